set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1000 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.txt 2>&1 || { tail -40 gpurun_out/gpu_tests.txt; exit 1; }
tail -3 gpurun_out/gpu_tests.txt
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
