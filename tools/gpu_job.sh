set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 python3 -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "audit or review" --timeout 120 --timeout-method thread > gpurun_out/gpu_audit.txt 2>&1 || { tail -60 gpurun_out/gpu_audit.txt; exit 1; }
tail -3 gpurun_out/gpu_audit.txt
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.txt 2>&1 || { tail -40 gpurun_out/gpu_tests.txt; exit 1; }
tail -3 gpurun_out/gpu_tests.txt
