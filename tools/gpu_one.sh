set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 python3 -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread -k "$1" > gpurun_out/gpu_one.txt 2>&1 || { tail -60 gpurun_out/gpu_one.txt; exit 1; }
tail -3 gpurun_out/gpu_one.txt
