# The GPU test suite (or the subset -k "$TESTS"), output kept in gpurun_out/gpu_tests.txt.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ -n "$TESTS" ]; then
  timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "$TESTS" > gpurun_out/gpu_tests.txt 2>&1 || { tail -60 gpurun_out/gpu_tests.txt; exit 1; }
else
  timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.txt 2>&1 || { tail -60 gpurun_out/gpu_tests.txt; exit 1; }
fi
tail -3 gpurun_out/gpu_tests.txt
