set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python3 -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread -k "audit or review or side_files" > gpurun_out/gpu_tests_audit.txt 2>&1 || { tail -40 gpurun_out/gpu_tests_audit.txt; exit 1; }
tail -2 gpurun_out/gpu_tests_audit.txt
timeout -k 10 300 python3 tools/audit_timing.py c2 200000 2>&1 | tail -2
