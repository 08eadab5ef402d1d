set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.txt 2>&1 || { tail -40 gpurun_out/gpu_tests.txt; exit 1; }
tail -3 gpurun_out/gpu_tests.txt
timeout -k 10 300 python3 tools/audit_timing.py c2 200000 > gpurun_out/audit_timing.txt 2>&1
timeout -k 10 200 python3 tools/audit_timing.py c5 100000 >> gpurun_out/audit_timing.txt 2>&1
timeout -k 10 200 python3 tools/audit_timing.py c3 200000 >> gpurun_out/audit_timing.txt 2>&1
grep -v amdgpu.ids gpurun_out/audit_timing.txt | cut -c1-230
