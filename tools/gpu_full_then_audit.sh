# The whole GPU suite and the smoke check, then the audit-mode timing and the bench line of the default build.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1000 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.txt 2>&1 || { tail -40 gpurun_out/gpu_tests.txt; exit 1; }
tail -3 gpurun_out/gpu_tests.txt
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
timeout -k 10 300 python3 tools/audit_timing.py c2 200000 2>&1 | tail -2
timeout -k 10 300 python3 tools/audit_timing.py c3c 200000 2>&1 | tail -2
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/bench_quick.json
python3 tools/show_bench.py gpurun_out/bench_quick.json
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 --config c3c --files 200000 > gpurun_out/bench_c3c_quick.json
python3 tools/show_bench.py gpurun_out/bench_c3c_quick.json
