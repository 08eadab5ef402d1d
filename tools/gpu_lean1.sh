set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread -k "device_api" > gpurun_out/gpu_tests_lean2.txt 2>&1 || { tail -40 gpurun_out/gpu_tests_lean2.txt; exit 1; }
tail -3 gpurun_out/gpu_tests_lean2.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 > gpurun_out/bench_lean.json 2> gpurun_out/bench_lean.err || { tail -20 gpurun_out/bench_lean.err; exit 1; }
cat gpurun_out/bench_lean.json
TAGPIPE_NO_LEAN=1 timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 > gpurun_out/bench_nolean.json 2> gpurun_out/bench_nolean.err || { tail -20 gpurun_out/bench_nolean.err; exit 1; }
cat gpurun_out/bench_nolean.json
