set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python3 -u -m pytest tests/wip_no_alloc.py -x -v --timeout 200 --timeout-method thread > gpurun_out/noalloc.txt 2>&1 || { tail -60 gpurun_out/noalloc.txt; exit 1; }
tail -8 gpurun_out/noalloc.txt
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err || { tail -20 gpurun_out/bench_now.err; exit 1; }
cat gpurun_out/bench_now.json
