set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
echo start
timeout -k 10 900 python3 bench.py --files 10000000 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/bench_10m.json 2> gpurun_out/bench_10m.err
cut -c1-800 gpurun_out/bench_10m.json
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
