set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
cut -c1-300 gpurun_out/bench_full.json
