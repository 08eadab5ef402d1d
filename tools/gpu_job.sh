set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
echo bench-done
for c in c1 c3 c4 c5; do
timeout -k 10 300 python3 bench.py --config $c --files 200000 --steps 5 --warmup 1 --cpu-sample 0 > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
echo $c-done
done
timeout -k 10 200 python3 tools/triage_timing.py c2 1000000 > gpurun_out/triage.txt 2>&1
cat gpurun_out/triage.txt
