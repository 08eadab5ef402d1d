set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_r01 gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout -k 10 300 python3 bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
echo bench-done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -o r01 -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/prof_r01.log 2>&1
echo stats-done
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/pmc_fetch.log 2>&1
echo fetch-done
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/pmc_write.log 2>&1
echo write-done
for c in c1 c3 c4 c5; do
timeout -k 10 300 python3 bench.py --config $c --files 200000 --steps 5 --warmup 1 --cpu-sample 0 > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
echo $c-done
done
