set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.txt 2>&1 || { tail -40 gpurun_out/gpu_tests.txt; exit 1; }
tail -3 gpurun_out/gpu_tests.txt
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1 || { tail -20 gpurun_out/smoke.txt; exit 1; }
tail -1 gpurun_out/smoke.txt
timeout -k 10 400 python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
cat gpurun_out/bench_final.json
rm -rf gpurun_out/prof_final
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_final -o final --output-format csv -- python3 bench.py --cpu-sample 0 > gpurun_out/prof_final.log 2>&1
find gpurun_out/prof_final -name "*kernel_stats*" | head
for c in c1 c3 c4 c5; do timeout -k 10 300 python3 bench.py --config $c --files 200000 --cpu-sample 0 > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err; done
python3 -c "
import json
for c in ('c1','c3','c4','c5'):
    d=json.load(open('gpurun_out/bench_%s.json'%c)); print(c, round(d['value']/1e6,1), d['config']['kernel_ms'], d['config']['files_in_general_kernel'])"
