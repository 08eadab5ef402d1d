set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread -k "edge or corpora or digests or long or spell or multi_line or outside_ascii" > gpurun_out/gpu_quick.txt 2>&1 || { tail -40 gpurun_out/gpu_quick.txt; exit 1; }
tail -2 gpurun_out/gpu_quick.txt
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err || { tail -20 gpurun_out/bench_now.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/bench_now.json')); print(d['value']/1e6, d['ms_per_step'], d['config']['kernel_ms'], d['roofline']['frac'])"
