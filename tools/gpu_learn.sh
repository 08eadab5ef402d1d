set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -s --timeout 300 --timeout-method thread -k "learned" 2>&1 | tail -15
timeout -k 10 300 python3 bench.py --config c2n --files 1000000 --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/bench_c2n.json
python3 -c "
import json; d=json.load(open('gpurun_out/bench_c2n.json')); print('c2n', round(d['value']/1e6,1),'M files/s', d['config']['kernel_ms'], 'general', d['config']['files_in_general_kernel'])"
