set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.txt 2>&1 || { tail -40 gpurun_out/gpu_tests.txt; exit 1; }
tail -3 gpurun_out/gpu_tests.txt
timeout -k 10 400 python3 tools/variant_timing.py 200000 plain cr_inside_2pc > gpurun_out/variants2.txt 2>&1
cat gpurun_out/variants2.txt
for k in 1 2; do
timeout -k 10 300 python3 bench.py --cpu-sample 0 > gpurun_out/bench.json 2> gpurun_out/bench.err
python3 -c "
import json
d=json.load(open('gpurun_out/bench.json')); print(round(d['value']/1e6,1), d['config']['kernel_ms'])"
done
