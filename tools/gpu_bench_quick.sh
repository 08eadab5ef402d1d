# Parity tests (subset by $TESTS), then the bench line of the default build without the host baseline.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 300 --timeout-method thread -k "${TESTS:-edge or benchmark_corpora or digests or empty or long_files or common or multi_line or alloc}" > gpurun_out/gpu_tests_ab.txt 2>&1 || { tail -40 gpurun_out/gpu_tests_ab.txt; exit 1; }
tail -2 gpurun_out/gpu_tests_ab.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 ${BENCH_ARGS} > gpurun_out/bench_quick.json
python3 -c "
import json; d=json.load(open('gpurun_out/bench_quick.json')); print(round(d['value']/1e6,1),'M files/s', d['config']['kernel_ms'], 'frac', round(d['roofline']['frac'],4))"
