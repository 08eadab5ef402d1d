# Bench lines of the other corpora (200k files per GPU), the no-vocabulary c2 variant excluded.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
for c in c1 c3 c3c c4 c5; do
  timeout -k 10 300 python3 bench.py --config $c --files 200000 --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/r02/bench_$c.json
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/r02/bench_$c.json')); print('$c', round(d['value']/1e6,1),'M files/s', {k:round(v,3) for k,v in d['config']['kernel_ms'].items()}, 'general', d['config']['files_in_general_kernel'])"
done
timeout -k 10 300 python3 tools/audit_timing.py c2 200000 > gpurun_out/r02/audit_timing.txt 2>&1 || true
tail -5 gpurun_out/r02/audit_timing.txt
timeout -k 10 300 python3 tools/scored_timing.py c2 200000 > gpurun_out/r02/scored_timing.txt 2>&1 || true
tail -4 gpurun_out/r02/scored_timing.txt
timeout -k 10 300 python3 tools/triage_timing.py c2 1000000 > gpurun_out/r02/triage_timing.txt 2>&1 || true
tail -4 gpurun_out/r02/triage_timing.txt
