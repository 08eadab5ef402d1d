# The side chain (later stages of the front half's files beside the back half) off / on / as the library decides: bench lines per corpus.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in c5 c1 c4 c3 c2; do
  n=200000; [ $c = c2 ] && n=1000000
  for v in 0 1 auto; do
    if [ $v = auto ]; then unset TAGPIPE_SIDE_CHAIN; else export TAGPIPE_SIDE_CHAIN=$v; fi
    timeout -k 10 300 python3 bench.py --config $c --files $n --steps 20 --warmup 5 --cpu-sample 0 --e2e-runs 0 > gpurun_out/side_${c}_$v.json
    python3 -c "
import json; d=json.load(open('gpurun_out/side_${c}_$v.json')); print('$c side=$v', round(d['value']/1e6,1),'M files/s', round(d['ms_per_step'],3), {k: round(x,3) for k,x in d['config']['kernel_ms'].items()})"
  done
done
