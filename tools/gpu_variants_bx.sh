# Bench of kernel variants built beforehand as libtagpipe_hip_<name>.so ($VARIANTS: names; "main" = the default build).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in $VARIANTS; do
  if [ "$v" = main ]; then unset TAGPIPE_LIBRARY; else export TAGPIPE_LIBRARY=$GRAFT_REPO_ROOT/imagecaptioner_amd/libtagpipe_hip_$v.so; fi
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --cpu-sample 0 --e2e-runs 0 ${BENCH_ARGS} > gpurun_out/bench_var_$v.json 2> gpurun_out/bench_var_$v.err || { echo "$v failed"; tail -3 gpurun_out/bench_var_$v.err; continue; }
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/bench_var_$v.json')); k=d['config']['kernel_ms']; print('$v', round(d['value']/1e6,1),'M files/s', {a: round(b,3) for a,b in k.items()}, 'reruns', d['config'].get('fused_reruns'))"
done
