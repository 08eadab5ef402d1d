set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/cmp.txt
for c in c1 c3 c5; do
for lib in libtagpipe_hip_old.so libtagpipe_hip.so; do
TAGPIPE_LIBRARY=imagecaptioner_amd/$lib timeout -k 10 200 python3 bench.py --config $c --files 200000 --cpu-sample 0 > gpurun_out/b.json 2> gpurun_out/b.err
python3 -c "
import json
d=json.load(open('gpurun_out/b.json'))
print('$c $lib', round(d['value']/1e6,1), d['config']['kernel_ms'])" >> gpurun_out/cmp.txt
done
done
cat gpurun_out/cmp.txt
