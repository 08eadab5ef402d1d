set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/cmp.txt
for c in c2 c1 c3 c5; do
TAGPIPE_DEBUG_TABLES=1 timeout -k 10 200 python3 bench.py --config $c --files 500000 --cpu-sample 0 > gpurun_out/b.json 2> gpurun_out/b.err
grep "tagpipe tables" gpurun_out/b.err | head -2 >> gpurun_out/cmp.txt
python3 -c "
import json
d=json.load(open('gpurun_out/b.json'))
print('$c', round(d['value']/1e6,1), d['config']['kernel_ms'])" >> gpurun_out/cmp.txt
done
cat gpurun_out/cmp.txt
