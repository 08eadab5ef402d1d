set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/blocks.txt
for b in 1280 2560 5120 10240 20480 40960; do
TAGPIPE_EMIT_BLOCKS=$b timeout -k 10 200 python3 tools/prof_sections.py c2 1000000 > gpurun_out/sections_b.txt 2>&1
echo "$b $(grep 'no hist' gpurun_out/sections_b.txt | cut -c1-160)" >> gpurun_out/blocks.txt
done
echo done
