set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/cmp.txt
for lib in libtagpipe_hip_noshort.so libtagpipe_hip.so; do
for e in 0 3; do
echo $lib >> gpurun_out/cmp.txt
TAGPIPE_LIBRARY=imagecaptioner_amd/$lib timeout -k 10 200 python3 tools/unknown_timing.py 200000 $e >> gpurun_out/cmp.txt 2>&1
done
done
cat gpurun_out/cmp.txt
