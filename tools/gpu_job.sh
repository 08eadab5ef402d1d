set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/bench.json 2> gpurun_out/bench.err
echo bench-done
TAGPIPE_LIBRARY=imagecaptioner_amd/libtagpipe_hip_prof.so timeout -k 10 200 python3 tools/prof_sections.py c2 200000 > gpurun_out/sections.txt 2>&1
echo sections-done
