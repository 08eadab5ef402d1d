set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 python3 tools/scored_timing.py c2 200000 > gpurun_out/scored.txt 2>&1
cat gpurun_out/scored.txt
