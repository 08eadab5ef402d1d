set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 200 python3 tools/triage_timing.py c2 1000000 > gpurun_out/triage.txt 2>&1
cat gpurun_out/triage.txt
timeout -k 10 300 python3 bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json
