set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gputests.log 2>&1
echo tests-done
timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/bench.json 2> gpurun_out/bench.err
echo bench-done
