set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gputests.log 2>&1
echo tests-done
for v in "" _w4 _w6; do
TAGPIPE_LIBRARY=imagecaptioner_amd/libtagpipe_hip$v.so timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/bench$v.json 2> gpurun_out/bench.err
echo bench$v-done
done
