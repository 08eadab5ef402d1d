set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gputests3.log 2>&1
echo tests-done
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_v2 -o v2 -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/prof_v2.log 2>&1
echo stats-done
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/pmc_fetch.log 2>&1
echo fetch-done
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/pmc_write.log 2>&1
echo write-done
