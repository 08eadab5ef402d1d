set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/bench.json 2> gpurun_out/bench.err
echo bench-done
rm -rf gpurun_out/pmc_now
timeout -s KILL 100 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc_now -- python3 tools/prof_sections.py c2 200000 > gpurun_out/pmc_now.log 2>&1
echo pmc-done
