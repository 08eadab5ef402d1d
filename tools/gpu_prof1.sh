set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_lean
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_lean -o lean -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/prof_lean.log 2>&1 || { tail -20 gpurun_out/prof_lean.log; exit 1; }
python3 tools/rocpd_stats.py gpurun_out/prof_lean/lean_results.db | head -8
