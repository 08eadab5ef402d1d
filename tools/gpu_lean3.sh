set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread -k "edge or benchmark_corpora or digests or empty or long_files or common or multi_line" > gpurun_out/gpu_tests_lean.txt 2>&1 || { tail -40 gpurun_out/gpu_tests_lean.txt; exit 1; }
tail -3 gpurun_out/gpu_tests_lean.txt
for v in b5 a6b5; do TAGPIPE_LIBRARY=imagecaptioner_amd/libtagpipe_hip_$v.so timeout -k 10 120 python3 tools/time_select.py c2 1000000 2>&1 | grep select_ms | sed 's/all.*total_ms/total_ms/' | cut -c1-200; done
timeout -k 10 120 python3 tools/time_select.py c2 1000000 2>&1 | grep select_ms | sed 's/all.*total_ms/total_ms/' | cut -c1-200
rm -rf gpurun_out/prof_lean
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_lean -o lean -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/prof_lean.log 2>&1 || { tail -20 gpurun_out/prof_lean.log; exit 1; }
python3 tools/rocpd_stats.py gpurun_out/prof_lean/lean_results.db | head -7
