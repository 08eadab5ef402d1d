# Parity tests on the default build, then select-stage timing of the default build and of the variants named in $VARIANTS.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 300 --timeout-method thread -k "${TESTS:-edge or benchmark_corpora or digests or empty or long_files or common or multi_line or alloc}" > gpurun_out/gpu_tests_ab.txt 2>&1 || { tail -40 gpurun_out/gpu_tests_ab.txt; exit 1; }
tail -2 gpurun_out/gpu_tests_ab.txt
timeout -k 10 120 python3 tools/time_select.py c2 1000000 2>&1 | grep -o "select_ms min.*" | cut -c1-400
for v in $VARIANTS; do
  TAGPIPE_LIBRARY=imagecaptioner_amd/libtagpipe_hip_$v.so timeout -k 10 120 python3 tools/time_select.py c2 1000000 2>&1 | grep -o "libtagpipe_hip_$v.*select_ms min.*" | cut -c1-400
done
