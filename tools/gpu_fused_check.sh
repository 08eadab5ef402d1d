# Parity subset with the fused back half switched on, then the bench line of both settings.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
TAGPIPE_FUSED=1 timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 300 --timeout-method thread -k "alloc or device_api or full_size or two_ranks" > gpurun_out/gpu_tests_fused.txt 2>&1 || { tail -40 gpurun_out/gpu_tests_fused.txt; exit 1; }
tail -1 gpurun_out/gpu_tests_fused.txt
VARIANTS=main bash tools/gpu_variants_bx.sh
