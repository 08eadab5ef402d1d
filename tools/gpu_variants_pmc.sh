cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in "$@"; do
  TAGPIPE_LIBRARY=imagecaptioner_amd/libtagpipe_hip_$v.so timeout -k 10 120 python3 tools/time_select.py c2 1000000 2>&1 | grep select_ms | cut -c1-90 || exit 1
  rm -rf gpurun_out/pmc_v
  TAGPIPE_LIBRARY=imagecaptioner_amd/libtagpipe_hip_$v.so timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_SMEM -d gpurun_out/pmc_v -o sq -- python3 tools/time_select.py c2 200000 > gpurun_out/pmc_v.log 2>&1 || exit 1
  python3 tools/pmc_summarise.py gpurun_out/pmc_v.json c2:200000 gpurun_out/pmc_v > /dev/null
  python3 - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_v.json"))
print("   k_lean", {c.replace("SQ_",""): round(v["avg"]/200000,1) for c,v in sorted(d["kernels"]["k_lean"].items())})
PY
done
