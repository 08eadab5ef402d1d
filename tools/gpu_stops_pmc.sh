cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for k in 1 2 3 5 6 4 7; do
  rm -rf gpurun_out/pmc_stop_$k
  TAGPIPE_LIBRARY=imagecaptioner_amd/libtagpipe_hip_ls$k.so timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM -d gpurun_out/pmc_stop_$k -o sq -- python3 tools/time_select.py c2 200000 > gpurun_out/pmc_stop_$k.log 2>&1 || exit 1
  python3 tools/pmc_summarise.py gpurun_out/pmc_stop_$k.json c2:200000 gpurun_out/pmc_stop_$k > /dev/null
  python3 - $k <<'PY'
import json,sys
d=json.load(open("gpurun_out/pmc_stop_%s.json" % sys.argv[1]))
print(sys.argv[1], {c.replace("SQ_",""): round(v["avg"]/200000,1) for c,v in sorted(d["kernels"]["k_lean"].items())})
PY
  grep select_ms gpurun_out/pmc_stop_$k.log | cut -c40-100
done
