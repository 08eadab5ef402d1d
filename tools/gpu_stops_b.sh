# Dynamic instruction counts and time of k_lean_b up to each section (diagnostic builds libtagpipe_hip_ls<k>.so, TP_LEAN_STOP=k).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for k in 1 2 3 4 5 6 full; do
  lib=imagecaptioner_amd/libtagpipe_hip_ls$k.so
  [ $k = full ] && lib=imagecaptioner_amd/libtagpipe_hip.so
  rm -rf gpurun_out/pmc_stop_$k
  TAGPIPE_LIBRARY=$lib timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM -d gpurun_out/pmc_stop_$k -o sq -- python3 tools/time_select.py c2 200000 > gpurun_out/pmc_stop_$k.log 2>&1 || exit 1
  python3 tools/pmc_summarise.py gpurun_out/pmc_stop_$k.json c2:200000 gpurun_out/pmc_stop_$k > /dev/null
  python3 - $k <<'PY'
import json,sys
d=json.load(open("gpurun_out/pmc_stop_%s.json" % sys.argv[1]))
print(sys.argv[1], {c.replace("SQ_",""): round(v["avg"]/200000,1) for c,v in sorted(d["kernels"]["k_lean_b"].items())})
PY
  TAGPIPE_LIBRARY=$lib timeout -k 10 120 python3 tools/time_select.py c2 1000000 2>&1 | grep -o "lean_back_ms[^,}]*"
  rm -rf gpurun_out/pmc_stop_$k
done
