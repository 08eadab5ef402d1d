set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 120 python3 tools/time_select.py c2 1000000 2>&1 | grep select_ms | sed 's/all.*total_ms/total_ms/' | cut -c1-260
rm -rf gpurun_out/pmc_sq_r2 gpurun_out/pmc_sq_r2b
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -d gpurun_out/pmc_sq_r2 -o sq -- python3 tools/time_select.py c2 200000 > gpurun_out/pmc_sq_r2.log 2>&1 || { tail -20 gpurun_out/pmc_sq_r2.log; exit 1; }
python3 tools/pmc_summarise.py gpurun_out/pmc_sq_r2.json c2:200000 gpurun_out/pmc_sq_r2 > /dev/null
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_sq_r2.json"))
for k in ("k_lean","k_fast","k_emit"):
    if k in d["kernels"]:
        print(k, {c.replace("SQ_",""): round(v["avg"]/200000,1) for c,v in sorted(d["kernels"][k].items())})
PY
