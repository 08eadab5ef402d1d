set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -d gpurun_out/pmc_a -o sq -- python3 tools/time_select.py c2 200000 > gpurun_out/pmc_a.log 2>&1 || { tail -20 gpurun_out/pmc_a.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE -d gpurun_out/pmc_b -o sq -- python3 tools/time_select.py c2 200000 > gpurun_out/pmc_b.log 2>&1 || { tail -20 gpurun_out/pmc_b.log; exit 1; }
python3 tools/pmc_summarise.py gpurun_out/pmc_ab.json c2:200000 gpurun_out/pmc_a gpurun_out/pmc_b > /dev/null
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_ab.json"))
for k in ("k_lean_a","k_lean_b","k_emit"):
    if k in d["kernels"]:
        print(k, {c.replace("SQ_",""): round(v["avg"]/200000,1) for c,v in sorted(d["kernels"][k].items())})
PY
