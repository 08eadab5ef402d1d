set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_t
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE -d gpurun_out/pmc_t -o t -- python3 tools/time_select.py c2 200000 > gpurun_out/pmc_t.log 2>&1 || { tail -20 gpurun_out/pmc_t.log; exit 1; }
python3 tools/pmc_summarise.py gpurun_out/pmc_t.json c2:200000 gpurun_out/pmc_t > /dev/null
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_t.json"))
for k in ("k_lean_a","k_lean_b","k_emit","k_hist"):
    if k in d["kernels"]:
        print(k, {c: round(v["avg"]/200000,1) for c,v in sorted(d["kernels"][k].items())})
PY
