# Round-3 evidence in one job: the kernel-trace summary and the HBM traffic / SQ passes of the bench command, then the bench line
# itself (it reads profiles/pmc_traffic.json, so the traffic passes come first), the other corpora, audit mode, the tagger-reply
# and triage entry points.  Everything lands in gpurun_out/r03/.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03
rm -rf $O && mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --e2e-runs 0 > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
echo "trace done"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e-runs 0 > $O/pmc_fetch.log 2>&1 || { tail -20 $O/pmc_fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e-runs 0 > $O/pmc_write.log 2>&1 || { tail -20 $O/pmc_write.log; exit 1; }
python3 tools/pmc_summarise.py $O/pmc_traffic.json c2:1000000 $O/pmc_fetch $O/pmc_write > /dev/null
cp $O/pmc_traffic.json profiles/pmc_traffic.json
echo "traffic done"
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -d $O/pmc_sq_a -o a -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --e2e-runs 0 > $O/pmc_sq_a.log 2>&1 || { tail -20 $O/pmc_sq_a.log; exit 1; }
python3 tools/pmc_summarise.py $O/pmc_sq.json c2:1000000 $O/pmc_sq_a > /dev/null
echo "sq done"
rm -rf $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_sq_a
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
echo "bench done"
for c in c1 c3 c3c c4 c5; do
  timeout -k 10 300 python3 bench.py --config $c --files 200000 --steps 20 --warmup 5 --cpu-sample 0 --e2e-runs 0 > $O/bench_$c.json 2> $O/bench_$c.err || { tail -5 $O/bench_$c.err; exit 1; }
done
timeout -k 10 300 python3 bench.py --config c2n --steps 10 --warmup 4 --cpu-sample 0 --e2e-runs 0 > $O/bench_c2n.json 2> $O/bench_c2n.err || { tail -5 $O/bench_c2n.err; exit 1; }
for a in "c2 200000" "c3c 200000" "c5 100000"; do timeout -k 10 300 python3 tools/audit_timing.py $a >> $O/audit_timing.txt 2>> $O/audit_timing.err || { tail -5 $O/audit_timing.err; exit 1; }; done
timeout -k 10 300 python3 tools/scored_timing.py c2 200000 > $O/scored_timing.txt 2> $O/scored_timing.err || { tail -5 $O/scored_timing.err; exit 1; }
timeout -k 10 300 python3 tools/triage_timing.py c2 1000000 > $O/triage_timing.txt 2> $O/triage_timing.err || { tail -5 $O/triage_timing.err; exit 1; }
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03/bench*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], round(d["value"] / 1e6, 1), "M files/s", {k: round(v, 3) for k, v in d["config"]["kernel_ms"].items()}, "frac", round(d["roofline"]["frac"], 4),
          "traffic", d["roofline"].get("traffic"), "e2e", d["config"].get("end_to_end_files_per_s"), "general", d["config"]["files_in_general_kernel"])
PY
tail -2 $O/scored_timing.txt; tail -2 $O/triage_timing.txt
