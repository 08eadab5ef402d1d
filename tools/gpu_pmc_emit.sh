cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/pmc_emit
rm -rf $O && mkdir -p $O
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -d $O/a -o a -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > $O/a.log 2>&1 || { tail -20 $O/a.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR -d $O/b -o b -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > $O/b.log 2>&1 || { tail -20 $O/b.log; exit 1; }
python3 tools/pmc_summarise.py $O/sq.json c2:1000000 $O/a $O/b > /dev/null
python3 -c "
import json
d=json.load(open('$O/sq.json'))
for k in d['kernels']:
    if 'emit' in k or 'hist' in k: print(k, {n.replace('SQ_',''):round(v['avg']/1e6,1) for n,v in d['kernels'][k].items()})
"
rm -rf $O/a $O/b
