# Round-2 evidence, part A: the bench line, the kernel-trace summary of the same command, the HBM traffic passes, the SQ and TCP passes.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02
rm -rf $O && mkdir -p $O
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
ls $O/trace/* | head -20
echo "trace done"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > $O/pmc_fetch.log 2>&1 || { tail -20 $O/pmc_fetch.log; exit 1; }
echo "fetch done"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > $O/pmc_write.log 2>&1 || { tail -20 $O/pmc_write.log; exit 1; }
echo "write done"
python3 tools/pmc_summarise.py $O/pmc_traffic.json c2:1000000 $O/pmc_fetch $O/pmc_write
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -d $O/pmc_sq_a -o a -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > $O/pmc_sq_a.log 2>&1 || { tail -20 $O/pmc_sq_a.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE -d $O/pmc_sq_b -o b -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > $O/pmc_sq_b.log 2>&1 || { tail -20 $O/pmc_sq_b.log; exit 1; }
python3 tools/pmc_summarise.py $O/pmc_sq.json c2:1000000 $O/pmc_sq_a $O/pmc_sq_b > /dev/null
echo "sq done"
timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE -d $O/pmc_tcp_a -o t -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 > $O/pmc_tcp.log 2>&1 || { tail -20 $O/pmc_tcp.log; exit 1; }
python3 tools/pmc_summarise.py $O/pmc_tcp.json c2:1000000 $O/pmc_tcp_a > /dev/null
rm -rf $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_sq_a $O/pmc_sq_b $O/pmc_tcp_a
cat $O/bench.json
