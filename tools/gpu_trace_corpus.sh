# Kernel-trace summary of bench.py on one corpus ($CORPUS, $FILES files): gpurun_out/trace_$CORPUS.csv
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/trace_$CORPUS
rm -rf $O && mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 bench.py --config $CORPUS --files $FILES --steps 10 --warmup 2 --cpu-sample 0 --e2e-runs 0 > $O.log 2>&1 || { tail -20 $O.log; exit 1; }
find $O -name "*kernel_stats.csv" -exec cp {} gpurun_out/trace_$CORPUS.csv \;
rm -rf $O
cut -d, -f1-4 gpurun_out/trace_$CORPUS.csv | head -24
tail -1 $O.log | cut -c1-400
