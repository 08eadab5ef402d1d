# Kernel times of the audit-mode run (tools/audit_timing.py) on one corpus.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/trace_audit
rm -rf $O && mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 tools/audit_timing.py ${CORPUS:-c2} ${FILES:-200000} > $O/trace.log 2>&1 || { tail -20 $O/trace.log; exit 1; }
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/trace
cut -d, -f1-4 $O/kernel_stats.csv | grep "tp::" | cut -c1-120
