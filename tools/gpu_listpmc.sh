set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -s KILL 120 rocprofv3 --list-avail > gpurun_out/pmc_avail.txt 2>&1 || true
grep -c . gpurun_out/pmc_avail.txt
grep -o "TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|TD_[A-Z0-9_]*" gpurun_out/pmc_avail.txt | sort -u | tr '\n' ' ' | cut -c1-6000
