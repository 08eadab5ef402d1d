set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/fpw.txt
for f in 64 16 4 1; do
echo "fpw $f" >> gpurun_out/fpw.txt
TAGPIPE_GENERAL_FPW=$f timeout -k 10 300 python3 tools/variant_timing.py 200000 cr_inside_2pc >> gpurun_out/fpw.txt 2>&1
done
echo "default" >> gpurun_out/fpw.txt
timeout -k 10 300 python3 tools/variant_timing.py 200000 >> gpurun_out/fpw.txt 2>&1
cat gpurun_out/fpw.txt
