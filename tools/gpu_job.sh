set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gputests.log 2>&1
echo tests-done
for c in c2 c4 c3; do
timeout -k 10 200 python3 tools/prof_sections.py $c 1000000 > gpurun_out/sections_$c.txt 2>&1
echo $c-done
done
