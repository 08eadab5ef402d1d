cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for b in 20480 40960 81920 250000; do
  TAGPIPE_LEAN_A_BLOCKS=$b TAGPIPE_LEAN_B_BLOCKS=$b timeout -k 10 120 python3 tools/time_select.py c2 1000000 2>&1 | grep -o "lean_front_ms[^}]*" | sed "s/^/blocks $b /"
done
