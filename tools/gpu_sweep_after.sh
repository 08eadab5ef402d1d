cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for c in c2:1000000 c3c:200000 c5:200000; do
for b in "4096 512"; do
  set -- $b
  TAGPIPE_AFTER_LEAN_BLOCKS=$1 TAGPIPE_AFTER_LEAN_BLOCKS2=$2 timeout -k 10 120 python3 tools/time_select.py ${c%%:*} ${c##*:} 2>&1 | grep -o "select_ms min [0-9.]*\|lean_front_ms[^}]*" | paste - - | sed "s/^/$c after-lean blocks $1 $2: /"
done; done
