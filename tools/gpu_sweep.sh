cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for b in 1280 2560 4096 8192 12288 20480; do
  echo "lean blocks $b"
  TAGPIPE_LEAN_BLOCKS=$b timeout -k 10 120 python3 tools/time_select.py c2 1000000 2>&1 | grep select_ms | cut -c1-120 || exit 1
done
