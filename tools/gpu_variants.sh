cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in "$@"; do
  TAGPIPE_LIBRARY=imagecaptioner_amd/libtagpipe_hip_$v.so timeout -k 10 120 python3 tools/time_select.py c2 1000000 2>&1 | grep select_ms | sed 's/all.*total_ms/total_ms/' | cut -c1-300 || exit 1
done
timeout -k 10 120 python3 tools/time_select.py c2 1000000 2>&1 | grep select_ms | sed 's/all.*total_ms/total_ms/' | cut -c1-300
