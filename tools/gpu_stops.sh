cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for k in 1 2 3 5 6 4 7; do
  TAGPIPE_LIBRARY=imagecaptioner_amd/libtagpipe_hip_ls$k.so timeout -k 10 120 python3 tools/time_select.py c2 1000000 2>&1 | grep select_ms | cut -c1-100 || exit 1
done
TAGPIPE_LIBRARY=imagecaptioner_amd/libtagpipe_hip_w5.so timeout -k 10 120 python3 tools/time_select.py c2 1000000 2>&1 | grep select_ms | cut -c1-100
