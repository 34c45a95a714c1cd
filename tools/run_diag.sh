cd $GRAFT_REPO_ROOT
for p in "12=0" "12=30" "12=45" "12=60" "12=80"; do
echo "== $p"
ST_SKELETON_PARAMS=$p python tools/time_config3.py 2>&1 | grep -A1 "blocking=blocks" | grep -o "'outlier_removal[^}]*components': [0-9.]*\|[0-9.]* ms per cloud"
ST_SKELETON_PARAMS=$p python tools/bench_knn.py 2>&1 | tail -3
done
