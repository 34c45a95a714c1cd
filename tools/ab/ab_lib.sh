# GPU box: alternate two builds of the library (_ab/old.so, _ab/new.so) under one command.  usage: ab_lib.sh '<cmd>' [rounds=2]
cd $GRAFT_REPO_ROOT
for round in $(seq 1 ${2:-2}); do
  for v in old new; do
    cp _ab/$v.so smart_tree_amd/libsmarttree_hip.so
    echo -n "$v: "; eval "$1" 2>&1 | grep -v amdgpu | tail -1
  done
done
cp _ab/new.so smart_tree_amd/libsmarttree_hip.so
