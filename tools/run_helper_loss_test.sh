#!/bin/bash
# Build HERE (no GPU needed) a TEST copy of the library whose helper workgroups give up after <life> ticks of the 100 MHz clock and whose
# component workgroups wait <wait> ticks for a job's answers, then run the full-size parity tests against it ON THE GPU BOX:
#   bash tools/run_helper_loss_test.sh build 2000 20000      (here)      ->  _helper_loss/libsmarttree_hip.so
#   gpurun -- 'bash tools/run_helper_loss_test.sh run'        (GPU box)
# Results must not change: a lost helper's share is claimed by the component's own workgroup.
cd "$(dirname "$0")/.."
if [ "$1" = "build" ]; then
  mkdir -p _helper_loss
  objs=""
  for f in smart_tree_amd/csrc/*.hip; do
    o=_helper_loss/$(basename ${f%.hip}).o
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DSK_HELP_LIFETIME=${2:-2000}ll -DSK_HELP_TIMEOUT=${3:-20000}ll -I include -c $f -o $o || exit 1
    objs="$objs $o"
  done
  hipcc --offload-arch=gfx950 -shared -fPIC -o _helper_loss/libsmarttree_hip.so $objs && rm -f _helper_loss/*.o
  ls -la _helper_loss/
else
  cp smart_tree_amd/libsmarttree_hip.so /tmp/keep.so
  cp _helper_loss/libsmarttree_hip.so smart_tree_amd/libsmarttree_hip.so
  touch smart_tree_amd/libsmarttree_hip.so
  timeout 1500 python -m pytest tests/test_full_size.py tests/test_batch.py -q -m gpu -x -k "oracle or batch or stagewise or canopy" 2>&1 | grep -E "passed|failed|error" | tail -3
  python tools/probe_sssp_batch.py 20 "" 2>&1 | grep params
  cp /tmp/keep.so smart_tree_amd/libsmarttree_hip.so
fi
