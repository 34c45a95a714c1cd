#!/bin/bash
# GPU box: the frontier SSSP's launch shape on the GROUND-TRUTH graph (2400 levels deep), one cloud and a launch set of twenty.
# codes (csrc/skeleton.hip "Tuning of one call"): 6 levels per launch, 7 launches per read-back, 8 lanes per vertex, 10 workgroups, 13 vertices per workgroup and local level
cd $GRAFT_REPO_ROOT
for p in "" "6=8" "6=16" "6=8,7=64" "6=16,7=64" "8=16" "6=8,8=16" "10=512" "6=8,10=512" "13=256,6=8"; do
  python tools/probe_gt.py 20 "$p" 0 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
one=[v for k,v in d.items() if isinstance(v,dict) and 'one cloud per call' in k][0]
st=[v for k,v in d.items() if isinstance(v,dict) and 'one launch set' in k][0]
print('params [%s]: one cloud %.2f ms (skeleton kernels %.2f, %d SSSP launches); set of 20: %.2f ms (skeleton kernels %.3f per cloud)' % (d['params'], one['ms'], one['stage_ms_per_cloud']['skeleton_kernels'], one['sssp_rounds'], st['ms_per_set'], st['stage_ms_per_cloud']['skeleton_kernels']))"
done
