#!/bin/bash
# GPU box: two runtime settings that change launch / wait latency, not results -- HIP_FORCE_DEV_KERNARG=1 (kernel arguments written
# to device memory: the packet processor does not fetch them over PCIe) and HSA_ENABLE_INTERRUPT=0 (host waits poll the completion
# signal instead of sleeping on an interrupt).  Alternating runs of the driver's command without the CPU leg.
cd $GRAFT_REPO_ROOT
run() {
  env "$@" ST_BENCH_MIN_UPTIME_S=12 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$*] %.1f M points/s, %.4f ms/step; upload-inclusive %.1f M; single cloud %.2f ms (min %.2f); host cores busy %.2f' % (d['value']/1e6, d['ms_per_step'], d['value_incl_host_upload']/1e6, d['single_cloud']['ms'], d['single_cloud']['ms_min'], d['config']['host_cpu_cores_busy_in_timed_region']))"
}
for rep in 1 2; do
  run X=0
  run HIP_FORCE_DEV_KERNARG=1
  run HSA_ENABLE_INTERRUPT=0
  run HIP_FORCE_DEV_KERNARG=1 HSA_ENABLE_INTERRUPT=0
done
