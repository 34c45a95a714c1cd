#!/bin/bash
# Developer aid: bench throughput vs clouds in flight and HIP hardware queues.  usage: sweep_streams.sh "4 8" "4 8"
for q in $1; do for s in $2; do
  GPU_MAX_HW_QUEUES=$q timeout -s KILL 160 python bench.py --steps 48 --warmup 2 --no-cpu-baseline --streams $s 2>/dev/null | tail -1 > /tmp/line.json
  python -c "
import json
d=json.load(open('/tmp/line.json')); print('queues $q streams', d['config']['clouds_in_flight_per_gpu'], round(d['value']/1e6,1), 'M pts/s', round(d['ms_per_step'],2), 'ms')"
done; done
