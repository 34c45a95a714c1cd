#!/bin/bash
# Developer aid: is the multi-stream bench bound by the host interpreter?  P processes x S streams on ONE GPU.
P=$1; S=$2
for i in $(seq 1 $P); do
  timeout -s KILL 200 python bench.py --steps 48 --warmup 2 --no-cpu-baseline --streams $S 2>/dev/null | tail -1 > /tmp/mp_$i.json &
done
wait
python - <<PY
import json
tot=0
for i in range(1,$P+1):
    d=json.load(open(f'/tmp/mp_{i}.json')); tot+=d['value']; print('proc',i,round(d['value']/1e6,1),'M pts/s')
print('processes $P x streams $S: total', round(tot/1e6,1), 'M pts/s')
PY
