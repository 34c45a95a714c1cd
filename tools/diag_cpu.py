import os, sys, time
sys.path.insert(0, '.')
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), flush=True)
try:
    print('cgroup cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip(), flush=True)
except Exception as e: print('no cgroup', e)
import torch
print('torch threads default', torch.get_num_threads(), flush=True)
from oracle import pipeline_oracle as po, unet_oracle as uo
from smart_tree_amd.synthetic import sample_tree_cloud
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if thr: torch.set_num_threads(thr)
t=time.time(); c = sample_tree_cloud(n, 0); print('gen', time.time()-t, flush=True)
w = uo.load_weights('smart_tree_amd/model/weights/noble-elevator-58.npz')
tm = {}
t=time.time(); trees = po.process_cloud(c['xyz'], c['rgb'], w, 0.02, timings=tm); print('oracle total', time.time()-t, tm, 'threads', torch.get_num_threads(), flush=True)
