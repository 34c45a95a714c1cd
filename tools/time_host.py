"""Developer aid: is the multi-stream bench host-bound?  For S clouds in flight prints wall ms per cloud and the
process CPU time (all threads) per cloud; then a cProfile of the single-stream loop."""
import sys, time, threading, cProfile, pstats
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import torch
import bench
from smart_tree_amd.data_types.cloud import Cloud
from smart_tree_amd.synthetic import sample_tree_cloud
dev = torch.device("cuda:0")
c = sample_tree_cloud(1_000_000, seed=0)
cloud = Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev))
SMAX = 8
pipes = [bench.build_pipeline(dev) for _ in range(SMAX)]
streams = [torch.cuda.Stream(device=dev) for _ in range(SMAX)]

def run(S, total):
    state = {"next": 0}; lock = threading.Lock()
    def worker(w):
        with torch.cuda.stream(streams[w]):
            while True:
                with lock:
                    i = state["next"]; state["next"] += 1
                if i >= total: break
                pipes[w].process_cloud(cloud=cloud)
            streams[w].synchronize()
    ts = [threading.Thread(target=worker, args=(w,)) for w in range(S)]
    [t.start() for t in ts]; [t.join() for t in ts]

run(SMAX, 2 * SMAX); torch.cuda.synchronize()
for S in (1, 2, 4, 8):
    n = 24
    w0, c0 = time.perf_counter(), time.process_time()
    run(S, n); torch.cuda.synchronize()
    w1, c1 = time.perf_counter(), time.process_time()
    print(f"S={S}: wall {1e3*(w1-w0)/n:.2f} ms/cloud, process cpu {1e3*(c1-c0)/n:.2f} ms/cloud", flush=True)
pr = cProfile.Profile(); pr.enable()
with torch.cuda.stream(streams[0]):
    for _ in range(10): pipes[0].process_cloud(cloud=cloud)
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumulative").print_stats(45)
