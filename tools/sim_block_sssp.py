"""CPU study (round 6): SSSP by BLOCKS RELAXED TO CONVERGENCE -- how many chip-wide rounds, block runs and inner levels?

    python tools/sim_block_sssp.py [gt|net] [seed=0] [block=64 ...]

The frontier SSSP costs one chip-wide step (~5 us) per hop level: 2400 levels on the ground-truth graph, 360 on the graph the
shipped checkpoint's output gives.  Here the vertices are ordered along a fine Morton curve and cut into blocks of B
consecutive vertices; a round relaxes every DIRTY block to convergence on its own (inside a wavefront: registers / LDS, no
global round trip per hop), reading the distances of outside neighbours as they were when the round started; a vertex that
improved marks the blocks of its outside neighbours dirty for the next round.  The result is the float32 fixed point whatever
the schedule (oracle/skeleton_oracle.c so_sssp: least fixed point reached from above); this script only COUNTS rounds, block
runs and the inner levels of the deepest block of a round.  Nothing here is product code.
"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import skeleton_oracle as so  # noqa: E402
from smart_tree_amd.synthetic import sample_tree_cloud  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "gt"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
blocks = [int(a) for a in sys.argv[3:]] or [64, 128, 256]

c = sample_tree_cloud(1_000_000, seed=seed)
if which == "gt":
    rng = np.random.RandomState(seed)
    sel = np.sort(rng.choice(1_000_000, 120000, replace=False))  # stands for the inner voxels' representative points
    xyz, mv = c["xyz"][sel], c["medial_vector"][sel]
else:
    from oracle import pipeline_oracle as po
    from oracle import unet_oracle as uo
    w = uo.load_weights(ROOT / "smart_tree_amd" / "model" / "weights" / "noble-elevator-58.npz")
    lc = po.labelled_cloud(c["xyz"], c["rgb"], w, 0.02)
    keep = lc["class_l"].reshape(-1) == 0
    xyz, mv = lc["xyz"][keep], lc["medial_vector"][keep]
medial = (xyz + mv).astype(np.float32)
radius = np.sqrt(((mv * mv)[:, 0] + (mv * mv)[:, 1]) + (mv * mv)[:, 2]).astype(np.float32)
keep = so.outlier_removal(medial, radius, 8)
xyz, medial, radius = xyz[keep], medial[keep], radius[keep]
edges, w = so.nn_graph(medial, np.maximum(radius, np.float32(0.02)), 16)
labels = so.cc_labels(len(xyz), edges)
roots, counts = np.unique(labels, return_counts=True)
ids = np.nonzero(labels == roots[np.argmax(counts)])[0]
inside = np.isin(edges[:, 0], ids)
le, lw = np.searchsorted(ids, edges[inside]), w[inside]
n = len(ids)
root = int(np.argmin(xyz[ids, 1]))
t0 = time.time()
dist_ref, pred_ref = so.sssp(n, le, lw, root)
src = np.concatenate((le[:, 0], le[:, 1]))
dst = np.concatenate((le[:, 1], le[:, 0]))
ww = np.concatenate((lw, lw)).astype(np.float32)
hops = np.zeros(n, np.int64)
for v in np.argsort(dist_ref, kind="stable"):
    if pred_ref[v] >= 0:
        hops[v] = hops[pred_ref[v]] + 1
print(f"{which} graph seed {seed}: {n} vertices, {len(src)} directed entries, oracle tree depth {hops.max()} hops, farthest {dist_ref.max():.2f} m")

P = medial[ids].astype(np.float64)


def morton_order(P, bits=20):
    lo, hi = P.min(0), P.max(0)
    q = ((P - lo) / (hi - lo).max() * ((1 << bits) - 1)).astype(np.uint64)
    key = np.zeros(len(P), dtype=object)
    code = [0] * len(P)
    qx, qy, qz = q[:, 0].tolist(), q[:, 1].tolist(), q[:, 2].tolist()
    for i in range(len(P)):
        x, y, z, k = qx[i], qy[i], qz[i], 0
        for b in range(bits - 1, -1, -1):
            k = (k << 3) | (((x >> b) & 1) << 2) | (((y >> b) & 1) << 1) | ((z >> b) & 1)
        code[i] = k
    return np.array(sorted(range(len(P)), key=code.__getitem__))


t0 = time.time()
import os
BITS = int(os.environ.get("MORTON_BITS", "20"))
order_m = morton_order(P, BITS)
print("morton bits per axis", BITS)
print(f"morton order {time.time() - t0:.1f} s")


def run(order, B, label):
    pos = np.empty(n, np.int64)
    pos[order] = np.arange(n)
    s, d_ = pos[src], pos[dst]
    bs, bd = s // B, d_ // B
    nb = (n + B - 1) // B
    inner_e = bs == bd
    D = np.full(n, np.inf, np.float32)
    D[pos[root]] = 0.0
    dirty = np.zeros(nb, bool)
    # the root "improved": its neighbours' blocks (and its own) are dirty
    dirty[bd[s == pos[root]]] = True
    dirty[pos[root] // B] = True
    rounds = runs = 0
    lvl_sum = 0  # sum over rounds of the deepest block's inner levels
    lvl_runs = 0  # sum over block runs of their inner levels
    max_dirty = 0
    while dirty.any():
        rounds += 1
        act = dirty.copy()
        dirty[:] = False
        runs += int(act.sum())
        max_dirty = max(max_dirty, int(act.sum()))
        e_act = act[bd]
        es, ed, ew, ein = s[e_act], d_[e_act], ww[e_act], inner_e[e_act]
        D0 = D.copy()
        ext_offer = np.full(n, np.inf, np.float32)
        np.minimum.at(ext_offer, ed[~ein], (D0[es[~ein]] + ew[~ein]).astype(np.float32))
        D = np.minimum(D, ext_offer)
        lv = np.zeros(nb, np.int64)
        ies, ied, iew = es[ein], ed[ein], ew[ein]
        live = np.ones(len(ies), bool)
        it = 0
        while True:
            it += 1
            offer = (D[ies] + iew).astype(np.float32)
            better = offer < D[ied]
            if not better.any():
                break
            new = D.copy()
            np.minimum.at(new, ied[better], offer[better])
            ch = new < D
            lv[np.unique(np.nonzero(ch)[0] // B)] = it
            D = new
        lvl_sum += int(lv.max()) + 1
        lvl_runs += int((lv[act] + 1).sum())
        changed = D < D0
        e_out = changed[s] & ~inner_e
        # only where the new value actually improves the neighbour (look before marking)
        e_out &= (D[s] + ww).astype(np.float32) < D[d_]
        dirty[bd[e_out]] = True
    ok = np.array_equal(D[pos], dist_ref)
    print(f"{label:>10} B={B:4d}: rounds {rounds:5d}, block runs {runs:7d} ({runs / nb:.1f} per block), max dirty {max_dirty}, "
          f"sum of deepest inner levels {lvl_sum}, mean levels per run {lvl_runs / max(runs, 1):.1f}; fixed point == oracle: {ok}")


for B in blocks:
    run(order_m, B, "morton")
for B in ([] if os.environ.get("MORTON_ONLY") else blocks[:1]):
    run(np.arange(n), B, "given")
    run(np.argsort(P[:, 1], kind="stable"), B, "y-sorted")
