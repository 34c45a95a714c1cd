"""CPU study for the SSSP's depth ("next" item d): how many POLICY ITERATIONS reach the float32 fixed point?

    python tools/sim_policy_sssp.py [seed=0] [n_sub=120000] [quantum_m=0.0 ...]

The frontier SSSP needs one dependent step per hop (2400 levels on a tree with exact medial vectors).  The float32 fixed point
d[v] = min_u fl(d[u] + w(u, v)) is the greatest fixed point; the float32 sums down ANY spanning tree are upper bounds of it, so a
procedure that keeps tree sums and ends in a state that satisfies the equation has the oracle's bits.  Policy iteration: take a predecessor tree from approximate distances (no bit constraint on how they
were obtained), evaluate the tree's float32 sums exactly (a chain is one lane doing dependent adds from registers -- ~10 ns a
hop instead of ~5 us a level), let every vertex switch to its best neighbour under the evaluated sums, repeat until nobody
switches.  This script counts the iterations on the bench tree's ground-truth graph: the graph is built by the oracle
(oracle/skeleton_oracle.py), the approximate distances are float64 Dijkstra distances rounded DOWN to a multiple of `quantum`
(0 = unrounded; a coarse value stands for "distance of the vertex's cell in a coarse graph").  Nothing here is product code.
"""
import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import dijkstra

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from oracle import skeleton_oracle as so  # noqa: E402
from smart_tree_amd.synthetic import sample_tree_cloud  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_sub = int(sys.argv[2]) if len(sys.argv) > 2 else 120000
quanta = [float(a) for a in sys.argv[3:]] or [0.0, 0.01, 0.06, 0.25]

c = sample_tree_cloud(1_000_000, seed=seed)
rng = np.random.RandomState(seed)
sel = np.sort(rng.choice(1_000_000, n_sub, replace=False))  # stands for the inner voxels' representative points
xyz, mv = c["xyz"][sel], c["medial_vector"][sel]
medial = xyz + mv
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
print(f"graph: {n} vertices, {len(le)} directed entries; oracle SSSP {time.time() - t0:.2f} s")

# directed adjacency as the oracle relaxes it: entry (a, b, w) offers d[a] + w to b; make both directions explicit
src = np.concatenate((le[:, 0], le[:, 1]))
dst = np.concatenate((le[:, 1], le[:, 0]))
ww = np.concatenate((lw, lw)).astype(np.float32)
hops = np.zeros(n, np.int64)
order = np.argsort(dist_ref, kind="stable")
for v in order:  # hop depth of the oracle's predecessor tree = levels of the frontier SSSP
    if pred_ref[v] >= 0:
        hops[v] = hops[pred_ref[v]] + 1
print(f"oracle tree depth {hops.max()} hops; farthest vertex {dist_ref.max():.3f} m")
flat = int((np.float32(dist_ref[src] + ww) == dist_ref[src]).sum())
print(f"{flat} adjacency entries do not move their source's distance (plateaus): the fixed point is then not unique, but tree sums are upper\n"
      f"bounds of the oracle's distances and every fixed point is a lower bound, so a policy iteration that stops IS at the oracle's")

A = sp.csr_matrix((ww.astype(np.float64), (src, dst)), shape=(n, n))
d64 = dijkstra(A, directed=True, indices=root)


def best_neighbour(D):
    """For every vertex the smallest float32 offer fl(D[u] + w) over its in-edges and the u that makes it (smallest u on a tie)."""
    offer = (D[src] + ww).astype(np.float32)
    o = np.lexsort((src, offer, dst))
    first = np.ones(len(o), bool)
    first[1:] = dst[o][1:] != dst[o][:-1]
    best = np.full(n, np.inf, np.float32)
    arg = np.full(n, -1, np.int64)
    best[dst[o][first]] = offer[o][first]
    arg[dst[o][first]] = src[o][first]
    return best, arg


def evaluate(pred, wpred):
    """Exact float32 sums down the tree, level by level (what one lane per chain would do with dependent adds)."""
    D = np.full(n, np.inf, np.float32)
    D[root] = 0
    depth = np.full(n, -1, np.int64)
    depth[root] = 0
    children_order = np.argsort(pred, kind="stable")
    ps = pred[children_order]
    frontier = np.array([root])
    lev = 0
    while len(frontier):
        lo, hi = np.searchsorted(ps, frontier, "left"), np.searchsorted(ps, frontier, "right")
        cnt = hi - lo
        if cnt.sum() == 0:
            break
        idx = np.repeat(lo - np.cumsum(cnt) + cnt, cnt) + np.arange(cnt.sum())
        kids = children_order[idx]
        par = np.repeat(frontier, cnt)
        D[kids] = (D[par] + wpred[kids]).astype(np.float32)
        lev += 1
        depth[kids] = lev
        frontier = kids
    return D, lev


for q in quanta:
    approx = d64 if q == 0 else np.floor(d64 / q) * q
    # first tree: the neighbour with the smallest approximate offer among those strictly closer in the exact order (acyclic by
    # construction; a coarse approximation leaves many ties inside a cell, the exact order stands for "any acyclic choice there")
    rank = np.empty(n, np.int64)
    rank[np.argsort(d64, kind="stable")] = np.arange(n)
    ok = rank[src] < rank[dst]
    offer = np.where(ok, approx[src] + ww, np.inf)
    o = np.lexsort((src, offer, dst))
    first = np.ones(len(o), bool)
    first[1:] = dst[o][1:] != dst[o][:-1]
    pred = np.full(n, -1, np.int64)
    wpred = np.zeros(n, np.float32)
    pred[dst[o][first]] = src[o][first]
    wpred[dst[o][first]] = ww[o][first]
    pred[root] = -1
    its, log = 0, []
    while True:
        D, depth = evaluate(pred, wpred)
        assert np.isfinite(D).all(), "the tree lost a vertex"
        best, arg = best_neighbour(D)
        best[root] = 0
        sw = best < D
        log.append((int(sw.sum()), int((D != dist_ref).sum()), depth))
        if not sw.any():
            break
        # the offer that made `best`: weight = the edge (arg -> v)
        key = arg[sw] * n + np.nonzero(sw)[0]
        ekey = src * n + dst
        eo = np.argsort(ekey)
        pos = np.searchsorted(ekey[eo], key)
        pred[sw] = arg[sw]
        wpred[sw] = ww[eo][pos]
        its += 1
        if its > 200:
            break
    same = bool(np.array_equal(D, dist_ref))
    print(f"quantum {q:g} m: {its} policy iterations to the fixed point, distances bit-identical to the oracle: {same}")
    print("   per iteration (vertices that switch, vertices whose distance differs from the oracle's, tree depth): "
          + " ".join(f"({a},{b},{c})" for a, b, c in log[:12]) + (" ..." if len(log) > 12 else ""))


# ---------------------------------------------------------------------------------------------------------------------------------
# Variant: evaluate AND switch in one depth-ordered pass (Gauss-Seidel).  Level L of the current tree takes the best offer of ALL its
# neighbours -- values of this pass from the levels above, values of the previous pass from the rest (both are path sums, i.e. upper
# bounds) -- and keeps its predecessor unless another offer is strictly better.  Passes until nobody changes.
o_dst = np.argsort(dst, kind="stable")
in_src, in_w = src[o_dst], ww[o_dst]
in_off = np.searchsorted(dst[o_dst], np.arange(n + 1))


def levels_of(pred):
    children_order = np.argsort(pred, kind="stable")
    ps = pred[children_order]
    out, frontier = [], np.array([root])
    while len(frontier):
        lo, hi = np.searchsorted(ps, frontier, "left"), np.searchsorted(ps, frontier, "right")
        cnt = hi - lo
        if cnt.sum() == 0:
            break
        frontier = children_order[np.repeat(lo - np.cumsum(cnt) + cnt, cnt) + np.arange(cnt.sum())]
        out.append(frontier)
    return out


for q in quanta:
    approx = d64 if q == 0 else np.floor(d64 / q) * q
    rank = np.empty(n, np.int64)
    rank[np.argsort(d64, kind="stable")] = np.arange(n)
    ok = rank[src] < rank[dst]
    offer = np.where(ok, approx[src] + ww, np.inf)
    o = np.lexsort((src, offer, dst))
    first = np.ones(len(o), bool)
    first[1:] = dst[o][1:] != dst[o][:-1]
    pred = np.full(n, -1, np.int64)
    wpred = np.zeros(n, np.float32)
    pred[dst[o][first]] = src[o][first]
    wpred[dst[o][first]] = ww[o][first]
    pred[root] = -1
    D, _ = evaluate(pred, wpred)  # pass 0: plain evaluation of the first tree
    passes, log = 0, []
    while True:
        lv = levels_of(pred)
        assert sum(len(a) for a in lv) == n - 1, "the tree lost a vertex"
        changed = 0
        for kids in lv:
            cnt = in_off[kids + 1] - in_off[kids]
            idx = np.repeat(in_off[kids] - np.cumsum(cnt) + cnt, cnt) + np.arange(cnt.sum())
            owner = np.repeat(np.arange(len(kids)), cnt)
            offers = (D[in_src[idx]] + in_w[idx]).astype(np.float32)
            cur = (D[pred[kids]] + wpred[kids]).astype(np.float32)  # the tree's own offer under this pass's values
            oo = np.lexsort((in_src[idx], offers, owner))
            fst = np.ones(len(oo), bool)
            fst[1:] = owner[oo][1:] != owner[oo][:-1]
            best, arg, bw = offers[oo][fst], in_src[idx][oo][fst], in_w[idx][oo][fst]
            better = best < cur
            newd = np.where(better, best, cur)
            changed += int((newd != D[kids]).sum())
            D[kids] = newd
            pred[kids[better]] = arg[better]
            wpred[kids[better]] = bw[better]
        passes += 1
        log.append((changed, int((D != dist_ref).sum()), len(lv)))
        if changed == 0 or passes > 100:
            break
    print(f"Gauss-Seidel passes, quantum {q:g} m: {passes} passes (the last one changes nothing), bit-identical: {bool(np.array_equal(D, dist_ref))}")
    print("   per pass (distances that changed, distances that differ from the oracle's after it, levels): " + " ".join(f"({a},{b},{c})" for a, b, c in log[:14]))
