"""Skeleton stage of the HIP path vs oracle/skeleton_oracle.{c,py}: everything bit-exact."""
import numpy as np
import pytest
import torch

from oracle import skeleton_oracle as so
from oracle import voxel_oracle as vo
from smart_tree_amd.data_types.cloud import Cloud
from smart_tree_amd.data_types.graph import Graph
from smart_tree_amd.skeleton import graph as G
from smart_tree_amd.skeleton.filter import outlier_removal
from smart_tree_amd.skeleton.skeletonize import (STAGE_SAMPLE, STAGE_SSSP, STAGE_TREE_DISTANCE, DeviceSkeleton, Skeletonizer,
                                                 run_components)
from smart_tree_amd.synthetic import sample_tree_cloud


def _tree(n=2500, seed=4, voxel=0.04, exact_medial=False):
    c = sample_tree_cloud(n, seed=seed, scale=0.6, max_depth=4)
    xyz = vo.centre_cloud(c["xyz"])
    vx = vo.voxelize_cloud(xyz, c["rgb"], voxel, block_size=2.0, buffer_size=0.2)
    keep = vx["mask"]
    pts = vx["feats"][keep, :3]
    mv = c["medial_vector"][vx["point"][keep]]
    if exact_medial:  # rings collapse onto (nearly) the same axis point: duplicates, plateaus, ties
        mv = np.round(mv * 50) / 50
        pts = np.round(pts * 50) / 50
    return pts.astype(np.float32), mv.astype(np.float32)


@pytest.mark.parametrize("K", [1, 8, 16, 4, 11, 32, 20, 64, 50, 40])  # (50 / 40: the defaults of the reference's knn / nn_graph, graph.py:12,36)
def test_knn_matches_oracle(backend, K):
    rng = np.random.RandomState(K)
    dst = rng.uniform(0, 1, (1500, 3)).astype(np.float32)
    dst[100:140] = dst[50]  # exact duplicates: index tie-break
    src = np.concatenate([dst[:300], rng.uniform(-0.2, 1.2, (150, 3)).astype(np.float32)])
    r = 0.12
    ref_idx, ref_d = so.knn(src, dst, K, r)
    idx, d, _ = G.knn(torch.from_numpy(src).to(backend), torch.from_numpy(dst).to(backend), K=K, r=r)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref_idx)
    np.testing.assert_array_equal(d.cpu().numpy(), ref_d)  # NaN == NaN under assert_array_equal


@pytest.mark.parametrize("cell", [0.0, 0.03], ids=["cell=r", "cell=3cm"])
@pytest.mark.parametrize("K", [1, 16])
def test_knn_dense_neighbourhoods(backend, K, cell):
    """Hundreds of points inside the radius: the key list in LDS is cut back to its K smallest several times and later
    candidates are dropped against the K-th key so far -- same rows as the oracle's exhaustive search, duplicates included.
    With 3 cm cells the search box is ~20 cells wide: the 3 x 3 x 3 cells around the query go first, and the rest is skipped
    when the K-th key is already closer than a cell edge (the clustered queries) or searched without the cube (the others)."""
    rng = np.random.RandomState(7 + K)
    dst = rng.uniform(0, 1, (1200, 3)).astype(np.float32)
    dst[200:330] = dst[10]  # 131 copies of one point: more equal keys than K, cut in the middle of a tie
    dst[400:420, 2] = dst[400, 2]
    dst[600:700] = dst[600] + rng.uniform(-0.004, 0.004, (100, 3)).astype(np.float32)  # a tight cluster: early exit
    src = np.concatenate([dst[:260], dst[590:620]])
    bound = rng.uniform(0.2, 0.6, len(src)).astype(np.float32)
    ref_idx, ref_d = so.knn(src, dst, K, 0.5)
    far = ref_d > bound[:, None]  # graph.py:38-40: neighbours beyond the query's own bound are dropped afterwards
    ref_idx[far], ref_d[far] = -1, np.nan
    idx, d, _ = G.knn(torch.from_numpy(src).to(backend), torch.from_numpy(dst).to(backend), K=K, r=0.5,
                      bound=torch.from_numpy(bound).to(backend), bound_mode=G.BOUND_LE, cell=cell)
    assert (ref_idx[:, -1] >= 0).mean() > 0.9
    np.testing.assert_array_equal(idx.cpu().numpy(), ref_idx)
    np.testing.assert_array_equal(d.cpu().numpy(), ref_d)


@pytest.mark.parametrize("nb,K", [(8, 16), (4, 10), (5, 8), (12, 32), (8, 24), (8, 40)])  # (40: nn_graph's default in the reference)
def test_outlier_and_graph_match_oracle(backend, nb, K):
    """nb_points / K other than the pipeline's (filter.py's own default is nb_points = 4): any K <= 16 is served by the next kernel width."""
    pts, mv = _tree()
    medial = pts + mv
    radius = np.sqrt(((mv * mv)[:, 0] + (mv * mv)[:, 1]) + (mv * mv)[:, 2]).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(backend)
    keep = outlier_removal(t(medial), t(radius).unsqueeze(1), nb_points=nb)
    ref_keep = so.outlier_removal(medial, radius, nb)
    np.testing.assert_array_equal(keep.cpu().numpy(), ref_keep)
    medial, radius = medial[ref_keep], np.maximum(radius[ref_keep], np.float32(0.02))
    g = G.nn_graph(t(medial), t(radius), K=K)
    ref_e, ref_w = so.nn_graph(medial, radius, K)
    np.testing.assert_array_equal(g.edges.cpu().numpy(), ref_e)
    np.testing.assert_array_equal(g.edge_weights.cpu().numpy(), ref_w)
    assert not (ref_e[:, 1] == 0).any() and (ref_e[:, 0] == ref_e[:, 1]).sum() == len(medial) - 1  # idx > 0 quirk
    comps = G.connected_components(g, minimum_vertices=32)
    labels = so.cc_labels(len(medial), ref_e)
    np.testing.assert_array_equal(comps.labels.cpu().numpy(), labels)
    roots, counts = np.unique(labels, return_counts=True)
    order = np.lexsort((roots, -counts))
    order = [i for i in order if counts[i] >= 32]
    assert comps.n_components == len(order)
    np.testing.assert_array_equal(comps.comp_size.cpu().numpy(), counts[order])
    ref_verts = np.concatenate([np.nonzero(labels == roots[i])[0] for i in order]) if order else np.zeros(0, int)
    np.testing.assert_array_equal(comps.vert_order.cpu().numpy(), ref_verts)


_PREPARED = {}


def _prepare_components(backend, pts, mv, cache_key=None):
    """Oracle result + the component set / graph the skeleton kernels start from (cached for the strategy sweep)."""
    key = (cache_key, str(backend))
    if cache_key is not None and key in _PREPARED:
        return _PREPARED[key]
    ref = so.skeletonize(pts, mv, K=16, min_connection_length=0.02, minimum_graph_vertices=32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(backend)
    cloud = Cloud(xyz=t(pts), medial_vector=t(mv))
    medial, radius = G.medial_points(cloud.xyz, cloud.medial_vector)
    keep = outlier_removal(medial, radius.unsqueeze(1), nb_points=8)
    np.testing.assert_array_equal(keep.cpu().numpy(), ref.keep_mask)
    cloud = cloud.filter(keep)
    medial, radius = medial[keep], radius[keep]
    g = G.nn_graph(medial, radius.clamp(min=0.02), K=16)
    comps = g.connected_cugraph_components(minimum_vertices=32)
    assert comps.n_components == len(ref.components) and comps.n_components > 0
    out = (ref, comps, medial, radius, cloud.xyz[:, 1].contiguous())
    if cache_key is not None:
        _PREPARED[key] = out
    return out


def _compare_components(backend, pts, mv, block_threads, cache_key=None):
    ref, comps, medial, radius, ys = _prepare_components(backend, pts, mv, cache_key)
    res = run_components(comps, medial, radius, ys,
                         stages=STAGE_SSSP | STAGE_TREE_DISTANCE | STAGE_SAMPLE, block_threads=block_threads)
    off = comps.comp_off.cpu().numpy()
    n_branches = 0
    for c, rc in enumerate(ref.components):
        a, b = off[c], off[c + 1]
        np.testing.assert_array_equal(comps.vert_order[a:b].cpu().numpy(), rc.vertex_ids)
        assert int(res.root_local[c]) == rc.root
        np.testing.assert_array_equal(res.dist[a:b].cpu().numpy(), rc.dist)
        np.testing.assert_array_equal(res.pred[a:b].cpu().numpy(), rc.preds)
        np.testing.assert_array_equal(res.tree_dist[a:b].cpu().numpy(), rc.tree_dist)
        np.testing.assert_array_equal(rc.tree_dist, rc.dist)  # the second SSSP is the identity
        assert int(res.n_branches[c]) == len(rc.branches)
        for br in rc.branches:
            i = a + br.branch_id
            assert int(res.branch_parent[i]) == br.parent_id
            s = a + int(res.branch_off[i])
            np.testing.assert_array_equal(res.path_verts[s: s + int(res.branch_len[i])].cpu().numpy(), br.verts)
        np.testing.assert_array_equal(res.branch_of[a:b].cpu().numpy(), rc.branch_of_point)
        n_branches += len(rc.branches)
    # the totals the select loop reports (they size the assembled skeleton without a read-back) against the counted ones
    assert res.stats["branches"] == n_branches
    assert res.stats["path_vertices"] == sum(len(br.verts) for rc in ref.components for br in rc.branches)
    DeviceSkeleton.from_components(comps, res, medial, radius, verify_counts=True)  # asserts equality with its own counts
    return n_branches


def test_components_match_oracle(backend):
    pts, mv = _tree()
    assert _compare_components(backend, pts, mv, block_threads=128) >= 2


@pytest.mark.parametrize("params", [
    {0: 0},                      # no pruning: every live window entry speculates
    {0: 4000},                   # aggressive pruning: wrong guesses stop the replay
    {2: 1, 3: 1},                # one round per launch, host read-back after every launch
    {5: -1},                     # no speculation: the whole workgroup per branch, point-centric
    {5: -1, 1: -1},              # ... every path through the chunk-pruned long-path claim
    {5: -1, 1: -1, 14: 0, 4: 1 << 30},  # ... path-centric inside the workgroup
    {5: -1, 1: -1, 14: 0, 4: -1},       # ... handed to the chip-wide claim kernel
    {5: 300, 1: 2000, 4: 200},   # a mix: speculative slots, plain and long-path claims
    {5: 300, 1: 2000, 14: 0, 4: 200},   # a mix: slots, plain, local and chip-wide claims
    {6: 1, 8: 16},               # SSSP: one level per launch, 16 lanes per vertex
    {6: 7, 7: 2},                # ... seven levels per launch, read-back every second launch
    {6: 6, 13: 2, 10: 3},        # ... three workgroups, at most two vertices per workgroup and local level (the rest goes back)
    {12: 1},                     # SSSP: every round in ONE persistent launch with grid barriers
    {12: 1, 6: 7, 10: 3},        # ... seven levels per round, three workgroups
], ids=["noprune", "prune4", "relaunch", "one", "long", "local", "wide", "mixed", "mixed-wide", "sssp-rows",
        "sssp-hops", "sssp-cap", "sssp-coop", "sssp-coop-hops"])
def test_sample_tree_strategies_agree(backend, params):
    """Branch selection has four claim strategies picked by size; each one alone must reproduce the oracle."""
    from smart_tree_amd.skeleton import tuning
    pts, mv = _tree()
    with tuning.override(params):  # per call: the library has no process-global knobs
        assert _compare_components(backend, pts, mv, block_threads=256, cache_key="strategies") >= 2


@pytest.mark.parametrize("mults", [(0, 0), (100, 45), (25, 15)], ids=["max-only", "default", "fine"])
def test_grid_cell_size_is_invisible_with_radius_outliers(backend, mults):
    """The search grids take their cell from the radii (max / DIV, capped at a multiple of the MEAN so that one radius the
    network got wrong does not coarsen every cell -- csrc/st_grid.h).  The cell changes the speed of a search, never its
    result: a cloud with a few 15x radius outliers gives the oracle's graph and skeleton under every setting."""
    from smart_tree_amd.skeleton import tuning
    pts, mv = _tree()
    mv = mv.copy()
    mv[::97] *= 15.0  # ~1 % of the points with a far-too-large radius
    with tuning.override({11: mults[0], tuning.KNN_CELL_MEAN_MULT: mults[1]}):
        assert _compare_components(backend, pts, mv, block_threads=256) >= 2


def test_components_with_duplicates_and_plateaus(backend):
    pts, mv = _tree(n=2000, seed=8, exact_medial=True)
    _compare_components(backend, pts, mv, block_threads=64)


@pytest.mark.parametrize("K,min_conn,min_vertices", [(16, 0.02, 32), (8, 0.02, 32), (16, 0.05, 10), (10, 0.03, 16)])
def test_skeletonizer_forward_objects(backend, K, min_conn, min_vertices):
    """conf/pipeline.yaml's skeletonizer keywords (K, min_connection_length, minimum_graph_vertices) other than the defaults, too."""
    pts, mv = _tree(n=2000, seed=6)
    ref = so.skeletonize(pts, mv, K=K, min_connection_length=min_conn, minimum_graph_vertices=min_vertices)
    t = lambda a: torch.from_numpy(a).to(backend)
    sk = Skeletonizer(K=K, min_connection_length=min_conn, minimum_graph_vertices=min_vertices, device=backend)
    sk.block_threads = 128
    out = sk.forward(Cloud(xyz=t(pts), medial_vector=t(mv)))
    assert len(out.skeletons) == len(ref.components) >= 1 and sum(len(rc.branches) for rc in ref.components) >= 2
    kept = np.nonzero(ref.keep_mask)[0]
    medial = (pts + mv)[kept]
    radius = np.sqrt(((mv * mv)[:, 0] + (mv * mv)[:, 1]) + (mv * mv)[:, 2]).astype(np.float32)[kept]
    for tree, rc in zip(out.skeletons, ref.components):
        assert list(tree.branches.keys()) == [b.branch_id for b in rc.branches]
        for br in rc.branches:
            got = tree.branches[br.branch_id]
            assert got.parent_id == br.parent_id and got.radii.shape == (len(br.verts), 1)
            np.testing.assert_array_equal(got.xyz.numpy(), medial[rc.vertex_ids[br.verts]])
            np.testing.assert_array_equal(got.radii.numpy()[:, 0], radius[rc.vertex_ids[br.verts]])


def test_skeletonizer_empty_cloud(backend):
    sk = Skeletonizer(K=16, min_connection_length=0.02, minimum_graph_vertices=32, device=backend)
    empty = Cloud(xyz=torch.zeros((0, 3), device=backend), medial_vector=torch.zeros((0, 3), device=backend))
    assert sk.forward(empty).skeletons == []
    few = Cloud(xyz=torch.rand((10, 3), device=backend), medial_vector=torch.rand((10, 3), device=backend) * 0.01)
    assert sk.forward(few).skeletons == []


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_non_finite_medial_vectors_are_nobodys_neighbours(backend, bad):
    """A network output that overflowed (exp of a large log-radius) gives NaN / infinite medial points.  Every distance to such a
    point is NaN or infinite, so it is nobody's neighbour and finds no neighbour (the oracle's compares fail the same way): the
    skeleton equals the one of the cloud WITHOUT those points -- and the search grid must not try to span them (round 5: the
    cell-size loop of the grid never ended on an infinite bounding box)."""
    c = sample_tree_cloud(3500 if backend.type == "cpu" else 6000, seed=2, scale=0.5, max_depth=3)
    mv = c["medial_vector"].copy()
    hit = np.zeros(len(mv), bool)
    hit[[17, 900]] = True
    hit[3000:3010] = True
    mv[17, 1] = bad
    mv[900] = bad
    mv[3000:3010, 2] = -bad
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(backend)
    sk = Skeletonizer(K=16, min_connection_length=0.02, minimum_graph_vertices=32, device=backend)
    zeros = lambda n: np.zeros((n, 1), np.float32)
    got = sk.forward(Cloud(xyz=t(c["xyz"]), medial_vector=t(mv), class_l=t(zeros(len(mv)))))
    ref = sk.forward(Cloud(xyz=t(c["xyz"][~hit]), medial_vector=t(c["medial_vector"][~hit]), class_l=t(zeros(int((~hit).sum())))))
    assert len(got.skeletons) == len(ref.skeletons) >= 1
    n_branches = 0
    for a, b in zip(got.skeletons, ref.skeletons):
        assert list(a.branches) == list(b.branches)
        for k in a.branches:
            assert a.branches[k].parent_id == b.branches[k].parent_id
            assert torch.equal(a.branches[k].xyz, b.branches[k].xyz) and torch.equal(a.branches[k].radii, b.branches[k].radii)
        n_branches += len(a.branches)
    assert n_branches >= 5
    # the searches themselves: -1 / NaN rows for the non-finite queries, no non-finite index anywhere
    pts = t(c["xyz"] + mv)
    idx, dist, _ = G.knn(pts, pts, K=8, r=0.1)
    idx = idx.cpu().numpy()
    assert (idx[hit] == -1).all() and not np.isin(idx, np.nonzero(hit)[0]).any()


def test_search_bounds_of_every_kind(backend):
    """Per-query bounds that are infinite, NaN, negative, zero or huge, against brute force: an infinite bound reaches every point
    (up to the search radius), a NaN / negative one admits nobody, zero admits exact duplicates only; and with the search radius
    reduced on the device from the bounds (r < 0: outlier_removal) one infinite bound must not spoil the other queries."""
    rng = np.random.RandomState(1)
    pts = rng.rand(3000, 3).astype(np.float32)
    bound = (0.02 + 0.05 * rng.rand(3000)).astype(np.float32)
    odd = {5: np.inf, 77: np.nan, 100: -1.0, 200: 0.0, 300: 1e30}
    for i, v in odd.items():
        bound[i] = v
    t = lambda a: torch.from_numpy(a).to(backend)
    K = 8
    idx, _, _ = G.knn(t(pts), t(pts), K=K, r=0.1, bound=t(bound), bound_mode=G.BOUND_LE)
    idx = idx.cpu().numpy()
    d = pts[:, None, :] - pts[None, :, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    with np.errstate(invalid="ignore"):
        for i in list(odd) + list(range(40)):
            cand = np.nonzero((d2[i] < np.float32(0.1) ** 2) & (np.sqrt(d2[i]) <= bound[i]))[0]
            want = cand[np.lexsort((cand, d2[i][cand]))][:K]
            np.testing.assert_array_equal(idx[i][idx[i] >= 0], want, err_msg=f"query {i} (bound {bound[i]})")
        keep = outlier_removal(t(pts), t(bound).unsqueeze(1), nb_points=8).cpu().numpy()
        np.testing.assert_array_equal(keep, (np.sqrt(d2) < bound[:, None]).sum(1) >= 8)
    assert keep[5] and keep[300] and not keep[77] and not keep[100] and not keep[200]


def test_sample_tree_reference_signature(backend):
    """skeleton/path.py sample_tree(medial_pts, medial_radii, preds, distances, all_points) on one component."""
    from smart_tree_amd.skeleton.path import sample_tree

    pts, mv = _tree(n=2000, seed=12)
    ref = so.skeletonize(pts, mv)
    comp = ref.components[0]
    kept = np.nonzero(ref.keep_mask)[0]
    medial = (pts + mv)[kept][comp.vertex_ids]
    radius = np.sqrt(((mv * mv)[:, 0] + (mv * mv)[:, 1]) + (mv * mv)[:, 2]).astype(np.float32)[kept][comp.vertex_ids]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(backend)
    got = sample_tree(t(medial), t(radius).unsqueeze(1), t(comp.preds), t(comp.tree_dist), t(medial),
                      block_threads=128 if backend.type == "cpu" else 0)
    assert list(got) == [b.branch_id for b in comp.branches]
    for b in comp.branches:
        assert got[b.branch_id].parent_id == b.parent_id
        np.testing.assert_array_equal(got[b.branch_id].xyz.numpy(), medial[b.verts])
        np.testing.assert_array_equal(got[b.branch_id].radii.numpy()[:, 0], radius[b.verts])


def test_shortest_paths_reference_signature(backend):
    """skeleton/shortest_path.py shortest_paths(root, edges, edge_weights) vs the oracle's float32 Dijkstra."""
    from smart_tree_amd.skeleton.shortest_path import shortest_paths

    rng = np.random.RandomState(3)
    n = 400
    pts = rng.rand(n, 3).astype(np.float32)
    idx, dist = so.knn(pts, pts, 8, 0.5)
    src = np.repeat(np.arange(n), 8)
    ok = (idx.reshape(-1) >= 0) & (idx.reshape(-1) != src)
    edges = np.stack([src[ok], idx.reshape(-1)[ok]], axis=1).astype(np.int64)
    w = dist.reshape(-1)[ok].astype(np.float32)
    root = 17
    ref_d, ref_p = so.sssp(n, edges, w, root)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(backend)
    verts, preds, d = shortest_paths(root, t(edges), t(w), points=t(pts))
    np.testing.assert_array_equal(verts.cpu().numpy(), np.arange(n))
    reach = np.isfinite(ref_d)
    np.testing.assert_array_equal(d.cpu().numpy()[reach], ref_d[reach])
    np.testing.assert_array_equal(preds.cpu().numpy()[reach], ref_p[reach])


@pytest.mark.parametrize("symmetric", [False, True], ids=["rows<=16", "rows>16"])
def test_shortest_paths_row_lengths(backend, symmetric):
    """SSSP on the directed K = 16 neighbour graph (rows of <= 16 edges) and on its symmetrised version (longer,
    uneven rows): several levels per launch must give the same least fixed point as the oracle's Dijkstra."""
    from smart_tree_amd.skeleton.shortest_path import shortest_paths

    rng = np.random.RandomState(5)
    n = 600
    pts = (rng.rand(n, 3) * np.array([1.0, 4.0, 1.0])).astype(np.float32)
    idx, dist = so.knn(pts, pts, 16, 0.6)
    src = np.repeat(np.arange(n), 16)
    ok = (idx.reshape(-1) >= 0) & (idx.reshape(-1) != src)
    edges = np.stack([src[ok], idx.reshape(-1)[ok]], axis=1).astype(np.int64)
    w = dist.reshape(-1)[ok].astype(np.float32)
    if symmetric:
        edges = np.concatenate([edges, edges[:, ::-1]])
        w = np.concatenate([w, w])
        assert np.bincount(edges[:, 0]).max() > 16
    root = 3
    ref_d, ref_p = so.sssp(n, edges, w, root)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(backend)
    _, preds, d = shortest_paths(root, t(edges), t(w), points=t(pts))
    reach = np.isfinite(ref_d)
    assert reach.sum() > n // 2
    np.testing.assert_array_equal(d.cpu().numpy()[reach], ref_d[reach])
    np.testing.assert_array_equal(preds.cpu().numpy()[reach], ref_p[reach])


def test_shortest_paths_isolated_root_above_the_largest_edge_id(backend):
    """cugraph.sssp accepts a source that no edge mentions; its id may exceed every edge end point (advisor, round 1)."""
    from smart_tree_amd.skeleton.shortest_path import shortest_paths

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(backend)
    edges = np.array([[0, 1], [1, 2], [2, 3]], np.int64)
    w = np.array([1.0, 2.0, 0.5], np.float32)
    verts, preds, d = shortest_paths(6, t(edges), t(w))  # vertices 4, 5 and the root 6 are isolated
    assert verts.shape[0] == 7 and float(d[6]) == 0.0 and int(preds[6]) == -1
    assert torch.isinf(d[:6]).all()
    _, preds, d = shortest_paths(0, t(edges), t(w), points=t(np.zeros((9, 3), np.float32)))  # trailing isolated vertices
    assert d.shape[0] == 9 and d[:4].cpu().tolist() == [0.0, 1.0, 3.0, 3.5] and torch.isinf(d[4:]).all()
    with pytest.raises(ValueError):
        shortest_paths(0, t(edges), t(w), surface_y=t(np.zeros(2, np.float32)))


def test_outlier_removal_count_kernel_equals_the_search(backend):
    """outlier_removal (filter.py:6-11) through the counting kernel (stops at the 8th hit, builds no lists) == the
    reference's formulation over the full search: the 8th slot of knn(K=8, bound = own radius, strict) is filled."""
    from smart_tree_amd.skeleton import graph as G
    from smart_tree_amd.skeleton.filter import outlier_removal

    rng = np.random.RandomState(11)
    pts = np.concatenate([rng.rand(1500, 3) * [0.3, 1.0, 0.3], rng.rand(40, 3) * 3.0]).astype(np.float32)  # dense core + stragglers
    rad = (0.02 + 0.08 * rng.rand(len(pts))).astype(np.float32)
    rad[::7] = 0.0  # radius 0: not even itself (d = 0 < 0 fails)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(backend)
    got = outlier_removal(t(pts), t(rad).unsqueeze(1), nb_points=8)
    idx, _, _ = G.knn(t(pts), t(pts), K=8, r=-1.0, bound=t(rad), bound_mode=G.BOUND_LT, cell=-G.SEARCH_CELL_DIV)
    ref = idx[:, 7] != -1
    assert torch.equal(got.cpu(), ref.cpu()) and 0 < int(ref.sum()) < len(pts)
    np.testing.assert_array_equal(got.cpu().numpy(), so.outlier_removal(pts, rad, 8))
    # two clouds in one call: each counts inside itself only
    off = torch.tensor([0, 700, len(pts)], dtype=torch.int32, device=backend)
    both = outlier_removal(t(pts), t(rad).unsqueeze(1), nb_points=8, seg_off=off)
    for a, b in ((0, 700), (700, len(pts))):
        one = outlier_removal(t(pts[a:b]), t(rad[a:b]).unsqueeze(1), nb_points=8)
        assert torch.equal(both[a:b].cpu(), one.cpu())


def test_components_from_knn_tables_equal_components_from_the_edge_list(backend):
    """KnnGraph: labels, layout and adjacency built straight from the search tables (st_connected_components_knn /
    st_component_csr_knn) against the same graph through make_edges' int64 edge list -- one cloud and a batch of two."""
    from smart_tree_amd.data_types.graph import KnnGraph
    from smart_tree_amd.skeleton import graph as G

    rng = np.random.RandomState(3)
    pts = np.concatenate([rng.rand(900, 3) * [0.4, 2.0, 0.4], 5 + rng.rand(300, 3) * 0.5, rng.rand(40, 3) * 9]).astype(np.float32)
    rad = (0.05 + 0.1 * rng.rand(len(pts))).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(backend)
    for seg in (None, torch.tensor([0, 700, len(pts)], dtype=torch.int32, device=backend)):
        a = G.nn_graph(t(pts), t(rad), K=16, seg_off=seg)
        assert isinstance(a, KnnGraph) and a._cap_cache is None
        ca = a.connected_cugraph_components(minimum_vertices=8)  # from the tables
        b = G.nn_graph(t(pts), t(rad), K=16, seg_off=seg)
        _ = b.padded  # materialise the edge list first: the edge-list path
        cb = b.connected_cugraph_components(minimum_vertices=8)
        assert ca.n_components == cb.n_components > 1
        for name in ("comp_size", "comp_off", "vert_order", "new_id", "labels"):
            assert torch.equal(getattr(ca, name).cpu(), getattr(cb, name).cpu()), name
        # adjacency: the table form keeps every (neighbour, weight) pair of a row ONCE, the edge-list form keeps both copies of
        # a mutual pair -- the same sets, and the table form's rows have no duplicates (the order inside a row is unspecified)
        roa, rob = ca.row_off.cpu().numpy().astype(np.int64), cb.row_off.cpu().numpy().astype(np.int64)
        assert len(roa) == len(rob)
        cola, wa, colb, wb = ca.col.cpu().tolist(), ca.wgt.cpu().tolist(), cb.col.cpu().tolist(), cb.wgt.cpu().tolist()
        shorter = 0
        for v in range(len(roa) - 1):
            ra = sorted(zip(cola[roa[v]: roa[v + 1]], wa[roa[v]: roa[v + 1]]))
            rb = sorted(set(zip(colb[rob[v]: rob[v + 1]], wb[rob[v]: rob[v + 1]])))
            assert ra == rb, v
            shorter += (rob[v + 1] - rob[v]) - (roa[v + 1] - roa[v])
        assert shorter > 0  # mutual pairs exist in a kNN graph


def test_csr_from_tables_falls_back_to_the_edge_list_rows_without_the_larger_workspace(backend):
    """st_component_csr_knn with only st_component_csr_workspace_bytes of scratch (or a K that is not a power of two) builds
    st_component_csr's rows -- both copies of a mutual pair -- instead of failing (include/smarttree_hip.h)."""
    from smart_tree_amd import _lib
    from smart_tree_amd.skeleton import graph as G

    rng = np.random.RandomState(11)
    pts = (rng.rand(700, 3) * [0.4, 1.5, 0.4]).astype(np.float32)
    rad = (0.05 + 0.1 * rng.rand(len(pts))).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(backend)
    g = G.nn_graph(t(pts), t(rad), K=16)
    comps = g.connected_cugraph_components(minimum_vertices=8)  # the deduplicated rows (full workspace)
    L = _lib.lib()
    n, K, m = len(pts), 16, comps.vert_order.shape[0]
    idxs, dists = g.idxs.contiguous(), g.dists.contiguous()
    row_off = torch.empty(m + 1, dtype=torch.int32, device=backend)
    col = torch.empty(2 * n * K, dtype=torch.int32, device=backend)
    wgt = torch.empty(2 * n * K, dtype=torch.float32, device=backend)
    ws = _lib.workspace(L.st_component_csr_workspace_bytes(m), backend)  # the SMALL workspace
    assert ws.numel() < L.st_component_csr_knn_workspace_bytes(m, n, K)
    _lib.check(L.st_component_csr_knn(_lib.ptr(idxs), _lib.ptr(dists), n, K, None, _lib.ptr(comps.new_id.contiguous()), m,
                                      _lib.ptr(row_off), _lib.ptr(col), _lib.ptr(wgt), _lib.ptr(ws), ws.numel(), _lib.stream(backend)))
    if backend.type == "cuda":
        torch.cuda.synchronize()
    ro_full, ro_dedup = row_off.cpu().numpy().astype(np.int64), comps.row_off.cpu().numpy().astype(np.int64)
    cf, wf = col.cpu().tolist(), wgt.cpu().tolist()
    cd, wd = comps.col.cpu().tolist(), comps.wgt.cpu().tolist()
    longer = 0
    for v in range(m):
        full = sorted(zip(cf[ro_full[v]: ro_full[v + 1]], wf[ro_full[v]: ro_full[v + 1]]))
        dedup = sorted(zip(cd[ro_dedup[v]: ro_dedup[v + 1]], wd[ro_dedup[v]: ro_dedup[v + 1]]))
        assert sorted(set(full)) == dedup, v
        longer += len(full) - len(dedup)
    assert longer > 0  # the fallback keeps both copies of the mutual pairs


@pytest.mark.parametrize("batched", [False, True], ids=["one-cloud", "two-clouds"])
def test_outlier_removal_over_a_subset_equals_filtering_first(backend, batched):
    """Round 5: `outlier_removal(..., valid=mask)` (the class filter folded into the outlier filter: st_radius_count_seg's `valid`)
    must give exactly what the reference's order of operations gives -- filter the cloud, then remove outliers (pipeline.py:67-71,
    skeletonize.py:33-37) -- scattered back to the unfiltered array: the points outside the subset are neither queries nor neighbours."""
    from smart_tree_amd.skeleton.filter import outlier_removal

    rng = np.random.RandomState(11 + int(batched))
    n = 6000
    pts = rng.rand(n, 3).astype(np.float32) * np.array([1.0, 2.0, 1.0], np.float32)
    rad = rng.uniform(0.01, 0.12, n).astype(np.float32)
    valid = rng.rand(n) < 0.6
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(backend)
    seg = torch.tensor([0, 2500, n], dtype=torch.int32, device=backend) if batched else None
    got = outlier_removal(t(pts), t(rad).unsqueeze(1), nb_points=8, seg_off=seg, valid=t(valid)).cpu().numpy()
    idx = np.nonzero(valid)[0]
    sub_seg = None
    if batched:
        sub_seg = torch.tensor([0, int((idx < 2500).sum()), len(idx)], dtype=torch.int32, device=backend)
    want_sub = outlier_removal(t(pts[idx]), t(rad[idx]).unsqueeze(1), nb_points=8, seg_off=sub_seg).cpu().numpy()
    want = np.zeros(n, bool)
    want[idx] = want_sub
    np.testing.assert_array_equal(got, want)
    assert 0 < want.sum() < len(idx)
    ref = so.outlier_removal(pts[idx][: int((idx < 2500).sum())] if batched else pts[idx],
                             rad[idx][: int((idx < 2500).sum())] if batched else rad[idx], 8)
    np.testing.assert_array_equal(want_sub[: len(ref)], ref)  # ... and the filtered form is the oracle's
    # the general search (nb_points != 8; the library searches K = 1, 8, 16) has no subset form: it filters, searches and scatters back
    got16 = outlier_removal(t(pts), t(rad).unsqueeze(1), nb_points=16, seg_off=seg, valid=t(valid)).cpu().numpy()
    want16 = np.zeros(n, bool)
    want16[idx] = outlier_removal(t(pts[idx]), t(rad[idx]).unsqueeze(1), nb_points=16, seg_off=sub_seg).cpu().numpy()
    np.testing.assert_array_equal(got16, want16)
