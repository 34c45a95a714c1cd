"""Generate tests/golden/*.npz by running the REFERENCE's own glue code (build container only).

The reference (/root/reference, pure Python) cannot run as a whole here: spconv, FRNN, cugraph, cudf,
cupy, open3d, hydra ... are CUDA-only / absent.  With permissive stub modules for those names every
reference module imports, and the parts of the hot path that are the reference's OWN code execute on
the CPU:
    CentreCloud, SingleTreeInference.compute_blocks, cube_filter      (blocking)
    outlier_removal, nn_graph / make_edges, remap_edges               (graph construction)
    sample_tree / trace_route / select_path_points, BranchSkeleton    (branch extraction)
    TreeSkeleton.prune / repair / smooth, DisjointTreeSkeleton        (post-processing)
The third-party calls they make are served by stand-ins that follow the canonical semantics the
oracle documents (FRNN: brute force, d2 < r^2, ties by index; cugraph SSSP: oracle/skeleton_oracle).
The outputs are committed as small data fixtures; no reference source is copied and nothing from
/root/reference is needed at test time.

    python tools/make_goldens.py
"""
from __future__ import annotations

import importlib
import sys
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
REF = Path("/root/reference")
OUT = ROOT / "tests" / "golden"


# ---------------------------------------------------------------------------------- stubs ---
class _Anything:
    """Attribute / call / subscript sink used for decorators, type hints and unused third-party names."""

    def __getattr__(self, name):
        return _Anything()

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]  # behaves as an identity decorator
        return _Anything()

    def __getitem__(self, item):
        return _Anything()

    def __mro_entries__(self, bases):
        return (object,)


class _ExactSqrt(torch.Tensor):
    """The reference takes `.sqrt()` of FRNN's squared distances (graph.py:26).  This container's
    torch-CPU sqrt is NOT correctly rounded (1 ulp off in ~0.5% of float32 values, an MKL artefact);
    a GPU sqrtf is.  The stand-in hands back a tensor whose sqrt() is the correctly rounded one."""

    @staticmethod
    def wrap(t):
        return torch.Tensor._make_subclass(_ExactSqrt, t)

    def sqrt(self):
        with np.errstate(invalid="ignore"):
            return torch.from_numpy(np.sqrt(self.detach().numpy().astype(np.float64)).astype(np.float32))


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    mod.__getattr__ = lambda attr: _Anything()
    sys.modules[name] = mod
    return mod


def install_stubs():
    for name in ["spconv", "spconv.pytorch", "spconv.pytorch.utils", "cumm", "open3d", "cugraph", "cudf", "cupy", "hydra",
                 "hydra.utils", "omegaconf", "wandb", "cmapy", "py_structs", "py_structs.torch", "laspy", "beartype",
                 "typeguard", "torchtyping"]:
        _stub(name)
    sys.modules["typeguard"].typechecked = lambda f=None, **k: f if f is not None else (lambda g: g)
    sys.modules["beartype"].beartype = lambda f: f
    sys.modules["torchtyping"].TensorType = _Anything()
    sys.modules["spconv.pytorch"].SparseModule = torch.nn.Module
    sys.modules["spconv"].pytorch = sys.modules["spconv.pytorch"]
    sys.modules["cudf"].DataFrame = _Anything()
    sys.modules["cugraph"].sssp = _Anything()
    # FRNN stand-in: canonical semantics (oracle/skeleton_oracle.c so_knn), FRNN's return convention
    from oracle import skeleton_oracle as so

    def frnn_grid_points(src, dst, src_len, dst_len, K, r, return_nn=False, return_sorted=True, **kw):
        idx, _ = so.knn(src[0].numpy(), dst[0].numpy(), K, float(r))
        d2 = np.where(idx >= 0, so_d2(src[0].numpy(), dst[0].numpy(), idx), np.float32(-1.0))
        return _ExactSqrt.wrap(torch.from_numpy(d2)[None]), torch.from_numpy(idx)[None], None, None

    def so_d2(src, dst, idx):
        j = np.clip(idx, 0, None)
        d = src[:, None, :] - dst[j]
        return ((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).astype(np.float32)

    _stub("frnn", frnn_grid_points=frnn_grid_points)

    # spconv PointToVoxel stand-in (CPU generator, max_num_points_per_voxel = 1): oracle/voxel_oracle.voxelize_block
    from oracle import voxel_oracle as vo

    class PointToVoxel:
        def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_voxels, max_num_points_per_voxel, device=None):
            assert max_num_points_per_voxel == 1 and len(set(float(v) for v in vsize_xyz)) == 1
            self.v, self.range = float(vsize_xyz[0]), [float(c) for c in coors_range_xyz]

        def generate_voxel_with_id(self, pc):
            pts = pc.numpy()
            assert np.array_equal(pts[:, :3].min(0), np.float32(self.range[:3])) and np.array_equal(pts[:, :3].max(0), np.float32(self.range[3:]))
            first, czyx = vo.voxelize_block(pts, self.v)
            vid = np.full(len(pts), -1, np.int64)
            return (torch.from_numpy(pts[first]).unsqueeze(1), torch.from_numpy(czyx), torch.ones(len(first), dtype=torch.int32),
                    torch.from_numpy(vid))

    sys.modules["spconv.pytorch.utils"].PointToVoxel = PointToVoxel
    install_graph_standins()


# ------------------------------------------------------------ cugraph / cudf / cupy stand-ins ---
class FSeries:
    """The few cudf.Series operations the reference's glue performs (data_types/graph.py:32-51, skeletonize.py:60-63)."""

    def __init__(self, a):
        self.a = np.asarray(a)

    def unique(self):
        return FSeries(np.unique(self.a))  # ascending: with labels = smallest member id this is the oracle's canonical order

    def to_pandas(self):
        return self.a.tolist()

    def __eq__(self, other):
        return FSeries(self.a == other)

    def count(self):
        return int(len(self.a))

    @property
    def values(self):
        return self.a

    def to_numpy(self):
        return self.a

    def __array__(self, dtype=None, copy=None):
        return self.a if dtype is None else self.a.astype(dtype)

    def __len__(self):
        return len(self.a)


class FFrame:
    def __init__(self, cols=None):
        self.cols = dict(cols or {})

    def __setitem__(self, k, v):
        self.cols[k] = np.asarray(v)

    def __getitem__(self, k):
        if isinstance(k, str):
            return FSeries(self.cols[k])
        mask = k.a if isinstance(k, FSeries) else np.asarray(k)
        return FFrame({c: v[mask] for c, v in self.cols.items()})

    @property
    def values(self):
        return np.stack([self.cols[c] for c in self.cols], 1)


class FGraph:
    """cugraph.Graph(directed=False) over an edge list with vertex ids 0 .. max (renumber=False)."""

    def __init__(self, directed=False):
        self.src = self.dst = np.zeros(0, np.int64)
        self.w = np.zeros(0, np.float32)

    def from_cudf_edgelist(self, d, edge_attr=None, renumber=False):
        assert not renumber
        self.src, self.dst = d.cols["source"].astype(np.int64), d.cols["destination"].astype(np.int64)
        self.w = d.cols[edge_attr].astype(np.float32)

    def n(self):
        return int(max(self.src.max(), self.dst.max())) + 1 if len(self.src) else 0

    def nodes(self):
        return FSeries(np.unique(np.concatenate([self.src, self.dst])))

    def edges(self):
        return FFrame({"src": self.src, "dst": self.dst})

    def edge_array(self):
        return np.stack([self.src, self.dst], 1)


def install_graph_standins():
    """cugraph / cudf / cupy are CUDA-only: connected components, sub-graphs and SSSP are served by the oracle's canonical
    restatements (oracle/skeleton_oracle.c: labels = smallest member, SSSP = least fixed point + smallest tight predecessor),
    so that the reference's OWN glue around them -- the >= minimum_vertices filter, the size ordering, the vertex-id /
    edge renumbering of process_subgraph, shortest_paths / pred_graph / the second sssp -- runs unmodified on the CPU."""
    from oracle import skeleton_oracle as so

    def connected_components(g):
        n = g.n()
        return FFrame({"labels": so.cc_labels(n, g.edge_array()), "vertex": np.arange(n, dtype=np.int64)})

    def subgraph(g, vertices):
        ids = np.asarray(vertices.a if isinstance(vertices, FSeries) else vertices)
        keep = np.isin(g.src, ids) & np.isin(g.dst, ids)
        out = FGraph()
        out.src, out.dst, out.w = g.src[keep], g.dst[keep], g.w[keep]
        return out

    def sssp(g, source=0):
        n = g.n()
        dist, pred = so.sssp(n, g.edge_array(), g.w, int(source))
        return {"vertex": np.arange(n, dtype=np.int64), "predecessor": pred, "distance": dist}

    def to_pandas_edgelist(g):
        return {"src": g.src, "dst": g.dst, "weights": g.w}

    cg = sys.modules["cugraph"]
    cg.Graph, cg.connected_components, cg.subgraph, cg.sssp, cg.to_pandas_edgelist = FGraph, connected_components, subgraph, sssp, to_pandas_edgelist
    sys.modules["cudf"].DataFrame = FFrame
    cp = sys.modules["cupy"]
    cp.asarray = lambda a: np.asarray(a.detach().cpu().numpy() if torch.is_tensor(a) else a)
    cp.unique = np.unique


def reference(module: str):
    if str(REF) not in sys.path:
        sys.path.insert(0, str(REF))
    return importlib.import_module(module)


# ---------------------------------------------------------------------------------- cases ---
def skeleton_case(name: str, xyz: np.ndarray, mv: np.ndarray):
    """Reference outlier_removal -> nn_graph -> (oracle CC + SSSP) -> reference sample_tree -> reference
    prune / repair / smooth on the largest component."""
    from oracle import skeleton_oracle as so

    r_filter = reference("smart_tree.skeleton.filter")
    r_graph = reference("smart_tree.skeleton.graph")
    r_path = reference("smart_tree.skeleton.path")
    r_tree = reference("smart_tree.data_types.tree")
    r_queries = reference("smart_tree.util.queries")
    r_tube = reference("smart_tree.data_types.tube")
    cpu = torch.device("cpu")
    r_queries.pts_to_nearest_tube_gpu.__defaults__ = (cpu,)
    r_tube.CollatedTube.to_gpu.__defaults__ = (cpu,)
    real_device = torch.device
    r_path.torch = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch)})
    r_path.torch.device = lambda *a, **k: real_device("cpu")  # path.py:74,78 hard-code "cuda"

    t = torch.from_numpy
    medial = xyz + mv
    radius = np.sqrt(((mv * mv)[:, 0] + (mv * mv)[:, 1]) + (mv * mv)[:, 2]).astype(np.float32)
    keep = r_filter.outlier_removal(t(medial), t(radius).unsqueeze(1), nb_points=8).numpy()
    raw_xyz, raw_mv = xyz, mv
    xyz, medial, radius = xyz[keep], medial[keep], radius[keep]
    clamped = np.maximum(radius, np.float32(0.02))
    # nn_graph builds a Graph dataclass (stubbed cugraph types are fine); call its pieces directly
    idxs, dists, _ = r_graph.knn(t(medial), t(medial), K=16, r=float(clamped.max()))
    idxs[dists > t(clamped).unsqueeze(1)] = -1
    edges, weights = r_graph.make_edges(dists, idxs)
    edges, weights = edges.numpy(), weights.numpy()
    labels = so.cc_labels(len(medial), edges)
    roots, counts = np.unique(labels, return_counts=True)
    big = roots[np.lexsort((roots, -counts))[0]]
    ids = np.nonzero(labels == big)[0]
    inside = np.isin(edges[:, 0], ids)
    local = r_graph.remap_edges(t(edges[inside])).numpy()
    root = int(np.argmin(xyz[ids, 1]))
    dist, pred = so.sssp(len(ids), local, weights[inside], root)
    branches = r_path.sample_tree(t(medial[ids]), t(radius[ids]).unsqueeze(1), t(pred), t(dist.copy()), t(xyz[ids]))
    out = {"raw_xyz": raw_xyz, "raw_medial_vector": raw_mv, "keep_mask": keep, "edges": edges, "weights": weights,
           "component": ids, "root": np.int64(root), "preds": pred, "dist": dist,
           "branch_ids": np.array(list(branches.keys()), np.int64),
           "branch_parent": np.array([b.parent_id for b in branches.values()], np.int64)}
    for k, b in branches.items():
        out[f"branch_{k}_xyz"] = b.xyz.numpy()
        out[f"branch_{k}_radii"] = b.radii.numpy()
    tree = r_tree.TreeSkeleton(0, branches)
    dis = r_tree.DisjointTreeSkeleton([tree])
    dis.prune(min_radius=0.01, min_length=0.02)
    dis.repair()
    dis.smooth(kernel_size=11)
    out["post_ids"] = np.array(list(tree.branches.keys()), np.int64)
    for k, b in tree.branches.items():
        out[f"post_{k}_xyz"] = b.xyz.numpy()
        out[f"post_{k}_radii"] = b.radii.numpy()
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print(name, "points", len(keep), "kept", keep.sum(), "component", len(ids), "branches", len(branches), "after post", len(tree.branches))


def quirk_cloud():
    """A small procedural tree + two isolated twigs whose graph components have exactly 32 and 31 vertices (the reference
    keeps components with count >= minimum_vertices = 32, data_types/graph.py:44-45)."""
    from oracle import voxel_oracle as vo
    from smart_tree_amd.synthetic import sample_tree_cloud

    c = sample_tree_cloud(40_000, seed=5, scale=0.5, max_depth=4)
    vx = vo.voxelize_cloud(vo.centre_cloud(c["xyz"]), c["rgb"], 0.02)
    m = vx["mask"]
    xyz, mv = [vx["feats"][m, :3]], [c["medial_vector"][vx["point"][m]]]
    rng = np.random.RandomState(3)
    for count, x0 in ((32, 4.0), (31, -4.0)):  # twig: medial points 5 mm apart on a vertical axis, radius 5 cm
        t = np.arange(count, dtype=np.float64) * 0.005
        th = rng.rand(count) * 2 * np.pi
        radial = np.stack([np.cos(th), np.zeros(count), np.sin(th)], 1)
        axis = np.stack([np.full(count, x0), 0.3 + t, np.zeros(count)], 1)
        xyz.append((axis + 0.05 * radial).astype(np.float32))
        mv.append((-0.05 * radial).astype(np.float32))
    return np.concatenate(xyz).astype(np.float32), np.concatenate(mv).astype(np.float32)


def quirks_case(name: str = "skeleton_quirks"):
    """The reference's OWN `Skeletonizer.forward` (skeleton/skeletonize.py:31-95), whole, on the CPU: outlier_removal ->
    nn_graph / make_edges -> Graph.connected_cugraph_components -> per component process_subgraph (vertex ids, Cloud.filter,
    decompose_cuda_graph + remap_edges, root_idx, shortest_paths, pred_graph, the second sssp, sample_tree) -> TreeSkeleton.
    Third-party calls go to the stand-ins above; the two torch-CPU roundings that differ from a GPU's (sqrt, documented at
    _ExactSqrt) are pinned to the correctly rounded value in Cloud.radius and pred_graph's torch.norm.
    The generator ASSERTS that every quirk SURVEY 8c(4) lists fires in this cloud."""
    r_cloud = reference("smart_tree.data_types.cloud")
    r_sk = reference("smart_tree.skeleton.skeletonize")
    r_sp = reference("smart_tree.skeleton.shortest_path")
    r_path = reference("smart_tree.skeleton.path")
    r_graph = reference("smart_tree.skeleton.graph")
    real_device = torch.device
    r_path.torch = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch)})
    r_path.torch.device = lambda *a, **k: real_device("cpu")  # path.py:74,78 hard-code "cuda"

    def exact_norm(t, dim=1):  # correctly rounded sqrt((dx^2 + dy^2) + dz^2): what sqrtf gives on a GPU
        a = t.numpy()
        s = ((a[:, 0] * a[:, 0] + a[:, 1] * a[:, 1]) + a[:, 2] * a[:, 2]).astype(np.float32)
        return torch.from_numpy(np.sqrt(s.astype(np.float64)).astype(np.float32))

    r_sp.torch = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch)})
    r_sp.torch.norm = exact_norm
    r_cloud.Cloud.radius = property(lambda self: exact_norm(self.medial_vector))
    seen = {"paths": [], "dist_pairs": []}
    real_trace, real_sample = r_path.trace_route, r_sk.sample_tree

    def trace_route(preds, idx, termination_pts):
        path, term = real_trace(preds, idx, termination_pts)
        seen["paths"].append((len(path), int(term)))
        return path, term

    r_path.trace_route = trace_route
    real_sp = r_sk.shortest_paths

    def shortest_paths(root, edges, w, renumber=True):
        out = real_sp(root, edges, w, renumber=renumber)
        seen["first"] = out[2].numpy().copy()
        return out

    def sample_tree(medial, radius, preds, distances, surface):
        seen["dist_pairs"].append(bool(np.array_equal(seen["first"], distances.numpy())))  # second SSSP == first, bit for bit
        seen.setdefault("inputs", []).append((preds.numpy().copy(), distances.numpy().copy()))
        return real_sample(medial, radius, preds, distances, surface)

    r_sk.shortest_paths, r_sk.sample_tree = shortest_paths, sample_tree
    xyz, mv = quirk_cloud()
    t = torch.from_numpy
    cloud = r_cloud.Cloud(xyz=t(xyz), medial_vector=t(mv))
    # graph-level facts for the quirk assertions (the same reference calls forward() makes)
    keep = reference("smart_tree.skeleton.filter").outlier_removal(cloud.medial_pts, cloud.radius.unsqueeze(1), nb_points=8).numpy()
    kept = cloud.filter(t(keep))
    idxs, dists, _ = r_graph.knn(kept.medial_pts, kept.medial_pts, K=16, r=float(kept.radius.clamp(min=0.02).max()))
    idxs[dists > kept.radius.clamp(min=0.02).unsqueeze(1)] = -1
    edges, weights = r_graph.make_edges(dists, idxs)
    sk = r_sk.Skeletonizer(K=16, min_connection_length=0.02, minimum_graph_vertices=32, device=real_device("cpu"))
    result = sk.forward(cloud)
    r_path.trace_route, r_sk.shortest_paths, r_sk.sample_tree = real_trace, real_sp, real_sample
    # ---- every quirk fires
    from oracle import skeleton_oracle as so
    assert (idxs[1:] == 0).any() and not (edges[:, 1] == 0).any(), "vertex 0 must have in-edges for graph.py:59 to drop"
    labels = so.cc_labels(int(keep.sum()), edges.numpy())
    sizes = np.unique(labels, return_counts=True)[1]
    assert 31 in sizes and 32 in sizes, sizes
    comp_sizes = [len(p[0]) for p in seen["inputs"]]
    assert 32 in comp_sizes and 31 not in comp_sizes and comp_sizes == sorted(comp_sizes, reverse=True), comp_sizes
    assert any(l == 1 for l, _ in seen["paths"]), "no single-vertex path (path.py:125-126)"
    assert sum(1 for _, term in seen["paths"] if term == -1) == len(result.skeletons), "one root-reaching path per component (path.py:132)"
    assert all(seen["dist_pairs"]), "second SSSP differs from the first"
    out = {"raw_xyz": xyz, "raw_medial_vector": mv, "keep_mask": keep, "edges": edges.numpy(), "weights": weights.numpy(),
           "component_sizes": np.array(comp_sizes, np.int64), "n_trees": np.int64(len(result.skeletons)),
           "single_vertex_paths": np.int64(sum(1 for l, _ in seen["paths"] if l == 1)),
           "iterations": np.int64(len(seen["paths"]))}
    for ti, tree in enumerate(result.skeletons):
        out[f"tree_{ti}_preds"], out[f"tree_{ti}_dist"] = seen["inputs"][ti]
        out[f"tree_{ti}_ids"] = np.array(list(tree.branches.keys()), np.int64)
        out[f"tree_{ti}_parent"] = np.array([b.parent_id for b in tree.branches.values()], np.int64)
        for k, b in tree.branches.items():
            out[f"tree_{ti}_branch_{k}_xyz"] = b.xyz.numpy()
            out[f"tree_{ti}_branch_{k}_radii"] = b.radii.numpy()
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print(name, "points", len(keep), "kept", int(keep.sum()), "components", comp_sizes, "branches",
          [len(tr.branches) for tr in result.skeletons], "iterations", len(seen["paths"]), "single-vertex paths", int(out["single_vertex_paths"]))


def blocking_case():
    from smart_tree_amd.synthetic import sample_tree_cloud

    r_cloud = reference("smart_tree.data_types.cloud")
    r_aug = reference("smart_tree.dataset.augmentations")
    r_ds = reference("smart_tree.dataset.dataset")
    c = sample_tree_cloud(50_000, seed=0)
    cloud = r_cloud.Cloud(xyz=torch.from_numpy(c["xyz"]), rgb=torch.from_numpy(c["rgb"]))
    centred = r_aug.CentreCloud()(cloud)
    ds = r_ds.SingleTreeInference(centred, voxel_size=0.02, block_size=4, buffer_size=0.4)
    out = {"centred_xyz": centred.xyz.numpy(), "block_centres": ds.block_centres.numpy(),
           "block_sizes": np.array([len(b) for b in ds.clouds], np.int64)}
    for i, b in enumerate(ds.clouds):
        out[f"block_{i}_first_xyz"] = b.xyz[:64].numpy()
        inner = reference("smart_tree.util.maths").cube_filter(b.xyz, ds.block_centres[i], 4)
        out[f"block_{i}_inner_count"] = np.int64(int(inner.sum()))
    # the reference's own __getitem__ (dataset.py:192-226: voxel range = block min/max, inner mask = cube_filter of the
    # voxel's representative point) and batch_collate (sparse.py:40-61), with spconv's PointToVoxel served by the
    # stand-in below (canonical semantics of oracle/voxel_oracle.voxelize_block)
    items = [ds[i] for i in range(len(ds))]
    feats, coords, mask, _ = reference("smart_tree.model.sparse").batch_collate(items)
    out["collated_coords"] = coords.numpy().astype(np.int32)
    out["collated_mask"] = mask.numpy().astype(bool)
    out["collated_xyz"] = feats[:, :3].numpy()
    np.savez_compressed(OUT / "blocking_50k.npz", **out)
    print("blocking_50k blocks", len(ds.clouds), out["block_sizes"].tolist())


def nearest_tube_case():
    """Reference pts_to_nearest_tube_gpu (util/queries.py:107-133) on the CPU: 400 points x 120 tubes."""
    r_queries = reference("smart_tree.util.queries")
    r_tube = reference("smart_tree.data_types.tube")
    cpu = torch.device("cpu")
    r_queries.pts_to_nearest_tube_gpu.__defaults__ = (cpu,)
    r_tube.CollatedTube.to_gpu.__defaults__ = (cpu,)
    rng = np.random.RandomState(42)
    m, n = 120, 400
    a = (rng.rand(m, 3) * 2).astype(np.float32)
    b = (a + rng.normal(0, 0.2, (m, 3))).astype(np.float32)
    r1 = (0.01 + 0.08 * rng.rand(m)).astype(np.float32)
    r2 = (r1 * (0.6 + 0.4 * rng.rand(m))).astype(np.float32)
    pts = (rng.rand(n, 3) * 2).astype(np.float32)
    t = torch.from_numpy
    tubes = [r_tube.Tube(t(a[i]), t(b[i]), t(r1[i:i + 1]), t(r2[i:i + 1])) for i in range(m)]
    vectors, idx, radii = r_queries.pts_to_nearest_tube_gpu(t(pts), tubes)
    # gap between the best two scores (float64), so the test knows where a different summation order may flip the index
    ab = (b - a).astype(np.float64)
    ap = pts[:, None, :].astype(np.float64) - a[None]
    tt = np.clip((ap * ab[None]).sum(2) / (ab * ab).sum(1)[None], 0, 1)
    proj = a[None] + tt[..., None] * ab[None]
    score = np.abs(np.linalg.norm(proj - pts[:, None, :], axis=2) - ((1 - tt) * r1[None] + tt * r2[None]))
    part = np.partition(score, 1, axis=1)
    np.savez_compressed(OUT / "nearest_tube.npz", pts=pts, a=a, b=b, r1=r1, r2=r2, vectors=vectors.numpy(),
                        idx=idx.numpy().astype(np.int64), radii=radii.numpy(), gap=(part[:, 1] - part[:, 0]))
    print("nearest_tube", n, "points", m, "tubes")


def skeleton_file_case():
    """A skeleton written by the reference's own save_skeleton and read back by its load_skeleton (util/file.py:73-116)."""
    r_file = reference("smart_tree.util.file")
    r_branch = reference("smart_tree.data_types.branch")
    r_tree = reference("smart_tree.data_types.tree")
    g = np.load(OUT / "skeleton_y_tree.npz")
    branches = {}
    for k, par in zip(g["branch_ids"].tolist(), g["branch_parent"].tolist()):
        branches[k] = r_branch.BranchSkeleton(k, par, torch.from_numpy(g[f"branch_{k}_xyz"]), torch.from_numpy(g[f"branch_{k}_radii"]))
    tree = r_tree.TreeSkeleton(3, branches)
    r_file.save_skeleton(tree, OUT / "ref_saved_skeleton.npz")
    back = r_file.load_skeleton(OUT / "ref_saved_skeleton.npz")
    assert list(back.branches) == list(branches)
    print("ref_saved_skeleton", len(branches), "branches", {k: v.shape for k, v in np.load(OUT / "ref_saved_skeleton.npz").items()})


def tube_mesh_case():
    """Reference tube_vertices / cylinder_triangles (o3d_abstractions/geometries.py:157-189) on a golden branch; the random
    start vector of the tangent frame is pinned."""
    r_geo = reference("smart_tree.o3d_abstractions.geometries")
    g = np.load(OUT / "skeleton_y_tree.npz")
    k = int(g["branch_ids"][0])
    pts, rad = g[f"branch_{k}_xyz"][:40], g[f"branch_{k}_radii"][:40].reshape(-1)
    start = np.array([0.36, -0.48, 0.8], np.float32)
    r_geo.random_unit = lambda dtype=np.float32: start
    verts = r_geo.tube_vertices(pts, rad, 10)
    tris = r_geo.cylinder_triangles(verts.shape[1], verts.shape[0])
    np.savez_compressed(OUT / "tube_mesh.npz", points=pts, radii=rad, start=start, vertices=verts, triangles=tris)
    print("tube_mesh", verts.shape, tris.shape)


def loss_case():
    """The reference's own compute_loss / L1Loss / cosine_similarity_loss / focal_loss / dice_loss (model/loss.py, pure torch) on
    seeded inputs: no mask, a loss mask, a loss mask + vector class, raw (non-log) radius targets."""
    r_loss = reference("smart_tree.model.loss")
    g = torch.Generator().manual_seed(7)
    n = 4000
    preds = {"radius": torch.randn(n, 1, generator=g) - 3.0, "direction": torch.nn.functional.normalize(torch.randn(n, 3, generator=g)),
             "class_l": torch.randn(n, 2, generator=g) * 2.0}
    preds["direction"][:5] = 0.0  # zero vectors: the eps clamp of CosineSimilarity
    t_dir = torch.nn.functional.normalize(torch.randn(n, 3, generator=g))
    t_dir[3:8] = 0.0
    targets = torch.cat([torch.rand(n, 1, generator=g) * 0.2 + 0.005, t_dir, (torch.rand(n, 1, generator=g) < 0.3).float()], 1)
    mask = torch.rand(n, generator=g) < 0.8
    out = {"radius": preds["radius"].numpy(), "direction": preds["direction"].numpy(), "class_l": preds["class_l"].numpy(),
           "targets": targets.numpy(), "mask": mask.numpy()}
    fns = dict(radius_loss_fn=r_loss.L1Loss, direction_loss_fn=r_loss.cosine_similarity_loss)
    cases = {"plain": dict(mask=None, vector_class=None, target_radius_log=True),
             "masked": dict(mask=mask, vector_class=None, target_radius_log=True),
             "masked_vector0": dict(mask=mask, vector_class=0, target_radius_log=True),
             "vector1_rawradius": dict(mask=None, vector_class=1, target_radius_log=False)}
    for name, kw in cases.items():
        for cls_name, cls in (("focal", r_loss.focal_loss), ("dice", r_loss.dice_loss)):
            if cls_name == "dice":  # dice_loss one-hots targets of shape [n] (a [n,1] target makes [n,1,C]: same numbers after .view(-1))
                res = r_loss.compute_loss(preds, targets, class_loss_fn=lambda o, t: cls(o, t.view(-1)), **fns, **kw)
            else:
                res = r_loss.compute_loss(preds, targets, class_loss_fn=cls, **fns, **kw)
            out[f"{name}_{cls_name}"] = np.array([float(res["radius"]), float(res["direction"]), float(res["class_l"])], np.float64)
    np.savez_compressed(OUT / "loss_vectors.npz", **out)
    print("loss_vectors", {k: v.tolist() for k, v in out.items() if k.endswith(("focal", "dice"))})


def sparse_attrs_case():
    """The reference's own `sparse_from_batch` (model/sparse.py:9-19) on the collated batch of blocking_50k.npz, with spconv's
    SparseConvTensor served by a recorder: WHAT the reference hands to spconv -- spatial_shape = the largest z / y / x (NOT + 1),
    batch_size = the number of voxels, int32 indices.  The attribute is mirrored (smart_tree_amd/model/sparse.py); the rulebooks
    deliberately do NOT take their extent from it (canonical choices)."""
    g = np.load(OUT / "blocking_50k.npz")
    seen = {}

    class SparseConvTensor:
        def __init__(self, features, indices, spatial_shape, batch_size=None):
            seen.update(features=features, indices=indices, spatial_shape=spatial_shape, batch_size=batch_size)

    sys.modules["spconv.pytorch"].SparseConvTensor = SparseConvTensor
    r_sparse = reference("smart_tree.model.sparse")
    r_sparse.spconv = sys.modules["spconv.pytorch"]
    feats = torch.from_numpy(np.concatenate([g["collated_xyz"], np.zeros_like(g["collated_xyz"])], 1))
    coords = torch.from_numpy(g["collated_coords"]).float()  # batch_collate emits float coordinates (dataset.py:216)
    r_sparse.sparse_from_batch(feats, coords, torch.device("cpu"))
    out = {"spatial_shape": seen["spatial_shape"].numpy().astype(np.int64), "batch_size": np.int64(seen["batch_size"]),
           "indices_dtype": np.array(str(seen["indices"].dtype)), "indices_equal_coords": np.bool_(
               bool(torch.equal(seen["indices"], coords.int()))), "true_extent": g["collated_coords"][:, 1:].max(0).astype(np.int64) + 1}
    np.savez_compressed(OUT / "sparse_attrs.npz", **out)
    print("sparse_attrs", {k: (v.tolist() if v.ndim else v.item()) for k, v in out.items()})


def tree_dataset_case():
    """The reference's TreeDataset.process_cloud (dataset.py:82-138) on a labelled cloud, no augmentation, with spconv's
    PointToVoxel served by the stand-in (canonical semantics of oracle/voxel_oracle.voxelize_block), then batch_collate of two
    items."""
    from smart_tree_amd.synthetic import sample_tree_cloud

    r_cloud = reference("smart_tree.data_types.cloud")
    r_ds = reference("smart_tree.dataset.dataset")
    r_sparse = reference("smart_tree.model.sparse")
    sys.modules["py_structs.torch"].map_tensors = lambda t, f: f(t)
    ds = r_ds.TreeDataset.__new__(r_ds.TreeDataset)
    ds.voxel_size, ds.augmentation, ds.device = 0.03, None, torch.device("cpu")
    ds.input_features, ds.target_features = ["xyz"], ["radius", "direction", "class_l"]
    items, out = [], {}
    for k, seed in enumerate((2, 5)):
        c = sample_tree_cloud(30_000, seed=seed, scale=0.7, max_depth=4, foliage_fraction=0.3)
        cloud = r_cloud.Cloud(xyz=torch.from_numpy(c["xyz"]), rgb=torch.from_numpy(c["rgb"]),
                              medial_vector=torch.from_numpy(c["medial_vector"]), class_l=torch.from_numpy(c["class_l"]).float().view(-1, 1))
        item = ds.process_cloud(cloud, f"tree_{k}.npz")
        items.append(item)
        out[f"seed_{k}"] = np.int64(seed)
    (inputs, targets), coords, mask, names = r_sparse.batch_collate(items)
    out.update(inputs=inputs.numpy(), targets=targets.numpy(), coords=coords.numpy().astype(np.int32), mask=mask.numpy())
    np.savez_compressed(OUT / "tree_dataset.npz", **out)
    print("tree_dataset", inputs.shape, targets.shape, coords.shape)


def y_tree(seed=0):
    """A small trunk + two limbs with exact medial vectors and a little noise."""
    rng = np.random.RandomState(seed)
    segs = [((0, 0, 0), (0, 1.2, 0), 0.12), ((0, 1.2, 0), (0.5, 2.0, 0.1), 0.07), ((0, 1.2, 0), (-0.4, 1.9, -0.2), 0.05)]
    pts, mvs = [], []
    for a, b, r in segs:
        a, b = np.array(a, float), np.array(b, float)
        n = int(900 * np.linalg.norm(b - a) * r / 0.1)
        tt = rng.rand(n)
        d = (b - a) / np.linalg.norm(b - a)
        u = np.cross(d, [1, 0, 0.3]); u /= np.linalg.norm(u)
        v = np.cross(d, u)
        th = rng.rand(n) * 2 * np.pi
        radial = np.cos(th)[:, None] * u + np.sin(th)[:, None] * v
        pts.append(a + tt[:, None] * (b - a) + r * radial + rng.normal(0, 0.002, (n, 3)))
        mvs.append(-r * radial)
    return np.concatenate(pts).astype(np.float32), np.concatenate(mvs).astype(np.float32)


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    install_stubs()
    xyz, mv = y_tree(0)
    skeleton_case("skeleton_y_tree", xyz, mv)
    # second case: procedural tree with voxel-representative points, many more branches
    from oracle import voxel_oracle as vo
    from smart_tree_amd.synthetic import sample_tree_cloud

    c = sample_tree_cloud(60_000, seed=11, scale=0.6, max_depth=4)
    vx = vo.voxelize_cloud(vo.centre_cloud(c["xyz"]), c["rgb"], 0.02)
    m = vx["mask"]
    skeleton_case("skeleton_small_tree", vx["feats"][m, :3], c["medial_vector"][vx["point"][m]])
    quirks_case()
    blocking_case()
    sparse_attrs_case()
    nearest_tube_case()
    skeleton_file_case()
    tube_mesh_case()
    loss_case()
    tree_dataset_case()


if __name__ == "__main__":
    main()
