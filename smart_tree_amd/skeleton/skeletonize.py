"""Skeletonizer: branch cloud -> DisjointTreeSkeleton (reference smart_tree/skeleton/skeletonize.py:18-95).

Same constructor and `forward(cloud)` contract.  The reference walks the components in a host
loop (`process_subgraph`, :57-95: cugraph sub-graph -> pandas edge list -> SSSP -> predecessor
graph -> second SSSP -> `sample_tree`); here every component is processed by one launch of
`st_skeleton_components` (csrc/skeleton.hip), one persistent workgroup per component, and the
branches come back in a single device-to-host copy.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import List

import torch

from .. import _lib, profiling
from ..data_types.branch import BranchSkeleton
from ..data_types.cloud import Cloud
from ..data_types.tree import DisjointTreeSkeleton, TreeSkeleton
from .filter import outlier_removal
from . import tuning
from .graph import ComponentSet, medial_points, nn_graph

STAGE_SSSP, STAGE_TREE_DISTANCE, STAGE_SAMPLE = 1, 2, 4


GRID_DIV = 4.0  # claim grid: cell = largest radius / GRID_DIV


@dataclass
class ComponentResult:
    """Device arrays in the renumbered vertex space of a ComponentSet (slices per component)."""
    dist: torch.Tensor
    pred: torch.Tensor
    root_local: torch.Tensor
    tree_dist: torch.Tensor
    branch_parent: torch.Tensor
    branch_off: torch.Tensor
    branch_len: torch.Tensor
    n_branches: torch.Tensor
    path_verts: torch.Tensor
    branch_of: torch.Tensor
    stats: dict = None


# What the helper workgroups of this process's skeleton calls did since reset_helper_stats(): {"calls", "with_helpers", "lost"}.
# A lost helper is a fall-back (slower, not wrong); the GPU suite and bench.py assert that the product settings never lose one.
_helper_lock = __import__("threading").Lock()
_helper_stats = {"calls": 0, "with_helpers": 0, "lost": 0}


def reset_helper_stats() -> None:
    with _helper_lock:
        _helper_stats.update(calls=0, with_helpers=0, lost=0)


def helper_stats() -> dict:
    with _helper_lock:
        return dict(_helper_stats)


last_run_stats: dict = {}  # stats of the most recent run_components call of this process (diagnostics: bench.py, tools/)


def _note_helpers(stats: dict) -> None:
    global last_run_stats
    last_run_stats = stats
    with _helper_lock:
        _helper_stats["calls"] += 1
        _helper_stats["with_helpers"] += 1 if stats["helpers"] > 0 else 0
        _helper_stats["lost"] += stats["helpers_lost"]


def run_components(comps: ComponentSet, medial_pts: torch.Tensor, radius: torch.Tensor, surface_y: torch.Tensor,
                   stages: int = STAGE_SSSP | STAGE_SAMPLE, block_threads: int = 0) -> ComponentResult:
    """SSSP from the lowest surface point, canonical predecessor tree, greedy branch extraction for
    every component.  The reference's second SSSP over the predecessor tree (skeletonize.py:80-85)
    re-adds the same float32 edge lengths in the same order and therefore reproduces the first
    SSSP's distances bit for bit (tests/test_skeleton.py proves it with STAGE_TREE_DISTANCE); the
    default pipeline feeds `dist` straight into sample_tree."""
    L = _lib.lib()
    dev = medial_pts.device
    C, m = comps.n_components, comps.vert_order.shape[0]
    order = comps.vert_order.long()
    pts = medial_pts[order].contiguous().float()
    rad = radius[order].contiguous().float()
    ys = surface_y[order].contiguous().float()
    f32 = lambda k: torch.empty((max(k, 1),), dtype=torch.float32, device=dev)
    i32 = lambda k: torch.empty((max(k, 1),), dtype=torch.int32, device=dev)
    res = ComponentResult(f32(m), i32(m), i32(C), f32(m), i32(m), i32(m), i32(m), i32(C), i32(m), i32(m))
    if C == 0:
        return res
    if os.environ.get("ST_DIAG_STAGES"):  # developer aid (tools/sweep_stages.sh): leave stages out to see what they cost
        stages = int(os.environ["ST_DIAG_STAGES"])
        res.n_branches.zero_()
    # no host-side facts needed: the library lays out the claim grid from comp_off and takes the grid cell as
    # max(rad) / GRID_DIV reduced on the device (grid_cell < 0), so this stage starts without a read-back
    stats = (ctypes.c_int64 * 16)()
    stats[7] = 1 if profiling.enabled() else 0  # bracket every k_sk_select launch with HIP events
    nseg = comps.n_seg  # batched clouds: every cloud's components use their own slab of the claim grid
    ws = _lib.workspace(L.st_skeleton_workspace_bytes_seg(m, C, nseg), dev)
    with profiling.stage("skeleton_kernels"):
        _lib.check(L.st_skeleton_components_seg(
            C, _lib.ptr(comps.comp_off.contiguous()), _lib.ptr(comps.comp_seg.contiguous()) if nseg > 1 else None,
            _lib.ptr(comps.vert_seg_off) if nseg > 1 else None, nseg, m, _lib.ptr(pts), _lib.ptr(rad), _lib.ptr(ys),
            _lib.ptr(comps.row_off), _lib.ptr(comps.col), _lib.ptr(comps.wgt), -float(GRID_DIV), int(stages),
            int(block_threads), _lib.ptr(res.dist), _lib.ptr(res.pred), _lib.ptr(res.root_local), _lib.ptr(res.tree_dist),
            _lib.ptr(res.branch_parent), _lib.ptr(res.branch_off), _lib.ptr(res.branch_len), _lib.ptr(res.n_branches),
            _lib.ptr(res.path_verts), _lib.ptr(res.branch_of), stats, _lib.ptr(ws), ws.numel(), _lib.stream(dev),
            tuning.skeleton_array()))
    res.stats = {"sssp_rounds": stats[0], "plateau_rounds": stats[1], "select_launches": stats[2], "lift_levels": stats[3],
                 "helpers_lost": int(stats[8]), "helpers": int(stats[9])}
    _note_helpers(res.stats)
    if stages & STAGE_SAMPLE:  # cloud totals from the select loop's last progress read-back
        res.stats["branches"], res.stats["path_vertices"] = int(stats[6] & 0xFFFFFFFF), int(stats[6] >> 32)
    if stats[7] and stats[5]:
        # algorithmic bytes of the branch selection (SURVEY.md 8d "sample_tree"): each path vertex is written once
        # (24 B: id + xyz lookups), each claimed point raced and stamped once (16 B), plus the sorted-order cursor (8 B/vertex).
        # Only a device scalar is kept for later (the thunk must not pin the result arrays of every batch of a timed region).
        claimed = (res.branch_of[:m] >= 0).sum()
        path_vertices = float(res.stats.get("path_vertices", 0))
        profiling.add_kernel_time("k_sk_select", stats[4] * 1e-6, stats[5],
                                  lambda: path_vertices * 24.0 + float(claimed.item()) * 16.0 + m * 8.0,
                                  chip_share=comps.n_components / float(profiling.compute_units(dev)))
    return res


class Skeletonizer:
    def __init__(self, K: int, min_connection_length: float, minimum_graph_vertices: int,
                 device: torch.device = torch.device("cuda:0")):
        self.K = K
        self.min_connection_length = min_connection_length
        self.minimum_graph_vertices = minimum_graph_vertices
        self.device = device
        self.block_threads = 0  # 0 = library default (1024 lanes in the per-component select workgroup)
        # optional callable, invoked once per forward() when the last chip-filling kernel of the call (the adjacency build) has been
        # enqueued: what follows (SSSP, branch selection, post-processing) occupies one compute unit per component.  A caller that
        # keeps several batches in flight can use it to let the next batch's chip-filling phase start (bench.py)
        self.on_wide_phase_done = None

    def forward(self, cloud: Cloud) -> DisjointTreeSkeleton:
        """`cloud` may be a batch of independent clouds (Cloud.collate): every stage below then runs ONCE for all of
        them, and `.split()` of the result gives each cloud's skeleton -- identical to what it gets on its own."""
        with profiling.stage("outlier_removal"):
            pending = cloud.pending() if hasattr(cloud, "pending") else None
            if pending is not None:
                # the inner-block mask and the class filter are still pending (MaskedCloud): the outlier filter runs over that
                # subset of the UNcompacted cloud, and one compaction (one host sync) serves all three selections
                cloud, valid = pending
                medial, radius = medial_points(cloud.xyz, cloud.medial_vector)
                mask = outlier_removal(medial, radius.unsqueeze(1), nb_points=8, seg_off=cloud.seg_off, valid=valid)
            else:
                medial, radius = medial_points(cloud.xyz, cloud.medial_vector)
                mask = outlier_removal(medial, radius.unsqueeze(1), nb_points=8, seg_off=cloud.seg_off)
            keep = mask.nonzero().view(-1)  # one compaction (one host sync) shared by every field
            cloud = cloud.filter(keep, assume_sorted=True)  # nonzero(): ascending
            medial, radius = medial.index_select(0, keep), radius.index_select(0, keep)
        with profiling.stage("nn_graph"):
            graph = nn_graph(medial, radius.clamp(min=self.min_connection_length), K=self.K, seg_off=cloud.seg_off)
        with profiling.stage("components"):
            comps = graph.connected_cugraph_components(minimum_vertices=self.minimum_graph_vertices)
        if self.on_wide_phase_done is not None:
            self.on_wide_phase_done()
        with profiling.stage("sssp_sample_tree"):
            res = run_components(comps, medial, radius, cloud.xyz[:, 1].contiguous(), block_threads=self.block_threads)
        with profiling.stage("assemble"):
            return DeviceSkeleton.from_components(comps, res, medial, radius)

    def process_subgraph(self, cloud: Cloud, subgraph, skeleton_id: int = 0) -> TreeSkeleton:
        """Reference entry point (skeletonize.py:57-95) for ONE component of a ComponentSet."""
        comps, c = subgraph
        medial, radius = medial_points(cloud.xyz, cloud.medial_vector)
        res = run_components(comps, medial, radius, cloud.xyz[:, 1].contiguous(), block_threads=self.block_threads)
        return DeviceSkeleton.from_components(comps, res, medial, radius).skeletons[c]



class _PackedBranch(BranchSkeleton):
    """A BranchSkeleton whose `xyz` / `radii` are cut out of the skeleton's packed host arrays when first read
    (a few hundred tensor views per cloud are not built unless somebody looks at them)."""

    def __getattr__(self, name):  # only reached while `xyz` / `radii` are not in __dict__ yet
        if name in ("xyz", "radii"):
            xyz_h, rad_h, a, b, smoothed = self.__dict__["_pack"]
            self.__dict__["xyz"] = xyz_h[a:b]
            r = rad_h[a:b]
            self.__dict__["radii"] = r if smoothed else r.unsqueeze(1)  # smooth flattens radii to 1-D (tree.py:130-134)
            return self.__dict__[name]
        raise AttributeError(name)


class _LazyTrees:
    """List-like `skeletons` of a materialised DeviceSkeleton: tree i is built from the packed host arrays when somebody
    asks for it (a cloud has ~170 trees, a batch thousands; building them all at the end of every launch set was ~1.5 ms
    of interpreter time per 20 clouds, exposed at the tail of a pass).  A tree keeps its identity: the same object comes
    back on every access, so edits made through it (host-side prune / repair / smooth) stay."""

    def __init__(self, host, lo, hi, parent=None):
        self._host, self._lo, self._n, self._parent, self._made = host, lo, hi - lo, parent, {}

    def __len__(self):
        return self._n

    def _make(self, i):
        t = self._made.get(i)
        if t is None:
            xyz_h, rad_h, rows, offs = self._host
            g = self._lo + i
            tree = _PackedTree.__new__(_PackedTree)
            tree._id = i
            tree._lazy = (xyz_h, rad_h, rows, offs[g], offs[g + 1])
            if self._parent is not None:
                # one cloud of a batch: this view and the batch-level tree are the SAME tree under two ids.  They share one
                # branch dictionary whichever of the two is read first (`_twin`: a tree that builds its branches hands them to
                # its twin), so a host-side prune / repair / smooth made through one is seen through the other, and the
                # parent's touched() / pack() notice edits made through a view.
                src = self._parent._make(g - self._parent._lo)
                tree.__dict__["_twin"], src.__dict__["_twin"] = src, tree
                if "_branches" in src.__dict__:
                    tree.__dict__["_branches"] = src.__dict__["_branches"]
            self._made[i] = t = tree
        return t

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._make(j) for j in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        return self._make(i)

    def __iter__(self):
        return (self._make(i) for i in range(self._n))

    def touched(self) -> bool:
        """Has anybody read the branch objects of a tree (and perhaps edited them)?"""
        if any("_branches" in t.__dict__ for t in self._made.values()):
            return True
        if self._parent is not None:  # ... or of the batch-level tree this view was cut from
            lo = self._lo - self._parent._lo
            return any("_branches" in t.__dict__ for i, t in self._parent._made.items() if lo <= i < lo + self._n)
        return False


class DeviceSkeleton(DisjointTreeSkeleton):
    """A DisjointTreeSkeleton whose branches still live on the GPU as flat arrays.

    `prune` / `repair` / `smooth` -- called in that order by `Pipeline.post_process` (reference
    pipeline.py:95-106) -- are recorded and executed by ONE call of `st_post_process`
    (csrc/postprocess.hip) when `.skeletons` is first read; only then are the geometry and the branch
    table copied to the host (one copy each) and the BranchSkeleton objects built.  Any other call
    order materialises first and falls back to the host implementations of the base class."""

    def __init__(self, tree_off, parent, start, length, xyz, rad, comp_seg_off=None):
        self._dev = (tree_off, parent, start, length, xyz, rad)  # device tensors (flat branch layout)
        self._ops = {}
        self._trees = None
        self._host = None  # packed host arrays of the materialised skeleton (valid while nobody has touched the objects)
        self._seg = comp_seg_off  # batched clouds: [n_seg+1] int32 (device) tree range of every cloud
        self._seg_host = None
        self._parts = None  # split()'s per-cloud views, made ONCE: a view and the batch-level tree are twins (one branch dict)

    # -- construction ---------------------------------------------------------------------------
    @staticmethod
    def from_components(comps: ComponentSet, res: ComponentResult, medial: torch.Tensor, radius: torch.Tensor,
                        verify_counts: bool = False):
        """Flat layout: branch k of the cloud owns geometry slots [start[k], start[k] + len[k] + 1); slot
        start[k] is reserved for the connection point `repair` prepends (radius pre-filled, tree.py:92)."""
        dev = medial.device
        C = comps.n_components
        i32 = dict(dtype=torch.int32, device=dev)
        if C == 0:
            z = torch.zeros(0, **i32)
            return DeviceSkeleton(torch.zeros(1, **i32), z, z, z, torch.zeros((0, 3), device=dev), torch.zeros(0, device=dev),
                                  comps.comp_seg_off if comps.n_seg > 1 else None)
        L = _lib.lib()
        m = comps.vert_order.shape[0]
        cap_b, cap_p = max(m, 1), max(2 * m, 1)  # every branch has >= 2 vertices; slots = path vertices + branches
        tree_off = torch.empty(C + 1, **i32)
        parent, start, length = torch.empty(cap_b, **i32), torch.empty(cap_b, **i32), torch.empty(cap_b, **i32)
        xyz = torch.empty((cap_p, 3), dtype=torch.float32, device=dev)
        rad = torch.empty(cap_p, dtype=torch.float32, device=dev)
        ws = _lib.workspace(L.st_assemble_workspace_bytes(cap_b), dev)
        known = res.stats is not None and "branches" in res.stats and not verify_counts
        counts = None if known else (ctypes.c_int64 * 2)()
        _lib.check((L.st_assemble_branches_nowait if known else L.st_assemble_branches)(
            C, _lib.ptr(comps.comp_off.contiguous()), _lib.ptr(res.n_branches), _lib.ptr(res.branch_parent), _lib.ptr(res.branch_off),
            _lib.ptr(res.branch_len), _lib.ptr(res.path_verts), _lib.ptr(comps.vert_order.contiguous()),
            _lib.ptr(medial.contiguous().float()), _lib.ptr(radius.contiguous().float()), _lib.ptr(tree_off), _lib.ptr(parent),
            _lib.ptr(start), _lib.ptr(length), _lib.ptr(xyz), _lib.ptr(rad), cap_b, cap_p, counts, _lib.ptr(ws), ws.numel(),
            _lib.stream(dev)))
        if known:  # no count read-back: the skeleton stage reported them
            B = res.stats["branches"]
            P = res.stats["path_vertices"] + B
        else:
            B, P = counts[0], counts[1]
            if res.stats is not None and "branches" in res.stats:
                assert (B, P) == (res.stats["branches"], res.stats["path_vertices"] + res.stats["branches"]), "select totals disagree"
        return DeviceSkeleton(tree_off, parent[:B], start[:B], length[:B], xyz[:P], rad[:P],
                              comps.comp_seg_off if comps.n_seg > 1 else None)

    # -- deferred post-processing ---------------------------------------------------------------
    def _can_defer(self, op: str) -> bool:
        order = ("prune", "repair", "smooth")
        return self._trees is None and op not in self._ops and all(o not in self._ops for o in order[order.index(op) + 1:])

    def prune(self, min_radius, min_length) -> None:
        if self._can_defer("prune"):
            self._ops["prune"] = (float(min_radius), float(min_length))
        else:
            trees = self.skeletons
            self._host = None
            if self._seg_host is not None:  # a batch: "skeleton 0" (tree.py:164-168) is the first tree OF EVERY CLOUD, as on the device
                for b in range(len(self._seg_host) - 1):
                    if self._seg_host[b] < self._seg_host[b + 1] and trees[self._seg_host[b]].branches:
                        trees[self._seg_host[b]].prune(min_radius=min_radius, min_length=min_length)
            else:
                super().prune(min_radius, min_length)

    def repair(self) -> None:
        if self._can_defer("repair"):
            self._ops["repair"] = True
        else:
            _ = self.skeletons
            self._host = None
            super().repair()

    def smooth(self, kernel_size: int = 7) -> None:
        if self._can_defer("smooth") and kernel_size > 0:
            self._ops["smooth"] = int(kernel_size)
        else:
            _ = self.skeletons
            self._host = None
            super().smooth(kernel_size)

    # -- materialisation ------------------------------------------------------------------------
    @property
    def skeletons(self) -> List[TreeSkeleton]:
        if self._trees is None:
            self._trees = self._materialise()
        return self._trees

    @skeletons.setter
    def skeletons(self, value):
        self._trees = value
        self._host = None
        self._parts = None

    def _materialise(self) -> List[TreeSkeleton]:
        tree_off, parent, start, length, xyz, rad = self._dev
        T, B = tree_off.shape[0] - 1, parent.shape[0]
        if B == 0:
            if self._seg is not None:
                self._seg_host = self._seg.cpu().tolist()
            return [TreeSkeleton(t, {}) for t in range(T)]
        dev = xyz.device
        u8 = lambda: torch.empty(B, dtype=torch.uint8, device=dev)
        keep, repaired, smoothed = u8(), u8(), u8()
        rad_out = torch.empty_like(rad)
        depth = torch.empty(B, dtype=torch.int32, device=dev)
        xyz = xyz.clone()
        pr = self._ops.get("prune")
        L = _lib.lib()
        # batched clouds: `prune` works on skeleton 0 OF EVERY CLOUD (tree.py:164-168); the first trees are comp_seg_off[:-1]
        first, n_first = (self._seg, int(self._seg.shape[0]) - 1) if self._seg is not None else (None, 0)
        _lib.check(L.st_post_process_seg(T, _lib.ptr(tree_off), _lib.ptr(parent), _lib.ptr(start), _lib.ptr(length), _lib.ptr(xyz),
                                         _lib.ptr(rad), _lib.ptr(rad_out), _lib.ptr(keep), _lib.ptr(repaired), _lib.ptr(smoothed),
                                         _lib.ptr(depth), int(pr is not None), pr[0] if pr else 0.0, pr[1] if pr else 0.0,
                                         int("repair" in self._ops), int("smooth" in self._ops), self._ops.get("smooth", 0),
                                         _lib.ptr(first), n_first, _lib.stream(dev)))
        # two copies (geometry, branch table).  Branch k's geometry is the slot range [a_k, b_k) of the packed host arrays;
        # the TreeSkeleton / BranchSkeleton objects are only built when somebody reads `.branches` (_PackedTree), their
        # tensors are views cut on first access (_PackedBranch): a few hundred Python objects per cloud are pure
        # interpreter time, and the interpreter is what the clouds in flight share.
        # ONE device-to-host copy (a blocking round trip costs ~1 ms beside other clouds' kernels):
        # geometry, radii, branch table and tree offsets travel as one float32 buffer (the integers bit-cast)
        P = xyz.shape[0]
        table = torch.stack((parent, start, length, keep.int(), repaired.int(), smoothed.int()), dim=1)
        parts = [xyz.reshape(-1), rad_out, table.reshape(-1).view(torch.float32), tree_off.view(torch.float32)]
        if self._seg is not None:
            parts.append(self._seg.view(torch.float32))
        blob = torch.cat(parts).cpu()
        xyz_h, rad_h = blob[: 3 * P].view(P, 3), blob[3 * P: 4 * P]
        rows = blob[4 * P: 4 * P + 6 * B].view(torch.int32).view(B, 6)
        offs = blob[4 * P + 6 * B: 4 * P + 6 * B + T + 1].view(torch.int32).tolist()
        if self._seg is not None:
            self._seg_host = blob[4 * P + 6 * B + T + 1:].view(torch.int32).tolist()
        self._host = (xyz_h, rad_h, rows, offs)
        return _LazyTrees(self._host, 0, T)

    def split(self) -> List[DisjointTreeSkeleton]:
        """Batched clouds: the skeleton of every cloud of the batch, trees renumbered from 0 inside each (what
        `Skeletonizer.forward` returns for that cloud alone).  A single cloud gives [self]."""
        if self._seg is None:
            return [self]
        if getattr(self, "_parts", None) is not None:
            # the same views on every call: a tree's `_twin` link is one-to-one, a second set of views would take it over and a
            # host-side edit made through the first set would no longer reach the batch-level tree (advisor, round 5)
            return self._parts
        trees = self.skeletons
        seg = self._seg_host
        out = []
        for b in range(len(seg) - 1):
            part = _CloudSkeleton.__new__(_CloudSkeleton)
            part._dev, part._ops, part._seg, part._seg_host = None, {}, None, None
            if isinstance(trees, _LazyTrees):
                part._trees = _LazyTrees(trees._host, seg[b], seg[b + 1], parent=trees)
            else:  # the trees were replaced by host objects (skeletons setter): plain copies with local ids
                part._trees = [TreeSkeleton(local, t.branches) for local, t in enumerate(trees[seg[b]: seg[b + 1]])]
            part._host = self._host
            part._tree_range = (seg[b], seg[b + 1])
            part._parts = None
            out.append(part)
        self._parts = out
        return out

    def pack(self, cloud_id: int = 0):
        """What sharding.pack_skeleton builds branch by branch -- (table int64 [B,6], geom float32 [P,4]) -- cut out of
        the packed host arrays with a dozen tensor operations instead of five per branch."""
        trees = self.skeletons
        edited = trees.touched() if isinstance(trees, _LazyTrees) else True
        if self._host is None or edited:
            return None  # host-side edits may have happened: the caller walks the objects instead
        xyz_h, rad_h, rows, offs = self._host
        t0, t1 = getattr(self, "_tree_range", (0, len(offs) - 1))  # one cloud of a batch: its slice of the branch table
        rows, offs = rows[offs[t0]: offs[t1]], [o - offs[t0] for o in offs[t0: t1 + 1]]
        B = rows.shape[0]
        if B == 0:
            return torch.zeros((0, 6), dtype=torch.int64), torch.zeros((0, 4), dtype=torch.float32)
        rows = rows.long()
        kept = rows[:, 3] != 0
        ids = torch.arange(B)
        tree_of = torch.bucketize(ids, torch.tensor(offs[1:], dtype=torch.int64), right=True)
        first = torch.tensor(offs, dtype=torch.int64)[tree_of]
        a = (rows[:, 1] + 1 - rows[:, 4])[kept]
        n = (rows[:, 2] + rows[:, 4])[kept]
        off = torch.cumsum(n, 0) - n
        table = torch.stack((torch.full_like(n, cloud_id), tree_of[kept], (ids - first)[kept], rows[kept, 0], off, n), dim=1)
        idx = torch.arange(int(n.sum())) + torch.repeat_interleave(a - off, n)
        geom = torch.cat((xyz_h[idx].float(), rad_h[idx].reshape(-1, 1).float()), dim=1)
        return table, geom


class _CloudSkeleton(DeviceSkeleton):
    """One cloud's share of a batched DeviceSkeleton (already materialised: host objects only)."""


class _PackedTree(TreeSkeleton):
    """A TreeSkeleton whose `branches` dict is built from the packed host arrays on first access."""

    @property
    def branches(self):
        d = self.__dict__.get("_branches")
        if d is None:
            xyz_h, rad_h, rows, o, e = self._lazy
            new, fill = _PackedBranch.__new__, dict.update
            d = {}
            for j, (par, st, ln, kp, rp, sm) in enumerate(rows[o:e].tolist()):
                if kp:
                    obj = new(_PackedBranch)
                    fill(obj.__dict__, _id=j, parent_id=par, child_id=None, _pack=(xyz_h, rad_h, st + 1 - rp, st + 1 + ln, sm))
                    d[j] = obj
            self.__dict__["_branches"] = d
            twin = self.__dict__.get("_twin")  # the same tree seen through the batch / through one cloud's view
            if twin is not None and "_branches" not in twin.__dict__:
                twin.__dict__["_branches"] = d
        return d

    @branches.setter
    def branches(self, value):
        self.__dict__["_branches"] = value
        twin = self.__dict__.get("_twin")
        if twin is not None:
            twin.__dict__["_branches"] = value
