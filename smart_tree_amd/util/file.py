"""Cloud / skeleton file I/O without open3d (reference smart_tree/util/file.py:73-167, .npz paths only)."""
from __future__ import annotations

from pathlib import Path

import numpy as np

from ..data_types.cloud import Cloud


def load_cloud(path) -> Cloud:
    """`.npz` clouds with the reference's keys (file.py:156-167 -> Cloud.from_numpy, cloud.py:233-252); any other
    suffix is read as a PLY point cloud (the reference hands those to open3d, file.py:163-166)."""
    path = Path(path)
    if path.suffix == ".npz":
        with np.load(path) as z:
            cloud = Cloud.from_numpy(**{k: z[k] for k in z.files})
    elif path.suffix == ".ply":
        xyz, rgb = read_ply_points(path)
        cloud = Cloud.from_numpy(xyz=xyz, rgb=rgb if rgb is not None else np.zeros_like(xyz))
    else:
        raise ValueError(f"unsupported cloud format: {path}")
    cloud.filename = path
    return cloud


_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4",
              "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4",
              "float32": "f4", "float64": "f8"}


def read_ply_points(path):
    """Vertex element of an ASCII / binary PLY: (xyz float32 [n,3], rgb float32 [n,3] in 0..1 or None)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline().decode("ascii", "replace").strip()
            if line == "end_header":
                break
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list properties on the vertex element are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=n, ndmin=2)
            cols = {name: data[:, i] for i, (name, _) in enumerate(props)}
        else:
            end = "<" if fmt == "binary_little_endian" else ">"
            rec = np.frombuffer(f.read(n * sum(np.dtype(t).itemsize for _, t in props)),
                                dtype=[(name, end + t) for name, t in props], count=n)
            cols = {name: rec[name] for name, _ in props}
    xyz = np.stack([cols["x"], cols["y"], cols["z"]], axis=1).astype(np.float32)
    rgb = None
    if all(k in cols for k in ("red", "green", "blue")):
        rgb = np.stack([cols["red"], cols["green"], cols["blue"]], axis=1).astype(np.float32)
        if dict(props)["red"] == "u1":
            rgb /= 255.0
    return xyz, rgb


def write_ply_points(path, xyz, rgb=None) -> None:
    """Binary little-endian PLY point cloud (what `save_o3d_cloud` produces in the reference, file.py:140-145)."""
    xyz = np.asarray(xyz, dtype=np.float32)
    fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
    if rgb is not None:
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
    rec = np.empty(len(xyz), dtype=fields)
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    if rgb is not None:
        c = np.clip(np.asarray(rgb, dtype=np.float32) * 255.0 + 0.5, 0, 255).astype(np.uint8)
        rec["red"], rec["green"], rec["blue"] = c[:, 0], c[:, 1], c[:, 2]
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {len(xyz)}", "property float x", "property float y",
              "property float z"] + (["property uchar red", "property uchar green", "property uchar blue"] if rgb is not None else [])
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    with open(path, "wb") as f:
        f.write(("\n".join(header + ["end_header"]) + "\n").encode("ascii"))
        f.write(rec.tobytes())


def write_ply_skeleton(path, skeleton) -> None:
    """Skeleton as a PLY line set: one vertex per skeleton point (with its radius), one edge per tube
    (the reference's `skeleton.ply` is open3d's LineSet of the same vertices/edges, pipeline.py:88-89)."""
    pts, rad, edges, base = [], [], [], 0
    for tree in skeleton.skeletons:
        for b in tree.branches.values():
            n = len(b)
            pts.append(b.xyz.numpy())
            rad.append(b.radii.reshape(-1).numpy())
            edges.append(np.stack([np.arange(base, base + n - 1), np.arange(base + 1, base + n)], axis=1))
            base += n
    pts = np.concatenate(pts) if pts else np.zeros((0, 3), np.float32)
    rad = np.concatenate(rad) if rad else np.zeros((0,), np.float32)
    edges = np.concatenate(edges).astype(np.int32) if edges else np.zeros((0, 2), np.int32)
    vrec = np.empty(len(pts), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("radius", "<f4")])
    vrec["x"], vrec["y"], vrec["z"], vrec["radius"] = pts[:, 0], pts[:, 1], pts[:, 2], rad
    erec = np.empty(len(edges), dtype=[("vertex1", "<i4"), ("vertex2", "<i4")])
    erec["vertex1"], erec["vertex2"] = edges[:, 0], edges[:, 1]
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {len(pts)}", "property float x", "property float y",
              "property float z", "property float radius", f"element edge {len(edges)}", "property int vertex1",
              "property int vertex2", "end_header"]
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(vrec.tobytes())
        f.write(erec.tobytes())


def save_cloud(path, cloud: Cloud) -> None:
    fields = {k: getattr(cloud, k) for k in ("xyz", "rgb", "medial_vector", "class_l", "branch_ids", "branch_direction")}
    np.savez(path, **{k: v.detach().cpu().numpy() for k, v in fields.items() if v is not None})


def save_skeleton(skeleton, save_location) -> None:
    """One TreeSkeleton in the reference's `.npz` layout (util/file.py:73-94): tree_id, skeleton_xyz [P,3], skeleton_radii
    (the branches' radii concatenated, plus a trailing axis), branch_id / branch_parent_id / branch_num_elements [B]."""
    branches = list(skeleton.branches.values())
    as_np = lambda t: t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
    data = {"tree_id": skeleton._id,
            "skeleton_xyz": np.concatenate([as_np(b.xyz) for b in branches]),
            # `smooth` flattens the radii of the branches it touches to [m] (tree.py:130-134) and leaves the short ones [m,1]:
            # every branch goes in as [m,1], the layout the reference writes for an unsmoothed tree ([P,1,1] in the file)
            "skeleton_radii": np.concatenate([as_np(b.radii).reshape(-1, 1) for b in branches])[..., np.newaxis],
            "branch_id": np.asarray([b._id for b in branches]),
            "branch_parent_id": np.asarray([b.parent_id for b in branches]),
            "branch_num_elements": np.asarray([len(b) for b in branches])}
    Path(save_location).parent.mkdir(parents=True, exist_ok=True)
    np.savez(save_location, **data)


def load_skeleton(path):
    """Reference util/file.py:97-116: -> TreeSkeleton(0, {branch_id: BranchSkeleton(id, parent, xyz, radii)}).  The file holds
    radii as [P,1,1] (save adds an axis the loader never removes: the reference hands BranchSkeleton a [m,1,1] numpy array);
    here they come back as the [m,1] tensors every other producer of BranchSkeleton uses."""
    import torch

    from ..data_types.branch import BranchSkeleton
    from ..data_types.tree import TreeSkeleton

    with np.load(path) as data:
        branch_id, parent_id = data["branch_id"], data["branch_parent_id"]
        xyz, radii, sizes = data["skeleton_xyz"], data["skeleton_radii"], data["branch_num_elements"]
    offsets = np.cumsum(np.append([0], sizes))
    branches = {}
    for size, off, _id, par in zip(sizes, offsets, branch_id, parent_id):
        branches[int(_id)] = BranchSkeleton(int(_id), int(par), torch.from_numpy(np.ascontiguousarray(xyz[off: off + size])),
                                            torch.from_numpy(np.ascontiguousarray(radii[off: off + size]).reshape(-1, 1)))
    return TreeSkeleton(0, branches)


def save_skeleton_npz(path, skeleton) -> None:
    """Flat arrays: per branch (tree id, branch id, parent id, offset, length) + concatenated xyz / radii."""
    rows, xyz, radii, off = [], [], [], 0
    for tree in skeleton.skeletons:
        for b in tree.branches.values():
            rows.append((tree._id, b._id, b.parent_id, off, len(b)))
            xyz.append(b.xyz.numpy())
            radii.append(b.radii.reshape(-1).numpy())
            off += len(b)
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    np.savez(path, branches=np.asarray(rows, dtype=np.int64).reshape(-1, 5),
             xyz=np.concatenate(xyz) if xyz else np.zeros((0, 3), np.float32),
             radii=np.concatenate(radii) if radii else np.zeros((0,), np.float32))
