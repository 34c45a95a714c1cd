"""st_points_to_nearest_tube (util/queries.py pts_to_nearest_tube_gpu / skeleton_to_points) against the oracle."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import queries_oracle as qo
from smart_tree_amd.util.queries import nearest_tube_device

GOLDEN = Path(__file__).resolve().parent / "golden"


def _case(n, m, seed, degenerate=False):
    rng = np.random.RandomState(seed)
    a = rng.rand(m, 3).astype(np.float32) * 4
    b = (a + rng.normal(0, 0.15, (m, 3))).astype(np.float32)
    r1 = (0.01 + 0.1 * rng.rand(m)).astype(np.float32)
    r2 = (r1 * (0.7 + 0.3 * rng.rand(m))).astype(np.float32)
    if degenerate:
        b[m // 3] = a[m // 3]  # zero-length tube: t = 0/0 = NaN, the NaN score wins (torch.argmin semantics)
    pts = (rng.rand(n, 3) * 4).astype(np.float32)
    return pts, a, b, r1, r2


@pytest.mark.parametrize("n,m,deg", [(1, 1, False), (700, 37, False), (3000, 1300, False), (500, 90, True)])
def test_nearest_tube_bit_exact(backend, n, m, deg):
    pts, a, b, r1, r2 = _case(n, m, seed=n + m, degenerate=deg)
    vec, idx, rad = qo.nearest_tube(pts, a, b, r1, r2)
    t = lambda x: torch.from_numpy(x).to(backend)
    gv, gi, gr = nearest_tube_device(t(pts), t(a), t(b), t(r1), t(r2))
    np.testing.assert_array_equal(gi.cpu().numpy(), idx)
    np.testing.assert_array_equal(gv.cpu().numpy(), vec)
    np.testing.assert_array_equal(gr.cpu().numpy(), rad)
    if deg:
        assert (idx == m // 3).all()


def test_oracle_against_reference_golden():
    """tests/golden/nearest_tube.npz was produced by the REFERENCE's pts_to_nearest_tube_gpu (torch CPU, device
    patched; tools/make_goldens.py).  Its einsum summation order is torch's, so: indices equal wherever the best two
    scores are further apart than 1e-5, vectors / radii to 1e-5."""
    z = np.load(GOLDEN / "nearest_tube.npz")
    vec, idx, rad = qo.nearest_tube(z["pts"], z["a"], z["b"], z["r1"], z["r2"])
    clear = z["gap"] > 1e-5
    assert clear.mean() > 0.98
    np.testing.assert_array_equal(idx[clear], z["idx"][clear])
    np.testing.assert_allclose(vec[clear], z["vectors"][clear], atol=1e-5)
    np.testing.assert_allclose(rad[clear], z["radii"][clear], atol=1e-5)


@pytest.mark.gpu
def test_skeleton_to_points_on_the_million_point_cloud():
    """Label the 1M-point cloud with a 4000-tube skeleton (4e9 pairs): properties instead of the oracle --
    a subset against the oracle bit for bit, and the chosen tube is at least as good as any of 64 random others."""
    dev = torch.device("cuda:0")
    pts, a, b, r1, r2 = _case(1_000_000, 4000, seed=3)
    t = lambda x: torch.from_numpy(x).to(dev)
    gv, gi, gr = nearest_tube_device(t(pts), t(a), t(b), t(r1), t(r2))
    sub = np.random.RandomState(0).choice(len(pts), 2000, replace=False)
    vec, idx, rad = qo.nearest_tube(pts[sub], a, b, r1, r2)
    np.testing.assert_array_equal(gi.cpu().numpy()[sub], idx)
    np.testing.assert_array_equal(gv.cpu().numpy()[sub], vec)
    best = (gv.norm(dim=1) - gr).abs()
    other = torch.from_numpy(np.random.RandomState(1).randint(0, 4000, 64)).to(dev)
    ov, _, orad = nearest_tube_device(t(pts), t(a)[other], t(b)[other], t(r1)[other], t(r2)[other])
    assert bool(((ov.norm(dim=1) - orad).abs() >= best - 1e-6).all())
