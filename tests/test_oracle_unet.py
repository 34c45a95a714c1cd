"""The oracle's rulebook formulation of the three sparse convs vs dense torch convolutions (fp64),
an independent restatement of the third-party (spconv) arithmetic."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import unet_oracle as uo


def _random_active(rng, shape=(9, 10, 11), n=180, batch=2):
    cells = set()
    while len(cells) < n:
        cells.add((rng.randint(batch), rng.randint(shape[0]), rng.randint(shape[1]), rng.randint(shape[2])))
    coords = np.array(sorted(cells), dtype=np.int32)
    coords[0, 1:] = [shape[0] - 1, shape[1] - 1, shape[2] - 1]  # make the extent exact
    return coords[np.argsort(rng.rand(len(coords)))]


def _dense(x, coords, batch, shape):
    d = torch.zeros((batch, x.shape[1]) + tuple(shape), dtype=torch.float64)
    c = torch.from_numpy(coords.astype(np.int64))
    d[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]] = x
    return d


def _sample(d, coords):
    c = torch.from_numpy(coords.astype(np.int64))
    return d[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]]


def test_subm_strided_inverse_match_dense_convs():
    rng = np.random.RandomState(0)
    coords = _random_active(rng)
    shape = tuple(int(v) + 1 for v in coords[:, 1:].max(0))
    cin, cout = 5, 7
    x = torch.from_numpy(rng.randn(len(coords), cin))
    w = torch.from_numpy(rng.randn(cout, 3, 3, 3, cin))  # checkpoint layout [Cout,kz,ky,kx,Cin]
    wd = w.permute(0, 4, 1, 2, 3)
    # submanifold: dense conv (padding 1) sampled at the active sites
    y = uo.sparse_conv(x, uo.subm_rulebook(coords), w, len(coords))
    ref = _sample(F.conv3d(_dense(x, coords, 2, shape), wd, padding=1), coords)
    torch.testing.assert_close(y, ref, rtol=1e-12, atol=1e-12)
    # strided k3 s2 p1: active outputs = every output position some active input reaches
    coarse = uo.strided_out_coords(coords)
    z = uo.sparse_conv(x, uo.down_rulebook(coarse, coords), w, len(coarse))
    dz = F.conv3d(_dense(x, coords, 2, shape), wd, stride=2, padding=1)
    torch.testing.assert_close(z, _sample(dz, coarse), rtol=1e-12, atol=1e-12)
    occupancy = F.conv3d(_dense(torch.ones(len(coords), 1, dtype=torch.float64), coords, 2, shape),
                         torch.ones(1, 1, 3, 3, 3, dtype=torch.float64), stride=2, padding=1)
    assert int((occupancy > 0).sum()) == len(coarse)
    # inverse: transposed conv of the coarse tensor sampled at the fine active sites, same k per pair
    wi = torch.from_numpy(rng.randn(cin, 3, 3, 3, cout))  # SparseInverseConv3d weight [Cout'=cin, k, Cin'=cout]
    zc = torch.from_numpy(rng.randn(len(coarse), cout))
    u = uo.sparse_conv(zc, uo.up_rulebook(coords, coarse), wi, len(coords))
    out_pad = [(shape[a] - 1) % 2 for a in range(3)]
    dt = F.conv_transpose3d(_dense(zc, coarse, 2, dz.shape[2:]), wi.permute(4, 0, 1, 2, 3), stride=2, padding=1,
                            output_padding=out_pad)
    torch.testing.assert_close(u, _sample(dt, coords), rtol=1e-12, atol=1e-12)
