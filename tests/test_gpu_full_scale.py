"""Size-independent properties of the projector at BASELINE.json's full sizes (configs 3 and 5), where an oracle
run would take hours: total path length = analytic chord through the grid box, the row-parallel and the
ray-parallel kernels agree bit for bit, a quarter-turn of the phantom is a shift by N_proj / 4 views, and an
energy-independent attenuation table collapses the polychromatic detection to one exponential."""
import numpy as np
import pytest
import torch

from conftest import small_scan

pytestmark = pytest.mark.gpu


def projector(ct, ph, **kw):
    from dex_ct_sim_amd import forward_project as fp
    return fp.Projector(ct, ph, **kw)


def box_chords(ct, half, views=None):
    """Analytic chord of every in-plane ray through the square [-half, half]^2 (float64, [views, channels])."""
    th = ct.thetas if views is None else ct.thetas[views[0]:views[1]]
    b, gm = th[:, None], ct.gammas[None, :]
    sx, sy = ct.SID * np.cos(b), ct.SID * np.sin(b)
    ex, ey = -np.cos(b + gm), -np.sin(b + gm)
    with np.errstate(divide='ignore', invalid='ignore'):
        ax0, ax1 = (-half - sx) / ex, (half - sx) / ex
        ay0, ay1 = (-half - sy) / ey, (half - sy) / ey
    t0 = np.maximum(np.minimum(ax0, ax1), np.minimum(ay0, ay1))
    t1 = np.minimum(np.maximum(ax0, ax1), np.maximum(ay0, ay1))
    return np.maximum(t1 - t0, 0.0)


def test_config3_512_cubed_1000x800x512(hip):
    """BASELINE config 3 geometry at full size: 512^3 phantom, 1000 views x 800 channels x 512 rows = 4.1e8 rays."""
    n = 512
    ct, ph = small_scan(n=n, nz=n, n_views=1000, n_channels=800, n_rows=n)
    mu = torch.tensor([[0.0002], [0.2], [0.5]], dtype=torch.float32, device='cuda')
    w = torch.tensor([[1000.0]], dtype=torch.float32, device='cuda')
    pr = projector(ct, ph)                                   # the kernel get_sino picks (row-parallel, 4 rows per lane)
    assert pr.native_layout == 1
    c3, p3 = pr.project_tables(mu, w, want_pathlen=True, layout=None)       # [1, V, C, R], [V, C, R, M]
    # (a) sum over materials of the path lengths = chord through the grid box, every ray
    chord = torch.tensor(box_chords(ct, 0.5 * n * ph.dx), device='cuda')    # [V, C] float64
    tot = p3.sum(-1, dtype=torch.float64)
    assert float((tot - chord[:, :, None]).abs().max()) < 2e-4
    del tot
    # (b) one energy bin: -log(counts / w) = sum_m mu_m L_m
    lin = (p3.double() * mu[:, 0].double()).sum(-1)
    assert float((-torch.log(c3[0].double() / 1000.0) - lin).abs().max()) < 2e-5
    del lin
    # (c) the ray-parallel kernel walks the same voxels: identical float32 path lengths, on a third of the views
    sub = (333, 666)
    c1, p1 = projector(ct, ph, view_range=sub, kernel=1).project_tables(mu, w, want_pathlen=True, layout=0)
    assert torch.equal(p1.permute(0, 2, 1, 3), p3[sub[0]:sub[1]])           # [v, R, C, M] vs [v, C, R, M]
    assert torch.allclose(c1[0].permute(0, 2, 1), c3[0, sub[0]:sub[1]], rtol=1e-6, atol=0)
    del c1, p1, p3
    # (d) a quarter-turn of the phantom about z = the same sinogram 250 views later (or earlier)
    ph.volume = np.ascontiguousarray(np.rot90(ph.volume, k=1, axes=(1, 2)))
    cr = projector(ct, ph).project_tables(mu, w, layout=None)
    err = [float(((torch.roll(cr[0], s, dims=0) - c3[0]).abs() / c3[0]).max()) for s in (250, -250)]
    assert min(err) < 1e-5, err
    assert max(err) > 1e-2, err                                              # the other direction is a different scan


def test_config5_1024_cubed_128_bins(hip):
    """BASELINE config 5: 1024^3 phantom (1 GiB of voxel ids), 128 energy bins, 2000 x 1024 geometry - a shard of
    16 views x 1024 channels x 1024 rows = 1.7e7 rays of it."""
    from dex_ct_sim_amd import synthetic
    n = 1024
    ct, ph = small_scan(n=n, nz=n, n_views=2000, n_channels=1024, n_rows=n)
    views = (1000, 1016)
    spec = synthetic.uniform_grid_spectrum(128)
    w = torch.tensor(np.asarray(spec.I0, dtype=np.float32)[None, :], device='cuda')          # [1, 128]
    mu_flat = torch.tensor([[0.0002], [0.2], [0.5]], dtype=torch.float32, device='cuda').repeat(1, 128).contiguous()
    pr = projector(ct, ph, view_range=views)
    c, p = pr.project_tables(mu_flat, w, want_pathlen=True, layout=None)     # [1, 16, C, R], [16, C, R, M]
    chord = torch.tensor(box_chords(ct, 0.5 * n * ph.dx, views), device='cuda')
    assert float((p.sum(-1, dtype=torch.float64) - chord[:, :, None]).abs().max()) < 2e-4
    # an attenuation table that does not depend on energy: 128 bins collapse to sum(w) exp(-sum_m mu_m L_m)
    lin = (p.double() * mu_flat[:, 0].double()).sum(-1)
    expect = w.double().sum() * torch.exp(-lin)
    assert float(((c[0].double() - expect).abs() / expect).max()) < 1e-5
    # ray-parallel kernel: same voxels
    c1, p1 = projector(ct, ph, view_range=views, kernel=1).project_tables(mu_flat, w, want_pathlen=True, layout=0)
    assert torch.equal(p1.permute(0, 2, 1, 3), p)
    assert torch.allclose(c1[0].permute(0, 2, 1), c[0], rtol=1e-6, atol=0)
    # water cylinder on axis: the central ray of every row crosses 0.8 * extent of non-air material
    mid = p[:, 512, :, 1:].sum(-1)
    assert float((mid - 0.8 * 51.2).abs().max()) < 0.15
