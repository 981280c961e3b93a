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


# ---------------------------------------------------------------------------------------------------------------
# Round 2: every BASELINE.json configuration at its full per-GPU size, compared with the float64 oracle on a sample
# of the rays the GPU computed (a few views x 8 detector rows x all channels: seconds of CPU), through the same
# calls bench.py times (Projector.project_tables -> dexct_siddon_project; matdecomp.gn_device ->
# dexct_reduce_max + dexct_gn_decompose with the fused air mask).  Tolerances: the north star's 1e-5 for the
# sinogram against the float64 textbook Siddon (oracle/dexct_oracle.c: orc_siddon_classic_ray), 1e-9 for the
# float64 Newton decomposition against the oracle's on the same counts.
REL_SINO = 1e-5
REL_GN = 1e-9
INPUT = None


def _input(*p):
    import os
    from conftest import INPUT as base
    return os.path.join(base, *p)


def _oracle_sample(ct, ph, mu64, w64, views, row0, n_rows=8, threads=8):
    """Float64 Siddon-1985 counts [S, len(views), n_rows, channels] of detector rows row0..row0+n_rows."""
    from oracle import c_oracle as co
    g = co.make_geom(ct.N_proj, ct.N_channels, n_rows, ph.z_index + row0, ph.Nx, ph.Ny, ph.Nz, ph.dx, ph.dy, ph.dz,
                     ct.SID, ct.SDD)
    vcs, ccs = ct.view_cs(), ct.chan_cs()
    return np.concatenate([co.project_classic(g, vcs, ccs, v, v + 1, ph.volume, mu64, w64, n_threads=threads)
                           for v in views], axis=1)


def _gpu_sample(counts_native, native_layout, views_local, row0, n_rows=8):
    """The same rays out of the kernel's output ([S, V, C, R] for the row-parallel kernels, else [S, V, R, C])."""
    idx = torch.tensor(list(views_local), device=counts_native.device)
    if native_layout == 1:
        return counts_native[:, idx, :, row0:row0 + n_rows].permute(0, 1, 3, 2).double().cpu().numpy()
    return counts_native[:, idx, row0:row0 + n_rows, :].double().cpu().numpy()


def _check_gn_sample(a_native, counts_native, native_layout, views_local, row0, i0, mus, n_iters, gmax, n_rows=8,
                     threads=8):
    """Decomposition of the sampled pixels: the oracle's Newton (float64) on the GPU's own counts; masked (air)
    pixels exactly 0 where counts_1 >= 0.95 * global max (matdecomp.py:195-196, :204-205)."""
    from oracle import c_oracle as co
    g = _gpu_sample(counts_native, native_layout, views_local, row0, n_rows)            # [2, v, r, c] float64
    idx = torch.tensor(list(views_local), device=a_native.device)
    if native_layout == 1:
        a = a_native[idx][:, :, row0:row0 + n_rows].permute(0, 2, 1, 3).cpu().numpy()    # [v, r, c, 2]
    else:
        a = a_native[idx][:, row0:row0 + n_rows].cpu().numpy()
    ref = co.gn_decompose(g[0].ravel(), g[1].ravel(), i0, mus, n_iters, n_threads=threads).reshape(a.shape)
    air = g[0] >= 0.95 * float(gmax)
    assert air.any() and not air.all()
    assert np.all(a[air] == 0.0)
    live = ~air
    assert np.isfinite(ref[live]).all() and np.isfinite(a[live]).all()
    err = np.abs(a[live] - ref[live]) / np.maximum(np.abs(ref[live]), 1.0)
    assert err.max() < REL_GN, err.max()
    return float(err.max()), int(live.sum())


def _dual_energy_shard(n, n_views, n_channels, view_range, sample_views, rows_at, n_iters=50, compare_exact=False):
    """Fused dual-spectrum projection (140 / 80 kVp) + Newton decomposition of one view shard, as bench.py runs it."""
    from dex_ct_sim_amd import forward_project as fp, matdecomp as md, synthetic
    from dex_ct_sim_amd._device import ptr, stream_ptr
    ct, ph = small_scan(n=n, nz=n, n_views=n_views, n_channels=n_channels, n_rows=n)
    specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
    pj = fp.Projector(ct, ph, view_range=view_range)
    E, mu_d, w_d, air = pj.upload_tables(specs)
    counts = pj.project_tables(mu_d, w_d, layout=None)                 # native layout of the kernel get_sino picks
    assert pj.native_layout == 1 and counts.shape == (2, view_range[1] - view_range[0], n_channels, n)
    _, mu64, w64 = fp.merged_tables(ct, ph, specs)
    vb = view_range[0]
    for row0 in rows_at:
        ref = _oracle_sample(ct, ph, mu64, w64, sample_views, row0)
        got = _gpu_sample(counts, 1, [v - vb for v in sample_views], row0)
        rel = np.abs(got - ref) / ref
        assert rel.max() < REL_SINO, (row0, rel.max())
    # the opt-in reduced energy quadrature (quadrature.py) on EVERY ray of the shard: <= 2e-6 from the full grid (the table's
    # verified bound is <= 1e-6; two float32 launches), and inside the full grid's parity bar against the float64 oracle
    _, mu_r, w_r, _ = pj.upload_tables(specs, 'reduced')
    info = pj.quadrature_info
    assert info is not None and 3 * info['nodes'] < info['n_full'] and info['max_rel_err'] <= 1e-6
    cr = pj.project_tables(mu_r, w_r, layout=None)
    worst = max(float(((cr[:, v0:v0 + 50].double() - counts[:, v0:v0 + 50].double()).abs() / counts[:, v0:v0 + 50].double()).max())
                for v0 in range(0, counts.shape[1], 50))
    assert worst <= 2e-6, worst
    ref = _oracle_sample(ct, ph, mu64, w64, sample_views, rows_at[0])
    got = _gpu_sample(cr, 1, [v - vb for v in sample_views], rows_at[0])
    assert (np.abs(got - ref) / ref).max() < REL_SINO
    del cr
    # decomposition exactly as get_basismat_sinos / bench.py run it on device tensors
    _, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
    gmax = torch.empty((), dtype=torch.float64, device='cuda')
    assert pj.lib.dexct_reduce_max(ptr(counts[0]), 0, counts[0].numel(), ptr(gmax), stream_ptr()) == 0
    assert float(gmax) == float(counts[0].max())
    a = md.gn_device(counts[0], counts[1], i0, mus, n_iters, 'f64', mask_max=gmax, mask_frac=0.95)     # the DEFAULT mode
    if compare_exact:
        # round 4: the default (tolerance stop) against the reference's fixed count (stop_tol = 0) on EVERY pixel of the
        # scan: within 1e-12, identical where the exact run is not finite; and the results written in the reference's
        # [view][row][channel] order by the kernel itself are the same bits
        st_d = md.last_gn_stats()
        st_default = st_d['pixel_iterations']
        # the default: start values from the gate's table of the reference's fixed points, then two full-table steps per
        # unmasked pixel (the second one is the tolerance rule's evidence of convergence)
        live = int((counts[0] < 0.95 * gmax).sum())
        assert st_d['mode'] == 'one' and 0.99 * live <= st_default <= 1.2 * live
        a_single = md.gn_device(counts[0], counts[1], i0, mus, n_iters, 'f64', mask_max=gmax, mask_frac=0.95, two_level=False)
        assert md.last_gn_stats()['mode'] == 'single' and md.last_gn_stats()['pixel_iterations'] > 6 * st_default
        a_exact = md.gn_device(counts[0], counts[1], i0, mus, n_iters, 'f64', mask_max=gmax, mask_frac=0.95, stop_tol=0.0)
        st_exact = md.last_gn_stats()['pixel_iterations']
        worst, V = 0.0, a.shape[0]
        for v0 in range(0, V, 100):
            d, x = a[v0:v0 + 100], a_exact[v0:v0 + 100]
            assert torch.equal(torch.isnan(d), torch.isnan(x))
            worst = max(worst, float(torch.nan_to_num((d - x).abs() / x.abs().clamp(min=1.0), nan=0.0).max()))
            s1 = a_single[v0:v0 + 100]
            assert torch.equal(torch.isnan(s1), torch.isnan(x))
            worst = max(worst, float(torch.nan_to_num((s1 - x).abs() / x.abs().clamp(min=1.0), nan=0.0).max()))
        assert worst <= 1e-12, worst
        assert st_default < 0.8 * st_exact, (st_default, st_exact)
        del a_exact, a_single
        R, C = int(counts.shape[3]), int(counts.shape[2])
        a_ref = md.gn_device(counts[0], counts[1], i0, mus, n_iters, 'f64', mask_max=gmax, mask_frac=0.95, out_rc=(R, C))
        assert a_ref.shape == (V, R, C, 2)
        for v0 in range(0, V, 100):
            assert torch.equal(a_ref[v0:v0 + 100].view(torch.int64),
                               a[v0:v0 + 100].permute(0, 2, 1, 3).contiguous().view(torch.int64))
        del a_ref
    out = []
    for row0 in rows_at:
        out.append(_check_gn_sample(a, counts, 1, [v - vb for v in sample_views], row0, i0, mus, n_iters, float(gmax)))
    # the decomposition inverts the forward model: counts predicted from the recovered thicknesses reproduce the
    # measured ones on every unmasked pixel of the whole shard (size-independent property, all 1e8 pixels)
    i0_d = torch.tensor(i0, device='cuda')
    mus_d = torch.tensor(mus, device='cuda')
    worst = 0.0
    for v0 in range(0, a.shape[0], 50):
        av = a[v0:v0 + 50]
        live = (av != 0).any(-1)
        ex = torch.exp(-(av[..., 0:1] * mus_d[0] + av[..., 1:2] * mus_d[1]))             # [v, C, R, nE]
        for k in range(2):
            pred = (ex * i0_d[k]).sum(-1)
            rel = ((pred - counts[k, v0:v0 + 50].double()).abs() / counts[k, v0:v0 + 50].double())[live]
            worst = max(worst, float(rel.max()))
        del ex
    assert worst < 1e-6, worst           # float32 counts in: the Newton fixed point reproduces them to their rounding
    return out


def test_config2_256_cubed_120kvp_forward_only(hip):
    """BASELINE configs[1]: 256^3 water/bone phantom, 360 views x 512 channels, ONE 120 kVp spectrum (the bundled
    120kV_1mGy_float32.bin, dose-scaled as main.py:68), forward projection only; all 256 detector rows."""
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import forward_project as fp
    n = 256
    ct, ph = small_scan(n=n, nz=n, n_views=360, n_channels=512, n_rows=n)
    spec = dx.xRaySpectrum(_input('spectrum', '120kV_1mGy_float32.bin'), '120kV')
    spec.rescale_counts(ct.A_iso * 5.0 / ct.N_proj)
    assert spec.E.size == 140 and (spec.I0 > 0).sum() >= 100              # real 120 kVp spectrum: ~120 weighted bins
    raw, log = dx.get_sino(ct, ph, spec)                                   # the public drop-in call (main.py:120)
    assert raw.shape == (360, n, 512) and raw.dtype == np.float32 and log.shape == raw.shape
    _, mu64, w64 = fp.merged_tables(ct, ph, [spec])
    air = w64.sum()
    views = [0, 45, 90, 101, 233, 359]                                     # axis-aligned and oblique
    for row0 in (0, 124, 248):
        ref = _oracle_sample(ct, ph, mu64, w64, views, row0)[0]            # [v, 8, C]
        got = raw[views][:, row0:row0 + 8].astype(np.float64)
        assert (np.abs(got - ref) / ref).max() < REL_SINO
        lref = np.log(air / ref)
        assert np.abs(log[views][:, row0:row0 + 8] - lref).max() < 2e-5 * max(1.0, lref.max())
    # the same public call with the reduced quadrature (opt-in): every ray of the scan <= 2e-6 from the full grid
    raw_r, log_r = dx.get_sino(ct, ph, spec, quadrature='reduced')
    assert (np.abs(raw_r.astype(np.float64) - raw) / raw).max() <= 2e-6
    assert np.abs(log_r.astype(np.float64) - log).max() <= 4e-6


def test_config3_dual_energy_gn_full_size(hip):
    """BASELINE configs[2] - the benchmark's step at full size: 512^3, 1000 x 800 x 512 rows, fused 140 / 80 kVp
    projection, global max, 50-iteration Newton with the fused air mask (gn_refill_kernel)."""
    res = _dual_energy_shard(512, 1000, 800, (0, 1000), sample_views=[0, 250, 333, 999], rows_at=(0, 252, 504),
                             compare_exact=True)
    assert all(n_live > 1000 for _, n_live in res)


def test_config4_2000x1024_view_shard(hip):
    """BASELINE configs[3]: 512^3, 2000 views x 1024 channels, dual energy + Newton, sharded over 8 GPUs - rank 3's
    250 views (750..1000) at full size.  (The gather itself: tests/test_shard_gloo.py, bench.py --gpus 2.)"""
    from dex_ct_sim_amd import _shard
    assert _shard.split(2000, 3, 8) == (750, 1000)
    res = _dual_energy_shard(512, 2000, 1024, (750, 1000), sample_views=[750, 875, 999], rows_at=(100, 300), compare_exact=True)
    assert all(n_live > 1000 for _, n_live in res)


def test_config5_energy_dependent_128_bins(hip):
    """BASELINE configs[4]: 1024^3 (1 GiB of ids), 128 energy bins on linspace(20, 147) with the real
    ENERGY-DEPENDENT attenuation table, 2000 x 1024 geometry, rank 5's 250-view shard = 2.6e8 rays."""
    from dex_ct_sim_amd import forward_project as fp, synthetic
    n = 1024
    ct, ph = small_scan(n=n, nz=n, n_views=2000, n_channels=1024, n_rows=n)
    views = (1250, 1500)
    spec = synthetic.uniform_grid_spectrum(128)
    pj = fp.Projector(ct, ph, view_range=views)
    E, mu_d, w_d, air = pj.upload_tables([spec])
    assert mu_d.shape == (3, 128) and float((mu_d[:, 0] - mu_d[:, -1]).abs().min()) > 0      # energy dependent
    counts = pj.project_tables(mu_d, w_d, layout=None)
    assert counts.shape == (1, 250, 1024, n)
    _, mu64, w64 = fp.merged_tables(ct, ph, [spec])
    sample = [1250, 1377, 1499]
    for row0 in (8, 512, 1000):
        ref = _oracle_sample(ct, ph, mu64, w64, sample, row0)
        got = _gpu_sample(counts, 1, [v - views[0] for v in sample], row0)
        assert (np.abs(got - ref) / ref).max() < REL_SINO
    # 128 bins -> the reduced quadrature's nodes (opt-in), every one of the shard's 2.6e8 rays
    _, mu_r, w_r, _ = pj.upload_tables([spec], 'reduced')
    assert pj.quadrature_info is not None and mu_r.shape[1] * 3 < 128
    cr = pj.project_tables(mu_r, w_r, layout=None)
    worst = max(float(((cr[:, v0:v0 + 25].double() - counts[:, v0:v0 + 25].double()).abs() / counts[:, v0:v0 + 25].double()).max())
                for v0 in range(0, 250, 25))
    assert worst <= 2e-6, worst


def test_trace_against_textbook_siddon(hip):
    """The kernel's own trace (dexct_siddon_trace: voxel-index sequence + float32 piece lengths of the fixed-point
    slab DDA) DIRECTLY against the float64 textbook Siddon 1985 (merged sorted alphas, midpoint voxel) - no mirror
    in between.  Pieces of one voxel are merged, segments shorter than 1e-7 cm (corner ties) dropped; every ray of
    two scans incl. the axis-aligned views."""
    from conftest import oracle_geom
    from dex_ct_sim_amd import forward_project as fp
    from oracle import c_oracle as co
    for n, nv, nc in ((64, 90, 128), (50, 72, 97)):
        ct, ph = small_scan(n=n, n_views=nv, n_channels=nc)
        pj = fp.Projector(ct, ph)
        g = oracle_geom(ct, ph)
        vcs, ccs = ct.view_cs(), ct.chan_cs()
        plan = pj.plan_host()
        rays = np.array([(v, 0, c) for v in range(nv) for c in range(nc)], dtype=np.int32)
        vox, ln, ns = pj.trace(rays)
        n_rays_with_material = 0
        for k, (v, _, c) in enumerate(rays):
            vc, lc = co.classic_ray(g, vcs, ccs, v, c)
            p = plan[v * nc + c]
            vd, ld = vox[k, :ns[k]], ln[k, :ns[k]].astype(np.float64) * float(p['len_per_u'])
            # merge consecutive pieces of the same voxel
            if len(vd):
                cut = np.flatnonzero(np.diff(vd)) + 1
                starts = np.concatenate([[0], cut])
                vm, lm = vd[starts], np.add.reduceat(ld, starts)
            else:
                vm, lm = vd, ld
            if len(vc) and len(vm) and vc[0] != vm[0]:
                vm, lm = vm[::-1], lm[::-1]                      # the DDA walks along +u, Siddon from the source
            cm, dm = lc > 1e-7, lm > 1e-7
            v0 = float(p['V0']) / 2.0 ** 40
            if abs(float(p['SV'])) < 2.0 ** 10 and abs(v0 - round(v0)) < 1e-9:
                assert cm.sum() == dm.sum()                      # a ray running ALONG a grid plane: either row is exact
                continue
            assert np.array_equal(vc[cm], vm[dm]), (v, c)
            assert np.max(np.abs(lc[cm] - lm[dm]), initial=0.0) < 3e-7
            n_rays_with_material += len(vc) > 0
        assert n_rays_with_material > 0.5 * len(rays)


@pytest.mark.parametrize('counts_per_ray', [1e6, 2e4])
def test_noisy_scan_default_mode_against_the_exact_count(hip, counts_per_ray):
    """A scan WITH quantum noise at configs[1]'s size (256^3, 360 x 512 x 256 = 4.7e7 pixels, 140 / 80 kVp; 1e6 photons
    per open-beam ray: clinical; 2e4: photon-starved behind 40 cm of water) through projection and decomposition: the
    default mode (short cut where its gate is open, the reference's walk elsewhere) against the exact count on EVERY
    pixel - within 1e-12 where the exact count is finite, the same NaN / inf pattern elsewhere."""
    from dex_ct_sim_amd import forward_project as fp, matdecomp as md, synthetic
    from dex_ct_sim_amd._device import ptr, stream_ptr
    n = 256
    ct, ph = small_scan(n=n, nz=n, n_views=360, n_channels=512, n_rows=n)
    specs = [synthetic.kramers_spectrum(140, total_counts=counts_per_ray), synthetic.kramers_spectrum(80, total_counts=counts_per_ray)]
    pj = fp.Projector(ct, ph)
    counts = pj.project(specs, noise=True, seed=11, layout=None)[0]
    clean = pj.project(specs, layout=None)[0]
    assert counts.shape == clean.shape and not torch.equal(counts, clean)
    _, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
    gmax = torch.empty((), dtype=torch.float64, device='cuda')
    assert pj.lib.dexct_reduce_max(ptr(counts[0]), 0, counts[0].numel(), ptr(gmax), stream_ptr()) == 0
    kw = dict(mask_max=gmax, mask_frac=0.95, out_rc=(n, 512))
    exact = md.gn_device(counts[0], counts[1], i0, mus, 50, 'f64', stop_tol=0.0, **kw)
    n_exact = md.last_gn_stats()['pixel_iterations']
    a = md.gn_device(counts[0], counts[1], i0, mus, 50, 'f64', **kw)
    st = md.last_gn_stats()
    assert st['mode'] == 'one'
    worst, bad_pattern, beyond = 0.0, 0, 0
    for v0 in range(0, 360, 60):
        d, x = a[v0:v0 + 60], exact[v0:v0 + 60]
        fin = torch.isfinite(x).all(-1)
        bad_pattern += int((torch.isfinite(d).all(-1) != fin).sum())
        e = ((d - x).abs() / x.abs().clamp(min=1.0)).amax(-1)[fin & (x.abs().amax(-1) < 1e6)]
        worst = max(worst, float(e.max()))
        beyond += int((e > 1e-12).sum())
    # every pixel the exact count leaves finite is finite here and vice versa, and every one of them is within 1e-12 (before
    # the tolerance rule asked the walk from 1e-6 for two contracting steps and the worse of their ratios: 481 wandering pixels
    # of the 2e4-photon scan ended finite where the reference ends NaN, 48 creeping ones up to 2e-5 away)
    assert bad_pattern == 0 and beyond == 0 and worst <= 1e-12, (bad_pattern, beyond, worst)
    assert st['pixel_iterations'] < (0.35 if counts_per_ray >= 1e6 else 0.8) * n_exact
