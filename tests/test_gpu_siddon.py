"""HIP Siddon projector against the oracle (all calls go through the C ABI via the host shims).

bit-exact: plan table, voxel-index sequences, float32 piece lengths, per-material path lengths
           (vs the DDA form of the oracle, which mirrors the kernel arithmetic and is itself pinned
           against the float64 textbook Siddon in test_siddon_oracle.py)
1e-5 rel : sinogram counts vs the float64 textbook Siddon + float64 detection (north-star tolerance)
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import INPUT, oracle_geom, small_scan
from oracle import c_oracle as co

pytestmark = pytest.mark.gpu
REL_TOL = 1e-5
C_byref = ctypes.byref


def projector(ct, ph, **kw):
    from dex_ct_sim_amd import forward_project as fp
    return fp.Projector(ct, ph, **kw)


def spectra():
    from dex_ct_sim_amd import synthetic
    return [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]


def ph_many(ph, n_mat):
    """Scramble the non-air voxels over n_mat - 1 materials (many material boundaries)."""
    from dex_ct_sim_amd.system import AIR, BONE, WATER, Material
    rng = np.random.default_rng(11)
    ph.volume = np.where(ph.volume > 0, rng.integers(1, n_mat, ph.volume.shape, dtype=np.uint8), 0).astype(np.uint8)
    ph.materials = [AIR, WATER, BONE] + [Material(f'm{i}', 1.0 + 0.1 * i, 'H(11.2)O(88.8)') for i in range(3, n_mat)]
    return ph


@pytest.mark.parametrize('n,nv,nc', [(48, 60, 96), (50, 72, 97), (64, 8, 300)])
def test_plan_bit_exact(hip, n, nv, nc):
    ct, ph = small_scan(n=n, n_views=nv, n_channels=nc)
    pj = projector(ct, ph)
    ref = co.plan(oracle_geom(ct, ph), ct.view_cs(), ct.chan_cs(), 0, nv)
    got = pj.plan_host()
    for f in ref.dtype.names:
        assert np.array_equal(got[f], ref[f]), f


def test_plan_nonsquare_anisotropic(hip):
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd.system import AIR, BONE, WATER
    rng = np.random.default_rng(5)
    vol = rng.integers(0, 3, (2, 40, 56), dtype=np.uint8)
    ph = dx.VoxelPhantom.from_array('aniso', vol, [AIR, WATER, BONE], dx=0.5, dy=0.7, dz=1.0)
    ct = dx.FanBeamGeometry(N_channels=70, N_proj=50, gamma_fan=0.7, SID=60.0, SDD=100.0, N_rows=2)
    pj = projector(ct, ph)
    g = oracle_geom(ct, ph)
    ref = co.plan(g, ct.view_cs(), ct.chan_cs(), 0, 50)
    got = pj.plan_host()
    for f in ref.dtype.names:
        assert np.array_equal(got[f], ref[f]), f
    E = np.array([40.0, 60.0, 80.0])
    mu, w = ph.mu_table(E), np.array([[1e4, 2e4, 1e4]])
    for kernel in (1, 2, 3, 5, 6): # nz = 2 is not a multiple of 4: the host pads the uploaded copy for kernel 3
        pj = projector(ct, ph, kernel=kernel)
        c, pl = pj.project_tables(torch.tensor(mu, dtype=torch.float32, device='cuda'),
                                  torch.tensor(w, dtype=torch.float32, device='cuda'), want_pathlen=True)
        _, rpl = co.project_dda(g, ct.view_cs(), ct.chan_cs(), 0, 50, ph.volume, mu, w, True)
        assert np.array_equal(pl.cpu().numpy(), rpl)
        cls = co.project_classic(g, ct.view_cs(), ct.chan_cs(), 0, 50, ph.volume, mu, w)
        assert np.max(np.abs(c.cpu().numpy() - cls) / cls) < REL_TOL


@pytest.mark.parametrize('n,nv,nc', [(64, 90, 128), (50, 72, 97)])
def test_voxel_index_sequence_bit_exact(hip, n, nv, nc):
    """Every ray of the scan, incl. the axis-aligned views: the sequence of voxel indices and the
    float32 piece lengths are identical to the oracle's."""
    ct, ph = small_scan(n=n, n_views=nv, n_channels=nc)
    pj = projector(ct, ph)
    g = oracle_geom(ct, ph)
    plan = co.plan(g, ct.view_cs(), ct.chan_cs(), 0, nv)
    rays = np.array([(v, 0, c) for v in range(nv) for c in range(nc)], dtype=np.int32)
    vox, ln, ns = pj.trace(rays)
    for k, (v, _, c) in enumerate(rays):
        rv, rl = co.dda_ray(g, plan[v * nc + c], 0)
        assert ns[k] == len(rv)
        assert np.array_equal(vox[k, :ns[k]], rv)
        assert np.array_equal(ln[k, :ns[k]], rl)


@pytest.mark.parametrize('kernel', [1, 2, 3, 4, 5, 6, 8])
@pytest.mark.parametrize('n_mat', [2, 3, 4, 7, 13, 16, 29])
def test_pathlen_bit_exact_and_counts(hip, kernel, n_mat):
    """Register accumulators (<= 4 materials), LDS accumulators (more), the packed-count 4-rows-per-lane
    kernel and its material-group form (kernel 4, up to 16 materials), the group passes on 2-bit packed codes (kernel
    8); 66 rows from slice 2 of 70: neither a multiple of 4 (or 16), so the host pads the uploaded volume, and the last
    lane is ragged."""
    from dex_ct_sim_amd._native import DexctError
    from dex_ct_sim_amd.system import AIR, BONE, WATER, Material
    ct, ph = small_scan(n=48, nz=70, n_views=24, n_channels=80, n_rows=66, z_index=2)      # unaligned on purpose
    if n_mat == 2:
        ph.volume = np.minimum(ph.volume, 1).astype(np.uint8)
        ph.materials = [AIR, WATER]
    if kernel in (3, 5, 6) and n_mat > 4:
        with pytest.raises(DexctError):
            projector(ct, ph_many(ph, n_mat), kernel=kernel).project(spectra())
        return
    if n_mat > 3:
        ph = ph_many(ph, n_mat)
    g = oracle_geom(ct, ph)
    pj = projector(ct, ph, kernel=kernel)
    sp = spectra()
    (counts, pl), air = pj.project(sp, want_pathlen=True)
    from dex_ct_sim_amd import forward_project as fp
    E, mu, w = fp.merged_tables(ct, ph, sp)
    _, rpl = co.project_dda(g, ct.view_cs(), ct.chan_cs(), 0, ct.N_proj, ph.volume, mu, w, True, n_threads=8)
    assert np.array_equal(pl.cpu().numpy(), rpl)
    cls = co.project_classic(g, ct.view_cs(), ct.chan_cs(), 0, ct.N_proj, ph.volume, mu, w, n_threads=8)
    rel = np.max(np.abs(counts.cpu().numpy() - cls) / cls)
    assert rel < REL_TOL, rel


def test_view_shard_equals_full(hip):
    ct, ph = small_scan(n=40, n_views=30, n_channels=64)
    sp = spectra()
    full, _ = projector(ct, ph).project(sp)
    a, _ = projector(ct, ph, view_range=(0, 13)).project(sp)
    b, _ = projector(ct, ph, view_range=(13, 30)).project(sp)
    assert torch.equal(torch.cat([a, b], dim=1), full)


def test_wide_fan_misses_and_empty_volume(hip):
    import dex_ct_sim_amd as dx
    _, ph = small_scan(n=16, n_views=4, n_channels=32)
    ct = dx.FanBeamGeometry(N_channels=32, N_proj=4, gamma_fan=2.4, SID=60.0, SDD=100.0)
    sp = spectra()
    counts, air = projector(ct, ph).project(sp)
    c = counts.cpu().numpy()
    plan = co.plan(oracle_geom(ct, ph), ct.view_cs(), ct.chan_cs(), 0, 4)
    miss = (plan['n_slabs'] == 0).reshape(4, 32)
    assert miss.any()
    for s in range(2):       # rays that miss the grid see the unattenuated spectrum
        assert np.allclose(c[s, :, 0, :][miss], air[s], rtol=2e-6)
    ph.volume[:] = 0          # all air: attenuation by air only, still matches the oracle
    counts, _ = projector(ct, ph).project(sp)
    from dex_ct_sim_amd import forward_project as fp
    E, mu, w = fp.merged_tables(ct, ph, sp)
    cls = co.project_classic(oracle_geom(ct, ph), ct.view_cs(), ct.chan_cs(), 0, 4, ph.volume, mu, w)
    assert np.max(np.abs(counts.cpu().numpy() - cls) / cls) < REL_TOL


def test_full_size_properties(hip):
    """256^2 x 64 slices, 360 views x 512 channels (BASELINE config 2 geometry, fewer slices):
    size-independent properties instead of an oracle run: (a) total path length over materials equals
    the analytic chord through the grid box, (b) all three kernels agree bit for bit, (c) linearity: with one
    energy bin, -log(counts/w) equals sum_m mu_m L_m."""
    ct, ph = small_scan(n=256, nz=64, n_views=360, n_channels=512, n_rows=64)
    mu = torch.tensor([[0.0002], [0.2], [0.5]], dtype=torch.float32, device='cuda')
    w = torch.tensor([[1000.0]], dtype=torch.float32, device='cuda')
    c1, p1 = projector(ct, ph, kernel=1).project_tables(mu, w, want_pathlen=True)
    c2, p2 = projector(ct, ph, kernel=2).project_tables(mu, w, want_pathlen=True)
    c3, p3 = projector(ct, ph, kernel=3).project_tables(mu, w, want_pathlen=True)
    assert torch.equal(p1, p2) and torch.equal(p1, p3)
    assert torch.allclose(c1, c2, rtol=1e-6, atol=0) and torch.allclose(c1, c3, rtol=1e-6, atol=0)
    tot = p1.sum(-1).double().cpu().numpy()[:, 0, :]
    b, gm = ct.thetas[:, None], ct.gammas[None, :]
    sx, sy = ct.SID * np.cos(b), ct.SID * np.sin(b)
    ex, ey = -np.cos(b + gm), -np.sin(b + gm)
    h = 0.5 * 256 * ph.dx
    with np.errstate(divide='ignore', invalid='ignore'):
        ax0, ax1 = (-h - sx) / ex, (h - sx) / ex
        ay0, ay1 = (-h - sy) / ey, (h - sy) / ey
    t0 = np.maximum(np.minimum(ax0, ax1), np.minimum(ay0, ay1))
    t1 = np.minimum(np.maximum(ax0, ax1), np.maximum(ay0, ay1))
    chord = np.maximum(t1 - t0, 0.0)
    assert np.max(np.abs(tot - chord)) < 2e-4
    lin = (p1.double() * mu[:, 0].double()).sum(-1)
    assert torch.max(torch.abs(-torch.log(c1[0].double() / 1000.0) - lin)) < 2e-5


def test_get_sino_surface(hip):
    """The reference call shape: two [N_proj, N_channels] arrays; log = ln(air / raw)."""
    import dex_ct_sim_amd as dx
    ct, ph = small_scan(n=32, n_views=20, n_channels=48)
    sp = spectra()[0]
    raw, log = dx.get_sino(ct, ph, sp)
    assert raw.shape == (20, 48) and log.shape == (20, 48) and raw.dtype == np.float32
    from dex_ct_sim_amd import forward_project as fp
    air = fp.effective_weights(ct, sp).sum()
    assert np.allclose(log, np.log(air / raw.astype(np.float64)), rtol=1e-5, atol=1e-6)
    both = dx.get_sinos(ct, ph, spectra())
    assert np.array_equal(both[0][0], raw)          # fused dual-spectrum traversal = single-spectrum result


def test_quantum_noise_statistics_and_reproducibility(hip):
    """noise=True: mean = noise-free sinogram, variance = sum_e gain^2 lambda_e (compound Poisson), same seed ->
    same sample, view shards reproduce the unsharded sample (counter-based RNG), either layout."""
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import forward_project as fp
    ct, ph = small_scan(n=32, nz=64, n_views=40, n_channels=64, n_rows=64)
    sp = spectra()
    for s in sp:
        s.rescale_counts(1e2)           # 1e8 photons per ray: Gaussian regime even behind 40 cm of water,
                                        # noise still far above float32 rounding of the mean
    clean, _ = projector(ct, ph, kernel=3).project(sp)
    n1, _ = projector(ct, ph, kernel=3).project(sp, noise=True, seed=7)
    n1b, _ = projector(ct, ph, kernel=1).project(sp, noise=True, seed=7)      # other kernel, other native layout
    n2, _ = projector(ct, ph, kernel=3).project(sp, noise=True, seed=8)
    n7, _ = projector(ct, ph, kernel=7).project(sp, noise=True, seed=7)       # rows16_kernel draws the sample itself
    assert torch.equal(n7, n1)
    assert not torch.equal(n1, clean) and not torch.equal(n1, n2)
    assert torch.allclose(n1, n1b, rtol=2e-6, atol=0)
    a, _ = projector(ct, ph, view_range=(0, 17), kernel=3).project(sp, noise=True, seed=7)
    b, _ = projector(ct, ph, view_range=(17, 40), kernel=3).project(sp, noise=True, seed=7)
    assert torch.equal(torch.cat([a, b], dim=1), n1)
    # standardise with the variance predicted by the float64 oracle (same weights2 through the classic Siddon)
    E, mu, w, w2 = fp.merged_tables(ct, ph, sp, with_variance=True)
    g = oracle_geom(ct, ph)
    var = co.project_classic(g, ct.view_cs(), ct.chan_cs(), 0, ct.N_proj, ph.volume, mu, w2, n_threads=8)
    zs = ((n1 - clean).double().cpu().numpy()) / np.sqrt(var)
    assert abs(zs.mean()) < 0.01 and abs(zs.std() - 1.0) < 0.01
    assert abs(np.mean(zs ** 3)) < 0.05 and abs(np.mean(zs ** 4) - 3.0) < 0.1     # normal sample
    raw, log = dx.get_sino(ct, ph, sp[0], noise=True, seed=7)
    assert raw.shape == (40, 64, 64) and np.isfinite(log).all()
    assert np.array_equal(raw, n1[0].cpu().numpy())


def test_native_layout_plus_transpose_equals_reference_layout(hip):
    """bench.py's pipeline shape: native (row-fastest) projection, then dexct_transpose_batched, equals the
    direct reference-order projection; pathlen likewise."""
    from dex_ct_sim_amd import _native
    from dex_ct_sim_amd._device import ptr, stream_ptr
    ct, ph = small_scan(n=40, nz=16, n_views=12, n_channels=50, n_rows=16)
    pj = projector(ct, ph, kernel=3)
    sp = spectra()
    _, mu_d, w_d, _ = pj.upload_tables(sp)
    ref = pj.project_tables(mu_d, w_d, layout=0)
    nat = pj.project_tables(mu_d, w_d, layout=None)
    assert pj.native_layout == 1 and nat.shape == (2, 12, 50, 16) and ref.shape == (2, 12, 16, 50)
    assert torch.equal(nat.permute(0, 1, 3, 2), ref)
    out = torch.empty_like(ref)
    _native.check(pj.lib.dexct_transpose_batched(ptr(nat), ptr(out), 2 * 12, 50, 16, 4, stream_ptr()), 'transpose')
    assert torch.equal(out, ref)
    a = torch.randn((12, 50, 16, 2), dtype=torch.float64, device='cuda')
    b = torch.empty((12, 16, 50, 2), dtype=torch.float64, device='cuda')
    _native.check(pj.lib.dexct_transpose_batched(ptr(a), ptr(b), 12, 50, 16, 16, stream_ptr()), 'transpose')
    assert torch.equal(b, a.permute(0, 2, 1, 3))


@pytest.mark.parametrize('rows,cols', [(64, 64), (100, 72), (800, 512), (52, 16), (37, 16), (40, 50), (3, 5)])
def test_transpose_with_the_log_sinogram_in_one_pass(hip, rows, cols):
    """dexct_transpose_log: [S n][rows][cols] counts -> [S n][cols][rows] and ln(air[s] / counts) of the transposed counts from
    the same registers - bit for bit the generic transpose followed by dexct_sino_log (the detection store's arithmetic), for
    tile-aligned, ragged and non-vectorisable shapes, with and without the log, and on a misaligned view."""
    import ctypes as C
    from dex_ct_sim_amd import _native
    from dex_ct_sim_amd._device import ptr, stream_ptr
    lib = _native.load()
    S, n = 2, 3
    gen = torch.Generator(device='cuda').manual_seed(rows * 1000 + cols)
    src = torch.rand((S, n, rows, cols), generator=gen, device='cuda') * 1e5 + 1e-3
    src[0, 0, 0, 0] = 0.0                                                    # counts == 0 -> +inf like the NumPy expression
    air = (C.c_float * S)(3.0e5, 1.5e5)
    want = src.permute(0, 1, 3, 2).contiguous()
    want_log = torch.empty_like(want)
    _native.check(lib.dexct_sino_log(ptr(want), air, S, n * rows * cols, ptr(want_log), stream_ptr()), 'dexct_sino_log')
    dst, log = torch.zeros_like(want), torch.zeros_like(want)
    _native.check(lib.dexct_transpose_log(ptr(src), ptr(dst), ptr(log), air, S, n, rows, cols, stream_ptr()), 'dexct_transpose_log')
    assert torch.equal(dst, want) and torch.equal(log, want_log) and torch.isinf(log[0, 0, 0, 0])
    only = torch.zeros_like(want)
    _native.check(lib.dexct_transpose_log(ptr(src), ptr(only), None, None, S, n, rows, cols, stream_ptr()), 'dexct_transpose_log')
    assert torch.equal(only, want)
    # a view that starts 4 bytes into its buffer: the 16-byte path does not apply, the values are the same
    buf = torch.zeros(src.numel() + 1, device='cuda')
    off = buf[1:].view_as(src).copy_(src)
    dst2, log2 = torch.zeros_like(want), torch.zeros_like(want)
    _native.check(lib.dexct_transpose_log(ptr(off), ptr(dst2), ptr(log2), air, S, n, rows, cols, stream_ptr()), 'dexct_transpose_log')
    assert torch.equal(dst2, want) and torch.equal(log2, want_log)
    assert lib.dexct_transpose_log(ptr(src), ptr(dst), ptr(log), None, S, n, rows, cols, stream_ptr()) < 0      # a log without air values


def test_more_than_512_slabs_per_ray(hip):
    """640 x 600 grid: rays cross up to 640 slabs, i.e. two staging passes of the packed-count kernel and
    more than 248 slabs between flushes of its byte counters; all kernels still equal the oracle bit for bit."""
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd.system import AIR, BONE, WATER
    rng = np.random.default_rng(9)
    vol = np.zeros((8, 600, 640), np.uint8)
    yy, xx = np.mgrid[0:600, 0:640]
    disc = (xx - 320) ** 2 + (yy - 300) ** 2 < 270 ** 2
    vol[:, disc] = 1
    blobs = rng.integers(0, 3, (8, 600 // 20, 640 // 20), dtype=np.uint8).repeat(20, axis=1).repeat(20, axis=2)
    vol = np.where(disc[None], np.maximum(blobs, 1), 0).astype(np.uint8)
    ph = dx.VoxelPhantom.from_array('wide', vol, [AIR, WATER, BONE], dx=0.08, dy=0.08, dz=0.08)
    ct = dx.FanBeamGeometry(N_channels=96, N_proj=24, gamma_fan=0.8230337, SID=60.0, SDD=100.0, N_rows=8)
    g = oracle_geom(ct, ph)
    mu = np.array([[0.0002, 0.0002], [0.2, 0.18], [0.6, 0.4]])
    w = np.array([[1e4, 2e4]])
    plan = co.plan(g, ct.view_cs(), ct.chan_cs(), 0, 24)
    assert plan['n_slabs'].max() > 600
    _, rpl = co.project_dda(g, ct.view_cs(), ct.chan_cs(), 0, 24, ph.volume, mu, w, True, n_threads=8)
    cls = co.project_classic(g, ct.view_cs(), ct.chan_cs(), 0, 24, ph.volume, mu, w, n_threads=8)
    for kernel in (1, 2, 3, 5, 6):
        c, pl = projector(ct, ph, kernel=kernel).project_tables(
            torch.tensor(mu, dtype=torch.float32, device='cuda'), torch.tensor(w, dtype=torch.float32, device='cuda'),
            want_pathlen=True)
        assert np.array_equal(pl.cpu().numpy(), rpl), kernel
        assert np.max(np.abs(c.cpu().numpy() - cls) / cls) < REL_TOL


@pytest.mark.parametrize('n_mat', [3, 6])
def test_cone_beam_matches_oracle(hip, n_mat):
    """True 3-D rays: per-material path lengths bit-identical to the oracle's mirror, counts
    within 1e-5 of the float64 3-D textbook Siddon; a flat 'cone' equals the stacked fan of that slice."""
    import dex_ct_sim_amd as dx
    ct, ph = small_scan(n=40, nz=24, n_views=20, n_channels=48, n_rows=10)
    if n_mat > 3:
        ph = ph_many(ph, n_mat)
    cone = dx.FanBeamGeometry(N_channels=48, N_proj=20, gamma_fan=0.8230337, SID=60.0, SDD=100.0, h_iso=0.8,
                              eid=True, detector_file=ct.detector_file, N_rows=10, cone=True, src_z=0.3)
    g = oracle_geom(cone, ph)
    sp = spectra()
    from dex_ct_sim_amd import forward_project as fp
    E, mu, w = fp.merged_tables(cone, ph, sp)
    _, rpl = co.project_cone(g, cone.view_cs(), cone.chan_cs(), 0, 20, cone.row_z(), 0.3, ph.volume, mu, w, dda=True,
                             n_threads=8)
    cls, _ = co.project_cone(g, cone.view_cs(), cone.chan_cs(), 0, 20, cone.row_z(), 0.3, ph.volume, mu, w, dda=False,
                             n_threads=8)
    # kernel 1: one thread per ray (any number of materials); kernel 2: the rows of a (view, channel) pair as lanes
    # with the shared in-plane records and one byte load per slab - <= 3 materials in one pass, more in one pass per group of
    # three + one detection pass (round 6) - the same bits
    for kernel in (1, 2):
        pjk = projector(cone, ph, kernel=kernel)
        assert pjk.cone_groups == (kernel == 2 and n_mat > 3)
        (counts, pl), _ = pjk.project(sp, want_pathlen=True)
        assert np.array_equal(pl.cpu().numpy(), rpl), kernel
        assert np.max(np.abs(counts.cpu().numpy() - cls) / cls) < REL_TOL
    # zero cone angle through the centre of slice 7 == the 2-D fan of slice 7
    flat = dx.FanBeamGeometry(N_channels=48, N_proj=20, gamma_fan=0.8230337, SID=60.0, SDD=100.0, h_iso=1.0,
                              eid=True, detector_file=ct.detector_file, N_rows=1, cone=True,
                              src_z=(7 + 0.5 - 12) * ph.dz)
    flat.row_z = lambda: np.array([flat.src_z])
    fan, ph7 = small_scan(n=40, nz=24, n_views=20, n_channels=48, n_rows=1, z_index=7)
    ph7.volume, ph7.materials = ph.volume, ph.materials
    a, _ = projector(flat, ph).project(sp)
    b, _ = projector(fan, ph7).project(sp)
    assert torch.allclose(a, b, rtol=3e-6, atol=0)
    # too steep a cone is refused
    from dex_ct_sim_amd._native import DexctError
    steep = dx.FanBeamGeometry(N_channels=48, N_proj=20, SID=60.0, SDD=100.0, h_iso=30.0, N_rows=4, cone=True)
    with pytest.raises(DexctError):
        projector(steep, ph).project(sp)


def test_per_bin_poisson_noise(hip):
    """noise='poisson': exact photon statistics.  High dose: standardised residuals ~ N(0, 1) against the
    oracle's compound-Poisson variance.  Photon-counting detector at very low dose: integer counts whose mean,
    variance and zero fraction follow Poisson(lambda_total); view shards reproduce the unsharded sample."""
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import forward_project as fp
    ct, ph = small_scan(n=32, nz=64, n_views=40, n_channels=64, n_rows=64)
    sp = spectra()
    for s in sp:
        s.rescale_counts(1e2)
    clean, _ = projector(ct, ph, kernel=3).project(sp)
    noisy, _ = projector(ct, ph, kernel=3).project(sp, noise='poisson', seed=11)
    E, mu, w, w2 = fp.merged_tables(ct, ph, sp, with_variance=True)
    g = oracle_geom(ct, ph)
    var = co.project_classic(g, ct.view_cs(), ct.chan_cs(), 0, ct.N_proj, ph.volume, mu, w2, n_threads=8)
    zs = (noisy - clean).double().cpu().numpy() / np.sqrt(var)
    assert abs(zs.mean()) < 0.01 and abs(zs.std() - 1.0) < 0.01
    a, _ = projector(ct, ph, view_range=(0, 17), kernel=3).project(sp, noise='poisson', seed=11)
    b, _ = projector(ct, ph, view_range=(17, 40), kernel=1).project(sp, noise='poisson', seed=11)
    assert torch.allclose(torch.cat([a, b], dim=1), noisy, rtol=1e-6, atol=0)
    # photon counting, ~3 photons per unattenuated ray
    pcd = dx.FanBeamGeometry(N_channels=64, N_proj=40, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=False, N_rows=64)
    lo = [dx.xRaySpectrum.from_arrays('lo', s.E, s.I0_raw * 3e-6) for s in sp]
    ph.volume[:] = 0                                   # air only: lambda is the same for every ray
    lam = np.array([fp.effective_weights(pcd, s).sum() for s in lo])
    mean_clean, _ = projector(pcd, ph, kernel=3).project(lo)
    cnt, _ = projector(pcd, ph, kernel=3).project(lo, noise='poisson', seed=3)
    c = cnt.cpu().numpy()
    lam_air = mean_clean.cpu().numpy().mean(axis=(1, 2, 3))          # includes the (tiny) attenuation by air
    assert np.allclose(lam_air, lam, rtol=2e-2)
    for s in range(2):
        x = c[s].ravel()
        x = np.where(x < 0.5, 0.0, x)                                # the 1e-20 floor stands for 0 photons
        assert np.all(np.abs(x - np.round(x)) < 1e-6)                # whole photons
        assert abs(x.mean() - lam_air[s]) < 0.02 * lam_air[s] and abs(x.var() - lam_air[s]) < 0.03 * lam_air[s]
        assert abs((x == 0).mean() - np.exp(-lam_air[s])) < 0.01


def on_plane_rays(g, ct, n_views):
    """[views, channels] mask of rays running exactly along a grid plane (slope 0, integer intercept): the one
    genuine tie, where the textbook algorithm and the DDA may each pick either neighbouring column
    (tests/test_siddon_oracle.py documents it)."""
    pl = co.plan(g, ct.view_cs(), ct.chan_cs(), 0, n_views)
    return ((pl['SV'] == 0) & (pl['V0'] % (1 << 40) == 0)).reshape(n_views, -1)


@pytest.mark.parametrize('seed', range(12))
def test_random_scans_bit_exact_path_lengths(hip, seed):
    """Randomised scanner / grid combinations: non-square anisotropic grids, odd sizes, off-centre slice ranges, 2..6
    materials, every kernel the host would pick or that can be forced.  Per-material path lengths equal the DDA
    form of the oracle bit for bit, counts agree with the float64 textbook Siddon to the north-star tolerance."""
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import forward_project as fp
    from dex_ct_sim_amd.system import AIR, BONE, WATER, Material
    from dex_ct_sim_amd._native import DexctError
    rng = np.random.default_rng(1000 + seed)
    nx, ny = int(rng.integers(17, 90)), int(rng.integers(17, 90))
    nz = int(rng.integers(1, 40))
    n_rows = int(rng.integers(1, nz + 1))
    z_index = int(rng.integers(0, nz - n_rows + 1))
    dxv, dyv, dzv = (float(v) for v in rng.uniform(0.05, 0.4, 3))
    half_diag = 0.5 * np.hypot(nx * dxv, ny * dyv)
    sid = float(half_diag * rng.uniform(1.15, 4.0))
    sdd = float(sid + half_diag * rng.uniform(1.0, 3.0))     # the detector clears the grid, like the source
    fan = float(rng.uniform(0.2, 2.4))                        # narrow fans, and fans wider than the grid (misses)
    n_views, n_ch = int(rng.integers(3, 40)), int(rng.integers(5, 130))
    n_mat = int(rng.integers(2, 7))
    vol = rng.integers(0, n_mat, (nz, ny, nx), dtype=np.uint8)
    vol[rng.random(vol.shape) < 0.5] = 0                     # half air, many material boundaries
    mats = ([AIR, WATER, BONE] + [Material(f'm{i}', 1.0 + 0.2 * i, 'H(11.2)O(88.8)') for i in range(3, n_mat)])[:n_mat]
    ph = dx.VoxelPhantom.from_array('rnd', vol, mats, dx=dxv, dy=dyv, dz=dzv, z_index=z_index)
    ct = dx.FanBeamGeometry(N_channels=n_ch, N_proj=n_views, gamma_fan=fan, SID=sid, SDD=sdd, N_rows=n_rows)
    g = co.make_geom(n_views, n_ch, n_rows, z_index, nx, ny, nz, dxv, dyv, dzv, sid, sdd)
    sp = spectra()
    E, mu, w = fp.merged_tables(ct, ph, sp)
    _, rpl = co.project_dda(g, ct.view_cs(), ct.chan_cs(), 0, n_views, vol, mu, w, True, n_threads=8)
    cls = co.project_classic(g, ct.view_cs(), ct.chan_cs(), 0, n_views, vol, mu, w, n_threads=8)
    tie = on_plane_rays(g, ct, n_views)[:, None, :].repeat(n_rows, 1)          # [views, rows, channels]
    for kernel in (0, 1, 2, 3, 4, 5, 6):
        try:
            pj = projector(ct, ph, kernel=kernel)
            (counts, pl), _ = pj.project(sp, want_pathlen=True)
        except (DexctError, ValueError):
            assert kernel in (3, 4, 5, 6)                     # packed / wave-per-ray kernels: material-count limits only
            continue
        assert np.array_equal(pl.cpu().numpy(), rpl), (seed, kernel)
        rel = np.abs(counts.cpu().numpy() - cls) / cls
        rel[:, tie] = 0.0
        assert rel.max() < REL_TOL, (seed, kernel, rel.max())


def test_source_and_detector_must_clear_the_grid(hip):
    """Line integrals run through the whole grid; a detector (or source) inside it is refused, not silently wrong."""
    import dex_ct_sim_amd as dx
    _, ph = small_scan(n=48)                                  # 51.2 cm grid: half diagonal 36.2 cm
    for sid, sdd in ((30.0, 100.0), (60.0, 80.0)):
        ct = dx.FanBeamGeometry(N_channels=32, N_proj=8, gamma_fan=0.5, SID=sid, SDD=sdd)
        with pytest.raises(ValueError):
            projector(ct, ph)


@pytest.mark.parametrize('seed', range(14))
def test_random_cone_beam_scans(hip, seed):
    """Randomised cone-beam scans (anisotropic grids, off-centre source heights, 2..6 materials): path lengths equal
    the oracle's mirror bit for bit, counts agree with the float64 3-D textbook Siddon."""
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import forward_project as fp
    from dex_ct_sim_amd.system import AIR, BONE, WATER, Material
    rng = np.random.default_rng(2000 + seed)
    nx, ny, nz = int(rng.integers(9, 60)), int(rng.integers(9, 60)), int(rng.integers(2, 40))
    dxv, dyv, dzv = (float(v) for v in rng.uniform(0.08, 0.4, 3))
    half_diag = 0.5 * np.hypot(nx * dxv, ny * dyv)
    sid = float(half_diag * rng.uniform(1.3, 4.0))
    sdd = float(sid + half_diag * rng.uniform(1.0, 3.0))
    n_rows, n_views, n_ch = int(rng.integers(1, 24)), int(rng.integers(3, 24)), int(rng.integers(5, 90))
    # the kernel takes at most one z-plane per dominant-axis slab: |dz/du| <= 1 for every ray
    reach = 0.6 * sdd * dzv / (max(dxv, dyv) * np.sqrt(2.0))
    src_z = float(rng.uniform(-0.3, 0.3) * reach)
    half_rows = max(0.5 * (n_rows - 1), 0.5)
    h_iso = float((reach - abs(src_z)) / half_rows * sid / sdd * rng.uniform(0.2, 1.0))
    n_mat = int(rng.integers(2, 7))
    vol = rng.integers(0, n_mat, (nz, ny, nx), dtype=np.uint8)
    vol[rng.random(vol.shape) < 0.4] = 0
    mats = ([AIR, WATER, BONE] + [Material(f'm{i}', 1.0 + 0.2 * i, 'H(11.2)O(88.8)') for i in range(3, n_mat)])[:n_mat]
    ph = dx.VoxelPhantom.from_array('rnd', vol, mats, dx=dxv, dy=dyv, dz=dzv)
    cone = dx.FanBeamGeometry(N_channels=n_ch, N_proj=n_views, gamma_fan=float(rng.uniform(0.2, 1.6)), SID=sid, SDD=sdd,
                              h_iso=h_iso, N_rows=n_rows, cone=True, src_z=src_z)
    g = co.make_geom(n_views, n_ch, n_rows, 0, nx, ny, nz, dxv, dyv, dzv, sid, sdd)
    sp = spectra()
    E, mu, w = fp.merged_tables(cone, ph, sp)
    _, rpl = co.project_cone(g, cone.view_cs(), cone.chan_cs(), 0, n_views, cone.row_z(), src_z, vol, mu, w, dda=True,
                             n_threads=8)
    cls, _ = co.project_cone(g, cone.view_cs(), cone.chan_cs(), 0, n_views, cone.row_z(), src_z, vol, mu, w, dda=False,
                             n_threads=8)
    for kernel in ((1, 2) if n_mat <= 3 else (1,)):
        (counts, pl), _ = projector(cone, ph, kernel=kernel).project(sp, want_pathlen=True)
        assert np.array_equal(pl.cpu().numpy(), rpl), (seed, kernel)
        rel = np.abs(counts.cpu().numpy() - cls) / cls
        rel[:, on_plane_rays(g, cone, n_views)[:, None, :].repeat(n_rows, 1)] = 0.0
        assert rel.max() < REL_TOL, (seed, kernel, rel.max())


@pytest.mark.parametrize('nz,n_rows', [(300, 300), (520, 40), (130, 257), (1030, 12)])
def test_cone_row_kernels_on_tall_volumes(hip, nz, n_rows, monkeypatch):
    """The row-parallel cone kernels on volumes and detectors the random scans above do not reach: columns of more
    than 288 and than 544 bytes (cone_cols_kernel<.., 544>, <.., 1056>), more than 1024 slices (falls back to
    cone_rows_kernel), more than one chunk of 256 rows; and the A/B forms (DEXCT_CONE_COLS=0, DEXCT_CONE_KB=8, a view tile) - path lengths equal the
    oracle's mirror bit for bit in every form, counts identical between the forms."""
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import forward_project as fp
    from dex_ct_sim_amd.system import AIR, BONE, WATER
    rng = np.random.default_rng(nz * 1000 + n_rows)
    nx, ny = 20, 17
    dxv, dyv, dzv = 0.2, 0.25, 0.05
    vol = rng.integers(0, 3, (nz, ny, nx), dtype=np.uint8)
    vol[rng.random(vol.shape) < 0.5] = 0
    ph = dx.VoxelPhantom.from_array('tall', vol, [AIR, WATER, BONE], dx=dxv, dy=dyv, dz=dzv)
    sid, sdd = 9.0, 16.0
    reach = 0.6 * sdd * dzv / (max(dxv, dyv) * np.sqrt(2.0))
    h_iso = float(reach / (0.5 * (n_rows - 1)) * sid / sdd * 0.9)
    n_views, n_ch = 5, 11
    cone = dx.FanBeamGeometry(N_channels=n_ch, N_proj=n_views, gamma_fan=0.9, SID=sid, SDD=sdd, h_iso=h_iso,
                              N_rows=n_rows, cone=True, src_z=0.1 * reach)
    g = co.make_geom(n_views, n_ch, n_rows, 0, nx, ny, nz, dxv, dyv, dzv, sid, sdd)
    sp = spectra()
    E, mu, w = fp.merged_tables(cone, ph, sp)
    _, rpl = co.project_cone(g, cone.view_cs(), cone.chan_cs(), 0, n_views, cone.row_z(), 0.1 * reach, vol, mu, w,
                             dda=True, n_threads=8)
    first = None
    for env in ({}, {'DEXCT_CONE_COLS': '0'}, {'DEXCT_CONE_KB': '8'}, {'DEXCT_CONE_VIEW_TILE': '3'}):
        for k in ('DEXCT_CONE_COLS', 'DEXCT_CONE_KB', 'DEXCT_CONE_VIEW_TILE'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        (counts, pl), _ = projector(cone, ph, kernel=2).project(sp, want_pathlen=True)
        assert np.array_equal(pl.cpu().numpy(), rpl), env
        if first is None:
            first = counts.clone()
        assert torch.equal(counts, first), env


@pytest.mark.parametrize('n_mat,nz,n_rows', [(4, 40, 36), (6, 300, 64), (12, 96, 257), (50, 64, 40), (5, 1030, 33)])
def test_cone_beam_material_groups(hip, n_mat, nz, n_rows, monkeypatch):
    """Round 6: more than 3 table rows on the row-parallel cone kernels - one cone_cols_kernel (cone_rows_kernel beyond 1024
    slices) pass per group of three materials into planes of path lengths, one detection pass over them
    (dexct_cone_project_grouped), as the stacked fan has had since round 3.  Path lengths bit-identical to the one-thread-per-ray
    kernel and to the oracle's mirror (the per-material sums are independent), counts identical for <= 48 table rows (the same
    detection arithmetic) and within 1e-5 of the float64 3-D Siddon; log sinogram, quantum noise (the same sample as kernel 1
    draws), view shards, view chunks of the scratch (DEXCT_GROUP_SCRATCH), what kernel 0 picks."""
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import forward_project as fp
    from dex_ct_sim_amd.system import AIR, BONE, WATER, Material
    rng = np.random.default_rng(n_mat * 100 + nz)
    nx, ny = 22, 19
    dxv, dyv, dzv = 0.2, 0.25, 0.05
    vol = rng.integers(0, n_mat, (nz, ny, nx), dtype=np.uint8)
    vol[rng.random(vol.shape) < 0.4] = 0
    mats = [AIR, WATER, BONE] + [Material(f'm{i}', 0.8 + 0.03 * i, 'H(11.2)O(88.8)' if i % 2 else 'H(10.2)C(14.3)N(3.4)O(71.0)Na(0.1)') for i in range(3, n_mat)]
    ph = dx.VoxelPhantom.from_array('groups', vol, mats, dx=dxv, dy=dyv, dz=dzv)
    sid, sdd = 9.0, 16.0
    reach = 0.6 * sdd * dzv / (max(dxv, dyv) * np.sqrt(2.0))
    h_iso = float(reach / (0.5 * (n_rows - 1)) * sid / sdd * 0.9)
    n_views, n_ch = 6, 13
    cone = dx.FanBeamGeometry(N_channels=n_ch, N_proj=n_views, gamma_fan=0.9, SID=sid, SDD=sdd, h_iso=h_iso,
                              N_rows=n_rows, cone=True, src_z=0.1 * reach)
    g = co.make_geom(n_views, n_ch, n_rows, 0, nx, ny, nz, dxv, dyv, dzv, sid, sdd)
    sp = spectra()
    for s_ in sp:
        s_.rescale_counts(1e2)
    E, mu, w = fp.merged_tables(cone, ph, sp)
    pj1, pj2, pj0 = projector(cone, ph, kernel=1), projector(cone, ph, kernel=2), projector(cone, ph)
    assert pj2.cone_groups and pj0.cone_groups and not pj1.cone_groups          # kernel 0 picks the group passes from 32 rows on
    M = pj2.n_mat
    _, rpl = co.project_cone(g, cone.view_cs(), cone.chan_cs(), 0, n_views, cone.row_z(), 0.1 * reach, vol, pj2.compact(mu), w,
                             dda=True, n_threads=8) if M == n_mat else (None, None)
    (c1, p1, l1), air = pj1.project(sp, want_pathlen=True, want_log=True)
    (c2, p2, l2), _ = pj2.project(sp, want_pathlen=True, want_log=True)
    assert torch.equal(p1, p2)
    if rpl is not None:
        assert np.array_equal(p2.cpu().numpy(), rpl)
    same_detection = 4 < M <= 48     # (4 rows: kernel 1 sums its energies in pairs; beyond 48 the detection pass forms its exponents
    if same_detection:               # unscaled: another rounding, the same numbers to 2e-6)
        assert torch.equal(c1, c2) and torch.equal(l1, l2)
    else:
        assert torch.allclose(c1, c2, rtol=2e-6, atol=0)
    cls, _ = co.project_cone(g, cone.view_cs(), cone.chan_cs(), 0, n_views, cone.row_z(), 0.1 * reach, vol, mu, w, dda=False, n_threads=8)
    assert np.max(np.abs(c2.cpu().numpy() - cls) / cls) < REL_TOL
    n1, _ = pj1.project(sp, noise=True, seed=4)
    (n2, nl2), _ = pj2.project(sp, noise=True, seed=4, want_log=True)
    if M <= 48:                   # the detection pass of the groups draws the sample itself: what dexct_add_noise draws from its variance
        from dex_ct_sim_amd import _native
        from dex_ct_sim_amd._device import ptr, stream_ptr
        _, mu_v, w_v, w2_v = fp.merged_tables(cone, ph, sp, with_variance=True)
        tabs = [torch.tensor(x, dtype=torch.float32, device='cuda').contiguous() for x in (pj2.compact(mu_v), w_v, w2_v)]
        nv, var = pj2.project_tables(tabs[0], tabs[1], w2_d=tabs[2], seed=4, want_variance=True)
        sampled = c2.clone()
        _native.check(pj2.lib.dexct_add_noise(ptr(sampled), ptr(var), 2, n_views, n_rows, n_ch, 0, 0, 4, stream_ptr()), 'dexct_add_noise')
        assert torch.equal(nv, n2) and torch.equal(sampled, n2)
    if same_detection:
        assert torch.equal(n1, n2)
    assert not torch.equal(n2, c2) and np.allclose(nl2.cpu().numpy(), _np_log(air, n2.cpu().numpy()), rtol=5e-6, atol=5e-7)
    a, _ = projector(cone, ph, kernel=2, view_range=(0, 2)).project(sp)
    b, _ = projector(cone, ph, kernel=2, view_range=(2, 6)).project(sp)
    assert torch.equal(torch.cat([a, b], dim=1), c2)
    monkeypatch.setattr(fp, '_GROUP_SCRATCH_BYTES', 2 * M * n_rows * n_ch * 4)      # two views of scratch: three chunks
    (c3, p3, l3), _ = pj2.project(sp, want_pathlen=True, want_log=True)
    assert torch.equal(c3, c2) and torch.equal(p3, p2) and torch.equal(l3, l2)
    n3, _ = pj2.project(sp, noise=True, seed=4)
    assert torch.equal(n3, n2)


def test_sino_allgather_entry_point_single_rank(hip):
    """dexct_sino_allgather with a one-rank RCCL communicator created through the RCCL copy torch has loaded: the
    gathered buffer equals the shard, out of place and in place (the 8-rank use is the driver's; ranks > 1 are
    covered by the gloo tests of _shard.py)."""
    import ctypes as C
    import os
    from dex_ct_sim_amd import _native
    from dex_ct_sim_amd._device import ptr, stream_ptr
    lib = _native.load()
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so'))
    comm = C.c_void_p()
    dev = (C.c_int * 1)(0)
    assert rccl.ncclCommInitAll(C.byref(comm), 1, dev) == 0
    try:
        a = torch.arange(100000, dtype=torch.float32, device='cuda') * 0.5
        out = torch.zeros_like(a)
        _native.check(lib.dexct_sino_allgather(ptr(a), ptr(out), a.numel(), comm, stream_ptr()), 'dexct_sino_allgather')
        torch.cuda.synchronize()
        assert torch.equal(out, a)
        _native.check(lib.dexct_sino_allgather(ptr(out), ptr(out), out.numel(), comm, stream_ptr()), 'in place')
        torch.cuda.synchronize()
        assert torch.equal(out, a)
        assert lib.dexct_sino_allgather(ptr(a), ptr(out), 0, comm, None) == -1
        assert lib.dexct_sino_allgather(ptr(a), ptr(out), 10, None, None) == -1
    finally:
        rccl.ncclCommDestroy(comm)


def test_degenerate_shapes_through_the_public_calls(hip):
    """1-2 views, 1-3 channels, 1-voxel phantoms, 1-64 rows through get_sinos / get_basismat_sinos / get_recon:
    right shapes, finite positive counts, no crash."""
    import itertools
    import warnings
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd.system import AIR, BONE, WATER
    hi, lo = spectra()
    rng = np.random.default_rng(0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for n_proj, n_ch, n, nz, rows in itertools.product((1, 2, 5), (1, 2, 65), (1, 2, 7), (1, 4, 64), (1, 4, 64)):
            if rows > nz:
                continue
            ct = dx.FanBeamGeometry(N_channels=n_ch, N_proj=n_proj, gamma_fan=0.5, SID=60.0, SDD=100.0, N_rows=rows)
            ph = dx.VoxelPhantom.from_array('t', rng.integers(0, 3, (nz, n, n), dtype=np.uint8), [AIR, WATER, BONE], dx=0.5)
            (r1, l1), (r2, _) = dx.get_sinos(ct, ph, [hi, lo])
            expect = (n_proj, n_ch) if rows == 1 else (n_proj, rows, n_ch)
            assert r1.shape == expect and np.isfinite(r1).all() and (r1 > 0).all(), (n_proj, n_ch, n, nz, rows)
            m1, m2 = dx.get_basismat_sinos(ct, r1, r2, hi, lo, n_iters=5)
            assert m1.shape == expect and m2.shape == expect
            if n_ch >= 2:
                img, _ = dx.get_recon(l1, ct, hi, 8, 10.0, 1.0)
                assert img.shape == ((8, 8) if rows == 1 else (rows, 8, 8))


@pytest.mark.parametrize('n_rows,nz,z_index', [(256, 256, 0), (256, 270, 2), (512, 512, 0), (1024, 1024, 0), (2048, 2048, 0),
                                               (200, 203, 1), (320, 320, 0), (768, 768, 0), (1100, 1100, 0), (50, 64, 3)])
@pytest.mark.parametrize('n_mat', [2, 3, 4])
def test_packed_volume_kernel_bit_identical(hip, n_rows, nz, z_index, n_mat):
    """rows16_kernel (kernel 7): 2 bits per voxel, 16 rows per lane, bit-sliced counters.  4, 2 or 1 (view, channel)
    pairs per wave (256 / 512 / >= 1024 rows), ragged channel groups (53 channels), an unaligned first slice (the host
    pads the uploaded volume to multiples of 16): per-material path lengths bit-identical to rows4_kernel and to the
    oracle mirror, counts equal to rows4_kernel's (same detection code) and within 1e-5 of the float64 Siddon."""
    from dex_ct_sim_amd import forward_project as fp
    from dex_ct_sim_amd.system import AIR, WATER
    ct, ph = small_scan(n=40, nz=nz, n_views=7, n_channels=53, n_rows=n_rows, z_index=z_index)
    if n_mat == 2:
        ph.volume = np.minimum(ph.volume, 1).astype(np.uint8)
        ph.materials = [AIR, WATER]
    if n_mat == 4:
        ph = ph_many(ph, 4)                                          # id 3 sets both flags: the third set of counters
    rng = np.random.default_rng(nz + n_mat)
    speck = rng.random(ph.volume.shape) < 0.02                      # isolated voxels: corrections in many rows
    ph.volume[speck] = rng.integers(0, n_mat, int(speck.sum()), dtype=np.uint8)
    sp = spectra()
    (c3, p3), _ = projector(ct, ph, kernel=3).project(sp, want_pathlen=True)
    (c7, p7), _ = projector(ct, ph, kernel=7).project(sp, want_pathlen=True)
    assert torch.equal(p7, p3)
    assert torch.equal(c7, c3)
    g = oracle_geom(ct, ph)
    E, mu, w = fp.merged_tables(ct, ph, sp)
    sub = co.make_geom(ct.N_proj, ct.N_channels, 16, z_index + n_rows - 16, ph.Nx, ph.Ny, ph.Nz, ph.dx, ph.dy, ph.dz,
                       ct.SID, ct.SDD)                               # the last 16 rows
    _, rpl = co.project_dda(sub, ct.view_cs(), ct.chan_cs(), 0, ct.N_proj, ph.volume, mu, w, True, n_threads=8)
    assert np.array_equal(p7[:, n_rows - 16:].cpu().numpy(), rpl)
    cls = co.project_classic(sub, ct.view_cs(), ct.chan_cs(), 0, ct.N_proj, ph.volume, mu, w, n_threads=8)
    assert np.max(np.abs(c7[:, :, n_rows - 16:].cpu().numpy() - cls) / cls) < REL_TOL


def dx_cone(ct):
    import dex_ct_sim_amd as dx
    return dx.FanBeamGeometry(N_channels=ct.N_channels, N_proj=ct.N_proj, gamma_fan=0.8230337, SID=60.0, SDD=100.0, h_iso=0.1,
                              N_rows=8, cone=True)


def test_packed_volume_kernel_refuses_what_it_cannot_do(hip):
    ct, ph = small_scan(n=40, nz=256, n_views=4, n_channels=16, n_rows=256)
    with pytest.raises(ValueError):
        projector(ct, ph_many(ph, 5), kernel=7)                      # ids above 2
    cone = dx_cone(ct)
    with pytest.raises(ValueError):
        projector(cone, ph, kernel=7)                                # not a stacked fan


def test_packed_volume_is_the_default_with_and_without_noise(hip):
    """kernel=0 picks the packed-volume kernel for <= 3 materials and 256 rows; with quantum noise too (round 6: the kernel
    sums the variance and draws the sample itself) and gives the sample the byte-volume kernel + dexct_add_noise give for the
    same seed."""
    ct, ph = small_scan(n=40, nz=256, n_views=6, n_channels=40, n_rows=256)
    sp = spectra()
    auto = projector(ct, ph)
    assert auto.use_packed and auto.vol_z2 is not None and auto.native_layout == 1
    clean0, _ = auto.project(sp)
    clean3, _ = projector(ct, ph, kernel=3).project(sp)
    assert torch.equal(clean0, clean3)
    n0, _ = auto.project(sp, noise=True, seed=5)
    n3, _ = projector(ct, ph, kernel=3).project(sp, noise=True, seed=5)
    assert torch.equal(n0, n3) and not torch.equal(n0, clean0)
    ct2, ph2 = small_scan(n=40, nz=96, n_views=6, n_channels=40, n_rows=96)          # 96 rows: 6 of 16 lanes: byte-volume kernel
    assert not projector(ct2, ph2).use_packed
    ct3, ph3 = small_scan(n=40, nz=320, n_views=6, n_channels=40, n_rows=320)        # 320 rows: 20 of 32 lanes: byte-volume kernel
    assert not projector(ct3, ph3).use_packed
    ct4, ph4 = small_scan(n=40, nz=768, n_views=6, n_channels=40, n_rows=768)        # 768 rows: 48 of 64 lanes: packed
    assert projector(ct4, ph4).use_packed


@pytest.mark.parametrize('n_rows,nz,z_index,n_mat,n_spec,staged', [
    (256, 256, 0, 3, 2, '1'), (256, 256, 0, 3, 2, '0'), (512, 512, 0, 3, 1, '1'), (66, 70, 2, 3, 2, '1'), (200, 208, 0, 2, 2, '1'),
    (1024, 1024, 0, 4, 2, '1'), (2048, 2048, 0, 3, 2, '1'), (64, 64, 0, 4, 1, '0')])
def test_noisy_scan_on_the_packed_kernel_equals_the_byte_volume_path(hip, n_rows, nz, z_index, n_mat, n_spec, staged, monkeypatch):
    """Round 6: rows16_kernel<NOISY> (the variance summed in the detection rounds, the sample drawn in registers, the log of
    the sampled counts from the same store) against rows4_kernel's variance output + dexct_add_noise + dexct_sino_log: the
    noisy counts, the variances and the log sinogram are the same bits - same Philox keys, same detection arithmetic.
    Staged whole-line stores and the per-round stores, ragged rows (66 of 70 from slice 2: the host pads; 200: idle lanes),
    4 / 2 / 1 pairs per wave and two z-chunks, 2 - 4 materials, one and two spectra; also in the reference's order
    (transpose + log in one pass) and without the sample (variance output only)."""
    from dex_ct_sim_amd import _native, forward_project as fp
    from dex_ct_sim_amd._device import ptr, stream_ptr
    from dex_ct_sim_amd.system import AIR, WATER
    monkeypatch.setenv('DEXCT_P16_STAGED', staged)
    ct, ph = small_scan(n=40, nz=nz, n_views=7, n_channels=23, n_rows=n_rows, z_index=z_index)
    if n_mat == 2:
        ph.volume = np.minimum(ph.volume, 1).astype(np.uint8)
        ph.materials = [AIR, WATER]
    elif n_mat == 4:
        ph = ph_many(ph, 4)
    sp = spectra()[:n_spec]
    for s_ in sp:
        s_.rescale_counts(1e-2)
    _, mu, w, w2 = fp.merged_tables(ct, ph, sp, with_variance=True)
    air = w.sum(axis=1)
    res = {}
    for k in (7, 3):
        pj = projector(ct, ph, kernel=k)
        mu_d, w_d, w2_d = (torch.tensor(x, dtype=torch.float32, device='cuda').contiguous() for x in (pj.compact(mu), w, w2))
        res[k] = pj.project_tables(mu_d, w_d, layout=None, w2_d=w2_d, seed=9, air=air, want_variance=True)
        res[k, 'ref'] = pj.project_tables(mu_d, w_d, layout=0, w2_d=w2_d, seed=9, air=air)
        res[k, 'clean'] = pj.project_tables(mu_d, w_d, layout=None)
        w_t = w_d.t().contiguous().t()                      # a column-major view of the same table (what torch.tensor(w[:, keep]) is)
        assert (n_spec == 1 or not w_t.is_contiguous()) and torch.equal(pj.project_tables(mu_d, w_t, layout=None), res[k, 'clean'])
        if k == 7:
            assert pj.use_packed
            # the variance alone (no sample): counts stay the expectation
            counts = torch.empty_like(res[7][0])
            var = torch.empty_like(counts)
            _native.check(pj.lib.dexct_siddon_project_packed(
                C_byref(pj.geom), pj.plan.data_ptr(), 0, ct.N_proj, ptr(pj.vol_z2), pj.n_mat, w_d.shape[1], n_spec, ptr(mu_d),
                ptr(w_d), ptr(counts), None, 1, None, ptr(w2_d), ptr(var), None, stream_ptr()), 'packed, variance only')
            assert torch.equal(counts, res[7, 'clean']) and torch.equal(var, res[7][2])
    for a, b in zip(res[7], res[3]):
        assert torch.equal(a, b)                                                 # noisy counts, log, variance
    for a, b in zip(res[7, 'ref'], res[3, 'ref']):
        assert torch.equal(a, b)
    assert torch.equal(res[7, 'ref'][0], res[7][0].permute(0, 1, 3, 2))
    assert torch.equal(res[7, 'clean'], res[3, 'clean']) and not torch.equal(res[7][0], res[7, 'clean'])
    assert torch.isfinite(res[7][1]).all() and (res[7][2] > 0).all()
    assert torch.equal(res[7][1], projector(ct, ph, kernel=7).sino_log(res[7][0], air))


def test_packed_noise_arguments_are_checked(hip):
    """struct dexct_noise: weights2 needs somewhere for the variance to go; a variance or a sample needs weights2; the log of a
    noisy sinogram needs the sample; at most two spectra."""
    from dex_ct_sim_amd import _native, forward_project as fp
    from dex_ct_sim_amd._device import ptr, stream_ptr
    ct, ph = small_scan(n=32, nz=64, n_views=4, n_channels=16, n_rows=64)
    pj = projector(ct, ph, kernel=7)
    sp = spectra()
    _, mu, w, w2 = fp.merged_tables(ct, ph, sp, with_variance=True)
    mu_d, w_d, w2_d = (torch.tensor(x, dtype=torch.float32, device='cuda').contiguous() for x in (pj.compact(mu), w, w2))
    counts = torch.empty((2, 4, 16, 64), dtype=torch.float32, device='cuda')
    var, log = torch.empty_like(counts), torch.empty_like(counts)

    def call(w2p, varp, nz, lo=None, n_spec=2):
        return pj.lib.dexct_siddon_project_packed(C_byref(pj.geom), pj.plan.data_ptr(), 0, 4, ptr(pj.vol_z2), pj.n_mat, w_d.shape[1],
                                                  n_spec, ptr(mu_d), ptr(w_d), ptr(counts), None, 1, lo, w2p, varp, nz, stream_ptr())
    assert call(ptr(w2_d), None, None) == -1                       # nowhere to go
    assert call(None, ptr(var), None) == -1 and call(None, None, _native.noise(3)) == -1
    assert call(ptr(w2_d), ptr(var), None, _native.log_out(ptr(log), [1.0, 1.0])) == -1       # the log needs the sample
    assert call(ptr(w2_d), None, _native.noise(3, sample=False)) == -1
    assert call(ptr(w2_d), ptr(var), _native.noise(3), _native.log_out(ptr(log), [1.0, 1.0])) == 0
    assert call(ptr(w2_d), None, _native.noise(3)) == 0


@pytest.mark.parametrize('seed', range(8))
def test_random_packed_volume_scans(hip, seed):
    """Randomised stacked fans through RANDOM volumes (a material boundary in nearly every crossing slab and every
    row: the correction path of rows16_kernel runs all the time), anisotropic non-square grids, fans wider than the
    grid, 256 / 512 / 1024 rows, 2 or 3 materials: bit-identical path lengths and counts to rows4_kernel, and the
    last rows against the oracle mirror."""
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import forward_project as fp
    from dex_ct_sim_amd.system import AIR, BONE, WATER
    rng = np.random.default_rng(7000 + seed)
    nx, ny = int(rng.integers(9, 70)), int(rng.integers(9, 70))
    n_rows = int(rng.choice([256, 256, 512, 1024]))
    dxv, dyv, dzv = (float(v) for v in rng.uniform(0.1, 0.5, 3))
    half_diag = 0.5 * np.hypot(nx * dxv, ny * dyv)
    sid = float(half_diag * rng.uniform(1.2, 4.0))
    sdd = float(sid + half_diag * rng.uniform(1.05, 3.0))
    n_views, n_ch = int(rng.integers(2, 9)), int(rng.integers(3, 70))
    n_mat = int(rng.integers(2, 4))
    vol = rng.integers(0, n_mat, (n_rows, ny, nx), dtype=np.uint8)
    vol[rng.random(vol.shape) < 0.3] = 0
    ph = dx.VoxelPhantom.from_array('rnd', vol, [AIR, WATER, BONE][:n_mat], dx=dxv, dy=dyv, dz=dzv)
    ct = dx.FanBeamGeometry(N_channels=n_ch, N_proj=n_views, gamma_fan=float(rng.uniform(0.2, 1.8)), SID=sid, SDD=sdd,
                            N_rows=n_rows)
    sp = spectra()
    (c3, p3), _ = projector(ct, ph, kernel=3).project(sp, want_pathlen=True)
    pj = projector(ct, ph)
    assert pj.use_packed
    (c7, p7), _ = pj.project(sp, want_pathlen=True)
    assert torch.equal(p7, p3), seed
    assert torch.equal(c7, c3), seed
    E, mu, w = fp.merged_tables(ct, ph, sp)
    sub = co.make_geom(n_views, n_ch, 8, n_rows - 8, nx, ny, n_rows, dxv, dyv, dzv, sid, sdd)
    _, rpl = co.project_dda(sub, ct.view_cs(), ct.chan_cs(), 0, n_views, vol, mu, w, True, n_threads=8)
    assert np.array_equal(p7[:, n_rows - 8:].cpu().numpy(), rpl), seed


@pytest.mark.parametrize('n_e,n_s', [(140, 2), (139, 2), (300, 2), (64, 1), (7, 2), (3, 1)])
def test_detection_skips_only_exact_zeros(hip, n_e, n_s):
    """The row-parallel kernels skip the accumulations of a spectrum slot over blocks of four energies it does not
    weight (the 80 kVp spectrum above 80 keV; the absent second slot of a single-spectrum scan).  Arbitrary zero
    patterns (isolated blocks, a slot that weights nothing, -0.0, more than 256 energies, energy counts that are not
    multiples of 4): counts bit-identical to rays_kernel, which has no such shortcut, and equal to the float64 sum."""
    from dex_ct_sim_amd import forward_project as fp
    ct, ph = small_scan(n=40, nz=256, n_views=5, n_channels=37, n_rows=256)
    rng = np.random.default_rng(n_e * 10 + n_s)
    mu = rng.uniform(0.01, 0.4, (3, n_e)).astype(np.float32)
    mu[0] *= 1e-3
    w = rng.uniform(0.5, 2.0, (n_s, n_e)).astype(np.float32)
    for s in range(n_s):
        for b in range(0, n_e, 4):
            r = rng.random()
            if r < 0.35:
                w[s, b:b + 4] = 0.0                                   # a whole block
            elif r < 0.45:
                w[s, b:b + 2] = 0.0                                   # part of a block: not skippable
            elif r < 0.5:
                w[s, b:b + 4] = -0.0                                  # sign bit set: runs, adds nothing
    if n_s == 2 and n_e >= 64:
        w[1, 32:] = 0.0                                               # a slot that ends early
    dev = torch.device('cuda:0')
    mu_d, w_d = torch.from_numpy(mu).to(dev), torch.from_numpy(w).to(dev)
    ref, pl = projector(ct, ph, kernel=1).project_tables(mu_d, w_d, want_pathlen=True)
    for kernel in (3, 5, 7):
        got = projector(ct, ph, kernel=kernel).project_tables(mu_d, w_d)
        assert torch.equal(got, ref), kernel
    exact = np.einsum('se,...e->s...', w.astype(np.float64),
                      np.exp(-np.einsum('...m,me->...e', pl.cpu().numpy().astype(np.float64), mu.astype(np.float64))))
    scale = np.abs(w).sum(axis=1).reshape(-1, 1, 1, 1)
    assert np.max(np.abs(ref.cpu().numpy() - exact) / scale) < REL_TOL


@pytest.mark.parametrize('n_rows', [256, 512, 200])
@pytest.mark.parametrize('n_mat', [4, 5, 7, 10, 13, 20])
def test_material_groups_on_the_packed_volume(hip, n_rows, n_mat):
    """Kernel 8: one rows16_kernel pass per group of three materials on 2-bit group codes (a full group uses all four
    codes: the "both flags" counters), then the detection pass of the byte-volume group path.  Path lengths and counts
    bit-identical to kernel 4 (byte codes) and kernel 1; the host picks it for more than 4 materials where it picks the
    packed kernel for fewer; with noise it gives the sample kernel 4 gives."""
    ct, ph = small_scan(n=40, nz=n_rows, n_views=6, n_channels=45, n_rows=n_rows)
    ph = ph_many(ph, n_mat)
    rng = np.random.default_rng(n_rows + n_mat)
    speck = rng.random(ph.volume.shape) < 0.02
    ph.volume[speck] = rng.integers(0, n_mat, int(speck.sum()), dtype=np.uint8)
    sp = spectra()
    (c1, p1), _ = projector(ct, ph, kernel=1).project(sp, want_pathlen=True)
    (c4, p4), _ = projector(ct, ph, kernel=4).project(sp, want_pathlen=True)
    pj8 = projector(ct, ph, kernel=8)
    (c8, p8), _ = pj8.project(sp, want_pathlen=True)
    assert pj8.grouped_packed and pj8.codes.shape == ((n_mat + 1) // 3, ph.Nx * ph.Ny * pj8.geom.nz // 4)
    assert torch.equal(p8, p1) and torch.equal(p8, p4)
    assert torch.equal(c8, c4)
    assert float(((c8 - c1).abs() / c1).max()) < REL_TOL          # kernel 1 detects from LDS columns beyond 4 materials
    auto = projector(ct, ph)
    assert auto.grouped_packed == (n_mat > 4) and auto.use_packed == (n_mat <= 4)      # 200, 256, 512 rows fill their lane groups
    ca = auto.project(sp)[0]
    assert torch.equal(ca, c4) if n_mat > 4 else float(((ca - c4).abs() / c4).max()) < REL_TOL
    n8, _ = pj8.project(sp, noise=True, seed=3)
    n4, _ = projector(ct, ph, kernel=4).project(sp, noise=True, seed=3)
    assert torch.equal(n8, n4) and not torch.equal(n8, c8)
    # round 6: the detection pass of the groups draws the sample itself (variance from the same exponentials, one Philox block per
    # ray) - the sample dexct_add_noise draws from the pass's own signal and variance, and the log of the sampled counts
    from dex_ct_sim_amd import _native, forward_project as fp
    from dex_ct_sim_amd._device import ptr, stream_ptr
    _, mu, w, w2 = fp.merged_tables(ct, ph, sp, with_variance=True)
    mu_d, w_d, w2_d = (torch.tensor(x, dtype=torch.float32, device='cuda').contiguous() for x in (pj8.compact(mu), w, w2))
    air = w.sum(axis=1)
    noisy, log, var = pj8.project_tables(mu_d, w_d, layout=None, w2_d=w2_d, seed=3, air=air, want_variance=True)
    assert pj8.native_layout == 1 and torch.equal(noisy.permute(0, 1, 3, 2), n8)        # (n8: the reference's order)
    sampled = pj8.project_tables(mu_d, w_d, layout=None).clone()
    _native.check(pj8.lib.dexct_add_noise(ptr(sampled), ptr(var), 2, ct.N_proj, ct.N_rows, ct.N_channels, pj8.native_layout, 0, 3,
                                          stream_ptr()), 'dexct_add_noise')
    assert torch.equal(sampled, noisy) and torch.equal(log, pj8.sino_log(noisy, air))
    # four spectra: beyond the fused form - the variance output + dexct_add_noise, as before; the same sample for the same spectra
    four, _ = pj8.project(sp + sp, noise=True, seed=3)
    assert torch.equal(four[0], n8[0]) and torch.equal(four[1], n8[1]) and not torch.equal(four[2], four[0])


# ---- round 4: the full uint8 id range
def label_map_phantom(n, nz, n_ids=200, n_distinct=60, seed=5):
    """A label map in the style of an XCAT phantom: ids 1..n_ids-1 scattered over the body, n_distinct different
    (density, composition) pairs among them, some ids absent, id 0 = air around it."""
    from conftest import small_scan
    from dex_ct_sim_amd.system import AIR, Material
    _, ph = small_scan(n=n, nz=nz)
    rng = np.random.default_rng(seed)
    comps = ['H(11.2)O(88.8)', 'H(10.2)C(14.3)N(3.4)O(70.8)Na(0.2)P(0.3)S(0.3)Cl(0.2)K(0.3)',
             'H(3.4)C(15.5)N(4.2)O(43.5)Na(0.1)Mg(0.2)P(10.3)S(0.3)Ca(22.5)', 'H(11.4)C(59.8)N(0.7)O(27.8)Na(0.1)S(0.1)Cl(0.1)']
    distinct = [(round(0.3 + 0.03 * k, 3), comps[k % len(comps)]) for k in range(n_distinct - 1)]     # + air = n_distinct
    which = rng.integers(0, len(distinct), n_ids)
    mats = [AIR] + [Material(f'organ{i}', *distinct[which[i]]) for i in range(1, n_ids)]
    used = rng.permutation(np.arange(1, n_ids))[: n_ids - 20]                 # 19 ids of the table never occur
    body = ph.volume > 0
    # blocks of 4 x 4 x 2 voxels share an id (organs are not salt and pepper), plus 3 % single-voxel specks
    zz, yy, xx = np.meshgrid(np.arange(ph.Nz) // 2, np.arange(ph.Ny) // 4, np.arange(ph.Nx) // 4, indexing='ij')
    ids = used[(zz * 7919 + yy * 104729 + xx * 1299709) % used.size]
    speck = rng.random(ph.volume.shape) < (0.03 if nz > 1 else 0.6)          # (one 40 x 40 slice: mostly specks, to reach 49+ rows)
    ids = np.where(speck, used[rng.integers(0, used.size, ph.volume.shape)], ids)
    ph.volume = np.where(body, ids, 0).astype(np.uint8)
    ph.materials = mats
    return ph, distinct


@pytest.mark.parametrize('n_rows,nz,kernel', [(1, 1, 0), (1, 1, 1), (256, 256, 0), (64, 64, 4), (256, 256, 8), (12, 12, 2)])
def test_full_id_range_200_ids_60_compositions(hip, n_rows, nz, kernel):
    """VERDICT round 3, missing 2: ids 48..255.  A 200-id label map with 60 distinct compositions (19 ids absent) through
    the single-row and the stacked-fan projector: the host merges ids of equal composition and drops absent ones (here 60
    rows remain: more than the 48 the fast group path took before), the kernels run on compact ids.  Path lengths bit for
    bit against the DDA mirror ON THE SAME compact ids, counts <= 1e-5 against the float64 textbook Siddon on the ORIGINAL
    200-id volume and table."""
    from dex_ct_sim_amd import forward_project as fp
    ph, distinct = label_map_phantom(40, nz)
    # (an even channel count: no ray runs exactly ALONG a grid plane, where the textbook algorithm and the slab DDA may
    # each pick either of the two voxel rows - equal on a symmetric phantom, not on a label map)
    ct, _ = small_scan(n=40, nz=nz, n_views=9, n_channels=46, n_rows=n_rows)
    pj = projector(ct, ph, kernel=kernel)
    keys = {(m.density, m.matcomp) for i, m in enumerate(ph.materials) if i == 0 or (ph.volume == i).any()}
    assert pj.n_mat == len(keys) and 49 <= pj.n_mat <= 60 and len(ph.materials) == 200
    sp = spectra()
    (counts, pl), _ = pj.project(sp, want_pathlen=True)
    E, mu, w = fp.merged_tables(ct, ph, sp)
    g = oracle_geom(ct, ph)
    # float64 Siddon 1985 on the original ids and the full 200-row table
    ref = co.project_classic(g, ct.view_cs(), ct.chan_cs(), 0, ct.N_proj, ph.volume, mu, w)
    assert float(np.max(np.abs(counts.cpu().numpy() - ref) / ref)) < REL_TOL
    # the mirror on the compact ids: bit-exact path lengths
    compact = pj.id_lut[ph.volume]
    mu_c = pj.compact(mu)
    _, ref_pl = co.project_dda(g, ct.view_cs(), ct.chan_cs(), 0, ct.N_proj, compact, mu_c.astype(np.float32), w.astype(np.float32),
                               want_pathlen=True)
    assert np.array_equal(pl.cpu().numpy(), ref_pl)
    # merged ids: the summed float64 lengths of the ids behind a compact row agree with the compact row's length
    _, pl_full = co.project_classic(g, ct.view_cs(), ct.chan_cs(), 0, ct.N_proj, ph.volume, mu, w, want_pathlen=True)
    summed = np.zeros(pl_full.shape[:-1] + (pj.n_mat,))
    for i in range(200):
        if i == 0 or (ph.volume == i).any():
            summed[..., pj.id_lut[i]] += pl_full[..., i]
    assert np.max(np.abs(summed - pl.cpu().numpy())) < 2e-4
    # the public call
    if kernel == 0:
        import dex_ct_sim_amd as dx
        raw, log = dx.get_sino(ct, ph, sp[0])
        want = ref[0] if n_rows > 1 else ref[0][:, 0]
        assert raw.shape == want.shape and float(np.max(np.abs(raw - want) / want)) < REL_TOL


def test_all_256_ids_distinct_and_view_chunks(hip, monkeypatch):
    """The worst case: 256 ids, every one a composition of its own (no merging possible): 85 group passes on the stacked
    fan with the scratch budget forcing view chunks, the LDS-column kernels (64 lanes x 256 materials) on the single row
    and the cone beam; noise and the log sinogram through the chunks."""
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import forward_project as fp
    from dex_ct_sim_amd.system import AIR, Material
    ct, ph = small_scan(n=40, nz=64, n_views=9, n_channels=34, n_rows=64)
    rng = np.random.default_rng(2)
    ph.volume = np.where(ph.volume > 0, rng.integers(1, 256, ph.volume.shape, dtype=np.uint16), 0).astype(np.uint8)
    ph.materials = [AIR] + [Material(f'm{i}', 0.2 + 0.007 * i, 'H(11.2)O(88.8)') for i in range(1, 256)]
    sp = spectra()
    E, mu, w = fp.merged_tables(ct, ph, sp)
    g = oracle_geom(ct, ph)
    ref = co.project_classic(g, ct.view_cs(), ct.chan_cs(), 0, ct.N_proj, ph.volume, mu, w)
    pj = projector(ct, ph)                                       # kernel 0: byte group codes (64 rows do not fill a lane group)
    assert pj.n_mat == 256 and (pj.grouped or pj.grouped_packed)
    (c_all, p_all, l_all), air = pj.project(sp, want_pathlen=True, want_log=True)
    assert float(np.max(np.abs(c_all.cpu().numpy() - ref) / ref)) < REL_TOL
    monkeypatch.setattr(fp, '_GROUP_SCRATCH_BYTES', 256 * 64 * 34 * 4 * 2)          # two views per chunk
    (c_ch, p_ch, l_ch), _ = pj.project(sp, want_pathlen=True, want_log=True)
    assert torch.equal(c_ch, c_all) and torch.equal(p_ch, p_all) and torch.equal(l_ch, l_all)
    n_ch, _ = pj.project(sp, noise=True, seed=9)
    monkeypatch.setattr(fp, '_GROUP_SCRATCH_BYTES', 16 << 30)
    n_all, _ = pj.project(sp, noise=True, seed=9)
    assert torch.equal(n_ch, n_all) and not torch.equal(n_all, c_all)
    _, ref_pl = co.project_dda(g, ct.view_cs(), ct.chan_cs(), 0, ct.N_proj, ph.volume, mu.astype(np.float32), w.astype(np.float32),
                               want_pathlen=True)
    assert np.array_equal(p_all.cpu().numpy(), ref_pl)
    for kern in (1, 2):                                          # one thread per ray / one row per lane: LDS columns
        (c, p), _ = projector(ct, ph, kernel=kern).project(sp, want_pathlen=True)
        assert torch.equal(p, p_all) and float(((c - c_all).abs() / c_all).max()) < REL_TOL
    # exact Poisson detection from the path lengths of 256 materials
    pz, _ = pj.project(sp, noise='poisson', seed=4)
    assert torch.isfinite(pz).all() and pz.shape == c_all.shape and float(pz.mean()) > 0
    # cone beam
    cone = dx.FanBeamGeometry(N_channels=34, N_proj=9, gamma_fan=0.8230337, SID=60.0, SDD=100.0, h_iso=0.8, N_rows=12, cone=True,
                              eid=True, detector_file=ct.detector_file)
    cj = projector(cone, ph)
    (cc, cp), _ = cj.project(sp, want_pathlen=True)
    gc = oracle_geom(cone, ph)
    _, ref_cp = co.project_cone(gc, cone.view_cs(), cone.chan_cs(), 0, cone.N_proj, cone.row_z(), cone.src_z, ph.volume, mu, w,
                                dda=True, n_threads=8)
    ref_c, _ = co.project_cone(gc, cone.view_cs(), cone.chan_cs(), 0, cone.N_proj, cone.row_z(), cone.src_z, ph.volume, mu, w,
                               dda=False, n_threads=8)
    assert np.array_equal(cp.cpu().numpy(), ref_cp)                # (id 255 is an id like any other: "outside" is 256)
    assert float(np.max(np.abs(cc.cpu().numpy() - ref_c) / ref_c)) < REL_TOL
    # an id beyond the table is refused
    ph.materials = ph.materials[:200]
    with pytest.raises(ValueError):
        projector(ct, ph)


# ---- round 3: the second output of get_sino from the device, whole-line stores, the O(1) cache key
def _np_log(air, counts):
    """What the host did before round 3 (and what the reference's caller sees): float32 ln(air / raw)."""
    with np.errstate(divide='ignore'):
        return np.log(np.float32(air)[:, None] / counts.reshape(counts.shape[0], -1)).reshape(counts.shape)


@pytest.mark.parametrize('kernel,n_rows,nz,n_mat', [(1, 1, 1, 3), (6, 1, 1, 3), (2, 66, 70, 3), (3, 64, 64, 3), (5, 64, 64, 4),
                                                    (7, 256, 256, 3), (7, 512, 512, 2), (7, 200, 208, 4), (7, 50, 64, 3),
                                                    (4, 64, 64, 7), (8, 256, 256, 7), (1, 8, 8, 6)])
def test_log_sinogram_from_the_detection_store(hip, kernel, n_rows, nz, n_mat):
    """sino_log = ln(air / counts) (main.py:120-122) is written by the projection kernels' own detection store
    (dexct_log_out): same counts bit for bit as without it, the log within float32 rounding of the NumPy expression
    the host used to evaluate, in the kernel's native layout and through the transpose."""
    from dex_ct_sim_amd.system import AIR, WATER
    ct, ph = small_scan(n=40, nz=nz, n_views=6, n_channels=37, n_rows=n_rows)
    if n_mat == 2:
        ph.volume = np.minimum(ph.volume, 1).astype(np.uint8)
        ph.materials = [AIR, WATER]
    elif n_mat > 3:
        ph = ph_many(ph, n_mat)
    pj = projector(ct, ph, kernel=kernel)
    _, mu_d, w_d, air = pj.upload_tables(spectra())
    for layout in (None, 0, 1):
        plain = pj.project_tables(mu_d, w_d, layout=layout)
        counts, log = pj.project_tables(mu_d, w_d, layout=layout, air=air)
        assert torch.equal(counts, plain)
        ref = _np_log(air, counts.cpu().numpy())
        assert np.allclose(log.cpu().numpy(), ref, rtol=5e-6, atol=5e-7), np.abs(log.cpu().numpy() - ref).max()
    alone = pj.sino_log(counts, air)                       # the pass of its own: same arithmetic
    assert torch.equal(alone, log)


def test_log_sinogram_of_noisy_and_cone_beam_scans(hip):
    """With quantum noise the log belongs to the noisy counts (dexct_sino_log after the sampling); cone-beam kernels
    write it from their detection like the fan kernels."""
    ct, ph = small_scan(n=32, nz=64, n_views=10, n_channels=40, n_rows=64)
    sp = spectra()
    for s in sp:
        s.rescale_counts(1e2)
    pj = projector(ct, ph)
    for noise in (True, 'poisson'):
        (counts, log), air = pj.project(sp, noise=noise, seed=3, want_log=True)
        clean, _ = pj.project(sp)
        assert not torch.equal(counts, clean)
        assert np.allclose(log.cpu().numpy(), _np_log(air, counts.cpu().numpy()), rtol=5e-6, atol=5e-7)
    cone = dx_cone(ct)
    for k in (1, 2):
        pjc = projector(cone, ph, kernel=k)
        (counts, log), air = pjc.project(sp, want_log=True)
        plain, _ = pjc.project(sp)
        assert torch.equal(counts, plain)
        assert np.allclose(log.cpu().numpy(), _np_log(air, counts.cpu().numpy()), rtol=5e-6, atol=5e-7)
        # round 4: quantum noise on cone-beam scans (Philox): reproducible, the same sample from both cone kernels and from
        # view shards, moments as predicted.  Round 6: ONE launch - the variance comes out of the detection's own energy
        # loop (the same bits as round 5's second launch with the variance weights as weights) and the kernel draws the
        # sample dexct_add_noise draws from the same signal and variance
        (noisy, nlog), _ = pjc.project(sp, noise=True, seed=11, want_log=True)
        from dex_ct_sim_amd import _native, forward_project as fp
        from dex_ct_sim_amd._device import ptr, stream_ptr
        _, mu, w, w2 = fp.merged_tables(cone, ph, sp, with_variance=True)
        mu_d, w_d, w2_d = (torch.tensor(x, dtype=torch.float32, device='cuda').contiguous() for x in (pjc.compact(mu), w, w2))
        second = pjc.project_tables(mu_d, w2_d)
        n_again, var = pjc.project_tables(mu_d, w_d, w2_d=w2_d, seed=11, want_variance=True)
        assert torch.equal(var, second) and torch.equal(n_again, noisy)
        sampled = plain.clone()
        _native.check(pjc.lib.dexct_add_noise(ptr(sampled), ptr(var), 2, cone.N_proj, cone.N_rows, cone.N_channels, 0, 0, 11,
                                              stream_ptr()), 'dexct_add_noise')
        assert torch.equal(sampled, noisy)
        again, _ = pjc.project(sp, noise=True, seed=11)
        other, _ = pjc.project(sp, noise=True, seed=12)
        assert torch.equal(noisy, again) and not torch.equal(noisy, other) and not torch.equal(noisy, plain)
        assert np.allclose(nlog.cpu().numpy(), _np_log(air, noisy.cpu().numpy()), rtol=5e-6, atol=5e-7)
        if k == 1:
            first = noisy
        else:
            assert torch.equal(noisy, first)
        lo_shard, _ = projector(cone, ph, kernel=k, view_range=(0, 4)).project(sp, noise=True, seed=11)
        hi_shard, _ = projector(cone, ph, kernel=k, view_range=(4, 10)).project(sp, noise=True, seed=11)
        assert torch.equal(torch.cat([lo_shard, hi_shard], dim=1), noisy)
        from dex_ct_sim_amd import forward_project as fp
        _, _, w, w2 = fp.merged_tables(cone, ph, sp, with_variance=True)
        z = ((noisy - plain).double() / plain.double().clamp(min=1e-30).sqrt())          # ~ N(0, var / counts)
        ratio = float((w2.sum() / w.sum()))                                             # unattenuated variance / signal
        assert abs(float(z.mean())) < 0.05 * np.sqrt(ratio) and 0.3 * ratio < float(z.var()) < 3.0 * ratio
        pz, _ = pjc.project(sp, noise='poisson', seed=5)
        assert torch.isfinite(pz).all() and not torch.equal(pz, plain)


@pytest.mark.parametrize('n_rows,n_channels,n_mat,n_spec', [(256, 53, 3, 2), (512, 7, 3, 2), (1024, 5, 3, 1), (200, 10, 3, 2),
                                                            (320, 9, 2, 1), (2048, 3, 4, 2), (256, 6, 2, 2)])
def test_whole_line_stores_change_no_bit(hip, n_rows, n_channels, n_mat, n_spec, monkeypatch):
    """rows16_kernel hands its results over through LDS so that one store instruction writes consecutive addresses
    (STAGED; DEXCT_P16_STAGED=0 restores the per-round 16-byte stores): same counts, same log sinogram, same path
    lengths; 4 / 2 / 1 pairs per wave, ragged channel groups, idle lanes (200 and 320 rows), two z-chunks (2048 rows),
    and the two-material / two-spectrum case that has no room for staged results and takes the old path by itself."""
    from dex_ct_sim_amd.system import AIR, WATER
    ct, ph = small_scan(n=40, nz=-(-n_rows // 16) * 16, n_views=5, n_channels=n_channels, n_rows=n_rows)
    if n_mat == 2:
        ph.volume = np.minimum(ph.volume, 1).astype(np.uint8)
        ph.materials = [AIR, WATER]
    elif n_mat == 4:
        ph = ph_many(ph, 4)
    pj = projector(ct, ph, kernel=7)
    _, mu_d, w_d, air = pj.upload_tables(spectra()[:n_spec])
    res = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('DEXCT_P16_STAGED', mode)
        res[mode] = pj.project_tables(mu_d, w_d, layout=None, air=air, want_pathlen=True)
    for a, b in zip(res['1'], res['0']):
        assert torch.equal(a, b)
    assert torch.isfinite(res['1'][0]).all() and (res['1'][0] > 0).all()


def test_get_sino_returns_page_locked_arrays_and_verifies_the_cached_volume(hip, monkeypatch):
    """Public boundary (NumPy in / NumPy out): results arrive through pinned memory that belongs to the
    returned arrays; the device-resident state is found by an O(1) key (version counter + a strided sample) and VERIFIED
    by a whole-volume checksum computed while the GPU works; DEXCT_VERIFY_VOLUME=0 opts out."""
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import forward_project as fp
    ct, ph = small_scan(n=48, nz=4, n_views=20, n_channels=48, n_rows=4)
    sp = spectra()[0]
    raw, log = dx.get_sino(ct, ph, sp)
    assert raw.dtype == np.float32 and log.dtype == np.float32 and raw.shape == (20, 4, 48)
    assert torch.from_numpy(raw).is_pinned()
    raw2, _ = dx.get_sino(ct, ph, sp)
    assert np.array_equal(raw, raw2) and not np.shares_memory(raw, raw2)       # each result owns its buffer
    pj, check = fp._projector(ct, ph, (0, 20))
    assert check is True and fp._hash64(ph.volume) == pj.volume_hash           # found in the cache: the caller verifies it
    assert fp._projector(ct, ph, (0, 20))[0] is pj                               # reused
    ph.volume[:, 10:30, 10:30] = 2                                               # in-place edit, announced
    ph.touch()
    raw3, _ = dx.get_sino(ct, ph, sp)
    assert fp._projector(ct, ph, (0, 20))[0] is not pj and not np.array_equal(raw3, raw)
    ph.volume = np.zeros_like(ph.volume)                                         # assignment bumps the version by itself
    raw4, log4 = dx.get_sino(ct, ph, sp)
    assert np.allclose(log4, np.log(np.float32(fp.effective_weights(ct, sp).sum()) / raw4), atol=1e-6)
    # opt-out: the key alone, never more than the sample is hashed
    monkeypatch.setenv('DEXCT_VERIFY_VOLUME', '0')
    calls = []
    real = fp._hash64
    monkeypatch.setattr(fp, '_hash64', lambda a: (calls.append(np.asarray(a).size), real(a))[1])
    assert fp._projector(ct, ph, (0, 20))[1] is False
    assert max(calls) <= 4200


@pytest.mark.parametrize('n,nz,rows', [(64, 64, 64), (128, 1, 1)])
def test_get_sino_sees_unannounced_in_place_edits_of_power_of_two_volumes(hip, n, nz, rows):
    """Advisor finding of round 3: with a sample stride that is a multiple of Nx (every power-of-two volume) all sampled
    voxels lie in the x = 0 face and an in-place edit of the interior WITHOUT touch() returned the stale sinogram.  Now the
    whole volume is checksummed (beside the GPU work) and the stride of the sample is coprime to the dimensions."""
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import forward_project as fp
    ct, ph = small_scan(n=n, nz=nz, n_views=16, n_channels=64, n_rows=rows)
    sp = spectra()[0]
    a, _ = dx.get_sino(ct, ph, sp)
    b, _ = dx.get_sino(ct, ph, sp)
    assert np.array_equal(a, b)
    ph.volume[nz // 2, n // 2 - 4:n // 2 + 4, n // 2 - 4:n // 2 + 4] = 2        # interior block, no touch()
    c, _ = dx.get_sino(ct, ph, sp)
    assert not np.array_equal(a, c)
    ct2, ph2 = small_scan(n=n, nz=nz, n_views=16, n_channels=64, n_rows=rows)
    ph2.volume[nz // 2, n // 2 - 4:n // 2 + 4, n // 2 - 4:n // 2 + 4] = 2
    d, _ = dx.get_sino(ct2, ph2, sp)
    assert np.array_equal(c, d)
    # the strided sample itself now reaches the interior: for these shapes the samples cover many x, y and z
    step = fp._sample_step(ph.volume.shape)
    idx = np.arange(0, ph.volume.size, step)
    z, y, x = np.unravel_index(idx, ph.volume.shape)
    assert len(np.unique(x)) > 16 and len(np.unique(y)) > 16 and (nz == 1 or len(np.unique(z)) > 16)


def test_integration_md_get_sino_stub_runs(hip):
    """The reference-side get_sino binding shown in INTEGRATION.md, executed as written: same counts as the package's
    get_sino bit for bit (same kernel), log sinogram within float32 rounding."""
    import os
    import re
    import dex_ct_sim_amd as dx
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, 'INTEGRATION.md')).read()
    blocks = [b for b in re.findall(r'```python\n(.*?)```', text, flags=re.S) if 'def get_sino' in b]
    assert len(blocks) == 1
    code = blocks[0].replace("C.CDLL('dex-ct-sim_amd/libdexct_hip.so')",
                             f"C.CDLL({os.path.join(root, 'dex-ct-sim_amd', 'libdexct_hip.so')!r})")
    ns = {}
    exec(compile(code, 'INTEGRATION.md', 'exec'), ns)
    ct, ph = small_scan(n=48, n_views=40, n_channels=64)
    sp = spectra()[0]
    raw, log = ns['get_sino'](ct, ph, sp)
    raw0, log0 = dx.get_sino(ct, ph, sp)
    assert np.array_equal(raw, raw0) and np.allclose(log, log0, rtol=0, atol=1e-6)


@pytest.mark.parametrize('kind', ['single_row', 'stacked_256', 'rows_64', 'cone'])
def test_reduced_quadrature_on_the_device(hip, kind, monkeypatch):
    """quadrature='reduced' (opt-in; dex-ct-sim_amd/quadrature.py): the same kernels on a shorter energy table.  Against the
    full grid on every ray <= 2e-6 relative (the table's verified bound is <= 1e-6; both launches round in float32),
    against the float64 detection of the device's own path lengths inside the parity bar of the full grid (1e-5); the path
    bounds the guarantee rests on really bound every ray; the default stays the full grid."""
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import forward_project as fp
    if kind == 'cone':
        ct, ph = small_scan(n=40, nz=24, n_views=20, n_channels=48, n_rows=10)
        ct = dx.FanBeamGeometry(N_channels=48, N_proj=20, gamma_fan=0.8230337, SID=60.0, SDD=100.0, h_iso=0.8, eid=True,
                                detector_file=ct.detector_file, N_rows=10, cone=True, src_z=0.3)
    else:
        n_rows = {'single_row': 1, 'stacked_256': 256, 'rows_64': 64}[kind]
        ct, ph = small_scan(n=64, nz=n_rows, n_views=24, n_channels=66, n_rows=n_rows)
    sp = spectra()
    pj = projector(ct, ph)
    (full, pl), air = pj.project(sp, want_pathlen=True, layout=0)
    assert pj.quadrature_info is None                                            # default: the full grid
    (red, pl2), _ = pj.project(sp, want_pathlen=True, layout=0, quadrature='reduced')
    info = pj.quadrature_info
    assert info is not None and 3 * info['nodes'] < info['n_full'] and info['max_rel_err'] <= 1.0e-6
    assert torch.equal(pl, pl2)                                                  # the traversal does not change
    # the domain of the guarantee: no ray leaves it
    l_max, c_max = pj.path_bounds()
    assert float(pl.sum(-1).max()) <= c_max * (1 + 1e-6)
    for k in range(pj.n_mat):
        assert float(pl[..., k].max()) <= l_max[k] * (1 + 1e-6) + 1e-6
    # NumPy bounding boxes of the same volume
    vol = ph.volume if ct.cone or ct.N_rows == ph.Nz else ph.volume[ph.z_index:ph.z_index + ct.N_rows]
    for k in range(1, pj.n_mat):
        zz, yy, xx = np.nonzero(vol == pj.mat_rows[k])
        ext = [(xx.max() - xx.min() + 1) * ph.dx, (yy.max() - yy.min() + 1) * ph.dy]
        if ct.cone:
            ext.append((zz.max() - zz.min() + 1) * ph.dz)
        assert abs(l_max[k] - min(c_max, float(np.sqrt(np.sum(np.square(ext)))))) < 1e-9
    rel = ((red.double() - full.double()).abs() / full.double()).max().item()
    assert rel <= 2.0e-6, rel
    # float64 detection of the device's path lengths on the FULL grid
    _, mu64, w64 = fp.merged_tables(ct, ph, sp)
    L = pl.double().cpu().numpy().reshape(-1, pj.n_mat)
    ref = (np.exp(-(L @ mu64[pj.mat_rows])) @ w64.T).T.reshape(red.shape)
    assert (np.abs(red.double().cpu().numpy() - ref) / ref).max() < REL_TOL
    # the public call, and the environment default
    raw_f, log_f = dx.get_sino(ct, ph, sp[0])
    raw_r, log_r = dx.get_sino(ct, ph, sp[0], quadrature='reduced')
    assert not np.array_equal(raw_f, raw_r)
    assert (np.abs(raw_r.astype(np.float64) - raw_f) / raw_f).max() <= 2.0e-6
    assert np.abs(log_r.astype(np.float64) - log_f).max() <= 4.0e-6
    monkeypatch.setenv('DEXCT_QUADRATURE', 'reduced')
    assert np.array_equal(dx.get_sino(ct, ph, sp[0])[0], raw_r)
    assert np.array_equal(dx.get_sino(ct, ph, sp[0], quadrature='full')[0], raw_f)
    # noisy sinograms keep the full grid (the variance and the photon counts are defined on its bins)
    noisy_env = dx.get_sino(ct, ph, sp[0], noise=True, seed=3)[0]
    monkeypatch.delenv('DEXCT_QUADRATURE')
    assert np.array_equal(dx.get_sino(ct, ph, sp[0], noise=True, seed=3)[0], noisy_env)
    with pytest.raises(ValueError):
        dx.get_sino(ct, ph, sp[0], quadrature='fast')


def test_reduced_quadrature_is_skipped_for_long_tables(hip):
    """More than 4 table rows: the domain cannot be sampled densely enough for a verified bound - the full grid runs."""
    ct, ph = small_scan(n=48, nz=1, n_views=12, n_channels=40)
    ph = ph_many(ph, 7)
    pj = projector(ct, ph)
    a, _ = pj.project(spectra())
    b, _ = pj.project(spectra(), quadrature='reduced')
    assert pj.quadrature_info is None and torch.equal(a, b)


@pytest.mark.parametrize('kind', ['single_row', 'stacked', 'packed', 'grouped', 'cone', 'cone_rows', 'noisy'])
def test_projection_of_a_view_slice_is_the_slice_of_the_projection(hip, kind):
    """Projector.project_tables(views=(a, b)) - what a step uses that hands its sinogram on in chunks (bench.py, sharded
    runs): both outputs of every kernel family, projected slice by slice, are bit for bit the slices of the whole projection
    (plans, Philox keys and view angles are indexed by the global view)."""
    import os
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import forward_project as fp, synthetic, system
    n_views = 11
    if kind in ('cone', 'cone_rows'):
        _, ph = small_scan(n=48, nz=48 if kind == 'cone' else 24, n_views=n_views, n_channels=40)
        ct = dx.FanBeamGeometry(N_channels=40, N_proj=n_views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, h_iso=0.5, eid=True,
                                detector_file=os.path.join(INPUT, 'detector', 'eta_eid_mv.bin'), N_rows=6 if kind == 'cone' else 24, cone=True, src_z=0.2)
    else:
        nz = {'single_row': 1, 'stacked': 64, 'packed': 192, 'grouped': 64, 'noisy': 16}[kind]
        ct, ph = small_scan(n=48, nz=nz, n_views=n_views, n_channels=40, n_rows=nz)
        if kind == 'grouped':                                   # more than four table rows: material groups + detection pass
            ph = ph_many(ph, 8)
    specs = [synthetic.kramers_spectrum(120), synthetic.kramers_spectrum(80)]
    pj = fp.Projector(ct, ph, view_range=(2, n_views))          # a shard that does not start at view 0
    if kind == 'noisy':
        _, mu, w, w2 = fp.merged_tables(ct, ph, specs, with_variance=True)
        mu_d, w_d = (torch.tensor(x, dtype=torch.float32, device='cuda') for x in (pj.compact(mu), w))
        kw = dict(w2_d=torch.tensor(w2, dtype=torch.float32, device='cuda'), seed=5)
        air = [float(x) for x in w.sum(1)]
    else:
        _, mu_d, w_d, air = pj.upload_tables(specs)
        kw = {}
    whole_c, whole_l = pj.project_tables(mu_d, w_d, layout=None, air=air, **kw)
    assert whole_c.shape[1] == n_views - 2 and torch.isfinite(whole_l).all()
    for a, b in ((0, 4), (4, 5), (5, 9)):
        c, l = pj.project_tables(mu_d, w_d, layout=None, air=air, views=(a, b), **kw)
        assert c.shape[1] == b - a
        assert torch.equal(c, whole_c[:, a:b]) and torch.equal(l, whole_l[:, a:b]), (kind, a, b)
    c = pj.project_tables(mu_d, w_d, layout=0, views=(3, 7), **kw)              # the reference's order (a transpose pass for stacked fans)
    assert torch.equal(c, pj.project_tables(mu_d, w_d, layout=0, **kw)[:, 3:7])
    with pytest.raises(ValueError):
        pj.project_tables(mu_d, w_d, views=(4, 4))
    with pytest.raises(ValueError):
        pj.project_tables(mu_d, w_d, views=(0, n_views))
