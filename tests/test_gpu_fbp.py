"""HIP fan-beam FBP against the oracle, and the projection -> reconstruction loop."""
import numpy as np
import pytest

from conftest import small_scan
from oracle import fbp_oracle as fo
from test_fbp_oracle import disc_sino

pytestmark = pytest.mark.gpu


def test_get_recon_matches_oracle(hip):
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import back_project as bp, synthetic
    ct = dx.FanBeamGeometry(N_channels=257, N_proj=360, gamma_fan=0.8230337, SID=60.0, SDD=100.0)
    s = disc_sino(ct.thetas, ct.gammas, [(0, 0, 10.0, 0.2), (5.0, -3.0, 2.0, 0.3)])
    spec = synthetic.kramers_spectrum(120)
    raw, hu = dx.get_recon(s, ct, spec, 128, 40.0, 0.8)
    mu_w = bp.water_mu(ct, spec)
    ref, ref_hu = fo.get_recon(s.astype(np.float32), ct.thetas, ct.gammas, 60.0, 128, 40.0, 0.8, mu_water=mu_w)
    assert raw.shape == (128, 128) and raw.dtype == np.float32 and hu.dtype == np.float32
    scale = np.abs(ref).max()
    assert np.max(np.abs(raw - ref)) < 2e-5 * scale          # float32 filter + accumulation vs float64
    assert np.max(np.abs(hu - ref_hu)) < 0.2


def test_multi_row_and_filter_linearity(hip):
    import dex_ct_sim_amd as dx
    import torch
    from dex_ct_sim_amd import back_project as bp
    ct = dx.FanBeamGeometry(N_channels=129, N_proj=90, gamma_fan=0.8230337, SID=60.0, SDD=100.0, N_rows=3)
    a = disc_sino(ct.thetas, ct.gammas, [(2.0, 1.0, 6.0, 0.2)])
    b = disc_sino(ct.thetas, ct.gammas, [(-3.0, 0.0, 3.0, 0.4)])
    stack = np.stack([a, b, a + 2 * b], axis=1).astype(np.float32)          # [views, rows, channels]
    img = bp.recon_device(torch.tensor(stack, device='cuda'), ct, 64, 30.0, 1.0).cpu().numpy()
    assert img.shape == (3, 64, 64)
    assert np.max(np.abs(img[2] - (img[0] + 2 * img[1]))) < 1e-5 * np.abs(img).max()
    ref, _ = fo.get_recon(a.astype(np.float32), ct.thetas, ct.gammas, 60.0, 64, 30.0, 1.0)
    assert np.max(np.abs(img[0] - ref)) < 2e-5 * np.abs(ref).max()


def test_project_then_reconstruct_phantom(hip):
    """Mono-energetic projection of the synthetic phantom on the GPU, reconstructed on the GPU: the image
    reproduces the phantom's attenuation map (closes the get_sino -> get_recon loop of main.py:120-134)."""
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import synthetic
    ct, ph = small_scan(n=128, n_views=400, n_channels=300)
    spec = dx.xRaySpectrum.from_arrays('mono60', [60.0], [1.0e6])
    raw, log = dx.get_sino(ct, ph, spec)
    img, _ = dx.get_recon(log, ct, spec, 128, 51.2, 1.0)
    truth = ph.M_mono(60.0)
    inner = np.zeros_like(truth, dtype=bool)
    c = (np.arange(128) - 64 + 0.5) * 0.4
    x, y = np.meshgrid(c, c)
    water = (ph.volume[0] == 1) & (x ** 2 + y ** 2 < 15.0 ** 2)
    # exclude a margin around bone inserts (blur): compare the water plateau and a bone centre
    from scipy import ndimage
    near_bone = ndimage.binary_dilation(ph.volume[0] == 2, iterations=4)
    plateau = water & ~near_bone
    assert abs(img[plateau].mean() - truth[plateau].mean()) < 0.01 * truth[plateau].mean()
    bone_core = ndimage.binary_erosion(ph.volume[0] == 2, iterations=3)
    if bone_core.sum() > 10:
        assert abs(img[bone_core].mean() - truth[bone_core].mean()) < 0.05 * truth[bone_core].mean()


def test_make_vmi_matches_reference_formula(hip):
    """plots.py:136-144 restated in NumPy by the test: u1*M1 + u2*M2, HU against water."""
    from dex_ct_sim_amd import back_project as bp, matdecomp as md, xcompy
    rng = np.random.default_rng(2)
    M1, M2 = rng.uniform(0, 1.2, (64, 64)), rng.uniform(0, 0.5, (64, 64))
    for E0 in (40.0, 70.0, 120.0):
        E = np.array([E0])
        u1, u2 = xcompy.mixatten(md.matcomp1, E), xcompy.mixatten(md.matcomp2, E)
        uw = 1.0 * xcompy.mixatten('H(11.2)O(88.8)', E)
        vmi = u1 * M1 + u2 * M2
        ref_hu = (1000 * (vmi - uw) / uw).astype(np.float32)
        got = bp.make_vmi(E0, M1, M2)
        assert got.dtype == np.float32 and got.shape == (64, 64)
        assert np.max(np.abs(got - ref_hu)) < 2e-4 * np.abs(ref_hu).max()
        assert np.allclose(bp.make_vmi(E0, M1, M2, HU=False), vmi.astype(np.float32), rtol=2e-6)


def test_volume_backprojection_shares_geometry_bit_exactly(hip):
    """11 rows take the 8-rows-per-thread kernel (incl. a partial last group): every slice equals the single-slice
    reconstruction of its own sinogram bit for bit."""
    import dex_ct_sim_amd as dx
    import torch
    from dex_ct_sim_amd import back_project as bp
    rows = 11
    ct1 = dx.FanBeamGeometry(N_channels=129, N_proj=90, gamma_fan=0.8230337, SID=60.0, SDD=100.0, N_rows=1)
    ctn = dx.FanBeamGeometry(N_channels=129, N_proj=90, gamma_fan=0.8230337, SID=60.0, SDD=100.0, N_rows=rows)
    rng = np.random.default_rng(4)
    stack = rng.uniform(0.0, 3.0, (90, rows, 129)).astype(np.float32)
    vol = bp.recon_device(torch.tensor(stack, device='cuda'), ctn, 70, 30.0, 0.9).cpu().numpy()
    assert vol.shape == (rows, 70, 70)
    for r in range(rows):
        one = bp.recon_device(torch.tensor(np.ascontiguousarray(stack[:, r]), device='cuda'), ct1, 70, 30.0, 0.9)
        assert np.array_equal(vol[r], one.cpu().numpy().reshape(70, 70)), r


@pytest.mark.parametrize('seed', range(8))
def test_random_reconstructions_match_oracle(hip, seed):
    """Randomised fan geometries, matrix sizes, fields of view, ramp cutoffs and row counts (1 takes the single-slice
    kernel, >= 8 the shared-geometry one): the HIP filter + back-projection follow the float64 oracle."""
    import dex_ct_sim_amd as dx
    import torch
    from dex_ct_sim_amd import back_project as bp
    rng = np.random.default_rng(3000 + seed)
    n_ch, n_views = int(rng.integers(16, 200)), int(rng.integers(8, 120))
    rows = int(rng.choice([1, 1, 2, 8, 9, 17]))
    sid = float(rng.uniform(30.0, 80.0))
    sdd = float(sid * rng.uniform(1.2, 2.0))
    fan = float(rng.uniform(0.3, 1.2))
    n_mat = int(rng.integers(8, 97))
    fov = float(rng.uniform(0.3, 1.0) * 2 * sid * np.sin(0.5 * fan))       # inside the fan's field of view, or beyond
    ramp = float(rng.uniform(0.2, 1.0))
    ct = dx.FanBeamGeometry(N_channels=n_ch, N_proj=n_views, gamma_fan=fan, SID=sid, SDD=sdd, N_rows=rows)
    stack = rng.uniform(0.0, 4.0, (n_views, rows, n_ch)).astype(np.float32)
    sino = stack if rows > 1 else stack[:, 0]
    img = bp.recon_device(torch.tensor(sino, device='cuda'), ct, n_mat, fov, ramp).cpu().numpy().reshape(rows, n_mat, n_mat)
    for r in sorted({0, rows // 2, rows - 1}):
        ref, _ = fo.get_recon(np.ascontiguousarray(stack[:, r]), ct.thetas, ct.gammas, sid, n_mat, fov, ramp)
        scale = np.abs(ref).max()
        assert np.max(np.abs(img[r] - ref)) < 5e-5 * scale, (seed, r, np.max(np.abs(img[r] - ref)) / scale)


@pytest.mark.parametrize('window', ['sinc', 'hann'])
def test_windowed_reconstruction_matches_oracle(hip, window):
    """get_recon(..., window=) against the oracle's independently computed taps; the window smooths: less noise
    than the plain ramp on a noisy sinogram, same plateau."""
    import dex_ct_sim_amd as dx
    ct = dx.FanBeamGeometry(N_channels=193, N_proj=240, gamma_fan=0.8230337, SID=60.0, SDD=100.0)
    rng = np.random.default_rng(9)
    s = disc_sino(ct.thetas, ct.gammas, [(0, 0, 12.0, 0.2)]) + 0.02 * rng.standard_normal((240, 193))
    spec = dx.xRaySpectrum.from_arrays('mono60', [60.0], [1.0e6])
    raw, _ = dx.get_recon(s, ct, spec, 96, 40.0, 0.9, window=window)
    ref, _ = fo.get_recon(s.astype(np.float32), ct.thetas, ct.gammas, 60.0, 96, 40.0, 0.9, window=window)
    assert np.max(np.abs(raw - ref)) < 3e-5 * np.abs(ref).max()
    plain, _ = dx.get_recon(s, ct, spec, 96, 40.0, 0.9)
    c = (np.arange(96) - 48 + 0.5) * (40.0 / 96)
    x, y = np.meshgrid(c, c)
    core = x ** 2 + y ** 2 < 8.0 ** 2
    assert raw[core].std() < {'sinc': 0.9, 'hann': 0.7}[window] * plain[core].std()
    assert abs(raw[core].mean() - 0.2) < 0.004 and abs(plain[core].mean() - 0.2) < 0.004


@pytest.mark.parametrize('seed', range(4))
def test_fdk_matches_oracle_on_random_data(hip, seed):
    """dexct_fdk_backproject (+ the fan filter) against the float64 FDK oracle: random cone geometries, source off
    the mid-plane, slice grids that partly leave the detector's coverage."""
    import dex_ct_sim_amd as dx
    import torch
    from dex_ct_sim_amd import back_project as bp
    rng = np.random.default_rng(4000 + seed)
    n_ch, n_views, rows = int(rng.integers(16, 90)), int(rng.integers(8, 60)), int(rng.integers(2, 20))
    sid = float(rng.uniform(30.0, 80.0))
    sdd = float(sid * rng.uniform(1.2, 2.0))
    ct = dx.FanBeamGeometry(N_channels=n_ch, N_proj=n_views, gamma_fan=float(rng.uniform(0.3, 1.0)), SID=sid, SDD=sdd,
                            h_iso=float(rng.uniform(0.1, 0.6)), N_rows=rows, cone=True, src_z=float(rng.uniform(-1, 1)))
    n_mat, fov = int(rng.integers(8, 70)), float(rng.uniform(5.0, 30.0))
    n_sl = int(rng.integers(1, 11))
    z0, dz = float(rng.uniform(-3.0, 0.0)), float(rng.uniform(0.1, 0.8))
    sino = rng.uniform(0.0, 4.0, (n_views, rows, n_ch)).astype(np.float32)
    img = bp.recon_device(torch.tensor(sino, device='cuda'), ct, n_mat, fov, 0.8, slices=(n_sl, z0, dz)).cpu().numpy()
    ref = fo.fdk_recon(sino, ct.thetas, ct.gammas, sid, sdd, ct.row_z(), ct.src_z, n_mat, fov, 0.8, z0 + dz * np.arange(n_sl))
    assert img.shape == ref.shape == (n_sl, n_mat, n_mat)
    assert np.max(np.abs(img - ref)) < 5e-5 * max(np.abs(ref).max(), 1e-30), seed


def test_cone_project_then_fdk_reproduces_the_phantom(hip):
    """Cone-beam projection of the synthetic phantom (HIP, mono-energetic) reconstructed with FDK (HIP): the water
    plateau of the mid-plane and of an off-centre slice reproduce the phantom's attenuation."""
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import synthetic
    from scipy import ndimage
    n, nz = 96, 24
    ph = synthetic.make_phantom(n, nz, extent=25.6, n_spheres=0)            # water cylinder; dz = 25.6 / 96
    dzv = ph.dz
    ct = dx.FanBeamGeometry(N_channels=200, N_proj=240, gamma_fan=0.55, SID=60.0, SDD=100.0, h_iso=dzv, N_rows=40,
                            cone=True, src_z=0.0)
    spec = dx.xRaySpectrum.from_arrays('mono60', [60.0], [1.0e6])
    raw, log = dx.get_sino(ct, ph, spec)
    assert log.shape == (240, 40, 200)
    zs = (np.arange(nz) + 0.5 - nz / 2) * dzv
    vol, _ = dx.get_recon(log, ct, spec, n, 25.6, 1.0, slices=(nz, float(zs[0]), dzv))
    truth = ph.M_mono(60.0, z=nz // 2)
    water = ndimage.binary_erosion(ph.volume[nz // 2] == 1, iterations=6)
    mu_w = truth[water].mean()
    for k in (nz // 2, nz // 2 + 6, 3):                                     # mid-plane, off-centre, near the edge
        assert abs(vol[k][water].mean() - mu_w) < 0.015 * mu_w, (k, vol[k][water].mean(), mu_w)
    air = ~ndimage.binary_dilation(ph.volume[nz // 2] == 1, iterations=6)
    assert abs(vol[nz // 2][air].mean()) < 0.02 * mu_w


def test_non_finite_sinogram_values_do_not_poison_the_image(hip):
    import dex_ct_sim_amd as dx
    ct = dx.FanBeamGeometry(N_channels=129, N_proj=90, gamma_fan=0.8230337, SID=60.0, SDD=100.0)
    s = disc_sino(ct.thetas, ct.gammas, [(0, 0, 8.0, 0.2)])
    spec = dx.xRaySpectrum.from_arrays('mono60', [60.0], [1.0e6])
    clean, _ = dx.get_recon(s, ct, spec, 64, 30.0, 1.0)
    s2 = s.copy()
    s2[17, 40], s2[60, 3] = np.nan, np.inf
    with pytest.warns(RuntimeWarning, match='2 non-finite'):
        dirty, _ = dx.get_recon(s2, ct, spec, 64, 30.0, 1.0)
    assert np.all(np.isfinite(dirty))
    assert np.abs(dirty - clean).max() < 0.2 * clean.max()        # two lost samples of 11 610: a local streak


@pytest.mark.parametrize('theta_tot', [np.pi + 0.8230337, 4.5, 6.0])
def test_short_scan_parker_matches_oracle(hip, theta_tot):
    """rotation_angle_total < 2 pi (input/params.txt:24): Parker-weighted short-scan FBP (dexct_fbp_parker) against the
    float64 oracle, through the public get_recon; the image agrees with the full-rotation image of the same object."""
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import synthetic
    ct = dx.FanBeamGeometry(N_channels=257, N_proj=400, gamma_fan=0.8230337, SID=60.0, SDD=100.0, theta_tot=theta_tot)
    assert np.isclose(ct.thetas[-1], theta_tot * 399 / 400)
    discs = [(0, 0, 10.0, 0.2), (5.0, -3.0, 2.0, 0.3), (-6.0, 4.0, 1.5, -0.1)]
    s = disc_sino(ct.thetas, ct.gammas, discs).astype(np.float32)
    spec = synthetic.kramers_spectrum(120)
    raw, _ = dx.get_recon(s, ct, spec, 128, 40.0, 0.9)
    ref, _ = fo.get_recon(s, ct.thetas, ct.gammas, 60.0, 128, 40.0, 0.9, theta_tot=theta_tot)
    assert np.max(np.abs(raw - ref)) < 3e-5 * np.abs(ref).max()
    full_ct = dx.FanBeamGeometry(N_channels=257, N_proj=400, gamma_fan=0.8230337, SID=60.0, SDD=100.0)
    full, _ = dx.get_recon(disc_sino(full_ct.thetas, full_ct.gammas, discs), full_ct, spec, 128, 40.0, 0.9)
    c = (np.arange(128) - 64 + 0.5) * (40 / 128)
    x, y = np.meshgrid(c, c)
    assert np.abs(raw - full)[x ** 2 + y ** 2 < 81.0].mean() < 2e-3


def test_short_scan_limits_rows_and_cone(hip):
    """Less than pi + fan angle is refused (projections are missing), more than one rotation too; a stacked sinogram is
    weighted row by row; a cone-beam short scan goes through Feldkamp with the same weights and stays close to the full
    rotation's image in the central slices."""
    import dex_ct_sim_amd as dx
    import torch
    from dex_ct_sim_amd import _native, back_project as bp
    fan = 0.8230337
    short = dx.FanBeamGeometry(N_channels=129, N_proj=200, gamma_fan=fan, SID=60.0, SDD=100.0, theta_tot=np.pi + 0.5 * fan)
    s = disc_sino(short.thetas, short.gammas, [(1.0, 2.0, 6.0, 0.2)]).astype(np.float32)
    with pytest.raises(ValueError, match='short scan'):
        bp.recon_device(torch.tensor(s, device='cuda'), short, 64, 30.0, 1.0)
    twice = dx.FanBeamGeometry(N_channels=129, N_proj=200, gamma_fan=fan, SID=60.0, SDD=100.0, theta_tot=4 * np.pi)
    with pytest.raises(ValueError, match='exceeds'):
        bp.recon_device(torch.tensor(s, device='cuda'), twice, 64, 30.0, 1.0)
    lib = _native.load()
    one = torch.zeros(16, device='cuda')
    assert lib.dexct_fbp_parker(one.data_ptr(), 2, 1, 8, 3.2, 0.1, 0, 2, one.data_ptr(), None) == -1      # pi + fan = 3.84 > 3.2
    assert lib.dexct_fbp_parker(one.data_ptr(), 2, 1, 8, 6.3, 0.1, 0, 2, one.data_ptr(), None) == -1      # >= 2 pi
    # stacked rows
    ct = dx.FanBeamGeometry(N_channels=129, N_proj=240, gamma_fan=fan, SID=60.0, SDD=100.0, N_rows=3, theta_tot=4.4)
    a = disc_sino(ct.thetas, ct.gammas, [(2.0, 1.0, 6.0, 0.2)])
    b = disc_sino(ct.thetas, ct.gammas, [(-3.0, 0.0, 3.0, 0.4)])
    stack = np.stack([a, b, a + 2 * b], axis=1).astype(np.float32)
    img = bp.recon_device(torch.tensor(stack, device='cuda'), ct, 64, 30.0, 1.0).cpu().numpy()
    ref, _ = fo.get_recon(a.astype(np.float32), ct.thetas, ct.gammas, 60.0, 64, 30.0, 1.0, theta_tot=4.4)
    assert np.max(np.abs(img[0] - ref)) < 3e-5 * np.abs(ref).max()
    assert np.max(np.abs(img[2] - (img[0] + 2 * img[1]))) < 1e-5 * np.abs(img).max()
    # cone beam: a centred ball seen by a short and by a full rotation
    def ball_sino(cone):
        th, gam, rz = cone.thetas, cone.gammas, cone.row_z()
        out = np.zeros((th.size, rz.size, gam.size))
        R, mu = 6.0, 0.2
        for i, b_ in enumerate(th):
            src = np.array([60.0 * np.cos(b_), 60.0 * np.sin(b_), cone.src_z])
            ang = b_ + np.pi + gam
            for r, z in enumerate(rz):
                d = np.stack([100.0 * np.cos(ang), 100.0 * np.sin(ang), np.full(gam.size, z - cone.src_z)], -1)
                d /= np.linalg.norm(d, axis=-1, keepdims=True)
                t = -(d @ src)
                dist2 = src @ src - t * t
                out[i, r] = mu * 2 * np.sqrt(np.maximum(R * R - dist2, 0))
        return out.astype(np.float32)
    kw = dict(N_channels=129, N_proj=240, gamma_fan=fan, SID=60.0, SDD=100.0, N_rows=9, cone=True, h_iso=0.5)
    cs, cf = dx.FanBeamGeometry(theta_tot=4.4, **kw), dx.FanBeamGeometry(**kw)
    vs = bp.recon_device(torch.tensor(ball_sino(cs), device='cuda'), cs, 64, 30.0, 1.0).cpu().numpy()
    vf = bp.recon_device(torch.tensor(ball_sino(cf), device='cuda'), cf, 64, 30.0, 1.0).cpu().numpy()
    c = (np.arange(64) - 32 + 0.5) * (30 / 64)
    x, y = np.meshgrid(c, c)
    inner = x ** 2 + y ** 2 < 16.0
    assert vs.shape == vf.shape == (9, 64, 64) and np.isfinite(vs).all()
    assert abs(vs[4][inner].mean() - 0.2) < 4e-3 and abs(vf[4][inner].mean() - 0.2) < 4e-3
    assert np.abs(vs[3:6] - vf[3:6])[:, inner].mean() < 5e-3
