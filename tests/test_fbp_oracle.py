"""Pins of the FBP oracle (parity unpinned: the reference's get_recon source is absent).  Analytic
sinograms of discs must reconstruct to the disc densities."""
import numpy as np
import pytest

from oracle import fbp_oracle as fo

SID = 60.0


def disc_sino(thetas, gammas, discs):
    s = np.zeros((thetas.size, gammas.size))
    for i, b in enumerate(thetas):
        sx, sy = SID * np.cos(b), SID * np.sin(b)
        ang = b + np.pi + gammas
        ex, ey = np.cos(ang), np.sin(ang)
        for x0, y0, R, mu in discs:
            d = (x0 - sx) * ey - (y0 - sy) * ex
            s[i] += mu * 2 * np.sqrt(np.maximum(R * R - d * d, 0))
    return s


def scan(n_views=360, n_ch=257):
    dg = 0.8230337 / n_ch
    return np.arange(n_views) * 2 * np.pi / n_views, (np.arange(n_ch) - (n_ch - 1) / 2) * dg


def test_discs_reconstruct_to_their_density():
    th, gam = scan()
    s = disc_sino(th, gam, [(0, 0, 10.0, 0.2), (5.0, -3.0, 2.0, 0.3)])
    raw, hu = fo.get_recon(s, th, gam, SID, 128, 40.0, 1.0, mu_water=0.2)
    c = (np.arange(128) - 64 + 0.5) * (40 / 128)
    x, y = np.meshgrid(c, c)
    bg = (x ** 2 + y ** 2 < 8.0 ** 2) & ~((x - 5) ** 2 + (y + 3) ** 2 < 2.6 ** 2)
    small = (x - 5) ** 2 + (y + 3) ** 2 < 1.5 ** 2
    assert abs(raw[bg].mean() - 0.2) < 1e-4
    assert abs(raw[small].mean() - 0.5) < 1e-3        # x = +5, y = -3: orientation [iy, ix] is right
    assert abs(raw[x ** 2 + y ** 2 > 12 ** 2].mean()) < 1e-3
    assert abs(hu[bg].mean()) < 0.5 and abs(hu[small].mean() - 1500) < 5


def test_ramp_cutoff_smooths_but_keeps_mean():
    th, gam = scan(180, 129)
    s = disc_sino(th, gam, [(0, 0, 8.0, 0.25)])
    full, _ = fo.get_recon(s, th, gam, SID, 64, 30.0, 1.0)
    soft, _ = fo.get_recon(s, th, gam, SID, 64, 30.0, 0.5)
    c = (np.arange(64) - 32 + 0.5) * (30 / 64)
    x, y = np.meshgrid(c, c)
    inner = x ** 2 + y ** 2 < 5.0 ** 2
    assert abs(full[inner].mean() - 0.25) < 1e-3 and abs(soft[inner].mean() - 0.25) < 5e-3
    # a lower cutoff blurs the disc edge (and rings inside: a hard band limit, not an apodisation)
    assert np.abs(np.diff(soft[32])).max() < np.abs(np.diff(full[32])).max()


def test_ramlak_taps():
    g = fo.ramp_taps(5, 0.01, 1.0)          # n = -4..4: classic Ram-Lak times the equiangular factor
    assert np.isclose(g[4], 0.5 / (4 * 0.01 ** 2))
    assert np.allclose(g[[2, 6]], 0.0, atol=1e-9)                  # even offsets vanish
    assert np.isclose(g[5], 0.5 * (0.01 / np.sin(0.01)) ** 2 * (-1 / (np.pi ** 2 * 0.01 ** 2)))


def test_windowed_ramp_taps_product_vs_oracle_and_properties():
    """Apodised ramps (Shepp-Logan 'sinc' - the reference's README names a sinc window -, cosine, Hann, Hamming): the
    product's Gauss-Legendre taps agree with the oracle's QUADPACK oscillatory rule; every ramp has no DC response;
    a window only removes high frequencies, so h(0) = 2 int f W df shrinks in the order rect > sinc > cosine > hann."""
    from dex_ct_sim_amd import back_project as bp
    n, dg = 96, 0.8230337 / 96
    h0 = {}
    for window in ('rect', 'sinc', 'cosine', 'hann', 'hamming'):
        for ramp in (1.0, 0.55):
            got = bp.ramp_taps(n, dg, ramp, window)
            ref = fo.ramp_taps(n, dg, ramp, window)
            assert got.shape == (2 * n - 1,)
            assert np.max(np.abs(got - ref)) < 1e-9 * np.abs(ref).max(), (window, ramp)
            assert np.array_equal(got, got[::-1])
        long = bp.ramp_taps(2000, dg, 1.0, window)
        t = np.arange(-1999, 2000) * dg
        with np.errstate(invalid='ignore', divide='ignore'):
            h = long / np.where(t == 0, 0.5, 0.5 * (t / np.sin(t)) ** 2)          # back to the parallel-beam taps
        assert abs(h.sum()) < 2e-3 * h[1999], window                             # H(0) = 0 up to the truncated tails
        h0[window] = h[1999]
    assert h0['rect'] > h0['sinc'] > h0['cosine'] > h0['hann']
    with pytest.raises(ValueError):
        bp.ramp_taps(16, dg, 1.0, 'boxcar')


def ball_cone_sino(ct, centre, radius, mu):
    """Analytic cone-beam projections of a uniform ball: mu * chord of the 3-D ray source -> (view, row, channel)."""
    out = np.zeros((ct.N_proj, ct.N_rows, ct.N_channels))
    rz = ct.row_z()
    for i, b in enumerate(ct.thetas):
        src = np.array([ct.SID * np.cos(b), ct.SID * np.sin(b), ct.src_z])
        for r in range(ct.N_rows):
            e = np.stack([-np.cos(b + ct.gammas), -np.sin(b + ct.gammas),
                          np.full(ct.N_channels, (rz[r] - ct.src_z) / ct.SDD)], axis=1)
            e /= np.linalg.norm(e, axis=1, keepdims=True)
            oc = np.asarray(centre, dtype=np.float64) - src
            along = e @ oc
            d2 = oc @ oc - along ** 2
            out[i, r] = mu * 2.0 * np.sqrt(np.maximum(radius ** 2 - d2, 0.0))
    return out


def test_fdk_oracle_reconstructs_a_ball():
    """Pin of the FDK oracle: analytic projections of a uniform ball reconstruct to its attenuation, exactly in the
    source plane (where Feldkamp's algorithm is the fan algorithm) and to a few percent off it."""
    import dex_ct_sim_amd as dx
    ct = dx.FanBeamGeometry(N_channels=97, N_proj=120, gamma_fan=0.6, SID=60.0, SDD=100.0, h_iso=0.5, N_rows=28,
                            cone=True, src_z=0.0)
    s = ball_cone_sino(ct, (1.0, -0.5, 0.0), 5.0, 0.2)
    zs = np.array([-2.0, 0.0, 1.5])
    vol = fo.fdk_recon(s, ct.thetas, ct.gammas, ct.SID, ct.SDD, ct.row_z(), ct.src_z, 48, 24.0, 1.0, zs)
    c = (np.arange(48) - 24 + 0.5) * 0.5
    x, y = np.meshgrid(c, c)
    for k, z in enumerate(zs):
        rad = np.sqrt(max(25.0 - z * z, 0.0))
        inside = (x - 1.0) ** 2 + (y + 0.5) ** 2 < (rad - 1.2) ** 2
        outside = (x - 1.0) ** 2 + (y + 0.5) ** 2 > (rad + 1.2) ** 2
        tol = 0.004 if z == 0.0 else 0.012
        assert abs(vol[k][inside].mean() - 0.2) < tol, (z, vol[k][inside].mean())
        assert abs(vol[k][outside].mean()) < 0.01, (z, vol[k][outside].mean())


def short_scan(theta_tot, n_views=400, n_ch=257):
    dg = 0.8230337 / n_ch
    return np.arange(n_views) * theta_tot / n_views, (np.arange(n_ch) - (n_ch - 1) / 2) * dg


@pytest.mark.parametrize('extra', [0.0, 0.35, 1.2])
def test_parker_weights_of_a_conjugate_pair_add_up_to_one(extra):
    """In this build's geometry ray (beta, gamma) is seen again as (beta + pi + 2 gamma, -gamma): wherever both lie in the
    scan their weights add up to 1, a ray seen once has weight 1, and the weights are continuous."""
    fan = 0.8230337
    theta_tot = np.pi + fan + extra
    G = 0.5 * (theta_tot - np.pi)
    rng = np.random.default_rng(0)
    beta = rng.uniform(0, theta_tot, 20000)
    gam = rng.uniform(-fan / 2, fan / 2, 20000)
    w = np.array([fo.parker_weights([b], [g], theta_tot)[0, 0] for b, g in zip(beta[:3000], gam[:3000])])
    bp = beta[:3000] + np.pi + 2 * gam[:3000]
    bm = beta[:3000] - np.pi + 2 * gam[:3000]               # (the partner of the late views lies earlier)
    partner = np.where(bp <= theta_tot, bp, np.where(bm >= 0, bm, np.nan))
    wp = np.array([fo.parker_weights([b], [-g], theta_tot)[0, 0] if np.isfinite(b) else 0.0 for b, g in zip(partner, gam[:3000])])
    assert np.allclose(w + wp, 1.0, atol=1e-12)
    assert (np.isnan(partner)).any() and (np.isfinite(partner)).any()
    # continuity along beta for a fixed channel
    b = np.linspace(0, theta_tot, 20001)
    for g in (-0.4, 0.0, 0.37):
        col = fo.parker_weights(b, [g], theta_tot)[:, 0]
        assert np.abs(np.diff(col)).max() < 2e-2 and col.min() >= 0 and col.max() <= 1 + 1e-15      # (steep near the fan edge, never a jump)
    with pytest.raises(ValueError):
        fo.parker_weights([0.0], [-fan / 2, fan / 2], np.pi + 0.9 * fan)
    assert np.isclose(G, 0.5 * (fan + extra))


@pytest.mark.parametrize('theta_tot', [np.pi + 0.8230337, 4.5, 6.0])
def test_short_scan_reconstructs_discs(theta_tot):
    """A short scan (pi + fan angle), a longer one and an almost full one reconstruct the analytic discs to their
    densities and agree with the 2 pi reconstruction of the same object."""
    th, gam = short_scan(theta_tot)
    discs = [(0, 0, 10.0, 0.2), (5.0, -3.0, 2.0, 0.3), (-6.0, 4.0, 1.5, -0.1)]
    s = disc_sino(th, gam, discs)
    raw, _ = fo.get_recon(s, th, gam, SID, 128, 40.0, 1.0, theta_tot=theta_tot)
    th2, _ = scan(400, 257)
    full, _ = fo.get_recon(disc_sino(th2, gam, discs), th2, gam, SID, 128, 40.0, 1.0)
    c = (np.arange(128) - 64 + 0.5) * (40 / 128)
    x, y = np.meshgrid(c, c)
    bg = (x ** 2 + y ** 2 < 8.0 ** 2) & ~((x - 5) ** 2 + (y + 3) ** 2 < 2.6 ** 2) & ~((x + 6) ** 2 + (y - 4) ** 2 < 2.1 ** 2)
    small = (x - 5) ** 2 + (y + 3) ** 2 < 1.5 ** 2
    assert abs(raw[bg].mean() - 0.2) < 5e-4
    assert abs(raw[small].mean() - 0.5) < 2e-3
    assert abs(raw[x ** 2 + y ** 2 > 12 ** 2].mean()) < 1e-3
    inner = x ** 2 + y ** 2 < 9.0 ** 2
    assert np.abs(raw - full)[inner].mean() < 2e-3                   # the same image up to discretisation
