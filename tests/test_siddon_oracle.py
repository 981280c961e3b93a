"""Pins of the Siddon oracle.  The reference's projector is absent (parity unpinned, see
oracle/dexct_oracle.c), so the float64 textbook form is pinned analytically and the fixed-point
DDA form (the arithmetic the HIP kernels use) is pinned against the textbook form."""
import numpy as np
import pytest

from conftest import oracle_geom, small_scan
from oracle import c_oracle as co


def merged(vox, ln):
    if len(vox) == 0:
        return vox, ln
    keep = np.r_[True, vox[1:] != vox[:-1]]
    out = np.zeros(int(keep.sum()))
    np.add.at(out, np.cumsum(keep) - 1, ln)
    return vox[keep], out


def ray_line(ct, v, c):
    b, gm = ct.thetas[v], ct.gammas[c]
    s = np.array([ct.SID * np.cos(b), ct.SID * np.sin(b)])
    e = -np.array([np.cos(b + gm), np.sin(b + gm)])
    return s, e


def box_chord(s, e, hx, hy):
    t0, t1 = -np.inf, np.inf
    for p, d, h in ((s[0], e[0], hx), (s[1], e[1], hy)):
        if d == 0:
            if abs(p) >= h:
                return 0.0
            continue
        a, b = (-h - p) / d, (h - p) / d
        t0, t1 = max(t0, min(a, b)), min(t1, max(a, b))
    return max(0.0, t1 - t0)


@pytest.mark.parametrize('n,nv,nc', [(32, 36, 64), (48, 40, 50)])
def test_classic_sum_equals_box_chord(n, nv, nc):
    ct, ph = small_scan(n=n, n_views=nv, n_channels=nc)
    g = oracle_geom(ct, ph)
    vcs, ccs = ct.view_cs(), ct.chan_cs()
    for v in range(nv):
        for c in range(nc):
            _, ln = co.classic_ray(g, vcs, ccs, v, c)
            s, e = ray_line(ct, v, c)
            assert abs(ln.sum() - box_chord(s, e, 0.5 * n * ph.dx, 0.5 * n * ph.dy)) < 1e-9


def test_classic_disc_chord_converges():
    # water disc radius 0.4*extent: the path length in water tends to the analytic chord
    n = 256
    from dex_ct_sim_amd import synthetic
    ct, _ = small_scan(n=8, n_views=8, n_channels=64)
    ph = synthetic.make_phantom(n, 1, n_spheres=0)
    g = oracle_geom(ct, ph)
    mu = np.array([[0.0], [1.0], [0.0]])
    w = np.array([[1.0]])
    _, pl = co.project_classic(g, ct.view_cs(), ct.chan_cs(), 0, 8, ph.volume, mu, w, want_pathlen=True)
    R = 0.4 * 51.2
    for v in range(8):
        for c in range(64):
            s, e = ray_line(ct, v, c)
            d = abs(s[0] * e[1] - s[1] * e[0])
            # the voxelised disc lies between the discs of radius R -+ one voxel diagonal
            h = ph.dx * np.sqrt(2.0)
            lo = 2 * np.sqrt(max((R - h) ** 2 - d * d, 0.0)) if d < R - h else 0.0
            hi = 2 * np.sqrt(max((R + h) ** 2 - d * d, 0.0))
            assert lo - 1e-9 <= pl[v, 0, c, 1] <= hi + 1e-9
    # and the total over materials is exactly the box chord
    s, e = ray_line(ct, 3, 20)
    assert abs(pl[3, 0, 20].sum() - box_chord(s, e, 25.6, 25.6)) < 1e-9


def test_rotation_symmetry():
    # a 4-fold symmetric phantom seen from views 90 degrees apart gives identical path lengths
    n = 40
    ct, ph = small_scan(n=n, n_views=8, n_channels=48)
    vol = np.zeros((1, n, n), np.uint8)
    yy, xx = np.mgrid[0:n, 0:n]
    vol[0][(np.abs(xx - n / 2 + .5) < 12) & (np.abs(yy - n / 2 + .5) < 12)] = 1
    vol[0][(np.abs(xx - n / 2 + .5) < 4) & (np.abs(yy - n / 2 + .5) < 4)] = 2
    ph.volume = vol
    g = oracle_geom(ct, ph)
    mu = np.array([[0.0], [0.2], [0.5]])
    _, pl = co.project_classic(g, ct.view_cs(), ct.chan_cs(), 0, 8, ph.volume, mu, np.array([[1.0]]), True)
    for v in (0, 1):
        assert np.max(np.abs(pl[v] - pl[v + 2])) < 1e-9
        assert np.max(np.abs(pl[v] - pl[v + 4])) < 1e-9


@pytest.mark.parametrize('n,nv,nc', [(64, 90, 128), (50, 72, 97)])
def test_dda_index_sequence_equals_classic(n, nv, nc):
    """Voxel-index sequences agree exactly (segments shorter than 1e-7 cm - ties at voxel corners -
    excluded); segment lengths to 3e-7 cm; incl. the axis-aligned views 0, 90, 180, 270 degrees."""
    ct, ph = small_scan(n=n, n_views=nv, n_channels=nc)
    g = oracle_geom(ct, ph)
    vcs, ccs = ct.view_cs(), ct.chan_cs()
    plan = co.plan(g, vcs, ccs, 0, nv)
    for v in range(nv):
        for c in range(nc):
            vc, lc = co.classic_ray(g, vcs, ccs, v, c)
            p = plan[v * nc + c]
            vd, ld = co.dda_ray(g, p, 0)
            vm, lm = merged(vd, ld.astype(np.float64) * float(p['len_per_u']))
            if len(vc) and len(vm) and vc[0] != vm[0]:
                vm, lm = vm[::-1], lm[::-1]
            cm, dm = lc > 1e-7, lm > 1e-7
            # the one genuine tie: a ray that runs ALONG a grid plane (axis-parallel to 1e-9 and on the
            # plane to 1e-9 voxel, e.g. the central channel at 180 degrees with an even grid) lies in both
            # neighbouring voxel rows at once; float64 and 40-bit fixed point may pick different rows.
            v0 = float(p['V0']) / 2.0 ** 40
            on_plane = abs(float(p['SV'])) < 2.0 ** 10 and abs(v0 - round(v0)) < 1e-9
            if on_plane:
                assert len(vc[cm]) == len(vm[dm])
                continue
            assert np.array_equal(vc[cm], vm[dm]), (v, c)
            assert np.max(np.abs(lc[cm] - lm[dm]), initial=0.0) < 3e-7
            assert abs(lc.sum() - float(p['chord_u']) * float(p['len_per_u'])) < 2e-5


def test_dda_projection_close_to_classic():
    ct, ph = small_scan(n=64, nz=2, n_views=45, n_channels=80, n_rows=2)
    g = oracle_geom(ct, ph)
    E = np.arange(20.0, 121.0, 5.0)
    mu = ph.mu_table(E)
    w = np.stack([np.exp(-((E - 60) / 25) ** 2), np.exp(-((E - 90) / 20) ** 2)]) * 1e5
    c1, p1 = co.project_classic(g, ct.view_cs(), ct.chan_cs(), 0, 45, ph.volume, mu, w, True)
    c2, p2 = co.project_dda(g, ct.view_cs(), ct.chan_cs(), 0, 45, ph.volume, mu, w, True)
    assert np.max(np.abs(p1 - p2)) < 2e-5
    assert np.max(np.abs(c1 - c2) / c1) < 2e-6


def test_miss_and_empty_rays():
    # a wide fan: the outermost channels miss the grid entirely
    import dex_ct_sim_amd as dx
    ct, ph = small_scan(n=16, n_views=4, n_channels=32)
    ct = dx.FanBeamGeometry(N_channels=32, N_proj=4, gamma_fan=2.4, SID=60.0, SDD=100.0)
    g = oracle_geom(ct, ph)
    plan = co.plan(g, ct.view_cs(), ct.chan_cs(), 0, 4)
    assert (plan['n_slabs'] == 0).any() and (plan['n_slabs'] > 0).any()
    for k in np.nonzero(plan['n_slabs'] == 0)[0][:8]:
        v, c = divmod(int(k), 32)
        vc, lc = co.classic_ray(g, ct.view_cs(), ct.chan_cs(), v, c)
        assert lc.sum() < 1e-9


def test_classic_against_brute_force_sampling():
    """Independent check of the textbook form on a random phantom: path length per material from dense point
    sampling along the ray (no plane arithmetic at all) agrees to the sampling step."""
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd.system import AIR, BONE, WATER
    rng = np.random.default_rng(21)
    n = 24
    vol = rng.integers(0, 3, (1, n, n), dtype=np.uint8)
    ph = dx.VoxelPhantom.from_array('rand', vol, [AIR, WATER, BONE], dx=0.7, dy=0.7, dz=0.7)
    ct = dx.FanBeamGeometry(N_channels=21, N_proj=17, gamma_fan=0.3, SID=40.0, SDD=80.0)
    g = oracle_geom(ct, ph)
    mu = np.array([[0.0], [1.0], [0.0]])
    _, pl = co.project_classic(g, ct.view_cs(), ct.chan_cs(), 0, 17, ph.volume, mu, np.array([[1.0]]), True)
    h = 0.5 * n * 0.7
    t = np.linspace(0.0, 80.0, 400001)
    dt = t[1] - t[0]
    for v in range(0, 17, 4):
        for c in range(0, 21, 5):
            s, e = ray_line(ct, v, c)
            x, y = s[0] + t * e[0], s[1] + t * e[1]
            ix, iy = np.floor((x + h) / 0.7).astype(int), np.floor((y + h) / 0.7).astype(int)
            ok = (ix >= 0) & (ix < n) & (iy >= 0) & (iy < n)
            ids = np.full(t.size, -1)
            ids[ok] = vol[0, iy[ok], ix[ok]]
            for m in range(3):
                assert abs((ids == m).sum() * dt - pl[v, 0, c, m]) < 100 * dt       # <= ~50 boundary samples


def test_cone_oracle_box_chords_and_mirror():
    """3-D textbook Siddon: total length = analytic chord through the grid box (1e-9); the fixed-point
    mirror of the cone kernel agrees with it to 2e-5 cm per material and 2e-6 in counts."""
    ct, ph = small_scan(n=40, nz=24, n_views=20, n_channels=48, n_rows=10)
    g = oracle_geom(ct, ph)
    row_z = (np.arange(10) - 4.5) * 1.2
    E = np.array([50.0, 80.0])
    mu, w = ph.mu_table(E), np.array([[1e4, 2e4]])
    c1, p1 = co.project_cone(g, ct.view_cs(), ct.chan_cs(), 0, 20, row_z, 0.3, ph.volume, mu, w, dda=False, n_threads=4)
    c2, p2 = co.project_cone(g, ct.view_cs(), ct.chan_cs(), 0, 20, row_z, 0.3, ph.volume, mu, w, dda=True, n_threads=4)
    assert np.abs(p1 - p2).max() < 2e-5 and np.max(np.abs(c1 - c2) / c1) < 2e-6
    b, gm, zd = ct.thetas[:, None, None], ct.gammas[None, None, :], row_z[None, :, None]
    sx, sy, dx_, dy_, dz_ = np.broadcast_arrays(ct.SID * np.cos(b), ct.SID * np.sin(b), -np.cos(b + gm) * ct.SDD,
                                                -np.sin(b + gm) * ct.SDD, zd - 0.3)

    def slab(p0, d, h):
        with np.errstate(divide='ignore', invalid='ignore'):
            a0, a1 = (-h - p0) / d, (h - p0) / d
        return np.minimum(a0, a1), np.maximum(a0, a1)
    hx, hz = 0.5 * 40 * ph.dx, 0.5 * 24 * ph.dz
    (lx, ux), (ly, uy), (lz, uz) = slab(sx, dx_, hx), slab(sy, dy_, hx), slab(np.full_like(dz_, 0.3), dz_, hz)
    chord = np.maximum(np.minimum(np.minimum(ux, uy), uz) - np.maximum(np.maximum(lx, ly), lz), 0)
    chord = chord * np.sqrt(dx_ ** 2 + dy_ ** 2 + dz_ ** 2)
    assert np.abs(p1.sum(-1) - chord).max() < 1e-9
