"""The reduced energy quadrature (dex-ct-sim_amd/quadrature.py): its error bound, re-checked here in float64 on points of the
domain that neither its linear programme nor its own validation saw."""
import os

import numpy as np
import pytest

import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, quadrature as q, synthetic

DET = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'dex-ct-sim_amd', 'input', 'detector',
                   'eta_eid_mv.bin')


def tables(kvps=(140, 80), eid=True):
    ct = dx.FanBeamGeometry(N_channels=64, N_proj=8, eid=eid, detector_file=DET)
    ph = synthetic.make_phantom(16, 1)
    return fp.merged_tables(ct, ph, [synthetic.kramers_spectrum(k) for k in kvps])


def independent_points(l_max, c_max, n, seed):
    rng = np.random.default_rng(seed)
    L = rng.uniform(0.0, 1.0, (n, len(l_max))) ** rng.choice([0.5, 1.0, 3.0], (n, 1)) * np.asarray(l_max)
    s = L.sum(axis=1)
    over = s > c_max
    L[over] *= (c_max / s[over] * rng.uniform(0.5, 1.0, over.sum()))[:, None]
    return L


@pytest.mark.parametrize('kvps,eid', [((140, 80), True), ((120,), False)])
def test_bound_holds_on_points_the_programme_never_saw(kvps, eid):
    E, mu, w = tables(kvps, eid)
    l_max, c_max = [72.4, 58.0, 44.5], 72.4
    cols, w_red, info = q.reduce_tables(mu, w, l_max, c_max)
    assert info['nodes'] == len(cols) < info['n_full'] // 3
    assert info['max_rel_err'] <= 1.0e-6
    assert np.all(w_red >= 0.0) and np.all(np.diff(cols) > 0)
    # every spectrum keeps its unattenuated signal (the point L = 0 is in the domain)
    assert np.allclose(w_red.sum(axis=1), w.sum(axis=1), rtol=1e-6, atol=0)
    # a spectrum's nodes are energies it weights itself (no weight appears above its kVp)
    assert not np.any((w_red > 0) & (w[:, cols] == 0))
    L = independent_points(l_max, c_max, 200000, seed=99)
    assert q.max_rel_error(mu, w, cols, w_red, L) <= 1.0e-6
    # thin, realistic rays (a few cm of bone in a water cylinder): the same bound
    L2 = np.stack([np.full(1000, 10.0), np.linspace(0, 41, 1000), np.linspace(0, 6, 1000)], axis=1)
    assert q.max_rel_error(mu, w, cols, w_red, L2) <= 1.0e-6
    # and the same call again comes from the cache
    assert q.reduce_tables(mu, w, l_max, c_max)[2] is info


def test_no_reduction_where_the_bound_cannot_be_checked():
    E, mu, w = tables()
    assert q.reduce_tables(np.tile(mu[:1], (5, 1)), w, [10.0] * 5, 10.0) is None            # more than MAX_MATERIALS rows
    assert q.reduce_tables(mu[:, :20], w[:, :20], [72.4, 58.0, 44.5], 72.4) is None           # nothing to gain: a short grid
    wn = w.copy()
    wn[0, 5] = -1.0
    assert q.reduce_tables(mu, wn, [72.4, 58.0, 44.5], 72.4) is None                          # not a quadrature with positive weights


def test_absent_material_and_opaque_rays():
    E, mu, w = tables((140,))
    # a material that does not occur (l_max 0) and one so dense that long paths fall below the signal floor
    mu2 = mu.copy()
    mu2[2] *= 40.0
    r = q.reduce_tables(mu2, w, [72.4, 0.0, 30.0], 72.4)
    if r is not None:                      # either no reduction, or one that holds where the signal is representable
        cols, w_red, info = r
        L = independent_points([72.4, 0.0, 30.0], 72.4, 50000, seed=5)
        assert q.max_rel_error(mu2, w, cols, w_red, L) <= 1.0e-6


def test_option_parsing():
    assert fp._want_reduced('reduced') and not fp._want_reduced('full') and not fp._want_reduced(None)
    with pytest.raises(ValueError):
        fp._want_reduced('fast')


def newton_tables():
    ct = dx.FanBeamGeometry(N_channels=8, N_proj=8, eid=True, detector_file=DET)
    from dex_ct_sim_amd import matdecomp as md
    return md.decomposition_tables(ct, synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80))


def test_gate_grid_and_start_array():
    """quadrature.newton_start_grid: a cell grid in data space that holds every ray of the domain (incl. water in a tissue /
    bone basis, whose second component is slightly negative); assemble_start: a cell needs the largest step count among its
    corners and those of its neighbours plus the margin, and is closed (infinity) around a corner that did not end by the
    rule or whose fixed point breaks the smoothness of its neighbours (another basin)."""
    _, i0, mus = newton_tables()
    p = q.newton_start_grid(i0, mus)
    head = p['head']
    n = int(head[3])
    assert n == q.GATE_CELLS and p['corner_g'].shape == ((n + 1) ** 2, 2) and np.all(p['corner_g'] > 0)
    rng = np.random.default_rng(8)
    a = np.stack([rng.uniform(0.05, 45, 5000), rng.uniform(0, 10, 5000)], 1)
    a[:1500, 1] = -0.008 * a[:1500, 0]                                   # water in a tissue / bone basis
    g = np.exp(-(a @ mus)) @ i0.T
    u = np.log(head[:2][None, :] / g) * head[2]
    fx = (np.log(u[:, 0]) - head[4]) * head[5]
    fy = (u[:, 1] / u[:, 0] - head[6]) * head[7]
    inside = u[:, 0] < 1.0
    assert fx[inside].min() >= 0 and fx[inside].max() < n and 0.15 * n < fy[inside].min() and fy[inside].max() < 0.85 * n
    # the corners' counts are what the header says: row = index along ln u0, column = index of the ratio
    i, j = 11, 40
    cg = p['corner_g'].reshape(n + 1, n + 1, 2)[i, j]
    u0 = np.exp(head[4] + i / head[5])
    assert np.allclose(np.log(head[:2] / cg) * head[2], [u0, u0 * (head[6] + j / head[7])], rtol=1e-12)
    # a smooth synthetic field of fixed points (made consistent: the corners' counts are set to the model's at those points),
    # one slow corner, one corner that did not end, one corner in another basin, one corner resting where it does not
    # reproduce its counts (a valley of the clipped likelihood)
    ii, jj = np.meshgrid(np.arange(n + 1.0) * 128.0 / n, np.arange(n + 1.0) * 128.0 / n, indexing='ij')      # (the field of the 128-cell grid this test was written on)
    roots = np.stack([0.15 * ii + 0.005 * ii * jj / 4, 0.1 * jj - 0.025 * ii], -1).reshape(-1, 2)
    p = dict(p, corner_g=np.exp(-(roots @ p['mus'])) @ p['i0'].T)
    steps = np.full((n + 1) ** 2, 17)
    steps[5 * (n + 1) + 7] = 30
    steps[30 * (n + 1) + 30] = 255
    roots = roots.copy()
    roots[45 * (n + 1) + 20] += [3.0, -2.0]
    roots[70 * (n + 1) + 60] += [1e-4, 0.0]
    start, share, stats = q.assemble_start(p, steps, roots)
    assert stats['walk_nonfinite_share'] == 0.0 and 0.0 < stats['walk_not_by_rule_share'] < 1e-3 and not q.pair_is_ill_posed(stats)
    assert np.allclose(start[8:10], np.log(start[0:2]), rtol=0, atol=0)          # ln of the open-beam signals, for the kernel's logarithm
    assert start.size == q.START_HEADER + 2 * (n + 1) ** 2 + 2 * n * n
    r0 = start[q.START_HEADER:q.START_HEADER + 2 * (n + 1) ** 2].reshape(n + 1, n + 1, 2)[:, :, 0]          # pairs (a0, a1)
    cells = start[q.START_HEADER + 2 * (n + 1) ** 2:].reshape(n, n, 2)                                        # pairs (need, radius)
    need, radius = cells[:, :, 0], cells[:, :, 1]
    ok = np.ones((n + 1) ** 2, bool)
    ok[[30 * (n + 1) + 30, 45 * (n + 1) + 20, 70 * (n + 1) + 60]] = False
    assert np.array_equal(r0.ravel()[ok], roots[ok, 0]) and r0[30, 30] == 0.0 and r0[70, 60] == 0.0         # (corners that do not count: zeroed)
    assert need[1, 1] == 17 + q.GATE_MARGIN and np.isinf(need[0, 5]) and np.isinf(need[7, n - 1]) and np.all(need[3:7, 5:9] == 30 + q.GATE_MARGIN) and need[2, 5] == need[7, 5] == 17 + q.GATE_MARGIN
    for ci, cj in ((30, 30), (45, 20), (70, 60)):
        assert np.all(np.isinf(need[ci - 2:ci + 2, cj - 2:cj + 2])) and np.isfinite(need[ci - 3, cj - 3]) and np.isfinite(need[ci + 2, cj])
    assert 0.93 < share < 1.0
    # the acceptance radius: a twentieth of the spread of the corners' fixed points
    c = roots.reshape(n + 1, n + 1, 2)
    want = max(np.abs(c[11, 11] - c[10, 10]).max(), np.abs(c[11, 10] - c[10, 11]).max(), np.abs(c[11, 10] - c[10, 10]).max(),
               np.abs(c[10, 11] - c[10, 10]).max(), np.abs(c[11, 11] - c[10, 11]).max(), np.abs(c[11, 11] - c[11, 10]).max())
    assert abs(radius[10, 10] - (q.GATE_RADIUS * want + 1e-9)) < 1e-12 and radius[29, 29] == 0.0


def test_gate_table_is_validated_at_the_cell_centres():
    """quadrature.validate_start: the reference's walk at the centre of every cell must end, within the cell's step budget,
    next to the Catmull-Rom interpolant the kernel would start from; a cell where it does not is closed with its neighbours."""
    _, i0, mus = newton_tables()
    p = q.newton_start_grid(i0, mus)
    n = int(p['head'][3])
    h = p['head']
    field = lambda x, t: np.stack([40.0 * np.exp(x) * (1.0 + 0.1 * t), 3.0 * np.exp(x) * (t - 1.2)], -1)      # smooth in (ln u0, ratio)
    xc, tc = h[4] + np.arange(n + 1) / h[5], h[6] + np.arange(n + 1) / h[7]
    roots = field(xc[:, None], tc[None, :]).reshape(-1, 2)
    p = dict(p, corner_g=np.exp(-(roots @ p['mus'])) @ p['i0'].T)
    start, share, _ = q.assemble_start(p, np.full((n + 1) ** 2, 17), roots)
    assert share > 0.85                                                    # (rows beyond GATE_U_MAX and the border are closed)
    xm, tm = h[4] + (np.arange(n) + 0.5) / h[5], h[6] + (np.arange(n) + 0.5) / h[7]
    centre_roots = field(xm[:, None], tm[None, :]).reshape(-1, 2)
    # the centres' counts must be the model's at those roots for the residual test: override what cell_centres would give
    g_c = np.exp(-(centre_roots @ p['mus'])) @ p['i0'].T
    orig = q.cell_centres
    try:
        q.cell_centres = lambda pieces: g_c
        ok, share_ok, n_bad = q.validate_start(start, p, np.full(n * n, 17), centre_roots)
        assert n_bad == 0 and share_ok == share                           # the interpolant is within the radius of a smooth field
        moved = centre_roots.copy().reshape(n, n, 2)
        moved[40, 50] += [0.5, 0.0]                                        # the walk from this centre ends somewhere else
        late = np.full((n, n), 17)
        late[60, 70] = 19                                                  # ... and this one needs more steps than the cell allows
        out, share_bad, n_bad = q.validate_start(start, p, late.ravel(), moved.reshape(-1, 2))
    finally:
        q.cell_centres = orig
    need = out[q.START_HEADER + 2 * (n + 1) ** 2:].reshape(n, n, 2)[:, :, 0]
    assert n_bad == 2 and share_bad < share
    for ci, cj in ((40, 50), (60, 70)):
        assert np.all(np.isinf(need[ci - 1:ci + 2, cj - 1:cj + 2])) and np.isfinite(need[ci - 2, cj]) and np.isfinite(need[ci, cj + 2])
    # cell_centres itself: the counts at (x_i + 1/2, t_j + 1/2)
    g = q.cell_centres(p).reshape(n, n, 2)
    u = np.log(h[:2] / g[7, 9]) * h[2]
    assert np.allclose([np.log(u[0]), u[1] / u[0]], [xm[7], tm[9]], rtol=1e-12)


def test_isolated_roots_are_told_from_families_of_them():
    """quadrature._counts_and_condition: the log-Jacobian of the forward model is well conditioned at physical thicknesses (an
    isolated root of the two equations) and singular when the two basis materials attenuate proportionally (every point of a
    line reproduces the counts) or when the exponent is clipped at every energy (no slope left)."""
    _, i0, mus = newton_tables()
    p = q.newton_start_grid(i0, mus)
    a = np.array([[0.01, 0.0], [1.0, 0.1], [20.0, 2.0], [40.0, 0.0], [5.0, 8.0], [30.0, -0.24]])
    g = np.exp(-(a @ p['mus'])) @ p['i0'].T
    resid, cond = q._counts_and_condition(p, a, g)
    assert resid.max() < 1e-14 and cond.max() < 200.0 < q.GATE_MAX_COND
    resid, _ = q._counts_and_condition(p, a + [0.0, 1e-3], g)
    assert resid.min() > 1e-5                                              # a point next to the root does not reproduce the counts
    same = dict(p, mus=np.stack([p['mus'][0], 2.0 * p['mus'][0]]))
    g2 = np.exp(-(a @ same['mus'])) @ same['i0'].T
    assert q._counts_and_condition(same, a, g2)[1].min() > 1e12            # a1 and 2 a0 are interchangeable: roots come in lines
    clipped = dict(p, mus=1e4 * p['mus'])
    far = np.array([[40.0, 5.0]])
    with np.errstate(all='ignore'):
        assert not np.isfinite(q._counts_and_condition(clipped, far, np.ones((1, 2)))[1][0]) or q._counts_and_condition(clipped, far, np.ones((1, 2)))[1][0] > 1e12


def test_kappa_table_of_the_one_step_acceptance():
    """quadrature.attach_kappa: per centre e1 / d1^2 (e1 = |probe - root|, not below 8 eps of the size; d1 = |probe - interpolant|),
    per cell KAPPA_SAFETY x the largest among the cell and the eight around it; infinity in closed cells, around a centre whose
    probe is not finite and around one whose single step does not land within stop_tol / 16; header [10] = 1, kappa appended."""
    _, i0, mus = newton_tables()
    p = q.newton_start_grid(i0, mus)
    n = int(p['head'][3])
    ii, jj = np.meshgrid(np.arange(n + 1.0) * 128.0 / n, np.arange(n + 1.0) * 128.0 / n, indexing='ij')
    roots = np.stack([0.15 * ii + 0.005 * ii * jj / 4, 0.1 * jj - 0.025 * ii], -1).reshape(-1, 2)
    p = dict(p, corner_g=np.exp(-(roots @ p['mus'])) @ p['i0'].T)
    start, _, _ = q.assemble_start(p, np.full((n + 1) ** 2, 17), roots)
    cells = start[q.START_HEADER + 2 * (n + 1) ** 2:].reshape(n, n, 2)
    s = q.centre_interpolant(start, n)
    probe = s + 1.0e-7                                                   # the step from the interpolant: d1 = 1e-7
    size = np.maximum(np.abs(probe).max(-1), 1.0)
    rc = probe + 1.0e-14 * size[:, :, None]                              # ... lands 1e-14 of the size from the root
    rc[40, 40] = probe[40, 40] + 1.0e-12 * size[40, 40]                  # one centre where the step leaves too much (> stop_tol / 16)
    probe[80, 80] = np.nan
    out, share = q.attach_kappa(start, p, rc.reshape(-1, 2), probe.reshape(-1, 2), 1e-12)
    assert out.size == start.size + n * n and out[10] == 1.0 and np.array_equal(out[:10], start[:10])
    assert np.array_equal(out[q.START_HEADER:start.size], start[q.START_HEADER:])
    kappa = out[start.size:].reshape(n, n)
    open_ = np.isfinite(cells[:, :, 0])
    assert np.all(np.isinf(kappa[~open_])) and 0.3 < share < open_.mean() + 1e-12
    i, j = 60, 50
    assert open_[i - 1:i + 2, j - 1:j + 2].all()
    want = q.KAPPA_SAFETY * np.max(1.0e-14 * size[i - 1:i + 2, j - 1:j + 2] / 1.0e-14)
    assert np.isclose(kappa[i, j], want, rtol=3e-2)                     # (1e-14 of the size is a dozen ulps: quantised)
    assert np.all(np.isinf(kappa[39:42, 39:42])) and np.all(np.isinf(kappa[79:82, 79:82])) and np.isfinite(kappa[43, 43])
