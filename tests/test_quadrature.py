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


def test_short_tables_of_the_coarse_newton_pass():
    """quadrature.coarse_newton_tables: a sixth of the energies, non-negative weights on the spectra's own energies, the
    forward model within a few 1e-6 over the decomposition's domain (attenuation down to exp(-16))."""
    _, i0, mus = newton_tables()
    cols, i0_s = q.coarse_newton_tables(i0, mus)
    assert 16 <= len(cols) <= i0.shape[1] // 4 and np.all(i0_s >= 0) and not np.any((i0_s > 0) & (i0[:, cols] == 0))
    rng = np.random.default_rng(3)
    mu_min = mus[:, (i0 > 0).any(0)].min(1)
    a = q._domain_points(16.0 / mu_min, 16.0, 40000, 600, rng, mu_min)
    assert q.max_rel_error(mus, i0, cols, i0_s, a) < 1e-5
    assert q.coarse_newton_tables(i0, -mus) is None and q.coarse_newton_tables(-i0, mus) is None


def test_start_polynomial_and_gate_table():
    """quadrature.newton_start_polynomial: start values within a few 1e-2 of max(|a|, 1) over the domain, including rays with
    a slightly negative second component; gate_table: a cell needs the largest step count among its corners and those of
    its neighbours plus the margin, and is closed (infinity) when any of them did not arrive at the truth."""
    _, i0, mus = newton_tables()
    p = q.newton_start_polynomial(i0, mus)
    head, coef = p['head'], p['coef']
    deg, n = int(head[3]), int(head[4])
    assert n == q.GATE_CELLS and coef.size == (deg + 1) * (deg + 2) and p['corners'].shape == ((n + 1) ** 2, 2)
    rng = np.random.default_rng(8)
    a = np.stack([rng.uniform(0, 45, 5000), rng.uniform(0, 10, 5000)], 1)
    a[:1500, 1] = -0.008 * a[:1500, 0]                                   # water in a tissue / bone basis
    g = np.exp(-(a @ mus)) @ i0.T
    u = np.log(head[:2][None, :] / g) * head[2]
    terms = [(i, j) for i in range(deg + 1) for j in range(deg + 1 - i)]
    V = np.stack([u[:, 0] ** i * u[:, 1] ** j for i, j in terms], 1)
    s = np.stack([V @ coef[:len(terms)], V @ coef[len(terms):]], 1)
    assert (np.abs(s - a).max(1) / np.maximum(np.abs(a).max(1), 1.0)).max() < 0.05
    # every such ray lies inside the cell grid (skewed coordinates), away from its lower edge
    f0 = (a[:, 0] * head[7] - head[5]) * head[6]
    f1 = ((a[:, 1] + head[9] * a[:, 0]) * head[8] - head[5]) * head[6]
    assert f0.min() >= 0 and f1.min() >= 0 and f0.max() < n and f1.max() < n
    steps = np.full((n + 1) ** 2, 17)
    steps[5 * (n + 1) + 7] = 30                                           # one slow corner
    found = p['corners'].copy()
    found[20 * (n + 1) + 20] += 1e-3                                      # one corner that ended somewhere else
    start, share = q.gate_table(p, steps, found)
    need = start[q.START_HEADER + coef.size:].reshape(n, n)
    assert start.size == q.START_HEADER + coef.size + n * n
    inside = ~np.isnan(p['corner_g'][:, 0]).reshape(n + 1, n + 1)
    assert need[0, 0] == 17 + q.GATE_MARGIN and np.all(need[3:7, 5:9] == 30 + q.GATE_MARGIN) and need[2, 5] == need[7, 5] == 17 + q.GATE_MARGIN
    assert np.all(np.isinf(need[18:22, 18:22])) and np.isfinite(need[17, 17]) and np.isfinite(need[22, 18]) == bool(inside[22:25, 18:21].all())
    assert np.all(np.isinf(need[~(inside[:-1, :-1] & inside[1:, 1:])]))   # cells that reach outside the domain are closed
    assert 0.3 < share < 0.6
