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
