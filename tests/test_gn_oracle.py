"""The GN oracle (NumPy and C restatements of matdecomp.py:87-207) against golden vectors captured
from the real reference (tests/golden/make_goldens.py)."""
import types

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import gn_oracle as go

ATOL = 1e-10    # relative to max(|a|, 1): rounding-level agreement, not bitwise


def close(a, b, tol=ATOL):
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)) < tol


@pytest.mark.parametrize('ci', [0, 1, 2])
def test_tables_match_reference_bitwise(golden, ci):
    g = golden
    ee, dE, i0 = go.decomposition_tables(g[f'gn{ci}_det_E'], g[f'gn{ci}_det_eta'], bool(g[f'gn{ci}_eid']),
                                         g[f'gn{ci}_spec1_E'], g[f'gn{ci}_spec1_I0'], g[f'gn{ci}_spec2_E'],
                                         g[f'gn{ci}_spec2_I0'])
    assert np.array_equal(ee, g[f'gn{ci}_ee'])
    assert np.array_equal(i0, g[f'gn{ci}_i0'])
    assert bool(g[f'gn{ci}_i0_tiled_same'])       # the reference tiles one spectrum over all channels
    assert ee.size == (239 if ci == 1 else 140)


@pytest.mark.parametrize('ci', [0, 1, 2])
@pytest.mark.parametrize('n_iters', [1, 2, 50])
def test_numpy_oracle_trajectory(golden, ci, n_iters):
    g = golden
    a = go.newton_solve(g[f'gn{ci}_g'], g[f'gn{ci}_i0'], g[f'gn{ci}_mus'], n_iters)
    assert close(a, g[f'gn{ci}_a_iters{n_iters}'])


@pytest.mark.parametrize('ci', [0, 2])
def test_numpy_oracle_iter5(golden, ci):
    g = golden
    a = go.newton_solve(g[f'gn{ci}_g'], g[f'gn{ci}_i0'], g[f'gn{ci}_mus'], 5)
    assert close(a, g[f'gn{ci}_a_iters5'])


@pytest.mark.parametrize('ci', [0, 1, 2])
@pytest.mark.parametrize('n_iters', [1, 2, 50])
def test_c_oracle_trajectory(golden, ci, n_iters):
    g = golden
    a = co.gn_decompose(g[f'gn{ci}_g'][0], g[f'gn{ci}_g'][1], g[f'gn{ci}_i0'], g[f'gn{ci}_mus'], n_iters)
    assert close(a, g[f'gn{ci}_a_iters{n_iters}'], 1e-9)


@pytest.mark.parametrize('n_iters', [1, 3, 30])
def test_channel_dependent_i0(golden, n_iters):
    g = golden
    a = go.newton_solve(g['opt_g'], g['opt_i0'], g['opt_mus'], n_iters)
    assert close(a, g[f'opt_a_iters{n_iters}'])


@pytest.mark.parametrize('ci', [0, 1, 2])
def test_get_basismat_sinos_mask(golden, ci):
    g = golden
    ct = types.SimpleNamespace(det_E=g[f'gn{ci}_det_E'], det_eta_E=g[f'gn{ci}_det_eta'], eid=bool(g[f'gn{ci}_eid']))
    s1 = types.SimpleNamespace(E=g[f'gn{ci}_spec1_E'], I0=g[f'gn{ci}_spec1_I0'])
    s2 = types.SimpleNamespace(E=g[f'gn{ci}_spec2_E'], I0=g[f'gn{ci}_spec2_I0'])
    basis = lambda ee: g[f'gn{ci}_mus']
    for key, kw in (('50', dict(n_iters=50)), ('default', {}), ('thresh50', dict(n_iters=50, mask_thresh=0.5))):
        m1, m2 = go.get_basismat_sinos(ct, g[f'gn{ci}_g'][0].copy(), g[f'gn{ci}_g'][1].copy(), s1, s2, basis, **kw)
        r1, r2 = g[f'gn{ci}_mat1_{key}'], g[f'gn{ci}_mat2_{key}']
        assert np.array_equal(m1 == 0, r1 == 0) and np.array_equal(m2 == 0, r2 == 0)   # masked pixels exactly 0
        assert close(m1, r1) and close(m2, r2)


def test_noise_free_truth_recovered(golden):
    g = golden
    a = go.newton_solve(g['gn0_g'], g['gn0_i0'], g['gn0_mus'], 50)
    t = g['gn0_a_true']
    assert np.max(np.abs(a - t)) < 1e-9


def test_oracle_on_unscreened_live_default_pair():
    """tests/golden/ref_extra.npz (make_goldens_r2.py): detunedMV / 80 kV pixels drawn once, no redraw loop.  On every
    pixel the reference completes and answers stably (spurious roots included) the oracle agrees to rounding; the
    pixels on which the reference raises LinAlgError come out non-finite here in all but (at most) two cases -
    the closed-form 2x2 solve does not flag the exactly singular pivot LAPACK reports (matdecomp.py:125)."""
    import os
    from conftest import GOLDEN
    e = np.load(os.path.join(GOLDEN, 'ref_extra.npz'))
    ok = ~e['uns_raised'] & ~e['uns_ill']
    with np.errstate(all='ignore'):
        a_np = go.newton_solve(e['uns_g'], e['uns_i0'], e['uns_mus'], int(e['uns_n_iters']))
        a_c = co.gn_decompose(e['uns_g'][0].ravel(), e['uns_g'][1].ravel(), e['uns_i0'], e['uns_mus'],
                              int(e['uns_n_iters'])).reshape(a_np.shape)
    assert close(a_np[ok], e['uns_a50'][ok]) and close(a_c[ok], e['uns_a50'][ok])
    assert (np.abs(e['uns_a50'][ok] - e['uns_a_true'][ok]).max(-1) > 1.0).sum() > 0     # spurious roots are in the set
    r = e['uns_raised']
    assert (~np.isfinite(a_c[r]).all(-1)).sum() >= r.sum() - 2


def test_nan_count_masks_nothing(golden):
    """One NaN in sinogram 1: np.max is NaN, every comparison False, nothing masked (matdecomp.py:195-196)."""
    import os
    from conftest import GOLDEN
    e, g = np.load(os.path.join(GOLDEN, 'ref_extra.npz')), golden
    ct = types.SimpleNamespace(det_E=g['gn0_det_E'], det_eta_E=g['gn0_det_eta'], eid=bool(g['gn0_eid']))
    s1 = types.SimpleNamespace(E=g['gn0_spec1_E'], I0=g['gn0_spec1_I0'])
    s2 = types.SimpleNamespace(E=g['gn0_spec2_E'], I0=g['gn0_spec2_I0'])
    basis = lambda ee: g['gn0_mus']
    with np.errstate(all='ignore'):
        m1, m2 = go.get_basismat_sinos(ct, e['nan_g'][0].copy(), e['nan_g'][1].copy(), s1, s2, basis, n_iters=30)
    fin = np.isfinite(e['nan_mat1'])
    assert np.array_equal(np.isfinite(m1), fin) and (~fin).sum() == 1
    assert np.array_equal(m1 == 0, e['nan_mat1'] == 0)
    assert close(m1[fin], e['nan_mat1'][fin]) and close(m2[fin], e['nan_mat2'][fin])
