"""ORACLE (test infrastructure, never shipped or timed as the product).

CPU restatement, in NumPy float64, of the reference's Gauss-Newton basis-material
decomposition.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module.

Pinned against golden vectors captured from the real reference
(tests/golden/gn_reference.npz, generator tests/golden/make_goldens.py).

Reference lines restated (paths relative to /root/reference):
  newton_solve        <- matdecomp.py:87-127  optimize_sino_cpu
  decomposition_tables<- matdecomp.py:136-160 do_matdecomp_gn (table building)
  do_matdecomp_gn     <- matdecomp.py:130-164
  get_basismat_sinos  <- matdecomp.py:167-207

The restatement is organised per pixel (every detector pixel is an independent
2-unknown problem) instead of the reference's per-view broadcast, and the 2x2
system is solved in closed form; agreement with the reference is therefore to
rounding (1e-12 relative in the tests), not bitwise.
"""
import numpy as np

EPS_INIT = 1e-6          # matdecomp.py:98-99
CLIP = 700.0             # matdecomp.py:116


def newton_solve(sino_gg, i0, mus, n_iters, return_sensitivity=False):
    """Per-pixel Newton iterations on the Poisson negative log-likelihood.

    sino_gg : [2, nViews, nBins] measured counts
    i0      : [2, nBins, nE] (reference layout) or [2, nE] effective spectra
    mus     : [2, nE] basis mass-attenuation tables
    returns : [nViews, nBins, 2] density line integrals  (matdecomp.py:127)

    ``return_sensitivity``: also return a dict of per-pixel arrays - the stability screen of tools/soak_gn.py:
      'walk'       the SUM over all iterations of cond(H_k) * |step_k| / max(|a_(k+1)|, 1); times the machine epsilon, a term is
                   the relative uncertainty of a computed step - any two float64 arithmetics (another order of the energy
                   sums, another 2x2 solve) differ by about that much after the step - and the sum what has accumulated by
                   the end if nothing amplifies it (inf / NaN -> inf; inf once eps * cond > 1e-4: a singular solve);
      'last_step'  |step| / max(|a|, 1) of the last iteration;
      'last_cond'  cond(H) at the last iteration;
      'twin'       what rounding may have done to the RESULT, measured: two twin trajectories run beside the pixel's own; after each
                   of their steps they receive a kick of the size of that step's rounding uncertainty (eps * cond(H) * |step|, at
                   least eps, of max(|a|, 1); fixed pseudo-random signs, the second twin the opposite ones).  'twin' is the larger
                   distance of the two from the pixel's own result, relative to max(|a|, 1) (inf when one ends non-finite and the
                   other does not).  A converging iteration forgets the kicks; one that wanders for dozens of steps before it
                   settles (|d map / d a| > 1 step after step: soak seed 1795, pixel 2221 - a 1e-15 kick at step 12 moves the
                   result by 3e-3) does not, and no sum of per-step uncertainties sees that.
    """
    sino_gg = np.asarray(sino_gg, dtype=np.float64)
    mus = np.asarray(mus, dtype=np.float64)
    i0 = np.asarray(i0, dtype=np.float64)
    n_meas, n_views, n_bins = sino_gg.shape
    assert n_meas == 2 and mus.shape[0] == 2, 'two measurements, two basis materials'
    if i0.ndim == 2:
        i0 = np.broadcast_to(i0[:, None, :], (2, n_bins, i0.shape[-1]))
    # product tables, same rounding as the reference's ssff / ssff2 (matdecomp.py:102,105)
    w_g = i0[:, None, :, :] * mus[None, :, None, :]                         # [k, m, bin, e]
    mm = mus[None, :, :] * mus[:, None, :]                                  # [m, n, e]
    w_h = i0[:, None, None, :, :] * mm[None, :, :, None, :]                 # [k, m, n, bin, e]

    a = np.full((n_views, n_bins, 2), EPS_INIT)
    g = np.moveaxis(sino_gg, 1, 0)                                          # [view, k, bin]

    def newton_map(a):
        """one iteration from the states a [view, bin, 2] -> (s0, s1, H, det): the step (matdecomp.py:116-125) and its Hessian"""
        expo = -(a[..., 0, None] * mus[0] + a[..., 1, None] * mus[1])       # [view, bin, e]
        att = np.exp(np.clip(expo, -CLIP, CLIP))
        nu = np.einsum('kbe,vbe->vkb', i0, att)                             # [view, k, bin]
        gr = np.einsum('kmbe,vbe->vkmb', w_g, att)                          # = -nu_grad
        hs = np.einsum('kmnbe,vbe->vkmnb', w_h, att)
        with np.errstate(all='ignore'):
            c = g / nu - 1.0
            q = g / (nu * nu)
            dF = np.einsum('vkb,vkmb->vmb', c, gr)                          # matdecomp.py:122
            H = -np.einsum('vkb,vkmnb->vmnb', c, hs) + np.einsum('vkb,vkmb,vknb->vmnb', q, gr, gr)
            det = H[:, 0, 0] * H[:, 1, 1] - H[:, 0, 1] * H[:, 1, 0]
            s0 = (H[:, 1, 1] * dF[:, 0] - H[:, 0, 1] * dF[:, 1]) / det
            s1 = (H[:, 0, 0] * dF[:, 1] - H[:, 1, 0] * dF[:, 0]) / det
        return s0, s1, H, det

    eps = np.finfo(np.float64).eps
    sens = {'walk': np.zeros((n_views, n_bins)), 'last_step': np.zeros((n_views, n_bins)), 'last_cond': np.ones((n_views, n_bins)),
            'twin': np.zeros((n_views, n_bins))}
    twins = [a.copy(), a.copy()] if return_sensitivity else []
    signs = np.random.default_rng(20261005).choice([-1.0, 1.0], (64, 2))    # the kicks' directions: fixed, the same for every case
    for it in range(n_iters):
        s0, s1, H, det = newton_map(a)
        a[..., 0] -= s0
        a[..., 1] -= s1
        if return_sensitivity:
            with np.errstate(all='ignore'):
                fro2 = H[:, 0, 0] ** 2 + H[:, 1, 1] ** 2 + H[:, 0, 1] ** 2 + H[:, 1, 0] ** 2
                cond = (fro2 + np.sqrt(np.maximum(fro2 * fro2 - 4.0 * det * det, 0.0))) / (2.0 * np.abs(det))     # sigma_max / sigma_min, 2 x 2
                step = np.maximum(np.abs(s0), np.abs(s1)) / np.maximum(np.abs(a).max(-1), 1.0)
                # (a 2 x 2 solve with eps * cond > 1e-4 is numerically singular: what it returns - even an exact 0, as with one energy,
                # where the numerators cancel - is rounding residue, and another arithmetic returns another one)
                now = np.where(cond * eps > 1.0e-4, np.inf, cond * step)
                sens['walk'] = np.where(np.isfinite(now), sens['walk'] + now, np.inf)
                sens['last_step'] = np.where(np.isfinite(step), step, np.inf)
                sens['last_cond'] = np.where(np.isfinite(cond), cond, np.inf)
                # the twins: their own step from their own state, then the kick
                kick = np.where(np.isfinite(now), np.maximum(eps * now, eps), 0.0)
                for t, tw in enumerate(twins):
                    t0, t1, _, _ = newton_map(tw)
                    tw[..., 0] -= t0
                    tw[..., 1] -= t1
                    size = np.maximum(np.abs(tw).max(-1), 1.0)
                    sign = (1.0 if t == 0 else -1.0) * signs[it % len(signs)]
                    tw[..., 0] += sign[0] * kick * size
                    tw[..., 1] += sign[1] * kick * size
    if return_sensitivity:
        with np.errstate(all='ignore'):
            size = np.maximum(np.abs(a).max(-1), 1.0)
            for tw in twins:
                same = np.isfinite(tw).all(-1) == np.isfinite(a).all(-1)
                d = np.abs(tw - a).max(-1) / size
                d = np.where(np.isfinite(a).all(-1), d, 0.0)                # (both non-finite: nothing to compare, the screen drops them anyway)
                sens['twin'] = np.maximum(sens['twin'], np.where(same, np.nan_to_num(d, nan=np.inf), np.inf))
    return (a, sens) if return_sensitivity else a


def decomposition_tables(det_E, det_eta, eid, spec1_E, spec1_I0, spec2_E, spec2_I0):
    """Union energy grid, bin widths and effective spectra (matdecomp.py:140-150)."""
    ee = np.unique(np.concatenate([np.asarray(spec1_E, float), np.asarray(spec2_E, float)]))
    dE = np.concatenate([[ee[0]], np.diff(ee)])           # first bin spans 0..ee[0]
    resp = np.interp(ee, det_E, det_eta)
    if eid:
        resp = resp * ee
    i0 = np.stack([np.interp(ee, spec1_E, spec1_I0) * resp * dE,
                   np.interp(ee, spec2_E, spec2_I0) * resp * dE])
    return ee, dE, i0


def do_matdecomp_gn(ct, sino1, sino2, spec1, spec2, n_iters, basis_mus):
    """basis_mus(ee) -> [2, nE] replaces the two xc.mixatten calls (matdecomp.py:155-160)."""
    ee, _, i0 = decomposition_tables(ct.det_E, ct.det_eta_E, ct.eid, spec1.E, spec1.I0, spec2.E, spec2.I0)
    mus = np.asarray(basis_mus(ee), dtype=np.float64)
    return newton_solve(np.array([sino1, sino2]), i0, mus, n_iters)


def get_basismat_sinos(ct, sino_raw_1, sino_raw_2, spec1, spec2, basis_mus, n_iters=30, mask_thresh=0.95):
    """Air mask from sinogram 1, decomposition, masked pixels set to 0 (matdecomp.py:194-207)."""
    sino_raw_1 = np.asarray(sino_raw_1)
    mask = sino_raw_1 >= mask_thresh * np.max(sino_raw_1)
    a = do_matdecomp_gn(ct, sino_raw_1, sino_raw_2, spec1, spec2, n_iters, basis_mus)
    m1 = a[..., 0]
    m2 = a[..., 1]
    m1[mask] = 0
    m2[mask] = 0
    return m1, m2
