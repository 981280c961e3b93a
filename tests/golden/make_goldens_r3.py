#!/usr/bin/env python3
"""Third set of golden vectors from the REAL reference (round 3): POISSON-NOISY measurements.

All earlier goldens are noise-free, and noise is what moves the conditioning of the 2x2 Hessians.  Runs only in the
build container (needs /root/reference); writes ``ref_noisy.npz`` next to this file (arrays only).

  ``noisy_*``  140 kV / 80 kV at doses 5 / 5 mGy with the dose scaling of main.py:68,
      ``spec.rescale_counts(ct.A_iso * dose / ct.N_proj)``, for the default scanner of input/params.txt:18-27
      (A_iso = SID * fan_angle_total / N_channels * detector_px_height, N_proj = 1200): ~7e7 photons per unattenuated
      ray.  6 views x 32 channels of tissue / bone thicknesses (two air channels per view); the measurement of every
      pixel is a per-energy-bin Poisson draw weighted by the energy-integrating detector's response - the compound
      statistics of the forward model inside the reference's decomposition (matdecomp.py:146-150).  Each pixel is handed
      to ``optimize_sino_cpu`` (matdecomp.py:87-127) for 1, 2 and 50 iterations, and to the frozen screen of
      make_goldens_r2.py (CRITERION_VERSION 2; nothing new is screened here).  ``get_basismat_sinos`` (:167-207) is run
      on the whole noisy pair as well (mask from the noisy maximum).

    python tests/golden/make_goldens_r3.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_goldens import REF, half_split, load_pkg_module, load_reference  # noqa: E402
from make_goldens_r2 import CRITERION_VERSION, screen  # noqa: E402


def main():
    ref = load_reference()
    xc = load_pkg_module('xcompy')
    ref.xc.mixatten = xc.mixatten
    captured = {}
    real_opt = ref.optimize_sino_cpu

    def spy(Sino_gg, ee, i0, mus, n_iters, verbose=True):
        captured.update(ee=np.array(ee), i0=np.array(i0), mus=np.array(mus))
        return real_opt(Sino_gg, ee, i0, mus, n_iters, verbose=False)

    # scanner numbers of the reference's input/params.txt:18-27
    SID, fan, n_ch, n_proj, h_iso = 60.0, 0.8230337, 800, 1200, 1.0
    A_iso = SID * (fan / n_ch) * h_iso
    det = half_split(f'{REF}/input/detector/eta_eid_mv.bin')
    s1, s2 = half_split(f'{REF}/input/spectrum/140kV_1mGy_float32.bin'), half_split(f'{REF}/input/spectrum/80kV_1mGy_float32.bin')
    ct = types.SimpleNamespace(det_E=det[0], det_eta_E=det[1], eid=True)
    sp1 = types.SimpleNamespace(E=s1[0], I0=s1[1] * (A_iso * 5.0 / n_proj))
    sp2 = types.SimpleNamespace(E=s2[0], I0=s2[1] * (A_iso * 5.0 / n_proj))
    ref.optimize_sino_cpu = spy
    ref.do_matdecomp_gn(ct, np.ones((1, 2)), np.ones((1, 2)), sp1, sp2, 1)
    ref.optimize_sino_cpu = real_opt
    ee, i0, mus = captured['ee'], captured['i0'][:, 0, :], captured['mus']

    rng = np.random.default_rng(20261005)
    nV, nB, n_iters = 6, 32, 50
    a_true = np.stack([rng.uniform(0.0, 35.0, (nV, nB)), rng.uniform(0.0, 6.0, (nV, nB))], axis=-1)
    a_true[:, :2] = 0.0                                    # two air channels per view
    # per-bin Poisson photons, detected with the response the decomposition assumes: i0 = photons * eta * E * dE, so the
    # signal of N photons in bin e is N * (i0[e] / photons[e]) with photons = I0_interp * dE (matdecomp.py:142,149-150)
    dE = np.append([ee[0]], ee[1:] - ee[:-1])
    photons = np.stack([np.interp(ee, sp1.E, sp1.I0) * dE, np.interp(ee, sp2.E, sp2.I0) * dE])       # [2, nE]
    gain = np.divide(i0, photons, out=np.zeros_like(i0), where=photons > 0)                            # eta * E
    att = np.exp(-np.tensordot(a_true, mus, axes=([-1], [0])))                                         # [nV, nB, nE]
    lam = photons[:, None, None, :] * att[None]                                                        # [2, nV, nB, nE]
    g = (rng.poisson(lam) * gain[:, None, None, :]).sum(-1)                                            # [2, nV, nB]
    g_clean = (lam * gain[:, None, None, :]).sum(-1)
    out = dict(noisy_ee=ee, noisy_i0=i0, noisy_mus=mus, noisy_a_true=a_true, noisy_g=g, noisy_g_clean=g_clean,
               noisy_n_iters=np.array(n_iters), noisy_criterion_version=np.array(CRITERION_VERSION),
               noisy_spec1_E=sp1.E, noisy_spec1_I0=sp1.I0, noisy_spec2_E=sp2.E, noisy_spec2_I0=sp2.I0,
               noisy_det_E=ct.det_E, noisy_det_eta=ct.det_eta_E)
    raised, raised_at, a50, ill, cond = screen(real_opt, g, ee, i0, mus, n_iters)
    out.update(noisy_raised=raised, noisy_raised_at=raised_at, noisy_a50=a50, noisy_ill=ill, noisy_cond=cond)
    # trajectories after 1 and 2 iterations pin the update rule on noisy data, not only the fixed point
    i0_1 = i0[:, None, :]
    for it in (1, 2):
        a_it = np.full((nV, nB, 2), np.nan)
        with np.errstate(all='ignore'):
            for j in range(nV):
                for b in range(nB):
                    try:
                        a_it[j, b] = real_opt(g[:, j:j + 1, b:b + 1], ee, i0_1, mus, it, verbose=False)[0, 0]
                    except np.linalg.LinAlgError:
                        pass
        out[f'noisy_a{it}'] = a_it
    # the public call on the whole noisy pair (mask threshold from the noisy maximum, :195-196)
    if not raised.any():
        with np.errstate(all='ignore'):
            m1, m2 = ref.get_basismat_sinos(ct, g[0].copy(), g[1].copy(), sp1, sp2, n_iters=n_iters)
        out.update(noisy_mat1=m1, noisy_mat2=m2)
    ok = ~raised & ~ill
    rel_noise = np.abs(g - g_clean) / g_clean
    err = np.abs(a50[ok] - a_true[ok])
    print(f'noisy 140kV/80kV at 5/5 mGy: {raised.sum()} of {raised.size} pixels raise, {ill.sum()} more flagged by the frozen '
          f'screen (v{CRITERION_VERSION}); relative noise of the counts median {np.median(rel_noise):.2e} max {rel_noise.max():.2e}; '
          f'decomposed thickness error median {np.median(err):.3f} max {err.max():.3f} g/cm2; worst Hessian condition '
          f'{np.nanmax(cond):.2e}')
    path = os.path.join(HERE, 'ref_noisy.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes,', len(out), 'arrays')


if __name__ == '__main__':
    main()
