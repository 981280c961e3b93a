"""The public calls on noisy scans with the reference's bundled spectra, dose-scaled as main.py:68 does: default mode against
the exact count on every pixel.  gpurun -- python tools/probes/gn_noisy_public.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import small_scan, INPUT
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import matdecomp as md

PAIR = tuple(sys.argv[1:3]) if len(sys.argv) > 2 else ('140kV_1mGy_float32.bin', '80kV_1mGy_float32.bin')
for n, nz, views, chans, rows in ((512, 1, 1200, 800, 1), (256, 64, 360, 512, 64)):
    ct, ph = small_scan(n=n, nz=nz, n_views=views, n_channels=chans, n_rows=rows)
    for dose in (5.0, 0.5, 0.02):                      # mGy-like scale factors of main.py:68 (A_iso * dose / N_proj)
        specs = []
        for name in PAIR:
            s = dx.xRaySpectrum(os.path.join(INPUT, 'spectrum', name), name[:5])
            s.rescale_counts(ct.A_iso * dose / ct.N_proj)
            specs.append(s)
        for noise in (True, 'poisson'):
            (r1, _), (r2, _) = dx.get_sinos(ct, ph, specs, noise=noise, seed=3)
            x1, x2 = md.get_basismat_sinos(ct, r1, r2, specs[0], specs[1], n_iters=50, stop_tol=0.0)
            n_exact = md.last_gn_stats()['pixel_iterations']
            m1, m2 = md.get_basismat_sinos(ct, r1, r2, specs[0], specs[1], n_iters=50)
            st = md.last_gn_stats()
            x, m = np.stack([x1, x2], -1), np.stack([m1, m2], -1)
            fin = np.isfinite(x).all(-1)
            pat = int((np.isfinite(m).all(-1) != fin).sum())
            e = (np.abs(m - x) / np.maximum(np.abs(x), 1.0)).max(-1)
            sel = fin & (np.abs(x).max(-1) < 1e6)
            print(f'{views} x {chans} x {rows}, dose {dose}, noise {noise}: open-beam counts {float(r1.max()):.3g}, min {float(r1.min()):.3g}; exact finite on '
                  f'{fin.mean():.4f}; pattern differs {pat}; beyond 1e-12: {int((e[sel] > 1e-12).sum())}, worst {float(e[sel].max()):.2e}; '
                  f'mode {st["mode"]}, full steps {st["pixel_iterations"] / n_exact:.3f} of the exact count\'s', flush=True)
