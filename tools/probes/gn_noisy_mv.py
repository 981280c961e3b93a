"""The MV pair at the lowest dose: where do the few differing pixels come from - the short cut or the tolerance rule of the walk?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import small_scan, INPUT
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import matdecomp as md
ct, ph = small_scan(n=512, nz=1, n_views=1200, n_channels=800, n_rows=1)
for pair in (('6MV_1mGy_float32.bin', '80kV_1mGy_float32.bin'), ('detunedMV_1mGy_float32.bin', '80kV_1mGy_float32.bin')):
    specs = []
    for name in pair:
        s = dx.xRaySpectrum(os.path.join(INPUT, 'spectrum', name), name[:5]); s.rescale_counts(ct.A_iso * 0.02 / ct.N_proj); specs.append(s)
    for noise in (True, 'poisson'):
        (r1, _), (r2, _) = dx.get_sinos(ct, ph, specs, noise=noise, seed=3)
        x = np.stack(md.get_basismat_sinos(ct, r1, r2, specs[0], specs[1], n_iters=50, stop_tol=0.0), -1)
        out = {}
        for mode in (None, False):
            m = np.stack(md.get_basismat_sinos(ct, r1, r2, specs[0], specs[1], n_iters=50, two_level=mode), -1)
            fin = np.isfinite(x).all(-1)
            with np.errstate(invalid='ignore'):
                e = (np.abs(m - x) / np.maximum(np.abs(x), 1.0)).max(-1)
            sel = fin & np.isfinite(m).all(-1) & (np.abs(x).max(-1) < 1e6)
            out[mode] = (int((np.isfinite(m).all(-1) != fin).sum()), int((e[sel] > 1e-12).sum()), float(e[sel].max()), m)
        same = np.array_equal(np.nan_to_num(out[None][3], nan=-7.0), np.nan_to_num(out[False][3], nan=-7.0))
        d = np.nan_to_num(out[None][3], nan=-7.0) != np.nan_to_num(out[False][3], nan=-7.0)
        print(pair[0][:9], noise, 'default (pattern, beyond, worst):', out[None][:3], '| every pixel from 1e-6:', out[False][:3], '| identical arrays:', same,
              '| pixels where they differ at all:', int(d.any(-1).sum()), flush=True)
