"""Which pixel of a noisy 140 kV / 80 kV scan does the chord step of the short cut leave furthest from the exact count, and who is
right: longdouble Newton on the pixel's counts as the judge."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('DEXCT_CACHE_DIR', 'off')
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import matdecomp as md, quadrature as q
from conftest import INPUT, small_scan

ct, ph = small_scan(n=512, nz=1, n_views=1200, n_channels=800, n_rows=1)
dose = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
specs = []
for name in ('140kV', '80kV'):
    s = dx.xRaySpectrum(os.path.join(INPUT, 'spectrum', f'{name}_1mGy_float32.bin'), name)
    s.rescale_counts(ct.A_iso * dose / ct.N_proj)
    specs.append(s)
(r1, _), (r2, _) = dx.get_sinos(ct, ph, specs, noise=True, seed=3)
x = np.stack(md.get_basismat_sinos(ct, r1, r2, specs[0], specs[1], n_iters=50, stop_tol=0.0), -1)
m = np.stack(md.get_basismat_sinos(ct, r1, r2, specs[0], specs[1], n_iters=50), -1)
st = md.last_gn_stats()
w = np.stack(md.get_basismat_sinos(ct, r1, r2, specs[0], specs[1], n_iters=50, two_level='start'), -1)
fin = np.isfinite(x).all(-1)
rel = np.abs(m - x).max(-1) / np.maximum(np.abs(x).max(-1), 1.0)
rel2 = np.abs(w - x).max(-1) / np.maximum(np.abs(x).max(-1), 1.0)
rel[~fin] = 0
rel2[~fin] = 0
print('mode', st['mode'], 'steps per pixel', st['pixel_iterations'] / fin.sum(), 'worst one-step', rel.max(), 'worst two-step', rel2.max(),
      'pixels > 2.5e-13:', int((rel > 2.5e-13).sum()), 'of', rel.size)
_, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
i0l, musl = i0.astype(np.longdouble), mus.astype(np.longdouble)
gate = None
for ent in md._table_cache.values():
    for k, v in ent.items():
        if isinstance(k, tuple) and k[0] == 'gate':
            gate = v
start = gate['start'].cpu().numpy()
n = int(start[3])
_, c0, k0, b0 = q.start_layout(n)
cells = start[c0:k0].reshape(n, n, 2)
one = start[k0:b0].reshape(n, n, 2)
for idx in np.argsort(rel.ravel())[::-1][:6]:
    v, c = np.unravel_index(idx, rel.shape)
    g = np.array([r1[v, c], r2[v, c]], dtype=np.longdouble)
    a = x[v, c].astype(np.longdouble)
    for _ in range(6):                                    # Newton on ln nu = ln g in extended precision
        att = np.exp(-(a @ musl))
        nu = i0l @ att
        G = (i0l * att) @ musl.T                          # [k, p]
        L = -G / nu[:, None]
        a = a + np.linalg.solve(L.astype(np.float64), np.asarray(g / nu - 1, dtype=np.float64)).astype(np.longdouble)
    size = max(np.abs(x[v, c]).max(), 1.0)
    u0 = np.log(start[0] / float(g[0])) * start[2]
    u1 = np.log(start[1] / float(g[1])) * start[2]
    fx, fy = (np.log(u0) - start[4]) * start[5], (u1 / u0 - start[6]) * start[7]
    ci, cj = int(fx), int(fy)
    print(f'pixel ({v}, {c}) g = {float(g[0]):.6g}, {float(g[1]):.6g}  cell ({ci}, {cj}) need {cells[ci, cj, 0]} kappa {one[ci, cj, 0]:.3g} eps {one[ci, cj, 1]:.3g}'
          f'\n    exact count {x[v, c]}  one step off by {rel[v, c]:.3e}, two steps off by {rel2[v, c]:.3e}'
          f'\n    longdouble root: exact count off by {float(np.abs(x[v, c] - a).max() / size):.3e}, one step off by {float(np.abs(m[v, c] - a).max() / size):.3e},'
          f' two steps off by {float(np.abs(w[v, c] - a).max() / size):.3e}', flush=True)
