import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from dex_ct_sim_amd import matdecomp as md, quadrature as q
golden = np.load(os.path.join(ROOT, 'tests', 'golden', 'gn_reference.npz'))
import importlib.util
spec = importlib.util.spec_from_file_location('tg', os.path.join(ROOT, 'tests', 'test_gpu_gn.py'))
rng = np.random.default_rng(77); n = 60000
i0, mus = golden['gn0_i0'], golden['gn0_mus']
a_true = np.stack([rng.uniform(0, 40, n) * rng.choice([0.02, 0.3, 1.0], n), rng.uniform(0, 8, n) * rng.choice([0.0, 0.1, 1.0], n)], -1)
a_true[: n // 3, 1] = -0.008 * a_true[: n // 3, 0]
ex = np.exp(-a_true @ mus)
cnt = (np.stack([(i0[k] * ex).sum(-1) for k in range(2)]) * (1 + 0.002 * rng.standard_normal((2, n)))).reshape(2, 100, n // 100)
exact = md.optimize_sino(cnt, None, i0, mus, 50, precision='f64', verbose=False, stop_tol=0.0).reshape(-1, 2)
for rep in range(2):
    a = md.optimize_sino(cnt, None, i0, mus, 50, precision='f64', verbose=False, two_level='start').reshape(-1, 2)
    bad = np.flatnonzero(~np.isfinite(a).all(-1) & np.isfinite(exact).all(-1))
    print('rep', rep, 'bad pixels', bad[:10], len(bad))
    for b in bad[:5]:
        print('  pixel', b, 'counts', cnt.reshape(2, -1)[:, b], 'true', a_true[b], 'exact', exact[b], 'two-level', a[b])
one = md.optimize_sino(cnt, None, i0, mus, 50, precision='f64', verbose=False, two_level=False).reshape(-1, 2)
for b in bad[:5]:
    print('  single launch', one[b])
    # trajectory of the exact count for this pixel
    for k in (1, 2, 3, 4, 5, 8, 12, 20):
        t = md.optimize_sino(cnt.reshape(2, -1)[:, b:b + 1].reshape(2, 1, 1), None, i0, mus, k, precision='f64', verbose=False, stop_tol=0.0)
        print('     after', k, t.ravel())
import torch
from dex_ct_sim_amd._device import to_dev
g = to_dev(cnt.reshape(2, -1), torch.float64, torch.device('cuda'))
for kern in (0, 1, 2):
    for tol in (0.0, None):
        r = md.gn_device(g[0], g[1], i0, mus, 50, 'f64', stop_tol=tol, kernel=kern, two_level=False)
        print('kernel', kern, 'stop_tol', tol, 'pixel 3937 ->', r[3937].cpu().numpy(), 'NaN pixels', int(torch.isnan(r).any(-1).sum()))
# the same pixel alone, and with a few neighbours
for lo, hi in ((3937, 3938), (3900, 3964), (3904, 3968)):
    r = md.gn_device(g[0, lo:hi].contiguous(), g[1, lo:hi].contiguous(), i0, mus, 50, 'f64', kernel=1, two_level=False)
    print('lane kernel on pixels', lo, hi, '->', r[3937 - lo].cpu().numpy())
    r = md.gn_device(g[0, lo:hi].contiguous(), g[1, lo:hi].contiguous(), i0, mus, 50, 'f64', two_level='start')
    print('start mode on pixels', lo, hi, '->', r[3937 - lo].cpu().numpy(), md.last_gn_stats())
