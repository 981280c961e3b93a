import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dex_ct_sim_amd import matdecomp as md
golden = np.load(os.path.join(ROOT, 'tests', 'golden', 'gn_reference.npz'))
rng = np.random.default_rng(5); n = 30000
i0, mus = golden['gn0_i0'], golden['gn0_mus']
a_true = np.stack([rng.uniform(0, 40, n) * rng.choice([0.02, 0.3, 1.0], n), rng.uniform(0, 8, n) * rng.choice([0.0, 0.1, 1.0], n)], -1)
a_true[: n // 3, 1] = -0.008 * a_true[: n // 3, 0]
ex = np.exp(-a_true @ mus)
cnt = np.stack([(i0[k] * ex).sum(-1) for k in range(2)]).reshape(2, 100, n // 100)
err = lambda a, b: np.abs(a - b) / np.maximum(np.abs(b), 1.0)
for n_iters in (4, 5, 8, 12, 20, 50):
    exact = md.optimize_sino(cnt, None, i0, mus, n_iters, precision='f64', verbose=False, stop_tol=0.0).reshape(-1, 2)
    single = md.optimize_sino(cnt, None, i0, mus, n_iters, precision='f64', verbose=False, two_level=False).reshape(-1, 2)
    for mode in ('start', 'coarse'):
        a = md.optimize_sino(cnt, None, i0, mus, n_iters, precision='f64', verbose=False, two_level=mode).reshape(-1, 2)
        e = err(a, exact).max(-1); e1 = err(single, exact).max(-1)
        w = int(np.argmax(e))
        print(n_iters, mode, 'two-level vs exact', e.max(), 'single vs exact', e1.max(), 'equal to single bits', np.array_equal(a.view(np.int64), single.view(np.int64)),
              '| worst pixel', w, 'true', a_true[w], 'exact', exact[w], 'two-level', a[w], 'single', single[w], md.last_gn_stats()['pixel_iterations'])
