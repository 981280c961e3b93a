import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dex_ct_sim_amd import matdecomp as md
from oracle import c_oracle as co
e = np.load(os.path.join(ROOT, 'tests/golden/ref_extra.npz'))
g, i0, mus = e['uns_g'], e['uns_i0'], e['uns_mus']
ok = ~e['uns_raised'] & ~e['uns_ill']
ref = e['uns_a50']
def show(tag, a):
    bad = ok & ~np.isfinite(a).all(-1)
    d = np.abs(a - ref) / np.maximum(np.abs(ref), 1.0)
    d = np.where(np.isfinite(d), d, np.inf).max(-1)
    print(tag, 'non-finite among ok:', int(bad.sum()), 'max err over finite ok:', np.nanmax(np.where(ok & np.isfinite(a).all(-1), d, 0)),
          'idx', np.argwhere(bad).tolist()[:6])
    return bad
a = md.optimize_sino(g, None, i0, mus, 50, precision='f64')
bad = show('refill default', a)
os.environ['DEXCT_GN_FULL_LOOP'] = '1'
show('refill full loop', md.optimize_sino(g, None, i0, mus, 50, precision='f64'))
os.environ.pop('DEXCT_GN_FULL_LOOP')
i0t = np.repeat(i0[:, None, :], g.shape[2], axis=1).copy(); i0t[0, 0, 0] *= 1.0000001     # per-bin kernel (gn_kernel<false,true>)
show('per-bin kernel', md.optimize_sino(g, None, i0t, mus, 50, precision='f64'))
with np.errstate(all='ignore'):
    c = co.gn_decompose(g[0].ravel(), g[1].ravel(), i0, mus, 50).reshape(ref.shape)
show('C oracle', c)
for (j, b) in np.argwhere(bad)[:3]:
    for it in (1, 2, 3, 4, 5, 6, 8, 10, 15, 20, 30, 50):
        ak = md.optimize_sino(g[:, j:j + 1, b:b + 1], None, i0, mus, it, precision='f64')[0, 0]
        with np.errstate(all='ignore'):
            ck = co.gn_decompose(g[0, j:j + 1, b], g[1, j:j + 1, b], i0, mus, it)[0]
        print((j, b), it, ak, ck)
