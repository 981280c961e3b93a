"""One case of tools/soak_gn.py (same draws) looked at closely: where do the two-level modes differ from the exact lane kernel?
gpurun -- python tools/probes/gn_soak_case.py <seed>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dex_ct_sim_amd import matdecomp as md

seed = int(sys.argv[1])
dev = torch.device('cuda:0')
rng = np.random.default_rng(770000 + seed)
n_e = int(rng.choice([1, 2, 3, 7, 33, 64, 140, 140, 239, 300]))
E = np.linspace(15.0, 150.0, n_e) if n_e > 1 else np.array([60.0])
pa, pb = rng.uniform(0.1, 0.4, 2), rng.uniform(0.1, 0.2, 2)
pp = np.array([rng.uniform(0.2, 1.0), rng.uniform(2.0, 3.2)])
mus = pa[:, None] * (E[None, :] / 60.0) ** (-pp[:, None]) + pb[:, None]
if rng.random() < 0.3 and n_e > 4:
    mus[:, : n_e // 8 + 1] *= 30.0
i0 = rng.uniform(0.2, 1.0, (2, n_e)) * 10.0 ** rng.uniform(0, 7)
if n_e > 6:
    lo, hi = sorted(rng.integers(0, n_e, 2))
    i0[0, lo:hi // 2] = 0.0
    i0[1, hi:] = 0.0
    i0[:, n_e // 2] = 0.0
    i0[:, -1] = np.maximum(i0[:, -1], 1.0)
    i0[:, 0] = np.maximum(i0[:, 0], 1.0)
n_v, n_c = int(rng.integers(1, 40)), int(rng.integers(1, 700))
a_true = np.stack([rng.uniform(0, 45, (n_v, n_c)), np.where(rng.random((n_v, n_c)) < 0.5, 0.0, rng.uniform(0, 8, (n_v, n_c)))], -1)
att = np.exp(-(a_true[..., :1] * mus[0] + a_true[..., 1:] * mus[1]))
g = np.einsum('ke,vce->kvc', i0, att)
kind = rng.choice(['clean', 'noisy', 'poisson', 'float32'])
if kind == 'noisy':
    g = g * (1 + 10.0 ** rng.uniform(-6, -1) * rng.standard_normal(g.shape))
elif kind == 'poisson':
    g = rng.poisson(np.minimum(g, 1e15)).astype(np.float64)
weird = rng.random(g.shape) < 0.01
g[weird] = rng.choice([0.0, -1.0, np.inf, np.nan, 1e-300, 1e300], int(weird.sum()))
dtype = torch.float32 if kind == 'float32' else torch.float64
g_d = torch.tensor(g, dtype=dtype, device=dev)
n_iters = int(rng.choice([0, 1, 2, 5, 9, 30, 50, 50, 50, 61, 80]))
print('seed', seed, n_e, 'energies', n_v, 'x', n_c, kind, n_iters, 'iterations; mus range', mus.min(), mus.max(), 'scaled low energies', bool(mus[0, 0] > 5))
run = lambda n, **kw: md.gn_device(g_d[0], g_d[1], i0, mus, n, 'f64', kernel=1, **kw).cpu().numpy().reshape(-1, 2)
exact = run(n_iters, stop_tol=0.0, two_level=False)
long = run(250, stop_tol=0.0, two_level=False)
single = run(n_iters, two_level=False)
gg = g_d.double().cpu().numpy().reshape(2, -1)
for mode in ('start', 'coarse'):
    a = run(n_iters, two_level=mode)
    st = md.last_gn_stats()
    size = np.maximum(np.abs(exact).max(-1), 1.0)
    d = np.abs(a - exact).max(-1) / size
    bad = np.flatnonzero((d > 1e-10) & np.isfinite(exact).all(-1))
    print(mode, st, 'pixels beyond 1e-10 of the exact count:', len(bad))
    for b in bad[:6]:
        print('   pixel', b, 'g', gg[:, b], 'true', a_true.reshape(-1, 2)[b], '| exact', exact[b], '| after 250', long[b], '| single', single[b], '| two-level', a[b])

# the gate for the pixels listed last: cell, need, radius, interpolated start value (host emulation of csrc/gn.hip gn_start)
from dex_ct_sim_amd import quadrature as q
ent = [v for v in md._table_cache.values()][0]
tabs = [v for k, v in ent.items() if isinstance(k, tuple) and k[0] == 'coarse'][0]
st = tabs[2].cpu().numpy()
n = int(st[3])
roots = st[q.START_HEADER:q.START_HEADER + 2 * (n + 1) ** 2].reshape(n + 1, n + 1, 2)
cells = st[q.START_HEADER + 2 * (n + 1) ** 2:].reshape(n, n, 2)
print('open cells', np.isfinite(cells[:, :, 0]).mean())
for b in bad[:6]:
    u0, u1 = np.log(st[0] / gg[0, b]) * st[2], np.log(st[1] / gg[1, b]) * st[2]
    fx, fy = (np.log(u0) - st[4]) * st[5], (u1 / u0 - st[6]) * st[7]
    i, j = int(fx), int(fy)
    inside = 0 <= fx < n and 0 <= fy < n
    print('   pixel', b, 'u', u0, u1, 'cell', (fx, fy), 'need/radius', cells[i, j] if inside else None, 'corner fixed points', roots[i:i + 2, j:j + 2].reshape(-1, 2).tolist() if inside else None)

# are both answers roots of the two equations?  residual of the counts and condition of the log-Jacobian at each
pc = q.newton_start_grid(i0, mus)
for b in bad[:4]:
    for name, pt in (('exact', exact[b]), ('two-level', a[b]), ('true', a_true.reshape(-1, 2)[b])):
        res, cond = q._counts_and_condition(pc, np.asarray(pt)[None, :], gg[:, b][None, :])
        print('   pixel', b, name, pt, 'residual of the counts', res[0], 'cond', cond[0])
