"""Is the exact lane kernel of round 4 bit-identical to round 3's?  Runs the soak's case `seed` through the library of the
round-3 tree (tools/probes/_build/r03, `git archive a34af18` + make) and through the current one (stop_tol = 0, kernel = 1), each
in its own process, and compares the bits.
    python tools/probes/gn_r03_vs_r04.py 468 319 525 0 1 2"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == '--child':
    tree, seed, out = sys.argv[2], int(sys.argv[3]), sys.argv[4]
    sys.path.insert(0, tree)
    import torch
    from dex_ct_sim_amd import matdecomp as md
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
    g_d = torch.tensor(g, dtype=dtype, device='cuda')
    n_iters = int(rng.choice([0, 1, 2, 5, 9, 30, 50, 50, 50, 61, 80]))
    kw = dict(stop_tol=0.0, kernel=1) if 'stop_tol' in md.gn_device.__code__.co_varnames else {}
    a = md.gn_device(g_d[0], g_d[1], i0, mus, n_iters, 'f64', **kw)
    np.save(out, a.cpu().numpy())
    print(f'seed {seed}: {n_e} energies, {n_v} x {n_c}, {kind}, {n_iters} iterations', flush=True)
    sys.exit(0)

for seed in [int(x) for x in sys.argv[1:]]:
    outs = []
    for name, tree in (('r03', os.path.join(ROOT, 'tools', 'probes', '_build', 'r03')), ('r04', ROOT)):
        out = f'/tmp/gn_{name}_{seed}.npy'
        subprocess.run([sys.executable, os.path.abspath(__file__), '--child', tree, str(seed), out], check=True,
                       stdout=subprocess.PIPE if name == 'r03' else None)
        outs.append(np.load(out))
    a, b = outs
    same = a.view(np.int64) == b.view(np.int64)
    d = np.abs(a - b) / np.maximum(np.abs(a), 1.0)
    fin = np.isfinite(a).all(-1) & np.isfinite(b).all(-1)
    print(f'  bit-identical: {bool(same.all())}; differing values {int((~same).sum())} of {same.size}; max rel diff on finite pixels '
          f'{float(np.max(d[fin], initial=0.0)):.2e}; finite in one only: {int((np.isfinite(a).all(-1) != np.isfinite(b).all(-1)).sum())}', flush=True)
