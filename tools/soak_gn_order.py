"""Randomised check of the Newton kernels' result ORDER (round 4): random (views, channels, rows), float32 / float64 input,
with / without the air mask, lane and cooperative kernel, default and exact stop: the results written as [view][row][channel]
from [view][channel][row] input must be the bits of the plain-order launch, permuted; every pixel written exactly once.
    python tools/soak_gn_order.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dex_ct_sim_amd import matdecomp as md

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'gn_reference.npz'))
i0, mus = g['gn0_i0'], g['gn0_mus']
fails, n_pix = 0, 0
for case in range(n_cases):
    rng = np.random.default_rng(880000 + seed0 + case)
    V, C, R = int(rng.integers(1, 7)), int(rng.integers(1, 90)), int(rng.integers(1, 90))
    n = V * C * R
    a_true = np.stack([rng.uniform(0, 35, n), rng.uniform(0, 6, n)], -1)
    cnt = np.stack([(i0[k] * np.exp(-a_true @ mus)).sum(-1) for k in range(2)]) * (1 + 0.002 * rng.standard_normal((2, n)))
    cnt[0, rng.random(n) < rng.uniform(0, 0.6)] = 2.0 * i0[0].sum()
    dt = torch.float32 if rng.random() < 0.5 else torch.float64
    g1, g2 = (torch.tensor(cnt[k], device='cuda').to(dt).reshape(V, C, R) for k in range(2))
    gmax = g1.max().double() if rng.random() < 0.7 else None
    kern = int(rng.choice([1, 2]))
    tol = None if rng.random() < 0.5 else 0.0
    n_iters = int(rng.choice([0, 1, 7, 30, 50]))
    plain = md.gn_device(g1, g2, i0, mus, n_iters, 'f64', mask_max=gmax, stop_tol=tol, kernel=kern)
    out = torch.full((V, R, C, 2), float('nan'), dtype=torch.float64, device='cuda')
    md.gn_device(g1, g2, i0, mus, n_iters, 'f64', mask_max=gmax, stop_tol=tol, kernel=kern, out_rc=(R, C), out=out)
    ok = torch.equal(out.view(torch.int64), plain.permute(0, 2, 1, 3).contiguous().view(torch.int64))
    n_pix += n
    if not ok:
        fails += 1
        print(f'FAIL case {case}: V {V} C {C} R {R} {dt} mask {gmax is not None} kernel {kern} tol {tol} iters {n_iters}: '
              f'{int((out.view(torch.int64) != plain.permute(0, 2, 1, 3).contiguous().view(torch.int64)).sum())} values differ', flush=True)
print(f'{n_cases} cases, {fails} failed, {n_pix:.3g} pixels', flush=True)
sys.exit(1 if fails else 0)
