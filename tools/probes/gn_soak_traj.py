"""One flagged case of tools/soak_gn.py followed step by step: for the pixels on which the lane kernel, the cooperative
kernel, the default tolerance stop and the NumPy restatement disagree, the state after every iteration in all three
arithmetics (exact mode, n_iters = 1 .. the case's count), the condition of the Hessian along the way, and the iteration at
which the trajectories part.    gpurun -- python tools/probes/gn_soak_traj.py 319 525 1179 468"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from soak_cases import draw, hessian_cond
from dex_ct_sim_amd import matdecomp as md
from oracle import gn_oracle

os.environ['DEXCT_CACHE_DIR'] = 'off'
dev = torch.device('cuda:0')
np.set_printoptions(precision=17, linewidth=250)


def run(g_d, i0, mus, n, **kw):
    return md.gn_device(g_d[0], g_d[1], i0, mus, n, 'f64', **kw).cpu().numpy().reshape(-1, 2)


for seed in [int(x) for x in sys.argv[1:]]:
    c = draw(seed)
    i0, mus, g, n_iters, n_e = c['i0'], c['mus'], c['g'], c['n_iters'], c['n_e']
    dtype = torch.float32 if c['kind'] == 'float32' else torch.float64
    g_d = torch.tensor(g, dtype=dtype, device=dev)
    gg = g.reshape(2, -1)
    print(f'==== seed {seed}: {n_e} energies, {c["n_v"]} x {c["n_c"]} pixels, {c["kind"]}, {n_iters} iterations; mus {mus.min():.3g} .. {mus.max():.3g}')
    print('i0', i0.tolist())
    print('mus', mus.tolist())
    lane = run(g_d, i0, mus, n_iters, stop_tol=0.0, kernel=1, two_level=False)
    coop = run(g_d, i0, mus, n_iters, stop_tol=0.0, kernel=2, two_level=False)
    dflt = run(g_d, i0, mus, n_iters, kernel=1, two_level=False)
    with np.errstate(all='ignore'):
        ref = gn_oracle.newton_solve(g, i0, mus, n_iters).reshape(-1, 2)
        ref_p = gn_oracle.newton_solve(g * (1 + 1e-13), i0, mus, n_iters).reshape(-1, 2)
        perm = c['rng'].permutation(n_e)
        ref_q = gn_oracle.newton_solve(g, i0[:, perm], mus[:, perm], n_iters).reshape(-1, 2)
        print('the screen\'s permutation of the energies:', perm.tolist())
        size = np.maximum(np.abs(ref).max(-1), 1.0)
        ok = np.isfinite(ref).all(-1) & (np.abs(ref).max(-1) < 1e6) & np.isfinite(gg).all(0) & (gg > 0).all(0)
        ok &= (np.abs(ref - ref_p).max(-1) <= 1e-11 * size) & (np.abs(ref - ref_q).max(-1) <= 1e-11 * size)
        d_coop = np.abs(coop - lane).max(-1) / size
        d_dflt = np.abs(dflt - lane).max(-1) / size
        d_ref = np.abs(lane - ref).max(-1) / size
    flagged = np.flatnonzero(ok & ((d_coop > 1e-9) | (d_dflt > 1e-10) | (d_ref > 1e-9)))
    print(f'{int(ok.sum())} stable pixels; beyond: coop vs lane {int((ok & (d_coop > 1e-9)).sum())}, default vs lane {int((ok & (d_dflt > 1e-10)).sum())}, '
          f'lane vs NumPy {int((ok & (d_ref > 1e-9)).sum())}; looking at {flagged[:6].tolist()}')
    for p in flagged[:6]:
        gp = np.ascontiguousarray(gg[:, p:p + 1])
        gp_d = torch.tensor(gp, dtype=dtype, device=dev)
        print(f'-- pixel {p}: counts {gp[:, 0].tolist()} true {c["a_true"].reshape(-1, 2)[p].tolist()}')
        print(f'   after {n_iters}: lane {lane[p]} coop {coop[p]} default {dflt[p]} NumPy {ref[p]} NumPy(1+1e-13) {ref_p[p]} NumPy(permuted) {ref_q[p]}')
        parted = None
        prev_l = np.array([1e-6, 1e-6])
        for k in range(1, n_iters + 1):
            l = run(gp_d, i0, mus, k, stop_tol=0.0, kernel=1, two_level=False, natural_order=True)[0]
            co = run(gp_d, i0, mus, k, stop_tol=0.0, kernel=2, two_level=False, natural_order=True)[0]
            with np.errstate(all='ignore'):
                r = gn_oracle.newton_solve(gp.reshape(2, 1, 1), i0, mus, k).reshape(2)
                cond = hessian_cond(prev_l[None, :], gp, i0, mus)[0]
                sz = max(np.abs(l).max(), 1.0)
                dl, dr = np.abs(co - l).max() / sz, np.abs(r - l).max() / sz
            if parted is None and (dl > 1e-12 or dr > 1e-12):
                parted = k
            if k <= 3 or k == n_iters or (parted is not None and k <= parted + 6) or k % 10 == 0:
                print(f'   k={k:3d} lane {l}  step {np.abs(l - prev_l).max():.3e}  cond(H before the step) {cond:.3e}  coop-lane {dl:.2e}  NumPy-lane {dr:.2e}')
            prev_l = l
        print(f'   trajectories part (1e-12) at iteration {parted}')
