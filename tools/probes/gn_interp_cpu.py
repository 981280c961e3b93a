"""CPU study: how far is the interpolant of the tabulated fixed points from a pixel's own fixed point, for Catmull-Rom (what the
kernel uses), 4-point and 6-point Lagrange interpolation, on grids of 128 and 256 cells - at random points of the cells, not only
their centres.    python tools/probes/gn_interp_cpu.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import matdecomp as md, quadrature as q, synthetic
from oracle import gn_oracle

det = os.path.join(ROOT, 'dex-ct-sim_amd', 'input', 'detector', 'eta_eid_mv.bin')
ct = dx.FanBeamGeometry(N_channels=8, N_proj=8, eid=True, detector_file=det)
_, i0, mus = md.decomposition_tables(ct, synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80))


def roots_of(g):
    with np.errstate(all='ignore'):
        return gn_oracle.newton_solve(g.T.reshape(2, 1, -1), i0, mus, 150).reshape(-1, 2)


def lagrange_w(t, nodes):
    w = np.ones((len(t), len(nodes)))
    for a, xa in enumerate(nodes):
        for b, xb in enumerate(nodes):
            if a != b:
                w[:, a] *= (t - xb) / (xa - xb)
    return w


def catmull_w(t):
    t2, t3 = t * t, t * t * t
    return np.stack([-0.5 * t3 + t2 - 0.5 * t, 1.5 * t3 - 2.5 * t2 + 1.0, -1.5 * t3 + 2.0 * t2 + 0.5 * t, 0.5 * t3 - 0.5 * t2], 1)


rng = np.random.default_rng(0)
for cells in (128, 256):
    q.GATE_CELLS = cells
    p = q.newton_start_grid(i0, mus)
    h = p['head']
    n = cells
    r = roots_of(p['corner_g']).reshape(n + 1, n + 1, 2)
    # random points in the physical part of the plane
    m = 20000
    fx = rng.uniform(3, n * (1 + np.log(q.GATE_U_MAX) / -np.log(q.GATE_U_MIN)) - 3, m)      # up to u0 = GATE_U_MAX
    fy = rng.uniform(0.3 * n, 0.7 * n, m)
    u0 = np.exp(h[4] + fx / h[5])
    u1 = u0 * (h[6] + fy / h[7])
    g = np.stack([h[0] * np.exp(-u0 / h[2]), h[1] * np.exp(-u1 / h[2])], 1)
    true = roots_of(g)
    i, j = fx.astype(int), fy.astype(int)
    tx, ty = fx - i, fy - j
    size = np.maximum(np.abs(true).max(1), 1.0)
    ok = np.isfinite(true).all(1)
    for name, wx, wy, off in (('Catmull-Rom 4x4', catmull_w(tx), catmull_w(ty), 1),
                              ('Lagrange 4x4', lagrange_w(tx, [-1, 0, 1, 2]), lagrange_w(ty, [-1, 0, 1, 2]), 1),
                              ('Lagrange 6x6', lagrange_w(tx, [-2, -1, 0, 1, 2, 3]), lagrange_w(ty, [-2, -1, 0, 1, 2, 3]), 2)):
        k = wx.shape[1]
        s = np.zeros((m, 2))
        for a in range(k):
            for b in range(k):
                s += (wx[:, a] * wy[:, b])[:, None] * r[i + a - off, j + b - off]
        d = np.abs(s - true).max(1) / size
        good = ok & np.isfinite(d)
        print(f'{cells} cells, {name}: |interpolant - root| / size: median {np.median(d[good]):.2e}  p90 {np.percentile(d[good], 90):.2e}  '
              f'p99 {np.percentile(d[good], 99):.2e}  max {d[good].max():.2e}')
