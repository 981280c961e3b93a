"""CPU studies behind the chord step of the Newton short cut (round 6; profiles/r06_gn_chord.md).  NumPy only.

    python tools/probes/gn_chord_cpu.py e0 [bundled|kramers]            where in the data plane the 6 x 6 Lagrange interpolant of the
                                                                        tabulated fixed points is far from a pixel's own (256 / 512 cells
                                                                        per axis, either axis refined alone)
    python tools/probes/gn_chord_cpu.py eps [bundled|kramers] [cells..] the misfit of the table's GRADIENT against the exact inverse
                                                                        Jacobian, and the share of noisy water-like rays whose one-step
                                                                        bound misses 1e-12 per component, per grid size and eps rule
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import matdecomp as md, quadrature as q, synthetic

what = sys.argv[1] if len(sys.argv) > 1 else 'e0'
which = sys.argv[2] if len(sys.argv) > 2 else 'bundled'
INPUT = os.path.join(ROOT, 'dex-ct-sim_amd', 'input')
ct = dx.FanBeamGeometry(N_channels=8, N_proj=8, eid=True, detector_file=os.path.join(INPUT, 'detector', 'eta_eid_mv.bin'))
if which == 'bundled':
    specs = [dx.xRaySpectrum(os.path.join(INPUT, 'spectrum', f'{n}_1mGy_float32.bin'), n) for n in ('140kV', '80kV')]
else:
    specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
_, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
used = np.any(i0 > 0, axis=0)
i0, mus = i0[:, used], mus[:, used]


def roots_of(g, a_init=None):
    # Newton on ln nu = ln g from a decent start (the exact root is what matters here, not the reference's branch)
    lg = np.log(g)
    a = np.zeros_like(g) if a_init is None else a_init.copy()
    if a_init is None:
        # crude start: effective attenuation
        mu_eff = (i0 @ mus.T) / i0.sum(1)[:, None]          # [k, p]
        a = np.linalg.solve(mu_eff, (np.log(i0.sum(1))[None, :] - lg).T).T
    for it in range(40):
        att = np.exp(-(a @ mus))
        nu = att @ i0.T
        G = np.einsum('ne,ke,pe->nkp', att, i0, mus)
        L = -G / nu[:, :, None]
        rhs = lg - np.log(nu)
        det = L[:, 0, 0] * L[:, 1, 1] - L[:, 0, 1] * L[:, 1, 0]
        d0 = (L[:, 1, 1] * rhs[:, 0] - L[:, 0, 1] * rhs[:, 1]) / det
        d1 = (-L[:, 1, 0] * rhs[:, 0] + L[:, 0, 0] * rhs[:, 1]) / det
        a = a + np.stack([d0, d1], 1)
        if np.nanmax(np.abs(rhs)) < 1e-15:
            break
    return a

def lagrange_w(t, nodes):
    w = np.ones((len(t), len(nodes)))
    for a_, xa in enumerate(nodes):
        for b_, xb in enumerate(nodes):
            if a_ != b_:
                w[:, a_] *= (t - xb) / (xa - xb)
    return w



with np.errstate(all='ignore'):
    if what == 'e0':
        rng = np.random.default_rng(0)
        q.GATE_CELLS = 512
        p = q.newton_start_grid(i0, mus)
        h = p['head']
        t0 = time.time()
        r512 = roots_of(p['corner_g']).reshape(513, 513, 2)
        print('table 512 in', time.time() - t0, 's', flush=True)
        m = 40000
        for (sx, sy, label) in ((2, 2, '256 x 256'), (2, 1, '256 (ln u0) x 512 (ratio)'), (1, 2, '512 x 256'), (1, 1, '512 x 512')):
            r = r512[::sx, ::sy]
            nx, ny = 512 // sx, 512 // sy
            xmax = 512 * (1 + np.log(q.GATE_U_MAX) / -np.log(q.GATE_U_MIN))
            FX = rng.uniform(8, xmax - 8, m)           # in 512-units
            FY = rng.uniform(0.15 * 512, 0.85 * 512, m)
            u0 = np.exp(h[4] + FX / h[5])
            u1 = u0 * (h[6] + FY / h[7])
            g = np.stack([h[0] * np.exp(-u0 / h[2]), h[1] * np.exp(-u1 / h[2])], 1)
            fx, fy = FX / sx, FY / sy
            i, j = fx.astype(int), fy.astype(int)
            tx, ty = fx - i, fy - j
            wx, wy = lagrange_w(tx, [-2, -1, 0, 1, 2, 3]), lagrange_w(ty, [-2, -1, 0, 1, 2, 3])
            s = np.zeros((m, 2))
            for a_ in range(6):
                for b_ in range(6):
                    s += (wx[:, a_] * wy[:, b_])[:, None] * r[i + a_ - 2, j + b_ - 2]
            true = roots_of(g, s)
            size = np.maximum(np.abs(true).max(1), 1.0)
            d = np.abs(s - true).max(1) / size
            ok = np.isfinite(d)
            print(f'{label}: e0 / size median {np.median(d[ok]):.2e} p90 {np.percentile(d[ok], 90):.2e} p99 {np.percentile(d[ok], 99):.2e} max {d[ok].max():.2e}')
            # where are the worst: by ratio band and by ln u0 band
            for name, coord, edges in (('ratio band', FY / 512, [0.15, 0.25, 0.4, 0.6, 0.75, 0.85]), ('ln u0 band', FX / xmax, [0, 0.25, 0.5, 0.75, 0.9, 1.0])):
                print('   ', name, ' '.join(f'[{lo:.2f},{hi:.2f}): max {d[ok & (coord >= lo) & (coord < hi)].max():.1e}' for lo, hi in zip(edges[:-1], edges[1:])))

    else:
        cells_list = [int(c) for c in sys.argv[3:]] or [256, 320, 384]
        for n in cells_list:
            q.GATE_CELLS = n
            p = q.newton_start_grid(i0, mus)
            h = p['head']
            r = roots_of(p['corner_g']).reshape(n + 1, n + 1, 2)
            start = np.concatenate([h, r.ravel(), np.zeros(4 * n * n)])
            # exact L at corners and centres
            def Lof(a):
                att = np.exp(-(a @ mus)); nu = att @ i0.T
                G = np.einsum('ne,ke,pe->nkp', att, i0, mus)
                return -G / nu[:, :, None]
            Lk = Lof(r.reshape(-1, 2)).reshape(n + 1, n + 1, 2, 2)
            gc = q.cell_centres(p)
            sc_ = q.centre_interpolant(start, n).reshape(-1, 2)
            rc = roots_of(gc, np.where(np.isfinite(sc_) & (sc_ != 0), sc_, 1.0))
            Lc = Lof(rc).reshape(n, n, 2, 2)
            def misfit(B, L):
                with np.errstate(all='ignore'):
                    m_ = np.abs(np.eye(2) - np.einsum('...pk,...kq->...pq', B, L)).sum(-1).max(-1)
                return np.where(np.isfinite(m_), m_, np.inf)
            e_c = misfit(q.table_gradient(start, p, 0.5, 0.5), Lc)
            e_k = e_c.copy()
            for tx_, ty_ in ((0., 0.), (0., 1.), (1., 0.), (1., 1.)):
                di, dj = int(tx_), int(ty_)
                e_k = np.maximum(e_k, misfit(q.table_gradient(start, p, tx_, ty_), Lk[di:n + di, dj:n + dj]))
            pad = np.pad(e_k, 1, mode='edge')
            e_nb = np.max([pad[1 + a:n + 1 + a, 1 + b:n + 1 + b] for a in (-1, 0, 1) for b in (-1, 0, 1)], axis=0)
            # water-like rays with noise: a0 in [1, 40], a1 = -0.0155 a0 + small, counts perturbed by noise
            rng = np.random.default_rng(3)
            m = 40000
            a0 = rng.uniform(1.0, 40.0, m)
            a = np.stack([a0, -0.0155 * a0 + rng.normal(0, 0.05, m)], 1)
            g = (np.exp(-(a @ mus)) @ i0.T) * (1 + rng.normal(0, 2e-3, (m, 2)))
            u = np.log(h[:2][None, :] / g) * h[2]
            fx = (np.log(u[:, 0]) - h[4]) * h[5]; tt = u[:, 1] / u[:, 0]; fy = (tt - h[6]) * h[7]
            inb = (fx > 3) & (fx < n - 3) & (fy > 3) & (fy < n - 3) & (u[:, 0] < q.GATE_U_MAX)
            fx, fy, g, tt, u0 = fx[inb], fy[inb], g[inb], tt[inb], u[inb, 0]
            i, j = fx.astype(int), fy.astype(int); tx, ty = fx - i, fy - j
            nodes = [-2, -1, 0, 1, 2, 3]
            wx, wy = lagrange_w(tx, nodes), lagrange_w(ty, nodes)
            s = np.zeros((len(fx), 2))
            for a_ in range(6):
                for b_ in range(6):
                    s += (wx[:, a_] * wy[:, b_])[:, None] * r[i + a_ - 2, j + b_ - 2]
            true = roots_of(g, s)
            e0_abs = np.abs(s - true).max(1)
            size_el = np.maximum(np.minimum(np.abs(true[:, 0]), np.abs(true[:, 1])), 1.0)
            size = np.maximum(np.abs(true).max(1), 1.0)
            print(f'{n} cells: water-like noisy rays: e0/size median {np.median(e0_abs / size):.2e} p90 {np.percentile(e0_abs / size, 90):.2e} p99 {np.percentile(e0_abs / size, 99):.2e}')
            for name, et in (('4 x max(3x3 cells)', 4 * e_nb[i, j]), ('2.5 x own cell', 2.5 * e_k[i, j]), ('4 x own cell', 4 * e_k[i, j])):
                print(f'   eps_tab = {name}: median {np.median(et):.2e}; fails per-component bar: {float((et * e0_abs > 2.5e-13 * size_el).mean()):.3f}; fails the norm bar: {float((et * e0_abs > 2.5e-13 * size).mean()):.4f}')
