"""CPU experiment behind DESIGN.md section 4.4 ("why the last two rows work"): how long does the float64 Newton iterate
wander in its last bits before a state repeats, as a function of HOW the expected counts nu are summed and how the
residual g / nu - 1 is formed?  NumPy emulation of the update (matdecomp.py:114-125) on random water / bone pixels with
the benchmark's tables; the exit rule is the kernel's (state equal to one of the previous 9).  No GPU needed.

    python tools/probes/gn_wander.py [n_pixels]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import matdecomp as md, synthetic

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
ct = dx.FanBeamGeometry(N_channels=800, N_proj=1000, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det,
                        N_rows=1)
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
_, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
i0, mus = np.asarray(i0, float), np.asarray(mus, float)
keep = (i0 != 0).any(axis=0)
i0, mus = i0[:, keep], mus[:, keep]
nE = i0.shape[1]
rng = np.random.default_rng(0)
a_true = np.stack([rng.uniform(0, 40, N), np.where(rng.random(N) < 0.5, 0.0, rng.uniform(0, 6, N))], 1)
g = (np.exp(-(a_true[:, :1] * mus[0] + a_true[:, 1:] * mus[1])) @ i0.T).astype(np.float32).astype(np.float64)
wg = i0[:, None, :] * mus[None, :, :]
wh = i0[:, None, None, :] * (mus[None, :, :] * mus[:, None, :])[None]
tabs = np.concatenate([i0.reshape(2, -1), wg.reshape(4, -1), wh.reshape(8, -1)], 0)            # [14, e]


def seqsum(att, cols, parts):
    accs = [np.zeros((att.shape[0], len(cols))) for _ in range(parts)]
    for e in range(nE):
        accs[e % parts] = accs[e % parts] + att[:, e:e + 1] * tabs[None, cols, e]
    while len(accs) > 1:
        accs = [accs[i] + accs[i + 1] if i + 1 < len(accs) else accs[i] for i in range(0, len(accs), 2)]
    return accs[0]


def run(nu_parts, residual, n_iters=50, hist=8):
    a = np.full((N, 2), 1e-6)
    states, exit_it, done = [a.copy()], np.full(N, n_iters), np.zeros(N, bool)
    for k in range(n_iters):
        att = np.exp(np.clip(-(a[:, :1] * mus[0] + a[:, 1:] * mus[1]), -700, 700))
        if nu_parts == 'longdouble':
            nu = (att[:, None, :].astype(np.longdouble) * tabs[None, :2].astype(np.longdouble)).sum(axis=2).astype(float)
        else:
            nu = seqsum(att, [0, 1], nu_parts)
        rest = seqsum(att, list(range(2, 14)), 1)
        gr, hs = rest[:, :4].reshape(-1, 2, 2), rest[:, 4:].reshape(-1, 2, 2, 2)
        inv = 1.0 / nu
        ratio = g * inv
        c = ratio - 1.0 if residual == 'quotient' else (g - nu) * inv
        q = ratio * inv
        dF = np.einsum('nk,nkm->nm', c, gr)
        H = -np.einsum('nk,nkmj->nmj', c, hs) + np.einsum('nk,nkm,nkj->nmj', q, gr, gr)
        det_h = H[:, 0, 0] * H[:, 1, 1] - H[:, 0, 1] * H[:, 1, 0]
        a = a - np.stack([(H[:, 1, 1] * dF[:, 0] - H[:, 0, 1] * dF[:, 1]) / det_h,
                          (H[:, 0, 0] * dF[:, 1] - H[:, 1, 0] * dF[:, 0]) / det_h], 1)
        rep = np.zeros(N, bool)
        for s in states[-(hist + 1):]:
            rep |= (s == a).all(axis=1)
        exit_it[rep & ~done] = k + 1
        done |= rep
        states.append(a.copy())
    conv = np.argmax([np.abs(s - states[-1]).max(axis=1) <= 1e-12 * np.abs(states[-1]).max(axis=1) for s in states], axis=0)
    return exit_it, conv


print(f'{N} pixels, {nE} energies; exit = first iteration whose state repeats one of the 9 before it')
for nu_parts, residual in [(1, 'quotient'), (2, 'quotient'), ('longdouble', 'quotient'), (1, 'subtracted'), (2, 'subtracted'),
                           (4, 'subtracted'), ('longdouble', 'subtracted')]:
    ex, conv = run(nu_parts, residual)
    print(f'nu in {nu_parts!s:>10} partial sum(s), residual {residual:10s}: mean exit {ex.mean():5.2f}   never within 50: '
          f'{(ex == 50).mean():.3f}   (converged to 1e-12 after {conv.mean():.1f})', flush=True)
