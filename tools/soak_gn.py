"""Randomised soak of the float64 Newton kernel's EXACT invariants on one MI355X (a one-off campaign, log under
profiles/).  Every case draws tables (1..300 energies, all energy classes, attenuation far above the clip-free bound now
and then), measurements (noise-free, noisy, photon-starved, zero, negative, inf, NaN pixels mixed in) and an iteration
count 0..80, and demands, BIT FOR BIT including the NaN payloads:

  * (exact mode, stop_tol = 0) the repeated-state exit returns what the full loop returns (DEXCT_GN_FULL_LOOP=1);
  * the 5-waves-per-SIMD register allocation (DEXCT_GN_MINW=5), any cap on the grid of the tile queue, the natural instead
    of the thick-first order of the hand-out (DEXCT_GN_SORT=0) and the results written in the reference's order from
    [view][channel][row] input (out_rc: 4 x 16 tiles collected in LDS) return the same;
  * the cooperative kernel (4 waves per 64 pixels) returns what ITS full loop returns;
  * with the air mask: masked pixels are exactly 0 and the others unchanged;
and, NOT bit for bit (round 4): the default tolerance stop and both modes of the two-level solve (start values only; with the
coarse launch - where the tables allow them, else they fall back to the single launch) within 1e-10 of the exact result, and
the cooperative kernel within 1e-9 of the lane kernel, on the pixels the NumPy restatement answers stably (below);
and, as a sanity check of the arithmetic (a statistic, not an invariant), agreement to 1e-9 with the NumPy restatement of
the reference on the pixels where that one is finite and insensitive both to a 1e-13 perturbation of its input and to the
order of its own sums.  The few pixels beyond 1e-9 that this screen lets through (about 1 in 1e5 here) are of two kinds,
traced iteration by iteration: photon-starved pixels still crawling after 60 iterations along an ill-conditioned valley
(differences of 1e-15 grow to 1e-6), and transients with a numerically SINGULAR Hessian (tables scaled x30, a negative
iterate, expected counts 1e16 times the measured ones: h00 h11 - h01^2 cancels to a rounding residue or to exactly 0),
where the restatement's residue yields a step of exactly 2^-9 and the kernel's closed-form
solve returns inf - the class the unscreened reference golden documents (tests/test_gpu_gn.py).

    python tools/soak_gn.py [n_cases] [first_seed]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dex_ct_sim_amd import matdecomp as md
from oracle import gn_oracle

KNOBS = ('DEXCT_GN_FULL_LOOP', 'DEXCT_GN_MINW', 'DEXCT_GN_BLOCKS_PER_CU', 'DEXCT_GN_SORT')
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device('cuda:0')


def run(g, i0, mus, n_iters, env, mask_max=None, stop_tol=0.0, kernel=1, out_rc=None, two_level=False):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    out = md.gn_device(g[0], g[1], i0, mus, n_iters, 'f64', mask_max=mask_max, stop_tol=stop_tol, kernel=kernel, out_rc=out_rc,
                       two_level=two_level)
    torch.cuda.synchronize()
    for k in KNOBS:
        os.environ.pop(k, None)
    return out


def hessian_cond(a, g, i0, mus):
    """Condition number of the 2x2 Hessian of the Poisson likelihood (matdecomp.py:116-123) at the states a [P, 2] for the
    measurements g [2, P]; inf where it is exactly singular or not finite."""
    with np.errstate(all='ignore'):
        at = np.exp(np.clip(-(a @ mus), -700, 700))                                   # [P, E]
        nu = at @ i0.T                                                                # [P, 2]
        gr = -np.einsum('ke,me,pe->pkm', i0, mus, at)
        hs = np.einsum('ke,me,ne,pe->pkmn', i0, mus, mus, at)
        c, q = g.T / nu - 1.0, g.T / nu ** 2
        H = -(c[:, :, None, None] * hs - q[:, :, None, None] * gr[:, :, :, None] * gr[:, :, None, :]).sum(1)
        out = np.full(a.shape[0], np.inf)
        fin = np.isfinite(H).all(axis=(1, 2))
        if fin.any():
            out[fin] = np.linalg.cond(H[fin])
    return out


t0 = time.time()
fails, n_pix, n_cmp, n_off = 0, 0, 0, 0
n_few, min_cond_few = 0, float('inf')
n_mode = {}
for case in range(n_cases):
    seed = seed0 + case
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
    weird = rng.random(g.shape) < 0.01                          # pathological measurements mixed in
    g[weird] = rng.choice([0.0, -1.0, np.inf, np.nan, 1e-300, 1e300], int(weird.sum()))
    dtype = torch.float32 if kind == 'float32' else torch.float64
    g_d = torch.tensor(g, dtype=dtype, device=dev)
    g = g_d.double().cpu().numpy()
    n_iters = int(rng.choice([0, 1, 2, 5, 9, 30, 50, 50, 50, 61, 80]))
    bad = []
    try:
        base = run(g_d, i0, mus, n_iters, {})
        bits = base.view(torch.int64)
        for env in ({'DEXCT_GN_FULL_LOOP': '1'}, {'DEXCT_GN_MINW': '5'}, {'DEXCT_GN_BLOCKS_PER_CU': '1'}, {'DEXCT_GN_BLOCKS_PER_CU': '3'},
                    {'DEXCT_GN_SORT': '0'}, {'DEXCT_GN_SORT': '0', 'DEXCT_GN_MINW': '5'}):
            got = run(g_d, i0, mus, n_iters, env)
            if not torch.equal(got.view(torch.int64), bits):
                bad.append(f'{env}: {int((got.view(torch.int64) != bits).sum())} values differ')
        # the same pixels read as [view][channel][row] with the results written as [view][row][channel]
        rr = int(np.random.default_rng(seed).choice([1, 3, 16, 17]))      # (its own stream: the case's draws stay those of round 3)
        if n_c % rr == 0 and n_c // rr >= 1:
            g3 = g_d.reshape(2, n_v, n_c // rr, rr)
            got = run(g3, i0, mus, n_iters, {}, out_rc=(rr, n_c // rr))
            want = base.reshape(n_v, n_c // rr, rr, 2).permute(0, 2, 1, 3).contiguous()
            if not torch.equal(got.view(torch.int64), want.view(torch.int64)):
                bad.append(f'out_rc=({rr}, {n_c // rr}): {int((got.view(torch.int64) != want.view(torch.int64)).sum())} values differ')
        coop = run(g_d, i0, mus, n_iters, {}, kernel=2)
        coop_full = run(g_d, i0, mus, n_iters, {'DEXCT_GN_FULL_LOOP': '1'}, kernel=2)
        if not torch.equal(coop.view(torch.int64), coop_full.view(torch.int64)):
            bad.append('cooperative kernel: exact exit differs from its full loop')
        default = run(g_d, i0, mus, n_iters, {}, stop_tol=None)
        # the short cut (round 4): start values from the tabulated fixed points (+ a coarse launch on a short quadrature), gated by the
        # reference iteration's own step counts - same contract as the tolerance stop
        modes = {}
        for mode in ('start', 'coarse'):
            modes[mode] = run(g_d, i0, mus, n_iters, {}, stop_tol=None, two_level=mode)
            n_mode[md.last_gn_stats()['mode']] = n_mode.get(md.last_gn_stats()['mode'], 0) + 1
        gmax = torch.tensor(float(np.nanmax(np.where(np.isfinite(g[0]), g[0], -np.inf))), dtype=torch.float64, device=dev)
        masked = run(g_d, i0, mus, n_iters, {}, mask_max=gmax)
        air = g_d[0].double() >= 0.95 * gmax
        if not (torch.equal(masked[air], torch.zeros_like(masked[air])) and
                torch.equal(masked[~air].view(torch.int64), base[~air].view(torch.int64))):
            bad.append('air mask: masked pixels not exactly 0 or others changed')
        few = n_e < 3        # one or two energies: two materials are separated badly or not at all - CHECKED below, not assumed
        with np.errstate(all='ignore'):
            ref = gn_oracle.newton_solve(g, i0, mus, n_iters)
            ref_p = gn_oracle.newton_solve(g * (1 + 1e-13), i0, mus, n_iters)
            # the same restatement with the energies in another order: its sums round differently, which is what the
            # kernel's do too - a pixel whose answer depends on that (an ill-conditioned transient amplifies rounding a
            # decade per step) says nothing about the kernel
            perm = rng.permutation(n_e)
            ref_q = gn_oracle.newton_solve(g, i0[:, perm], mus[:, perm], n_iters)
            ok = np.isfinite(ref).all(-1) & (np.abs(ref).max(-1) < 1e6) & np.isfinite(g).all(0) & (g > 0).all(0)
            size = np.maximum(np.abs(ref).max(-1), 1.0)
            ok &= (np.abs(ref - ref_p).max(-1) <= 1e-11 * size) & (np.abs(ref - ref_q).max(-1) <= 1e-11 * size)
            err = np.abs(base.cpu().numpy() - ref)[ok] / np.maximum(np.abs(ref[ok]).max(-1, keepdims=True), 1.0)
            # (not with one or two energies: the Hessian is singular there and any two arithmetics part ways - the case the
            # condition-number check below handles for the comparison with the restatement)
            for name, other, tol in (() if few else (('default tolerance stop', default, 1e-10), ('cooperative kernel', coop, 1e-9),
                                                     ('two-level solve, start values only', modes['start'], 1e-10),
                                                     ('two-level solve with the coarse launch', modes['coarse'], 1e-10))):
                d = (np.abs(other.cpu().numpy() - base.cpu().numpy())[ok] / np.maximum(np.abs(ref[ok]).max(-1, keepdims=True), 1.0)) if ok.any() else np.zeros(1)
                n_bad = int((~(d.max(-1) <= tol)).sum()) if ok.any() else 0
                if n_bad > max(2, 1e-3 * ok.sum()):
                    bad.append(f'{name}: {n_bad} of {int(ok.sum())} stable pixels beyond {tol:g} of the exact lane kernel')
        if err.size:
            n_cmp += int(ok.sum())
            dev_px = ~(err.max(-1) <= 1e-9)
            off = int(dev_px.sum())
            if few and off:
                # tables of one or two energies: every deviating pixel must have a (near-)singular Hessian at the restatement's
                # own answer - condition number >= 1e8 (one energy: exactly singular, the two attenuation vectors are parallel)
                cond = hessian_cond(ref[ok][dev_px], g[:, ok][:, dev_px], i0, mus)
                n_few += off
                min_cond_few = min(min_cond_few, float(np.min(cond)))
                if not np.all(cond >= 1e8):
                    bad.append(f'{n_e} energies: {int((cond < 1e8).sum())} deviating pixels with a WELL-conditioned Hessian (min cond {np.min(cond):.2e})')
            else:
                n_off += off
                if off > max(2, 1e-3 * ok.sum()):
                    bad.append(f'vs the NumPy restatement: {off} of {int(ok.sum())} stable pixels beyond 1e-9')
    except StopIteration:
        pass
    except Exception as exc:
        bad = [f'{type(exc).__name__}: {exc}']
    n_pix += n_v * n_c
    if bad:
        fails += 1
        print(f'FAIL seed {seed}: {n_e} energies, {n_v} x {n_c} pixels, {kind}, {n_iters} iterations: ' + '; '.join(bad), flush=True)
    if case % 100 == 99 or case == n_cases - 1:
        print(f'{case + 1} cases, {fails} failed, {n_pix:.3g} pixels x 12 launches, {n_cmp:.3g} stable pixels compared with the '
              f'NumPy restatement ({n_off} beyond 1e-9 with >= 3 energies; with 1-2 energies {n_few} beyond 1e-9, the best '
              f'conditioned of them has Hessian cond {min_cond_few:.1e}); two-level launches ended up as {n_mode}; {time.time() - t0:.0f} s', flush=True)
sys.exit(1 if fails else 0)
