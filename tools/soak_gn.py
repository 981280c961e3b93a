"""Randomised soak of the float64 Newton kernel's EXACT invariants on one MI355X (a one-off campaign, log under
profiles/).  Every case draws tables (1..300 energies, all energy classes, attenuation far above the clip-free bound now
and then), measurements (noise-free, noisy, photon-starved, zero, negative, inf, NaN pixels mixed in) and an iteration
count 0..80, and demands, BIT FOR BIT including the NaN payloads:

  * (exact mode, stop_tol = 0) the repeated-state exit returns what the full loop returns (DEXCT_GN_FLAG_FULL_LOOP);
  * any cap on the grid of the tile queue (blocks_per_cu), the natural instead of the thick-first order of the hand-out
    (DEXCT_GN_FLAG_NATURAL_ORDER) and the results written in the reference's order from [view][channel][row] input
    (out_rc: 4 x 16 tiles collected in LDS) return the same;
  * the cooperative kernel (4 waves per 64 pixels) returns what ITS full loop returns;
  * with the air mask: masked pixels are exactly 0 and the others unchanged;
and, NOT bit for bit (round 4): the default tolerance stop and the short cut (where the tables allow it, else it falls back
to the single launch) within 1e-10 of the exact result, and
the cooperative kernel within 1e-9 of the lane kernel, on the pixels the NumPy restatement answers stably (below);
and agreement to 1e-9 with the NumPy restatement of the reference, all on the STABLE pixels.

THE STABILITY SCREEN (version 3, round 5).  Two float64 arithmetics - another order of the energy sums, another 2x2 solve -
differ after one Newton step by about eps * cond(H) * |step| / size: the uncertainty of the computed step itself.  Over the
iterations these add up, a creeping iteration (Jacobian of the Newton map near 1) keeps them, a wild transient amplifies them
and can carry the two arithmetics into different basins of noisy counts that admit two fixed points.  With
w = eps * SUM over all iterations of cond(H_k) |step_k| / max(|a_(k+1)|, 1) of the restatement (gn_oracle.newton_solve(...,
return_sensitivity=True); infinite once eps * cond(H_k) > 1e-4: such a solve is numerically singular and even the exact 0 it may
return - one energy: the numerators cancel - is one arithmetic's rounding residue), a pixel is compared only if
  * the restatement's answer is finite, below 1e6, and its counts are finite and positive;
  * the answer moves by at most 1e-11 relative under a 1e-13 perturbation of the counts;
  * w <= 1e-11: what the computed steps are uncertain by, all added up, is a hundredth of the tightest tolerance asked below;
  * (version 3) two TWIN trajectories of the restatement that receive, after every step, a kick of that step's uncertainty
    (eps cond(H_k) |step_k|, at least eps; fixed signs, the second twin the opposite ones) end within 1e-11 of the pixel's own
    result: w adds the uncertainties up as if nothing amplified them, and an iteration that WANDERS before it settles does.
    Seed 1795, pixel 2221 (0.45 and 0.24 counts; found when the campaign went on to seeds 1000 - 2499, profiles/r05_soak_gn3.log):
    35 irregular iterations of size 1e-3 around the solution before it converges at the 47th; w = 4.5e-12, yet a 1e-15 kick at
    iteration 12 moves the result by 3e-3, and the cooperative kernel ended 1.3 away from the lane kernel.  The rule costs 0.5 %
    of the pixels version 2 compared (seeds 0 - 39).
Version 1 had the permutation of the energies in place of the last rule; with 2 or 3 energies a permutation changes little or
nothing (seed 319's was a swap of two, 468's the identity) and photon-starved pixels still creeping at the last iteration through
Hessians of condition 1e9 - 1e11 passed it: the cooperative kernel, the lane kernel and the restatement then part by 1e-9
at the iteration where cond(H) passes 1e7 (tools/probes/gn_soak_traj.py, profiles/r05_soak_traj.log: seeds 319, 525, 468 -
three arithmetics, three answers, the differences of the same size).  (A first form of version 2 took the largest term instead
of the sum and forgave a rough transient to a pixel that converged afterwards: seeds 29 and 468 of profiles/r05_soak_gn.log - a
transient uncertain by 2e-10 that ends in another basin, a creep of 61 iterations whose 5e-12 per step add up to 1e-9.)  With it
every comparison below is exact: no pixel may exceed its tolerance.

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
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from soak_cases import draw

dev = None


def run(g, i0, mus, n_iters, knobs, mask_max=None, stop_tol=0.0, kernel=1, out_rc=None, two_level=False):
    """``knobs``: full_loop / natural_order / blocks_per_cu of gn_device (dexct_gn_options.flags / .blocks_per_cu)"""
    out = md.gn_device(g[0], g[1], i0, mus, n_iters, 'f64', mask_max=mask_max, stop_tol=stop_tol, kernel=kernel, out_rc=out_rc,
                       two_level=two_level, **knobs)
    torch.cuda.synchronize()
    return out


def check_case(seed, stats=None):
    """One random case through every invariant; returns (description, list of violations).  ``stats``: a dict the campaign's
    counters are added to."""
    global dev
    if dev is None:
        dev = torch.device('cuda:0')
        os.environ['DEXCT_CACHE_DIR'] = 'off'      # every case calibrates its own gate: nothing is read from or left on the disk
    stats = stats if stats is not None else {}
    c = draw(seed)
    n_e, i0, mus, n_v, n_c, g, kind, n_iters = (c[k] for k in ('n_e', 'i0', 'mus', 'n_v', 'n_c', 'g', 'kind', 'n_iters'))
    what = f'{n_e} energies, {n_v} x {n_c} pixels, {kind}, {n_iters} iterations'
    g_d = torch.tensor(g, dtype=torch.float32 if kind == 'float32' else torch.float64, device=dev)
    bad = []
    try:
        base = run(g_d, i0, mus, n_iters, {})
        bits = base.view(torch.int64)
        for env in ({'full_loop': True}, {'blocks_per_cu': 1}, {'blocks_per_cu': 3}, {'natural_order': True},
                    {'natural_order': True, 'blocks_per_cu': 2}):
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
            tol = md.DEFAULT_STOP_TOL or 1e-12
            got = run(g3, i0, mus, n_iters, {}, out_rc=(rr, n_c // rr), stop_tol=tol, two_level='start')
            want = run(g_d, i0, mus, n_iters, {}, stop_tol=tol, two_level='start').reshape(n_v, n_c // rr, rr, 2).permute(0, 2, 1, 3).contiguous()
            if not torch.equal(got.view(torch.int64), want.view(torch.int64)):
                bad.append(f'short cut, out_rc=({rr}, {n_c // rr}): {int((got.view(torch.int64) != want.view(torch.int64)).sum())} values differ')
        coop = run(g_d, i0, mus, n_iters, {}, kernel=2)
        coop_full = run(g_d, i0, mus, n_iters, {'full_loop': True}, kernel=2)
        if not torch.equal(coop.view(torch.int64), coop_full.view(torch.int64)):
            bad.append('cooperative kernel: exact exit differs from its full loop')
        default = run(g_d, i0, mus, n_iters, {}, stop_tol=md.DEFAULT_STOP_TOL or 1e-12)
        # the short cut: start values from the tabulated fixed points, gated by the reference iteration's own step counts -
        # same contract as the tolerance stop (an explicit tolerance: also on pairs the calibration calls ill-posed)
        short = run(g_d, i0, mus, n_iters, {}, stop_tol=md.DEFAULT_STOP_TOL or 1e-12, two_level='start')
        short1 = run(g_d, i0, mus, n_iters, {}, stop_tol=md.DEFAULT_STOP_TOL or 1e-12, two_level='one')
        mode = md.last_gn_stats()['mode']
        stats[mode] = stats.get(mode, 0) + 1
        gmax = torch.tensor(float(np.nanmax(np.where(np.isfinite(g[0]), g[0], -np.inf))), dtype=torch.float64, device=dev)
        masked = run(g_d, i0, mus, n_iters, {}, mask_max=gmax)
        air = g_d[0].double() >= 0.95 * gmax
        if not (torch.equal(masked[air], torch.zeros_like(masked[air])) and
                torch.equal(masked[~air].view(torch.int64), base[~air].view(torch.int64))):
            bad.append('air mask: masked pixels not exactly 0 or others changed')
        with np.errstate(all='ignore'):
            ref, sens = gn_oracle.newton_solve(g, i0, mus, n_iters, return_sensitivity=True)
            ref_p = gn_oracle.newton_solve(g * (1 + 1e-13), i0, mus, n_iters)
            size = np.maximum(np.abs(ref).max(-1), 1.0)
            ok = np.isfinite(ref).all(-1) & (np.abs(ref).max(-1) < 1e6) & np.isfinite(g).all(0) & (g > 0).all(0)
            eps = np.finfo(np.float64).eps
            ok &= (np.abs(ref - ref_p).max(-1) <= 1e-11 * size) & (eps * sens['walk'] <= 1e-11)
            # what the twin-trajectory rule of screen v3 ALONE takes out of the comparison (advisor finding of round 5: the rule was
            # added after seed 1795 failed under v2, so how much it hides is counted per case and bounded over the campaign)
            twin_only = ok & ~(sens['twin'] <= 1e-11)
            ok &= sens['twin'] <= 1e-11
        stats['twin_only'] = stats.get('twin_only', 0) + int(twin_only.sum())
        stats['pixels'] = stats.get('pixels', 0) + n_v * n_c
        stats['stable'] = stats.get('stable', 0) + int(ok.sum())
        if ok.any():
            lane = base.cpu().numpy()
            for name, other, tol in (('NumPy restatement', ref, 1e-9), ('default tolerance stop', default.cpu().numpy(), 1e-10),
                                     ('cooperative kernel', coop.cpu().numpy(), 1e-9), ('short cut, two steps', short.cpu().numpy(), 1e-10),
                                     ('short cut, one step', short1.cpu().numpy(), 1e-10)):
                d = (np.abs(other - lane).max(-1) / size)[ok]
                n_bad = int((~(d <= tol)).sum())
                if n_bad:
                    k = np.flatnonzero(ok.ravel())[np.flatnonzero(~(d <= tol))[0]]
                    first = (f'; first: pixel {k} counts {g.reshape(2, -1)[:, k].tolist()} lane {lane.reshape(-1, 2)[k].tolist()} other '
                             f'{other.reshape(-1, 2)[k].tolist()} walk {float(np.finfo(np.float64).eps * sens["walk"].ravel()[k]):.1e}')
                    bad.append(f'{name}: {n_bad} of {int(ok.sum())} stable pixels beyond {tol:g} of the exact lane kernel (worst {np.nanmax(d):.2e}, {int(np.isnan(d).sum())} not comparable{first})')
    except Exception as exc:
        bad = [f'{type(exc).__name__}: {exc}']
    return what, bad


TWIN_ONLY_MAX_SHARE = 0.02


if __name__ == '__main__':
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    fails, stats = 0, {}
    for case in range(n_cases):
        what, bad = check_case(seed0 + case, stats)
        if bad:
            fails += 1
            print(f'FAIL seed {seed0 + case}: {what}: ' + '; '.join(bad), flush=True)
        if case % 100 == 99 or case == n_cases - 1:
            print(f'{case + 1} cases from seed {seed0}, {fails} failed, {stats.get("pixels", 0):.3g} pixels x 15 launches, {stats.get("stable", 0):.3g} stable '
                  f'pixels compared (screen v3; {stats.get("twin_only", 0)} excluded by the twin rule alone); short-cut launches ended up as '
                  f'{ {k: v for k, v in stats.items() if k not in ("pixels", "stable", "twin_only")} }; {time.time() - t0:.0f} s', flush=True)
    # the twin rule may only remove a small share: results on wandering pixels are kernel-dependent, but a rule that
    # grew to hide a kernel's loss of digits would show here
    twin_share = stats.get('twin_only', 0) / max(stats.get('pixels', 0), 1)
    if twin_share > TWIN_ONLY_MAX_SHARE:
        print(f'FAIL: the twin-trajectory rule alone excluded {twin_share:.2%} of the pixels (bound {TWIN_ONLY_MAX_SHARE:.0%})', flush=True)
        fails += 1
    sys.exit(1 if fails else 0)
