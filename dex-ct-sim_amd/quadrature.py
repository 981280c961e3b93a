"""Host-side table preparation for two things the kernels run on (NumPy / SciPy, like the reference's own spectrum handling;
no ray and no pixel is computed here):

1. ``reduce_tables`` - a shorter energy quadrature for the DETECTION sum, with a verified error bound (opt-in:
   ``quadrature='reduced'``), described below;
2. the table of the Newton decomposition's short cut (matdecomp.gn_device; csrc/gn.hip gn_start; include/dexct.h,
   dexct_gn_options.pass / .start): ``newton_start_grid`` lays a cell grid over the plane of the two counts,
   ``assemble_start`` / ``validate_start`` turn what the library's own kernel returns on its corners and centres - where the
   reference's walk from 1e-6 ends, after how many steps - into the table of start values, step budgets and acceptance radii
   (profiles/r04_gn_two_level.md); ``pair_is_ill_posed`` reads off the same calibration whether the pair of spectra determines
   two thicknesses at all (the MV / kV pairs do not: they run the reference's fixed count).

The detected signal of a ray is ``sum_e w[s][e] exp(-sum_m mu[m][e] L_m)`` over the spectrum's energy grid (the weighting the
reference's decomposition assumes, matdecomp.py:146-150; 134 weighted bins for the 140 kVp spectrum).  As functions of the
energy these exponentials span a space of small numerical dimension (the attenuation curves of air / water / bone are smooth
and monotone), so a handful of the SAME energies with NEW positive weights reproduces every signal in a bounded domain of path
lengths to a few 1e-7 relative: a generalised Gauss quadrature.  The kernels run unchanged on the shorter table; the work per
ray falls with the node count.

What makes this usable rather than a guess:

* the domain is rigorous for the phantom at hand: 0 <= L_m <= l_max[m] (the diagonal of material m's bounding box; the grid's
  diagonal for id 0) and sum_m L_m <= c_max (the grid's diagonal) - no ray can leave it; signals attenuated below 1e-30 of
  the unattenuated one (which float32 counts cannot hold either) are outside the bound;
* the weights come from a linear programme (non-negative weights, every training signal within ``tol`` relative) whose vertex
  solutions are sparse; nodes already chosen for one spectrum cost nothing for the next, so spectra share nodes (the fused
  kernels evaluate one exponential per energy of the UNION grid);
* the result is then evaluated in float64 on an independent validation set (a regular grid over the domain, its edges and
  corners, and random points); points that exceed ``tol`` are added to the programme and it is solved again; the reduction is
  only handed out when the validation maximum is <= ``max_err`` (default 1e-6 relative, half the 2e-6 budget named in the
  round-3 review; the reference parity bar for this path is 1e-5), otherwise ``reduce_tables`` returns None and the caller
  keeps the full grid.

Everything here is host-side table preparation in NumPy / SciPy (like the reference's own spectrum handling); the detection
itself stays in the HIP kernels.
"""
import numpy as np

MAX_MATERIALS = 4          # the domain is sampled, not enumerated: only small tables
FLOOR = 1.0e-30            # signals below FLOOR x the unattenuated one are outside the bound (float32 cannot hold them either)
_cache = {}


def _domain_points(l_max, c_max, n_random, n_axis, rng, cw):
    """Points of {0 <= L_m <= l_max[m], sum_m cw[m] L_m <= c_max}: random ones (dense near the faces, where few materials are
    present), every axis and every pairwise edge."""
    l_max = np.asarray(l_max, dtype=np.float64)
    M = l_max.size
    pts = []
    # random: a Dirichlet direction (alpha < 1 favours sparse mixes) scaled into the polytope
    d = rng.dirichlet(np.full(M + 1, 0.7), n_random)[:, :M] * (c_max / cw)[None, :]
    d = np.minimum(d, l_max[None, :])
    # every third one pulled towards the origin on a logarithmic scale: spectra with weight at a few keV have features at
    # thicknesses of 1e-4 of the range (exp(-mu L) with mu in the thousands)
    d[::3] *= 10.0 ** (-6.0 * rng.random((len(d[::3]), 1)))
    pts.append(d)
    t = np.linspace(0.0, 1.0, n_axis)
    t[1::3] = 10.0 ** (-6.0 * (1.0 - t[1::3]))
    for m in range(M):                                   # one material alone
        p = np.zeros((n_axis, M))
        p[:, m] = t * min(l_max[m], c_max / cw[m])
        pts.append(p)
    for a in range(M):                                   # two materials filling the longest chord
        for b in range(a + 1, M):
            p = np.zeros((n_axis, M))
            p[:, a] = np.minimum(t * c_max / cw[a], l_max[a])
            p[:, b] = np.minimum((c_max - cw[a] * p[:, a]) / cw[b], l_max[b])
            pts.append(p)
    return np.vstack(pts)


def _grid_points(l_max, c_max, per_axis, cw):
    """A regular grid over the box, cut by sum_m cw[m] L_m <= c_max (points outside are projected onto the face)."""
    l_max = np.asarray(l_max, dtype=np.float64)
    axes = [np.linspace(0.0, min(l, c_max / c), per_axis) for l, c in zip(l_max, cw)]
    g = np.stack(np.meshgrid(*axes, indexing='ij'), axis=-1).reshape(-1, l_max.size)
    s = g @ cw
    over = s > c_max
    g[over] *= (c_max / s[over])[:, None]
    return g


def _solve(expo, F, w_s, cols, tol, cost):
    """min cost . u  s.t.  |sum_e u_e w_e exp_e(L) / F(L) - 1| <= tol for every training point, u >= 0."""
    from scipy.optimize import linprog
    A = expo[:, cols] * (w_s[cols] / 1.0)[None, :] / F[:, None]
    n = len(F)
    res = linprog(cost, A_ub=np.vstack([A, -A]), b_ub=np.concatenate([np.full(n, 1.0 + tol), np.full(n, tol - 1.0)]),
                  bounds=(0, None), method='highs')
    if res.status != 0:
        return None
    u = np.where(res.x > 1e-12 * res.x.max(), res.x, 0.0)
    return u


def reduce_tables(mu, w, l_max, c_max, tol=2.5e-7, max_err=1.0e-6, seed=0, rounds=6, total_weights=None):
    """mu [M, nE] (1/cm), w [S, nE] (>= 0) -> (cols, w_red [S, len(cols)], info) with
    ``|sum_c w_red[s][c] exp(-mu[:, cols[c]] . L) / sum_e w[s][e] exp(-mu[:, e] . L) - 1| <= info['max_rel_err'] <= max_err``
    at every validated point of the domain, or None when no shorter table with that property was found (more than
    MAX_MATERIALS rows, a failed programme, or a table that would not be shorter).  The domain: 0 <= L_m <= l_max[m] and
    sum_m total_weights[m] L_m <= c_max (weights 1 by default: a bound on the total path)."""
    mu = np.asarray(mu, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    cw = np.ones(mu.shape[0]) if total_weights is None else np.asarray(total_weights, dtype=np.float64)
    l_max = np.minimum(np.asarray(l_max, dtype=np.float64), c_max / cw)
    M, nE = mu.shape
    S = w.shape[0]
    if M > MAX_MATERIALS or M != l_max.size or np.any(w < 0.0) or nE < 24:
        return None
    key = (mu.tobytes(), w.tobytes(), l_max.tobytes(), cw.tobytes(), float(c_max), float(tol), float(max_err), int(seed))
    if key in _cache:
        return _cache[key]
    rng = np.random.default_rng(seed)
    train = _domain_points(l_max, c_max, 2500, 160, rng, cw)
    per_axis = {1: 4001, 2: 301, 3: 41, 4: 17}[M]
    valid = np.vstack([_grid_points(l_max, c_max, per_axis, cw), _domain_points(l_max, c_max, 60000, 1200, rng, cw)])
    expo_v = np.exp(-(valid @ mu))
    F_v = expo_v @ w.T                                                  # [n_valid, S]
    air = w.sum(axis=1)
    chosen = np.zeros(nE, dtype=bool)
    u_all = np.zeros((S, nE))
    worst = 0.0
    order = np.argsort(-np.count_nonzero(w, axis=1))                    # the widest spectrum first: the others reuse its nodes
    for s in order:
        cols = np.flatnonzero(w[s])
        pts = train
        u = None
        for _ in range(rounds):
            expo = np.exp(-(pts @ mu))
            F = expo @ w[s]
            live = F >= FLOOR * air[s]
            expo, F = expo[live], F[live]
            # a random positive cost makes the vertex unique; nodes other spectra already use are (almost) free
            cost = rng.uniform(0.5, 1.5, cols.size)
            cost[chosen[cols]] *= 1e-3
            u = _solve(expo, F, w[s], cols, tol, cost)
            if u is None:
                break
            approx = expo_v[:, cols] @ (u * w[s][cols])
            with np.errstate(divide='ignore', invalid='ignore'):
                err = np.where(F_v[:, s] >= FLOOR * air[s], np.abs(approx / F_v[:, s] - 1.0), 0.0)
            bad = np.flatnonzero(err > 2.0 * tol)
            if bad.size == 0:
                break
            bad = bad[np.argsort(-err[bad])[:400]]                      # cutting planes: the worst validation points join
            pts = np.vstack([pts, valid[bad]])
        if u is None:                                                   # infeasible / solver failure: keep the full grid
            return _remember(key, None)
        # (a programme that still misses 2 x tol somewhere after the last round may yet meet max_err: settled below)
        worst = max(worst, float(err.max()))
        u_all[s, cols] = u * w[s][cols]
        chosen |= u_all[s] > 0.0
    keep = np.flatnonzero(chosen)
    # the verdict comes from points the programme never saw: a finer grid and fresh random points
    fresh = np.vstack([_grid_points(l_max, c_max, per_axis + per_axis // 2 + 1, cw), _domain_points(l_max, c_max, 60000, 1777, rng, cw)])
    worst = max(worst, max_rel_error(mu, w, keep, u_all[:, keep], fresh))
    n_checked = int(valid.shape[0] + fresh.shape[0])
    if worst > max_err or keep.size * 4 > nE * 3:
        return _remember(key, None)
    w_red = u_all[:, keep]
    info = dict(nodes=int(keep.size), n_full=int(nE), nodes_per_spectrum=[int(np.count_nonzero(w_red[s])) for s in range(S)],
                max_rel_err=worst, n_validated=n_checked, tol=float(tol), l_max=[float(x) for x in l_max],
                c_max=float(c_max))
    return _remember(key, (keep, w_red, info))


def _remember(key, value):
    if len(_cache) >= 16:
        _cache.clear()
    _cache[key] = value
    return value


def max_rel_error(mu, w, cols, w_red, L):
    """The reduction's relative error at the path lengths ``L`` [n, M], in float64 (tests, tools)."""
    mu = np.asarray(mu, dtype=np.float64)
    expo = np.exp(-(np.asarray(L, dtype=np.float64) @ mu))
    w = np.asarray(w, dtype=np.float64)
    full = expo @ w.T
    red = expo[:, cols] @ np.asarray(w_red, dtype=np.float64).T
    with np.errstate(divide='ignore', invalid='ignore'):
        err = np.where(full >= FLOOR * w.sum(axis=1)[None, :], np.abs(red / full - 1.0), 0.0)
    return float(err.max())


GATE_VERSION = 9           # part of the key of the on-disk copy of a gate table (matdecomp._gate_cache_path): bump with any change here
START_HEADER = 12          # doubles before the tables (csrc/gn.hip, gn_start)
GATE_CELLS = 384           # cells per axis of the grid over (ln u0, u1 / u0).  The error of the kernel's 6 x 6 Lagrange interpolant of the fixed
                           # points goes with the sixth power of the cell size, and it is NOT uniform over the plane: at 256 cells (round 5)
                           # 2e-11 of |a| at the median point but 1.4e-9 at the median WATER ray - water in a tissue / bone basis sits at the edge
                           # of the physical ratios (a1 = -0.016 a0), where the fixed points bend - so that 7 % (round 5's Gauss-Newton step) to
                           # 40 % (round 6's chord step, held to 1e-12 per component) of a water scan's pixels took a second step; at 384 cells:
                           # 1.0e-10 at the median water ray, no second steps (profiles/r06_gn_chord.md).  7.1 MB instead of 3.2
GATE_U_MIN = 1.0e-4        # smallest u0 = ln(air_0 / g_0) / log_range of the grid: thinner rays walk from 1e-6 (a handful of steps)
GATE_U_MAX = 0.75          # largest u0 with open cells: attenuation exp(-12), six counts per million.  Beyond, the long walk from 1e-6 is
                           # fragile - photon-starved counts inside a cell whose corners all arrive have been seen to end at another
                           # root or to diverge (tools/soak_gn.py seeds 245, 283: 10 and 23 counts of 7e6) - and is left to the reference
GATE_MARGIN = 2            # steps added to the largest count seen around a cell
GATE_RADIUS = 0.01         # acceptance radius of a result around the interpolated fixed point, in units of the spread of the cell's
                           # corner fixed points.  The interpolant is within ~1e-6 of |a| of the pixel's own fixed point on the
                           # reference's branch (the spread is ~0.1 |a|), so this leaves a margin of a thousand.  What it keeps out, on
                           # tables far from anything physical (tools/soak_gn.py): a second root of the two equations 0.9 spreads away
                           # (seed 270), and a critical point of the likelihood that is no root at all - singular Jacobian, counts
                           # reproduced to 2e-3 only - 0.04 spreads away (seed 969); both reached by the coarse launch only


def newton_start_grid(i0, mus, log_range=16.0):
    """The grid of the Newton short cut's gate (csrc/gn.hip, gn_start; include/dexct.h, dexct_gn_options.start),
    laid out in DATA space: cells over (ln u0, u1 / u0), u_k = ln(air_k / g_k) / log_range, u0 from GATE_U_MIN to 1 and the ratio
    over what the forward model produces on the decomposition's domain (attenuation down to exp(-log_range), second component
    down to a quarter of its physical lower bound), widened by a quarter on both sides for noisy counts.  Returns a dict:
    ``head`` (the START_HEADER doubles) and ``corner_g`` [(n+1)^2, 2], the counts at the cell corners (row = index along ln u0)
    - what the reference's iteration is run on (matdecomp._device_tables) before ``assemble_start`` builds the tables -
    or None."""
    i0 = np.asarray(i0, dtype=np.float64)
    mus = np.asarray(mus, dtype=np.float64)
    if i0.ndim != 2 or mus.shape != (2, i0.shape[1]):
        return None
    used = np.any(i0 > 0.0, axis=0)
    air = i0.sum(axis=1)
    if not used.any() or np.any(air <= 0.0) or np.any(i0 < 0.0) or not np.all(np.isfinite(mus)):
        return None
    i0, mus = i0[:, used], mus[:, used]
    mu_min = mus.min(axis=1)
    if np.any(mu_min <= 0.0):
        return None
    # the ratio u1 / u0 over the domain: a0, a1 >= 0 plus rays whose second component is negative (water in a tissue / bone
    # basis: a1 = -0.008 a0; physically a0 mu0(E) + a1 mu1(E) >= 0 bounds it by -a0 min_E(mu0 / mu1); a quarter of that)
    skew = 0.25 * float(np.min(mus[0] / mus[1]))
    f = np.linspace(0.0, 1.0, 48)
    f0, f1 = np.meshgrid(f, f, indexing='ij')
    a = np.stack([f0.ravel() * log_range / mu_min[0], f1.ravel() * log_range / mu_min[1] - skew * f0.ravel() * log_range / mu_min[0]], axis=1)
    a = a[(a @ mu_min <= log_range) & (np.abs(a).sum(axis=1) > 0.0)]
    a = np.vstack([a, a * 1e-2, a * 1e-4])                           # thin rays: the ratio depends on the thickness (beam hardening)
    with np.errstate(over='ignore', invalid='ignore'):
        nu = np.exp(-(a @ mus)) @ i0.T
    u = np.log(air[None, :] / nu)
    ok = np.all(np.isfinite(u), axis=1) & (u[:, 0] > 0.0)
    if not ok.any():
        return None
    t = u[ok, 1] / u[ok, 0]
    t_lo, t_hi = float(t.min()), float(t.max())
    if not (np.isfinite(t_lo) and t_hi > t_lo):
        return None
    w = t_hi - t_lo
    t_lo, t_hi = t_lo - 0.25 * w, t_hi + 0.25 * w
    n = GATE_CELLS
    per_x = n / -np.log(GATE_U_MIN)
    per_t = n / (t_hi - t_lo)
    head = np.array([air[0], air[1], 1.0 / log_range, float(n), np.log(GATE_U_MIN), per_x, t_lo, per_t, np.log(air[0]), np.log(air[1]), 0.0, 0.0])
    x = np.log(GATE_U_MIN) + np.arange(n + 1) / per_x
    tt = t_lo + np.arange(n + 1) / per_t
    u0 = np.exp(x)[:, None] * np.ones((1, n + 1))
    u1 = u0 * tt[None, :]
    g = np.stack([air[0] * np.exp(-u0.ravel() * log_range), air[1] * np.exp(-u1.ravel() * log_range)], axis=1)
    return dict(head=head, corner_g=g, i0=i0, mus=mus)


GATE_MAX_COND = 1.0e4      # largest condition number of the forward model's log-Jacobian d ln nu_k / d a_m at a tabulated fixed point (physical tables: 15 - 140)


def _model_sums(pieces, a, third=False):
    """The energy sums of the forward model at the states a [n, 2] (with the reference's clip of the exponent, matdecomp.py:116):
    nu [n, k], G [n, k, m] = sum_e i0_k mu_m att over the energies whose exponent is not clipped (the clipped exponent has no
    slope), and with ``third`` S [n, k, m, p] = sum_e i0_k mu_m mu_p att - one exponential pass and one matrix product."""
    i0, mus = pieces['i0'], pieces['mus']
    a = np.where(np.isfinite(a), a, 0.0)
    dev = pieces.get('device')
    if dev is not None:
        # the same sums on the device the tables are calibrated for, by the library's own kernel (dexct_gn_model_sums: 0.2 s per
        # call in NumPy for the 1.5e5 states x 140 energies of a calibration, a millisecond there); they feed thresholds and
        # bounds, no bit of a result.  torch only carries the arrays (round 6 first ran these sums as torch float64 kernels: the
        # GPU suite then aborted once in a few runs, from a thread of the runtime, inside exactly these passes).
        import torch
        from . import _native
        from ._device import ptr, stream_ptr
        lib = _native.load()
        t = lambda x: torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64), device=dev)
        a_t, i0_t, mus_t = t(a), t(i0), t(mus)
        n = a_t.shape[0]
        nu = torch.empty((n, 2), dtype=torch.float64, device=dev)
        G = torch.empty((n, 2, 2), dtype=torch.float64, device=dev)
        S = torch.empty((n, 2, 2, 2), dtype=torch.float64, device=dev) if third else None
        if n:
            _native.check(lib.dexct_gn_model_sums(ptr(a_t), n, ptr(i0_t), ptr(mus_t), i0.shape[1], ptr(nu), ptr(G),
                                                  ptr(S) if third else None, stream_ptr()), 'dexct_gn_model_sums')
        return nu.cpu().numpy(), G.cpu().numpy(), (S.cpu().numpy() if third else None)
    expo = -(a @ mus)
    att = np.exp(np.clip(expo, -700.0, 700.0))
    nu = att @ i0.T
    live = att * (np.abs(expo) < 700.0)
    w1 = (i0[:, None, :] * mus[None, :, :]).reshape(4, -1)                               # (k, m)
    G = (live @ w1.T).reshape(-1, 2, 2)
    if not third:
        return nu, G, None
    w2 = (i0[:, None, None, :] * mus[None, :, None, :] * mus[None, None, :, :]).reshape(8, -1)      # (k, m, p)
    return nu, G, (live @ w2.T).reshape(-1, 2, 2, 2)


def _counts_and_condition(pieces, a, g, sums=None):
    """For fixed points a [n, 2] and the counts g [n, 2] they belong to: how well the forward model (with the reference's clip of
    the exponent, matdecomp.py:116) reproduces the counts, and the condition number of its log-Jacobian there - an ISOLATED root
    of the two equations has a small one; where tables far from anything physical (attenuation x 30 at the low energies: clipped
    exponents) make the equations dependent, roots come in families and a start value next to one rests next to it, not on it."""
    with np.errstate(all='ignore'):
        nu, G, _ = sums if sums is not None else _model_sums(pieces, a)
        jac = -G / nu[:, :, None]
        resid = np.abs(nu / g - 1.0).max(axis=1)
        fro2 = (jac ** 2).sum(axis=(1, 2))
        det = np.abs(jac[:, 0, 0] * jac[:, 1, 1] - jac[:, 0, 1] * jac[:, 1, 0])
        cond = (fro2 + np.sqrt(np.maximum(fro2 * fro2 - 4.0 * det * det, 0.0))) / (2.0 * det)       # sigma_max / sigma_min of a 2 x 2
    return resid, np.where(np.isfinite(cond), cond, np.inf)


def assemble_start(pieces, steps, roots):
    """The start array (csrc/gn.hip, gn_start) from the reference iteration run on the cell corners (by the library's own
    kernel: matdecomp._device_tables): ``steps`` [(n+1)^2] = steps after which the tolerance rule ended the corner's pixel
    (255: it did not), ``roots`` [(n+1)^2, 2] = where.  A corner counts only if its fixed point REPRODUCES its counts (the
    forward model with the reference's clip of the exponent, matdecomp.py:116, within 1e-8) and the model's log-Jacobian is
    well conditioned there (GATE_MAX_COND): an isolated root of the two equations.  Where the clip is active (attenuation tables far above anything physical) or the gradient vanishes for another
    reason the iteration also comes to rest - on a valley of the clipped likelihood, at points that are not isolated: another
    start value rests elsewhere on it, so there is nothing to tabulate.  A cell is open when all four corners count, at finite
    fixed points that vary smoothly over it and along the grid lines through its corners (mixed and axial second differences at
    most half the largest first difference: no boundary between two basins crosses or borders it); it needs the largest step count among its own corners and those of the eight cells around
    it - and of the sixteen around those: the 5 x 5 cells whose corners the kernel interpolates over - plus GATE_MARGIN (infinity if any of those cells is closed); its acceptance radius is GATE_RADIUS x the spread of its
    corners' fixed points; kappa (the one-step acceptance, newton_kappa) follows the cells.  Returns the array, the share of open
    cells, and what the walk did on the corner grid (``stats``, for pair_is_ill_posed)."""
    n = int(pieces['head'][3])
    steps = np.asarray(steps).reshape(n + 1, n + 1).astype(np.float64)
    r = np.asarray(roots, dtype=np.float64).reshape(n + 1, n + 1, 2)
    finite = np.all(np.isfinite(r), axis=2)
    good = (steps < 255) & finite & np.all(np.isfinite(pieces['corner_g']), axis=1).reshape(n + 1, n + 1)
    with np.errstate(all='ignore'):
        sums = _model_sums(pieces, r.reshape(-1, 2), third=True)
    resid, cond = _counts_and_condition(pieces, r.reshape(-1, 2), pieces['corner_g'], sums)
    # what the reference's walk did on the part of the grid a detector can deliver (attenuation up to exp(-12), GATE_U_MAX)
    x_c = pieces['head'][4] + np.arange(n + 1) / pieces['head'][5]
    dom = np.broadcast_to((x_c <= np.log(GATE_U_MAX))[:, None], (n + 1, n + 1))
    root_ok = good & (resid.reshape(n + 1, n + 1) <= 1.0e-8)
    cg = cond.reshape(n + 1, n + 1)[root_ok & dom]
    stats = {'corners': int(dom.sum()), 'walk_nonfinite_share': float((~finite & dom).sum() / dom.sum()),
             'walk_not_by_rule_share': float(((steps >= 255) & finite & dom).sum() / dom.sum()),
             'not_a_root_share': float((good & dom & ~root_ok).sum() / dom.sum()),
             'cond_median': float(np.median(cg)) if cg.size else float('inf'),
             'cond_p90': float(np.percentile(cg, 90)) if cg.size else float('inf')}
    good &= (resid.reshape(n + 1, n + 1) <= 1.0e-8) & (cond.reshape(n + 1, n + 1) <= GATE_MAX_COND)
    c00, c01, c10, c11 = r[:-1, :-1], r[:-1, 1:], r[1:, :-1], r[1:, 1:]
    with np.errstate(invalid='ignore'):
        edges = np.max([np.abs(c01 - c00), np.abs(c11 - c10), np.abs(c10 - c00), np.abs(c11 - c01)], axis=0).max(axis=2)
        twist = np.abs(c00 + c11 - c01 - c10).max(axis=2)
        spread = np.max([np.abs(c01 - c00), np.abs(c10 - c00), np.abs(c11 - c00), np.abs(c11 - c01), np.abs(c11 - c10),
                         np.abs(c10 - c01)], axis=0).max(axis=2)
    # ... nor runs along a grid line: along either axis the step from a corner to its neighbour may not differ from the step
    # before by more than half the larger of the two (smooth fields change by a few per cent from cell to cell)
    kink = np.zeros((n + 1, n + 1), dtype=bool)
    with np.errstate(invalid='ignore'):
        for ax in (0, 1):
            rm = np.moveaxis(r, ax, 0)
            d0, d1 = rm[1:-1] - rm[:-2], rm[2:] - rm[1:-1]
            k = np.abs(d1 - d0).max(axis=-1) > 0.5 * np.maximum(np.abs(d1).max(axis=-1), np.abs(d0).max(axis=-1)) + 1e-12
            gm = np.moveaxis(good, ax, 0)
            k &= gm[1:-1] & gm[:-2] & gm[2:]                 # (corners that do not count close their cells anyway)
            np.moveaxis(kink, ax, 0)[1:-1] |= k
    good = good & ~kink
    cell_ok = good[:-1, :-1] & good[:-1, 1:] & good[1:, :-1] & good[1:, 1:] & (twist <= 0.5 * edges + 1e-12)
    k = np.max([steps[:-1, :-1], steps[:-1, 1:], steps[1:, :-1], steps[1:, 1:]], axis=0)
    k = np.where(cell_ok, k, np.inf)
    pad = np.pad(k, 2, mode='edge')
    need = np.max([pad[2 + di:n + 2 + di, 2 + dj:n + 2 + dj] for di in (-2, -1, 0, 1, 2) for dj in (-2, -1, 0, 1, 2)], axis=0) + GATE_MARGIN
    # the kernel interpolates the fixed points over the 6 x 6 corners around a cell (the corners of the 5 x 5 cells above): the
    # cells within two of the border of the grid, which lack the rings of neighbours, are closed; so are the cells of
    # photon-starved counts (GATE_U_MAX)
    need[:2, :] = need[-2:, :] = need[:, :2] = need[:, -2:] = np.inf
    x_hi = pieces['head'][4] + (np.arange(n) + 1.0) / pieces['head'][5]            # ln u0 at the upper edge of each cell row
    need[x_hi > np.log(GATE_U_MAX), :] = np.inf
    radius = np.where(cell_ok, GATE_RADIUS * spread + 1e-9, 0.0)
    # the one-step acceptance (csrc/gn.hip: the chord step, whose inverse Jacobian is the gradient of this table): per cell kappa -
    # KAPPA_SAFETY x the largest value at the corners of the cell and of the eight around it - and eps, which validate_start
    # measures at the cells' corners and centres (infinity until then: a table that was not validated takes two steps everywhere)
    _, kc = chord_tables(pieces, r.reshape(-1, 2), sums)
    kc = np.where(good, kc.reshape(n + 1, n + 1), np.inf)
    kcell = np.maximum.reduce([kc[:-1, :-1], kc[:-1, 1:], kc[1:, :-1], kc[1:, 1:]])
    padk = np.pad(kcell, 1, mode='edge')
    kappa = KAPPA_SAFETY * np.max([padk[1 + di:n + 1 + di, 1 + dj:n + 1 + dj] for di in (-1, 0, 1) for dj in (-1, 0, 1)], axis=0)
    kappa = np.where(np.isfinite(need), kappa, np.inf)
    stats['one_step_share'] = float(np.isfinite(kappa).mean())
    r = np.where(good[:, :, None], r, 0.0)
    head = pieces['head'].copy()
    head[10] = 2.0                        # the pairs of the one-step acceptance follow the cells
    # pairs (a0, a1), pairs (need, radius), pairs (kappa, eps)
    out = np.concatenate([head, r.ravel(), np.stack([need, radius], axis=-1).ravel(),
                          np.stack([kappa, np.full_like(kappa, np.inf)], axis=-1).ravel()])
    return out, float(np.isfinite(need).mean()), stats


def start_layout(n):
    """Offsets (in doubles) of the parts of a start array with n cells per axis: roots, cells, one-step pairs, end."""
    c0 = START_HEADER + 2 * (n + 1) ** 2
    return START_HEADER, c0, c0 + 2 * n * n, c0 + 4 * n * n


# A pair of spectra is ILL-POSED when it does not determine two thicknesses - MV against kV: above a few hundred keV both basis
# materials attenuate by Compton scattering alone, their curves are parallel.  The calibration sees it in the reference's own
# iteration, run on noise-free counts of its own forward model over the part of the data plane a detector can deliver
# (profiles/r06_pair_classes.log - every pair of the bundled spectra, the benchmark's Kramers pair, the three golden cases):
#                                 kV / kV (6 cases)      kV / MV (7 cases)      6MV / detunedMV
#   open cells                    0.857 .. 0.952         0.24 .. 0.73           0.45
#   walk rests where it does not  0.00006 .. 0.0016      0.17 .. 0.54           0.011
#     reproduce its counts
#   median cond of the log-       22 .. 69               31 .. 78               3667
#     Jacobian at the roots
# (the share of corners where the walk ends non-finite, 0.1 - 26 %, does not tell the classes apart: those are the corners in
# the margin the grid adds around the physical ratios).  Ill-posed = the walk comes to rest at points that are no solutions on
# more than ILL_POSED_NOT_A_ROOT of the plane, or fewer than ILL_POSED_OPEN of the cells are open, or the roots themselves
# are ill-conditioned.  For such a pair the library runs the reference's fixed count (matdecomp.gn_device): its walk wanders on
# noisy data, the tolerance rule is no safer there than anywhere on a wandering sequence (the only divergence from the exact
# count ever seen, profiles/r04_gn_noisy_public.log), and the short cut saved 15 % there.  A false positive costs speed only.
ILL_POSED_NOT_A_ROOT = 0.05
ILL_POSED_OPEN = 0.78
ILL_POSED_COND = 1.0e3


def pair_is_ill_posed(stats):
    """``stats`` of assemble_start + validate_start (via matdecomp.calibrate_gate)."""
    if not stats or not stats.get('grid', True) or 'not_a_root_share' not in stats:
        return False
    return bool(stats['not_a_root_share'] > ILL_POSED_NOT_A_ROOT or stats.get('open_share', 1.0) < ILL_POSED_OPEN
                or not stats['cond_median'] <= ILL_POSED_COND)


def cell_centres(pieces):
    """The counts at the centres of the grid's cells [n^2, 2] (row = index along ln u0): what validate_start is given the
    reference's walk on."""
    h = pieces['head']
    n = int(h[3])
    x = h[4] + (np.arange(n) + 0.5) / h[5]
    t = h[6] + (np.arange(n) + 0.5) / h[7]
    u0 = np.exp(x)[:, None] * np.ones((1, n))
    u1 = u0 * t[None, :]
    return np.stack([h[0] * np.exp(-u0.ravel() / h[2]), h[1] * np.exp(-u1.ravel() / h[2])], axis=1)


def lagrange6(t):
    """Weights of the 6-point Lagrange interpolation on the nodes -2 .. 3 at t (what csrc/gn.hip gn_start forms per axis)."""
    nodes = np.arange(-2.0, 4.0)
    w = np.ones(6)
    for a in range(6):
        for b in range(6):
            if a != b:
                w[a] *= (t - nodes[b]) / (nodes[a] - nodes[b])
    return w


def centre_interpolant(start, n):
    """The 6 x 6 Lagrange interpolant of the tabulated fixed points at the centre of every cell that has its two rings of
    neighbours [n, n, 2] (what the kernel starts a pixel from there; the others: 0)."""
    r = np.asarray(start)[START_HEADER:START_HEADER + 2 * (n + 1) ** 2].reshape(n + 1, n + 1, 2)
    w = lagrange6(0.5)                                               # [3, -25, 150, 150, -25, 3] / 256
    s = np.zeros((n, n, 2))
    for p in range(6):
        for q_ in range(6):
            s[2:-2, 2:-2] += w[p] * w[q_] * r[p:n - 4 + p, q_:n - 4 + q_]     # corner (i + p - 2, j + q - 2) of cell (i, j)
    return s


EPS_SAFETY = 4.0           # on the largest |I - Bt L| measured at the corners and the centre of a cell and of the eight around it (the
                           # derivative of an even-order interpolant is worst at the ends of the middle interval, i.e. at the corners)


def chord_tables(pieces, roots, sums=None):
    """What the chord step of the short cut (csrc/gn.hip, chord_residuals_f64 / gn_binv) needs at the fixed points ``roots`` [n, 2]
    of the counts they reproduce: B [n, 2, 2] = L^-1, the inverse of the model's log-Jacobian L_kp = d ln nu_k / d a_p = -G_kp /
    nu_k, and the constant of what the step leaves beyond the Jacobian's own error: with c_k(a) = g_k / nu_k(a) - 1 (root:
    c = 0, Dc = -L) the step m = s + B c(s) lands at m - a* = (I - B L*) e0 + 1/2 B D2c(xi) [e0, e0], and at a root
    D2c_k,pq = 2 G_kp G_kq / nu_k^2 - S_kpq / nu_k; kappa = 1/2 max_i sum_k |B_ik| sum_pq (2 G_kp G_kq / nu_k^2 + S_kpq / nu_k)
    (the two terms bounded separately: they partly cancel at the root, not necessarily next to it).  inf where L is singular."""
    with np.errstate(all='ignore'):
        nu, G, S = sums if sums is not None else _model_sums(pieces, roots, third=True)
        inv = 1.0 / nu
        L = -G * inv[:, :, None]                                        # [n, k, p]
        det = L[:, 0, 0] * L[:, 1, 1] - L[:, 0, 1] * L[:, 1, 0]
        B = np.stack([np.stack([L[:, 1, 1], -L[:, 0, 1]], -1), np.stack([-L[:, 1, 0], L[:, 0, 0]], -1)], -2) / det[:, None, None]   # [n, p, k]
        d2 = (2.0 * inv ** 2)[:, :, None, None] * np.abs(G[:, :, :, None] * G[:, :, None, :]) + np.abs(inv)[:, :, None, None] * np.abs(S)
        per_k = 0.5 * d2.sum(axis=(2, 3))                               # [n, k]
        kap = np.einsum('npk,nk->np', np.abs(B), per_k).max(axis=1)
    ok = np.isfinite(kap) & np.all(np.isfinite(roots), axis=1) & np.isfinite(B).all(axis=(1, 2))
    return np.where(ok[:, None, None], B, np.nan), np.where(ok, kap, np.inf)


def dlagrange6(t):
    """d/dt of lagrange6's weights (the derivative weights csrc/gn.hip gn_start<DERIV> forms by the product rule)."""
    nodes = np.arange(-2.0, 4.0)
    dw = np.zeros(6)
    for a in range(6):
        for c in range(6):
            if c == a:
                continue
            term = 1.0 / (nodes[a] - nodes[c])
            for b in range(6):
                if b != a and b != c:
                    term *= (t - nodes[b]) / (nodes[a] - nodes[b])
            dw[a] += term
    return dw


def table_gradient(start, pieces, tx, ty):
    """The inverse log-Jacobian the chord step of the kernel uses at the place (tx, ty) in [0, 1]^2 of every cell that has its two
    rings of neighbours [n, n, 2 (p), 2 (k)] (nan elsewhere): the gradient of the 6 x 6 Lagrange interpolant of the tabulated fixed
    points along (x = ln u0, t = u1 / u0), times d (x, t) / d ln g (csrc/gn.hip, gn_start<DERIV>):
    B_p0 = Bx_p - t Bt_p, B_p1 = Bt_p with Bx = d a / d x * (-1 / (log_range u0)), Bt = d a / d t * (-1 / (log_range u0))."""
    h = pieces['head']
    n = int(h[3])
    r = np.asarray(start)[START_HEADER:START_HEADER + 2 * (n + 1) ** 2].reshape(n + 1, n + 1, 2)
    wx, wy, dx, dy = lagrange6(tx), lagrange6(ty), dlagrange6(tx), dlagrange6(ty)
    ax = np.zeros((n - 4, n - 4, 2))
    at = np.zeros((n - 4, n - 4, 2))
    for p in range(6):
        for q_ in range(6):
            blk = r[p:n - 4 + p, q_:n - 4 + q_]                        # corner (i + p - 2, j + q - 2) of cell (i, j), i, j = 2 .. n - 3
            ax += dx[p] * wy[q_] * blk
            at += wx[p] * dy[q_] * blk
    u0 = np.exp(h[4] + (np.arange(2, n - 2) + tx) / h[5])[:, None, None]
    t = (h[6] + (np.arange(2, n - 2) + ty) / h[7])[None, :, None]
    bx = ax * (-(h[5] * h[2]) / u0)
    bt = at * (-(h[7] * h[2]) / u0)
    out = np.full((n, n, 2, 2), np.nan)
    out[2:-2, 2:-2, :, 0] = bx - t * bt
    out[2:-2, 2:-2, :, 1] = bt
    return out


KAPPA_SAFETY = 2.5         # on the largest kappa at the corners of a cell and of the eight around it: covers its variation across the
                           # cells (a few per cent) and the factor (1 - kappa e0)^-2 <= 1.25 between e0^2 and the measured d1^2


def newton_kappa(pieces, roots, sums=None):
    """The contraction constant of Newton's iteration on the Poisson likelihood F(a) = sum_k nu_k(a) - g_k ln nu_k(a) at the fixed
    points ``roots`` [n, 2] of the counts they reproduce (g_k = nu_k there): a step from a0 lands at a1 with
    a1 - a* = 1/2 H^-1 D3F [e0, e0], hence |e1| <= kappa |e0|^2 in the max norm with
    kappa = 1/2 max_i sum_j |H^-1_ij| sum_pq |D3F_jpq|.  With G_km = sum_e i0_k mu_m att, S_kmp = sum_e i0_k mu_m mu_p att:
    H_mp = sum_k G_km G_kp / nu_k and D3F_mpq = sum_k [2 G_km G_kp G_kq / nu_k^2 - (S_kpq G_km + S_kmq G_kp + S_kmp G_kq) / nu_k]
    (the terms with g_k / nu_k - 1 vanish at a root that reproduces its counts).  inf where H is singular or anything overflows.
    (Round 5's one-step acceptance was built on this constant; round 6's chord step carries its own, chord_tables - this one
    stays as the yardstick: what a FULL Newton step leaves, tests/test_quadrature.py.)"""
    with np.errstate(all='ignore'):
        nu, G, S = sums if sums is not None else _model_sums(pieces, roots, third=True)
        # (broadcast products over [n, k, m, p, q] summed over k: the same sums as the einsum forms in the docstring, 15 x faster)
        inv = 1.0 / nu
        Gm, Gp, Gq = G[:, :, :, None, None], G[:, :, None, :, None], G[:, :, None, None, :]
        H = (inv[:, :, None, None] * G[:, :, :, None] * G[:, :, None, :]).sum(axis=1)
        T = ((2.0 * inv ** 2)[:, :, None, None, None] * Gm * Gp * Gq
             - inv[:, :, None, None, None] * (S[:, :, None, :, :] * Gm + S[:, :, :, None, :] * Gp + S[:, :, :, :, None] * Gq)).sum(axis=1)
        det = H[:, 0, 0] * H[:, 1, 1] - H[:, 0, 1] * H[:, 1, 0]
        Hinv = np.stack([np.stack([H[:, 1, 1], -H[:, 0, 1]], -1), np.stack([-H[:, 1, 0], H[:, 0, 0]], -1)], -2) / det[:, None, None]
        per_row = 0.5 * np.abs(T).sum(axis=(2, 3))
        kap = np.einsum('nij,nj->ni', np.abs(Hinv), per_row).max(axis=1)
    return np.where(np.isfinite(kap) & np.all(np.isfinite(roots), axis=1), kap, np.inf)


def validate_start(start, pieces, steps, roots):
    """The table checked against the thing it stands for, at one interior point per cell: the reference's walk run on the counts
    at the cell CENTRES (``steps``, ``roots`` as in assemble_start, n^2 of them).  An open cell stays open only if that walk
    ended by the rule within the cell's step budget (need - 1) at a fixed point that reproduces the centre's counts and lies
    within the cell's acceptance radius of the interpolant the kernel would start from; a cell that fails is closed
    together with the two rings of cells around it (every cell whose interpolation uses its corners).  Returns the array and the share of open cells."""
    h = pieces['head']
    n = int(h[3])
    out = np.array(start, dtype=np.float64, copy=True)
    _, c0, k0, k1 = start_layout(n)
    cells = out[c0:k0].reshape(n, n, 2)
    onestep = out[k0:k1].reshape(n, n, 2)                              # (kappa, eps)
    kappa = onestep[:, :, 0]
    steps = np.asarray(steps, dtype=np.float64).reshape(n, n)
    rc = np.asarray(roots, dtype=np.float64).reshape(n, n, 2)
    s = centre_interpolant(out, n)
    g = cell_centres(pieces)
    with np.errstate(all='ignore'):
        sums_c = _model_sums(pieces, rc.reshape(-1, 2))
    resid, cond = _counts_and_condition(pieces, rc.reshape(-1, 2), g, sums_c)
    # eps of the chord step: the gradient of the table as the kernel forms it, against the exact log-Jacobian L = -G / nu of the
    # model at the cell's four corners (their tabulated roots) and at its centre (the centre's own root): |I - Bt L| (max row
    # sum), EPS_SAFETY x the largest of a cell and the eight around it
    def misfit(B, nu_, G_):
        # (component by component: reductions over trailing axes of length 2 cost NumPy 20 ms per call at 1.5e5 cells)
        with np.errstate(all='ignore'):
            L = -G_ / nu_[..., None]                                                                 # [.., k, p]
            e = [[(1.0 if p_ == q_ else 0.0) - (B[..., p_, 0] * L[..., 0, q_] + B[..., p_, 1] * L[..., 1, q_]) for q_ in (0, 1)]
                 for p_ in (0, 1)]
            m_ = np.maximum(np.abs(e[0][0]) + np.abs(e[0][1]), np.abs(e[1][0]) + np.abs(e[1][1]))
        return np.where(np.isfinite(m_), m_, np.inf)
    nu_c, G_c, _ = sums_c
    eps_c = misfit(table_gradient(out, pieces, 0.5, 0.5), nu_c.reshape(n, n, 2), G_c.reshape(n, n, 2, 2))
    with np.errstate(all='ignore'):
        nu_k, G_k, _ = _model_sums(pieces, out[START_HEADER:c0].reshape(-1, 2))
    nu_k, G_k = nu_k.reshape(n + 1, n + 1, 2), G_k.reshape(n + 1, n + 1, 2, 2)
    for tx_, ty_ in ((0.0, 0.0), (0.0, 1.0), (1.0, 0.0), (1.0, 1.0)):
        di, dj = int(tx_), int(ty_)
        eps_c = np.maximum(eps_c, misfit(table_gradient(out, pieces, tx_, ty_), nu_k[di:n + di, dj:n + dj], G_k[di:n + di, dj:n + dj]))
    pade = np.pad(eps_c, 1, mode='edge')
    onestep[:, :, 1] = EPS_SAFETY * np.max([pade[1 + di:n + 1 + di, 1 + dj:n + 1 + dj] for di in (-1, 0, 1) for dj in (-1, 0, 1)], axis=0)
    with np.errstate(all='ignore'):
        off = np.abs(rc - s).max(axis=2)
        fine = ((steps < 255) & np.all(np.isfinite(rc), axis=2) & (resid.reshape(n, n) <= 1.0e-8) & (cond.reshape(n, n) <= GATE_MAX_COND)
                & (off <= cells[:, :, 1]) & (steps <= cells[:, :, 0] - 1.0))
    bad = np.isfinite(cells[:, :, 0]) & ~fine
    pad = np.pad(bad, 2, mode='constant')
    near = np.any([pad[2 + di:n + 2 + di, 2 + dj:n + 2 + dj] for di in (-2, -1, 0, 1, 2) for dj in (-2, -1, 0, 1, 2)], axis=0)
    cells[near, 0] = np.inf
    kappa[near] = np.inf
    onestep[~np.isfinite(kappa), 1] = np.inf
    kappa[~np.isfinite(onestep[:, :, 1])] = np.inf
    return out, float(np.isfinite(cells[:, :, 0]).mean()), int(bad.sum())
