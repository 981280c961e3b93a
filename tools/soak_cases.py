"""The random cases of tools/soak_gn.py (tables, measurements, iteration count), shared with the probes that look at one case
closely (tools/probes/gn_soak_traj.py): one generator, the same draws."""
import numpy as np


def draw(seed):
    """-> dict(rng, n_e, i0 [2, n_e], mus [2, n_e], n_v, n_c, a_true [n_v, n_c, 2], g [2, n_v, n_c] float64, kind, n_iters);
    ``rng`` is left where soak_gn.py continues drawing (the permutation of its stability screen)."""
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
    kind = str(rng.choice(['clean', 'noisy', 'poisson', 'float32']))
    if kind == 'noisy':
        g = g * (1 + 10.0 ** rng.uniform(-6, -1) * rng.standard_normal(g.shape))
    elif kind == 'poisson':
        g = rng.poisson(np.minimum(g, 1e15)).astype(np.float64)
    weird = rng.random(g.shape) < 0.01                          # pathological measurements mixed in
    g[weird] = rng.choice([0.0, -1.0, np.inf, np.nan, 1e-300, 1e300], int(weird.sum()))
    if kind == 'float32':
        g = g.astype(np.float32).astype(np.float64)
    n_iters = int(rng.choice([0, 1, 2, 5, 9, 30, 50, 50, 50, 61, 80]))
    return dict(rng=rng, n_e=n_e, i0=i0, mus=mus, n_v=n_v, n_c=n_c, a_true=a_true, g=g, kind=kind, n_iters=n_iters)


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
