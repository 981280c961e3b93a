"""ORACLE (test infrastructure): equiangular fan-beam filtered back-projection in NumPy float64.

PARITY UNPINNED: the reference's ``get_recon`` (main.py:134,168) lives in the un-vendored
x-tomo-sim submodule (``xtomosim/back_project.py``, "fan-beam filtered back projection with a sinc
window filter", README.md:30-31).  Restated here from the textbook it implements - Kak & Slaney,
"Principles of Computerized Tomographic Imaging", section 3.4.1 (equiangular rays):

    R'(beta, n dg) = R(beta, n dg) * D * cos(n dg)                         (weighting, D = SID)
    Q(beta, n dg)  = dg * sum_m R'(beta, m dg) g((n - m) dg)               (filtering)
    g(n dg)        = 1/2 (n dg / sin(n dg))^2 h(n dg)
    h(t)           = 2 fc^2 sinc(2 fc t) - fc^2 sinc^2(fc t),  fc = ramp / (2 dg)   (band-limited ramp)
    f(x, y)        = dbeta * sum_beta Q(beta, gamma'(x, y)) / L^2(x, y, beta)     (back-projection, 2 pi scan)

with this build's geometry (source at SID (cos b, sin b), channel angle measured from the central
ray, image pixel (ix, iy) at ((ix - N/2 + 1/2) FOV/N, (iy - N/2 + 1/2) FOV/N), linear interpolation
between channels).  Pins: reconstruction of analytically projected discs (tests/test_fbp_oracle.py).
"""
import numpy as np


WINDOWS = {          # W(x), x = f / f_cutoff: apodisation of the ramp (Shepp-Logan "sinc", cosine, Hann, Hamming)
    'sinc': lambda x: np.sinc(0.5 * x),
    'cosine': lambda x: np.cos(0.5 * np.pi * x),
    'hann': lambda x: 0.5 * (1.0 + np.cos(np.pi * x)),
    'hamming': lambda x: 0.54 + 0.46 * np.cos(np.pi * x),
}


def windowed_ramp(t, fc, window):
    """h(t) = 2 int_0^fc f W(f / fc) cos(2 pi f t) df by QUADPACK's oscillatory rule (scipy quad, weight='cos'),
    one lag at a time - an algorithm unrelated to the product's Gauss-Legendre panels."""
    from scipy.integrate import quad
    W = WINDOWS[window]
    out = np.empty(len(t))
    for i, ti in enumerate(np.asarray(t, dtype=np.float64)):
        if ti == 0.0:
            out[i] = 2.0 * quad(lambda f: f * W(f / fc), 0.0, fc, epsabs=0, epsrel=1e-13, limit=400)[0]
        else:
            out[i] = 2.0 * quad(lambda f: f * W(f / fc), 0.0, fc, weight='cos', wvar=2.0 * np.pi * abs(ti),
                                epsabs=1e-9 * fc * fc, epsrel=1e-12, limit=400)[0]
    return out


def ramp_taps(n_channels, dgamma, ramp=1.0, window='rect'):
    """g((n) dg) for n = -(N-1) .. (N-1): the equiangular filter with cutoff ramp * Nyquist."""
    n = np.arange(-(n_channels - 1), n_channels, dtype=np.float64)
    c = float(ramp)
    if window == 'rect':
        h = (c * c / (2 * dgamma ** 2)) * np.sinc(c * n) - (c * c / (4 * dgamma ** 2)) * np.sinc(c * n / 2) ** 2
    else:
        pos = windowed_ramp(n[n_channels - 1:] * dgamma, c / (2.0 * dgamma), window)
        h = np.concatenate([pos[:0:-1], pos])
    t = n * dgamma
    with np.errstate(invalid='ignore', divide='ignore'):
        ratio = np.where(n == 0, 1.0, t / np.sin(t))
    return 0.5 * ratio ** 2 * h


def filter_sino(sino, gammas, sid, ramp=1.0, window='rect'):
    """sino [..., N_channels] line integrals -> Q [..., N_channels]."""
    sino = np.asarray(sino, dtype=np.float64)
    n = sino.shape[-1]
    dg = float(gammas[1] - gammas[0])
    g = ramp_taps(n, dg, ramp, window)
    rp = sino * (sid * np.cos(gammas))
    idx = np.arange(n)[:, None] - np.arange(n)[None, :] + (n - 1)      # taps[n - m]
    return dg * np.einsum('...m,nm->...n', rp, g[idx])


def back_project(q, thetas, gammas, sid, n_matrix, fov):
    """q [N_proj, N_channels] -> image [n_matrix, n_matrix] (index [iy, ix]) in 1/cm."""
    q = np.asarray(q, dtype=np.float64)
    n_views, n_ch = q.shape
    dg = float(gammas[1] - gammas[0])
    dbeta = 2 * np.pi / n_views if n_views > 1 else 2 * np.pi
    if n_views > 1:
        dbeta = float(thetas[1] - thetas[0])
    c = (np.arange(n_matrix) - n_matrix / 2 + 0.5) * (fov / n_matrix)
    x, y = np.meshgrid(c, c)             # x varies along the last axis
    img = np.zeros((n_matrix, n_matrix))
    for i in range(n_views):
        cb, sb = np.cos(thetas[i]), np.sin(thetas[i])
        dx, dy = x - sid * cb, y - sid * sb
        c0x, c0y = -cb, -sb
        dot = c0x * dx + c0y * dy
        cross = c0x * dy - c0y * dx
        gam = np.arctan2(cross, dot)
        pos = gam / dg + 0.5 * (n_ch - 1)
        k = np.floor(pos).astype(np.int64)
        w = pos - k
        ok = (k >= 0) & (k < n_ch - 1)
        kk = np.clip(k, 0, n_ch - 2)
        val = (1 - w) * q[i, kk] + w * q[i, kk + 1]
        img += np.where(ok, val / (dx * dx + dy * dy), 0.0)
    return img * dbeta


def parker_weights(thetas, gammas, theta_tot):
    """Short-scan weights w[view, channel] (Parker, Med. Phys. 9, 254 (1982); Silver's virtual fan angle
    G = (theta_tot - pi) / 2 for scans longer than pi + fan), for this build's geometry, in which ray (beta, gamma) is
    measured a second time as (beta + pi + 2 gamma, -gamma).  The weights of such a pair add up to 1."""
    b = np.asarray(thetas, dtype=np.float64)[:, None]
    g = np.asarray(gammas, dtype=np.float64)[None, :]
    G = 0.5 * (theta_tot - np.pi)
    if G < np.max(np.abs(g)) * (1 - 1e-12):
        raise ValueError('less than a short scan')
    w = np.ones(np.broadcast(b, g).shape)
    early = b < 2.0 * (G - g)
    late = b > np.pi - 2.0 * g
    with np.errstate(divide='ignore', invalid='ignore'):
        w = np.where(early, np.sin(0.25 * np.pi * b / (G - g)) ** 2, w)
        w = np.where(late & ~early, np.sin(0.25 * np.pi * (np.pi + 2.0 * G - b) / (G + g)) ** 2, w)
    return w


def get_recon(sino_log, thetas, gammas, sid, n_matrix, fov, ramp, mu_water=None, window='rect', theta_tot=None):
    """theta_tot: None or 2 pi = a full rotation; less: a short scan, weighted by 2 * parker_weights first (the
    full-scan formula counts every ray twice)."""
    sino_log = np.asarray(sino_log, dtype=np.float64)
    if theta_tot is not None and theta_tot < 2 * np.pi - 1e-9:
        w = 2.0 * parker_weights(thetas, gammas, theta_tot)
        sino_log = sino_log * (w if sino_log.ndim == 2 else w[:, None, :])
    raw = back_project(filter_sino(sino_log, gammas, sid, ramp, window), thetas, gammas, sid, n_matrix, fov)
    hu = None if mu_water is None else 1000.0 * (raw - mu_water) / mu_water
    return raw, hu


def fdk_recon(sino, thetas, gammas, sid, sdd, row_z, src_z, n_matrix, fov, ramp, slices_z, window='rect'):
    """Feldkamp reconstruction for the cylindrical detector of the cone-beam projector (rows at heights row_z on the
    cylinder of radius sdd around the source axis, source at height src_z): the fan algorithm above with the
    projections weighted by cos(kappa_r) = sdd / sqrt(sdd^2 + (row_z[r] - src_z)^2) and voxel (x, y, z) reading
    the detector at height src_z + (z - src_z) * sdd / L, L the in-plane source-voxel distance (Feldkamp, Davis,
    Kress 1984, adapted to equiangular rays as in Kak & Slaney 3.6).  sino [N_proj, N_rows, N_channels] ->
    [len(slices_z), n_matrix, n_matrix].  PARITY UNPINNED (the reference has no cone beam)."""
    sino = np.asarray(sino, dtype=np.float64)
    n_views, n_rows, n_ch = sino.shape
    row_z = np.asarray(row_z, dtype=np.float64)
    q = filter_sino(sino, gammas, sid, ramp, window) * (sdd / np.sqrt(sdd ** 2 + (row_z - src_z) ** 2))[None, :, None]
    dg = float(gammas[1] - gammas[0])
    dbeta = float(thetas[1] - thetas[0]) if n_views > 1 else 2 * np.pi
    dzr = float(row_z[1] - row_z[0])
    c = (np.arange(n_matrix) - n_matrix / 2 + 0.5) * (fov / n_matrix)
    x, y = np.meshgrid(c, c)
    vol = np.zeros((len(slices_z), n_matrix, n_matrix))
    for i in range(n_views):
        cb, sb = np.cos(thetas[i]), np.sin(thetas[i])
        dx, dy = x - sid * cb, y - sid * sb
        gam = np.arctan2(-cb * dy + sb * dx, -(cb * dx + sb * dy))
        pos = gam / dg + 0.5 * (n_ch - 1)
        k = np.floor(pos).astype(np.int64)
        w = pos - k
        ok = (k >= 0) & (k < n_ch - 1)
        kk = np.clip(k, 0, n_ch - 2)
        l2 = dx * dx + dy * dy
        for s_i, z in enumerate(slices_z):
            rpos = (src_z + (z - src_z) * sdd / np.sqrt(l2) - row_z[0]) / dzr
            r0 = np.floor(rpos).astype(np.int64)
            wr = rpos - r0
            okr = ok & (r0 >= 0) & (r0 < n_rows - 1)
            rr = np.clip(r0, 0, n_rows - 2)
            va = (1 - w) * q[i, rr, kk] + w * q[i, rr, kk + 1]
            vb = (1 - w) * q[i, rr + 1, kk] + w * q[i, rr + 1, kk + 1]
            vol[s_i] += np.where(okr, ((1 - wr) * va + wr * vb) / l2, 0.0)
    return vol * dbeta
