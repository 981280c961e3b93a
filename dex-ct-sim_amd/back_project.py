"""Fan-beam filtered back-projection: drop-in for the reference's ``xtomosim.back_project.get_recon``.

``recon_raw, recon_HU = get_recon(sino, ct, spec, N_matrix, FOV, ramp)`` is called at main.py:134 (log
sinogram of one spectrum) and main.py:168 (basis-material sinograms, ``spec`` a filler there).  The
reference source is in the un-vendored x-tomo-sim submodule; README.md:30-31 describes it as fan-beam
FBP with a windowed ramp filter.  This build implements Kak & Slaney 3.4.1 for the equiangular fan of
``FanBeamGeometry`` (see oracle/fbp_oracle.py for the formulas): ``ramp`` is the cutoff of the
band-limited ramp as a fraction of the Nyquist frequency (``ramp_filter_percent_Nyquist``,
input/params.txt:35), ``recon_raw`` is in 1/cm on an ``N_matrix`` x ``N_matrix`` grid over ``FOV`` cm,
``recon_HU = 1000 (raw - mu_w) / mu_w`` with the spectrum- and detector-weighted water attenuation.
The convolution and the back-projection run in the HIP library (dexct_fbp_filter, dexct_fbp_backproject).  Rotations shorter than 2 pi (``rotation_angle_total``,
input/params.txt:24) down to pi + the fan angle are reconstructed with Parker's short-scan weights (dexct_fbp_parker).
"""
import numpy as np
import torch

from . import _native, xcompy
from ._device import device, ptr, stream_ptr, to_dev
from .forward_project import effective_weights

WATER = 'H(11.2)O(88.8)'       # the composition plots.py:140 uses for HU


WINDOWS = {                      # apodisation W(x), x = f / f_cutoff in [0, 1], applied to the ramp |f|
    'rect': None,                                                        # band-limited ramp (Ram-Lak)
    'sinc': lambda x: np.sinc(0.5 * x),                                  # Shepp-Logan: sin(pi x / 2) / (pi x / 2)
    'cosine': lambda x: np.cos(0.5 * np.pi * x),
    'hann': lambda x: 0.5 * (1.0 + np.cos(np.pi * x)),
    'hamming': lambda x: 0.54 + 0.46 * np.cos(np.pi * x),
}


def default_window():
    """The reference's README (README.md:30-31) speaks of "a sinc window filter"; its source is not available, so
    the plain band-limited ramp stays the default and the window is a choice (``DEXCT_FBP_WINDOW``)."""
    import os
    return os.environ.get('DEXCT_FBP_WINDOW', 'rect')


def _windowed_ramp(t, fc, window):
    """h(t) = 2 int_0^fc f W(f / fc) cos(2 pi f t) df for an array of lags t, by Gauss-Legendre panels shorter
    than half a period of the fastest cosine (float64, error ~1e-15 of h(0))."""
    t = np.asarray(t, dtype=np.float64)
    n_panels = int(np.ceil(4.0 * fc * np.max(np.abs(t)))) + 2
    x, w = np.polynomial.legendre.leggauss(12)
    edges = np.linspace(0.0, fc, n_panels + 1)
    half = 0.5 * (edges[1:] - edges[:-1])
    f = (0.5 * (edges[1:] + edges[:-1])[:, None] + half[:, None] * x[None, :]).ravel()       # nodes
    wf = (half[:, None] * w[None, :]).ravel() * f * window(f / fc)
    out = np.empty(t.shape)
    step = max(1, int(4e6 // f.size))
    for i in range(0, t.size, step):
        out[i:i + step] = 2.0 * (np.cos(2.0 * np.pi * t[i:i + step, None] * f[None, :]) @ wf)
    return out


def ramp_taps(n_channels, dgamma, ramp=1.0, window='rect'):
    """Equiangular filter taps g(n dgamma), n = -(N-1)..(N-1), cutoff ``ramp`` x Nyquist (float64);
    ``window`` apodises the ramp below the cutoff (WINDOWS)."""
    if window not in WINDOWS:
        raise ValueError(f'unknown filter window {window!r}; choose from {sorted(WINDOWS)}')
    n = np.arange(-(n_channels - 1), n_channels, dtype=np.float64)
    c = float(ramp)
    t = n * dgamma
    if WINDOWS[window] is None:
        h = (c * c / (2 * dgamma ** 2)) * np.sinc(c * n) - (c * c / (4 * dgamma ** 2)) * np.sinc(c * n / 2) ** 2
    else:
        pos = _windowed_ramp(t[n_channels - 1:], c / (2.0 * dgamma), WINDOWS[window])      # h is even in t
        h = np.concatenate([pos[:0:-1], pos])
    with np.errstate(invalid='ignore', divide='ignore'):
        ratio = np.where(n == 0, 1.0, t / np.sin(t))
    return 0.5 * ratio ** 2 * h


def water_mu(ct, spec):
    """Effective water attenuation [1/cm] seen by (spectrum, detector): weighted mean over the spectrum."""
    w = effective_weights(ct, spec)
    return float(np.sum(w * xcompy.mixatten(WATER, spec.E)) / np.sum(w))


def default_slices(ct):
    """Slice grid of a cone-beam reconstruction when none is given: one slice per detector row, spaced by the row
    height at the isocentre, centred on z = 0 (the centre of the phantom grid)."""
    return ct.N_rows, -0.5 * (ct.N_rows - 1) * ct.h_iso, ct.h_iso


def recon_device(sino_d, ct, N_matrix, FOV, ramp, window=None, slices=None):
    """sino_d: device float32 [N_proj, N_channels] or [N_proj, N_rows, N_channels] -> image tensor
    [N_matrix, N_matrix] or [N_rows, N_matrix, N_matrix] (float32, 1/cm).  A cone-beam scanner (``ct.cone``) is
    reconstructed with Feldkamp's algorithm onto ``slices = (n_slices, z_first [cm], dz [cm])``."""
    lib = _native.load()
    dev = sino_d.device
    three_d = sino_d.dim() == 3
    s = sino_d if three_d else sino_d[:, None, :]
    s = s.contiguous()
    n_views, n_rows, n_ch = s.shape
    if n_views != ct.N_proj or n_ch != ct.N_channels:
        raise ValueError(f'sinogram {tuple(sino_d.shape)} does not match the scanner ({ct.N_proj} x {ct.N_channels})')
    st = stream_ptr()
    if ct.theta_tot > 2 * np.pi + 1e-4:
        raise ValueError(f'rotation_angle_total = {ct.theta_tot:.6g} exceeds 2 pi: more than one rotation is not reconstructed')
    if ct.theta_tot < 2 * np.pi - 1e-4:
        # a short scan (rotation_angle_total, input/params.txt:24): Parker's weights make every ray count once
        need = np.pi + (n_ch - 1) * ct.dgamma
        if ct.theta_tot < need * (1 - 1e-12):
            raise ValueError(f'rotation_angle_total = {ct.theta_tot:.6g} rad is less than a short scan (pi + fan angle = '
                             f'{need:.6g} rad): projections are missing')
        sw = torch.empty_like(s)
        _native.check(lib.dexct_fbp_parker(ptr(s), n_views, n_rows, n_ch, float(ct.theta_tot), float(ct.dgamma), 0, n_views,
                                           ptr(sw), st), 'dexct_fbp_parker')
        s = sw
    cone = bool(getattr(ct, 'cone', False)) and n_rows > 1
    taps = to_dev(ramp_taps(n_ch, ct.dgamma, ramp, window or default_window()), torch.float32, dev)
    weight = to_dev(ct.SID * np.cos(ct.gammas), torch.float32, dev)
    view_cs = to_dev(ct.view_cs(), torch.float64, dev)
    q = torch.empty_like(s)
    _native.check(lib.dexct_fbp_filter(ptr(s), ptr(taps), ptr(weight), n_views * n_rows, n_ch, ct.dgamma, ptr(q), st),
                  'dexct_fbp_filter')
    if cone:
        if n_rows != ct.N_rows:
            raise ValueError(f'sinogram has {n_rows} rows, the scanner {ct.N_rows}')
        n_slices, z0, dz = slices if slices is not None else default_slices(ct)
        row_z = ct.row_z()
        row_w = to_dev(ct.SDD / np.sqrt(ct.SDD ** 2 + (row_z - ct.src_z) ** 2), torch.float32, dev)
        img = torch.empty((int(n_slices), N_matrix, N_matrix), dtype=torch.float32, device=dev)
        _native.check(lib.dexct_fdk_backproject(ptr(q), ptr(view_cs), ptr(row_w), n_views, n_ch, n_rows, ct.SID, ct.SDD,
                                                ct.dgamma, ct.theta_tot / ct.N_proj, float(row_z[0]), float(ct.h),
                                                float(ct.src_z), int(N_matrix), float(FOV), int(n_slices), float(z0),
                                                float(dz), ptr(img), st), 'dexct_fdk_backproject')
        return img
    img = torch.empty((n_rows, N_matrix, N_matrix), dtype=torch.float32, device=dev)
    _native.check(lib.dexct_fbp_backproject(ptr(q), ptr(view_cs), n_views, n_ch, n_rows, ct.SID, ct.dgamma,
                                            ct.theta_tot / ct.N_proj, int(N_matrix), float(FOV), ptr(img), st),
                  'dexct_fbp_backproject')
    return img if three_d else img[0]


def get_recon(sino, ct, spec, N_matrix, FOV, ramp, window=None, slices=None):
    """Drop-in for ``recon_raw, recon_HU = get_recon(sino, ct, spec, N_matrix, FOV, ramp)`` (main.py:134).
    ``window``: apodisation of the ramp (WINDOWS; default ``DEXCT_FBP_WINDOW`` or the plain band-limited ramp).
    Cone-beam sinograms ([N_proj, N_rows, N_channels] of a ``cone=True`` scanner) are reconstructed with Feldkamp's
    algorithm into ``slices = (n_slices, z_first, dz)`` (default: default_slices)."""
    dev = device()
    sino_d = to_dev(np.asarray(sino, dtype=np.float32), torch.float32, dev)
    bad = int((~torch.isfinite(sino_d)).sum().item())
    if bad:
        # one NaN (a photon-starved ray whose decomposition diverged, ln of a zero count) would spread over its whole
        # detector line in the filter and over the whole image in the back-projection
        import warnings
        warnings.warn(f'get_recon: {bad} non-finite sinogram value(s) set to 0 before filtering', RuntimeWarning)
        sino_d = torch.nan_to_num(sino_d, nan=0.0, posinf=0.0, neginf=0.0)
    raw = recon_device(sino_d, ct, N_matrix, FOV, ramp, window, slices).cpu().numpy()
    mu_w = water_mu(ct, spec)
    return raw, (1000.0 * (raw - mu_w) / mu_w).astype(np.float32)


def make_vmi(E0, M1, M2, HU=True, matcomp1=None, matcomp2=None):
    """Virtual monoenergetic image at E0 [keV] from the two basis-material images - the reference's
    ``make_vmi`` (plots.py:136-144): ``u_p_1 * M1 + u_p_2 * M2`` with the basis mass attenuations at E0, in HU
    against water ('H(11.2)O(88.8)', plots.py:140) unless ``HU=False``.  float32 out, like the reference."""
    from . import matdecomp as md
    lib = _native.load()
    dev = device()
    E = np.array([float(E0)])
    u1 = float(xcompy.mixatten(matcomp1 or md.matcomp1, E)[0])
    u2 = float(xcompy.mixatten(matcomp2 or md.matcomp2, E)[0])
    uw = float(xcompy.mixatten(WATER, E)[0])
    m1 = to_dev(np.asarray(M1, dtype=np.float32), torch.float32, dev)
    m2 = to_dev(np.asarray(M2, dtype=np.float32), torch.float32, dev)
    if m1.shape != m2.shape:
        raise ValueError('basis images differ in shape')
    out = torch.empty_like(m1)
    _native.check(lib.dexct_vmi(ptr(m1), ptr(m2), m1.numel(), u1, u2, uw, int(bool(HU)), ptr(out), stream_ptr()),
                  'dexct_vmi')
    return out.cpu().numpy()
