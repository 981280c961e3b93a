"""Measurements of the reference's analysis script (plots.py), without its figures.

The reference's ``plots.py`` mixes matplotlib panels with a handful of numerical helpers; this module provides
those helpers under the same names and argument meaning, so that its measurement cells (ROI signal / noise /
CNR, plots.py:236-263, 332-418; RMSE of virtual monoenergetic images against the phantom's ground truth,
plots.py:276-312) can be run on the engine's outputs:

  make_vmi(E0, M1, M2, HU, matcomp1, matcomp2)   plots.py:136-144   (HIP: dexct_vmi)
  measure_roi(M, roi_info, give_roi, ax)         plots.py:146-158   (HIP: dexct_label_moments)
  crop_img(M, crop)                              plots.py:167-170
  get_xcat_mask(M, threshold)                    plots.py:226-231
  get_img_ct / get_img_basismats                 plots.py:173-207   (paths of main.py's output tree)

and two sweeps the reference writes as Python loops over energies (one make_vmi + one measurement per energy):

  vmi_roi_sweep(Evals, M1, M2, roi_signal, roi_background)  ->  signal, noise, CNR per energy  (plots.py:386-395)
  vmi_rmse_sweep(Evals, M1, M2, gt_labels, gt_values, mask) ->  RMSE per energy                (plots.py:297-303)

A VMI is ``u1(E) M1 + u2(E) M2``; mean, variance and squared error against a piecewise-constant ground truth
are therefore closed forms in the second-order moments of (M1, M2) per region, which one pass of
``dexct_label_moments`` delivers for all energies at once (float64 sums of the float32 images; the reference
rounds each VMI to float32 first, so results agree to float32 rounding, not bitwise).
"""
import os

import numpy as np
import torch

from . import _native, xcompy
from ._device import device, ptr, stream_ptr, to_dev
from .back_project import WATER, make_vmi          # noqa: F401  (re-export under the reference's module name)


def label_moments(M1, M2=None, labels=None, n_labels=1):
    """[n_labels, 6] float64: count, S m1, S m2, S m1^2, S m1 m2, S m2^2 per label (dexct_label_moments)."""
    lib = _native.load()
    dev = device()
    m1 = to_dev(np.ascontiguousarray(M1, dtype=np.float32), torch.float32, dev)
    if m1.numel() == 0:
        return np.zeros((n_labels, 6))
    m2 = None
    if M2 is not None:
        m2 = to_dev(np.ascontiguousarray(M2, dtype=np.float32), torch.float32, dev)
        if m2.shape != m1.shape:
            raise ValueError('images differ in shape')
    lab = None
    if labels is not None:
        lab = to_dev(np.ascontiguousarray(labels, dtype=np.uint8), torch.uint8, dev)
        if lab.shape != m1.shape:
            raise ValueError('label image differs in shape')
    out = torch.empty((int(n_labels), 6), dtype=torch.float64, device=dev)
    _native.check(lib.dexct_label_moments(ptr(m1), ptr(m2) if m2 is not None else None,
                                          ptr(lab) if lab is not None else None, m1.numel(), int(n_labels), ptr(out),
                                          stream_ptr()), 'dexct_label_moments')
    return out.cpu().numpy()


def _roi_slices(roi_info, shape):
    x0, y0, dx, dy = (int(v) for v in roi_info)
    # NumPy slice semantics of ``mask[y0:y0+dy, x0:x0+dx] = 1`` (plots.py:149): clipped to the image
    return slice(max(y0, 0), max(min(y0 + dy, shape[0]), 0)), slice(max(x0, 0), max(min(x0 + dx, shape[1]), 0))


def measure_roi(M, roi_info, give_roi=False, ax=None):
    """Mean and (population) variance of the rectangle ``roi_info = [x0, y0, dx, dy]`` of image ``M``
    (plots.py:146-158).  ``give_roi`` returns the ROI's pixels instead; ``ax`` draws the outline like the reference."""
    M = np.asarray(M)
    ys, xs = _roi_slices(roi_info, M.shape)
    roi = M[ys, xs]
    if ax is not None:
        x0, y0, dx, dy = roi_info
        ax.plot([x0 + dx, x0, x0, x0 + dx, x0 + dx], [y0, y0, y0 + dy, y0 + dy, y0], 'r-', lw=0.5)
    if give_roi:
        return roi.ravel()
    if roi.size == 0:
        return float('nan'), float('nan')          # np.mean / np.var of an empty selection
    c, s1, _, s11, _, _ = label_moments(roi)[0]
    u = s1 / c
    return u, max(s11 / c - u * u, 0.0)


def crop_img(M, crop):
    """Central ``crop`` x ``crop`` window (plots.py:167-170)."""
    r0 = M.shape[0] // 2
    return M[r0 - crop // 2:r0 + crop // 2, r0 - crop // 2:r0 + crop // 2]


def get_xcat_mask(M, threshold=-900):
    """Boolean mask of values above ``threshold`` (non-air pixels of a HU image by default; plots.py:226-231)."""
    return np.asarray(M) > threshold


def get_img_ct(phantom_id, spec_id, dose, crop=None, units='HU', N_matrix=512, out_dir='output', run_prefix='mvkv_'):
    """Reconstruction written by main.py for (phantom, spectrum, dose [mGy]) (plots.py:173-182; main.py:102,135-136)."""
    assert units in ('HU', 'raw')
    path = os.path.join(out_dir, f'{run_prefix}{phantom_id}', f'{spec_id}_{int(dose * 1000):04}uGy',
                        f'recon_{units}_float32.bin')
    M = np.fromfile(path, dtype=np.float32).reshape([N_matrix, N_matrix])
    return M if crop is None else crop_img(M, crop)


def get_img_basismats(phantom_id, spec_id1, spec_id2, dose1, dose2, crop=None, N_matrix=512, out_dir='output',
                      run_prefix='mvkv_'):
    """Basis-material reconstructions of a spectral pair (plots.py:199-207; main.py:144,169-170)."""
    d = os.path.join(out_dir, f'{run_prefix}{phantom_id}',
                     f'matdecomp_{spec_id1}_{spec_id2}_{int(dose1 * 1000):04}uGy_{int(dose2 * 1000):04}uGy')
    M1 = np.fromfile(os.path.join(d, 'mat1_recon_float32.bin'), dtype=np.float32).reshape([N_matrix, N_matrix])
    M2 = np.fromfile(os.path.join(d, 'mat2_recon_float32.bin'), dtype=np.float32).reshape([N_matrix, N_matrix])
    return (M1, M2) if crop is None else (crop_img(M1, crop), crop_img(M2, crop))


def _basis_mu(Evals, matcomp1, matcomp2):
    from . import matdecomp as md
    E = np.atleast_1d(np.asarray(Evals, dtype=np.float64))
    return (xcompy.mixatten(matcomp1 or md.matcomp1, E), xcompy.mixatten(matcomp2 or md.matcomp2, E),
            xcompy.mixatten(WATER, E))


def vmi_roi_sweep(Evals, M1, M2, roi_signal, roi_background, HU=True, matcomp1=None, matcomp2=None):
    """Signal, noise and CNR of the VMIs at every energy of ``Evals`` - the loop of plots.py:386-395
    (``make_vmi``; ``measure_roi`` on a signal and a background rectangle; ``(u1 - u2) / sqrt(v1 + v2)``).
    Returns a dict of arrays ``u_signal, v_signal, u_background, v_background, noise, cnr``."""
    M1, M2 = np.asarray(M1), np.asarray(M2)
    if M1.shape != M2.shape or M1.ndim != 2:
        raise ValueError('basis images must be 2-D and of one shape')
    labels = np.full(M1.shape, 255, dtype=np.uint8)
    ys, xs = _roi_slices(roi_background, M1.shape)
    labels[ys, xs] = 1
    ys, xs = _roi_slices(roi_signal, M1.shape)
    labels[ys, xs] = 0                  # overlapping rectangles: the signal ROI wins (the reference measures each alone)
    mom = label_moments(M1, M2, labels, 2)
    u1, u2, uw = _basis_mu(Evals, matcomp1, matcomp2)
    scale = 1000.0 / uw if HU else np.ones_like(uw)
    res = {}
    for name, (c, s1, s2, s11, s12, s22) in zip(('signal', 'background'), mom):
        if c == 0:
            mean = var = np.full(u1.shape, np.nan)
        else:
            m1, m2 = s1 / c, s2 / c
            c11, c12, c22 = s11 / c - m1 * m1, s12 / c - m1 * m2, s22 / c - m2 * m2
            mean = u1 * m1 + u2 * m2
            var = np.maximum(u1 * u1 * c11 + 2 * u1 * u2 * c12 + u2 * u2 * c22, 0.0) * scale * scale
            mean = (mean - uw) * scale if HU else mean
        res['u_' + name], res['v_' + name] = mean, var
    res['noise'] = np.sqrt(res['v_signal'] + res['v_background'])
    with np.errstate(divide='ignore', invalid='ignore'):
        res['cnr'] = (res['u_signal'] - res['u_background']) / res['noise']
    return res


def vmi_rmse_sweep(Evals, M1, M2, gt_labels, gt_values, mask=None, HU=True, matcomp1=None, matcomp2=None):
    """RMSE of the VMIs against a piecewise-constant ground truth at every energy - the loop of plots.py:297-303.

    ``gt_labels`` [Ny, Nx] uint8 region ids (e.g. a phantom slice's material ids), ``gt_values`` [n_regions, nE]
    the true linear attenuation [1/cm] of each region at each energy (``VoxelPhantom.mu_table(Evals)``),
    ``mask`` optional boolean image of the pixels that count (plots.py:290 uses the non-air pixels).
    With ``HU`` both images are compared in Hounsfield units, as the reference does."""
    M1, M2 = np.asarray(M1), np.asarray(M2)
    gt_values = np.asarray(gt_values, dtype=np.float64)
    n_reg = gt_values.shape[0]
    labels = np.asarray(gt_labels).astype(np.uint8)
    if labels.shape != M1.shape or M2.shape != M1.shape:
        raise ValueError('images and labels must have one shape')
    if n_reg > 64:
        raise ValueError('at most 64 ground-truth regions')
    if labels.max(initial=0) >= n_reg:
        raise ValueError('a label has no row in gt_values')
    if mask is not None:
        labels = np.where(np.asarray(mask, dtype=bool), labels, 255).astype(np.uint8)
    mom = label_moments(M1, M2, labels, n_reg)
    u1, u2, uw = _basis_mu(Evals, matcomp1, matcomp2)
    if gt_values.shape[1] != u1.shape[0]:
        raise ValueError('gt_values needs one column per energy')
    sse = np.zeros_like(u1)
    for (c, s1, s2, s11, s12, s22), g in zip(mom, gt_values):
        sse += u1 * u1 * s11 + 2 * u1 * u2 * s12 + u2 * u2 * s22 - 2 * g * (u1 * s1 + u2 * s2) + g * g * c
    n = mom[:, 0].sum()
    rmse = np.sqrt(np.maximum(sse, 0.0) / n) if n > 0 else np.full(u1.shape, np.nan)
    return rmse * (1000.0 / uw) if HU else rmse
