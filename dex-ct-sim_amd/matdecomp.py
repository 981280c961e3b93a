"""Dual-energy basis-material decomposition on the GPU: drop-in for the reference's matdecomp.py.

Same public names and argument meaning as /root/reference/matdecomp.py:
``mat1, matcomp1, density1, mat2, matcomp2, density2`` (:11-17, imported by plots.py:18),
``optimize_sino`` / ``optimize_sino_cpu`` (:20-127), ``do_matdecomp_gn`` (:130-164),
``get_basismat_sinos`` (:167-207).  The Newton iterations run in the HIP kernel behind
``dexct_gn_decompose``; this module only builds the energy tables the way the reference does
and moves arrays.  There is no NumPy or CuPy compute path here.

Error behaviour: the reference raises ``numpy.linalg.LinAlgError`` when a pixel's 2x2 Hessian is
exactly singular (matdecomp.py:125); the kernel never traps and leaves inf/NaN in such a pixel
(masked air pixels are set to 0 afterwards exactly as in the reference, :204-205).
``get_basismat_sinos(..., strict=True)`` restores an exception: it raises ``SingularHessianError`` (a
``LinAlgError``) when a pixel outside the air mask ends non-finite.
"""
import os

import numpy as np
import torch

from . import _native, _shard, xcompy as xc
from ._device import LazyPinnedResult, device, locked_arrays, pinned_empty, pool_wanted, side_streams, ptr, stream_ptr, to_dev, to_host

# Basis materials, as data (matdecomp.py:11-17).
mat1 = 'ICRU tissue'
matcomp1 = 'H(10.2)C(14.3)N(3.4)O(70.8)Na(0.2)P(0.3)S(0.3)Cl(0.2)K(0.3)'
density1 = 1.06  # [g/cm3]
mat2 = 'ICRU bone'
matcomp2 = 'H(3.4)C(15.5)N(4.2)O(43.5)Na(0.1)Mg(0.2)P(10.3)S(0.3)Ca(22.5)'
density2 = 1.92  # [g/cm3]

class SingularHessianError(np.linalg.LinAlgError):
    """A pixel outside the air mask ended inf/NaN: its 2x2 Hessian became singular on the way (the reference's
    ``np.linalg.inv`` raises LinAlgError for an exactly singular one, matdecomp.py:125)."""


# 'f64': float64 throughout (the reference's arithmetic).  'mixed': float32 bulk iterations, then
# N_POLISH float64 iterations (same total count).  Override with DEXCT_GN_PRECISION.
DEFAULT_PRECISION = os.environ.get('DEXCT_GN_PRECISION', 'f64')
N_POLISH = 4


def _as_device_counts(x, dev):
    """Counts as a contiguous device tensor, float64 if given float64, else float32."""
    if isinstance(x, torch.Tensor):
        t = x.to(dev)
        if t.dtype not in (torch.float32, torch.float64):
            t = t.to(torch.float64)
        return t.contiguous()
    a = np.asarray(x)
    dt = torch.float32 if a.dtype == np.float32 else torch.float64
    return to_dev(a.astype(np.float32 if dt == torch.float32 else np.float64, copy=False), dt, dev)


# Tolerance stop of the Newton iteration (include/dexct.h, dexct_gn_options).  None = the default below; 0 = the reference's
# fixed count, bit for bit.  The default is resolved ONCE, here, from the environment (DEXCT_GN_EXACT=1 -> 0,
# DEXCT_GN_STOP_TOL=<t>, else 1e-12) and every call hands the library an explicit value: one parser, no drift between what the
# host calibrates the short cut for and what the library runs.
def _default_stop_tol():
    if os.environ.get('DEXCT_GN_EXACT', '')[:1] == '1':
        return 0.0
    t = os.environ.get('DEXCT_GN_STOP_TOL')
    if t is None or t.strip() == '':
        return 1.0e-12
    try:
        v = float(t)
    except ValueError:
        raise ValueError(f'DEXCT_GN_STOP_TOL={t!r} is not a number') from None
    if not v >= 0.0:
        raise ValueError(f'DEXCT_GN_STOP_TOL={t!r} must be >= 0')
    return v


DEFAULT_STOP_TOL = _default_stop_tol()

# The short cut of the Newton solve (include/dexct.h, dexct_gn_options.pass / .start; csrc/gn.hip gn_start, gn_shortcut_kernel).
# Most of the reference's ~17 Newton steps per pixel are the walk from its start value 1e-6 to the neighbourhood of the
# solution; what it returns is the fixed point its walk ends at.  That is a function of the pixel's two counts alone, so it is
# tabulated once per pair of spectra: the library's own single launch (full tables, from 1e-6, counting steps) is run on the
# counts at the corners of a 384 x 384 cell grid (quadrature.GATE_CELLS) over (ln u0, u1 / u0), u_k = ln(air_k / g_k) / 16, and
# once more on the cell centres as a check (quadrature.newton_start_grid / assemble_start / validate_start: where does the walk
# end, after how many steps, is that an isolated root of the two equations, how smoothly does it vary).  A pixel whose counts
# fall in an open cell - the walk ends by the tolerance rule within n_iters steps at all corners of the 5 x 5 cells around it, at
# well-conditioned roots that vary smoothly (no boundary between two basins), the attenuation is not beyond exp(-12) - starts
# from the 6 x 6 Lagrange interpolant of those fixed points (1e-10 of |a| from its own at the median water ray) and takes ONE step
# where the table vouches for it, else two - the second being the tolerance rule's evidence that the FULL model has converged to
# stop_tol.  The result is accepted only within the cell's radius of the interpolant (the reference's branch); a pixel without
# that evidence or acceptance, or in a closed cell (few steps asked for, counts outside the grid, NaN), is solved from 1e-6 with
# all n_iters steps in the same launch.  What comes out is, per pixel, a fixed point of the full model verified to stop_tol on
# the reference's branch, or the reference's own trajectory: the contract of the single launch with the tolerance stop, asserted
# against the exact count on every pixel of the benchmark (bench.py, tests/test_gpu_full_scale.py), on the reference goldens at
# 1 / 2 / 5 / 50 iterations and in tools/soak_gn.py.  1.0 steps per pixel instead of ~17.
#
# ILL-POSED PAIRS run the reference's fixed count.  Where the calibration itself shows that the pair of spectra does not
# determine two thicknesses (quadrature.pair_is_ill_posed: the walk comes to rest where it does not reproduce its counts on more
# than 5 % of the data plane, fewer than 78 % of the cells are open, or the roots are ill-conditioned - the MV / kV pairs, the
# class of the reference's live pair, main.py:101; every bundled pair's class is pinned in tests/test_gpu_gn.py) the default is
# stop_tol = 0 for that pair: the tolerance rule on the wandering pixels of such a pair was the only place where the default
# ever differed from the exact count (profiles/r04_gn_noisy_public.log), and it saved 15 % there.  An explicit stop_tol > 0 is
# honoured.
#
# ONE STEP where the table vouches for it (the default since round 5; round 6: the CHORD step - csrc/gn.hip chord_residuals_f64,
# gn_start<DERIV>; quadrature.chord_tables).  What a step from 1e-10 of the fixed point has to get right is the residual - the
# relative misfit of the counts, c_k = g_k / nu_k(s) - 1: the full energy sum of nu, 2 of the 12 accumulations per energy - not
# the Jacobian: and the inverse Jacobian is already in the table, as the GRADIENT of the tabulated fixed points with respect to
# the (logarithms of the) counts, formed from the same 36 loads as the start value.  The step m = s + B c leaves at most
# eps e0 + kappa e0^2 - kappa from the second derivatives of the misfit at the tabulated fixed points, eps from the table's
# gradient against the exact Jacobians at every cell's corners and centre - and its own length d1 measures e0: a pixel with
# (kappa d1 + eps) d1 <= stop_tol / 4 * max(min(|a0|, |a1|), 1) ends there (a bound on the distance it still has to go, per
# component, which is what the tolerance rule extracts from two steps); every other pixel goes on with full Newton steps and the
# rule.  (Round 5's one step was of the Gauss-Newton form: 6 of the 12 sums.)
#
# Modes (``two_level=`` of the calls below; DEXCT_GN_TWO_LEVEL in the environment; DEFAULT_TWO_LEVEL):
#   None / True / 'one'     the short cut, one step where kappa allows (one launch)
#   'start'                 the short cut, always two steps and the tolerance rule (round 4's form)
#   False / '0'             the single launch from 1e-6
# It applies to float64 with one shared spectrum, the tolerance stop on, 4 <= n_iters <= 254 and >= 48 energies; anything
# else runs the single launch.
SHORTCUT_MODES = ('one', 'start')
DEFAULT_TWO_LEVEL = {'0': False, '1': True, 'start': 'start', 'one': 'one'}.get(os.environ.get('DEXCT_GN_TWO_LEVEL', ''), None)

# Sampled audit of the short cut (``audit=`` of the calls below; DEXCT_GN_AUDIT=<ppm>[,strict] in the environment): that many
# pixels per million, chosen by the device's Philox generator, are solved again with the reference's fixed count in the same
# call and compared (<= 1e-12 relative, same finite / NaN pattern); a difference warns (GnAuditWarning) or, strict, raises
# (GnAuditError) with the worst pixel's counts.  Off by default; main.py switches it on.
def _default_audit():
    v = os.environ.get('DEXCT_GN_AUDIT', '')
    if not v:
        return 0.0, False
    parts = v.split(',')
    return float(parts[0]), (len(parts) > 1 and parts[1].strip() == 'strict')


DEFAULT_AUDIT_PPM, DEFAULT_AUDIT_STRICT = _default_audit()
AUDIT_TOL = 1.0e-12


class GnAuditWarning(UserWarning):
    pass


class GnAuditError(RuntimeError):
    pass


_last_ws = []          # workspaces of the most recent call (one per view chunk of the pipelined boundary)
_last_events = []      # (before, after) events of the launches of the most recent call(s)
_last_zeroed = None
_last_mode = 'single'
_last_audit = None
_table_cache = {}


def last_gn_stats():
    """Diagnostics of the most recent gn_device call - or of ALL the chunk launches of the most recent pipelined
    get_basismat_sinos call - (synchronises): ``pixel_iterations`` = Newton steps the float64 shared-spectrum kernels
    actually executed (the exits end pixels before n_iters; masked air pixels run none), ``stalled_lane_steps`` = lane-steps
    a wave could not hand out because all its result slots waited for stragglers.  0 for the mixed-precision and
    per-channel-spectrum kernels, which do not count.  ``mode``: 'one' / 'start' (the short cut), 'single', or 'exact (ill-posed
    pair)'; ``audit``: the last sampled audit, if any."""
    if not _last_ws:
        return None
    words = torch.stack([w[72:104].view(torch.int64) for w in _last_ws]).sum(dim=0).tolist()
    st = {'pixel_iterations': int(words[0]), 'stalled_lane_steps': int(words[3]), 'launches': len(_last_ws), 'mode': _last_mode}
    if _last_events:
        torch.cuda.synchronize()
        st['main_ms'] = sum(e[0].elapsed_time(e[1]) for e in _last_events)
    if _last_audit is not None:
        st['audit'] = _last_audit
    return st


def _host_tables(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x, dtype=np.float64)


def _walk(lib, dev, i0_d, mus_d, n_e, g, cal_tol):
    """The reference's iteration (the library's own kernel, full tables, from 1e-6, counting steps) on counts g [n, 2]:
    (steps until the tolerance rule fired | 255, where it ended)."""
    g_d = to_dev(np.ascontiguousarray(g.T), torch.float64, dev)
    n_c = g_d.shape[1]
    a_c = torch.empty((n_c, 2), dtype=torch.float64, device=dev)
    k_c = torch.empty(n_c, dtype=torch.uint8, device=dev)
    ws = torch.empty(lib.dexct_gn_workspace_bytes(n_e, 1), dtype=torch.uint8, device=dev)
    _native.check(lib.dexct_gn_decompose(ptr(g_d[0]), ptr(g_d[1]), 1, n_c, ptr(i0_d), ptr(mus_d), n_e, 1, 1, 254, 0, 0, None, 0.95,
                                         ptr(a_c), _native.gn_options(cal_tol, 0, 0, 1, _native.GN_PASS_COUNT, k_c.data_ptr()),
                                         ptr(ws), stream_ptr()), 'dexct_gn_decompose (gate calibration)')
    return k_c.cpu().numpy(), a_c.cpu().numpy()


def calibrate_gate(i0_h, mus_h, i0_d, mus_d, dev, cal_tol):
    """THE GATE of the short cut (csrc/gn.hip, gn_start): the reference's iteration run on the counts at the corners of a cell
    grid in data space; where it ends, after how many steps, and how smoothly that varies decides where pixels may take the
    short cut and where they start; then the table is checked against what it stands for at every cell's centre.
    Returns (start array or None, stats) - stats says what the calibration saw of the pair (quadrature.pair_is_ill_posed)."""
    from . import quadrature
    i0_2 = i0_h.reshape(2, -1)
    pieces = quadrature.newton_start_grid(i0_2, mus_h)
    if pieces is None:
        return None, {'grid': False}
    pieces['device'] = dev                    # (the energy sums of the table assembly run there: quadrature._model_sums)
    lib = _native.load()
    n_e = i0_2.shape[1]
    steps, roots = _walk(lib, dev, i0_d, mus_d, n_e, pieces['corner_g'], cal_tol)
    start_h, share, stats = quadrature.assemble_start(pieces, steps, roots)
    start_h, share, n_bad = quadrature.validate_start(start_h, pieces, *_walk(lib, dev, i0_d, mus_d, n_e, quadrature.cell_centres(pieces), cal_tol))
    stats = dict(stats, grid=True, open_share=float(share), centres_failed=int(n_bad))
    return start_h, stats


def _lib_fingerprint():
    """sha256 of the built library (once per process): a cached gate table is only as good as the kernel that made it."""
    global _lib_hash
    if _lib_hash is None:
        import hashlib
        h = hashlib.sha256()
        with open(_native.LIB_PATH, 'rb') as f:
            for blk in iter(lambda: f.read(1 << 20), b''):
                h.update(blk)
        _lib_hash = h.hexdigest()
    return _lib_hash


_lib_hash = None


def _cache_dir():
    d = os.environ.get('DEXCT_CACHE_DIR')
    if d is not None and d.strip().lower() in ('', '0', 'off', 'none'):
        return None
    return d or os.path.join(os.path.expanduser('~'), '.cache', 'dexct')


def _gate_cache_path(i0_h, mus_h, cal_tol):
    d = _cache_dir()
    if d is None:
        return None
    import hashlib
    from . import quadrature
    h = hashlib.sha256()
    h.update(f'{_native.ABI_VERSION}|{quadrature.GATE_VERSION}|{cal_tol!r}|{i0_h.shape}|{_lib_fingerprint()}'.encode())
    h.update(i0_h.tobytes())
    h.update(mus_h.tobytes())
    return os.path.join(d, f'gate_{h.hexdigest()[:32]}.npz')


def _gate_from_disk(path, i0_h, mus_h):
    """(start, stats) of an earlier process, or None: the file must be complete, carry its own checksum and agree with the
    grid this process would lay out for the tables (a stale, truncated or foreign file is ignored, never trusted)."""
    if path is None or not os.path.exists(path):
        return None
    try:
        import hashlib
        import json
        from . import quadrature
        with np.load(path, allow_pickle=False) as z:
            start_h = np.ascontiguousarray(z['start'], dtype=np.float64) if z['start'].size else None
            stats = json.loads(str(z['stats']))
            digest = str(z['digest'])
        blob = b'' if start_h is None else start_h.tobytes()
        if hashlib.sha256(blob + json.dumps(stats, sort_keys=True).encode()).hexdigest() != digest:
            return None
        if start_h is not None:
            pieces = quadrature.newton_start_grid(i0_h.reshape(2, -1), mus_h)
            n = quadrature.GATE_CELLS
            head = None if pieces is None else pieces['head'].copy()
            if head is not None:
                head[10] = 2.0                # (the tables of the one-step acceptance follow the cells)
            if (head is None or start_h.shape != (quadrature.start_layout(n)[-1],)
                    or not np.array_equal(start_h[:quadrature.START_HEADER], head)):
                return None
        return start_h, stats
    except Exception:
        return None


def _gate_to_disk(path, start_h, stats):
    if path is None:
        return
    try:
        import hashlib
        import json
        os.makedirs(os.path.dirname(path), exist_ok=True)
        blob = b'' if start_h is None else start_h.tobytes()
        digest = hashlib.sha256(blob + json.dumps(stats, sort_keys=True).encode()).hexdigest()
        tmp = f'{path}.{os.getpid()}.tmp'
        with open(tmp, 'wb') as f:
            np.savez(f, start=np.zeros(0) if start_h is None else start_h, stats=json.dumps(stats, sort_keys=True), digest=digest)
        os.replace(tmp, path)             # atomic: a reader sees the old file or the whole new one
    except OSError:
        pass                              # (a read-only home directory: the table is simply not kept)


def _device_tables(i0, mus, dev, want_gate, cal_tol=1.0e-12):
    """(i0_d [2, nBins, nE], mus_d [2, nE], gate) for host or device tables, cached by content; ``want_gate``: prepare the short
    cut - gate = {'start': device array | None, 'ill_posed': bool, 'stats': ...} (None when not wanted); ``cal_tol``: the
    tolerance the gate is calibrated for.  Tables given as device tensors are used as they are - and read back once per call
    when the short cut is wanted: pass host arrays to avoid that synchronisation.  The gate comes from this process's cache,
    else from DEXCT_CACHE_DIR (default ~/.cache/dexct; validated on load), else from a calibration (two launches + host NumPy,
    ~0.1 s) whose result is stored in both - only after it succeeded: a calibration that raises leaves nothing behind and is
    tried again by the next call."""
    if isinstance(i0, torch.Tensor) and isinstance(mus, torch.Tensor) and not want_gate:
        i0_d, mus_d = to_dev(i0, torch.float64, dev), to_dev(mus, torch.float64, dev)
        return (i0_d[:, None, :].contiguous() if i0_d.dim() == 2 else i0_d), mus_d, None
    i0_h, mus_h = np.ascontiguousarray(_host_tables(i0), dtype=np.float64), np.ascontiguousarray(_host_tables(mus), dtype=np.float64)
    key = (i0_h.shape, i0_h.tobytes(), mus_h.tobytes(), str(dev))
    ent = _table_cache.get(key)
    if ent is None:
        if len(_table_cache) >= 8:
            _table_cache.clear()
        i0_d, mus_d = to_dev(i0_h, torch.float64, dev), to_dev(mus_h, torch.float64, dev)
        if i0_d.dim() == 2:
            i0_d = i0_d[:, None, :].contiguous()
        ent = _table_cache[key] = {'i0': i0_d, 'mus': mus_d}
    if want_gate and ('gate', cal_tol) not in ent:
        gate = {'start': None, 'ill_posed': False, 'stats': {'grid': False}, 'source': 'none'}
        if i0_h.ndim == 2 or i0_h.shape[1] == 1:
            from . import quadrature
            path = _gate_cache_path(i0_h, mus_h, cal_tol)
            got = _gate_from_disk(path, i0_h, mus_h)
            if got is not None:
                start_h, stats = got
                gate['source'] = 'disk'
            else:
                start_h, stats = calibrate_gate(i0_h, mus_h, ent['i0'], ent['mus'], dev, cal_tol)     # (may raise: nothing is stored then)
                _gate_to_disk(path, start_h, stats)
                gate['source'] = 'calibration'
            gate['stats'] = stats
            gate['ill_posed'] = bool(quadrature.pair_is_ill_posed(stats))
            if start_h is not None and not gate['ill_posed'] and stats.get('open_share', 0.0) >= 0.1:
                gate['start'] = to_dev(start_h, torch.float64, dev)
        ent[('gate', cal_tol)] = gate
    return ent['i0'], ent['mus'], (ent.get(('gate', cal_tol)) if want_gate else None)


_audit_calls = 0


def _audit(g1, g2, a, i0, mus, n_iters, ppm, strict, out_rc, mask_max, mask_frac, merge=False):
    """Sampled audit of a default-mode result ``a`` (see DEFAULT_AUDIT_PPM): re-solve ~ppm pixels per million with the
    reference's fixed count (stop_tol = 0, the single launch) and compare.  ``merge``: add to the record of the call's earlier
    chunks."""
    global _last_audit, _audit_calls
    before = _last_audit if merge else None
    n_pix = g1.numel()
    m = int(min(max(round(n_pix * ppm * 1e-6), min(n_pix, 64)), n_pix, 1 << 22))
    gen = torch.Generator(device=g1.device)           # Philox on the device; another sample in every call, the same in every run
    gen.manual_seed(0x5EED + _audit_calls)
    _audit_calls += 1
    idx = torch.randint(n_pix, (m,), generator=gen, device=g1.device)
    s1, s2 = g1.reshape(-1)[idx].contiguous(), g2.reshape(-1)[idx].contiguous()
    if out_rc is not None:                             # where pixel (v, c, r) of the input went: [v][r][c]
        rows, chans = out_rc
        v, rem = idx // (rows * chans), idx % (rows * chans)
        c, r = rem // rows, rem % rows
        oidx = (v * rows + r) * chans + c
    else:
        oidx = idx
    got = a.reshape(-1, 2)[oidx]
    global _last_ws, _last_events, _last_mode
    keep = (_last_ws, _last_events, _last_mode)
    want = gn_device(s1, s2, i0, mus, n_iters, 'f64', mask_max=mask_max, mask_frac=mask_frac, stop_tol=0.0, kernel=1, two_level=False, audit=0)
    _last_ws, _last_events, _last_mode = keep
    rel = (got - want).abs() / want.abs().amax(dim=-1, keepdim=True).clamp(min=1.0)
    same_nan = torch.isnan(got) == torch.isnan(want)
    bad = (~same_nan.all(dim=-1)) | (torch.nan_to_num(rel, nan=0.0).amax(dim=-1) > AUDIT_TOL)
    n_bad = int(bad.sum().item())
    worst = float(torch.nan_to_num(rel, nan=0.0).max().item())
    _last_audit = {'pixels': m + (before['pixels'] if before else 0), 'differing': n_bad + (before['differing'] if before else 0),
                   'max_rel_diff': max(worst, before['max_rel_diff'] if before else 0.0), 'tolerance': AUDIT_TOL}
    if before and 'worst' in before:
        _last_audit['worst'] = before['worst']
    if n_bad:
        k = int(torch.nan_to_num(rel, nan=float('inf')).amax(dim=-1).argmax().item())
        msg = (f'Newton short cut audit: {n_bad} of {m} sampled pixels differ from the reference\'s fixed iteration count by more than '
               f'{AUDIT_TOL:g} (or in their NaN pattern); worst: pixel {int(idx[k])} counts ({float(s1[k])!r}, {float(s2[k])!r}) '
               f'default {got[k].tolist()} exact {want[k].tolist()}.  Use stop_tol=0 / DEXCT_GN_EXACT=1 and report the tables.')
        _last_audit['worst'] = {'pixel': int(idx[k]), 'counts': [float(s1[k]), float(s2[k])], 'default': got[k].tolist(), 'exact': want[k].tolist()}
        if strict:
            raise GnAuditError(msg)
        import warnings
        warnings.warn(msg, GnAuditWarning, stacklevel=3)
    return _last_audit


def gn_device(g1, g2, i0, mus, n_iters, precision=None, n_polish=N_POLISH, out=None, bin_div=1, mask_max=None,
              mask_frac=0.95, stop_tol=None, out_rc=None, kernel=0, accumulate_stats=False, two_level=None, full_loop=False,
              natural_order=False, blocks_per_cu=0, audit=None, audit_strict=None):
    """g1, g2: device tensors of equal shape; mus: [2, nE] float64; i0: [2, nE] (one spectrum for all
    pixels) or [2, nBins, nE] (pixel p uses row (p // bin_div) % nBins: the reference's general layout).
    ``mask_max``: device float64 scalar (the global maximum of sinogram 1) - pixels with g1 >= mask_frac * max are
    the air pixels get_basismat_sinos zeroes (:204-205); they are written as 0 and not iterated.
    ``stop_tol``: None = default (DEFAULT_STOP_TOL = 1e-12 unless the environment said otherwise at import; the fixed count
    for an ill-posed pair of spectra, see above), 0 = the fixed iteration count exactly.
    ``out_rc=(rows, channels)``: the sinograms are [..., channel, row] (row fastest) and the result is written as
    [..., row, channel, 2], the reference's order, by the kernel itself.  ``kernel``: 0 choose, 1 lane per pixel,
    2 cooperative (dexct_gn_options).  ``two_level``: None = DEFAULT_TWO_LEVEL (see there), True / False.
    ``full_loop``, ``natural_order``, ``blocks_per_cu``: dexct_gn_options.flags / .blocks_per_cu (checking and tuning: results do
    not depend on the last two, and on the first only through stop_tol = 0).  ``audit`` (pixels per million; None =
    DEFAULT_AUDIT_PPM), ``audit_strict``: the sampled audit of the short cut (see DEFAULT_AUDIT_PPM).
    Returns a device tensor of shape g1.shape + (2,) float64 (with ``out_rc``: the last two sinogram dimensions swapped)."""
    lib = _native.load()
    dev = g1.device
    precision = precision or DEFAULT_PRECISION
    if precision not in ('f64', 'mixed'):
        raise ValueError(f'precision {precision!r}')
    if g1.shape != g2.shape or g1.dtype != g2.dtype:
        raise ValueError('the two sinograms must agree in shape and dtype')
    explicit_tol = stop_tol is not None
    if stop_tol is None:
        stop_tol = DEFAULT_STOP_TOL
    stop_tol = float(stop_tol)
    if not stop_tol >= 0.0:
        raise ValueError(f'stop_tol={stop_tol!r}')
    if full_loop:
        stop_tol = 0.0
    shp_i0, shp_mu = (tuple(x.shape) if hasattr(x, 'shape') else np.shape(x) for x in (i0, mus))
    if len(shp_i0) not in (2, 3) or shp_i0[0] != 2 or shp_mu != (2, shp_i0[-1]):
        raise ValueError('i0 must be [2, nE] or [2, nBins, nE] and mus [2, nE]')
    n_bins, n_e = (shp_i0[1] if len(shp_i0) == 3 else 1), shp_i0[-1]
    if n_bins > 1 and precision == 'mixed':
        precision = 'f64'               # mixed precision exists for the shared-spectrum fast path only
    if two_level is None:
        two_level = DEFAULT_TWO_LEVEL
    if two_level is None or two_level is True:
        two_level = 'one'
    if two_level not in (False, 'one', 'start'):
        raise ValueError(f'two_level={two_level!r}')
    applies = (precision == 'f64' and n_bins == 1 and kernel != 2 and 4 <= int(n_iters) <= 254 and n_e >= 48 and stop_tol > 0.0)
    # (the gate is consulted for the default tolerance even with two_level=False: an ill-posed pair runs the fixed count)
    want_gate = applies and (bool(two_level) or not explicit_tol)
    i0_d, mus_d, gate = _device_tables(i0, mus, dev, want_gate, min(stop_tol, 1.0e-12) or 1.0e-12)
    start = None
    mode = 'single'
    if gate is not None:
        if gate['ill_posed'] and not explicit_tol:
            stop_tol, mode = 0.0, 'exact (ill-posed pair)'
        elif two_level and gate['start'] is not None:
            start, mode = gate['start'], (two_level if gate['stats'].get('one_step_share', 0.0) > 0.0 else 'start')
    shape = tuple(g1.shape)
    rows = chans = 0
    if out_rc is not None:
        rows, chans = int(out_rc[0]), int(out_rc[1])
        if shape[-2:] != (chans, rows):
            raise ValueError(f'out_rc={out_rc}: the sinograms must end in [channel={chans}, row={rows}], got {shape}')
        shape = shape[:-2] + (rows, chans)
    a = out if out is not None else torch.empty(shape + (2,), dtype=torch.float64, device=dev)
    if out is not None and (a.numel() != 2 * g1.numel() or a.dtype != torch.float64 or not a.is_contiguous()):
        raise ValueError('out must be a contiguous float64 tensor with two values per pixel')
    global _last_zeroed, _last_ws, _last_events, _last_mode, _last_audit
    is64 = int(g1.dtype == torch.float64)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ws = torch.empty(lib.dexct_gn_workspace_bytes(n_e, n_bins), dtype=torch.uint8, device=dev)
    ws[72:104].zero_()       # executed-iteration, progress, queue and stall counters: defined before anybody polls them
    _last_zeroed = torch.cuda.Event()
    _last_zeroed.record()    # a progress poller on another stream waits for this (never reads uninitialised bytes)
    flags = ((_native.GN_FLAG_FULL_LOOP if full_loop else 0) | (_native.GN_FLAG_NATURAL_ORDER if natural_order else 0)
             | (_native.GN_FLAG_ONE_STEP if mode == 'one' else 0))
    if start is not None:
        opts = _native.gn_options(stop_tol, rows, chans, 1, _native.GN_PASS_SHORTCUT, None, start.data_ptr(), flags, blocks_per_cu)
    else:
        opts = _native.gn_options(stop_tol, rows, chans, kernel, 0, None, None, flags, blocks_per_cu)
    ev[0].record()
    _native.check(lib.dexct_gn_decompose(ptr(g1), ptr(g2), is64, g1.numel(), ptr(i0_d),
                                         ptr(mus_d), n_e, n_bins, int(bin_div), int(n_iters),
                                         int(precision == 'mixed'), int(n_polish), ptr(mask_max), float(mask_frac), ptr(a),
                                         opts, ptr(ws), stream_ptr()),
                  'dexct_gn_decompose')
    ev[1].record()
    _last_ws = (_last_ws + [ws]) if accumulate_stats else [ws]
    _last_events = (_last_events if accumulate_stats else []) + [ev]
    _last_mode = mode
    if not accumulate_stats:
        _last_audit = None
    ppm = DEFAULT_AUDIT_PPM if audit is None else float(audit)
    if ppm > 0.0 and mode in SHORTCUT_MODES:
        _audit(g1, g2, a, i0, mus, n_iters, ppm, DEFAULT_AUDIT_STRICT if audit_strict is None else bool(audit_strict),
               (rows, chans) if out_rc is not None else None, mask_max, mask_frac, merge=accumulate_stats)
    return a


def _progress_lines(ws, n_views, n_bins, done_event, t0, every=20, poll_s=0.05):
    """The reference's progress line ``j / nViews t=...s`` (matdecomp.py:111-112, one per 20 views), driven by the
    kernel's own finished-pixel counter (workspace byte 80), read on a side stream while the kernel runs."""
    import time
    side = torch.cuda.Stream()
    if _last_zeroed is not None:
        side.wait_event(_last_zeroed)
    host = torch.zeros(1, dtype=torch.int64).pin_memory()
    counter = ws[80:88].view(torch.int64)
    n_pix = n_views * max(n_bins, 1)
    next_view = 0
    while True:
        finished = done_event.query()
        with torch.cuda.stream(side):
            host.copy_(counter, non_blocking=True)
        side.synchronize()
        done_pix = int(host.item())
        if not 0 <= done_pix <= n_pix:      # never a value the kernel can have written: ignore
            done_pix = 0
        views_done = n_views if finished else min(done_pix // max(n_bins, 1), n_views)
        while next_view < n_views and next_view <= views_done:
            print(next_view, '/', n_views, f't={time.time() - t0:.2f}s')
            next_view += every
        if finished:
            return
        time.sleep(poll_s)


def optimize_sino(Sino_gg, ee, i0, mus, n_iters, verbose=True, dtype=None, precision=None, stop_tol=None, two_level=None, audit=None,
                  audit_strict=None, **gn_knobs):
    """Newton iterations for every pixel (signature of matdecomp.py:20 / :87).
    ``verbose`` prints the reference's progress line every 20 views (:111-112) from the kernel's finished-pixel
    counter; the drop-in callers below pass verbose=False unless asked (a benchmark should not print).

    Sino_gg [2, nViews, nBins] counts; i0 [2, nBins, nEnergies] (channel-dependent spectra are handled by a
    slower per-lane-table kernel; the tiled spectrum do_matdecomp_gn builds, :151, takes the fast path) or
    [2, nEnergies]; mus [2, nEnergies].
    Returns Sino_aa [nViews, nBins, 2] float64 NumPy.  ``ee`` is unused, as in the reference.
    """
    i0 = np.asarray(i0, dtype=np.float64)
    if i0.ndim == 3 and np.all(i0 == i0[:, :1, :]):
        i0 = i0[:, 0, :]                 # one spectrum tiled over the channels (:151): the fast path
    if i0.ndim == 3 and i0.shape[1] != np.asarray(Sino_gg).shape[2]:
        raise ValueError('i0 has a different number of bins than the sinogram')
    dev = device()
    g = _as_device_counts(np.asarray(Sino_gg), dev)
    import time
    t0 = time.time()
    a = gn_device(g[0], g[1], i0, np.asarray(mus, dtype=np.float64), n_iters, precision, stop_tol=stop_tol, two_level=two_level, audit=audit,
                  audit_strict=audit_strict, **gn_knobs)          # (kernel, full_loop, natural_order, blocks_per_cu: see gn_device)
    if verbose:
        done = torch.cuda.Event()
        done.record()
        _progress_lines(_last_ws[-1], int(g.shape[1]), int(g[0].numel() // max(int(g.shape[1]), 1)), done, t0)
    return to_host(a)


optimize_sino_cpu = optimize_sino   # the reference's NumPy twin (:87); same engine here


def decomposition_tables(ct, spec1, spec2):
    """Union energy grid and effective spectra exactly as matdecomp.py:140-150 builds them."""
    ee = np.array(sorted(set(np.append(spec1.E, spec2.E))))
    dE = np.append([ee[0]], ee[1:] - ee[:-1])          # 1st energy bin is 0 to E[0]
    detresponse = np.interp(ee, ct.det_E, ct.det_eta_E)
    if ct.eid:
        detresponse = detresponse * ee
    i0 = np.stack([np.interp(ee, spec1.E, spec1.I0) * detresponse * dE,
                   np.interp(ee, spec2.E, spec2.I0) * detresponse * dE])
    mus = np.stack([xc.mixatten(matcomp1, ee), xc.mixatten(matcomp2, ee)])   # mass attenuation (:158)
    return ee, i0, mus


def do_matdecomp_gn(ct, sino1, sino2, spec1, spec2, n_iters, precision=None, stop_tol=None, two_level=None, audit=None, audit_strict=None):
    """[N_proj, N_channels, 2] density line integrals (matdecomp.py:130-164)."""
    _, i0, mus = decomposition_tables(ct, spec1, spec2)
    dev = device()
    g1 = _as_device_counts(sino1, dev)
    g2 = _as_device_counts(sino2, dev).to(g1.dtype)
    a = gn_device(g1, g2, i0, mus, n_iters, precision, stop_tol=stop_tol, two_level=two_level, audit=audit, audit_strict=audit_strict)
    return a if isinstance(sino1, torch.Tensor) else to_host(a)


_PIPE_CHUNKS = 8                 # view chunks of the pipelined host boundary (tools/probes/boundary_gn.py at configs[2] size,
                                 # Newton kernel with its run queue: 2 / 3 / 4 / 6 / 8 / 12 / 16 chunks 0.878 / 0.856 / 0.848 /
                                 # 0.834 / 0.832 / 0.832 / 0.836 s, the plain sequence 0.975 s.  With the static runs of rounds
                                 # 1-2 short launches cost the kernel efficiency and 3 chunks were the optimum.)
_PIPE_MIN_PIXELS = 1 << 24       # below 16.8 M pixels (64 MiB per float32 sinogram) the plain sequence is as fast


def _basismat_sinos_pipelined(lib, dev, s1, s2, i0, mus, n_iters, mask_thresh, precision, strict, stop_tol, two_level=None, audit=None,
                              audit_strict=None):
    """get_basismat_sinos for NumPy sinograms of benchmark size: sinogram 1 goes to the device first (the mask needs its
    global maximum, matdecomp.py:195-196), then per view chunk: sinogram 2's chunk arrives on an upload stream, the Newton
    kernel runs on it, and the finished chunk leaves for page-locked host memory on a download stream while the next chunk
    computes.  Same kernels on the same pixels as the plain sequence: bit-identical results."""
    a1 = np.ascontiguousarray(s1)
    a2 = np.ascontiguousarray(s2)
    npdt = np.float32 if a1.dtype == np.float32 else np.float64
    a1, a2 = a1.astype(npdt, copy=False), a2.astype(npdt, copy=False)
    # the inputs are locked for the time of the call: their uploads run as DMA, whoever allocated them
    with locked_arrays(lib, [a1, a2], dev.index or 0):
        comp, up, down = side_streams(dev)
        comp.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(comp):
            return _pipeline(lib, dev, a1, a2, i0, mus, n_iters, mask_thresh, precision, strict, stop_tol, two_level, audit, audit_strict,
                             up, down)


def _pipeline(lib, dev, a1, a2, i0, mus, n_iters, mask_thresh, precision, strict, stop_tol, two_level, audit, audit_strict, copy, down):
    dt = torch.float32 if a1.dtype == np.float32 else torch.float64
    h1 = torch.from_numpy(a1)
    h2 = torch.from_numpy(a2)
    n_views = h1.shape[0]
    main = torch.cuda.current_stream()
    # (copy / down: the upload and the download stream - PCIe carries both directions at once, and a result chunk must not queue
    # behind the uploads of ALL later input chunks; the kernels run on the current stream, _device.side_streams)
    g1 = h1.to(dev, non_blocking=True)                      # (one DMA: the array is page-locked)
    g2 = torch.empty_like(g1)
    gmax = torch.empty((), dtype=torch.float64, device=dev)
    _native.check(lib.dexct_reduce_max(ptr(g1), int(dt == torch.float64), g1.numel(), ptr(gmax), stream_ptr()),
                  'dexct_reduce_max')
    a = torch.empty(tuple(g1.shape) + (2,), dtype=torch.float64, device=dev)
    bounds = [_shard.split(n_views, k, _PIPE_CHUNKS) for k in range(_PIPE_CHUNKS)]
    # where the results land: a block of host memory that is touched (a new one) and locked chunk by chunk while the pipeline
    # runs (_device.LazyPinnedResult)
    lazy = host = None
    if pool_wanted(16 * g1.numel()):
        row = 16 * int(np.prod(g1.shape[1:]))
        lazy = LazyPinnedResult(lib, tuple(g1.shape) + (2,), np.float64, [b * row for b, _ in bounds] + [n_views * row], dev.index or 0)
    else:
        host = pinned_empty(tuple(g1.shape) + (2,), torch.float64)
    arrived = []
    copy.wait_stream(main)
    with torch.cuda.stream(copy):                           # all of sinogram 2 is queued at once, chunk by chunk
        for b, e in bounds:
            g2[b:e].copy_(h2[b:e], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy)
            arrived.append(ev)
    # Stacked fans arrive as [view][row][channel].  The kernel is ~8 % faster on [view][channel][row] (z-neighbours are
    # nearly the same problem, so the lanes of a wave end together; tools/probes/overlap.py: 0.87 vs 0.80 s), which is also
    # the order the projection writes: transpose in (2 x 1.7 ms) and solve; the kernel writes its results in the reference's
    # order itself (out_rc: 4 x 16 tiles collected in LDS - round 3 needed a 2.3 ms transpose pass over 6.5 GB for it).
    row_fastest = g1.dim() == 3 and g1.shape[1] >= 8
    if row_fastest:
        nR, nC = int(g1.shape[1]), int(g1.shape[2])
        t1 = torch.empty((n_views, nC, nR), dtype=dt, device=dev)
        t2 = torch.empty_like(t1)
        eb = 4 if dt == torch.float32 else 8
    global _last_ws, _last_events, _last_audit
    _last_ws, _last_events, _last_audit = [], [], None
    for k, ((b, e), ev) in enumerate(zip(bounds, arrived)):
        main.wait_event(ev)
        kw = dict(out=a[b:e], mask_max=gmax, mask_frac=float(mask_thresh), stop_tol=stop_tol, accumulate_stats=True,
                  two_level=two_level, audit=audit, audit_strict=audit_strict)
        if row_fastest:
            for src, dst in ((g1, t1), (g2, t2)):
                _native.check(lib.dexct_transpose_batched(ptr(src[b:e]), ptr(dst[b:e]), e - b, nR, nC, eb, stream_ptr()),
                              'dexct_transpose_batched')
            gn_device(t1[b:e], t2[b:e], i0, mus, n_iters, precision, out_rc=(nR, nC), **kw)
        else:
            gn_device(g1[b:e], g2[b:e], i0, mus, n_iters, precision, **kw)
        done = torch.cuda.Event()
        done.record(main)
        with torch.cuda.stream(down):
            down.wait_event(done)
            if lazy is not None:
                lazy.download(k, ptr(a[b:e]), down)
            else:
                host[b:e].copy_(a[b:e], non_blocking=True)
    main.wait_stream(copy)
    main.wait_stream(down)
    main.synchronize()
    down.synchronize()
    out = lazy.finish() if lazy is not None else host.numpy()          # (every copy has landed)
    if strict:
        bad = ~torch.isfinite(a).all(dim=-1)
        n_bad = int(bad.sum().item())
        if n_bad:
            first = bad.flatten().nonzero()[:1].flatten().tolist()
            raise SingularHessianError(f'Singular matrix: {n_bad} pixel(s) outside the air mask ended non-finite '
                                       f'after {n_iters} Newton iterations (first flat index on this rank: {first})')
    return out[..., 0], out[..., 1]


def get_basismat_sinos(ct, sino_raw_1, sino_raw_2, spec1, spec2, n_iters=30, mask_thresh=0.95, precision=None,
                       strict=False, verbose=False, stop_tol=None, two_level=None, audit=None, audit_strict=None):
    """Basis-material sinograms (matdecomp.py:167-207): air mask from sinogram 1
    (``>= mask_thresh * max``), Newton decomposition, masked pixels set to exactly 0.

    NumPy in -> NumPy float64 out (two views of one buffer, like the reference); device tensors
    in -> device tensors out.  Under torch.distributed the inputs are either each rank's own view
    shard or the full gathered sinograms (then each rank decomposes its own views and the result
    is all-gathered); the mask threshold always uses the all-reduced global maximum.
    ``verbose=True`` prints the reference's progress line every 20 views (matdecomp.py:111-112; the reference always
    prints it) from the kernel's finished-pixel counter.
    ``stop_tol``: None = the default tolerance stop (1e-12 relative step with a contraction check, include/dexct.h
    dexct_gn_options; DEXCT_GN_EXACT=1 in the environment makes the default exact; a pair of spectra the calibration finds
    ill-posed - the MV / kV pairs - runs the fixed count by default); 0 = the reference's fixed iteration
    count bit for bit (matdecomp.py:114); the two agree to ~1e-14 on converging pixels and are identical on the others.
    ``audit`` (pixels per million; None = DEXCT_GN_AUDIT, 0 = off), ``audit_strict``: re-solve a Philox-chosen sample with the
    fixed count in the same call and warn (GnAuditWarning) or raise (GnAuditError) when the default's result differs by more
    than 1e-12 or in its NaN pattern.
    ``strict=True``: raise ``SingularHessianError`` (a ``numpy.linalg.LinAlgError``, what :125 raises) if a pixel
    outside the air mask ends non-finite; the default returns the inf/NaN in place, as documented above.
    """
    lib = _native.load()
    dev = device()
    _, i0, mus = decomposition_tables(ct, spec1, spec2)
    rank, world = _shard.world()
    n_views = getattr(ct, 'N_proj', None)
    full_in = world > 1 and n_views is not None and sino_raw_1.shape[0] == n_views
    if full_in:                       # every rank holds the gathered sinograms: take this rank's views
        vb, ve = _shard.split(n_views, rank, world)
        sino_raw_1, sino_raw_2 = sino_raw_1[vb:ve], sino_raw_2[vb:ve]
    # Large NumPy inputs on one process: pipeline the host boundary (sinogram 2 arrives and the results leave in view
    # chunks while the Newton kernel works on the chunk in between) instead of copy-in, compute, copy-out in sequence
    if (world == 1 and not verbose and not isinstance(sino_raw_1, torch.Tensor) and np.ndim(sino_raw_1) >= 2
            and np.shape(sino_raw_1)[0] >= 2 * _PIPE_CHUNKS and np.size(sino_raw_1) >= _PIPE_MIN_PIXELS
            and np.shape(sino_raw_1) == np.shape(sino_raw_2)):
        return _basismat_sinos_pipelined(lib, dev, sino_raw_1, sino_raw_2, i0, mus, n_iters, mask_thresh, precision, strict, stop_tol,
                                         two_level, audit, audit_strict)
    g1 = _as_device_counts(sino_raw_1, dev)
    g2 = _as_device_counts(sino_raw_2, dev).to(g1.dtype)
    is64 = int(g1.dtype == torch.float64)
    gmax = torch.empty((), dtype=torch.float64, device=dev)
    _native.check(lib.dexct_reduce_max(ptr(g1), is64, g1.numel(), ptr(gmax), stream_ptr()), 'dexct_reduce_max')
    gmax = _shard.global_max(gmax)
    # the mask is applied inside the kernel (threshold read from the device scalar: no host round trip)
    import time
    t0 = time.time()
    a = gn_device(g1, g2, i0, mus, n_iters, precision, mask_max=gmax, mask_frac=float(mask_thresh), stop_tol=stop_tol,
                  two_level=two_level, audit=audit, audit_strict=audit_strict)
    if verbose and rank == 0 and g1.dim() >= 2:
        done = torch.cuda.Event()
        done.record()
        _progress_lines(_last_ws[-1], int(g1.shape[0]), int(g1.numel() // max(int(g1.shape[0]), 1)), done, t0)
    if strict:
        bad = ~torch.isfinite(a).all(dim=-1)          # masked pixels are exactly 0, hence finite
        n_bad = int(bad.sum().item())
        if world > 1:
            t = torch.tensor(float(n_bad), dtype=torch.float64,
                             device='cpu' if torch.distributed.get_backend() == 'gloo' else dev)
            torch.distributed.all_reduce(t)            # every rank raises, or none
            n_bad = int(t.item())
        if n_bad:
            first = bad.flatten().nonzero()[:1].flatten().tolist()
            raise SingularHessianError(f'Singular matrix: {n_bad} pixel(s) outside the air mask ended non-finite '
                                       f'after {n_iters} Newton iterations (first flat index on this rank: {first})')
    if full_in:
        a = _shard.gather_views(a, n_views, view_dim=0, tag='get_basismat_sinos', mode=_shard.dropin_mode())     # a new tensor: the caller owns it
    if isinstance(sino_raw_1, torch.Tensor):
        return a[..., 0], a[..., 1]
    a = to_host(a)          # page-locked: one DMA; the two results are views of this one buffer, like the reference's
    return a[..., 0], a[..., 1]
