"""ORACLE (test infrastructure): ctypes access to oracle/_build/libdexct_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Build with ``make -C oracle`` (``__graft_entry__.build()`` does it).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, '_build', 'libdexct_oracle.so')


class FanGeom(C.Structure):
    """Mirror of dexct_fan_geom (include/dexct.h)."""
    _fields_ = [('n_views', C.c_int32), ('n_channels', C.c_int32), ('n_rows', C.c_int32),
                ('z_first', C.c_int32), ('nx', C.c_int32), ('ny', C.c_int32), ('nz', C.c_int32),
                ('pad_', C.c_int32), ('dx', C.c_double), ('dy', C.c_double), ('dz', C.c_double),
                ('sid', C.c_double), ('sdd', C.c_double)]


PLAN_DTYPE = np.dtype([('V0', '<i8'), ('SV', '<i8'), ('i_first', '<i4'), ('n_slabs', '<i4'),
                       ('kf', '<f4'), ('len_per_u', '<f4'), ('chord_u', '<f4'), ('flags', '<u4')])
assert PLAN_DTYPE.itemsize == 40

_lib = None


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, 'dexct_oracle.c')):
        subprocess.check_call(['make', '-C', HERE, '-s'])


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_siddon_classic_ray.restype = C.c_int
        _lib.orc_dda_ray.restype = C.c_int
        _lib.orc_max_threads.restype = C.c_int
        _lib.orc_count_segments.restype = C.c_longlong
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_geom(n_views, n_channels, n_rows, z_first, nx, ny, nz, dx, dy, dz, sid, sdd):
    return FanGeom(n_views, n_channels, n_rows, z_first, nx, ny, nz, 0, dx, dy, dz, sid, sdd)


def classic_ray(g, view_cs, chan_cs, view, chan):
    n = g.nx + g.ny + 4
    vox = np.zeros(n, np.int32)
    ln = np.zeros(n, np.float64)
    k = lib().orc_siddon_classic_ray(C.byref(g), _p(view_cs), _p(chan_cs), int(view), int(chan), n, _p(vox), _p(ln))
    return vox[:k].copy(), ln[:k].copy()


def plan(g, view_cs, chan_cs, view_begin, view_end):
    out = np.zeros((view_end - view_begin) * g.n_channels, PLAN_DTYPE)
    lib().orc_plan(C.byref(g), _p(view_cs), _p(chan_cs), int(view_begin), int(view_end), _p(out))
    return out


def dda_ray(g, plan_entry, z):
    n = 2 * max(g.nx, g.ny) + 4
    vox = np.zeros(n, np.int32)
    ln = np.zeros(n, np.float32)
    pe = np.array([plan_entry], PLAN_DTYPE)
    k = lib().orc_dda_ray(C.byref(g), _p(pe), int(z), n, _p(vox), _p(ln))
    return vox[:k].copy(), ln[:k].copy()


def project_classic(g, view_cs, chan_cs, view_begin, view_end, vol, mu, w, want_pathlen=False, n_threads=1):
    mu = np.ascontiguousarray(mu, np.float64)
    w = np.ascontiguousarray(w, np.float64)
    vol = np.ascontiguousarray(vol, np.uint8)
    n_mat, n_e = mu.shape
    n_spec = w.shape[0]
    nV = view_end - view_begin
    counts = np.zeros((n_spec, nV, g.n_rows, g.n_channels), np.float64)
    pl = np.zeros((nV, g.n_rows, g.n_channels, n_mat), np.float64) if want_pathlen else None
    lib().orc_project_classic(C.byref(g), _p(view_cs), _p(chan_cs), int(view_begin), int(view_end), _p(vol),
                              n_mat, n_e, n_spec, _p(mu), _p(w), _p(counts), _p(pl), int(n_threads))
    return (counts, pl) if want_pathlen else counts


def project_dda(g, view_cs, chan_cs, view_begin, view_end, vol, mu, w, want_pathlen=False, n_threads=1):
    mu = np.ascontiguousarray(mu, np.float64)
    w = np.ascontiguousarray(w, np.float64)
    vol = np.ascontiguousarray(vol, np.uint8)
    n_mat, n_e = mu.shape
    n_spec = w.shape[0]
    nV = view_end - view_begin
    counts = np.zeros((n_spec, nV, g.n_rows, g.n_channels), np.float64)
    pl = np.zeros((nV, g.n_rows, g.n_channels, n_mat), np.float32) if want_pathlen else None
    lib().orc_project_dda(C.byref(g), _p(view_cs), _p(chan_cs), int(view_begin), int(view_end), _p(vol),
                          n_mat, n_e, n_spec, _p(mu), _p(w), _p(counts), _p(pl), int(n_threads))
    return (counts, pl) if want_pathlen else counts


def gn_decompose(g1, g2, i0, mus, n_iters, n_threads=1):
    g1 = np.ascontiguousarray(g1, np.float64)
    g2 = np.ascontiguousarray(g2, np.float64)
    i0 = np.ascontiguousarray(i0, np.float64)
    mus = np.ascontiguousarray(mus, np.float64)
    out = np.zeros(g1.shape + (2,), np.float64)
    lib().orc_gn_decompose(_p(g1), _p(g2), C.c_long(g1.size), _p(i0), _p(mus), int(i0.shape[-1]), int(n_iters),
                           _p(out), int(n_threads))
    return out


def max_threads():
    return lib().orc_max_threads()


def count_segments(g, plan_table, n_threads=0):
    plan_table = np.ascontiguousarray(plan_table)
    return lib().orc_count_segments(C.byref(g), _p(plan_table), C.c_long(plan_table.size),
                                    int(n_threads or max_threads()))


def project_cone(g, view_cs, chan_cs, view_begin, view_end, row_z, src_z, vol, mu, w, dda=False, n_threads=1):
    """Cone-beam projection: classic float64 3-D Siddon (dda=False) or the kernel-arithmetic mirror (dda=True).
    Returns counts [S, nV, rows, ch] float64 and pathlen [nV, rows, ch, M] (float64 classic / float32 mirror)."""
    mu = np.ascontiguousarray(mu, np.float64)
    w = np.ascontiguousarray(w, np.float64)
    vol = np.ascontiguousarray(vol, np.uint8)
    row_z = np.ascontiguousarray(row_z, np.float64)
    n_mat, n_e = mu.shape
    n_spec = w.shape[0]
    nV = view_end - view_begin
    counts = np.zeros((n_spec, nV, g.n_rows, g.n_channels), np.float64)
    plc = np.zeros((nV, g.n_rows, g.n_channels, n_mat), np.float64) if not dda else None
    pld = np.zeros((nV, g.n_rows, g.n_channels, n_mat), np.float32) if dda else None
    lib().orc_project_cone(C.byref(g), _p(view_cs), _p(chan_cs), int(view_begin), int(view_end), _p(row_z),
                           C.c_double(src_z), _p(vol), n_mat, n_e, n_spec, _p(mu), _p(w), _p(counts), _p(plc), _p(pld),
                           int(bool(dda)), int(n_threads))
    return counts, (pld if dda else plc)
