"""ctypes binding of libdexct_hip.so (include/dexct.h).  No fallback: if the library is missing or
does not export the ABI the import of the compute modules fails loudly."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libdexct_hip.so')
ABI_VERSION = 6

# every entry point include/dexct.h declares
SYMBOLS = ['dexct_strerror', 'dexct_abi_version', 'dexct_last_hip_error', 'dexct_volume_layouts', 'dexct_fan_plan',
           'dexct_siddon_project', 'dexct_siddon_trace', 'dexct_gn_decompose', 'dexct_gn_apply_mask', 'dexct_gn_model_sums',
           'dexct_reduce_max', 'dexct_transpose_batched', 'dexct_fbp_filter', 'dexct_fbp_backproject',
           'dexct_add_noise', 'dexct_volume_groups', 'dexct_siddon_project_grouped', 'dexct_cone_project',
           'dexct_cone_layout', 'dexct_cone_project_rows', 'dexct_volume_pack2', 'dexct_siddon_project_packed', 'dexct_volume_groups_pack2',
           'dexct_siddon_project_grouped_packed', 'dexct_poisson_detect', 'dexct_vmi', 'dexct_label_moments', 'dexct_fdk_backproject', 'dexct_sino_allgather', 'dexct_sino_gather', 'dexct_transpose_log', 'dexct_host_pin', 'dexct_host_touch', 'dexct_host_unpin', 'dexct_download',
           'dexct_volume_ids', 'dexct_volume_remap', 'dexct_fbp_parker', 'dexct_sino_log', 'dexct_cone_layout_groups', 'dexct_cone_project_grouped',
           'dexct_cone_layout_bytes', 'dexct_gn_workspace_bytes']


class FanGeom(C.Structure):
    """dexct_fan_geom"""
    _fields_ = [('n_views', C.c_int32), ('n_channels', C.c_int32), ('n_rows', C.c_int32), ('z_first', C.c_int32),
                ('nx', C.c_int32), ('ny', C.c_int32), ('nz', C.c_int32), ('pad_', C.c_int32),
                ('dx', C.c_double), ('dy', C.c_double), ('dz', C.c_double), ('sid', C.c_double), ('sdd', C.c_double)]


class LogOut(C.Structure):
    """dexct_log_out: the optional second output of get_sino, sino_log = ln(air / counts)"""
    _fields_ = [('sino_log', C.c_void_p), ('air', C.c_float * 4)]


def log_out(sino_log_ptr, air):
    """byref-able dexct_log_out, or None when no log sinogram is wanted."""
    if sino_log_ptr is None:
        return None
    lo = LogOut()
    lo.sino_log = sino_log_ptr
    for k in range(4):
        lo.air[k] = float(air[k]) if k < len(air) else 1.0
    return C.byref(lo)


class Noise(C.Structure):
    """dexct_noise (ABI 6): quantum noise drawn by the projection kernel itself"""
    _fields_ = [('seed', C.c_uint64), ('sample', C.c_int32), ('reserved_', C.c_int32)]


def noise(seed, sample=True):
    """byref-able dexct_noise, or None (``seed`` None): the kernel samples counts = max(signal + sqrt(variance) z, 1e-20)."""
    if seed is None:
        return None
    return C.byref(Noise(int(seed) & (2 ** 64 - 1), 1 if sample else 0, 0))


class GnOptions(C.Structure):
    """dexct_gn_options (ABI 5)"""
    _fields_ = [('stop_tol', C.c_double), ('out_rows', C.c_int32), ('out_channels', C.c_int32), ('kernel', C.c_int32),
                ('gn_pass', C.c_int32), ('iterations', C.c_void_p), ('start', C.c_void_p), ('flags', C.c_int32),
                ('blocks_per_cu', C.c_int32)]


GN_PASS_COUNT, GN_PASS_SHORTCUT = 1, 2
GN_FLAG_FULL_LOOP, GN_FLAG_NATURAL_ORDER, GN_FLAG_ONE_STEP = 1, 2, 4


def gn_options(stop_tol=None, out_rows=0, out_channels=0, kernel=0, gn_pass=0, iterations=None, start=None, flags=0,
               blocks_per_cu=0):
    """byref-able dexct_gn_options; ``stop_tol=None`` asks for the library default (a negative value in the struct).
    ``gn_pass`` / ``iterations`` (device address of n_pix bytes: the step counts of GN_PASS_COUNT) / ``start`` (device address of
    the table of the reference's fixed points: GN_PASS_SHORTCUT): the Newton short cut; ``flags``: GN_FLAG_*;
    ``blocks_per_cu``: workgroups per CU of the queue kernels, 0 = default (include/dexct.h)."""
    o = GnOptions(-1.0 if stop_tol is None else float(stop_tol), int(out_rows), int(out_channels), int(kernel), int(gn_pass),
                  iterations, start, int(flags), int(blocks_per_cu))
    return C.byref(o)


PLAN_BYTES = 40   # sizeof(dexct_ray_plan)


class DexctError(RuntimeError):
    pass


_lib = None


def load():
    """Load the HIP library; raises if it is absent (build with __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DexctError(f'{LIB_PATH} not found: the HIP extension is not built '
                         f'(run `python -c "import __graft_entry__ as g; g.build()"` or `make -C dex-ct-sim_amd/csrc`). '
                         f'There is no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name in SYMBOLS:
        if not hasattr(lib, name):
            raise DexctError(f'{LIB_PATH} does not export {name}')
    lib.dexct_strerror.restype = C.c_char_p
    lib.dexct_strerror.argtypes = [C.c_int]
    if lib.dexct_abi_version() != ABI_VERSION:
        raise DexctError(f'ABI version mismatch: library {lib.dexct_abi_version()}, binding {ABI_VERSION}')
    vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    lib.dexct_volume_layouts.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    lib.dexct_volume_ids.argtypes = [vp, i64, vp, vp]
    lib.dexct_volume_remap.argtypes = [vp, i64, C.POINTER(C.c_uint8), vp]
    lib.dexct_fan_plan.argtypes = [C.POINTER(FanGeom), vp, vp, i32, i32, vp, vp]
    lib.dexct_siddon_project.argtypes = [C.POINTER(FanGeom), vp, i32, i32, vp, vp, vp, i32, i32, i32, vp, vp, vp,
                                         vp, i32, i32, vp, vp, vp, vp]
    lib.dexct_sino_log.argtypes = [vp, C.POINTER(C.c_float), i32, i64, vp, vp]
    lib.dexct_add_noise.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, C.c_uint64, vp]
    lib.dexct_volume_groups.argtypes = [vp, i64, i32, vp, vp]
    lib.dexct_vmi.argtypes = [vp, vp, i64, f64, f64, f64, i32, vp, vp]
    lib.dexct_label_moments.argtypes = [vp, vp, vp, i64, i32, vp, vp]
    lib.dexct_sino_allgather.argtypes = [vp, vp, i64, vp, vp]
    lib.dexct_sino_gather.argtypes = [vp, vp, C.POINTER(i64), C.POINTER(i64), i32, i32, i32, vp, vp]
    lib.dexct_host_pin.argtypes = [vp, i64, i32]
    lib.dexct_host_unpin.argtypes = [vp, i32]
    lib.dexct_host_touch.argtypes = [vp, i64, i32]
    lib.dexct_download.argtypes = [vp, vp, i64, vp]
    lib.dexct_fdk_backproject.argtypes = [vp, vp, vp, i32, i32, i32, f64, f64, f64, f64, f64, f64, f64, i32, f64, i32,
                                          f64, f64, vp, vp]
    lib.dexct_poisson_detect.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, C.c_uint64, vp, vp]
    lib.dexct_cone_project.argtypes = [C.POINTER(FanGeom), vp, vp, vp, vp, f64, f64, i32, i32, vp, vp, i32, i32, i32, vp, vp,
                                       vp, vp, vp, vp, vp, vp, vp]
    lib.dexct_volume_pack2.argtypes = [vp, i64, vp, vp]
    lib.dexct_volume_groups_pack2.argtypes = [vp, i64, i32, vp, vp]
    lib.dexct_siddon_project_packed.argtypes = [C.POINTER(FanGeom), vp, i32, i32, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp,
                                                vp, vp, vp]
    lib.dexct_cone_layout.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.dexct_cone_project_rows.argtypes = [C.POINTER(FanGeom), vp, vp, vp, vp, f64, f64, i32, i32, vp, i32, i32, i32, vp, vp, vp,
                                            vp, vp, vp, vp, vp, vp]
    lib.dexct_cone_layout_groups.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    lib.dexct_cone_project_grouped.argtypes = [C.POINTER(FanGeom), vp, vp, vp, vp, f64, f64, i32, i32, vp, i32, i32, i32, vp, vp, vp,
                                               vp, vp, vp, vp, vp, vp, vp]
    lib.dexct_cone_layout_bytes.argtypes = [i32, i32, i32]
    lib.dexct_cone_layout_bytes.restype = i64
    lib.dexct_siddon_project_grouped.argtypes = [C.POINTER(FanGeom), vp, i32, i32, vp, i32, i32, i32, vp, vp, vp, vp, vp,
                                                 i32, vp, vp, vp, vp, vp]
    lib.dexct_siddon_project_grouped_packed.argtypes = lib.dexct_siddon_project_grouped.argtypes
    lib.dexct_transpose_batched.argtypes = [vp, vp, i64, i32, i32, i32, vp]
    lib.dexct_transpose_log.argtypes = [vp, vp, vp, C.POINTER(C.c_float), i32, i64, i32, i32, vp]
    lib.dexct_fbp_filter.argtypes = [vp, vp, vp, i64, i32, f64, vp, vp]
    lib.dexct_fbp_parker.argtypes = [vp, i32, i32, i32, f64, f64, i32, i32, vp, vp]
    lib.dexct_fbp_backproject.argtypes = [vp, vp, i32, i32, i32, f64, f64, f64, i32, f64, vp, vp]
    lib.dexct_siddon_trace.argtypes = [C.POINTER(FanGeom), vp, vp, i32, i32, vp, vp, vp, vp]
    lib.dexct_gn_decompose.argtypes = [vp, vp, i32, i64, vp, vp, i32, i32, i32, i32, i32, i32, vp, f64, vp, C.POINTER(GnOptions), vp, vp]
    lib.dexct_gn_workspace_bytes.argtypes = [i32, i32]
    lib.dexct_gn_workspace_bytes.restype = i64
    lib.dexct_gn_apply_mask.argtypes = [vp, i32, i64, f64, vp, vp]
    lib.dexct_gn_model_sums.argtypes = [vp, i64, vp, vp, i32, vp, vp, vp, vp]
    lib.dexct_reduce_max.argtypes = [vp, i32, i64, vp, vp]
    for name in SYMBOLS[3:-2]:
        getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        lib = load()
        msg = lib.dexct_strerror(code).decode()
        extra = f' (hipError {lib.dexct_last_hip_error()})' if code == -3 else ''
        raise DexctError(f'{what}: {msg}{extra}')
