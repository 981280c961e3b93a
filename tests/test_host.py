"""Host-side mirror of the reference interface (no GPU, no HIP calls)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from conftest import INPUT, ROOT


def test_import_names_and_aliases():
    import dex_ct_sim_amd as dx
    assert dx.ScannerGeometry is dx.FanBeamGeometry and dx.Phantom is dx.VoxelPhantom
    assert dx.Spectrum is dx.xRaySpectrum
    for n in ('read_parameter_file', 'get_sino', 'get_basismat_sinos'):
        assert n in dx.__all__


def test_spectrum_files_half_split():
    import dex_ct_sim_amd as dx
    for sid, n in (('80kV', 140), ('120kV', 140), ('140kV', 140), ('detunedMV', 100), ('6MV', 100)):
        s = dx.xRaySpectrum(os.path.join(INPUT, 'spectrum', f'{sid}_1mGy_float32.bin'), sid)
        assert s.E.shape == (n,) and s.I0.shape == (n,) and np.all(np.diff(s.E) > 0) and s.name == sid
    s = dx.xRaySpectrum(os.path.join(INPUT, 'spectrum', '80kV_1mGy_float32.bin'), '80kV')
    assert s.E[0] == 1.0 and s.E[-1] == 140.0 and s.I0[100:].sum() == 0.0
    tot = s.I0.sum()
    s.rescale_counts(0.5)
    assert np.isclose(s.I0.sum(), 0.5 * tot)


def test_geometry_matches_params_defaults():
    import dex_ct_sim_amd as dx
    ct = dx.FanBeamGeometry(N_channels=800, N_proj=1200, gamma_fan=0.8230337, SID=60.0, SDD=100.0, h_iso=1.0,
                            eid=True, detector_file=os.path.join(INPUT, 'detector', 'eta_eid_mv.bin'))
    assert ct.det_E.shape == (6000,) and ct.det_E[0] == 1.0 and ct.det_E[-1] == 6000.0
    assert np.isclose(ct.A_iso, 60.0 * 0.8230337 / 800 * 1.0)
    assert ct.view_cs().shape == (1200, 2) and ct.chan_cs().shape == (800, 2)
    assert np.isclose(ct.gammas.sum(), 0.0, atol=1e-12) and np.isclose(ct.thetas[-1], 2 * np.pi * 1199 / 1200)
    r = ct.detector_response(np.array([50.0, 100.0]))
    assert np.allclose(r, np.interp([50.0, 100.0], ct.det_E, ct.det_eta_E) * [50.0, 100.0])
    pcd = dx.FanBeamGeometry(eid=False, detector_file=os.path.join(INPUT, 'detector', 'eta_pcd_Si_30mm.bin'))
    assert pcd.det_E.shape == (5999,) and not pcd.eid


def test_parameter_file_reference_keys(tmp_path):
    """The reference's own params.txt keys (input/params.txt:1-37), with a phantom written here."""
    import dex_ct_sim_amd as dx
    vol = np.zeros((1, 16, 16), np.uint8)
    vol[0, 4:12, 4:12] = 1
    vol.tofile(tmp_path / 'ph.bin')
    (tmp_path / 'mats.csv').write_text('id,name,density,composition\n0,air,0.0012,N(75.5)O(23.2)Ar(1.3)\n'
                                       '1,water,1.0,H(11.2)O(88.8)\n')
    p = {"RUN_ID": "t", "forward_project": True, "back_project": True, "phantom_type": "voxel",
         "phantom_id": "box", "phantom_filename": str(tmp_path / 'ph.bin'), "matcomp_filename": str(tmp_path / 'mats.csv'),
         "Nx": 16, "Ny": 16, "Nz": 1, "dx": 0.1, "dy": 0.1, "dz": 0.1, "z_index": 0, "scanner_geometry": "fan_beam",
         "SID": 60.0, "SDD": 100.0, "N_channels": 800, "N_projections": 1200, "fan_angle_total": 0.8230337,
         "rotation_angle_total": 6.283185, "detector_px_height": 1.0, "detector_mode": "eid",
         "detector_filename": os.path.join(INPUT, 'detector', 'eta_eid_mv.bin'), "spectrum_id": "NA",
         "spectrum_filename": "NA", "N_photons_per_cm2_per_scan": "NA", "N_recon_matrix": 512, "FOV_recon": 50.0,
         "ramp_filter_percent_Nyquist": 0.8}
    f = tmp_path / 'params.txt'
    f.write_text(json.dumps(p))
    runs = dx.read_parameter_file(str(f))
    assert len(runs) == 1
    run_id, do_fp, do_bp, ct, ph, spec, N, FOV, ramp = runs[0]
    assert (run_id, do_fp, do_bp, spec, N, FOV, ramp) == ('t', True, True, None, 512, 50.0, 0.8)
    assert ct.N_channels == 800 and ct.N_proj == 1200 and ct.eid
    assert ph.volume.shape == (1, 16, 16) and ph.n_materials == 2 and ph.materials[1].name == 'water'
    assert ph.M_mono(60.0).shape == (16, 16)
    f.write_text(json.dumps([p, dict(p, RUN_ID='u', detector_mode='pcd')]))     # a list of runs
    runs = dx.read_parameter_file(str(f))
    assert [r[0] for r in runs] == ['t', 'u'] and not runs[1][3].eid


def test_mixatten_surface(golden):
    from dex_ct_sim_amd import matdecomp as md, xcompy
    E = golden['xc_E']
    assert np.array_equal(xcompy.mixatten(md.matcomp1, E), golden['xc_tissue'])
    assert np.array_equal(xcompy.mixatten(md.matcomp2, E), golden['xc_bone'])
    assert xcompy.mixatten('H(11.2)O(88.8)', np.array([60.0])).shape == (1,)
    assert md.density1 == golden['const_density'][0] and md.density2 == golden['const_density'][1]
    xcompy.register_table('Xx(100)', [10.0, 100.0], [2.0, 0.2])
    assert np.isclose(xcompy.mixatten('Xx(100)', np.array([10.0]))[0], 2.0)
    with pytest.raises(KeyError):
        xcompy.mixatten('Qq(100)', E)


def test_merged_tables_keep_each_spectrum_quadrature():
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import forward_project as fp, synthetic
    ct = dx.FanBeamGeometry(64, 10, detector_file=os.path.join(INPUT, 'detector', 'eta_eid_mv.bin'))
    ph = synthetic.make_phantom(16, 1)
    s1 = dx.xRaySpectrum(os.path.join(INPUT, 'spectrum', 'detunedMV_1mGy_float32.bin'), 'mv')
    s2 = synthetic.kramers_spectrum(80)
    E, mu, w = fp.merged_tables(ct, ph, [s1, s2])
    assert mu.shape == (3, E.size) and w.shape == (2, E.size)
    assert np.isclose(w[0].sum(), fp.effective_weights(ct, s1).sum())
    assert np.isclose(w[1].sum(), fp.effective_weights(ct, s2).sum())
    assert np.all(w[0][np.isin(E, s1.E, invert=True)] == 0)


def test_c_abi_exports_every_declared_symbol():
    """The built library exports exactly what include/dexct.h declares (no compute calls here)."""
    from dex_ct_sim_amd import _native
    hdr = open(os.path.join(ROOT, 'include', 'dexct.h')).read()
    declared = sorted(set(re.findall(r'\b(dexct_[a-z_0-9]+)\s*\(', hdr)))
    assert declared == sorted(_native.SYMBOLS)
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.dexct_abi_version.restype = ctypes.c_int
    assert lib.dexct_abi_version() == 6 == _native.ABI_VERSION
    lib.dexct_strerror.restype = ctypes.c_char_p
    assert lib.dexct_strerror(-2) == b'size out of supported range'
    # struct layouts the binding mirrors
    assert ctypes.sizeof(_native.FanGeom) == 72 and _native.PLAN_BYTES == 40 and ctypes.sizeof(_native.GnOptions) == 48


def test_product_does_not_import_oracle():
    """The shipped path never imports, includes, loads or executes anything under oracle/ (comments may
    cite it): scan the package sources for import / include / dlopen-style uses."""
    pkg = os.path.join(ROOT, 'dex-ct-sim_amd')
    py_use = re.compile(r'^\s*(from\s+oracle\b|import\s+oracle\b)|importlib[^\n]*oracle|CDLL\([^\n]*oracle|'
                        r'subprocess[^\n]*oracle', re.M)
    c_use = re.compile(r'#\s*include[^\n]*oracle|dlopen\([^\n]*oracle')
    n = 0
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            txt = None
            if fn.endswith('.py'):
                txt, pat = open(os.path.join(dirpath, fn)).read(), py_use
            elif fn.endswith(('.hip', '.h', 'Makefile')):
                txt, pat = open(os.path.join(dirpath, fn)).read(), c_use
            if txt is not None:
                n += 1
                assert not pat.search(txt), fn
    assert n > 10


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import _native, synthetic
    ct = dx.FanBeamGeometry(32, 8)
    with pytest.raises(_native.DexctError):
        dx.get_sino(ct, synthetic.make_phantom(16, 1), synthetic.kramers_spectrum(80))


def test_c_abi_rejects_bad_arguments_without_touching_the_gpu():
    """Argument validation happens before any HIP call: error codes instead of exceptions/crashes."""
    import ctypes as C
    from dex_ct_sim_amd import _native
    lib = _native.load()
    g = _native.FanGeom(10, 16, 1, 0, 8, 8, 1, 0, 0.1, 0.1, 0.1, 60.0, 100.0)
    one = C.c_void_p(8)      # non-null dummy pointer; never dereferenced on these paths
    EINVAL, ERANGE = -1, -2
    assert lib.dexct_fan_plan(None, one, one, 0, 10, one, None) == EINVAL
    assert lib.dexct_fan_plan(C.byref(g), one, one, 5, 5, one, None) == EINVAL          # empty view range
    assert lib.dexct_fan_plan(C.byref(g), one, one, 0, 11, one, None) == EINVAL         # beyond n_views
    big = _native.FanGeom(10, 16, 1, 0, 9000, 8, 1, 0, 0.1, 0.1, 0.1, 60.0, 100.0)
    assert lib.dexct_fan_plan(C.byref(big), one, one, 0, 10, one, None) == ERANGE       # fixed-point range
    assert lib.dexct_volume_layouts(one, 8, 8, 1, None, None, None) == EINVAL           # nothing to write
    args = [C.byref(g), one, 0, 10, one, one, None]
    assert lib.dexct_siddon_project(*args, 0, 10, 2, one, one, one, None, 0, 0, None, None, None, None) == EINVAL   # no materials
    assert lib.dexct_siddon_project(*args, 257, 10, 2, one, one, one, None, 0, 0, None, None, None, None) == ERANGE  # > DEXCT_MAX_MATERIALS (every uint8 id)
    assert lib.dexct_volume_ids(one, 0, one, None) == EINVAL and lib.dexct_volume_remap(one, 16, None, None) == EINVAL
    assert lib.dexct_siddon_project(*args, 3, 10, 5, one, one, one, None, 0, 0, None, None, None, None) == ERANGE   # > DEXCT_MAX_SPECTRA
    assert lib.dexct_siddon_project(*args, 3, 10, 2, one, one, one, None, 3, 0, None, None, None, None) == EINVAL   # kernel 3 needs vol_zf
    assert lib.dexct_siddon_project(*args, 3, 10, 2, one, one, one, None, 1, 7, None, None, None, None) == EINVAL   # layout
    assert lib.dexct_siddon_project(*args, 3, 10, 2, one, one, one, None, 1, 0, one, None, None, None) == EINVAL   # weights2 without variance
    lo = _native.log_out(8, [1.0])                  # a log sinogram together with a variance output: the log of the noisy
    assert lib.dexct_siddon_project(*args, 3, 10, 2, one, one, one, None, 1, 0, one, one, lo, None) == EINVAL   # counts comes from dexct_sino_log
    assert lib.dexct_sino_log(one, None, 2, 16, one, None) == EINVAL and lib.dexct_sino_log(one, (C.c_float * 5)(), 5, 16, one, None) == ERANGE
    assert C.sizeof(_native.LogOut) == 24
    assert lib.dexct_add_noise(one, one, 2, 4, 1, 8, 2, 0, 1, None) == EINVAL
    assert lib.dexct_gn_decompose(one, one, 1, 0, one, one, 10, 1, 1, 5, 0, 0, None, 0.0, one, None, one, None) == EINVAL    # no pixels
    assert lib.dexct_gn_decompose(one, one, 1, 4, one, one, 10, 1, 1, 5, 2, 0, None, 0.0, one, None, one, None) == EINVAL    # precision
    assert lib.dexct_gn_decompose(one, one, 1, 4, one, one, 10, 8, 1, 5, 1, 0, None, 0.0, one, None, one, None) == EINVAL    # mixed + per-bin
    al = C.c_void_p(16)      # out_a must be 16-byte aligned (a pixel's pair is one 16-byte store)
    assert lib.dexct_gn_decompose(one, one, 1, 4, one, one, 5000, 1, 1, 5, 0, 0, None, 0.0, al, None, one, None) == ERANGE   # energies
    assert lib.dexct_gn_decompose(one, one, 1, 4, one, one, 10, 1, 1, 5, 0, 0, None, 0.0, one, None, one, None) == EINVAL    # out_a misaligned
    # dexct_gn_options: the result order needs n_pix = k * rows * channels; kernel ids 0..2
    assert lib.dexct_gn_decompose(one, one, 1, 12, one, one, 10, 1, 1, 5, 0, 0, None, 0.0, al, _native.gn_options(0.0, 5, 2), one, None) == EINVAL
    assert lib.dexct_gn_decompose(one, one, 1, 12, one, one, 10, 1, 1, 5, 0, 0, None, 0.0, al, _native.gn_options(0.0, 3, 0), one, None) == EINVAL
    assert lib.dexct_gn_decompose(one, one, 1, 12, one, one, 10, 1, 1, 5, 0, 0, None, 0.0, al, _native.gn_options(0.0, 0, 0, 3), one, None) == EINVAL
    assert lib.dexct_gn_workspace_bytes(140, 1) > 140 * 14 * 12 and lib.dexct_gn_workspace_bytes(0, 1) == 0
    # the guarded cone-beam layout (include/dexct.h): (nx ny + 1) columns of ((nz + 15) & ~15) + 32 bytes
    for nx, ny, nz in ((3, 5, 1), (8, 8, 16), (20, 17, 300), (512, 512, 512)):
        assert lib.dexct_cone_layout_bytes(nx, ny, nz) == (nx * ny + 1) * (((nz + 15) & ~15) + 32)
    assert lib.dexct_cone_layout_bytes(0, 5, 5) == 0
    assert lib.dexct_cone_layout(one, 8, 8, 0, one, None) == EINVAL
    assert lib.dexct_transpose_batched(one, one, 1, 4, 4, 3, None) == EINVAL            # element size
    assert lib.dexct_transpose_log(one, one, one, None, 2, 4, 8, 8, None) == EINVAL     # a log sinogram without air values
    assert lib.dexct_transpose_log(one, one, None, None, 5, 4, 8, 8, None) == ERANGE    # > DEXCT_MAX_SPECTRA
    assert lib.dexct_transpose_log(one, None, None, None, 1, 4, 8, 8, None) == EINVAL
    assert lib.dexct_host_pin(None, 4096, 0) == EINVAL and lib.dexct_host_unpin(None, 0) == EINVAL
    assert lib.dexct_download(None, one, 16, None) == EINVAL
    assert lib.dexct_fbp_filter(one, one, one, 1, 1, 0.01, one, None) == EINVAL
    assert lib.dexct_fbp_backproject(one, one, 10, 16, 1, 60.0, 0.0, 0.1, 32, 20.0, one, None) == EINVAL
    assert lib.dexct_label_moments(None, None, None, 16, 1, one, None) == EINVAL
    assert lib.dexct_sino_allgather(one, one, 16, None, None) == EINVAL                    # no communicator
    assert lib.dexct_label_moments(one, None, None, 16, 65, one, None) == ERANGE          # more than 64 labels
    assert lib.dexct_vmi(one, one, 16, 0.2, 0.3, 0.0, 1, one, None) == EINVAL              # HU without water
    for code, text in ((0, b'ok'), (-1, b'invalid argument'), (-3, b'HIP runtime error (see dexct_last_hip_error)'),
                       (-4, b'RCCL library not found, or the collective failed (see dexct_last_hip_error)')):
        assert lib.dexct_strerror(code) == text


def test_xcom_directory_tables_override_the_surrogate(tmp_path, monkeypatch):
    """Real NIST tables can be supplied as <Symbol>.txt files (E [keV], mu/rho [cm^2/g]); mixtures are then
    weight-fraction sums of log-log interpolated element tables."""
    import importlib
    from dex_ct_sim_amd import xcompy
    E = np.array([10.0, 100.0, 1000.0])
    (tmp_path / 'H.txt').write_text('\n'.join(f'{e} {m}' for e, m in zip(E, [0.4, 0.3, 0.1])))
    (tmp_path / 'O.txt').write_text('\n'.join(f'{e} {m}' for e, m in zip(E, [6.0, 0.15, 0.07])))
    monkeypatch.setenv('DEXCT_XCOM_DIR', str(tmp_path))
    saved = dict(xcompy._user_tables)
    try:
        xcompy._user_tables.clear()
        got = xcompy.mixatten('H(11.2)O(88.8)', np.array([100.0]))[0]
        assert np.isclose(got, 0.112 * 0.3 + 0.888 * 0.15)
        mid = xcompy.mixatten('O(100)', np.array([np.sqrt(10.0 * 100.0)]))[0]
        assert np.isclose(mid, np.sqrt(6.0 * 0.15))                     # log-log interpolation
    finally:
        xcompy._user_tables.clear()
        xcompy._user_tables.update(saved)


def test_analysis_helpers_without_gpu():
    """crop_img / get_xcat_mask / measure_roi(give_roi) / output-tree readers of plots.py (plots.py:146-231)."""
    import tempfile, os
    from dex_ct_sim_amd import plots
    M = np.arange(100, dtype=np.float32).reshape(10, 10)
    assert plots.crop_img(M, 4).shape == (4, 4) and plots.crop_img(M, 4)[0, 0] == M[3, 3]
    assert plots.get_xcat_mask(np.array([[-1000.0, -899.0], [0.0, -900.0]])).tolist() == [[False, True], [True, False]]
    assert plots.measure_roi(M, [2, 3, 2, 2], give_roi=True).tolist() == [32.0, 33.0, 42.0, 43.0]
    with tempfile.TemporaryDirectory() as d:
        sub = os.path.join(d, 'mvkv_p', '80kV_1000uGy')
        os.makedirs(sub)
        M.tofile(os.path.join(sub, 'recon_HU_float32.bin'))
        assert np.array_equal(plots.get_img_ct('p', '80kV', 1.0, N_matrix=10, out_dir=d), M)
        sub = os.path.join(d, 'mvkv_p', 'matdecomp_140kV_80kV_5000uGy_5000uGy')
        os.makedirs(sub)
        M.tofile(os.path.join(sub, 'mat1_recon_float32.bin'))
        (2 * M).tofile(os.path.join(sub, 'mat2_recon_float32.bin'))
        a, b = plots.get_img_basismats('p', '140kV', '80kV', 5, 5, crop=4, N_matrix=10, out_dir=d)
        assert a.shape == (4, 4) and np.array_equal(b, 2 * a)


def test_register_table_replaces_cached_mixtures():
    """Advisor finding of round 3: registering a table again under an EXISTING name (element or whole formula) must
    change what mixatten returns - the mixture cache is emptied by register_table."""
    from dex_ct_sim_amd import xcompy
    E = np.array([10.0, 100.0, 1000.0])
    q = np.array([50.0, 200.0])
    saved = dict(xcompy._user_tables)
    try:
        xcompy.register_table('Zz(100)', E, [1.0, 0.5, 0.1])
        a = xcompy.mixatten('Zz(100)', q)
        assert np.array_equal(a, xcompy.mixatten('Zz(100)', q))
        xcompy.register_table('Zz(100)', E, [2.0, 1.0, 0.2])                  # same name, same table count
        b = xcompy.mixatten('Zz(100)', q)
        assert np.allclose(b, 2.0 * a)
        xcompy.register_table('H', E, [0.4, 0.3, 0.1])
        w1 = xcompy.mixatten('H(11.2)O(88.8)', q)
        xcompy.register_table('H', E, [0.8, 0.6, 0.2])                        # an element under a mixture
        w2 = xcompy.mixatten('H(11.2)O(88.8)', q)
        assert np.all(w2 > w1)
    finally:
        xcompy._user_tables.clear()
        xcompy._user_tables.update(saved)
        xcompy._mix_cache.clear()


def test_volume_sample_stride_is_coprime_to_the_dimensions():
    """The O(1) cache key samples ~4096 voxels; for power-of-two volumes a stride of size / 4096 is a multiple of Nx and
    only ever sees the x = 0 face (advisor finding of round 3)."""
    from math import gcd
    from dex_ct_sim_amd import forward_project as fp
    for shape in ((256, 256, 256), (512, 512, 512), (1024, 1024, 1024), (1, 512, 512), (48, 48, 1), (300, 200, 100), (3, 5, 7)):
        step = fp._sample_step(shape)
        size = int(np.prod(shape))
        assert all(d == 1 or gcd(step, d) == 1 for d in shape)
        assert size / step <= 4200 and (size < 4096 or size / step >= 2000)
        idx = np.arange(0, size, step)
        z, y, x = np.unravel_index(idx, shape)
        for coord, d in ((x, shape[2]), (y, shape[1]), (z, shape[0])):
            assert len(np.unique(coord)) >= min(d, 16)


def test_compact_ids_merges_equal_compositions_and_drops_absent_ids():
    """The renumbering the projector applies to a label map (round 4: all 256 ids): equal (density, composition) share a
    row, absent ids get none, id 0 stays row 0, at least two rows."""
    from dex_ct_sim_amd import forward_project as fp
    from dex_ct_sim_amd.system import AIR, BONE, WATER, Material
    mats = [AIR, WATER, BONE, Material('w2', 1.0, WATER.matcomp), Material('dense water', 1.1, WATER.matcomp),
            Material('air again', AIR.density, AIR.matcomp), Material('unused', 3.0, 'H(100)')]
    present = np.zeros(256, dtype=bool)
    present[[0, 1, 3, 4, 5]] = True                       # bone (2) and id 6 do not occur
    rows, lut = fp.compact_ids(present, mats)
    assert rows == [0, 1, 4]                               # air, water (ids 1 and 3), dense water
    assert list(lut[:7]) == [0, 1, 0, 1, 2, 0, 0]         # id 5 = air's composition -> row 0; absent ids -> 0 (never read)
    present[:] = False
    present[0] = True
    rows, lut = fp.compact_ids(present, mats)
    assert rows == [0, 1] and not lut.any()                # an empty volume still gets the kernels' smallest table
    full = [AIR] + [Material(f'm{i}', 0.5 + 0.01 * i, WATER.matcomp) for i in range(1, 256)]
    rows, lut = fp.compact_ids(np.ones(256, dtype=bool), full)
    assert rows == list(range(256)) and list(lut) == list(range(256))


def test_default_tolerance_is_resolved_once_by_one_parser(monkeypatch):
    """matdecomp._default_stop_tol: DEXCT_GN_EXACT=1 -> 0, DEXCT_GN_STOP_TOL=<t> (a number >= 0 or an error: nothing is
    guessed from a malformed value - the library is always handed an explicit tolerance), else 1e-12."""
    from dex_ct_sim_amd import matdecomp as md
    for k in ('DEXCT_GN_EXACT', 'DEXCT_GN_STOP_TOL'):
        monkeypatch.delenv(k, raising=False)
    assert md._default_stop_tol() == 1e-12
    monkeypatch.setenv('DEXCT_GN_STOP_TOL', '1e-10')
    assert md._default_stop_tol() == 1e-10
    monkeypatch.setenv('DEXCT_GN_STOP_TOL', '0')
    assert md._default_stop_tol() == 0.0
    for bad in ('1e-12x', '-1', 'nan'):
        monkeypatch.setenv('DEXCT_GN_STOP_TOL', bad)
        with pytest.raises(ValueError):
            md._default_stop_tol()
    monkeypatch.setenv('DEXCT_GN_EXACT', '1')                     # wins over the tolerance variable
    assert md._default_stop_tol() == 0.0


def test_gate_table_on_disk_is_validated_before_use(tmp_path, monkeypatch):
    """The gate of the Newton short cut is kept under DEXCT_CACHE_DIR between processes (matdecomp._gate_to_disk /
    _gate_from_disk): a complete file for the same tables, tolerance and library is taken; a truncated, edited or foreign one
    (other tables -> another grid header) is ignored; DEXCT_CACHE_DIR=off keeps nothing."""
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import matdecomp as md, quadrature as q, synthetic
    ct = dx.FanBeamGeometry(N_channels=8, N_proj=8, eid=True, detector_file=os.path.join(INPUT, 'detector', 'eta_eid_mv.bin'))
    _, i0, mus = md.decomposition_tables(ct, synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80))
    i0, mus = np.ascontiguousarray(i0), np.ascontiguousarray(mus)
    monkeypatch.setenv('DEXCT_CACHE_DIR', str(tmp_path))
    path = md._gate_cache_path(i0, mus, 1e-12)
    assert path.startswith(str(tmp_path)) and md._gate_cache_path(i0, mus, 1e-14) != path and md._gate_cache_path(2 * i0, mus, 1e-12) != path
    p = q.newton_start_grid(i0, mus)
    n = q.GATE_CELLS
    rng = np.random.default_rng(0)
    head = p['head'].copy()
    head[10] = 2.0                                                        # (with the one-step pairs, as the calibration leaves it)
    start = np.concatenate([head, rng.random(q.start_layout(n)[-1] - q.START_HEADER)])
    stats = {'grid': True, 'open_share': 0.9, 'walk_nonfinite_share': 0.0, 'walk_not_by_rule_share': 0.0}
    assert md._gate_from_disk(path, i0, mus) is None                      # nothing there yet
    md._gate_to_disk(path, start, stats)
    got = md._gate_from_disk(path, i0, mus)
    assert got is not None and np.array_equal(got[0], start) and got[1] == stats
    assert md._gate_from_disk(path, 1.5 * i0, mus) is None                # the file of other tables: its grid header does not match
    raw = open(path, 'rb').read()
    open(path, 'wb').write(raw[: len(raw) // 2])                          # truncated
    assert md._gate_from_disk(path, i0, mus) is None
    edited = start.copy()
    edited[100] += 1e-9
    md._gate_to_disk(path, edited, stats)
    z = dict(np.load(path))
    z['start'] = start                                                    # contents no longer match the digest
    np.savez(open(path, 'wb'), **z)
    assert md._gate_from_disk(path, i0, mus) is None
    md._gate_to_disk(path, None, {'grid': False})                         # "no short cut for these tables" is remembered too
    assert md._gate_from_disk(path, i0, mus) == (None, {'grid': False})
    monkeypatch.setenv('DEXCT_CACHE_DIR', 'off')
    assert md._gate_cache_path(i0, mus, 1e-12) is None
    md._gate_to_disk(None, start, stats)                                  # no-ops
    assert md._gate_from_disk(None, i0, mus) is None


def test_ill_posed_pairs_are_told_from_the_calibration():
    """quadrature.pair_is_ill_posed on the statistics the GPU calibration recorded for the bundled pairs
    (profiles/r06_pair_classes.log): kV / kV pairs keep the short cut, every pair with an MV spectrum runs the fixed count."""
    from dex_ct_sim_amd import quadrature as q
    kv = {'grid': True, 'walk_nonfinite_share': 0.082, 'walk_not_by_rule_share': 0.0019, 'not_a_root_share': 0.0019, 'cond_median': 21.7,
          'open_share': 0.845}                                                      # 140 kV / 80 kV
    mv = dict(kv, walk_nonfinite_share=0.042, not_a_root_share=0.244, cond_median=30.9, open_share=0.405)     # 140 kV / detunedMV
    mvmv = dict(kv, not_a_root_share=0.011, cond_median=3674.0, open_share=0.42)                              # 6MV / detunedMV
    assert not q.pair_is_ill_posed(kv) and not q.pair_is_ill_posed({'grid': False}) and not q.pair_is_ill_posed(None)
    assert q.pair_is_ill_posed(mv) and q.pair_is_ill_posed(mvmv)
    assert q.pair_is_ill_posed(dict(kv, open_share=0.70)) and q.pair_is_ill_posed(dict(kv, cond_median=float('inf')))


def test_the_committed_pair_class_log_agrees_with_the_rule_line_by_line():
    """profiles/r06_pair_classes.log (tools/probes/gn_pair_classes.py on an MI355X, every pair of the bundled spectra + the
    benchmark's Kramers pair + the three goldens): the class printed on each line is what quadrature.pair_is_ill_posed returns
    for the statistics printed on that line, under the thresholds committed NOW - evidence and rule cannot drift apart again
    (round 5's log said the opposite of the final rule for two pairs).  The GPU test test_every_bundled_pair_has_its_class
    re-measures the statistics; this one keeps the documentation honest on CPU."""
    from dex_ct_sim_amd import quadrature as q
    lines = [ln for ln in open(os.path.join(ROOT, 'profiles', 'r06_pair_classes.log')) if 'ill_posed=' in ln]
    assert len(lines) == 14
    classes = {}
    for ln in lines:
        name = ln.split('energies')[0].rsplit(None, 1)[0].strip()
        logged = 'ill_posed=True' in ln
        stats = {}
        for k, v in re.findall(r'(\w+)=([-+.\w]+)', ln.split('calibration')[1]):
            stats[k] = {'True': True, 'False': False}.get(v, None)
            if stats[k] is None:
                stats[k] = float(v)
        assert q.pair_is_ill_posed(stats) == logged, (name, stats)
        classes[name] = logged
    assert [k for k, v in classes.items() if not v] == ['140kV / 120kV', '140kV / 80kV', '120kV / 80kV', 'kramers140 / kramers80',
                                                        'golden0 / (reference)', 'golden2 / (reference)']


def test_host_pages_are_made_resident_without_a_gpu():
    """dexct_host_touch (include/dexct.h): the pages of a host range are made resident by 1 .. 64 threads and the bytes stay as
    they are - pure host code (what a large result block gets before it is page-locked, _device.LazyPinnedResult); bad
    arguments are refused."""
    import ctypes as C
    from dex_ct_sim_amd import _native
    lib = _native.load()
    n = 24 << 20
    for threads in (1, 2, 7):
        a = np.empty(n + 8192, dtype=np.uint8)
        a[::4096] = np.arange(a[::4096].size, dtype=np.uint8)             # something to keep in some of the pages
        a[n // 2 + 5] = 77
        base = a.ctypes.data + 3                                          # any address, any length
        assert lib.dexct_host_touch(base, n, threads) == 0
        assert a[n // 2 + 5] == 77 and np.array_equal(a[::4096], np.arange(a[::4096].size, dtype=np.uint8))
    assert lib.dexct_host_touch(None, n, 1) == -1 and lib.dexct_host_touch(a.ctypes.data, 0, 1) == -1
    assert lib.dexct_host_touch(a.ctypes.data, n, 0) == -1 and lib.dexct_host_touch(a.ctypes.data, n, 65) == -1


def test_page_spans_of_arrays_that_share_pages():
    """_device._page_spans: the page-aligned spans that cover byte ranges, merged where two ranges share or touch a page (two
    halves of one array - sino[0], sino[1] - must be locked as ONE region: a page cannot be registered twice)."""
    from dex_ct_sim_amd import _device
    P = 4096
    assert _device._page_spans([(10 * P + 5, 100)]) == [(10 * P, P)]
    assert _device._page_spans([(10 * P + 5, P), (11 * P + 5, 8)]) == [(10 * P, 2 * P)]                # overlap in page 11
    assert _device._page_spans([(10 * P, P), (11 * P, P)]) == [(10 * P, 2 * P)]                        # touching
    assert _device._page_spans([(20 * P, P), (10 * P, 3)]) == [(10 * P, P), (20 * P, P)]               # apart, sorted
    assert _device._page_spans([(10 * P, 0)]) == []


def test_host_pool_lets_the_oldest_blocks_go(monkeypatch):
    """_device's pool of resident result blocks (no GPU needed for its bookkeeping): blocks return when their arrays are garbage,
    the pool stays within DEXCT_HOST_POOL_GB by dropping the blocks that have waited longest - sizes nobody asks for any more
    cannot hog it - and a block larger than the limit is not kept at all."""
    from dex_ct_sim_amd import _device
    _device.empty_pool()
    monkeypatch.setattr(_device, 'POOL_MAX_BYTES', 10 << 20)
    blk = lambda mb: np.empty(mb << 20, dtype=np.uint8)
    for mb in (4, 4, 3):                       # 11 MB offered: the oldest 4 MB block leaves
        _device._give_back(blk(mb))
    # (a returned block waits in a lock-free deque - the finalizer may run inside a locked region of the same thread - and joins
    # the pool the next time somebody holds the lock: pooled() does)
    assert not _device._pool and len(_device._returned) == 3
    assert {k >> 20: n for k, n in _device.pooled().items()} == {4: 1, 3: 1} and _device._pool_age == [4 << 20, 3 << 20]
    _device._give_back(blk(6))                 # 13 MB: the other 4 MB block leaves, 3 + 6 stay
    assert {k >> 20: n for k, n in _device.pooled().items()} == {3: 1, 6: 1}
    _device._give_back(blk(11))                # larger than the pool: not kept, nothing else disturbed
    assert {k >> 20: n for k, n in _device.pooled().items()} == {3: 1, 6: 1}
    with _device._pool_lock:                   # the finalizer inside a locked region of the same thread: no deadlock
        _device._give_back(blk(1))
    assert _device.empty_pool() == 10 << 20 and not _device._pool and not _device._pool_age and not _device._returned


def test_the_timed_region_of_the_bench_imports_no_checker():
    """bench/step.py is the timed region and nothing else: it imports the product and torch - neither the oracle (the checker)
    nor NumPy-side solvers; bench/cpu.py is the only part of the bench that touches oracle/ (after the timed loop)."""
    import ast
    bdir = os.path.join(ROOT, 'bench')
    mods = {}
    for fn in sorted(os.listdir(bdir)):
        if fn.endswith('.py'):
            tree = ast.parse(open(os.path.join(bdir, fn)).read())
            names = set()
            for node in ast.walk(tree):
                if isinstance(node, ast.Import):
                    names.update(a.name.split('.')[0] for a in node.names)
                elif isinstance(node, ast.ImportFrom) and node.level == 0 and node.module:
                    names.add(node.module.split('.')[0])
            mods[fn] = names
    assert 'oracle' not in mods['step.py'] and 'numpy' not in mods['step.py'] and 'scipy' not in mods['step.py']
    assert {fn for fn, names in mods.items() if 'oracle' in names} == {'cpu.py'}
    src = open(os.path.join(bdir, 'step.py')).read()
    assert 'gn_oracle' not in src and 'c_oracle' not in src
    main_src = open(os.path.join(ROOT, 'bench.py')).read()
    assert main_src.index('wl.timed_steps(args.steps, args.warmup)') < main_src.index('from bench import cpu')


def test_gather_auto_picks_the_fastest_step_and_prefers_the_north_stars_gather_on_a_tie():
    """bench/multi.py pick_mode: the mode of the timed loop from measured warm-up steps (the gloo worlds of
    tests/test_gpu_bench.py exercise the whole selection; this is its rule)."""
    import sys
    sys.path.insert(0, ROOT)
    from bench.multi import pick_mode
    assert pick_mode({'root': 10.0, 'direct': 12.0, 'all': 30.0}) == 'root'
    assert pick_mode({'root': 14.0, 'direct': 12.0, 'all': 30.0}) == 'direct'
    assert pick_mode({'root': 12.2, 'direct': 12.0, 'all': 11.99}) == 'root'           # within 2 %: root, direct, all in that order
    assert pick_mode({'root': 13.0, 'direct': 12.1, 'all': 12.0}) == 'direct'
    assert pick_mode({'root': 13.0, 'direct': 12.5, 'all': 12.0}) == 'all'


def test_the_committed_bench_line_keeps_the_contract():
    """profiles/r06_final_bench.json (the plain `python bench.py` line of the final tree, cited by README.md / DESIGN.md): every key
    of the bench contract, the metric and unit BASELINE.json names, `roofline` (issued flops / peak, with the survey-unit figure and
    the instruction-issue view beside it) and `cpu_baseline` (a bounded sample on the host cores), and the numbers the documents
    quote from it."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    line = json.load(open(os.path.join(root, 'profiles', 'r06_final_bench.json')))
    base = json.load(open(os.path.join(root, 'BASELINE.json')))
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in line, key
    assert line['n_gpus'] == 1 and line['higher_is_better'] is True and line['vs_baseline'] is None and line['data'] == 'synthetic'
    assert line['unit'] == base.get('unit', line['unit']) or 'integrals' in line['unit']
    assert 'workload' in line['config'] and 'model' not in line['config'] and line['config']['baseline_config'] == 'configs[2]'
    roof, cpu = line['roofline'], line['cpu_baseline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'frac_by_survey_unit', 'issue'):
        assert key in roof, key
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-9 and 0.0 < roof['frac'] < 1.0
    assert 0.5 < roof['issue']['frac'] < 1.0 and roof['issue']['unit'].startswith('G wave-instructions')
    for key in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert key in cpu, key
    assert cpu['kind'] in ('port', 'reference') and cpu['cores'] >= 1 and line['value'] / cpu['value'] > 1e3
    # what README.md and DESIGN.md quote
    assert abs(line['ms_per_step'] - 39.9) < 0.2 and abs(line['value'] / 1e12 - 2.14) < 0.02
    assert abs(line['kernel_ms']['gn_decompose'] - 27.6) < 0.2 and abs(line['noisy_step']['ms_per_step'] - 44.3) < 0.2
    assert line['gn_exact']['default_within_1e-12_of_exact_on_every_pixel'] is True and line['gn_exact']['pixels_compared'] == 409600000
    assert line['noisy_step']['same_sample_as_round5_path_bit_for_bit'] is True
