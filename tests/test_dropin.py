"""The reference's own import lines (main.py:17-22, plots.py:16-18) resolve to this engine when
dex-ct-sim_amd/dropin is put on sys.path, and main.py reproduces the reference's
output tree."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import INPUT, ROOT

DROPIN = os.path.join(ROOT, 'dex-ct-sim_amd', 'dropin')

REFERENCE_IMPORTS = '''
import sys
sys.path.insert(0, %r)
from xtomosim.system import read_parameter_file, xRaySpectrum          # main.py:19
from xtomosim.forward_project import get_sino                           # main.py:20
from xtomosim.back_project import get_recon                             # main.py:21
from matdecomp import get_basismat_sinos                                # main.py:22
import xcompy as xc                                                     # plots.py:16
from xtomosim.system import FanBeamGeometry, VoxelPhantom               # plots.py:17
from matdecomp import matcomp1, matcomp2                                # plots.py:18
import numpy as np
from plots import make_vmi, measure_roi, crop_img, get_xcat_mask        # helpers plots.py defines itself (:136-231)
assert xc.mixatten(matcomp1, np.array([60.0])).shape == (1,)
ct = FanBeamGeometry(N_channels=800, N_proj=1200, gamma_fan=0.8230337, SID=60.0, SDD=100.0, h_iso=1.0, eid=True,
                     detector_file=%r)                                  # plots.py:109-111
print(get_sino.__module__, get_recon.__module__, get_basismat_sinos.__module__, ct.A_iso)
'''


def test_reference_import_lines_resolve():
    code = REFERENCE_IMPORTS % (DROPIN, os.path.join(INPUT, 'detector', 'eta_eid_mv.bin'))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300, cwd='/tmp')
    assert out.returncode == 0, out.stderr[-2000:]
    mods = out.stdout.split()
    assert mods[:3] == ['dex_ct_sim_amd.forward_project', 'dex_ct_sim_amd.back_project', 'dex_ct_sim_amd.matdecomp']


@pytest.mark.gpu
def test_main_writes_the_reference_output_tree(tmp_path):
    params = json.load(open(os.path.join(INPUT, 'params.txt')))
    params.update(RUN_ID='tiny', Nx=64, Ny=64, dx=0.8, dy=0.8, dz=0.8, N_channels=96, N_projections=90,
                  N_recon_matrix=64, FOV_recon=50.0,
                  detector_filename=os.path.join(INPUT, 'detector', 'eta_eid_mv.bin'))
    pf = tmp_path / 'params.txt'
    pf.write_text(json.dumps(params))
    out_dir = tmp_path / 'output'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'dex-ct-sim_amd', 'main.py'), '--params', str(pf), '--out',
                        str(out_dir), '--pairs', '140kV:80kV:5:5'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    base = out_dir / 'tiny'
    expect = {'140kV_5000uGy': ['sino_raw', 'sino_log', 'recon_raw', 'recon_HU'],
              '80kV_5000uGy': ['sino_raw', 'sino_log', 'recon_raw', 'recon_HU'],
              'matdecomp_140kV_80kV_5000uGy_5000uGy': ['mat1_sino', 'mat2_sino', 'mat1_recon', 'mat2_recon']}
    assert (base / 'params.txt').exists()                      # main.py:98
    for sub, names in expect.items():
        for nme in names:
            f = base / sub / f'{nme}_float32.bin'
            assert f.exists(), f
            a = np.fromfile(f, dtype=np.float32)
            assert a.size == (64 * 64 if 'recon' in nme else 90 * 96)
            assert np.isfinite(a).all() or 'sino_log' in nme
    m1 = np.fromfile(base / 'matdecomp_140kV_80kV_5000uGy_5000uGy' / 'mat1_sino_float32.bin', dtype=np.float32)
    assert (m1 == 0).any() and (m1 > 1).any()                   # masked air rays and real thicknesses


@pytest.mark.gpu
@pytest.mark.parametrize('gather', [None, 'root'])
def test_main_sharded_over_two_ranks_writes_identical_files(tmp_path, gather):
    """The N > 1 product path end to end (view shards, sinogram all-gather, global max for the air mask,
    sharded decomposition + gather): two ranks - gloo, sharing the one GPU of the test box, collectives staged
    through the host; on a real node the backend is RCCL - write byte-identical files to a single process.
    DEXCT_GATHER=root (a documented value: the default of step loops that assemble on one rank) must not reach the drop-in
    calls, which return the whole array on every rank (advisor finding of round 5: the other ranks got None and crashed)."""
    params = json.load(open(os.path.join(INPUT, 'params.txt')))
    params.update(RUN_ID='tiny', Nx=48, Ny=48, dx=1.0, dy=1.0, dz=1.0, N_channels=64, N_projections=45,   # ragged: 23 + 22
                  N_recon_matrix=32, FOV_recon=50.0, back_project=False,
                  detector_filename=os.path.join(INPUT, 'detector', 'eta_eid_mv.bin'))
    pf = tmp_path / 'params.txt'
    pf.write_text(json.dumps(params))
    main_py = os.path.join(ROOT, 'dex-ct-sim_amd', 'main.py')
    common = ['--params', str(pf), '--pairs', '140kV:80kV:5:5']
    r1 = subprocess.run([sys.executable, main_py, '--out', str(tmp_path / 'one')] + common, capture_output=True,
                        text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    env = dict(os.environ, DEXCT_DIST_BACKEND='gloo', **({'DEXCT_GATHER': gather} if gather else {}))
    port = 29600 + os.getpid() % 300 + (17 if gather else 0)
    r2 = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                         '--master-addr', '127.0.0.1', '--master-port', str(port), main_py, '--out',
                         str(tmp_path / 'two')] + common, capture_output=True, text=True, timeout=900, env=env)
    assert r2.returncode == 0, r2.stderr[-3000:]
    n = 0
    for dirpath, _, files in os.walk(tmp_path / 'one' / 'tiny'):
        for fn in files:
            if fn.endswith('.bin'):
                a = np.fromfile(os.path.join(dirpath, fn), dtype=np.float32)
                b = np.fromfile(os.path.join(dirpath.replace(str(tmp_path / 'one'), str(tmp_path / 'two')), fn),
                                dtype=np.float32)
                assert np.array_equal(a, b, equal_nan=True), fn
                n += 1
    assert n == 6


@pytest.mark.gpu
def test_main_cone_beam_run_with_window(tmp_path):
    """A cone-beam parameter file (extension keys scanner_geometry = cone_beam, N_rows) through main.py: cone
    projection, decomposition, Feldkamp reconstructions with an apodised ramp (--window)."""
    params = json.load(open(os.path.join(INPUT, 'params.txt')))
    params.update(RUN_ID='cone', Nx=48, Ny=48, Nz=16, dx=0.6, dy=0.6, dz=0.6, N_channels=80, N_projections=72,
                  N_recon_matrix=40, FOV_recon=28.0, scanner_geometry='cone_beam', N_rows=12, detector_px_height=0.6,
                  detector_filename=os.path.join(INPUT, 'detector', 'eta_eid_mv.bin'))
    pf = tmp_path / 'params.txt'
    pf.write_text(json.dumps(params))
    out_dir = tmp_path / 'output'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'dex-ct-sim_amd', 'main.py'), '--params', str(pf), '--out',
                        str(out_dir), '--pairs', '140kV:80kV:5:5', '--window', 'hann'], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    base = out_dir / 'cone'
    sino = np.fromfile(base / '140kV_5000uGy' / 'sino_log_float32.bin', dtype=np.float32)
    assert sino.size == 72 * 12 * 80
    rec = np.fromfile(base / '140kV_5000uGy' / 'recon_raw_float32.bin', dtype=np.float32)
    assert rec.size == 12 * 40 * 40 and np.isfinite(rec).all()             # one slice per detector row by default
    vol = rec.reshape(12, 40, 40)
    assert 0.1 < vol[6, 18:22, 18:22].mean() < 0.3                          # water at the centre of the mid-plane [1/cm]
    m1 = np.fromfile(base / 'matdecomp_140kV_80kV_5000uGy_5000uGy' / 'mat1_recon_float32.bin', dtype=np.float32)
    assert m1.size == 12 * 40 * 40 and np.isfinite(m1).all()


@pytest.mark.gpu
def test_main_noise_switch_follows_the_dose(tmp_path):
    """--noise: the dose of a spectrum (mGy, main.py:68) sets the noise of its sinogram - four times the dose, half
    the relative noise; the same seed reproduces the files, the default stays noise-free."""
    params = json.load(open(os.path.join(INPUT, 'params.txt')))
    params.update(RUN_ID='n', Nx=64, Ny=64, dx=0.4, dy=0.4, dz=0.4, N_channels=96, N_projections=60, back_project=False,
                  detector_filename=os.path.join(INPUT, 'detector', 'eta_eid_mv.bin'))
    pf = tmp_path / 'params.txt'
    pf.write_text(json.dumps(params))
    main_py = os.path.join(ROOT, 'dex-ct-sim_amd', 'main.py')

    def run(tag, pairs, *extra):
        out = tmp_path / tag
        r = subprocess.run([sys.executable, main_py, '--params', str(pf), '--out', str(out), '--pairs', pairs, *extra],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return out / 'n'

    clean = run('clean', '140kV:80kV:0.001:0.004')
    a = run('a', '140kV:80kV:0.001:0.004', '--noise', 'gaussian', '--seed', '5')
    b = run('b', '140kV:80kV:0.001:0.004', '--noise', 'gaussian', '--seed', '5')

    def rel_noise(base, sub, ref_base):
        x = np.fromfile(base / sub / 'sino_raw_float32.bin', dtype=np.float32).astype(np.float64)
        m = np.fromfile(ref_base / sub / 'sino_raw_float32.bin', dtype=np.float32).astype(np.float64)
        air = m > 0.9 * m.max()
        return np.std(x[air] / m[air] - 1.0)

    n_lo, n_hi = rel_noise(a, '140kV_0001uGy', clean), rel_noise(a, '80kV_0004uGy', clean)
    assert n_lo > 0 and n_hi > 0
    assert rel_noise(clean, '140kV_0001uGy', clean) == 0.0
    for sub in ('140kV_0001uGy', '80kV_0004uGy'):
        assert (a / sub / 'sino_raw_float32.bin').read_bytes() == (b / sub / 'sino_raw_float32.bin').read_bytes()
    # noise ~ 1 / sqrt(photons): compare each spectrum at two doses
    c = run('c', '140kV:80kV:0.004:0.016', '--noise', 'gaussian', '--seed', '5')
    clean4 = run('clean4', '140kV:80kV:0.004:0.016')
    assert abs(rel_noise(c, '140kV_0004uGy', clean4) / n_lo - 0.5) < 0.1


@pytest.mark.gpu
def test_config1_default_params_full_size_relative_paths(tmp_path):
    """BASELINE configs[0]: the bundled default params.txt at FULL size (512^2 phantom, 1200 views x 800 channels,
    80 / 140 kVp spectra, reconstructions at 512^2) through main.py, started the way the reference is (``cd`` next to
    ``input/`` and relative paths: --input-dir input --params input/params.txt; main.py must not change directory);
    the raw sinograms against the float64 Siddon oracle on a sample of views, the decomposition against the float64
    Newton oracle on the same counts."""
    import shutil
    from oracle import c_oracle as co
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import forward_project as fp, matdecomp as md
    work = tmp_path / 'work'
    shutil.copytree(INPUT, work / 'input')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'dex-ct-sim_amd', 'main.py'), '--input-dir', 'input', '--params',
                        os.path.join('input', 'params.txt'), '--out', 'output', '--pairs', '140kV:80kV:5:5'],
                       capture_output=True, text=True, timeout=900, cwd=str(work))
    assert r.returncode == 0, r.stderr[-2000:]
    assert '1180 / 1200' in r.stdout                                  # the reference's progress lines (matdecomp.py:111-112)
    base = work / 'output' / 'synthetic_512'
    nV, nC = 1200, 800
    raw = {k: np.fromfile(base / f'{k}_5000uGy' / 'sino_raw_float32.bin', dtype=np.float32).reshape(nV, nC)
           for k in ('140kV', '80kV')}
    run = dx.read_parameter_file(str(work / 'input' / 'params.txt'))[0]
    ct, ph = run[3], run[4]
    specs = []
    for sid in ('140kV', '80kV'):
        s = dx.xRaySpectrum(str(work / 'input' / 'spectrum' / f'{sid}_1mGy_float32.bin'), sid)
        s.rescale_counts(ct.A_iso * 5.0 / ct.N_proj)
        specs.append(s)
    _, mu64, w64 = fp.merged_tables(ct, ph, specs)
    g = co.make_geom(ct.N_proj, ct.N_channels, 1, ph.z_index, ph.Nx, ph.Ny, ph.Nz, ph.dx, ph.dy, ph.dz, ct.SID, ct.SDD)
    for v in (0, 150, 300, 601, 1199):
        ref = co.project_classic(g, ct.view_cs(), ct.chan_cs(), v, v + 1, ph.volume, mu64, w64, n_threads=8)[:, 0, 0, :]
        for k, key in enumerate(('140kV', '80kV')):
            assert np.max(np.abs(raw[key][v] - ref[k]) / ref[k]) < 1e-5, (v, key)
    m = [np.fromfile(base / 'matdecomp_140kV_80kV_5000uGy_5000uGy' / f'mat{i}_sino_float32.bin', dtype=np.float32).reshape(nV, nC)
         for i in (1, 2)]
    _, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
    air = raw['140kV'] >= 0.95 * raw['140kV'].max()
    assert air.any() and np.all(m[0][air] == 0) and np.all(m[1][air] == 0)
    views = [0, 300, 601]
    a = co.gn_decompose(raw['140kV'][views].astype(np.float64).ravel(), raw['80kV'][views].astype(np.float64).ravel(), i0, mus,
                        50, n_threads=8).reshape(len(views), nC, 2)
    live = ~air[views] & np.isfinite(a).all(-1)
    for i in (0, 1):
        got, want = m[i][views][live], a[..., i][live]
        assert np.max(np.abs(got - want) / np.maximum(np.abs(want), 1.0)) < 1e-6          # files are float32
    for sub, names in (('140kV_5000uGy', ['recon_raw', 'recon_HU']), ('matdecomp_140kV_80kV_5000uGy_5000uGy', ['mat1_recon'])):
        for nme in names:
            img = np.fromfile(base / sub / f'{nme}_float32.bin', dtype=np.float32)
            assert img.size == 512 * 512 and np.isfinite(img).all()


@pytest.mark.gpu
def test_get_sino_sees_in_place_changes():
    """The reference rebuilds its state on every get_sino call; the cached device state here is keyed on the geometry
    numbers and a checksum of the whole volume (computed beside the GPU work), so an in-place edit of phantom.volume or of a scanner number is seen."""
    import dex_ct_sim_amd as dx
    from conftest import small_scan
    from dex_ct_sim_amd import synthetic
    ct, ph = small_scan(n=48, n_views=30, n_channels=64)
    spec = synthetic.kramers_spectrum(100)
    a, _ = dx.get_sino(ct, ph, spec)
    b, _ = dx.get_sino(ct, ph, spec)
    assert np.array_equal(a, b)
    ph.volume[0, 20:28, 20:28] = 2                                      # in place: a bone block
    c, _ = dx.get_sino(ct, ph, spec)
    assert not np.array_equal(a, c)
    fresh_ct, fresh_ph = small_scan(n=48, n_views=30, n_channels=64)
    fresh_ph.volume[0, 20:28, 20:28] = 2
    d, _ = dx.get_sino(fresh_ct, fresh_ph, spec)
    assert np.array_equal(c, d)
    ct.SID = 55.0                                                       # a scanner number changed in place
    e, _ = dx.get_sino(ct, ph, spec)
    assert not np.array_equal(c, e)
