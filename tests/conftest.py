import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the on-disk copy of the Newton short cut's gate tables (matdecomp._gate_cache_path) stays inside the test session
if 'DEXCT_CACHE_DIR' not in os.environ:
    import tempfile
    os.environ['DEXCT_CACHE_DIR'] = tempfile.mkdtemp(prefix='dexct_cache_')

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
INPUT = os.path.join(ROOT, 'dex-ct-sim_amd', 'input')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs an MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build the HIP library (hipcc
    cross-compiles without a GPU) and the oracle once, exactly as __graft_entry__.build() does.  On the GPU
    box the prebuilt libraries travel with the snapshot and nothing is rebuilt."""
    import subprocess
    lib = os.path.join(ROOT, 'dex-ct-sim_amd', 'libdexct_hip.so')
    csrc = os.path.join(ROOT, 'dex-ct-sim_amd', 'csrc')
    newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc) if f.endswith(('.hip', '.h')))
    if not os.path.exists(lib) or os.path.getmtime(lib) < newest:
        if os.path.exists('/opt/rocm/bin/hipcc'):
            subprocess.check_call(['make', '-C', csrc, '-j4', '-s'])
    orc = os.path.join(ROOT, 'oracle', '_build', 'libdexct_oracle.so')
    if not os.path.exists(orc) or os.path.getmtime(orc) < os.path.getmtime(os.path.join(ROOT, 'oracle', 'dexct_oracle.c')):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), '-s'])


@pytest.fixture(scope='session')
def golden():
    return np.load(os.path.join(GOLDEN, 'gn_reference.npz'))


@pytest.fixture(scope='session')
def hip():
    """The HIP library and a device; fails (not skips) if either is missing on a GPU run."""
    import torch
    from dex_ct_sim_amd import _native
    lib = _native.load()
    assert torch.cuda.is_available(), 'gpu-marked test started without a HIP device'
    return lib


def small_scan(n=48, nz=1, n_views=60, n_channels=96, n_rows=1, z_index=0, materials=None, seed=1234):
    """A small fan-beam scan of a synthetic phantom, reference geometry scaled down."""
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import synthetic
    ct = dx.FanBeamGeometry(N_channels=n_channels, N_proj=n_views, gamma_fan=0.8230337, SID=60.0, SDD=100.0,
                            eid=True, detector_file=os.path.join(INPUT, 'detector', 'eta_eid_mv.bin'), N_rows=n_rows)
    ph = synthetic.make_phantom(n, nz, seed=seed, z_index=z_index)
    if materials is not None:
        ph.materials = materials
    return ct, ph


def oracle_geom(ct, ph):
    from oracle import c_oracle as co
    return co.make_geom(ct.N_proj, ct.N_channels, ct.N_rows, ph.z_index, ph.Nx, ph.Ny, ph.Nz, ph.dx, ph.dy, ph.dz,
                        ct.SID, ct.SDD)
