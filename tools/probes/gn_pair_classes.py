"""What the gate calibration sees of every pair of bundled spectra (and of the benchmark's Kramers pair): the statistics
quadrature.pair_is_ill_posed decides on.  gpurun -- python tools/probes/gn_pair_classes.py  -> profiles/r06_pair_classes.log
(the classes themselves are pinned by tests/test_gpu_gn.py::test_every_bundled_pair_has_its_class)"""
import itertools
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import INPUT
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import matdecomp as md, quadrature as q, synthetic
from dex_ct_sim_amd._device import to_dev

os.environ['DEXCT_CACHE_DIR'] = 'off'
dev = torch.device('cuda:0')
det = os.path.join(INPUT, 'detector', 'eta_eid_mv.bin')
ct = dx.FanBeamGeometry(N_channels=800, N_proj=1200, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=1)
names = ['140kV', '120kV', '80kV', '6MV', 'detunedMV']
spec = {n: dx.xRaySpectrum(os.path.join(INPUT, 'spectrum', f'{n}_1mGy_float32.bin'), n) for n in names}
pairs = [(a, b, spec[a], spec[b]) for a, b in itertools.combinations(names, 2)]
pairs.append(('kramers140', 'kramers80', synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)))
from conftest import GOLDEN
gold = np.load(os.path.join(GOLDEN, 'gn_reference.npz'))
cases = [(a, b) + tuple(md.decomposition_tables(ct, s1, s2)[1:]) for a, b, s1, s2 in pairs]
cases += [(f'golden{ci}', '(reference)', gold[f'gn{ci}_i0'], gold[f'gn{ci}_mus']) for ci in range(3)]
for a, b, i0, mus in cases:
    i0_d, mus_d = to_dev(i0, torch.float64, dev)[:, None, :].contiguous(), to_dev(mus, torch.float64, dev)
    t0 = time.perf_counter()
    start_h, stats = md.calibrate_gate(np.ascontiguousarray(i0), np.ascontiguousarray(mus), i0_d, mus_d, dev, 1e-12)
    dt = time.perf_counter() - t0
    print(f'{a:>10} / {b:<10} {i0.shape[1]:4d} energies  ill_posed={q.pair_is_ill_posed(stats)!s:5}  calibration {dt:.3f} s  '
          + '  '.join(f'{k}={v:.4g}' if isinstance(v, float) else f'{k}={v}' for k, v in stats.items()), flush=True)
