"""Which streams make the pipelined get_basismat_sinos fast?  Ten fresh (compute, upload, download) triples, three calls each; then
the default stream as the compute stream with fresh pairs beside it (what the calls did before _device.side_streams: every other
pair shared a hardware queue with the default stream - 0.157 / 0.180 s alternating, profiles/r05_notes_boundary.md).
gpurun -- python tools/probes/boundary_streams.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import _device, synthetic

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
n = 512
ct = dx.FanBeamGeometry(N_channels=800, N_proj=1000, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=n)
ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
r1, _ = dx.get_sino(ct, ph, specs[0])
r2, _ = dx.get_sino(ct, ph, specs[1])
print('GPU_MAX_HW_QUEUES =', os.environ.get('GPU_MAX_HW_QUEUES'))
dx.get_basismat_sinos(ct, r1, r2, specs[0], specs[1], n_iters=50)
for k in range(10):
    pair = (torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream())
    _device._side['cuda:0'] = pair
    ts = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m1, m2 = dx.get_basismat_sinos(ct, r1, r2, specs[0], specs[1], n_iters=50)
        ts.append(time.perf_counter() - t0)
        del m1, m2
    print(f'triple {k}: stream ids {pair[0].stream_id} {pair[1].stream_id} {pair[2].stream_id}: ' + ' '.join(f'{t:.3f}' for t in ts), flush=True)
for k in range(6):
    pair = (torch.cuda.default_stream(), torch.cuda.Stream(), torch.cuda.Stream())
    _device._side['cuda:0'] = pair
    ts = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m1, m2 = dx.get_basismat_sinos(ct, r1, r2, specs[0], specs[1], n_iters=50)
        ts.append(time.perf_counter() - t0)
        del m1, m2
    print(f'default stream + pair {k}: stream ids {pair[1].stream_id} {pair[2].stream_id}: ' + ' '.join(f'{t:.3f}' for t in ts), flush=True)
