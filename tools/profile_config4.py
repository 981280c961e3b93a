"""BASELINE configs[4] as one rank of eight runs it - 1024^3 phantom, 128 energy bins (one spectrum on linspace(20, 147)), 250 of the
2000 views x 1024 channels x 1024 rows, forward projection only - for the rocprofv3 passes of tools/profile_config4.sh: `reps`
launches of the projection (both outputs of get_sino), then one JSON line with the HIP-event time and the algorithmic bytes
(segments x bytes per stored voxel + outputs).    python tools/profile_config4.py [reps]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, synthetic

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n, views, chans, gpus = 1024, 2000, 1024, 8
det = os.path.join(ROOT, 'dex-ct-sim_amd', 'input', 'detector', 'eta_eid_mv.bin')
ct = dx.FanBeamGeometry(N_channels=chans, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=n)
ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
specs = [synthetic.uniform_grid_spectrum(128)]
pj = fp.Projector(ct, ph, view_range=(0, views // gpus))
_, mu_d, w_d, air = pj.upload_tables(specs)
out, log = pj.project_tables(mu_d, w_d, layout=None, air=air)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    pj.project_tables(mu_d, w_d, out=out, layout=None, air=air, log_out=log)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
from oracle import c_oracle as co          # (the segment count of the shard: the checker's plan, not the product's)
geom = co.make_geom(ct.N_proj, ct.N_channels, n, ph.z_index, n, n, n, ph.dx, ph.dy, ph.dz, ct.SID, ct.SDD)
plan = co.plan(geom, ct.view_cs(), ct.chan_cs(), 0, views // gpus)
seg_vc = int(co.count_segments(geom, plan))
n_rays = out[0].numel()
b_vox = 0.25 if getattr(pj, 'use_packed', False) else 1.0
alg = seg_vc * n * b_vox + 2 * 4 * n_rays                      # sino_raw and sino_log of the one spectrum
print(json.dumps({'workload': 'configs[4] shard: 1024^3, 128 bins, 250 of 2000 views x 1024 channels x 1024 rows, forward only',
                  'kernel': 'rows16_kernel' if b_vox == 0.25 else 'rows4_kernel', 'reps': reps, 'projection_ms': ms, 'rays': n_rays,
                  'weighted_bins': int((w_d != 0).sum().item()), 'segments': seg_vc * n, 'bytes_per_stored_voxel': b_vox,
                  'packed_volume_MiB': n ** 3 * b_vox / 2 ** 20, 'algorithmic_bytes_per_launch': alg,
                  'algorithmic_GBps': alg / (ms * 1e-3) / 1e9, 'rays_per_s': n_rays / (ms * 1e-3),
                  'ray_energy_integrals_per_s': n_rays * int((w_d != 0).sum().item()) / (ms * 1e-3)}))
