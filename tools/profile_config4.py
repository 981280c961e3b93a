"""BASELINE configs[4] as one rank of eight runs it - 1024^3 phantom, 128 energy bins (one spectrum on linspace(20, 147)), 250 of the
2000 views x 1024 channels x 1024 rows, forward projection only - for the rocprofv3 passes of tools/profile_config4.sh: `reps`
launches of the projection (both outputs of get_sino), then one JSON line with the HIP-event time and the algorithmic bytes
(segments x bytes per stored voxel + outputs).    python tools/profile_config4.py [reps] [key=value ...]
Keys (round 6): noisy=1 - the scan WITH quantum noise (the kernel sums the variance and draws the sample: rows16_kernel<NOISY>);
n= views= chans= gpus= spec=grid128|dual - another stacked fan through the same passes (n=1600 views=1000 chans=800 gpus=1
spec=dual: the beyond-the-Infinity-Cache point of README.md, profiles/r06_n1600.md)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, synthetic

reps = int(sys.argv[1]) if len(sys.argv) > 1 and '=' not in sys.argv[1] else 5
opt = dict(a.split('=', 1) for a in sys.argv[1:] if '=' in a)
n, views, chans, gpus = int(opt.get('n', 1024)), int(opt.get('views', 2000)), int(opt.get('chans', 1024)), int(opt.get('gpus', 8))
noisy = opt.get('noisy', '0') not in ('0', '')
det = os.path.join(ROOT, 'dex-ct-sim_amd', 'input', 'detector', 'eta_eid_mv.bin')
ct = dx.FanBeamGeometry(N_channels=chans, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=n)
ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
specs = ([synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)] if opt.get('spec', 'grid128') == 'dual' else
         [synthetic.uniform_grid_spectrum(128)])
pj = fp.Projector(ct, ph, view_range=(0, views // gpus))
_, mu_d, w_d, air = pj.upload_tables(specs)
kw = {}
if noisy:
    _, _, _, w2 = fp.merged_tables(ct, ph, specs, with_variance=True)
    kw = dict(w2_d=torch.from_numpy(w2).to(device=w_d.device, dtype=torch.float32).contiguous(), seed=5)
out, log = pj.project_tables(mu_d, w_d, layout=None, air=air, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    pj.project_tables(mu_d, w_d, out=out, layout=None, air=air, log_out=log, **kw)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
from oracle import c_oracle as co          # (the segment count of the shard: the checker's plan, not the product's)
geom = co.make_geom(ct.N_proj, ct.N_channels, n, ph.z_index, n, n, n, ph.dx, ph.dy, ph.dz, ct.SID, ct.SDD)
plan = co.plan(geom, ct.view_cs(), ct.chan_cs(), 0, views // gpus)
seg_vc = int(co.count_segments(geom, plan))
n_rays = out[0].numel()
b_vox = 0.25 if getattr(pj, 'use_packed', False) else 1.0
S = len(specs)
alg = seg_vc * n * b_vox + S * 2 * 4 * n_rays                  # sino_raw and sino_log of every spectrum
label = ('configs[4] shard' if (n, views, chans, gpus) == (1024, 2000, 1024, 8) else 'stacked fan') + \
    f': {n}^3, {int((w_d != 0).any(dim=0).sum().item())} bins, {views // gpus} of {views} views x {chans} channels x {n} rows, forward only' + \
    (', WITH quantum noise (variance + sample in the kernel)' if noisy else '')
print(json.dumps({'workload': label, 'noisy': noisy, 'n': n, 'spectra': S,
                  'kernel': 'rows16_kernel' if b_vox == 0.25 else 'rows4_kernel', 'reps': reps, 'projection_ms': ms, 'rays': n_rays,
                  'weighted_bins': int((w_d != 0).sum().item()), 'segments': seg_vc * n, 'bytes_per_stored_voxel': b_vox,
                  'packed_volume_MiB': n ** 3 * b_vox / 2 ** 20, 'algorithmic_bytes_per_launch': alg,
                  'algorithmic_GBps': alg / (ms * 1e-3) / 1e9, 'rays_per_s': n_rays / (ms * 1e-3),
                  'ray_energy_integrals_per_s': n_rays * int((w_d != 0).sum().item()) / (ms * 1e-3)}))
