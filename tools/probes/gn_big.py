"""The Newton kernel on configs[3]'s full 2000 x 1024 x 512 sinograms (1.05e9 pixels) ran at 2.67 ns/pixel where every smaller
launch runs at 1.95-2.0 (profiles/r03_shard_of.md).  Pixels per lane of a wave's run (DEXCT_GN_CHUNK) against time."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, matdecomp as md, synthetic

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
n, views, chans = 512, int(os.environ.get('VIEWS', 2000)), 1024
ct = dx.FanBeamGeometry(N_channels=chans, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=n)
ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
pj = fp.Projector(ct, ph)
_, mu_d, w_d, _ = pj.upload_tables(specs)
counts = pj.project_tables(mu_d, w_d, layout=None)
_, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
gmax = counts[0].max().double()
a = torch.empty(tuple(counts[0].shape) + (2,), dtype=torch.float64, device='cuda')
npix = counts[0].numel()
for chunk in os.environ.get('CHUNKS', ',16,32,64,128').split(','):
    os.environ.pop('DEXCT_GN_CHUNK', None)
    if chunk:
        os.environ['DEXCT_GN_CHUNK'] = chunk
    ts = []
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        md.gn_device(counts[0], counts[1], i0, mus, 50, 'f64', out=a, mask_max=gmax)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    st = md.last_gn_stats()
    print(f'chunk {chunk or "default"}: {ts[0]:.0f} {ts[1]:.0f} ms = {min(ts) * 1e6 / npix:.2f} ns/pixel, executed iterations per pixel '
          f'{st["pixel_iterations"] / npix:.2f}', flush=True)
os.environ.pop('DEXCT_GN_CHUNK', None)
# halves: the same pixels in two launches (default run length)
half = views // 2
for sl in (slice(0, half), slice(half, views)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    md.gn_device(counts[0][sl], counts[1][sl], i0, mus, 50, 'f64', out=a[sl], mask_max=gmax)
    e1.record()
    torch.cuda.synchronize()
    print(f'views {sl.start}..{sl.stop}: {e0.elapsed_time(e1):.0f} ms = {e0.elapsed_time(e1) * 1e6 / (npix / 2):.2f} ns/pixel')
