"""Two-level Newton solve (coarse launch on a short quadrature + refining launch on the full tables) against the single
launch, at the benchmark's size: kernel times, executed steps, and the distance of both from the exact mode on every pixel.
gpurun -- python tools/probes/gn_two_level.py [views] > gpurun_out/gn_two_level.log"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, matdecomp as md, synthetic
from dex_ct_sim_amd._device import ptr, stream_ptr

views = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
det = os.path.join(ROOT, 'dex-ct-sim_amd', 'input', 'detector', 'eta_eid_mv.bin')
ct = dx.FanBeamGeometry(N_channels=800, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=n)
ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
pj = fp.Projector(ct, ph)
_, mu_d, w_d, air = pj.upload_tables(specs)
counts = pj.project_tables(mu_d, w_d, layout=None)
_, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
gmax = torch.empty((), dtype=torch.float64, device='cuda')
pj.lib.dexct_reduce_max(ptr(counts[0]), 0, counts[0].numel(), ptr(gmax), stream_ptr())
R, C = n, 800
out = [torch.empty((views, R, C, 2), dtype=torch.float64, device='cuda') for _ in range(3)]


def run(tag, o, **kw):
    for _ in range(2):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        md.gn_device(counts[0], counts[1], i0, mus, 50, 'f64', out=o, out_rc=(R, C), mask_max=gmax, mask_frac=0.95, **kw)
        e1.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    st = md.last_gn_stats()
    print(f'{tag}: {e0.elapsed_time(e1):.1f} ms (host wall of the call {wall * 1e3:.1f} ms)  {st}', flush=True)


run('exact (stop_tol 0)', out[0], stop_tol=0.0, two_level=False)
run('default, one launch', out[1], two_level=False)
t0 = time.perf_counter()
run('default, two-level', out[2], two_level=True)
print(f'   (first two-level call includes the host preparation of the short tables: {time.perf_counter() - t0:.2f} s for both runs)')


def dist(a, b):
    worst, nan_same = 0.0, True
    for v0 in range(0, views, 50):
        x, y = a[v0:v0 + 50], b[v0:v0 + 50]
        nan_same &= bool(torch.equal(torch.isnan(x), torch.isnan(y)))
        worst = max(worst, float(torch.nan_to_num((x - y).abs() / y.abs().clamp(min=1.0), nan=0.0).max()))
    return worst, nan_same


print('one launch vs exact :', dist(out[1], out[0]))
print('two-level vs exact  :', dist(out[2], out[0]))
print('two-level vs one    :', dist(out[2], out[1]))
n_nan = sum(int(torch.isnan(out[0][v0:v0 + 50]).any(-1).sum()) for v0 in range(0, views, 50))
print('pixels NaN in the exact result:', n_nan, 'of', views * R * C)
