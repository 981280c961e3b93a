"""Reduced energy quadrature (quadrature.py) against the full grid: kernel time and the largest relative deviation of the
detected counts, at the benchmark's size (1000 x 800 x 512 rows of a 512^3 phantom: rows16_kernel), at configs[1]'s
(2000 x 1024, one row: rays_kernel) and on a cone beam.  gpurun -- python tools/probes/quad_probe.py > gpurun_out/quad_probe.log"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, synthetic

det = os.path.join(ROOT, 'dex-ct-sim_amd', 'input', 'detector', 'eta_eid_mv.bin')
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), float(np.mean(ts))


def case(name, ct, ph, sp):
    pj = fp.Projector(ct, ph)
    _, mu_f, w_f, air = pj.upload_tables(sp, 'full')
    t0 = time.perf_counter()
    _, mu_r, w_r, _ = pj.upload_tables(sp, 'reduced')
    t_lp = time.perf_counter() - t0
    info = pj.quadrature_info
    if info is None:
        print(f'{name}: no reduction found', flush=True)
        return
    full = pj.project_tables(mu_f, w_f, layout=None, air=air)
    red = pj.project_tables(mu_r, w_r, layout=None, air=air)
    dev = ((red[0].double() - full[0].double()).abs() / full[0].double()).max().item()
    dlog = (red[1].double() - full[1].double()).abs().max().item()
    tf = timed(lambda: pj.project_tables(mu_f, w_f, layout=None, air=air))
    tr = timed(lambda: pj.project_tables(mu_r, w_r, layout=None, air=air))
    print(f'{name}: nodes {info["nodes"]} of {info["n_full"]} ({info["nodes_per_spectrum"]}), bound {info["max_rel_err"]:.2e} over '
          f'{info["n_validated"]} points, l_max {np.round(info["l_max"], 1)}, LP {t_lp:.1f} s | full {tf[0]:.3f} ms (mean {tf[1]:.3f}), '
          f'reduced {tr[0]:.3f} ms (mean {tr[1]:.3f}) = {tf[0] / tr[0]:.2f} x | max rel deviation of the counts (two float32 kernels) '
          f'{dev:.2e}, of the log sinogram {dlog:.2e} abs', flush=True)


ph512 = synthetic.make_phantom(512, 512, extent=51.2, seed=1234)
ct = dx.FanBeamGeometry(N_channels=800, N_proj=1000, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=512)
case('bench 1000x800x512, dual', ct, ph512, specs)
case('bench 1000x800x512, single 140', ct, ph512, specs[:1])
ph1 = synthetic.make_phantom(512, 1, extent=51.2, seed=1234)
ct1 = dx.FanBeamGeometry(N_channels=1024, N_proj=2000, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=1)
case('configs[1] 2000x1024x1, single 140', ct1, ph1, specs[:1])
case('configs[1] 2000x1024x1, dual', ct1, ph1, specs)
ph256 = synthetic.make_phantom(256, 256, extent=51.2, seed=1234)
ctc = dx.FanBeamGeometry(N_channels=512, N_proj=360, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=256,
                         h_iso=0.15, cone=True)
case('cone 360x512x256 on 256^3, dual', ctc, ph256, specs)
