"""Round 4: the one-lane-per-pixel kernel (dexct_gn_options.kernel = 1) against the cooperative kernel (2) from the reference's
own size (1200 views x 800 channels, one row = 9.6e5 pixels) up to where the chip is full, default tolerance stop and exact
mode; the two kernels' results compared (another summation order: rounding level, not bits).
    [DEXCT_GN_BLOCKS_PER_CU=n] python tools/probes/gn_small3.py [lib.so]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from dex_ct_sim_amd import _native
if len(sys.argv) > 1:
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, matdecomp as md, synthetic

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
cases = ((60, 256, 256, 1), (100, 512, 256, 1), (200, 512, 256, 1), (360, 512, 256, 1), (1200, 800, 512, 1), (1200, 800, 512, 2), (1200, 800, 512, 4), (100, 800, 512, 64), (1000, 800, 512, 16),
         (1000, 800, 512, 32))
if os.environ.get('SMALL_ONLY'):
    cases = cases[:6]
for views, chans, n, rows in cases:
    ct = dx.FanBeamGeometry(N_channels=chans, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=rows)
    ph = synthetic.make_phantom(n, rows, extent=51.2, seed=1234)
    specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
    pj = fp.Projector(ct, ph)
    _, mu_d, w_d, _ = pj.upload_tables(specs)
    counts = pj.project_tables(mu_d, w_d, layout=None)
    _, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
    gmax = counts[0].max().double()
    masked = float((counts[0] >= 0.95 * gmax).float().mean())
    res = {}
    line = f'{views} x {chans} x {rows} = {counts[0].numel():.3g} pixels ({100 * masked:.0f} % air):'
    for tol, tname in ((None, 'default'), (0.0, 'exact')):
        for kern in (1, 2):
            a = torch.empty(tuple(counts[0].shape) + (2,), dtype=torch.float64, device='cuda')
            kw = dict(out=a, mask_max=gmax, stop_tol=tol, kernel=kern)
            md.gn_device(counts[0], counts[1], i0, mus, 50, 'f64', **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                md.gn_device(counts[0], counts[1], i0, mus, 50, 'f64', **kw)
            e1.record()
            torch.cuda.synchronize()
            res[(tname, kern)] = a
            its = md.last_gn_stats()['pixel_iterations'] / max((1 - masked) * counts[0].numel(), 1)
            line += f'  {tname} k{kern} {e0.elapsed_time(e1) / 5:.3f} ms ({its:.1f} it)'
    d = (res[('exact', 2)] - res[('exact', 1)]).abs() / res[('exact', 1)].abs().clamp(min=1.0)
    line += f'   coop vs lane (exact): {float(torch.nan_to_num(d, nan=0.0).max()):.1e}'
    print(line, flush=True)
