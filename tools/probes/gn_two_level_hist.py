"""What the coarse launch leaves behind: histogram of the per-pixel step counts (255 = not ended by the tolerance rule), and
the refining launch timed on it.  gpurun -- python tools/probes/gn_two_level_hist.py [views]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import _native, forward_project as fp, matdecomp as md, synthetic
from dex_ct_sim_amd._device import ptr, stream_ptr

views = int(sys.argv[1]) if len(sys.argv) > 1 else 250
n = 512
det = os.path.join(ROOT, 'dex-ct-sim_amd', 'input', 'detector', 'eta_eid_mv.bin')
ct = dx.FanBeamGeometry(N_channels=800, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=n)
ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
pj = fp.Projector(ct, ph)
_, mu_d, w_d, air = pj.upload_tables(specs)
counts = pj.project_tables(mu_d, w_d, layout=None)
_, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
gmax = torch.empty((), dtype=torch.float64, device='cuda')
lib = pj.lib
lib.dexct_reduce_max(ptr(counts[0]), 0, counts[0].numel(), ptr(gmax), stream_ptr())
R, C = n, 800
npx = counts[0].numel()
i0_d, mus_d, coarse = md._device_tables(i0, mus, counts.device, True)
i0_s, mus_s, start = coarse
use_start = [True]
a = torch.empty((views, R, C, 2), dtype=torch.float64, device='cuda')
iters = torch.empty(npx, dtype=torch.uint8, device='cuda')


def launch(i0t, mut, tol, gn_pass, it_t):
    ne = int(mut.shape[1])
    ws = torch.empty(lib.dexct_gn_workspace_bytes(ne, 1), dtype=torch.uint8, device='cuda')
    ws[72:104].zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _native.check(lib.dexct_gn_decompose(ptr(counts[0]), ptr(counts[1]), 0, npx, ptr(i0t), ptr(mut), ne, 1, 1, 50, 0, 0, ptr(gmax), 0.95,
                                         ptr(a), _native.gn_options(tol, R, C, 1, gn_pass, it_t.data_ptr() if it_t is not None else None,
                                                                    start.data_ptr() if (gn_pass == 1 and use_start[0]) else None),
                                         ptr(ws), stream_ptr()), 'gn')
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), int(ws[72:80].view(torch.int64).item())


for tol1, us in ((1e-7, False), (1e-7, True), (1e-6, True), (1e-5, True), (1e-4, True)):
    use_start[0] = us
    for rep in range(2):
        t1, n1 = launch(i0_s, mus_s, tol1, 1, iters)
        h = torch.bincount(iters.int(), minlength=256).cpu().numpy()
        mask = (counts[0] >= 0.95 * gmax)
        n_live = int((~mask).sum())
        # iters is in result order [v][r][c]; the mask in input order [v][c][r]
        live = (~mask).permute(0, 2, 1).reshape(-1)
        hl = torch.bincount(iters[live].int(), minlength=256).cpu().numpy()
        t2, n2 = launch(i0_d, mus_d, None, 2, iters)
    print(f'coarse tol {tol1:g}, polynomial start {us}: coarse {t1:.1f} ms ({n1 / n_live:.2f} steps per live pixel), refine {t2:.1f} ms ({n2 / n_live:.2f} full steps per live pixel); '
          f'live pixels {n_live}, marked 255: {hl[255]} = {hl[255] / n_live:.4%}; mean k of the rest {np.dot(np.arange(255), hl[:255]) / max(hl[:255].sum(), 1):.2f}; '
          f'k histogram (k: pixels) {dict((int(k), int(hl[k])) for k in np.flatnonzero(hl)[:40])}', flush=True)
use_start[0] = True
# tiles reserved per atomic on the queue head
for tpf in (1, 2, 4, 8, 16):
    os.environ['DEXCT_GN_TILES_PER_FETCH'] = str(tpf)
    best = [1e9, 1e9, 1e9]
    for rep in range(3):
        t1, _ = launch(i0_s, mus_s, 1e-7, 1, iters)
        t2, _ = launch(i0_d, mus_d, None, 2, iters)
        t0, _ = launch(i0_d, mus_d, None, 0, None)
        best = [min(best[0], t1), min(best[1], t2), min(best[2], t0)]
    print(f'tiles per fetch {tpf}: coarse {best[0]:.1f} ms, refine {best[1]:.1f} ms, single launch {best[2]:.1f} ms', flush=True)
del os.environ['DEXCT_GN_TILES_PER_FETCH']
# the refining launch with every pixel marked 255 = a single launch
iters.fill_(255)
t2, n2 = launch(i0_d, mus_d, None, 2, iters)
t0, n0 = launch(i0_d, mus_d, None, 0, None)
print(f'refine with everything marked 255: {t2:.1f} ms ({n2} steps); single launch: {t0:.1f} ms ({n0} steps)')
