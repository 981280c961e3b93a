"""A/B of the traversal kernels on one MI355X: rows4_kernel (kernel 3: one workgroup per (view, channel), every
workgroup at its own pace) against rows4t_kernel (kernel 5: a tile of pairs per workgroup in step), with the tile
shapes given on the command line.  Checks that both produce the same bits, prints one line per variant.

    python tools/bench_tiled.py --n 1024 --views 200 --channels 1024 [--variants 16:4:16 16:4:8 8:4:16 ...]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, synthetic

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=1024)
ap.add_argument('--views', type=int, default=200)
ap.add_argument('--view-begin', type=int, default=0)
ap.add_argument('--total-views', type=int, default=2000)
ap.add_argument('--channels', type=int, default=1024)
ap.add_argument('--variants', nargs='*', default=['16:4:16', '16:4:8', '16:4:32', '16:4:64', '16:8:16', '16:2:16', '16:16:16',
                                                  '16:1:16', '8:4:16', '8:2:16', '8:8:16', '8:4:8'])
ap.add_argument('--spectra', default='grid128')
ap.add_argument('--reps', type=int, default=3)
args = ap.parse_args()

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
n = args.n
ct = dx.FanBeamGeometry(N_channels=args.channels, N_proj=args.total_views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True,
                        detector_file=det, N_rows=n)
ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
specs = [synthetic.uniform_grid_spectrum(128)] if args.spectra == 'grid128' else [synthetic.kramers_spectrum(140),
                                                                                   synthetic.kramers_spectrum(80)]


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


pj3 = fp.Projector(ct, ph, view_range=(args.view_begin, args.view_begin + args.views), kernel=3)
_, mu_d, w_d, _ = pj3.upload_tables(specs)
ref, ref_pl = pj3.project_tables(mu_d, w_d, want_pathlen=True, layout=None)
out = torch.empty_like(ref)
ms3 = timed(lambda: pj3.project_tables(mu_d, w_d, out=out, layout=None), args.reps)
n_rays = ref[0].numel()
print(f'rows4_kernel  (kernel 3)            {ms3:8.2f} ms  {n_rays / ms3 * 1e3:.3g} rays/s  ({n_rays:.3g} rays, n={n})', flush=True)
pj7 = fp.Projector(ct, ph, view_range=(args.view_begin, args.view_begin + args.views), kernel=7)
got, pl = pj7.project_tables(mu_d, w_d, want_pathlen=True, layout=None)
same = bool(torch.equal(pl, ref_pl)) and bool(torch.equal(got, ref))
del got, pl
ms7 = timed(lambda: pj7.project_tables(mu_d, w_d, out=out, layout=None), args.reps)
print(f'rows16_kernel (kernel 7, 2-bit volume) {ms7:8.2f} ms  {n_rays / ms7 * 1e3:.3g} rays/s  bit-identical to kernel 3: {same}', flush=True)
del pj7
pj5 = fp.Projector(ct, ph, view_range=(args.view_begin, args.view_begin + args.views), kernel=5)
for var in args.variants:
    pairs, tv, sub, *fb = var.split(':')
    os.environ['DEXCT_TILE_FB'] = fb[0] if fb else '8'
    os.environ['DEXCT_TILE_PAIRS'], os.environ['DEXCT_TILE_V'], os.environ['DEXCT_TILE_SUB'] = pairs, tv, sub
    got, pl = pj5.project_tables(mu_d, w_d, want_pathlen=True, layout=None)
    same = bool(torch.equal(pl, ref_pl)) and bool(torch.equal(got, ref))
    del got, pl
    ms = timed(lambda: pj5.project_tables(mu_d, w_d, out=out, layout=None), args.reps)
    print(f'rows4t_kernel pairs={pairs:>2} tile_v={tv:>2} sub={sub:>2}  {ms:8.2f} ms  {n_rays / ms * 1e3:.3g} rays/s  '
          f'bit-identical to kernel 3: {same}', flush=True)
