"""Kernel times on one MI355X for the per-GPU share of every configuration BASELINE.json lists (configs[0..4]); bench.py
itself times configs[2].  Rows = Nz (stacked fan) as in bench.py, and the reference's single-row form next to it; the Newton
decomposition in its four modes (default: the short cut with one step where the table vouches for it; two steps; every pixel
walked from 1e-6 in one launch; the reference's fixed count).  Prints one markdown table row per line."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, matdecomp as md, synthetic

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


CONFIGS = [
    # name, n, total views, channels, GPUs sharing the views, spectra, decomposition
    ('configs[0] input/params.txt geometry: 512^2 slice, 1200 x 800, 140/80 kVp + GN', 512, 1200, 800, 1, [140, 80], True),
    ('configs[1] 256^3, 360 x 512, 120 kVp, forward only', 256, 360, 512, 1, [120], False),
    ('configs[2] 512^3, 1000 x 800, 140/80 kVp + GN', 512, 1000, 800, 1, [140, 80], True),
    ('configs[3] 512^3, 2000 x 1024, 140/80 kVp + GN, 1/8 of the views', 512, 2000, 1024, 8, [140, 80], True),
    ('configs[4] 1024^3, 128 bins, 2000 x 1024, forward only, 1/8 of the views', 1024, 2000, 1024, 8, ['grid128'], False),
]
GN_MODES = [('default', {}), ('two steps', dict(two_level='start')), ('single launch', dict(two_level=False)), ('exact', dict(stop_tol=0.0))]
print('| configuration (per-GPU share) | rows | rays | projection ms | WITH quantum noise ms (x noise-free) | rays/s | ray-energy integrals/s | Newton ms, 50 it: '
      + ' / '.join(m for m, _ in GN_MODES) + ' (full-table steps per unmasked pixel) |')
print('|---|---|---|---|---|---|---|---|')
for name, n, views, chans, gpus, kvs, gn in CONFIGS:
    specs = [synthetic.uniform_grid_spectrum(128) if kv == 'grid128' else synthetic.kramers_spectrum(kv) for kv in kvs]
    ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
    for rows in ((1,) if name.startswith('configs[0]') else (n, 1)):
        if rows == 1:
            ph.z_index = n // 2
        ct = dx.FanBeamGeometry(N_channels=chans, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True,
                                detector_file=det, N_rows=rows)
        pj = fp.Projector(ct, ph, view_range=(0, views // gpus))
        _, mu_d, w_d, _ = pj.upload_tables(specs)
        n_e = int((w_d != 0).sum().item())
        out = pj.project_tables(mu_d, w_d, layout=None)
        ms = timed(lambda: pj.project_tables(mu_d, w_d, out=out, layout=None))
        # the same scan WITH quantum noise (round 6: variance and sample inside the default kernels)
        _, _, _, w2 = fp.merged_tables(ct, ph, specs, with_variance=True)
        w2_d = torch.from_numpy(w2).to(device=w_d.device, dtype=torch.float32).contiguous()
        ms_n = timed(lambda: pj.project_tables(mu_d, w_d, out=out, layout=None, w2_d=w2_d, seed=5))
        pj.project_tables(mu_d, w_d, out=out, layout=None)
        n_rays = out[0].numel()
        gn_ms = ''
        if gn:
            _, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
            gmax = out[0].max().double()
            a = torch.empty(out[0].numel() * 2, dtype=torch.float64, device=out.device)
            rc = (rows, chans) if pj.native_layout == 1 else None          # [view][channel][row] in, the reference's order out
            live = float((out[0] < 0.95 * gmax).sum())
            parts = []
            for _, kw in GN_MODES:
                t = timed(lambda: md.gn_device(out[0], out[1], i0, mus, 50, 'f64', out=a, mask_max=gmax, out_rc=rc, **kw), reps=2)
                parts.append('%.2f (%.2f)' % (t, md.last_gn_stats()['pixel_iterations'] / live))
            gn_ms = ' / '.join(parts)
            del a
        print(f'| {name} | {rows} | {n_rays:.3g} | {ms:.2f} | {ms_n:.2f} ({ms_n / ms:.2f}) | {n_rays / ms * 1e3:.3g} | {n_rays * n_e / ms * 1e3:.3g} | {gn_ms} |',
              flush=True)
        del pj, out
        torch.cuda.empty_cache()
