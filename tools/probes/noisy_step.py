"""Projection time of the NOISY scan (the reference's live mode: spectra scaled to a dose, main.py:68,101) next to the noise-free
one, on the default kernels: the stacked fan of configs[2] (512^3, 1000 x 800 x 512, 140 / 80 kVp), the configs[4] shard
(1024^3, 250 x 1024 x 1024, 128 bins) and a cone beam (512^3, 100 x 800 x 512).  Each line: projection + what it takes to have
sino_raw and sino_log in the reference's order (dexct_transpose_log), HIP events."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, synthetic

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def run(name, ct, ph, specs, views=None, kernel=0):
    pj = fp.Projector(ct, ph, view_range=views, kernel=kernel)
    _, mu, w, w2 = fp.merged_tables(ct, ph, specs, with_variance=True)
    mu_d, w_d, w2_d = (torch.tensor(x, dtype=torch.float32, device='cuda').contiguous() for x in (pj.compact(mu), w, w2))
    air = w.sum(axis=1)
    out = {}
    for label, kw in (('noise-free', {}), ('noisy', dict(w2_d=w2_d, seed=5))):
        out[label, 'native'] = timed(lambda: pj.project_tables(mu_d, w_d, layout=None, **kw))
        out[label, 'reference order + log'] = timed(lambda: pj.project_tables(mu_d, w_d, layout=0, air=air, **kw))
    poisson = timed(lambda: pj.project(specs, noise='poisson', seed=5, layout=None), reps=2)
    n_rays = pj.n_local_views * ct.N_rows * ct.N_channels
    print(f'{name}: {n_rays:.3e} rays x {w_d.shape[0]} spectra, kernel={kernel}, packed={pj.use_packed}')
    for what in ('native', 'reference order + log'):
        a, b = out['noise-free', what], out['noisy', what]
        print(f'    {what:24s} noise-free {a:8.3f} ms   noisy {b:8.3f} ms   ratio {b / a:5.2f}')
    print(f'    per-bin Poisson (native)  {poisson:8.3f} ms', flush=True)


which = sys.argv[1:] or ['c2', 'c4', 'cone', 'groups']
if 'c2' in which:
    ph = synthetic.make_phantom(512, 512, extent=51.2, seed=1234)
    ct = dx.FanBeamGeometry(N_channels=800, N_proj=1000, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=512)
    specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
    run('configs[2] stacked fan', ct, ph, specs)
    run('configs[2] stacked fan, byte-volume kernel (round 5\'s noisy path)', ct, ph, specs, kernel=3)
if 'c4' in which:
    ph = synthetic.make_phantom(1024, 1024, extent=51.2, seed=1234)
    ct = dx.FanBeamGeometry(N_channels=1024, N_proj=2000, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=1024)
    run('configs[4] shard (1/8 of the views)', ct, ph, [synthetic.uniform_grid_spectrum(128)], views=(0, 250))
    run('configs[4] shard, byte-volume kernel', ct, ph, [synthetic.uniform_grid_spectrum(128)], views=(0, 250), kernel=3)
if 'cone' in which:
    ph = synthetic.make_phantom(512, 512, extent=51.2, seed=1234)
    cone = dx.FanBeamGeometry(800, 100, detector_file=det, N_rows=512, cone=True, h_iso=0.1)
    run('cone beam 100 x 800 x 512', cone, ph, [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)])
if 'groups' in which:
    # an XCAT-like case: 12 table rows on the benchmark's stacked fan (4 group passes of rows16_kernel + the detection pass, which
    # since round 6 sums the variance with the signal and draws the sample itself) and on the cone beam (4 group passes of
    # cone_cols_kernel + the same detection pass)
    import numpy as np
    from dex_ct_sim_amd.system import AIR, BONE, WATER, Material
    ph = synthetic.make_phantom(512, 512, extent=51.2, seed=1234)
    v = ph.volume
    z = np.arange(v.shape[0])[:, None, None]
    ph.volume = np.where(v == 2, 2 + (z // 8) % 10, v).astype(np.uint8)
    ph.materials = [AIR, WATER, BONE] + [Material(f'm{i}', 1.0 + 0.05 * i, 'H(11.2)O(88.8)') for i in range(3, 12)]
    specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
    ct = dx.FanBeamGeometry(N_channels=800, N_proj=250, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=512)
    run('12 table rows, stacked fan 250 x 800 x 512 (material groups)', ct, ph, specs)
    cone = dx.FanBeamGeometry(800, 100, detector_file=det, N_rows=512, cone=True, h_iso=0.1)
    run('12 table rows, cone beam 100 x 800 x 512 (material groups)', cone, ph, specs)
