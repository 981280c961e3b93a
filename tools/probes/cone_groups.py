"""Cone beam with more than 3 table rows (round 6): one cone_cols_kernel pass per group of three materials + one detection pass
(dexct_cone_project_grouped) against the one-thread-per-ray kernel (round 5's only path for > 3 rows), at the benchmark's cone
scan: 512^3, 100 views x 800 channels x 512 rows, 140 / 80 kVp.  Path lengths compared bit for bit.
    python tools/probes/cone_groups.py [n_mat ...]      (default 3 4 6 12)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, synthetic
from dex_ct_sim_amd.system import AIR, BONE, WATER, Material

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
views = 100
only = os.environ.get('CONE_GROUPS_ONLY')          # 'groups': just the group passes (profiling runs)


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


ct = dx.FanBeamGeometry(800, views, detector_file=det, N_rows=512, cone=True, h_iso=0.1)
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
rays = views * 512 * 800
base_ms = None
for n_mat in [int(a) for a in sys.argv[1:]] or [3, 4, 6, 12]:
    ph = synthetic.make_phantom(512, 512)
    if n_mat > 3:
        # the bone spheres become n_mat - 2 materials (ids 2 .. n_mat - 1), by slice band: every group sees boundaries
        v = ph.volume
        z = np.arange(v.shape[0])[:, None, None]
        ph.volume = np.where(v == 2, 2 + (z // 8) % (n_mat - 2), v).astype(np.uint8)
        ph.materials = [AIR, WATER, BONE] + [Material(f'm{i}', 1.0 + 0.05 * i, 'H(11.2)O(88.8)') for i in range(3, n_mat)]
    res = {}
    for kernel in ((2,) if only == 'groups' else (2, 1)):
        pj = fp.Projector(ct, ph, kernel=kernel)
        _, mu_d, w_d, _ = pj.upload_tables(specs)
        out, pl = pj.project_tables(mu_d, w_d, want_pathlen=True)
        ms = timed(lambda: pj.project_tables(mu_d, w_d, out=out))
        res[kernel] = (ms, pl, out.clone(), pj.n_mat, pj.cone_groups)
        del pj
    ms2, pl2, c2, M, grouped = res[2]
    line = f'{n_mat} materials ({M} table rows): row kernels{" in " + str((M + 2) // 3) + " group passes + detection" if grouped else ""} {ms2:.2f} ms = {rays / ms2 * 1e3:.3g} rays/s'
    if 1 in res:
        ms1, pl1, c1, _, _ = res[1]
        line += (f'; one thread per ray {ms1:.2f} ms ({ms1 / ms2:.1f}x); path lengths identical: {bool(torch.equal(pl1, pl2))}; '
                 f'counts max rel diff {float(((c1 - c2).abs() / c1).max()):.1e}')
    if n_mat == 3:
        base_ms = ms2
    elif base_ms is not None:
        line += f'; {(M + 2) // 3} x the 3-material pass = {(M + 2) // 3 * base_ms:.2f} ms -> ratio {ms2 / ((M + 2) // 3 * base_ms):.2f}'
    print(line, flush=True)
