"""Round 4: the full uint8 id range at benchmark scale (512^3, 250 views x 800 channels x 512 rows, both spectra; and the
single-row scan 1200 x 800).  A label map with N ids of which D have distinct compositions (XCAT style: many organs share a
tissue), ids scattered over the 40 spheres and the body: what the host makes of it (table rows after merging / dropping),
which kernel it picks, and the projection time next to the plain 3-id phantom.
    python tools/bench_ids.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, synthetic
from dex_ct_sim_amd.system import AIR, Material

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
comps = ['H(11.2)O(88.8)', 'H(10.2)C(14.3)N(3.4)O(70.8)Na(0.2)P(0.3)S(0.3)Cl(0.2)K(0.3)',
         'H(3.4)C(15.5)N(4.2)O(43.5)Na(0.1)Mg(0.2)P(10.3)S(0.3)Ca(22.5)', 'H(11.4)C(59.8)N(0.7)O(27.8)Na(0.1)S(0.1)Cl(0.1)']


def label_map(n, nz, n_ids, n_distinct):
    ph = synthetic.make_phantom(n, nz, n_spheres=40)
    if n_ids <= 3:
        return ph
    rng = np.random.default_rng(7)
    distinct = [(round(0.9 + 0.02 * k, 3), comps[k % len(comps)]) for k in range(n_distinct - 1)]
    which = rng.integers(0, len(distinct), n_ids)
    mats = [AIR] + [Material(f'organ{i}', *distinct[which[i]]) for i in range(1, n_ids)]
    # organs = blocks of 16 x 16 x 16 voxels of the body, bone spheres keep an id of their own per block too
    zz, yy, xx = np.meshgrid(np.arange(nz) // 16, np.arange(n) // 16, np.arange(n) // 16, indexing='ij')
    ids = 1 + (zz * 7919 + yy * 104729 + xx * 1299709) % (n_ids - 1)
    ph.volume = np.where(ph.volume > 0, ids, 0).astype(np.uint8)
    ph.materials = mats
    return ph


for rows, views in ((512, 250), (1, 1200)):
    ct = dx.FanBeamGeometry(800, views, detector_file=det, N_rows=rows)
    for n_ids, n_distinct in ((3, 3), (40, 8), (200, 30), (200, 60), (256, 256)):
        ph = label_map(512, rows, n_ids, n_distinct)
        t0 = time.perf_counter()
        pj = fp.Projector(ct, ph)
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t0
        _, mu_d, w_d, _ = pj.upload_tables(specs)
        out = pj.project_tables(mu_d, w_d, layout=None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2):
            pj.project_tables(mu_d, w_d, out=out, layout=None)
        e1.record()
        torch.cuda.synchronize()
        path = ('rows16 (one pass)' if pj.use_packed else 'rows16 per group of 3' if pj.grouped_packed else
                'rows4 per group of 3' if pj.grouped else 'rays_kernel' + (' (LDS columns)' if pj.n_mat > 4 else ''))
        print(f'{views} x 800 x {rows} rows, {n_ids:3d} ids / {n_distinct:3d} compositions -> {pj.n_mat:3d} table rows, {path:24s}: '
              f'{e0.elapsed_time(e1) / 2:8.2f} ms   (device state built in {t_build * 1e3:.0f} ms)', flush=True)
        del pj, out
        torch.cuda.empty_cache()
