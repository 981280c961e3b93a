"""Projection time vs number of materials (512^3, 1000 x 800 x 512 rays, both spectra): the byte-volume packed-count
kernel in a single pass (3, <= 4 materials) or per material group (4) against the 2-bit-volume kernel, single pass (7,
<= 4 materials) or per material group (8); results compared bit for bit."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, synthetic
from dex_ct_sim_amd.system import AIR, BONE, WATER, Material

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
ct = dx.FanBeamGeometry(800, 1000, detector_file=det, N_rows=512)
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
MS = [int(x) for x in sys.argv[1:]] or [3, 7, 16]
for M in MS:
    ph = synthetic.make_phantom(512, 512, n_spheres=40)
    if M > 3:
        # give the 40 spheres materials 2..M-1 in turn (labels of connected blobs would do the same job)
        rng = np.random.default_rng(1234)
        radii = rng.uniform(0.02, 0.06, 40) * 51.2
        centres = rng.uniform(-0.3, 0.3, (40, 3)) * 51.2
        c = (np.arange(512) + 0.5) * 0.1 - 25.6
        for k, (r, (cx, cy, cz)) in enumerate(zip(radii, centres)):
            ix, iy, iz = (np.nonzero(np.abs(c - q) <= r)[0] for q in (cx, cy, cz))
            sub = ((c[ix][None, None, :] - cx) ** 2 + (c[iy][None, :, None] - cy) ** 2 + (c[iz][:, None, None] - cz) ** 2) <= r * r
            blk = ph.volume[iz[0]:iz[-1] + 1, iy[0]:iy[-1] + 1, ix[0]:ix[-1] + 1]
            blk[sub] = 2 + k % (M - 2)
        ph.materials = [AIR, WATER, BONE] + [Material(f'm{i}', 1.0 + 0.05 * i, 'H(11.2)O(88.8)') for i in range(3, M)]
    ref = None
    for kernel in ((3, 7, 4, 8) if M <= 4 else (4, 8)):
        pj = fp.Projector(ct, ph, kernel=kernel)
        _, mu_d, w_d, _ = pj.upload_tables(specs)
        out = pj.project_tables(mu_d, w_d, layout=None)
        torch.cuda.synchronize()
        same = '' if ref is None else f'   counts bit-identical to kernel 4: {bool(torch.equal(out, ref))}'
        if kernel == 4:
            ref = out.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            pj.project_tables(mu_d, w_d, out=out, layout=None)
        e1.record()
        torch.cuda.synchronize()
        print(f'materials {M:2d} kernel {kernel}: {e0.elapsed_time(e1) / 3:.1f} ms{same}', flush=True)
        del pj, out
        torch.cuda.empty_cache()
