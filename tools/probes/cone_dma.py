"""(NOTE: the LDS-DMA variants lost and were not committed - profiles/r04_notes_cone.md; with the committed library every
setting of DEXCT_CONE_DMA runs the register-staged kernel.)
cone_cols_kernel at benchmark scale (512^3, 100 views x 800 channels x 512 rows, both spectra): the staging variants -
DEXCT_CONE_DMA = 0 (round 3: global -> VGPR -> ds_write, double buffer), 2 / 3 / 4 (LDS-DMA, ring depth) - timed, and
their per-material path lengths and counts compared bit for bit with the one-thread-per-ray cone_kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, synthetic

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
views = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ph = synthetic.make_phantom(512, 512)
ct = dx.FanBeamGeometry(800, views, detector_file=det, N_rows=512, cone=True, h_iso=ph.dz)
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
pj1 = fp.Projector(ct, ph, kernel=1)
_, mu_d, w_d, air = pj1.upload_tables(specs)
ref_c, ref_p = pj1.project_tables(mu_d, w_d, want_pathlen=True)
del pj1
pj = fp.Projector(ct, ph, kernel=2)
out = torch.empty_like(ref_c)
first = None
for rep in range(2):
    for dma in ('0', '2', '3', '4'):
        os.environ['DEXCT_CONE_DMA'] = dma
        c, p = pj.project_tables(mu_d, w_d, want_pathlen=True)
        same_p = bool(torch.equal(p, ref_p))
        if first is None:
            first = c.clone()
        same_c = bool(torch.equal(c, first))
        del c, p
        ts = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            pj.project_tables(mu_d, w_d, out=out)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f'DEXCT_CONE_DMA={dma}: {min(ts):.3f} ms (median {sorted(ts)[1]:.3f})   path lengths == cone_kernel: {same_p}   '
              f'counts == DMA=0: {same_c}   vs cone_kernel counts: {float(((out - ref_c).abs() / ref_c).max()):.1e}', flush=True)
os.environ.pop('DEXCT_CONE_DMA', None)
