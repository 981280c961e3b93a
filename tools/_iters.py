import sys, os, shutil, numpy as np, torch
sys.path.insert(0, os.getcwd())
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, matdecomp as md, synthetic
det = 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin'
ct = dx.FanBeamGeometry(800, 40, detector_file=det, N_rows=512)
ph = synthetic.make_phantom(512, 512)
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
pj = fp.Projector(ct, ph)
_, mu_d, w_d, _ = pj.upload_tables(specs)
c = pj.project_tables(mu_d, w_d, layout=None)
_, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
a = md.gn_device(c[0], c[1], i0, mus, 50, 'f64')
it = a[..., 1].cpu().numpy().ravel(); per = a[..., 0].cpu().numpy().ravel()
air = (c[0] >= 0.95 * c[0].max()).cpu().numpy().ravel()
it = it[~air]; per = per[~air]
print('period histogram', {int(k): int((per == k).sum()) for k in np.unique(per)})
print('pixels', it.size, 'mean exit it', it.mean(), 'percentiles', np.percentile(it, [10, 50, 90, 99, 99.9]), 'frac ran all 50:', (it >= 50).mean())
w = it[: it.size // 64 * 64].reshape(-1, 64).max(1)
print('wave-max mean', w.mean(), 'percentiles', np.percentile(w, [10, 50, 90]))
