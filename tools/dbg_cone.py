import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, synthetic
from conftest import small_scan
ct, ph = small_scan(n=40, nz=24, n_views=20, n_channels=48, n_rows=10)
for src_z, h_iso in ((0.0, 1e-9), (0.3, 0.8)):
    cone = dx.FanBeamGeometry(N_channels=48, N_proj=20, gamma_fan=0.8230337, SID=60.0, SDD=100.0, h_iso=h_iso,
                              eid=True, detector_file=ct.detector_file, N_rows=10, cone=True, src_z=src_z)
    sp = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
    res = {}
    for k in (1, 2):
        (c, pl), _ = fp.Projector(cone, ph, kernel=k).project(sp, want_pathlen=True)
        res[k] = (c.cpu().numpy(), pl.cpu().numpy())
    d = res[2][1] - res[1][1]
    bad = np.argwhere(np.abs(d).max(-1) > 0)
    print('src_z', src_z, 'h_iso', h_iso, 'rays differing', len(bad), 'of', d[..., 0].size, 'max abs diff', np.abs(d).max())
    for (v, r, c) in bad[:8]:
        print('  v,r,c', v, r, c, 'k1', res[1][1][v, r, c], 'k2', res[2][1][v, r, c])
    print('  sum over materials k1 vs k2 (first bad):', None if not len(bad) else (res[1][1][tuple(bad[0])].sum(), res[2][1][tuple(bad[0])].sum()))
