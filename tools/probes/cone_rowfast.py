"""cone_cols_kernel with row-fastest outputs (whole-line stores; DEXCT_CONE_ROWFAST=1) + the transpose pass into the reference's
order, against the kernel as it is (4-byte stores one line apart): time and bits.  The benchmark's cone-beam scan (100 x 800 x 512).
    gpurun -- python tools/probes/cone_rowfast.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import _native, forward_project as fp, synthetic
from dex_ct_sim_amd._device import ptr, stream_ptr

n, views, chans = 512, 100, 800
det = os.path.join(ROOT, 'dex-ct-sim_amd', 'input', 'detector', 'eta_eid_mv.bin')
ph_dz = 51.2 / n                               # detector rows one voxel apart at the isocentre, as bench.py's cone-beam leg
ct = dx.FanBeamGeometry(N_channels=chans, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, h_iso=ph_dz, eid=True, detector_file=det,
                        N_rows=n, cone=True)
ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
pj = fp.Projector(ct, ph)
_, mu_d, w_d, air = pj.upload_tables(specs)


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


ref_c, ref_l = pj.project_tables(mu_d, w_d, air=air)
t_ref = timed(lambda: pj.project_tables(mu_d, w_d, air=air, out=ref_c, log_out=ref_l))
os.environ['DEXCT_CONE_ROWFAST'] = '1'
nat_c, nat_l = torch.empty_like(ref_c), torch.empty_like(ref_l)
t_nat = timed(lambda: pj.project_tables(mu_d, w_d, air=air, out=nat_c, log_out=nat_l))
os.environ.pop('DEXCT_CONE_ROWFAST')
out_c, out_l = torch.empty_like(ref_c), torch.empty_like(ref_l)


def transposes():
    for src, dst in ((nat_c, out_c), (nat_l, out_l)):
        _native.check(pj.lib.dexct_transpose_batched(ptr(src), ptr(dst), 2 * views, chans, n, 4, stream_ptr()), 'transpose')


t_tr = timed(transposes)
print(f'cone kernel, outputs in the reference\'s order (4-byte stores one line apart): {t_ref:.3f} ms')
print(f'cone kernel, row-fastest outputs (whole lines): {t_nat:.3f} ms + transposes of counts and log {t_tr:.3f} ms = {t_nat + t_tr:.3f} ms')
print('same bits after the transpose:', bool(torch.equal(out_c, ref_c) and torch.equal(out_l, ref_l)))
