"""Where the time of get_basismat_sinos goes at the reference's own size (1200 x 800, one row): phases timed with a
synchronisation in between (so the sum is an upper bound of the pipelined call), and the call itself."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import _native, matdecomp as md, synthetic
from dex_ct_sim_amd._device import device, ptr, stream_ptr, to_host

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
ct = dx.FanBeamGeometry(N_channels=800, N_proj=1200, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=1)
ph = synthetic.make_phantom(512, 1, extent=51.2, seed=1234)
s1, s2 = synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)
(r1, _), (r2, _) = dx.get_sinos(ct, ph, [s1, s2])
lib = _native.load()
dev = device()


def t(f, n=20):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = f()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


ms, (ee, i0, mus) = t(lambda: md.decomposition_tables(ct, s1, s2)); print(f'decomposition_tables      {ms:.3f} ms')
ms, g1 = t(lambda: md._as_device_counts(r1, dev)); print(f'H2D one sinogram          {ms:.3f} ms')
g2 = md._as_device_counts(r2, dev)
gmax = torch.empty((), dtype=torch.float64, device=dev)
ms, _ = t(lambda: lib.dexct_reduce_max(ptr(g1), 0, g1.numel(), ptr(gmax), stream_ptr())); print(f'reduce_max                {ms:.3f} ms')
ms, a = t(lambda: md.gn_device(g1, g2, i0, mus, 50, mask_max=gmax)); print(f'gn_device (tables + kernel) {ms:.3f} ms')
i0_d, mus_d = torch.tensor(i0, device=dev), torch.tensor(mus, device=dev)
ms, a = t(lambda: md.gn_device(g1, g2, i0_d, mus_d, 50, mask_max=gmax)); print(f'gn_device, tables resident {ms:.3f} ms')
ms, h = t(lambda: to_host(a)); print(f'D2H result ({a.numel() * 8 / 1e6:.1f} MB)      {ms:.3f} ms')
ms, _ = t(lambda: md.get_basismat_sinos(ct, r1, r2, s1, s2, n_iters=50)); print(f'get_basismat_sinos        {ms:.3f} ms')
