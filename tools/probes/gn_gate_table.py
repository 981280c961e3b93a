"""Print the gate's step table (quadrature.gate_table) for the benchmark's spectra: steps the reference iteration needs on the
cell corners, and which corners fail and why.  gpurun -- python tools/probes/gn_gate_table.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import _native, matdecomp as md, quadrature as q, synthetic
from dex_ct_sim_amd._device import ptr, stream_ptr, to_dev

det = os.path.join(ROOT, 'dex-ct-sim_amd', 'input', 'detector', 'eta_eid_mv.bin')
ct = dx.FanBeamGeometry(N_channels=8, N_proj=8, eid=True, detector_file=det)
_, i0, mus = md.decomposition_tables(ct, synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80))
p = q.newton_start_polynomial(i0, mus)
dev = torch.device('cuda')
g = np.nan_to_num(p['corner_g'], nan=1.0)
g_d = to_dev(np.ascontiguousarray(g.T), torch.float64, dev)
n_c = g.shape[0]
a_c = torch.empty((n_c, 2), dtype=torch.float64, device=dev)
k_c = torch.empty(n_c, dtype=torch.uint8, device=dev)
lib = _native.load()
i0_d, mus_d = to_dev(i0[:, None, :], torch.float64, dev), to_dev(mus, torch.float64, dev)
ws = torch.empty(lib.dexct_gn_workspace_bytes(i0.shape[1], 1), dtype=torch.uint8, device=dev)
_native.check(lib.dexct_gn_decompose(ptr(g_d[0]), ptr(g_d[1]), 1, n_c, ptr(i0_d), ptr(mus_d), i0.shape[1], 1, 1, 254, 0, 0, None, 0.95,
                                     ptr(a_c), _native.gn_options(1e-12, 0, 0, 1, _native.GN_PASS_COARSE, k_c.data_ptr()), ptr(ws),
                                     stream_ptr()), 'cal')
k = k_c.cpu().numpy().reshape(41, 41)
a = a_c.cpu().numpy()
err = (np.abs(a - p['corners']).max(1) / np.maximum(np.abs(p['corners']).max(1), 1)).reshape(41, 41)
np.set_printoptions(linewidth=250)
print('steps at the corners (rows: f0 index, columns: f1 index), first 16 x 16:')
print(k[:16, :16])
print('log10 error of the found a against the truth, first 12 x 12:')
with np.errstate(divide='ignore'):
    print(np.round(np.log10(err[:12, :12]), 1))
start, share = q.gate_table(p, k_c.cpu().numpy(), a)
need = start[q.START_HEADER + p['coef'].size:].reshape(40, 40)
print('need, first 16 x 16:')
print(need[:16, :16])
print('share of cells that allow the short cut', share)
