"""Print the gate's tables (quadrature.newton_start_grid / assemble_start) for the benchmark's spectra and for the reference's
bundled ones (golden case 0): steps the reference iteration needs on the cell corners of the data-space grid, open cells,
acceptance radii.  gpurun -- python tools/probes/gn_gate_table.py"""
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
golden = np.load(os.path.join(ROOT, 'tests', 'golden', 'gn_reference.npz'))
dev = torch.device('cuda')
lib = _native.load()
np.set_printoptions(linewidth=250)
cases = [('140 / 80 kVp Kramers (the benchmark)',) + md.decomposition_tables(ct, synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80))[1:]]
cases += [(f'golden case {ci}', golden[f'gn{ci}_i0'], golden[f'gn{ci}_mus']) for ci in range(3)]
for name, i0, mus in cases:
    p = q.newton_start_grid(i0, mus)
    n = int(p['head'][3])
    g_d = to_dev(np.ascontiguousarray(p['corner_g'].T), torch.float64, dev)
    n_c = g_d.shape[1]
    a_c = torch.empty((n_c, 2), dtype=torch.float64, device=dev)
    k_c = torch.empty(n_c, dtype=torch.uint8, device=dev)
    i0_d, mus_d = to_dev(i0[:, None, :], torch.float64, dev), to_dev(mus, torch.float64, dev)
    ws = torch.empty(lib.dexct_gn_workspace_bytes(i0.shape[1], 1), dtype=torch.uint8, device=dev)
    _native.check(lib.dexct_gn_decompose(ptr(g_d[0]), ptr(g_d[1]), 1, n_c, ptr(i0_d), ptr(mus_d), i0.shape[1], 1, 1, 254, 0, 0, None, 0.95,
                                         ptr(a_c), _native.gn_options(1e-12, 0, 0, 1, _native.GN_PASS_COARSE, k_c.data_ptr()), ptr(ws),
                                         stream_ptr()), 'cal')
    k = k_c.cpu().numpy().reshape(n + 1, n + 1)
    start, share0 = q.assemble_start(p, k_c.cpu().numpy(), a_c.cpu().numpy())
    # the same walk on the cell centres: the table checked against what it stands for
    gc = to_dev(np.ascontiguousarray(q.cell_centres(p).T), torch.float64, dev)
    a2 = torch.empty((gc.shape[1], 2), dtype=torch.float64, device=dev)
    k2 = torch.empty(gc.shape[1], dtype=torch.uint8, device=dev)
    _native.check(lib.dexct_gn_decompose(ptr(gc[0]), ptr(gc[1]), 1, gc.shape[1], ptr(i0_d), ptr(mus_d), i0.shape[1], 1, 1, 254, 0, 0, None, 0.95,
                                         ptr(a2), _native.gn_options(1e-12, 0, 0, 1, _native.GN_PASS_COARSE, k2.data_ptr()), ptr(ws),
                                         stream_ptr()), 'cal')
    start, share, n_bad = q.validate_start(start, p, k2.cpu().numpy(), a2.cpu().numpy())
    print(f'   open cells before / after the check at the centres: {share0:.3f} / {share:.3f} ({n_bad} centres failed)')
    cells = start[q.START_HEADER + 2 * (n + 1) ** 2:].reshape(n, n, 2)
    need, radius = cells[:, :, 0], cells[:, :, 1]
    print(f'== {name}: ratio u1/u0 from {p["head"][6]:.3f} in cells of {1 / p["head"][7]:.4f}; open cells {share:.3f}')
    print('steps at every 4th corner (rows: ln u0 from 1e-4 to 1, columns: ratio):')
    print(k[::4, ::4])
    print('open cells (1) at every 2nd cell:')
    print(np.isfinite(need)[::2, ::2].astype(int))
    with np.errstate(invalid='ignore'):
        print('need: min', np.nanmin(np.where(np.isfinite(need), need, np.nan)), 'max finite', np.nanmax(np.where(np.isfinite(need), need, np.nan)),
              '| radius relative to |a| at the open cells: median',
              np.nanmedian(np.where(np.isfinite(need), radius, np.nan) / np.maximum(np.abs(a_c.cpu().numpy()).max(1).reshape(n + 1, n + 1)[:-1, :-1], 1e-3)))
