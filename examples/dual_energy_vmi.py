#!/usr/bin/env python3
"""The reference's whole workflow in one page, on the MI355X engine: simulate a dual-energy scan of a voxel phantom
(main.py:120), decompose it into basis-material sinograms (main.py:153), reconstruct them (main.py:168), form
virtual monoenergetic images and measure them (plots.py:136-158, 297-303, 386-395).

    python examples/dual_energy_vmi.py [--n 256] [--views 720] [--channels 512] [--dose 1e8] [--noise]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dex_ct_sim_amd as dx                                   # noqa: E402
from dex_ct_sim_amd import plots, synthetic                   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=256, help='phantom slice is n x n voxels over 51.2 cm')
    ap.add_argument('--views', type=int, default=720)
    ap.add_argument('--channels', type=int, default=512)
    ap.add_argument('--dose', type=float, default=1e8, help='photons per detector pixel and view, unattenuated')
    ap.add_argument('--noise', action='store_true', help='quantum noise for that dose')
    args = ap.parse_args()

    ct = dx.FanBeamGeometry(N_channels=args.channels, N_proj=args.views, gamma_fan=0.8230337, SID=60.0, SDD=100.0)
    phantom = synthetic.make_phantom(args.n, 1)                # water cylinder with bone inserts, one slice
    spec_hi, spec_lo = synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)
    for s in (spec_hi, spec_lo):
        s.rescale_counts(args.dose / s.I0.sum())

    t0 = time.perf_counter()
    raw_hi, _ = dx.get_sino(ct, phantom, spec_hi, noise=args.noise, seed=1)
    raw_lo, _ = dx.get_sino(ct, phantom, spec_lo, noise=args.noise, seed=2)
    m1_sino, m2_sino = dx.get_basismat_sinos(ct, raw_hi, raw_lo, spec_hi, spec_lo, n_iters=50)
    M1, _ = dx.get_recon(m1_sino, ct, spec_hi, args.n, 51.2, 0.8)       # basis images [g/cm^3]
    M2, _ = dx.get_recon(m2_sino, ct, spec_hi, args.n, 51.2, 0.8)
    print(f'projection x2, decomposition, reconstruction x2: {time.perf_counter() - t0:.2f} s')

    ids = phantom.volume[0]
    energies = np.arange(40, 141, 10)
    rmse = plots.vmi_rmse_sweep(energies, M1, M2, ids, phantom.mu_table(energies.astype(float)), mask=ids > 0)
    from scipy import ndimage
    comp, n_comp = ndimage.label(ndimage.binary_erosion(ids == 2, iterations=2))     # the largest bone insert
    if n_comp:
        sizes = ndimage.sum(np.ones_like(comp), comp, index=np.arange(1, n_comp + 1))
        ys, xs = np.nonzero(comp == 1 + int(np.argmax(sizes)))
        bone_roi = [int(xs.mean()) - 1, int(ys.mean()) - 1, 3, 3]
    else:
        bone_roi = [args.n // 2 + 10, args.n // 2, 3, 3]
    water_roi = [args.n // 2 - 4, args.n // 2 - 4, 8, 8]
    cnr = plots.vmi_roi_sweep(energies, M1, M2, bone_roi, water_roi)
    print(' keV   RMSE [HU]   signal [HU]   background [HU]   CNR')
    for k, e in enumerate(energies):
        print(f'{e:4d}  {rmse[k]:9.1f}  {cnr["u_signal"][k]:11.1f}  {cnr["u_background"][k]:15.1f}  {cnr["cnr"][k]:6.1f}')
    vmi70 = plots.make_vmi(70.0, M1, M2)
    u, v = plots.measure_roi(vmi70, water_roi)
    print(f'70 keV VMI, water ROI: mean {u:.1f} HU, sd {np.sqrt(v):.1f} HU')
    return rmse


if __name__ == '__main__':
    main()
