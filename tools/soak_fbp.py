"""Randomised soak of the fan-beam FBP kernels (filter + back-projection) against the float64 oracle on one MI355X: random
fan geometries, matrix sizes, fields of view (inside the fan's field of view or beyond), ramp cutoffs, apodisation
windows and 1..17 detector rows (1 takes the single-slice kernel, >= 8 the shared-geometry one).  Tolerance 5e-5 of the
image maximum (float32 filtered values, float64 geometry).  A one-off campaign; the worst case is printed.

    python tools/soak_fbp.py [n_cases] [first_seed]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import back_project as bp
from oracle import fbp_oracle as fo

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0, fails, worst = time.time(), 0, 0.0
for case in range(n_cases):
    seed = seed0 + case
    rng = np.random.default_rng(440000 + seed)
    n_ch, n_views = int(rng.integers(8, 260)), int(rng.integers(4, 160))
    rows = int(rng.choice([1, 1, 2, 8, 9, 17]))
    sid = float(rng.uniform(30.0, 80.0))
    sdd = float(sid * rng.uniform(1.2, 2.0))
    fan = float(rng.uniform(0.2, 1.3))
    n_mat = int(rng.integers(4, 130))
    fov = float(rng.uniform(0.2, 1.1) * 2 * sid * np.sin(0.5 * fan))
    ramp = float(rng.uniform(0.1, 1.0))
    window = str(rng.choice(['rect', 'rect', 'sinc', 'cosine', 'hann', 'hamming']))
    bad = []
    try:
        ct = dx.FanBeamGeometry(N_channels=n_ch, N_proj=n_views, gamma_fan=fan, SID=sid, SDD=sdd, N_rows=rows)
        stack = rng.uniform(0.0, 4.0, (n_views, rows, n_ch)).astype(np.float32)
        sino = stack if rows > 1 else stack[:, 0]
        img = bp.recon_device(torch.tensor(sino, device='cuda'), ct, n_mat, fov, ramp, window=window).cpu().numpy()
        img = img.reshape(rows, n_mat, n_mat)
        for r in sorted({0, rows - 1}):
            ref, _ = fo.get_recon(np.ascontiguousarray(stack[:, r]), ct.thetas, ct.gammas, sid, n_mat, fov, ramp, window=window)
            err = float(np.max(np.abs(img[r] - ref)) / np.abs(ref).max())
            worst = max(worst, err)
            if not err < 5e-5:
                bad.append(f'row {r}: {err:.2e} of the image maximum')
    except Exception as exc:
        bad = [f'{type(exc).__name__}: {exc}']
    if bad:
        fails += 1
        print(f'FAIL seed {seed}: {n_views} views x {rows} rows x {n_ch} ch, matrix {n_mat}, fov {fov:.1f}, ramp {ramp:.2f}, {window}: '
              + '; '.join(bad), flush=True)
    if case % 100 == 99 or case == n_cases - 1:
        print(f'{case + 1} cases, {fails} failed, worst {worst:.2e} of the image maximum, {time.time() - t0:.0f} s', flush=True)
sys.exit(1 if fails else 0)
