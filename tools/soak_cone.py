"""Randomised differential soak of the cone-beam kernels on one MI355X (a one-off campaign, log under profiles/): random
anisotropic grids, source heights and row pitches up to the kernel's limit (one z-plane per slab), 1..300 rows, 2..6
materials, dense random volumes or small objects in air.  Demanded in every case:

  * per-material path lengths of cone_kernel (one thread per ray) bit-identical to the oracle's mirror (CPU);
  * with <= 3 materials, path lengths AND counts of cone_rows_kernel (rows of a (view, channel) pair as lanes, shared
    in-plane records, guarded volume layout) bit-identical to cone_kernel's;
  * (round 6) with more, the group passes of the row-parallel kernels: path lengths bit-identical, counts identical for 5 - 48
    table rows; the noisy sample of the host's choice = dexct_add_noise on the same call's signal and variance.

    python tools/soak_cone.py [n_cases] [first_seed]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import _native, forward_project as fp
from dex_ct_sim_amd._device import ptr, stream_ptr
from dex_ct_sim_amd.system import AIR, BONE, WATER, Material
from oracle import c_oracle as co

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device('cuda:0')
t0 = time.time()
fails, rays, n_rows_kernel, n_groups = 0, 0, 0, 0
for case in range(n_cases):
    seed = seed0 + case
    rng = np.random.default_rng(330000 + seed)
    nx, ny, nz = int(rng.integers(9, 70)), int(rng.integers(9, 70)), int(rng.integers(2, 80))
    dxv, dyv, dzv = (float(v) for v in rng.uniform(0.08, 0.4, 3))
    half_diag = 0.5 * np.hypot(nx * dxv, ny * dyv)
    sid = float(half_diag * rng.uniform(1.3, 4.0))
    sdd = float(sid + half_diag * rng.uniform(1.0, 3.0))
    n_rows = int(rng.choice([1, 2, 7, 24, 31, 32, 33, 64, 100, 256, 300]))
    n_views, n_ch = int(rng.integers(1, 8)), int(rng.integers(1, 120))
    # the kernels take at most one z-plane per dominant-axis slab: |dz/du| <= 1 for every ray
    reach = 0.6 * sdd * dzv / (max(dxv, dyv) * np.sqrt(2.0))
    src_z = float(rng.uniform(-0.3, 0.3) * reach)
    half_rows = max(0.5 * (n_rows - 1), 0.5)
    h_iso = float((reach - abs(src_z)) / half_rows * sid / sdd * rng.uniform(0.05, 1.0))
    n_mat = int(rng.choice([2, 3, 3, 3, 4, 6, 9, 13]))
    vol = np.zeros((nz, ny, nx), dtype=np.uint8)
    style = rng.choice(['dense', 'blob', 'empty'])
    if style == 'dense':
        vol = rng.integers(0, n_mat, vol.shape, dtype=np.uint8)
        vol[rng.random(vol.shape) < 0.4] = 0
    elif style == 'blob':
        cx, cy = rng.integers(0, nx), rng.integers(0, ny)
        r = max(1, int(min(nx, ny) * rng.uniform(0.05, 0.4)))
        yy, xx = np.ogrid[:ny, :nx]
        disc = (xx - cx) ** 2 + (yy - cy) ** 2 <= r * r
        z0, z1 = sorted(rng.integers(0, nz, 2))
        vol[z0:z1 + 1, disc] = rng.integers(1, n_mat, (z1 - z0 + 1, int(disc.sum())), dtype=np.uint8)
    mats = ([AIR, WATER, BONE] + [Material(f'm{i}', 1.0 + 0.2 * i, 'H(11.2)O(88.8)') for i in range(3, n_mat)])[:n_mat]
    ph = dx.VoxelPhantom.from_array('soak', vol, mats, dx=dxv, dy=dyv, dz=dzv)
    bad = []
    try:
        cone = dx.FanBeamGeometry(N_channels=n_ch, N_proj=n_views, gamma_fan=float(rng.uniform(0.1, 1.8)), SID=sid, SDD=sdd,
                                  h_iso=h_iso, N_rows=n_rows, cone=True, src_z=src_z)
        n_e, n_s = int(rng.choice([1, 5, 64, 140])), int(rng.integers(1, 3))
        mu = rng.uniform(0.01, 0.4, (n_mat, n_e)).astype(np.float32)
        mu[0] *= 1e-3
        w = rng.uniform(0.5, 2.0, (n_s, n_e)).astype(np.float32)
        mu_d, w_d = torch.from_numpy(mu).to(dev), torch.from_numpy(w).to(dev)
        g = co.make_geom(n_views, n_ch, n_rows, 0, nx, ny, nz, dxv, dyv, dzv, sid, sdd)
        _, rpl = co.project_cone(g, cone.view_cs(), cone.chan_cs(), 0, n_views, cone.row_z(), src_z, vol, mu, w, dda=True,
                                 n_threads=8)
        pj1 = fp.Projector(cone, ph, kernel=1)
        c1, p1 = pj1.project_tables(mu_d, w_d, want_pathlen=True)
        # (round 4: the kernels see compact ids - ids the volume does not hold are dropped, Projector.mat_rows)
        if not np.array_equal(p1.cpu().numpy(), rpl[..., pj1.mat_rows]):
            bad.append('cone_kernel vs the oracle mirror: path lengths differ')
        if pj1.n_mat <= 3:
            c2, p2 = fp.Projector(cone, ph, kernel=2).project_tables(mu_d, w_d, want_pathlen=True)
            n_rows_kernel += 1
            if not torch.equal(p2, p1):
                bad.append(f'cone_rows_kernel: path lengths differ ({int((p2 != p1).sum())} values)')
            if not torch.equal(c2, c1):
                bad.append(f'cone_rows_kernel: counts differ ({int((c2 != c1).sum())} values)')
        # round 6: more than 3 table rows in group passes of the row-parallel kernels (kernel=2 asks for them whatever the row
        # count): path lengths bit-identical to cone_kernel, counts identical where the detection arithmetic is the same (5 - 48
        # table rows), else to 2e-6
        if pj1.n_mat > 3:
            pjg = fp.Projector(cone, ph, kernel=2)
            if pjg.cone_groups:
                cg, pg = pjg.project_tables(mu_d, w_d, want_pathlen=True)
                n_groups += 1
                if not torch.equal(pg, p1):
                    bad.append(f'cone group passes: path lengths differ ({int((pg != p1).sum())} values)')
                if pj1.n_mat > 4 and not torch.equal(cg, c1):
                    bad.append(f'cone group passes: counts differ ({int((cg != c1).sum())} values)')
                if not torch.allclose(cg, c1, rtol=2e-6, atol=0):
                    bad.append(f'cone group passes: counts off by {float(((cg - c1).abs() / c1.abs().clamp_min(1e-30)).max()):.2e}')
        # round 6: quantum noise - what the host picks (variance + sample inside the cone kernels / the group passes' detection
        # pass, or variance output + dexct_add_noise) draws the sample dexct_add_noise draws from the same call's signal and variance
        auto = fp.Projector(cone, ph)
        w_n, w_nd = w, w_d
        if rng.random() < 0.3:                   # three or four spectra: beyond what the fused forms of the stacked fan hold (fallback)
            w_n = np.concatenate([w, (0.7 * w[::-1]).astype(np.float32), (1.3 * w).astype(np.float32)])[:int(rng.integers(3, 5))]
            w_nd = torch.from_numpy(np.ascontiguousarray(w_n)).to(dev)
        w2_d = torch.from_numpy((w_n * rng.uniform(0.5, 3.0, w_n.shape)).astype(np.float32)).to(dev)
        nseed = int(rng.integers(0, 2 ** 31))
        noisy, var = auto.project_tables(mu_d, w_nd, layout=None, w2_d=w2_d, seed=nseed, want_variance=True)
        sampled = auto.project_tables(mu_d, w_nd, layout=None).clone()
        _native.check(auto.lib.dexct_add_noise(ptr(sampled), ptr(var), w_n.shape[0], n_views, n_rows, n_ch, auto.native_layout, 0, nseed,
                                               stream_ptr()), 'dexct_add_noise')
        if not torch.equal(noisy, sampled):
            bad.append(f'noise: the host-choice sample differs from dexct_add_noise on its own variance ({int((noisy != sampled).sum())} values)')
    except Exception as exc:
        bad = [f'{type(exc).__name__}: {exc}']
    rays += n_views * n_rows * n_ch
    if bad:
        fails += 1
        print(f'FAIL seed {seed}: grid {nx}x{ny}x{nz}, {n_rows} rows, {n_views} views x {n_ch} ch, {n_mat} materials, {style}: '
              + '; '.join(bad), flush=True)
    if case % 500 == 499 or case == n_cases - 1:
        print(f'{case + 1} cases, {fails} failed, {rays:.3g} rays, {n_rows_kernel} cases through both kernels, {n_groups} through the group passes, every case with the noise leg (round 6), {time.time() - t0:.0f} s',
              flush=True)
sys.exit(1 if fails else 0)
