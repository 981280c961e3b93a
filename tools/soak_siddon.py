"""Randomised differential soak of the stacked-fan traversal kernels on one MI355X (not part of the test suite: a
one-off campaign whose log is kept under profiles/).  Every case draws a geometry (grid 9..90 voxels per side,
anisotropic voxels, fan narrower or much wider than the grid, 16..1400 rows with an arbitrary first slice), a volume
(blobs in air so that whole waves of rays see nothing but air, or dense random voxels so that every crossing slab
corrects) and detection tables (1 or 2 spectra, random runs of zero weights, 1..300 energies) and demands

  * per-material path lengths of kernels 3 (rows4), 5 (tiled), 7 (rows16, 2-bit volume; 2..4 materials) and of the
    material-group forms 4 (byte codes) and 8 (2-bit codes; 5..9 materials here) bit-identical to kernel 1 (one thread
    per ray: no shared lists, no packed counters, no detection shortcuts), counts bit-identical up to 4 materials and
    within 1e-5 above (kernel 1 then detects from LDS columns), and kernels 4 and 8 bit-identical to each other;
  * the path lengths of a random block of rows bit-identical to the oracle's mirror (CPU).

    python tools/soak_siddon.py [n_cases] [first_seed]
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


def rel(a, b):
    """largest difference relative to b (a spectrum of all-zero weights gives counts 0: then absolute)"""
    return float(((a - b).abs() / b.abs().clamp_min(1e-30)).max())


t0 = time.time()
fails = 0
stats = {'packed_auto': 0, 'air_cases': 0, 'rays': 0, 'hit': 0, 'counts_sum': 0.0}
for case in range(n_cases):
    seed = seed0 + case
    rng = np.random.default_rng(910000 + seed)
    nx, ny = int(rng.integers(9, 90)), int(rng.integers(9, 90))
    n_rows = int(rng.choice([16, 48, 64, 100, 192, 256, 256, 320, 512, 512, 700, 768, 1024, 1100, 1400]))
    z_index = int(rng.integers(0, 6))
    nz = n_rows + z_index + int(rng.integers(0, 5))
    dxv, dyv, dzv = (float(v) for v in rng.uniform(0.1, 0.5, 3))
    half_diag = 0.5 * np.hypot(nx * dxv, ny * dyv)
    sid = float(half_diag * rng.uniform(1.2, 4.0))
    sdd = float(sid + half_diag * rng.uniform(1.05, 3.0))
    n_views, n_ch = int(rng.integers(1, 10)), int(rng.integers(1, 200))
    n_mat = int(rng.choice([2, 3, 3, 4, 4, 5, 7, 9]))
    style = rng.choice(['blob', 'dense', 'empty', 'slab'])
    vol = np.zeros((nz, ny, nx), dtype=np.uint8)
    if style == 'dense':
        vol = rng.integers(0, n_mat, vol.shape, dtype=np.uint8)
        vol[rng.random(vol.shape) < 0.3] = 0
    elif style == 'blob':                      # a small object in a wide field: most rays miss it
        cx, cy = rng.integers(0, nx), rng.integers(0, ny)
        r = max(1, int(min(nx, ny) * rng.uniform(0.05, 0.3)))
        yy, xx = np.ogrid[:ny, :nx]
        disc = (xx - cx) ** 2 + (yy - cy) ** 2 <= r * r
        z0, z1 = sorted(rng.integers(0, nz, 2))
        vol[z0:z1 + 1, disc] = 1
        if n_mat > 2:
            inner = disc & (rng.random((ny, nx)) < 0.3)
            vol[z0:z1 + 1][:, inner] = rng.integers(2, n_mat, int(inner.sum()), dtype=np.uint8)
        stats['air_cases'] += 1
    elif style == 'slab':                      # z-invariant except a few slices
        vol[:, ny // 4:3 * ny // 4, nx // 4:3 * nx // 4] = 1
        vol[rng.integers(0, nz, 3)] = n_mat - 1
    mats = ([AIR, WATER, BONE] + [Material(f'm{i}', 1.0 + 0.1 * i, 'H(11.2)O(88.8)') for i in range(3, n_mat)])[:n_mat]
    ph = dx.VoxelPhantom.from_array('soak', vol, mats, dx=dxv, dy=dyv, dz=dzv, z_index=z_index)
    ct = dx.FanBeamGeometry(N_channels=n_ch, N_proj=n_views, gamma_fan=float(rng.uniform(0.1, 2.2)), SID=sid, SDD=sdd,
                            N_rows=n_rows)
    n_e, n_s = int(rng.choice([1, 3, 4, 7, 64, 140, 140, 141, 300])), int(rng.integers(1, 3))
    mu = rng.uniform(0.01, 0.4, (n_mat, n_e)).astype(np.float32)
    mu[0] *= 1e-3
    w = rng.uniform(0.5, 2.0, (n_s, n_e)).astype(np.float32)
    for s in range(n_s):
        b = 0
        while b < n_e:                          # runs of zero weights of random length
            run = int(rng.integers(1, 40))
            if rng.random() < 0.4:
                w[s, b:b + run] = 0.0
            b += run
    mu_d, w_d = torch.from_numpy(mu).to(dev), torch.from_numpy(w).to(dev)
    try:
        pj1 = fp.Projector(ct, ph, kernel=1)
        ref, pl = pj1.project_tables(mu_d, w_d, want_pathlen=True)
        n_mat = pj1.n_mat                       # the table rows the kernels work with (compact ids)
        bad = []
        stats['hit'] += int((pl[..., 1:].sum(dim=-1) > 0).sum())
        stats['counts_sum'] += float(ref.double().sum())
        for kernel in ((3, 5, 7, 8) if n_mat <= 4 else (4, 8)):
            got, gpl = fp.Projector(ct, ph, kernel=kernel).project_tables(mu_d, w_d, want_pathlen=True)
            if not torch.equal(gpl, pl):
                bad.append(f'kernel {kernel}: path lengths differ ({int((gpl != pl).sum())} values)')
            if n_mat <= 4 and not torch.equal(got, ref):
                bad.append(f'kernel {kernel}: counts differ ({int((got != ref).sum())} values)')
            if n_mat > 4:
                if not rel(got, ref) <= 1e-5:
                    bad.append(f'kernel {kernel}: counts off by {rel(got, ref):.2e}')
                if kernel == 4:
                    ref4 = got
                elif not torch.equal(got, ref4):
                    bad.append('kernel 8: counts differ from kernel 4')
        auto = fp.Projector(ct, ph)
        stats['packed_auto'] += int(bool(auto.use_packed or auto.grouped_packed))
        air = w.astype(np.float64).sum(axis=1)
        got0, log0 = auto.project_tables(mu_d, w_d, air=air)          # round 3: the log sinogram from the detection store
        want_log = torch.log(torch.tensor(air, dtype=torch.float32, device=dev)[:, None, None, None] / got0)
        live = (got0 > 0) & torch.isfinite(want_log)
        if not torch.allclose(log0[live], want_log[live], rtol=5e-6, atol=5e-7):
            bad.append(f'kernel 0: log sinogram off by {float((log0[live] - want_log[live]).abs().max()):.2e}')
        if n_mat <= 4:
            same0 = torch.equal(got0, ref)
        elif auto.grouped or auto.grouped_packed:
            same0 = torch.equal(got0, ref4)
        else:                                   # few rows: a one-row-per-lane kernel with LDS accumulators
            same0 = rel(got0, ref) <= 1e-5
        if not same0:
            bad.append('kernel 0 (host choice): counts differ')
        # round 6: quantum noise.  Whatever the host picks (variance + sample inside rows16_kernel<NOISY> / the group passes'
        # detection kernel, or a variance output + dexct_add_noise) must draw the sample dexct_add_noise draws from the same
        # call's signal and variance, deliver that variance, and - up to 4 table rows - agree bit for bit with the byte-volume
        # path (kernel 3: separate variance loop + dexct_add_noise)
        w_n, w_nd = w, w_d
        if rng.random() < 0.3:                   # three or four spectra: beyond what the fused forms of the stacked fan hold (fallback)
            w_n = np.concatenate([w, (0.7 * w[::-1]).astype(np.float32), (1.3 * w).astype(np.float32)])[:int(rng.integers(3, 5))]
            w_nd = torch.from_numpy(np.ascontiguousarray(w_n)).to(dev)
        w2_d = torch.from_numpy((w_n * rng.uniform(0.5, 3.0, w_n.shape)).astype(np.float32)).to(dev)
        nseed = int(rng.integers(0, 2 ** 31))
        noisy, var = auto.project_tables(mu_d, w_nd, layout=None, w2_d=w2_d, seed=nseed, want_variance=True)
        clean = auto.project_tables(mu_d, w_nd, layout=None)
        sampled = clean.clone()
        _native.check(auto.lib.dexct_add_noise(ptr(sampled), ptr(var), w_n.shape[0], n_views, n_rows, n_ch, auto.native_layout, 0, nseed,
                                               stream_ptr()), 'dexct_add_noise')
        if not torch.equal(noisy, sampled):
            bad.append(f'noise: the host-choice sample differs from dexct_add_noise on its own variance ({int((noisy != sampled).sum())} values)')
        if n_mat <= 4:
            n3, v3 = fp.Projector(ct, ph, kernel=3).project_tables(mu_d, w_nd, layout=auto.native_layout, w2_d=w2_d, seed=nseed, want_variance=True)
            if not (torch.equal(n3, noisy) and torch.equal(v3, var)):
                bad.append(f'noise: sample / variance differ from the byte-volume path ({int((n3 != noisy).sum())} / {int((v3 != var).sum())} values)')
        stats['noisy'] = stats.get('noisy', 0) + 1
        nsub = min(8, n_rows)
        r0 = int(rng.integers(0, n_rows - nsub + 1))
        sub = co.make_geom(n_views, n_ch, nsub, z_index + r0, nx, ny, nz, dxv, dyv, dzv, sid, sdd)
        _, rpl = co.project_dda(sub, ct.view_cs(), ct.chan_cs(), 0, n_views, vol, mu, w, True, n_threads=8)
        # (round 4: the kernels see compact ids - ids the scanned slices do not hold are dropped, Projector.mat_rows)
        if not np.array_equal(pl[:, r0:r0 + nsub].cpu().numpy(), rpl[..., pj1.mat_rows]):
            bad.append('kernel 1 vs the oracle mirror: path lengths differ')
    except Exception as exc:                    # a refusal is a finding too
        bad = [f'{type(exc).__name__}: {exc}']
    stats['rays'] += n_views * n_rows * n_ch
    if bad:
        fails += 1
        print(f'FAIL seed {seed}: grid {nx}x{ny}x{nz}, rows {n_rows} from {z_index}, {n_views} views x {n_ch} ch, {n_mat} materials, '
              f'{style}, {n_s} spectra x {n_e} energies: ' + '; '.join(bad), flush=True)
    if case % 500 == 499 or case == n_cases - 1:
        print(f'{case + 1} cases, {fails} failed, {stats["rays"]:.3g} rays ({stats["hit"] / max(stats["rays"], 1):.2f} of them through the '
              f'object, sum of all counts {stats["counts_sum"]:.6g}), host picked the packed kernel in {stats["packed_auto"]}, '
              f'{stats["air_cases"]} small-object cases, {stats.get("noisy", 0)} with the noise leg (round 6), {time.time() - t0:.0f} s', flush=True)
sys.exit(1 if fails else 0)
