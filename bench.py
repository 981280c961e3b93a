#!/usr/bin/env python3
"""Benchmark of the dex-ct hot path on MI355X.

A step = one pass of the hot path over BASELINE.json's metric configuration (configs[2]):
512^3 synthetic water/bone phantom, 1000 views x 800 channels, dual 80/140 kVp spectra, i.e.
  plan -> Siddon traversal + polychromatic detection of BOTH spectra (one fused traversal)
       -> Gauss-Newton decomposition (50 iterations, as main.py:153) + air mask
       -> (N > 1) all-gather of the two raw sinograms over RCCL, overlapped with the decomposition.
Detector rows: BASELINE.json does not name a row count and a single row touches one slice of the
512^3 volume, so the workload is the stacked fan N_rows = Nz = 512; the
single-row case is reported under "single_row".  Inputs are resident in HBM before the timed region.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: projection angles are sharded; weak scaling - every rank projects `--views` angles of an
N x views scan of the same phantom.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # vector FP64


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--n', type=int, default=512, help='phantom is n^3')
    ap.add_argument('--views', type=int, default=1000, help='views per GPU')
    ap.add_argument('--channels', type=int, default=800)
    ap.add_argument('--rows', type=int, default=0, help='detector rows (0: n)')
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--gn-precision', default=None, choices=[None, 'f64', 'mixed'])
    ap.add_argument('--kernel', type=int, default=0, help='0 choose, 1 ray-parallel, 2 row-parallel')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    ap.add_argument('--skip-single-row', action='store_true')
    ap.add_argument('--skip-gn-full-loop', action='store_true',
                    help='omit the extra full-loop Newton launch (keeps rocprof per-kernel averages clean)')
    return ap.parse_args()


def segment_count(co, geom, view_cs, chan_cs, n_views_total, view_begin, view_end):
    """Exact number of Siddon segments per (view, channel) of this rank's shard, from the CPU oracle."""
    plan = co.plan(geom, view_cs, chan_cs, view_begin, view_end)
    return int(co.count_segments(geom, plan)), plan


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        # RCCL ("nccl") over xGMI; DEXCT_DIST_BACKEND=gloo only for rehearsing the N > 1 control flow on a
        # single-GPU box (ranks then share one device and collectives are staged through the host)
        backend = os.environ.get('DEXCT_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import _shard, forward_project as fp, matdecomp as md, synthetic

    dev = torch.device('cuda', local_rank)
    n, rows = args.n, (args.rows or args.n)
    total_views = args.views * world
    det = os.path.join(ROOT, 'dex-ct-sim_amd', 'input', 'detector', 'eta_eid_mv.bin')
    ct = dx.FanBeamGeometry(N_channels=args.channels, N_proj=total_views, gamma_fan=0.8230337, SID=60.0, SDD=100.0,
                            eid=True, detector_file=det, N_rows=rows)
    ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
    specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
    vb, ve = _shard.split(total_views, rank, world)
    pj = fp.Projector(ct, ph, view_range=(vb, ve), kernel=args.kernel)
    E, mu_d, w_d, air = pj.upload_tables(specs)
    n_e_spec = [int((w_d[k] != 0).sum().item()) for k in range(2)]
    _, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
    i0_d = torch.tensor(i0, dtype=torch.float64, device=dev)
    mus_d = torch.tensor(mus, dtype=torch.float64, device=dev)
    lib = pj.lib
    import ctypes as C
    from dex_ct_sim_amd import _native
    from dex_ct_sim_amd._device import ptr, stream_ptr

    nV = pj.n_local_views
    n_rays = nV * rows * args.channels
    native = pj.native_layout          # 1: [view][channel][row] (row-parallel kernels), 0: [view][row][channel]
    nat_shape = (nV, args.channels, rows) if native == 1 else (nV, rows, args.channels)
    counts_nat = torch.empty((2,) + nat_shape, dtype=torch.float32, device=dev)
    a_nat = torch.empty(nat_shape + (2,), dtype=torch.float64, device=dev)
    # results in the reference's order ([view][row][channel]) are part of the step
    counts = torch.empty((2, nV, rows, args.channels), dtype=torch.float32, device=dev) if native == 1 else counts_nat
    a_out = torch.empty((nV, rows, args.channels, 2), dtype=torch.float64, device=dev) if native == 1 else a_nat
    gmax = torch.empty((), dtype=torch.float64, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    precision = args.gn_precision or md.DEFAULT_PRECISION

    def step(timed):
        st = stream_ptr()
        _native.check(lib.dexct_fan_plan(C.byref(pj.geom), ptr(pj.view_cs), ptr(pj.chan_cs), vb, ve, ptr(pj.plan), st),
                      'plan')
        if timed:
            ev[0].record()
        pj.project_tables(mu_d, w_d, out=counts_nat, layout=None)
        if timed:
            ev[1].record()
        _native.check(lib.dexct_reduce_max(ptr(counts_nat[0]), 0, counts_nat[0].numel(), ptr(gmax), st), 'max')
        gm = _shard.global_max(gmax)
        finish_gather = None
        if world > 1:
            # the one data-path collective: assemble the raw sinograms (reference order) on every rank; it is
            # started here and overlaps the Newton kernel, which only needs the local shard
            if native == 1:
                _native.check(lib.dexct_transpose_batched(ptr(counts_nat), ptr(counts), 2 * nV, args.channels, rows,
                                                          4, st), 'transpose counts')
            finish_gather = _shard.gather_views(counts, total_views, view_dim=1, async_op=True)
        if timed:
            ev[2].record()
        # air mask fused into the Newton kernel: threshold = 0.95 * (all-reduced) max, read from the device scalar
        md.gn_device(counts_nat[0], counts_nat[1], i0_d, mus_d, args.iters, precision, out=a_nat, mask_max=gm,
                     mask_frac=0.95)
        if timed:
            ev[3].record()
        if native == 1:       # hand the results over in the reference's [view][row][channel] order
            if world == 1:
                _native.check(lib.dexct_transpose_batched(ptr(counts_nat), ptr(counts), 2 * nV, args.channels, rows,
                                                          4, st), 'transpose counts')
            _native.check(lib.dexct_transpose_batched(ptr(a_nat), ptr(a_out), nV, args.channels, rows, 16, st),
                          'transpose mats')
        if world > 1:
            # basis-material sinograms stay view-sharded (each rank owns its angles, as a view-sharded
            # back-projection would consume them); only the raw sinogram is assembled, as the north star says
            return finish_gather(), a_out
        return counts, a_out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    barrier()
    t_sid, t_gn = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
        torch.cuda.synchronize()
        t_sid.append(ev[0].elapsed_time(ev[1]))
        t_gn.append(ev[2].elapsed_time(ev[3]))
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor(elapsed, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    integrals_per_step = world * n_rays * sum(n_e_spec)
    value = integrals_per_step / (elapsed / args.steps)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    sid_ms, gn_ms = float(np.mean(t_sid)), float(np.mean(t_gn))
    out = {
        'metric': 'Siddon ray-energy integrals/sec', 'value': value, 'unit': 'ray-energy integrals/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f64' if precision == 'f64' else 'f32+f64', 'data': 'synthetic',
        'config': {'workload': f'{n}^3 water/bone phantom, {args.views} views/GPU x {args.channels} channels x '
                               f'{rows} rows (stacked fan), dual 140/80 kVp Kramers spectra ({n_e_spec[0]}+{n_e_spec[1]} '
                               f'energy bins), fused dual-spectrum Siddon + {args.iters}-iteration Gauss-Newton',
                   'rays_per_gpu': n_rays, 'parallelism': f'views sharded x{world}',
                   'arithmetic': 'voxel indices int64 fixed point, path lengths + detection f32, Newton '
                                 + ('f64 (reference arithmetic)' if precision == 'f64' else 'f32 bulk + f64 polish')},
        'kernel_ms': {'siddon_project': sid_ms, 'gn_decompose': gn_ms},
        'siddon_only_integrals_per_s': n_rays * sum(n_e_spec) / (sid_ms * 1e-3),
        'siddon_rays_per_s': n_rays / (sid_ms * 1e-3),
        'gn_pixel_solves_per_s': n_rays / (gn_ms * 1e-3),
    }

    # ---- roofline of the traversal kernel: algorithmic bytes = exact segment count x 1 B + outputs
    from oracle import c_oracle as co
    geom = co.make_geom(ct.N_proj, ct.N_channels, rows, ph.z_index, n, n, n, ph.dx, ph.dy, ph.dz, ct.SID, ct.SDD)
    seg_vc, _ = segment_count(co, geom, ct.view_cs(), ct.chan_cs(), total_views, vb, ve)
    alg_bytes = seg_vc * rows * 1 + 4 * 2 * n_rays
    achieved = alg_bytes / (sid_ms * 1e-3) / 1e9
    kname = {1: 'rays_kernel', 2: 'rows_kernel', 3: 'rows4_kernel'}[args.kernel or (3 if native == 1 else 1)]
    traffic = None
    import glob
    pmc = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json')))
    if pmc:      # newest committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/profile_gpu.sh)
        j = json.load(open(pmc[-1]))
        if j.get('rays_per_gpu') == n_rays and kname in j.get('siddon_kernel', ''):   # same workload and kernel only
            traffic = j.get('siddon_hbm_bytes_per_launch')
    out['roofline'] = {'kernel': kname,
                       'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                       'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                       'traffic_GBps': None if traffic is None else traffic / (sid_ms * 1e-3) / 1e9,
                       'note': 'achieved = ALGORITHMIC bytes / launch time; the 128 MiB volume is resident in L2 + '
                               'Infinity Cache, so it can exceed the HBM peak; traffic = PMC FETCH_SIZE + WRITE_SIZE '
                               '(fabric side, calibrated as in profiles/*_summary.md).  True-HBM regime (1024^3 volume): '
                               'profiles/r01d_1024_*',
                       'algorithmic_bytes_per_launch': alg_bytes, 'segments_per_launch': seg_vc * rows,
                       'avg_launch_ms': sid_ms}
    # air pixels (zeroed by the mask afterwards, matdecomp.py:204-205) are not iterated: count the others only
    masked = float((counts_nat[0] >= 0.95 * gmax).float().mean().item())
    out['gn_masked_fraction'] = masked
    gn_flops = (1.0 - masked) * n_rays * args.iters * i0.shape[1] * (28 + 1)   # 28 flops + 1 exp per energy-iteration
    out['roofline_gn'] = {'kernel': 'gn_refill_kernel' if precision == 'f64' else 'gn_kernel<true,false>', 'bound': 'valu_fp64' if precision == 'f64' else 'valu_fp32+fp64',
                          'achieved': gn_flops / (gn_ms * 1e-3) / 1e12, 'peak': FP64_VALU_PEAK_TFLOPS,
                          'unit': 'TFLOP/s', 'frac': gn_flops / (gn_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                          'avg_launch_ms': gn_ms,
                          'note': 'ALGORITHMIC flops of n_iters iterations (exp counted as 1 flop; unmasked pixels only; '
                                  'not HBM bound, 24 B/pixel).  Iterations the exact repeated-state exit skips are counted '
                                  'as done; full_loop_* is the same launch with every iteration executed '
                                  '(DEXCT_GN_FULL_LOOP=1), i.e. the executed-flop rate of the kernel'}
    if precision == 'f64' and world == 1 and not args.skip_gn_full_loop:
        os.environ['DEXCT_GN_FULL_LOOP'] = '1'
        try:
            a_full = torch.empty_like(a_nat)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            md.gn_device(counts_nat[0], counts_nat[1], i0_d, mus_d, args.iters, 'f64', out=a_full, mask_max=gmax,
                         mask_frac=0.95)
            e1.record()
            torch.cuda.synchronize()
        finally:
            os.environ.pop('DEXCT_GN_FULL_LOOP', None)
        full_ms = e0.elapsed_time(e1)
        out['roofline_gn'].update({'full_loop_ms': full_ms, 'full_loop_achieved': gn_flops / (full_ms * 1e-3) / 1e12,
                                   'full_loop_frac': gn_flops / (full_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                   'full_loop_bit_identical': bool(torch.equal(a_full.view(torch.int64),
                                                                               a_nat.view(torch.int64)))})

    # ---- opt-in tolerance stop (float64, DEXCT_GN_STOP_TOL), never part of `value`: what giving up "exactly the
    # reference's 50 iterations" would buy
    if precision == 'f64' and world == 1 and not args.skip_gn_full_loop:
        os.environ['DEXCT_GN_STOP_TOL'] = '1e-12'
        try:
            a_tol = torch.empty_like(a_nat)
            md.gn_device(counts_nat[0], counts_nat[1], i0_d, mus_d, args.iters, 'f64', out=a_tol, mask_max=gmax,
                         mask_frac=0.95)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            md.gn_device(counts_nat[0], counts_nat[1], i0_d, mus_d, args.iters, 'f64', out=a_tol, mask_max=gmax,
                         mask_frac=0.95)
            e1.record()
            torch.cuda.synchronize()
        finally:
            os.environ.pop('DEXCT_GN_STOP_TOL', None)
        diff = ((a_tol - a_nat).abs() / a_nat.abs().clamp(min=1.0))
        out['gn_stop_tol'] = {'tol': 1e-12, 'gn_ms': e0.elapsed_time(e1),
                              'max_diff_vs_exact': float(torch.nan_to_num(diff, nan=0.0).max().item()),
                              'note': 'DEXCT_GN_STOP_TOL=1e-12: float64, a pixel also stops when a step moves it by '
                                      '<= tol * max(|a|, 1); opt-in, not the fixed iteration count of the reference, '
                                      'not used for value'}
        del a_tol

    # ---- opt-in mixed-precision Newton (float32 bulk + float64 polish), never part of `value`
    if precision == 'f64' and world == 1:
        a_mixed = torch.empty_like(a_nat)
        md.gn_device(counts_nat[0], counts_nat[1], i0_d, mus_d, args.iters, 'mixed', out=a_mixed, mask_max=gmax,
                     mask_frac=0.95)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        md.gn_device(counts_nat[0], counts_nat[1], i0_d, mus_d, args.iters, 'mixed', out=a_mixed, mask_max=gmax,
                     mask_frac=0.95)
        e1.record()
        torch.cuda.synchronize()
        diff = ((a_mixed - a_nat).abs() / a_nat.abs().clamp(min=1.0))
        out['gn_mixed_precision'] = {'gn_ms': e0.elapsed_time(e1),
                                     'max_diff_vs_f64': float(torch.nan_to_num(diff, nan=0.0).max().item()),
                                     'note': 'DEXCT_GN_PRECISION=mixed: first n-4 iterations float32, last 4 float64; '
                                             'opt-in, not the reference arithmetic, not used for value'}
        del a_mixed, diff

    # ---- single-row (the reference's own 2-D case), ray-parallel kernel
    if not args.skip_single_row:
        ct1 = dx.FanBeamGeometry(N_channels=args.channels, N_proj=args.views, gamma_fan=0.8230337, SID=60.0,
                                 SDD=100.0, eid=True, detector_file=det, N_rows=1)
        ph1 = synthetic.make_phantom(n, 1, extent=51.2, seed=1234)
        pj1 = fp.Projector(ct1, ph1, kernel=1)
        c1 = torch.empty((2, args.views, 1, args.channels), dtype=torch.float32, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pj1.project_tables(mu_d, w_d, out=c1)
        e0.record()
        for _ in range(10):
            pj1.project_tables(mu_d, w_d, out=c1)
        e1.record()
        torch.cuda.synchronize()
        ms1 = e0.elapsed_time(e1) / 10
        out['single_row'] = {'rays': args.views * args.channels, 'siddon_ms': ms1,
                             'integrals_per_s': args.views * args.channels * sum(n_e_spec) / (ms1 * 1e-3)}

    # ---- cone beam (true 3-D rays, untuned one-thread-per-ray kernel) on a slice of the same scan
    if not args.skip_single_row and rows >= 8:
        cv = max(1, min(args.views, 100))
        ctc = dx.FanBeamGeometry(N_channels=args.channels, N_proj=cv, gamma_fan=0.8230337, SID=60.0, SDD=100.0,
                                 eid=True, detector_file=det, N_rows=rows, cone=True, h_iso=ph.dz)
        pjc = fp.Projector(ctc, ph)
        cc = torch.empty((2, cv, rows, args.channels), dtype=torch.float32, device=dev)
        pjc.project_tables(mu_d, w_d, out=cc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pjc.project_tables(mu_d, w_d, out=cc)
        e1.record()
        torch.cuda.synchronize()
        msc = e0.elapsed_time(e1)
        out['cone_beam'] = {'rays': cv * rows * args.channels, 'siddon_ms': msc,
                            'integrals_per_s': cv * rows * args.channels * sum(n_e_spec) / (msc * 1e-3)}
        del pjc, cc

    # ---- CPU baseline: the oracle (float64 textbook Siddon + detection, then float64 Newton) on a bounded
    # sample of the same workload, all host cores
    if world > 1:
        out['cpu_baseline'] = None          # timed at N = 1 only (rank 0 shares its host cores with the other ranks here)
    elif not args.no_cpu_baseline:
        threads = co.max_threads()
        sample_rows, sample_views = 8, 2
        gs = co.make_geom(ct.N_proj, ct.N_channels, sample_rows, n // 2 - sample_rows // 2, n, n, n, ph.dx, ph.dy,
                          ph.dz, ct.SID, ct.SDD)
        mu64, w64 = mu_d.double().cpu().numpy(), w_d.double().cpu().numpy()
        t0 = time.perf_counter()
        cs = co.project_classic(gs, ct.view_cs(), ct.chan_cs(), 0, sample_views, ph.volume, mu64, w64,
                                n_threads=threads)
        dt = time.perf_counter() - t0
        # scale the sample to about cpu_seconds of work
        # the Newton leg of the same rays costs about 6x the projection leg: aim the pair at cpu_seconds
        sample_views = int(max(2, min(args.views, sample_views * args.cpu_seconds / 7.0 / max(dt, 1e-3))))
        t0 = time.perf_counter()
        cs = co.project_classic(gs, ct.view_cs(), ct.chan_cs(), 0, sample_views, ph.volume, mu64, w64,
                                n_threads=threads)
        t_proj = time.perf_counter() - t0
        n_sample = sample_views * sample_rows * args.channels
        # the Newton leg runs on the GPU's own (float32) sinogram values of those rays, so that its result is at
        # the same time the parity reference for the GPU decomposition at benchmark scale
        r0 = n // 2 - sample_rows // 2
        have_gpu = world == 1 and rows == n and sample_views <= nV
        g_cnt = counts[:, :sample_views, r0:r0 + sample_rows, :].double().cpu().numpy() if have_gpu else cs
        t0 = time.perf_counter()
        a_cpu = co.gn_decompose(g_cnt[0].ravel(), g_cnt[1].ravel(), i0, mus, args.iters, n_threads=threads)
        t_gn_cpu = time.perf_counter() - t0
        if have_gpu:
            a_gpu = a_out[:sample_views, r0:r0 + sample_rows].cpu().numpy().reshape(-1, 2)
            live = (a_gpu != 0).any(axis=1) & np.isfinite(a_cpu).all(axis=1)        # masked air pixels are exactly 0
            out['parity_sample'] = {
                'rays': n_sample,
                'sinogram_max_rel_err_vs_float64_siddon': float(np.max(np.abs(g_cnt - cs) / cs)),
                'decomposition_max_err_vs_float64_newton': float(np.max(
                    np.abs(a_gpu[live] - a_cpu[live]) / np.maximum(np.abs(a_cpu[live]), 1.0))),
                'decomposed_pixels_compared': int(live.sum()),
                'note': 'oracle (CPU) results of the cpu_baseline sample against the GPU results of the same rays of '
                        'the timed step; tolerances of the north star: 1e-5'}
        out['cpu_baseline'] = {'value': n_sample * sum(n_e_spec) / (t_proj + t_gn_cpu), 'unit': 'ray-energy integrals/s',
                               'cores': threads, 'kind': 'port',
                               'sample': f'{sample_views} views x {sample_rows} rows x {args.channels} channels of the same '
                                         f'scan (oracle: float64 Siddon 1985 + detection {t_proj:.1f} s, float64 Newton '
                                         f'{t_gn_cpu:.1f} s, OpenMP over rays / pixels)',
                               'siddon_only_integrals_per_s': n_sample * sum(n_e_spec) / t_proj,
                               'gn_pixel_solves_per_s': n_sample / t_gn_cpu}
        # also the NumPy restatement of optimize_sino_cpu (the reference's own style of CPU code), on a
        # few views of the same sinograms, with the thread counts that apply to it
        from oracle import gn_oracle
        np_views = min(16, g_cnt.shape[1])
        g_np = g_cnt[:, :np_views, 0, :] if g_cnt.ndim == 4 else g_cnt.reshape(2, -1, args.channels)[:, :np_views]
        t0 = time.perf_counter()
        gn_oracle.newton_solve(g_np, i0, mus, args.iters)
        t_np = time.perf_counter() - t0
        blas = None
        try:
            from threadpoolctl import threadpool_info
            blas = [{'api': t.get('user_api'), 'threads': t.get('num_threads')} for t in threadpool_info()]
        except Exception:
            pass
        out['cpu_baseline']['numpy_restatement'] = {
            'pixel_iters_per_s': g_np.shape[1] * g_np.shape[2] * args.iters / t_np,
            'pixel_solves_per_s': g_np.shape[1] * g_np.shape[2] / t_np,
            'sample': f'{g_np.shape[1]} views x {g_np.shape[2]} channels x {args.iters} iterations, {t_np:.1f} s',
            'os_cpu_count': os.cpu_count(), 'omp_threads_c_oracle': threads, 'blas_threadpools': blas}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
