"""`cpu_baseline` and the parity sample: the ONLY part of bench.py that touches oracle/ (the checker and the reported CPU
baseline; never the thing measured as `value`).  Also the exact segment / slab counts of a scan that the rooflines need - integer
geometry from the oracle's plan, computed after the timed loop."""
import os
import time

import numpy as np

from oracle import c_oracle as co


def segment_count(wl):
    """Exact number of Siddon segments per (view, channel) of this rank's shard, from the CPU oracle's plan."""
    ct, ph, n = wl.ct, wl.ph, wl.n
    geom = co.make_geom(ct.N_proj, ct.N_channels, wl.rows, ph.z_index, n, n, n, ph.dx, ph.dy, ph.dz, ct.SID, ct.SDD)
    plan = co.plan(geom, ct.view_cs(), ct.chan_cs(), wl.vb, wl.ve)
    return int(co.count_segments(geom, plan))


def plan_slabs(ct, ph, nz):
    """in-plane slabs of every (view, channel) pair of a scan, summed"""
    n = ph.Nx
    geom = co.make_geom(ct.N_proj, ct.N_channels, 1, 0, n, n, nz, ph.dx, ph.dy, ph.dz, ct.SID, ct.SDD)
    plan = co.plan(geom, ct.view_cs(), ct.chan_cs(), 0, ct.N_proj)
    return float(plan['n_slabs'].sum())


def cpu_baseline(wl, args, out):
    """The oracle (float64 textbook Siddon + detection, then float64 Newton) on a bounded sample of the same workload, all host
    cores; its Newton leg runs on the GPU's own sinogram values of those rays, so it is at the same time the parity reference of
    the timed step at benchmark scale (`parity_sample`)."""
    ct, ph, n, rows = wl.ct, wl.ph, wl.n, wl.rows
    n_e_spec, i0, mus = wl.n_e_spec, wl.i0, wl.mus
    threads = co.max_threads()
    sample_rows, sample_views = 8, 2
    gs = co.make_geom(ct.N_proj, ct.N_channels, sample_rows, n // 2 - sample_rows // 2, n, n, n, ph.dx, ph.dy,
                      ph.dz, ct.SID, ct.SDD)
    mu64, w64 = wl.mu_d.double().cpu().numpy(), wl.w_d.double().cpu().numpy()
    t0 = time.perf_counter()
    cs = co.project_classic(gs, ct.view_cs(), ct.chan_cs(), 0, sample_views, ph.volume, mu64, w64, n_threads=threads)
    dt = time.perf_counter() - t0
    # scale the sample to about cpu_seconds of work: the Newton leg of the same rays costs about 6x the projection leg
    sample_views = int(max(2, min(args.views, sample_views * args.cpu_seconds / 7.0 / max(dt, 1e-3))))
    t0 = time.perf_counter()
    cs = co.project_classic(gs, ct.view_cs(), ct.chan_cs(), 0, sample_views, ph.volume, mu64, w64, n_threads=threads)
    t_proj = time.perf_counter() - t0
    n_sample = sample_views * sample_rows * args.channels
    r0 = n // 2 - sample_rows // 2
    have_gpu = wl.world == 1 and rows == n and sample_views <= wl.nV
    g_cnt = wl.counts[:, :sample_views, r0:r0 + sample_rows, :].double().cpu().numpy() if have_gpu else cs
    t0 = time.perf_counter()
    a_cpu = co.gn_decompose(g_cnt[0].ravel(), g_cnt[1].ravel(), i0, mus, args.iters, n_threads=threads)
    t_gn_cpu = time.perf_counter() - t0
    if have_gpu:
        a_gpu = wl.a_out[:sample_views, r0:r0 + sample_rows].cpu().numpy().reshape(-1, 2)
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
