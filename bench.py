#!/usr/bin/env python3
"""Benchmark of the dex-ct hot path on MI355X.

A step = one pass of the hot path over BASELINE.json's metric configuration (configs[2]):
512^3 synthetic water/bone phantom, 1000 views x 800 channels, dual 80/140 kVp spectra, i.e.
  plan -> Siddon traversal + polychromatic detection of BOTH spectra (one fused traversal; both outputs of get_sino,
          sino_raw and sino_log, in the reference's order)
       -> Newton decomposition (n_iters = 50 asked, as main.py:153) + air mask
       -> (N > 1) assembly of the two raw sinograms over RCCL, overlapped with the decomposition.
Detector rows: BASELINE.json does not name a row count and a single row touches one slice of the
512^3 volume, so the workload is the stacked fan N_rows = Nz = 512; the
single-row case is reported under "single_row".  Inputs are resident in HBM before the timed region.

    python bench.py [--gpus N --steps K --warmup W]            (N > 1: starts the N rank processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: projection angles are sharded contiguously over the ranks (one process per GPU, RCCL).
  --scaling strong (default): the scan is FIXED - the metric's 1000 x 800 scan (or --workload config3: BASELINE
      configs[3], 2000 views x 1024 channels) - and rank r projects and decomposes views/N of it; one assembly
      of the raw sinograms over xGMI (--gather auto: the fastest of three measured modes), overlapped with the Newton
      kernel, plus one scalar all-reduce(max).
  --scaling weak: every rank projects `--views` angles of an N x views scan of the same phantom.
At N = 1 both are the same workload (BASELINE configs[2]).

This file parses the arguments and puts the JSON line together; what it runs lives in bench/ (bench/step.py: the timed
region and nothing else; launch, multi, modes, roofline, dropin, cpu: see bench/__init__.py).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench import launch  # noqa: E402


def main():
    args = launch.parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch.launch_ranks(args, __file__))
    world, rank, local_rank, backend = launch.init_ranks(args)
    import torch
    import torch.distributed as dist
    from dex_ct_sim_amd import matdecomp as md
    from bench import modes, multi, roofline
    from bench.step import Workload, mean

    wl = Workload(args, world, rank, local_rank)
    n, rows, total_views, n_rays, n_e_spec, precision = wl.n, wl.rows, wl.total_views, wl.n_rays, wl.n_e_spec, wl.precision

    # the table of the Newton short cut (once per pair of spectra, cached by content; outside the timed region like every
    # other table): timed here so that the bench line says what it costs
    t0 = time.perf_counter()
    if precision == 'f64':
        md._device_tables(wl.i0, wl.mus, wl.dev, True)
        torch.cuda.synchronize()
    gate_prep_s = time.perf_counter() - t0

    chosen, auto = (multi.choose_gather(wl, args) if world > 1 else (None, None))
    # ---- THE TIMED REGION (bench/step.py): W warm-up steps, exactly K steps between barrier + synchronize, max over ranks
    elapsed, t_sid, t_gn, t_exposed = wl.timed_steps(args.steps, args.warmup)
    ms_per_step = 1e3 * elapsed / args.steps
    rays_all = total_views * rows * args.channels           # rays of all ranks together (ragged shards included)
    if args.shard_of > 1:
        rays_all = n_rays                                   # only this shard is computed here
    integrals_per_step = rays_all * sum(n_e_spec)
    value = integrals_per_step / (elapsed / args.steps)
    sid_ms, gn_ms = mean(t_sid), mean(t_gn)
    gstats = md.last_gn_stats()             # of the last timed step's launches (the default mode)

    multi_gpu = None
    if world > 1:
        multi_gpu = multi.measure(wl, args, backend, chosen, auto, ms_per_step, t_sid, t_gn, t_exposed, integrals_per_step)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    masked = float((wl.counts_nat[0] >= 0.95 * wl.gmax).float().mean().item())
    live = max((1.0 - masked) * n_rays, 1.0)
    steps_pp = (gstats['pixel_iterations'] / live) if (gstats and gstats.get('pixel_iterations')) else None
    short = bool(gstats) and gstats.get('mode') in md.SHORTCUT_MODES
    out = {
        'metric': 'Siddon ray-energy integrals/sec', 'value': value, 'unit': 'ray-energy integrals/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
        'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
        'dtype': 'f64' if precision == 'f64' else 'f32+f64', 'data': 'synthetic',
        'config': {'workload': f'{n}^3 water/bone phantom, {total_views} views x {args.channels} channels x '
                               f'{rows} rows (stacked fan), dual 140/80 kVp Kramers spectra ({n_e_spec[0]}+{n_e_spec[1]} '
                               f'energy bins), fused dual-spectrum Siddon + Newton decomposition: n_iters = {args.iters} asked, '
                               + (f'{steps_pp:.2f} full-table steps executed per unmasked pixel' if steps_pp is not None else
                                  'executed steps not counted in this mode')
                               + (' (short cut: start values from the tabulated fixed points of the reference\'s walk; every pixel '
                                  'compared with the exact count, see gn_exact)' if short else ''),
                   'baseline_config': 'configs[2]' if (args.workload == 'config2' and total_views == 1000) else
                                      ('configs[3]' if args.workload == 'config3' else 'configs[2] x N views (weak scaling)'),
                   'n': n, 'rays_per_gpu': n_rays, 'rays_total': rays_all,
                   'shard': None if args.shard_of <= 1 else f'views [{wl.vb}, {wl.ve}) = rank {args.shard_rank} of {args.shard_of} '
                                                            f'(one rank\'s share, measured alone on one GPU)',
                   'parallelism': f'{total_views} views sharded x{world} ({args.scaling} scaling: '
                                  + ('fixed scan, views/N per rank)' if args.scaling == 'strong' else
                                     f'{args.views} views per rank)'),
                   'arithmetic': 'voxel indices int64 fixed point, path lengths + detection f32, Newton '
                                 + ('f64 (reference arithmetic)' if precision == 'f64' else 'f32 bulk + f64 polish')},
        'kernel_ms': {'siddon_project': sid_ms, 'gn_decompose': gn_ms},
        'siddon_only_integrals_per_s': n_rays * sum(n_e_spec) / (sid_ms * 1e-3),
        'siddon_rays_per_s': n_rays / (sid_ms * 1e-3),
        'gn_pixel_solves_per_s': n_rays / (gn_ms * 1e-3),
    }
    if multi_gpu is not None:
        out['multi_gpu'] = multi_gpu

    # ---- rooflines (bench/roofline.py).  The step's dominant kernel is the Newton launch (FP64 vector issue); the traversal
    # kernel is stated against the bound its counters show.  Integer geometry (segment counts) from the oracle's plan.
    from bench import cpu
    kname = roofline.siddon_kernel_name(wl, args)
    prof, traffic_src = roofline.matching_profile(n_rays, kname, n)
    out['roofline_siddon'], sid_info = roofline.siddon(wl, args, sid_ms, cpu.segment_count(wl), prof, traffic_src, wl.gmax)
    out['gn_masked_fraction'] = masked      # air pixels (zeroed by the mask afterwards, matdecomp.py:204-205) are not iterated
    roof, gn_info = roofline.newton(wl, args, gstats, gn_ms, masked, prof, traffic_src, gate_prep_s, md)
    out['roofline'] = roof
    if precision == 'f64' and world == 1 and not args.skip_gn_full_loop:
        modes.newton_modes(wl, args, out, gstats, gn_info['two_level'], masked, integrals_per_step, gn_info['flops_per_pixel_iter'])
    if 'frac' not in roof:
        roof.update({'achieved': None, 'frac': None, 'note': 'executed-iteration count not available in this mode'})
    if world == 1:
        if precision == 'f64':
            modes.mixed_precision(wl, args, out)
        if not args.skip_noisy:
            modes.noisy_step(wl, args, out, sid_ms, ms_per_step, sid_info)
        if not args.skip_quadrature:
            modes.reduced_quadrature(wl, args, out, sid_ms)
        if not args.skip_single_row:
            modes.single_row(wl, args, out, prof, traffic_src, sid_info, cpu.plan_slabs)
            if rows >= 8:
                modes.cone_beam(wl, args, out, prof, traffic_src, sid_info, cpu.plan_slabs)
        if not args.skip_dropin:
            import dex_ct_sim_amd as dx
            from dex_ct_sim_amd import forward_project as fp
            from bench.dropin import dropin_e2e
            out['dropin_e2e'] = dropin_e2e(args, dx, fp, md, wl.ct, wl.ph, wl.specs, wl.det, wl.dev)
    # ---- CPU baseline (bench/cpu.py): timed at N = 1 only (rank 0 shares its host cores with the other ranks otherwise)
    if world > 1:
        out['cpu_baseline'] = None
    elif not args.no_cpu_baseline:
        cpu.cpu_baseline(wl, args, out)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
