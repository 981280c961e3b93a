"""The `roofline_siddon` and `roofline` objects of the JSON line (after the timed loop; nothing here is timed).

Counters cannot be read inside the timed run: traffic and instruction counts are those of the newest matching
profiles/*_pmc_traffic.json (rocprofv3 --pmc passes of the same command, tools/profile_gpu.sh), labelled `traffic_source`."""
import glob
import json
import os

from . import CLOCK_GHZ, FP64_VALU_PEAK_TFLOPS, HBM_PEAK_GBS, ISSUE_PEAK_G, SIMDS
from .launch import ROOT


def matching_profile(n_rays, kname, n):
    """(counters, source) of the newest profiles/*_pmc_traffic.json taken on this workload with this traversal kernel"""
    prof, src = {}, None
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json'))):     # r01a < ... < r06a: the last match wins
        j = json.load(open(f))
        if j.get('rays_per_gpu') == n_rays and kname in j.get('siddon_kernel', '') and j.get('n', 512) == n:
            prof, src = j, 'profiles/' + os.path.basename(f)        # same workload and kernel only
    return prof, src


def siddon_kernel_name(wl, args):
    if getattr(wl.pj, 'use_packed', False):
        return 'rows16_kernel'
    return {1: 'rays_kernel', 2: 'rows_kernel', 3: 'rows4_kernel', 5: 'rows4t_kernel', 6: 'wave_ray_kernel'}[args.kernel or (3 if wl.native == 1 else 1)]


def detection_floor_slots(lanes4, n_e_any, n_e_spec, noisy=False):
    """vector-issue slots of the detection per group of 4 rays: per energy bin any spectrum weights 6 v_pk_fma (3 materials x 2
    ray pairs) + 4 v_exp_f32 (2 issue slots each), + 2 v_pk_fma per spectrum that weights the bin (noisy: twice that - the
    variance rides in the same loop)"""
    return lanes4 * (14.0 * n_e_any + (4.0 if noisy else 2.0) * sum(n_e_spec))


def siddon(wl, args, sid_ms, seg_vc, prof, traffic_src, gmax):
    """The traversal kernel against the bound its counters show: vector issue while the volume is cache
    resident (<= 256 MiB Infinity Cache), HBM beyond.  No fraction here can exceed 1."""
    n, rows, n_rays, native, pj = wl.n, wl.rows, wl.n_rays, wl.native, wl.pj
    kname = siddon_kernel_name(wl, args)
    # algorithmic bytes: S_ray x bytes per stored voxel + outputs; the packed volume stores a voxel in 2 bits
    b_vox = 0.25 if getattr(pj, 'use_packed', False) else 1.0
    # outputs of the timed launch: sino_raw of both spectra; sino_log too where the kernel writes it itself (row-parallel kernels
    # leave it to the pass that brings both outputs into the reference's order, dexct_transpose_log)
    alg_bytes = seg_vc * rows * b_vox + (2 if native == 1 else 4) * 4 * n_rays
    alg_gbps = alg_bytes / (sid_ms * 1e-3) / 1e9
    traffic = prof.get('siddon_hbm_bytes_per_launch')
    vol_bytes = int(n * n * n * b_vox)
    cache_resident = vol_bytes <= 256 * 2 ** 20
    # vector-issue floor of the packed traversal + detection: per voxel dword visited
    #   rows4_kernel (1 B / voxel, 4 rows per dword): 2 vector instructions (bit-plane AND + its add; the weighted-sum
    #     add shared by two visits through v_add3)
    #   rows16_kernel (2 bits / voxel, 16 rows per dword): 3 (the lane's address add + 21 / 8 for the seven carry-save
    #     adders per 8 words; the ripple into the high counter bits can be amortised away)
    # and the detection (detection_floor_slots); one slot = 4 cycles of one of the 1024 SIMDs.
    n_e_any = int(((wl.w_d != 0).any(dim=0)).sum().item())
    # rays that crossed air only (they are the pixels the decomposition masks) are detected once per (view, channel)
    # pair, not per row: they are left out of the floor (their one detection per pair is not counted either)
    air_rays = float((wl.counts_nat[0] >= 0.95 * gmax).float().mean().item())
    lanes = n_rays * (1.0 - air_rays) / 4.0
    rows_per_dword, per_visit = (16.0, 3.0) if kname == 'rows16_kernel' else (4.0, 2.0)
    traversal_slots = per_visit * seg_vc * rows / rows_per_dword
    floor_slots = traversal_slots + detection_floor_slots(lanes, n_e_any, wl.n_e_spec)
    slots_per_s = SIMDS * CLOCK_GHZ * 1e9 / 4.0                 # wave-instruction issue slots per second, whole chip
    floor_ms = floor_slots / 64.0 / slots_per_s * 1e3
    sid = {'kernel': kname, 'avg_launch_ms': sid_ms,
           'algorithmic_bytes_per_launch': alg_bytes, 'segments_per_launch': seg_vc * rows,
           'algorithmic_GBps': alg_gbps, 'traffic': traffic, 'traffic_source': traffic_src,
           'traffic_GBps': None if traffic is None else traffic / (sid_ms * 1e-3) / 1e9,
           'traffic_frac_of_hbm_peak': None if traffic is None else traffic / (sid_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           'volume_cache_resident': cache_resident,
           'valu_floor': {'floor_wave_instructions': floor_slots / 64.0, 'floor_ms_at_%.1f_GHz' % CLOCK_GHZ: floor_ms,
                          'achieved_over_floor': floor_ms / sid_ms,
                          'measured_valu_instructions': prof.get('siddon_valu_insts'),
                          'measured_valu_busy': prof.get('siddon_valu_busy'), 'counters_source': traffic_src}}
    if cache_resident:
        sid.update({'bound': 'valu_issue', 'achieved': floor_slots / 64.0 / (sid_ms * 1e-3) / 1e9, 'peak': slots_per_s / 1e9,
                    'unit': 'G wave-instructions/s', 'frac': floor_ms / sid_ms,
                    'note': 'the %d MiB volume is L2 / Infinity-Cache resident: the algorithmic byte rate (%.0f GB/s) is a cache-served '
                            'request rate, not an HBM rate, and is reported as algorithmic_GBps only; the counters show vector issue as '
                            'the binding resource, so frac = instruction floor / time' % (vol_bytes >> 20, alg_gbps)})
    else:
        hbm_gbps = sid['traffic_GBps']
        sid.update({'bound': 'hbm', 'achieved': hbm_gbps if hbm_gbps is not None else min(alg_gbps, HBM_PEAK_GBS),
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': (hbm_gbps if hbm_gbps is not None else min(alg_gbps, HBM_PEAK_GBS)) / HBM_PEAK_GBS,
                    'note': 'volume larger than the Infinity Cache: achieved = measured fabric traffic (PMC) when a '
                            'matching profile exists, else the algorithmic byte rate capped at the peak; '
                            'traffic / algorithmic bytes = %s' % (None if traffic is None else round(traffic / alg_bytes, 3))})
    info = {'kname': kname, 'n_e_any': n_e_any, 'lanes4': lanes, 'traversal_slots': traversal_slots, 'slots_per_s': slots_per_s,
            'air_rays': air_rays}
    return sid, info


def newton(wl, args, gstats, gn_ms, masked, prof, traffic_src, gate_prep_s, md):
    """The step's dominant kernel, the Newton launch (FP64 vector issue).

    `frac` (round 6) = the FP64 flops the launch ISSUES / its time / 78.6 TFLOP/s - what the hardware does, not what the reference's
    formulation would have done: the short cut's one step is of the Gauss-Newton form (6 of the 12 accumulations per energy)
    and an energy only one spectrum weights takes half of them, so counting SURVEY 8d's 29 flops per energy for it made the
    fraction RISE when work was removed.  That figure stays as `frac_by_survey_unit`.  `issue`: the instruction-issue view
    (wave-instructions per second of the chip's 614.4 G, from the PMC file) - the ceiling this kernel actually sits at."""
    i0, n_rays, precision = wl.i0, wl.n_rays, wl.precision
    # 28 flops + 1 exp per energy-iteration
    flops_per_pixel_iter = i0.shape[1] * (28 + 1)
    # what the hardware issues of those: an energy only one spectrum weights takes 6 accumulations instead of 12 (17 of the
    # 29 flop), an energy no spectrum weights is dropped
    n_both = int(((i0[0] != 0) & (i0[1] != 0)).sum())
    n_one = int(((i0[0] != 0) ^ (i0[1] != 0)).sum())
    hw_share = (29.0 * n_both + 17.0 * n_one) / (29.0 * i0.shape[1])
    two_level = bool(gstats) and gstats.get('mode') in md.SHORTCUT_MODES
    gn_name = ('gn_shortcut_kernel (start values from the table of the reference\'s fixed points + full-table steps)' if two_level
               else 'gn_refill_kernel<false>') if precision == 'f64' else 'gn_kernel<true,false>'
    main_ms = gstats['main_ms'] if gstats else gn_ms      # HIP events around the launch, on its stream, last timed step
    gn_form = two_level and gstats.get('mode') == 'one'    # the one step of the short cut is the CHORD step: 2 of the 12 sums (nu alone)
    if gn_form:
        # per energy: exponent 2 FMA = 4 flops, the exponential counted as 1, one accumulation (2 flops) per weighting spectrum
        hw_share = (9.0 * n_both + 7.0 * n_one) / (29.0 * i0.shape[1])
    roof = {'kernel': gn_name, 'bound': 'valu_fp64' if precision == 'f64' else 'valu_fp32+fp64',
            'unit': 'TFLOP/s', 'peak': FP64_VALU_PEAK_TFLOPS, 'avg_launch_ms': main_ms, 'gn_ms_all_launches': gn_ms,
            'traffic': ((prof.get('gn_fetch_bytes_raw', 0) if prof.get('gn_fetch_counted_in_full') else prof.get('gn_fetch_bytes_x2_corrected', 0))
                        + prof.get('gn_write_bytes', 0)) or None,
            'traffic_source': traffic_src, 'algorithmic_bytes_per_launch': 24 * n_rays,
            'traffic_note': 'HBM-side bytes (FETCH_SIZE + WRITE_SIZE) of this kernel from the rocprofv3 --pmc passes of the same command '
                            'recorded in traffic_source (counters cannot be read from inside the timed run); algorithmic: 8 B of counts '
                            'in and 16 B of results out per pixel',
            'bound_note': 'neither HBM (24 - 41 B/pixel against >= 2e3 flops/pixel) nor MFMA (no dense contraction; FP64 MFMA and '
                          'FP64 VALU do not overlap on gfx950, DESIGN.md 4.4): bound = FP64 vector issue'}
    info = {'flops_per_pixel_iter': flops_per_pixel_iter, 'two_level': two_level, 'main_ms': main_ms}
    if gstats and gstats.get('pixel_iterations'):
        # EXECUTED work of the timed launch itself: the kernel counts the pixel-iterations it ran
        ex_flops = gstats['pixel_iterations'] * flops_per_pixel_iter
        live = max((1.0 - masked) * n_rays, 1.0)
        by_unit = ex_flops / (main_ms * 1e-3) / 1e12
        issued = hw_share * by_unit
        roof.update({'achieved': issued, 'frac': issued / FP64_VALU_PEAK_TFLOPS,
                     'achieved_by_survey_unit': by_unit, 'frac_by_survey_unit': by_unit / FP64_VALU_PEAK_TFLOPS,
                     'executed_pixel_iterations': gstats['pixel_iterations'],
                     'mean_iterations_per_unmasked_pixel': gstats['pixel_iterations'] / live,
                     'exit_saving': 1.0 - gstats['pixel_iterations'] / max((1.0 - masked) * n_rays * args.iters, 1.0),
                     'stalled_lane_steps': gstats.get('stalled_lane_steps'),
                     'hardware_fp64_flop_share': hw_share,
                     'hardware_fp64_utilisation': issued / FP64_VALU_PEAK_TFLOPS,
                     'note': 'achieved / frac = the FP64 flops the timed launch ISSUED (iterations counted by the kernel x the flops its '
                             'energy loop issues per iteration) / its time / the FP64 vector peak.  Of the %d energies of the union grid '
                             '%d carry both spectra, %d only one (half the accumulations) and %d none (dropped); SURVEY 8d\'s unit - one '
                             'Newton iteration of one pixel at the reference\'s 28 flops + 1 exp per energy - counts %.2f x as many flops as '
                             'are issued (hardware_fp64_flop_share = %.2f): frac_by_survey_unit keeps that figure, which rises when work is '
                             'removed and is therefore not the roofline.  The rest of the busy vector pipe is the exponential (14 of the '
                             '~16 instructions per energy), integer / move work and the per-pixel gate and interpolations: see '
                             '`issue`.%s  exit_saving = share of the n_iters x pixels full-table iterations the short cut and the exits '
                             'made unnecessary - reported separately, not as throughput'
                             % (i0.shape[1], n_both, n_one, i0.shape[1] - n_both - n_one, 1.0 / hw_share, hw_share,
                                '  The ONE step per pixel of the default short cut is the chord step (round 6): the relative misfit of the counts '
                                '- the full energy sum of nu, 2 of the 12 accumulations per energy - times the TABULATED inverse log-Jacobian '
                                '(what it leaves is bounded by the table\'s (kappa, eps) per cell).' if gn_form else '')})
        valu = prof.get('gn_valu_insts')
        if valu:
            rate = valu / (main_ms * 1e-3) / 1e9
            roof['issue'] = {'bound': 'valu_issue', 'unit': 'G wave-instructions/s', 'peak': ISSUE_PEAK_G, 'achieved': rate,
                             'frac': rate / ISSUE_PEAK_G, 'wave_instructions_per_launch': valu,
                             'valu_busy': prof.get('gn_valu_busy'), 'wait_any_share': prof.get('gn_wait_any_share'),
                             'counters_source': traffic_src,
                             'note': 'SQ_INSTS_VALU of the launch (PMC pass of the same command) / this run\'s launch time / the chip\'s '
                                     'issue peak (1024 SIMDs x 2.4 GHz / 4 cycles): the kernel sits at the issue ceiling - only fewer '
                                     'instructions make it faster'}
        if gn_form:
            roof['one_step_form'] = 'chord step: misfit of the counts (2 of 12 sums per energy) x tabulated inverse log-Jacobian; |e1| <= eps e0 + kappa e0^2, both tabulated per cell'
        if two_level:
            roof['short_cut'] = {
                'mode': gstats['mode'], 'launch_ms': main_ms, 'full_energies': int(i0.shape[1]),
                'table_preparation_s_once_per_pair_of_spectra': gate_prep_s,
                'full_steps_per_unmasked_pixel': gstats['pixel_iterations'] / live,
                'note': 'what the reference returns is the fixed point its walk from 1e-6 ends at - a function of the two counts, '
                        'tabulated once per pair of spectra by running the single launch on a 257 x 257 grid of counts.  A pixel '
                        'in a cell where that walk ends by the tolerance rule within n_iters steps, smoothly, starts from the '
                        '6 x 6 Lagrange interpolant of the tabulated fixed points (1e-10 of |a| from its own) and takes ONE '
                        'full-table step where the cell\'s tabulated kappa - an analytic bound on Newton\'s quadratic constant '
                        'from the Hessian and third derivatives of the likelihood at the tabulated fixed points - times the '
                        'squared step puts what is left below stop_tol / 4; else two, the second being the tolerance rule\'s '
                        'evidence of convergence of the FULL model (mode start: always two: value_two_step); accepted only on '
                        'the reference\'s branch; every other pixel is solved from 1e-6 with all n_iters steps in the same '
                        'launch.  Compared with the exact count on every pixel below (gn_exact)'}
    return roof, info
