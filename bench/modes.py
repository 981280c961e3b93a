"""What the JSON line reports BESIDE `value` (N = 1; each block runs after the timed loop and restores the default state):
the Newton modes against the reference's exact count on every pixel, the step WITH quantum noise, the opt-in mixed precision
and reduced quadrature, the single-row and cone-beam geometries."""
import time

import torch

import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, matdecomp as md, synthetic

from . import CLOCK_GHZ, FP64_VALU_PEAK_TFLOPS
from .step import mean


def _max_rel(a, b):
    return float(torch.nan_to_num((a - b).abs() / b.abs().clamp(min=1.0), nan=0.0).max().item())


def _timed_mode(wl, args):
    """the step in the mode the knobs of `wl` say: (seconds of args.steps steps, mean Newton ms, stats, results)"""
    wl.step(False)
    torch.cuda.synchronize()
    t_gn = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step(True)
        torch.cuda.synchronize()
        t_gn.append(wl.ev[2].elapsed_time(wl.ev[3]))
    return time.perf_counter() - t0, mean(t_gn), md.last_gn_stats(), wl.a_out.clone()


def newton_modes(wl, args, out, gstats, two_level, masked, integrals_per_step, flops_per_pixel_iter):
    """The reference's fixed iteration count, EXACTLY (stop_tol = 0): the same step timed the same way -> value_exact; checked bit
    for bit against a launch that executes every iteration, and the results of the default step - and of the two modes in
    between - checked against it on every pixel (the run aborts above 1e-12)."""
    n_rays = wl.n_rays
    live = max((1.0 - masked) * n_rays, 1.0)
    gn_flops_all = (1.0 - masked) * n_rays * args.iters * flops_per_pixel_iter
    a_default = wl.a_out.clone()
    a_two = a_single = None
    if two_level and gstats.get('mode') == 'one':
        # ---- the short cut with two steps and the tolerance rule for every pixel (round 4's form, 'start')
        wl.gn_mode = 'start'
        elapsed_2, g2_ms, st2, a_two = _timed_mode(wl, args)
        wl.gn_mode = None
    if two_level:
        # ---- the default tolerance stop in ONE launch from the reference's start value (round 4's first form of the default)
        wl.gn_mode = False
        elapsed_1, g1_ms, st1, a_single = _timed_mode(wl, args)
        wl.gn_mode = None
    wl.gn_tol = 0.0
    elapsed_ex, gn_ex_ms, ex_stats, a_exact = _timed_mode(wl, args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    md.gn_device(wl.counts_nat[0], wl.counts_nat[1], wl.i0_d, wl.mus_d, args.iters, 'f64', out=wl.a_out, out_rc=wl.out_rc, mask_max=wl.gmax,
                 mask_frac=0.95, full_loop=True)             # DEXCT_GN_FLAG_FULL_LOOP: every iteration executed
    e1.record()
    torch.cuda.synchronize()
    full_ms = e0.elapsed_time(e1)
    exact_is_full = bool(torch.equal(wl.a_out.view(torch.int64), a_exact.view(torch.int64)))
    diff = _max_rel(a_default, a_exact)
    same_nan = bool(torch.equal(torch.isnan(a_default), torch.isnan(a_exact)))
    if not exact_is_full:
        raise SystemExit('bench.py: the exact launch (stop_tol = 0) differs from the full 50-iteration loop')
    if not (diff <= 1e-12 and same_nan):
        raise SystemExit(f'bench.py: the default mode moved a pixel by {diff:.3e} (> 1e-12) from the exact launch')
    if a_two is not None:
        diff2 = _max_rel(a_two, a_exact)
        if not (diff2 <= 1e-12 and bool(torch.equal(torch.isnan(a_two), torch.isnan(a_exact)))):
            raise SystemExit(f'bench.py: the two-step short cut moved a pixel by {diff2:.3e} (> 1e-12) from the exact launch')
        out['value_two_step'] = integrals_per_step / (elapsed_2 / args.steps)
        out['gn_two_step'] = {
            'gn_ms': g2_ms, 'ms_per_step': 1e3 * elapsed_2 / args.steps, 'executed_pixel_iterations': st2['pixel_iterations'],
            'mean_iterations_per_unmasked_pixel': st2['pixel_iterations'] / live,
            'achieved': st2['pixel_iterations'] * flops_per_pixel_iter / (g2_ms * 1e-3) / 1e12,
            'frac': st2['pixel_iterations'] * flops_per_pixel_iter / (g2_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
            'max_diff_vs_exact': diff2,
            'note': "two_level='start' / DEXCT_GN_TWO_LEVEL=start: the short cut with two full-table steps and the tolerance rule for "
                    'every pixel (the default of round 4, on round 5\'s kernel): the same launch doing twice the counted work - its '
                    'frac (by SURVEY 8d\'s unit: these are full Newton steps) is the kernel\'s rate with the per-pixel work (gate, start '
                    'value, result) spread over two steps instead of one'}
        del a_two
    if two_level:
        diff1 = _max_rel(a_single, a_exact)
        if not (diff1 <= 1e-12 and bool(torch.equal(torch.isnan(a_single), torch.isnan(a_exact)))):
            raise SystemExit(f'bench.py: the single-launch tolerance stop moved a pixel by {diff1:.3e} (> 1e-12) from the exact launch')
        out['value_single_launch'] = integrals_per_step / (elapsed_1 / args.steps)
        out['gn_single_launch'] = {
            'gn_ms': g1_ms, 'ms_per_step': 1e3 * elapsed_1 / args.steps, 'executed_pixel_iterations': st1['pixel_iterations'],
            'mean_iterations_per_unmasked_pixel': st1['pixel_iterations'] / live,
            'achieved': st1['pixel_iterations'] * flops_per_pixel_iter / (g1_ms * 1e-3) / 1e12,
            'frac': st1['pixel_iterations'] * flops_per_pixel_iter / (g1_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
            'max_diff_vs_exact': diff1,
            'note': 'two_level=False / DEXCT_GN_TWO_LEVEL=0: every pixel from the reference\'s start value 1e-6 on the full '
                    'tables, ended by the same tolerance rule (what `value` was before the short cut)'}
        del a_single
    out['value_exact'] = integrals_per_step / (elapsed_ex / args.steps)
    out['gn_exact'] = {'stop_tol': 0.0, 'gn_ms': gn_ex_ms, 'ms_per_step': 1e3 * elapsed_ex / args.steps,
                       'exact_bit_identical_to_full_loop': True, 'full_loop_ms': full_ms,
                       'executed_pixel_iterations': ex_stats['pixel_iterations'],
                       'mean_iterations_per_unmasked_pixel': ex_stats['pixel_iterations'] / live,
                       'achieved': ex_stats['pixel_iterations'] * flops_per_pixel_iter / (gn_ex_ms * 1e-3) / 1e12,
                       'frac': ex_stats['pixel_iterations'] * flops_per_pixel_iter / (gn_ex_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                       'full_loop_achieved': gn_flops_all / (full_ms * 1e-3) / 1e12,
                       'full_loop_frac': gn_flops_all / (full_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                       'default_max_diff_vs_exact': diff, 'pixels_compared': int(a_exact[..., 0].numel()),
                       'default_within_1e-12_of_exact_on_every_pixel': True,
                       'note': 'value_exact: the same step with stop_tol = 0 - the fixed iteration count of '
                               'matdecomp.py:114, every bit of it (checked here against a launch that executes all '
                               'iterations).  `value` is the default mode: every pixel ends at a fixed point of the full model '
                               'that the tolerance rule has verified to 1e-12 * max(|a|, 1) - reached from the tabulated fixed '
                               'points of the reference\'s walk - or after the reference\'s own n_iters steps; its results are '
                               'compared with the exact ones on every pixel above'}
    wl.gn_tol = None
    wl.step(False)                                                                  # the default results are back in place
    torch.cuda.synchronize()
    assert torch.equal(wl.a_out.view(torch.int64), a_default.view(torch.int64))
    del a_exact, a_default


def noisy_step(wl, args, out, sid_ms, ms_per_step, sid_info):
    """The step WITH quantum noise - the reference scales every spectrum to a dose before it projects (main.py:68, doses at
    :101) and reads sino_raw as photon counts (matdecomp.py:30,179): a noise-free scan makes the dose irrelevant.  Round 6: the
    default kernels sum the variance of the signal in the detection's own energy loop and draw the sample in registers
    (struct dexct_noise, ABI 6); round 5 fell back to the byte-volume kernel + a variance array + dexct_add_noise."""
    from .roofline import detection_floor_slots
    seed = 20261005
    n_rays, pj = wl.n_rays, wl.pj
    # ---- the projection alone, HIP events, like the step's projection (counts in the kernel's layout)
    def proj_ms(fn, reps=5):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    clean = wl.counts_nat.clone()
    kw = dict(out=wl.counts_nat, layout=None)
    ms_gauss = proj_ms(lambda: pj.project_tables(wl.mu_d, wl.w_d, w2_d=wl.w2_d, seed=seed, **kw))
    noisy = wl.counts_nat.clone()
    rel_noise = float(((noisy[:, ::50].double() - clean[:, ::50].double()) / clean[:, ::50].double()).std().item())
    ms_poisson = proj_ms(lambda: pj.project(wl.specs, noise='poisson', seed=seed, layout=None), reps=2)
    # round 5's path for the same scan: the byte-volume kernel with its variance output + dexct_add_noise (kernel=3 forces it)
    old_ms = None
    if getattr(pj, 'use_packed', False):
        pj3 = fp.Projector(wl.ct, wl.ph, view_range=(wl.vb, wl.ve), kernel=3)
        buf = torch.empty_like(wl.counts_nat)
        old_ms = proj_ms(lambda: pj3.project_tables(wl.mu_d, wl.w_d, w2_d=wl.w2_d, seed=seed, out=buf, layout=None), reps=3)
        same_sample = bool(torch.equal(buf, noisy))
        del pj3, buf
        if not same_sample:
            raise SystemExit('bench.py: the packed kernel\'s noisy sinogram differs from the byte-volume path\'s for the same seed')
    # ---- the whole step on the noisy scan, timed like the headline (barrier / synchronize around args.steps steps)
    wl.noise_seed = seed
    el, t_sid, t_gn, _ = wl.timed_steps(args.steps, 1)
    st = md.last_gn_stats()
    a_noisy = wl.a_out.clone()
    masked = float((wl.counts_nat[0] >= 0.95 * wl.gmax).float().mean().item())
    live = max((1.0 - masked) * n_rays, 1.0)
    # the default mode against the exact count on every pixel of the NOISY scan too
    wl.gn_tol = 0.0
    wl.step(False)
    torch.cuda.synchronize()
    diff = _max_rel(a_noisy, wl.a_out)
    same_nan = bool(torch.equal(torch.isnan(a_noisy), torch.isnan(wl.a_out)))
    wl.gn_tol = None
    wl.noise_seed = None
    if not (diff <= 1e-12 and same_nan):
        raise SystemExit(f'bench.py: noisy scan: the default mode moved a pixel by {diff:.3e} (> 1e-12) from the exact launch')
    step_ms = 1e3 * el / args.steps
    kname = sid_info['kname']
    noisy_floor = (sid_info['traversal_slots'] + detection_floor_slots(sid_info['lanes4'], sid_info['n_e_any'], wl.n_e_spec, noisy=True)
                   + 130.0 * n_rays) / 64.0 / sid_info['slots_per_s'] * 1e3
    out['noisy_step'] = {
        'projection_ms': {'noise_free': sid_ms, 'gaussian': ms_gauss, 'poisson': ms_poisson,
                          'gaussian_round5_path': old_ms},
        'ratio_gaussian_to_noise_free': ms_gauss / sid_ms,
        'kernels': {'gaussian': f'{kname}<NOISY> (variance summed in the detection rounds, one Philox block per ray, the sample drawn '
                                'in registers: no variance array, no sampling pass)' if getattr(pj, 'use_packed', False) or pj.cone
                                else f'{kname} with its variance output + add_noise_kernel',
                    'poisson': f'{kname} (path lengths) + poisson_detect_kernel (per-bin photon counts: inversion / rounded normal)',
                    'gaussian_round5_path': 'rows4_kernel + variance loop + add_noise_kernel'},
        'same_sample_as_round5_path_bit_for_bit': None if old_ms is None else True,
        'relative_noise_of_the_sample': rel_noise, 'photons_per_ray_and_spectrum': 1.0e6,
        'ms_per_step': step_ms, 'value': (n_rays * sum(wl.n_e_spec)) / (step_ms * 1e-3),
        'kernel_ms': {'siddon_project': mean(t_sid), 'gn_decompose': mean(t_gn)},
        'gn': {'mode': st.get('mode'), 'full_steps_per_unmasked_pixel': st['pixel_iterations'] / live if st.get('pixel_iterations') else None,
               'masked_fraction': masked, 'default_max_diff_vs_exact': diff, 'pixels_compared': int(a_noisy[..., 0].numel())},
        'valu_floor_ms': noisy_floor, 'achieved_over_floor': noisy_floor / ms_gauss,
        'note': 'the same step on the scan WITH quantum noise (Gaussian sample of the compound-Poisson signal; 1e6 photons per ray and '
                'spectrum -> relative_noise_of_the_sample): value counts the same ray-energy integrals; the Newton launch then meets '
                'counts off the noise-free manifold (its short cut applies per pixel as before; the result is compared with the exact '
                'count on every pixel).  Never part of the headline `value`, which is the noise-free parity configuration'}
    wl.step(False)                               # the noise-free results are back in place
    torch.cuda.synchronize()
    del clean, noisy, a_noisy


def mixed_precision(wl, args, out):
    """opt-in mixed-precision Newton (float32 bulk + float64 polish), never part of `value`"""
    a_mixed = torch.empty_like(wl.a_out)
    call = lambda: md.gn_device(wl.counts_nat[0], wl.counts_nat[1], wl.i0_d, wl.mus_d, args.iters, 'mixed', out=a_mixed, out_rc=wl.out_rc,
                                mask_max=wl.gmax, mask_frac=0.95)
    call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call()
    e1.record()
    torch.cuda.synchronize()
    diff = ((a_mixed - wl.a_out).abs() / wl.a_out.abs().clamp(min=1.0))
    out['gn_mixed_precision'] = {'gn_ms': e0.elapsed_time(e1),
                                 'max_diff_vs_f64': float(torch.nan_to_num(diff, nan=0.0).max().item()),
                                 'note': 'DEXCT_GN_PRECISION=mixed: first n-4 iterations float32, last 4 float64; '
                                         'opt-in, not the reference arithmetic, not used for value'}


def reduced_quadrature(wl, args, out, sid_ms):
    """the opt-in reduced energy quadrature (dex-ct-sim_amd/quadrature.py): same kernel, shorter table with a verified error
    bound; reported beside the step, never part of `value` (the step detects on the full grid)"""
    pj, nV = wl.pj, wl.nV
    t0 = time.perf_counter()
    _, mu_r, w_r, _ = pj.upload_tables(wl.specs, 'reduced')
    prep_s = time.perf_counter() - t0
    qi = pj.quadrature_info
    if qi is None:
        out['siddon_reduced_quadrature'] = {'applied': False}
        return
    c_red, l_red = torch.empty_like(wl.counts_nat), torch.empty_like(wl.log_nat)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # both grids' outputs in the kernel's own layout for the comparison below (untimed; the step takes its log with the
    # transpose), then the reduced grid timed like the step's projection: counts only
    pj.project_tables(wl.mu_d, wl.w_d, out=wl.counts_nat, layout=None, air=wl.air, log_out=wl.log_nat)
    pj.project_tables(mu_r, w_r, out=c_red, layout=None, air=wl.air, log_out=l_red)
    e0.record()
    for _ in range(5):
        pj.project_tables(mu_r, w_r, out=c_red, layout=None)
    e1.record()
    torch.cuda.synchronize()
    ms_r = e0.elapsed_time(e1) / 5
    cn, ln = wl.counts_nat, wl.log_nat
    dev_c = max(float(((c_red[:, v0:v0 + 50].double() - cn[:, v0:v0 + 50].double()).abs() / cn[:, v0:v0 + 50].double()).max())
                for v0 in range(0, nV, 50))
    dev_l = max(float((l_red[:, v0:v0 + 50] - ln[:, v0:v0 + 50]).abs().max()) for v0 in range(0, nV, 50))
    if dev_c > 2e-6:
        raise SystemExit(f'bench.py: reduced quadrature {dev_c:.2e} from the full grid (bound 2e-6)')
    out['siddon_reduced_quadrature'] = {
        'applied': True, 'opt_in': "get_sino(..., quadrature='reduced') / DEXCT_QUADRATURE=reduced", 'siddon_ms': ms_r,
        'full_grid_siddon_ms': sid_ms, 'speedup': sid_ms / ms_r, 'nodes': qi['nodes'], 'full_grid_bins': qi['n_full'],
        'nodes_per_spectrum': qi['nodes_per_spectrum'], 'verified_max_rel_err_f64': qi['max_rel_err'],
        'points_verified': qi['n_validated'], 'path_bounds_cm': qi['l_max'],
        'max_rel_deviation_of_counts_all_rays': dev_c, 'max_abs_deviation_of_log_sinogram_all_rays': dev_l,
        'host_preparation_s_once_per_phantom_and_spectra': prep_s,
        'rays_per_s': wl.n_rays / (ms_r * 1e-3),
        'note': 'positive-weight generalised Gauss quadrature on a subset of the grid (linear programme), verified in '
                'float64 over every path length the phantom allows; deviation measured here on every ray of the '
                'step against the full-grid launch (two float32 kernels); not used for value'}


def single_row(wl, args, out, prof, traffic_src, sid_info, plan_slabs):
    """single-row (the reference's own 2-D case), ray-parallel kernel; and the same scan with ONE WAVEFRONT PER RAY (the north
    star's mapping).  ``plan_slabs(ct, nz)``: in-plane slabs of a scan from the oracle's plan (cpu.py)."""
    n, dev = wl.n, wl.dev
    n_e_any, slots_per_s, n_e_spec = sid_info['n_e_any'], sid_info['slots_per_s'], wl.n_e_spec
    ct1 = dx.FanBeamGeometry(N_channels=args.channels, N_proj=args.views, gamma_fan=0.8230337, SID=60.0,
                             SDD=100.0, eid=True, detector_file=wl.det, N_rows=1)
    ph1 = synthetic.make_phantom(n, 1, extent=51.2, seed=1234)
    pj1 = fp.Projector(ct1, ph1, kernel=1)
    c1 = torch.empty((2, args.views, 1, args.channels), dtype=torch.float32, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pj1.project_tables(wl.mu_d, wl.w_d, out=c1)
    e0.record()
    for _ in range(10):
        pj1.project_tables(wl.mu_d, wl.w_d, out=c1)
    e1.record()
    torch.cuda.synchronize()
    ms1 = e0.elapsed_time(e1) / 10
    out['single_row'] = {'rays': args.views * args.channels, 'siddon_ms': ms1, 'kernel': 'rays_kernel (lanes = channels)',
                         'integrals_per_s': args.views * args.channels * sum(n_e_spec) / (ms1 * 1e-3)}
    # roofline of the reference's own geometry (one row, input/params.txt:12,18): vector issue.  Floor per slab and
    # lane of this formulation: fixed-point step 1, two slices 2, two range tests 2, two offsets + their selects 4,
    # two id selects 2, two counts (compare + add-with-carry) 4, id difference 2 = 17 vector instructions (the compiled
    # loop issues 20.5, profiles/r03_kernels.md), 2 byte loads; detection per ray and weighted energy 3 FMA + v_exp_f32
    # (2 slots) + 1 FMA per weighting spectrum.  Slabs from the oracle's plan of the same scan.
    slabs1 = plan_slabs(ct1, ph1, 1)
    floor1 = (17.0 * slabs1 + args.views * args.channels * (5.0 * n_e_any + sum(n_e_spec))) / 64.0
    floor1_ms = floor1 / slots_per_s * 1e3
    o1 = (prof.get('other_kernels') or {}).get('single_row', {})
    out['single_row']['roofline'] = {
        'kernel': 'rays_kernel<3, 64, 4>', 'bound': 'valu_issue', 'unit': 'G wave-instructions/s', 'peak': slots_per_s / 1e9,
        'achieved': floor1 / (ms1 * 1e-3) / 1e9, 'frac': floor1_ms / ms1, 'floor_wave_instructions': floor1,
        'floor_ms_at_%.1f_GHz' % CLOCK_GHZ: floor1_ms, 'slabs_per_launch': slabs1,
        'algorithmic_bytes_per_launch': 2.0 * slabs1 + 8.0 * args.views * args.channels,
        'measured_valu_instructions': o1.get('valu_insts'), 'measured_valu_busy': o1.get('valu_busy'),
        'measured_wait_any_share': o1.get('wait_any_share'), 'traffic': (o1.get('fetch_bytes_raw', 0) + o1.get('write_bytes', 0)) or None,
        'counters_source': traffic_src if o1 else None,
        'note': 'one %d x %d slice is L2 resident (%.0f KiB): not an HBM-bound kernel; frac = instruction floor / time' %
                (n, n, n * n / 1024.0)}
    pj6 = fp.Projector(ct1, ph1, kernel=6)
    pj6.project_tables(wl.mu_d, wl.w_d, out=c1)
    e0.record()
    for _ in range(10):
        pj6.project_tables(wl.mu_d, wl.w_d, out=c1)
    e1.record()
    torch.cuda.synchronize()
    ms6 = e0.elapsed_time(e1) / 10
    out['single_row']['wave_per_ray'] = {'kernel': 'wave_ray_kernel (lanes = slabs of one ray)', 'siddon_ms': ms6,
                                         'integrals_per_s': args.views * args.channels * sum(n_e_spec) / (ms6 * 1e-3)}
    del pj6, pj1


def cone_beam(wl, args, out, prof, traffic_src, sid_info, plan_slabs):
    """cone beam (true 3-D rays) on a slice of the same scan: the row-parallel kernel (what the host picks for <= 3 materials)
    and the one-thread-per-ray kernel beside it; the noisy scan in the same launch (round 6)"""
    n, rows, dev, ph = wl.n, wl.rows, wl.dev, wl.ph
    n_e_any, slots_per_s, n_e_spec = sid_info['n_e_any'], sid_info['slots_per_s'], wl.n_e_spec
    cv = max(1, min(args.views, 100))
    ctc = dx.FanBeamGeometry(N_channels=args.channels, N_proj=cv, gamma_fan=0.8230337, SID=60.0, SDD=100.0,
                             eid=True, detector_file=wl.det, N_rows=rows, cone=True, h_iso=ph.dz)
    cc = torch.empty((2, cv, rows, args.channels), dtype=torch.float32, device=dev)
    res = {}
    for kk, name in ((0, 'cone_rows_kernel'), (1, 'cone_kernel')):
        pjc = fp.Projector(ctc, ph, kernel=kk)
        pjc.project_tables(wl.mu_d, wl.w_d, out=cc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pjc.project_tables(wl.mu_d, wl.w_d, out=cc)
        e1.record()
        torch.cuda.synchronize()
        msc = e0.elapsed_time(e1)
        res[name] = {'siddon_ms': msc, 'rays_per_s': cv * rows * args.channels / (msc * 1e-3),
                     'integrals_per_s': cv * rows * args.channels * sum(n_e_spec) / (msc * 1e-3)}
        if kk == 0:
            pjc.project_tables(wl.mu_d, wl.w_d, out=cc, w2_d=wl.w2_d, seed=7)
            e0.record()
            pjc.project_tables(wl.mu_d, wl.w_d, out=cc, w2_d=wl.w2_d, seed=7)
            e1.record()
            torch.cuda.synchronize()
            res[name]['noisy_siddon_ms'] = e0.elapsed_time(e1)
        del pjc
    out['cone_beam'] = {'rays': cv * rows * args.channels, **res['cone_rows_kernel'],
                        'kernel': 'cone_cols_kernel (rows of a (view, channel) pair as lanes, voxel columns of 4 slabs staged in LDS)',
                        'thread_per_ray': res['cone_kernel']}
    # roofline: vector issue.  Floor per slab and lane (= detector row) of this formulation: 64-bit z step 1, slice
    # (shift + clamp) 2, count the b voxel 1, id comparison 1 = 5 (the compiled loop issues 6.3 with the two middle
    # voxels of a v-crossing slab), plus the staging of the two voxel columns of a slab by the 256 lanes of the
    # workgroup (4 instructions per 16-byte piece); exact corrections only at material boundaries; detection with the
    # energies in pairs: 1.5 exponent FMAs + 1 v_exp per energy, 0.5 per energy and weighting spectrum.  What the kernel
    # measures against this floor is latency, not issue: each batch of 4 slabs waits for its staged loads, and waves in
    # flight (8 per SIMD) are what hides it (profiles/r03_kernels.md).
    slabs_pair = plan_slabs(ctc, ph, n)                     # in-plane slabs, shared by the rows of a pair
    slabsc = slabs_pair * rows
    zs_col = ((n + 15) // 16) * 16 + 32                     # bytes per guarded voxel column (cone_zs)
    chunks = (rows + 255) // 256
    staging = slabs_pair * chunks * 2 * (zs_col / 16) * 4.0
    floorc = (5.0 * slabsc + staging + cv * rows * args.channels * (2.5 * n_e_any + 0.5 * sum(n_e_spec))) / 64.0
    floorc_ms = floorc / slots_per_s * 1e3
    oc = (prof.get('other_kernels') or {}).get('cone_rows', {})
    msc = res['cone_rows_kernel']['siddon_ms']
    out['cone_beam']['roofline'] = {
        'kernel': 'cone_cols_kernel<3, 4, 544>', 'bound': 'valu_issue', 'unit': 'G wave-instructions/s', 'peak': slots_per_s / 1e9,
        'achieved': floorc / (msc * 1e-3) / 1e9, 'frac': floorc_ms / msc, 'floor_wave_instructions': floorc,
        'floor_ms_at_%.1f_GHz' % CLOCK_GHZ: floorc_ms, 'lane_slabs_per_launch': slabsc,
        'algorithmic_bytes_per_launch': 2.0 * slabsc + 8.0 * cv * rows * args.channels,
        'measured_valu_instructions': oc.get('valu_insts'), 'measured_valu_busy': oc.get('valu_busy'),
        'measured_wait_any_share': oc.get('wait_any_share'), 'traffic': (oc.get('fetch_bytes_raw', 0) + oc.get('write_bytes', 0)) or None,
        'counters_source': traffic_src if oc else None}
    del cc
