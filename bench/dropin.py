"""The PUBLIC boundary (NumPy in / NumPy out): get_sino x 2 + get_basismat_sinos(n_iters=50) as main.py:120,153 call
them, wall-clock, at configs[0]'s size (1200 x 800, one row) and at the workload's size.  PCIe-inclusive: never `value`."""
import os
import time

import numpy as np


def dropin_e2e(args, dx, fp, md, ct, ph, specs, det, dev):
    """Wall seconds of the reference's own call sequence through the public NumPy boundary.  'cold' = what a process's FIRST
    sequence costs - the device state is built (volume upload, layouts, plans), no page-locked memory is in the allocator's
    reserve (the result of get_basismat_sinos is locked chunk by chunk while the pipeline runs), the table of the Newton short
    cut is not in the process - in two variants: with the table on disk from an earlier process (DEXCT_CACHE_DIR, the normal
    case after a machine's first run) and without ('cold_no_disk_cache': the calibration runs inside the call).  'warm' is the
    second identical sequence of the same process."""
    import gc
    import tempfile
    import torch
    from dex_ct_sim_amd import _device, synthetic

    def fresh_process_state(cache_dir):
        fp.invalidate()
        md._table_cache.clear()
        gc.collect()
        torch._C._host_emptyCache()          # page-locked blocks of earlier results go back to the system
        _device.empty_pool()
        os.environ['DEXCT_CACHE_DIR'] = cache_dir

    def sequence(ct_, ph_, s1, s2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r1, l1 = dx.get_sino(ct_, ph_, s1)
        t1 = time.perf_counter()
        r2, l2 = dx.get_sino(ct_, ph_, s2)
        t2 = time.perf_counter()
        m1, m2 = dx.get_basismat_sinos(ct_, r1, r2, s1, s2, n_iters=50)
        t3 = time.perf_counter()
        ok = bool(np.isfinite(l1).all() and r1.dtype == np.float32 and m1.dtype == np.float64 and m1.shape == r1.shape)
        n = r1.size
        del r1, l1, r2, l2, m1, m2
        gc.collect()
        return {'get_sino_1_s': t1 - t0, 'get_sino_2_s': t2 - t1, 'get_basismat_sinos_s': t3 - t2, 'total_s': t3 - t0,
                'ok': ok}, n

    def kernel_ms(ct_, ph_, s1):            # one single-spectrum projection with both outputs, device resident
        pj = fp._projector(ct_, ph_, (0, ct_.N_proj))[0]
        _, mu_d, w_d, air = pj.upload_tables([s1])
        pj.project_tables(mu_d, w_d, air=air)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            pj.project_tables(mu_d, w_d, air=air)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 3

    res = {}
    ct0 = dx.FanBeamGeometry(N_channels=800, N_proj=1200, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True,
                             detector_file=det, N_rows=1)
    ph0 = synthetic.make_phantom(512, 1, extent=51.2, seed=1234)
    cases = [('configs[0] size: 1200 views x 800 channels x 1 row, 512^2 slice', ct0, ph0)]
    if not args.skip_dropin_full:
        cases.append((f'this workload: {ct.N_proj} x {ct.N_channels} x {ct.N_rows} rows, {ph.Nx}^3', ct, ph))
    keep_dir = os.environ.get('DEXCT_CACHE_DIR')
    tmp = tempfile.mkdtemp(prefix='dexct_bench_cache_')
    for label, ct_, ph_ in cases:
        fresh_process_state(tmp)             # an empty directory: the calibration runs in the call and leaves its table there
        for f in os.listdir(tmp):
            os.remove(os.path.join(tmp, f))
        cold_nodisk, n = sequence(ct_, ph_, specs[0], specs[1])
        fresh_process_state(tmp)             # ... where the next "process" finds it
        cold, _ = sequence(ct_, ph_, specs[0], specs[1])
        second, _ = sequence(ct_, ph_, specs[0], specs[1])
        warm, _ = sequence(ct_, ph_, specs[0], specs[1])
        k_ms = kernel_ms(ct_, ph_, specs[0])
        d2h_sino = 2 * n * 4                       # sino_raw + sino_log, float32
        floor_s = k_ms * 1e-3 + d2h_sino / 50e9
        res[label] = {'cold': cold, 'cold_no_disk_cache': cold_nodisk, 'second': second, 'warm': warm, 'rays': n,
                      'bytes': {'h2d_volume_once': int(ph_.volume.size), 'd2h_per_get_sino': d2h_sino,
                                'h2d_get_basismat_sinos': 2 * n * 4, 'd2h_get_basismat_sinos': n * 16},
                      'get_sino_kernels_ms': k_ms,
                      'get_sino_floor_s': floor_s, 'get_sino_over_floor': warm['get_sino_1_s'] / floor_s,
                      'note': 'floor = projection kernels (single spectrum, both outputs) + its device-to-host bytes at '
                              '50 GB/s; warm get_sino / floor is the boundary overhead factor'}
    fp.invalidate()
    if keep_dir is None:
        os.environ.pop('DEXCT_CACHE_DIR', None)
    else:
        os.environ['DEXCT_CACHE_DIR'] = keep_dir
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return res
