#!/usr/bin/env python3
"""Benchmark of the dex-ct hot path on MI355X.

A step = one pass of the hot path over BASELINE.json's metric configuration (configs[2]):
512^3 synthetic water/bone phantom, 1000 views x 800 channels, dual 80/140 kVp spectra, i.e.
  plan -> Siddon traversal + polychromatic detection of BOTH spectra (one fused traversal; both outputs of get_sino,
          sino_raw and sino_log, from the kernel's detection store)
       -> Gauss-Newton decomposition (50 iterations, as main.py:153) + air mask
       -> (N > 1) all-gather of the two raw sinograms over RCCL, overlapped with the decomposition.
Detector rows: BASELINE.json does not name a row count and a single row touches one slice of the
512^3 volume, so the workload is the stacked fan N_rows = Nz = 512; the
single-row case is reported under "single_row".  Inputs are resident in HBM before the timed region.

    python bench.py [--gpus N --steps K --warmup W]            (N > 1: starts the N rank processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: projection angles are sharded contiguously over the ranks (one process per GPU, RCCL).
  --scaling strong (default): the scan is FIXED - the metric's 1000 x 800 scan (or --workload config3: BASELINE
      configs[3], 2000 views x 1024 channels) - and rank r projects and decomposes views/N of it; one all-gather
      of the raw sinograms over xGMI, overlapped with the Newton kernel, plus one scalar all-reduce(max).
  --scaling weak: every rank projects `--views` angles of an N x views scan of the same phantom.
At N = 1 both are the same workload (BASELINE configs[2]).
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
CLOCK_GHZ = 2.4                # MI355X_MICROARCH.md: max clock (the traversal kernel holds ~2.2 under load)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--n', '--phantom-n', dest='n', type=int, default=512,
                    help='phantom is n^3 (--phantom-n: the spelling to use behind torch.distributed.run, whose own parser takes '
                         '--n for an abbreviation of its options)')
    ap.add_argument('--views', type=int, default=None, help='views of the scan (strong) / per GPU (weak)')
    ap.add_argument('--channels', type=int, default=None)
    ap.add_argument('--scaling', default='strong', choices=['strong', 'weak'])
    ap.add_argument('--workload', default='config2', choices=['config2', 'config3'],
                    help='config2: 1000 views x 800 channels (the metric); config3: 2000 x 1024 (BASELINE configs[3])')
    ap.add_argument('--rows', type=int, default=0, help='detector rows (0: n)')
    ap.add_argument('--shard-of', type=int, default=0,
                    help='single-GPU measurement of ONE rank\'s share of a K-GPU strong-scaling run (no collectives): '
                         'views [rank K-th] of the fixed scan; e.g. --workload config3 --shard-of 8 is the per-GPU work of '
                         'BASELINE configs[3]')
    ap.add_argument('--shard-rank', type=int, default=0)
    ap.add_argument('--gather', default='root', choices=['root', 'direct', 'all'],
                    help='N > 1: how the raw sinograms are assembled (dex-ct-sim_amd/_shard.py): root = the north star\'s gather to '
                         'rank 0 (point-to-point, one transfer per peer link); direct = the same transfers to every rank (an '
                         'all-gather that does not depend on RCCL\'s algorithm); all = one all_gather_into_tensor per spectrum')
    ap.add_argument('--gather-chunks', type=int, default=0,
                    help='N > 1: view chunks per rank; a chunk\'s transfer starts when its projection is done and overlaps the '
                         'projection and the Newton launches of the following chunks (0: 4 for root / direct, 1 for all)')
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--gn-precision', default=None, choices=[None, 'f64', 'mixed'])
    ap.add_argument('--kernel', type=int, default=0, help='0 choose, 1 ray-parallel, 2 row-parallel')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    ap.add_argument('--skip-single-row', action='store_true')
    ap.add_argument('--skip-quadrature', action='store_true', help='leave out the reduced-quadrature measurement')
    ap.add_argument('--skip-dropin', action='store_true', help='omit the public-boundary (NumPy in/out) timing')
    ap.add_argument('--skip-dropin-full', action='store_true', help='public-boundary timing at configs[0] size only')
    ap.add_argument('--skip-gn-full-loop', action='store_true',
                    help='omit the extra full-loop Newton launch (keeps rocprof per-kernel averages clean)')
    args = ap.parse_args()
    dv, dc = {'config2': (1000, 800), 'config3': (2000, 1024)}[args.workload]
    args.views = args.views or dv
    args.channels = args.channels or dc
    return args


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N rank processes from here - BEFORE this process
    makes any GPU call, and as children (never an exec of a process that has touched the GPU) - relay rank 0's
    JSON line and exit non-zero if any rank failed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               LOCAL_WORLD_SIZE=str(args.gpus))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if 'DEXCT_DIST_BACKEND' not in env:
        import torch          # device_count() does not initialise the GPU
        if torch.cuda.device_count() < args.gpus:
            # fewer devices than ranks (a one-GPU box): ranks share devices, which RCCL cannot do - rehearse the
            # N-rank control flow over gloo (collectives staged through the host) and say so in the output
            env['DEXCT_DIST_BACKEND'] = 'gloo'
            print(f'bench.py: {torch.cuda.device_count()} device(s) for {args.gpus} ranks - gloo rehearsal, ranks share '
                  f'devices (not an RCCL measurement)', file=sys.stderr)
    # every rank's stdout / stderr go to gpurun_out/rank<r>.log (rank 0's stdout carries the JSON line and is piped):
    # the first real RCCL run must be able to say what went wrong on WHICH rank
    log_dir = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(log_dir, exist_ok=True)
    procs, logs = [], []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        lf = open(os.path.join(log_dir, f'rank{r}.log'), 'wb')
        logs.append(lf)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else lf, stderr=lf))
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    # a rank that dies leaves the others waiting in a collective: watch all of them, and when one fails end the
    # others (exactly the processes started above)
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            time.sleep(2.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    codes = [p.wait() for p in procs]
    reader.join(timeout=10)
    for lf in logs:
        lf.close()
    # rank 0's JSON line goes to stdout; anything else a library printed there (gloo's connection banner) to stderr
    for ln in b''.join(buf).decode().splitlines():
        print(ln, file=sys.stdout if ln.startswith('{') else sys.stderr)
    sys.stdout.flush()
    if failed or any(codes):
        print(f'bench.py: rank exit codes {codes}', file=sys.stderr)
        first_bad = [r for r, c in enumerate(codes) if c not in (0, -9)] or [r for r, c in enumerate(codes) if c]
        for r in first_bad[:2]:                 # the rank(s) that failed by themselves (-9: killed by this launcher afterwards)
            try:
                tail = open(os.path.join(log_dir, f'rank{r}.log'), 'rb').read()[-3000:].decode(errors='replace')
            except OSError:
                tail = '(no log)'
            print(f'---- tail of gpurun_out/rank{r}.log (exit code {codes[r]}) ----\n{tail}', file=sys.stderr)
        return 1
    return 0


def segment_count(co, geom, view_cs, chan_cs, n_views_total, view_begin, view_end):
    """Exact number of Siddon segments per (view, channel) of this rank's shard, from the CPU oracle."""
    plan = co.plan(geom, view_cs, chan_cs, view_begin, view_end)
    return int(co.count_segments(geom, plan)), plan


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


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args))
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
        import datetime
        tmo = datetime.timedelta(seconds=float(os.environ.get('DEXCT_DIST_TIMEOUT_S', '300')))
        where = (f'rank {rank}/{world} local_rank {local_rank} device {torch.cuda.current_device()} of '
                 f'{torch.cuda.device_count()} backend {backend} rendezvous {os.environ.get("MASTER_ADDR")}:'
                 f'{os.environ.get("MASTER_PORT")}')
        try:
            if backend == 'nccl':
                dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank), timeout=tmo)
            else:
                dist.init_process_group(backend, timeout=tmo)
            probe = torch.ones(1, device='cuda' if backend == 'nccl' else 'cpu')
            dist.all_reduce(probe)                    # the communicator really works before anything is timed
            if float(probe.item()) != world:
                raise RuntimeError(f'all_reduce probe returned {float(probe.item())}, expected {world}')
        except Exception as exc:
            print(f'bench.py: process group did not come up within {tmo.total_seconds():.0f} s ({where}): {exc!r}\n'
                  f'  check: one process per GPU, HSA_ENABLE_IPC_MODE_LEGACY=0 (is {os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")!r}), '
                  f'MASTER_ADDR=127.0.0.1, a free MASTER_PORT, DEXCT_DIST_TIMEOUT_S to wait longer', file=sys.stderr, flush=True)
            raise
        print(f'bench.py: process group up ({where})', file=sys.stderr, flush=True)
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (the launcher sets WORLD_SIZE; plain '
                         f'`python bench.py --gpus N` starts its own ranks)')
    backend = (os.environ.get('DEXCT_DIST_BACKEND', 'nccl') if world > 1 else None)
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import _shard, forward_project as fp, matdecomp as md, synthetic

    dev = torch.device('cuda', local_rank)
    n, rows = args.n, (args.rows or args.n)
    # strong: the scan is fixed (args.views angles in all), each rank takes views/N of it; weak: args.views per rank
    total_views = args.views if args.scaling == 'strong' else args.views * world
    det = os.path.join(ROOT, 'dex-ct-sim_amd', 'input', 'detector', 'eta_eid_mv.bin')
    ct = dx.FanBeamGeometry(N_channels=args.channels, N_proj=total_views, gamma_fan=0.8230337, SID=60.0, SDD=100.0,
                            eid=True, detector_file=det, N_rows=rows)
    ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
    specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
    vb, ve = _shard.split(total_views, rank, world)
    if args.shard_of > 1:
        if world != 1 or args.scaling != 'strong':
            raise SystemExit('--shard-of is a single-process, strong-scaling rehearsal')
        vb, ve = _shard.split(total_views, args.shard_rank, args.shard_of)
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
    log_nat = torch.empty_like(counts_nat)          # get_sino's second output (main.py:120-122), from the same kernel
    # results in the reference's order ([view][row][channel]) are part of the step: the sinograms by a transpose pass, the
    # decomposition directly from the Newton kernel (dexct_gn_options.out_rows / out_channels, ABI 3)
    counts = torch.empty((2, nV, rows, args.channels), dtype=torch.float32, device=dev) if native == 1 else counts_nat
    log_ref = torch.empty_like(counts) if native == 1 else log_nat
    air_c = [(C.c_float * 2)(float(air[0]), float(air[1])), (C.c_float * 1)(float(air[1]))]      # host floats: both spectra / the second
    a_out = torch.empty((nV, rows, args.channels, 2), dtype=torch.float64, device=dev)
    out_rc = (rows, args.channels) if native == 1 else None
    # None: the default of get_basismat_sinos / dexct_gn_decompose (tolerance stop, 1e-12); 0.0: the fixed count exactly
    gn_tol = [None]
    gn_mode = [None]                    # two_level of gn_device: None = its default (the two-level solve), False = one launch
    gmax = torch.empty((), dtype=torch.float64, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    precision = args.gn_precision or md.DEFAULT_PRECISION

    n_chunks = args.gather_chunks or (1 if args.gather == 'all' else 4)
    n_chunks = max(1, min(n_chunks, nV))
    if world > 1:
        # the sharded step works chunk by chunk: per chunk a compact [2, views, channel, row] projection output (the kernel's
        # own layout), its transposed copy [2, views, row, channel] (what travels), and the chunk of the results
        cb = [_shard.split(nV, j, n_chunks) for j in range(n_chunks)]
        cn = [torch.empty((2, e - b) + nat_shape[1:], dtype=torch.float32, device=dev) for b, e in cb]
        cl = [torch.empty_like(t) for t in cn]
        cr = [torch.empty((2, e - b, rows, args.channels), dtype=torch.float32, device=dev) if native == 1 else cn[j] for j, (b, e) in enumerate(cb)]
        clr = [torch.empty_like(t) if native == 1 else cl[j] for j, t in enumerate(cr)]
        cmax = torch.empty(n_chunks, dtype=torch.float64, device=dev)
        # where the assembled sinogram lands: on every rank for 'direct' / 'all'; 'root' needs it on rank 0 only (the other ranks
        # keep the buffer for the per-mode comparison below: gather_views ignores out= on ranks that receive nothing)
        full_out = torch.empty((2, total_views, rows, args.channels), dtype=torch.float32, device=dev)
        full_all = full_out

    gather_mode = [args.gather]

    def step_sharded(timed):
        """N > 1.  Plan (whole shard, once); per chunk of this rank's views: projection, its maximum,
        transpose into the reference's order (sino_raw and, from the same pass, sino_log) and - point-to-point modes - the START of the chunk's transfer; then the global
        maximum (one scalar all-reduce), the Newton launches chunk by chunk, and the wait for the transfers.  Mode 'all': one all_gather_into_tensor per spectrum, started after the last chunk (rounds 1-4)."""
        mode = gather_mode[0]
        st = stream_ptr()
        _native.check(lib.dexct_fan_plan(C.byref(pj.geom), ptr(pj.view_cs), ptr(pj.chan_cs), vb, ve, ptr(pj.plan), st), 'plan')
        if timed:
            ev[0].record()
        finishes = []
        for j, (b, e) in enumerate(cb):
            # (row-parallel kernels: sino_log comes with the transpose into the reference's order, dexct_transpose_log)
            if native == 1:
                pj.project_tables(mu_d, w_d, out=cn[j], layout=None, views=(b, e))
            else:
                pj.project_tables(mu_d, w_d, out=cn[j], layout=None, air=air, log_out=cl[j], views=(b, e))
            _native.check(lib.dexct_reduce_max(ptr(cn[j][0]), 0, cn[j][0].numel(), ptr(cmax[j]), st), 'max')
            if mode == 'all':                    # the whole shard in one buffer [2, views, row, channel]
                for k in range(2):
                    if native == 1:
                        _native.check(lib.dexct_transpose_log(ptr(cn[j][k]), ptr(counts[k, b:e]), ptr(clr[j][k]), air_c[k], 1, e - b,
                                                              args.channels, rows, st), 'transpose counts + log')
                    else:
                        counts[k, b:e].copy_(cn[j][k])
            else:
                if native == 1:
                    _native.check(lib.dexct_transpose_log(ptr(cn[j]), ptr(cr[j]), ptr(clr[j]), air_c[0], 2, e - b, args.channels, rows, st),
                                  'transpose counts + log')
                finishes.append(_shard.gather_views(cr[j], total_views, view_dim=1, async_op=True, out=full_out, mode=mode, root=0,
                                                    part=(j, n_chunks), tag='bench'))
        if mode == 'all':
            finishes.append(_shard.gather_views(counts, total_views, view_dim=1, async_op=True, out=full_all, mode='all', tag='bench'))
        if timed:
            ev[1].record()
        gmax.copy_(cmax.max())                   # NaN-propagating like np.max (torch.max returns NaN if any element is NaN)
        gm = _shard.global_max(gmax)
        if timed:
            ev[2].record()
        for j, (b, e) in enumerate(cb):
            md.gn_device(cn[j][0], cn[j][1], i0, mus, args.iters, precision, out=a_out[b:e], out_rc=out_rc, mask_max=gm, mask_frac=0.95,
                         stop_tol=gn_tol[0], two_level=gn_mode[0], accumulate_stats=j > 0)
        if timed:
            ev[3].record()
        # basis-material sinograms stay view-sharded (each rank owns its angles, as a view-sharded back-projection would
        # consume them); only the raw sinogram is assembled, as the north star says
        if timed:
            ev[4].record()
        full = None
        for f in finishes:
            full = f()                           # the stream waits here for whatever of the transfers is not yet done
        if timed:
            ev[5].record()
        return full, a_out

    def step(timed):
        if world > 1:
            return step_sharded(timed)
        st = stream_ptr()
        _native.check(lib.dexct_fan_plan(C.byref(pj.geom), ptr(pj.view_cs), ptr(pj.chan_cs), vb, ve, ptr(pj.plan), st),
                      'plan')
        if timed:
            ev[0].record()
        if native == 1:       # sino_log comes with the transpose into the reference's order below (one pass for both outputs)
            pj.project_tables(mu_d, w_d, out=counts_nat, layout=None)
        else:
            pj.project_tables(mu_d, w_d, out=counts_nat, layout=None, air=air, log_out=log_nat)      # sino_raw AND sino_log
        if timed:
            ev[1].record()
        _native.check(lib.dexct_reduce_max(ptr(counts_nat[0]), 0, counts_nat[0].numel(), ptr(gmax), st), 'max')
        gm = _shard.global_max(gmax)
        if timed:
            ev[2].record()
        # air mask fused into the Newton kernel: threshold = 0.95 * (all-reduced) max, read from the device scalar
        # (the tables as host arrays: gn_device keeps their device copies, and those of the two-level solve, by content)
        md.gn_device(counts_nat[0], counts_nat[1], i0, mus, args.iters, precision, out=a_out, out_rc=out_rc, mask_max=gm,
                     mask_frac=0.95, stop_tol=gn_tol[0], two_level=gn_mode[0])
        if timed:
            ev[3].record()
        if native == 1:       # hand the sinograms over in the reference's [view][row][channel] order
            _native.check(lib.dexct_transpose_log(ptr(counts_nat), ptr(counts), ptr(log_ref), air_c[0], 2, nV, args.channels, rows, st),
                          'transpose counts + log')
        return counts, a_out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the table of the Newton short cut (once per pair of spectra, cached by content; outside the timed region like every
    # other table): timed here so that the bench line says what it costs
    t0 = time.perf_counter()
    if precision == 'f64':
        md._device_tables(i0, mus, dev, True)
        torch.cuda.synchronize()
    gate_prep_s = time.perf_counter() - t0
    def timed_steps(n_steps, n_warm):
        """n_warm untimed steps, then exactly n_steps bracketed by barrier + synchronize; the MAX over ranks of the wall time"""
        for _ in range(n_warm):
            step(False)
        barrier()
        ts, tg, tx = [], [], []
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step(True)
            torch.cuda.synchronize()
            ts.append(ev[0].elapsed_time(ev[1]))
            tg.append(ev[2].elapsed_time(ev[3]))
            if world > 1:
                tx.append(ev[4].elapsed_time(ev[5]))
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor(el, dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, ts, tg, tx

    elapsed, t_sid, t_gn, t_exposed = timed_steps(args.steps, args.warmup)
    ms_per_step = 1e3 * elapsed / args.steps
    rays_all = total_views * rows * args.channels           # rays of all ranks together (ragged shards included)
    if args.shard_of > 1:
        rays_all = n_rays                                   # only this shard is computed here
    integrals_per_step = rays_all * sum(n_e_spec)
    value = integrals_per_step / (elapsed / args.steps)
    sid_ms, gn_ms = float(np.mean(t_sid)), float(np.mean(t_gn))

    multi = None
    if world > 1:
        # every mode of the assembly: its transfers alone (nothing else on the GPU), and the step with it - what the step still
        # waits for after its last kernel (gather_exposed_ms) and the step time; the timed loop above ran args.gather
        pj.project_tables(mu_d, w_d, out=counts_nat, layout=None, air=air, log_out=log_nat)          # the whole shard, for the statistics below
        if native == 1:
            _native.check(lib.dexct_transpose_batched(ptr(counts_nat), ptr(counts), 2 * nV, args.channels, rows, 4, stream_ptr()), 'transpose counts')
        by_mode = {}
        gather_allocs = None
        for mode in _shard.GATHER_MODES:
            barrier()
            _shard.gather_views(counts, total_views, view_dim=1, out=full_out, mode=mode, tag='bench')        # (buffers of the mode exist)
            barrier()
            n_alloc0 = torch.cuda.memory_stats().get('allocation.all.allocated', 0)
            g0 = time.perf_counter()
            for _ in range(3):
                _shard.gather_views(counts, total_views, view_dim=1, out=full_out, mode=mode, tag='bench')
                torch.cuda.synchronize()
            alone_ms = 1e3 * (time.perf_counter() - g0) / 3
            allocs = (torch.cuda.memory_stats().get('allocation.all.allocated', 0) - n_alloc0) / 3
            if mode == args.gather:
                gather_allocs = allocs
                exposed, step_ms = float(np.mean(t_exposed)), ms_per_step
            else:
                gather_mode[0] = mode
                el, _, _, tx = timed_steps(min(3, args.steps), 1)
                gather_mode[0] = args.gather
                exposed, step_ms = float(np.mean(tx)), 1e3 * el / min(3, args.steps)
            t = torch.tensor([alone_ms, exposed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            recv_bytes = 2 * total_views * rows * args.channels * 4 * (world - 1) / world          # everybody else's views, both spectra
            by_mode[mode] = {'gather_ms': float(t[0]), 'gather_exposed_ms': float(t[1]), 'ms_per_step': step_ms,
                             'received_bytes_per_receiving_rank': recv_bytes, 'receiving_ranks': 1 if mode == 'root' else world,
                             'GBps_into_a_receiving_rank': recv_bytes / (float(t[0]) * 1e-3) / 1e9}
        step(False)                              # the selected mode's results are back in place
        torch.cuda.synchronize()
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {'rank': rank, 'views': [vb, ve], 'siddon_ms': sid_ms, 'gn_ms': gn_ms,
                                          'gather_exposed_ms': float(np.mean(t_exposed))})
        multi = {'backend': 'nccl (RCCL)' if backend == 'nccl' else f'{backend} (REHEARSAL: ranks share devices, host-staged '
                                                                      f'transfers; not an RCCL measurement)',
                 'gather': args.gather, 'view_chunks_per_rank': n_chunks if args.gather != 'all' else 1,
                 'collectives_per_step': {'root': 'gather of the raw sinograms (reference order) to rank 0: one point-to-point transfer per peer '
                                                  'and chunk in one RCCL group', 'direct': 'the same transfers to every rank (all-gather as '
                                                  'world-1 sends + receives per rank)', 'all': 'all_gather_into_tensor per spectrum'}[args.gather]
                                         + ' + all_reduce(max) of one float64',
                 'gather_ms': by_mode[args.gather]['gather_ms'], 'gather_exposed_ms': by_mode[args.gather]['gather_exposed_ms'],
                 'gather_device_allocations_per_call': gather_allocs,     # buffers are allocated once
                 'by_mode': by_mode, 'per_rank': per_rank,
                 'note': 'gather_ms: the assembly alone (whole shard, nothing else running); gather_exposed_ms: what the step still '
                         'waits for after its last kernel (transfers start chunk by chunk during the projection and overlap the '
                         'Newton launches); by_mode: the same two numbers and the step time for every mode, measured in this run '
                         '(3 steps each for the modes that are not --gather)'}

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    out = {
        'metric': 'Siddon ray-energy integrals/sec', 'value': value, 'unit': 'ray-energy integrals/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
        'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
        'dtype': 'f64' if precision == 'f64' else 'f32+f64', 'data': 'synthetic',
        'config': {'workload': f'{n}^3 water/bone phantom, {total_views} views x {args.channels} channels x '
                               f'{rows} rows (stacked fan), dual 140/80 kVp Kramers spectra ({n_e_spec[0]}+{n_e_spec[1]} '
                               f'energy bins), fused dual-spectrum Siddon + {args.iters}-iteration Gauss-Newton',
                   'baseline_config': 'configs[2]' if (args.workload == 'config2' and total_views == 1000) else
                                      ('configs[3]' if args.workload == 'config3' else 'configs[2] x N views (weak scaling)'),
                   'n': n, 'rays_per_gpu': n_rays, 'rays_total': rays_all,
                   'shard': None if args.shard_of <= 1 else f'views [{vb}, {ve}) = rank {args.shard_rank} of {args.shard_of} '
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
    if multi is not None:
        out['multi_gpu'] = multi

    # ---- rooflines.  The step's dominant kernel is the Newton kernel (98 % of the time): FP64 vector bound.  The
    # traversal kernel is stated against the bound its counters show: vector issue while the
    # volume is cache resident (<= 256 MiB Infinity Cache), HBM beyond.  No fraction here can exceed 1.
    from oracle import c_oracle as co
    import glob
    geom = co.make_geom(ct.N_proj, ct.N_channels, rows, ph.z_index, n, n, n, ph.dx, ph.dy, ph.dz, ct.SID, ct.SDD)
    seg_vc, _ = segment_count(co, geom, ct.view_cs(), ct.chan_cs(), total_views, vb, ve)
    # algorithmic bytes: S_ray x bytes per stored voxel + outputs; the packed volume stores a voxel in 2 bits
    b_vox = 0.25 if getattr(pj, 'use_packed', False) else 1.0
    # outputs of the timed launch: sino_raw of both spectra; sino_log too where the kernel writes it itself (row-parallel kernels
    # leave it to the pass that brings both outputs into the reference's order, dexct_transpose_log)
    alg_bytes = seg_vc * rows * b_vox + (2 if native == 1 else 4) * 4 * n_rays
    alg_gbps = alg_bytes / (sid_ms * 1e-3) / 1e9
    kname = 'rows16_kernel' if getattr(pj, 'use_packed', False) else \
        {1: 'rays_kernel', 2: 'rows_kernel', 3: 'rows4_kernel', 5: 'rows4t_kernel', 6: 'wave_ray_kernel'}[args.kernel or (3 if native == 1 else 1)]
    traffic = traffic_src = None
    prof = {}
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json'))):     # r01a < ... < r03a: the last match wins
        j = json.load(open(f))         # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (tools/profile_gpu.sh)
        if j.get('rays_per_gpu') == n_rays and kname in j.get('siddon_kernel', '') and j.get('n', 512) == n:
            prof, traffic_src = j, 'profiles/' + os.path.basename(f)        # same workload and kernel only
    traffic = prof.get('siddon_hbm_bytes_per_launch')
    vol_bytes = int(n * n * n * b_vox)
    cache_resident = vol_bytes <= 256 * 2 ** 20
    # vector-issue floor of the packed traversal + detection: per voxel dword visited
    #   rows4_kernel (1 B / voxel, 4 rows per dword): 2 vector instructions (bit-plane AND + its add; the weighted-sum
    #     add shared by two visits through v_add3)
    #   rows16_kernel (2 bits / voxel, 16 rows per dword): 3 (the lane's address add + 21 / 8 for the seven carry-save
    #     adders per 8 words; the ripple into the high counter bits can be amortised away)
    # and, per 4 rays and energy bin that any spectrum weights, 6 v_pk_fma (3 materials x 2 ray pairs) + 4 v_exp_f32
    # (2 issue slots each) + 2 v_pk_fma per spectrum that weights the bin; one slot = 4 cycles of one of the 1024 SIMDs.
    n_e_any = int(((w_d != 0).any(dim=0)).sum().item())
    # rays that crossed air only (they are the pixels the decomposition masks) are detected once per (view, channel)
    # pair, not per row: they are left out of the floor (their one detection per pair is not counted either)
    air_rays = float((counts_nat[0] >= 0.95 * gmax).float().mean().item())
    lanes = n_rays * (1.0 - air_rays) / 4.0
    rows_per_dword, per_visit = (16.0, 3.0) if kname == 'rows16_kernel' else (4.0, 2.0)
    floor_slots = per_visit * seg_vc * rows / rows_per_dword + lanes * (14.0 * n_e_any + 2.0 * sum(n_e_spec))
    slots_per_s = 1024 * CLOCK_GHZ * 1e9 / 4.0                 # wave-instruction issue slots per second, whole chip
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
                    'note': 'the %d MiB volume is L2 / Infinity-Cache resident: the algorithmic byte rate (%.0f GB/s) is not '
                            'an HBM rate and is reported as algorithmic_GBps only; the counters show vector issue as the '
                            'binding resource, so frac = instruction floor / time' % (vol_bytes >> 20, alg_gbps)})
    else:
        hbm_gbps = sid['traffic_GBps']
        sid.update({'bound': 'hbm', 'achieved': hbm_gbps if hbm_gbps is not None else min(alg_gbps, HBM_PEAK_GBS),
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': (hbm_gbps if hbm_gbps is not None else min(alg_gbps, HBM_PEAK_GBS)) / HBM_PEAK_GBS,
                    'note': 'volume larger than the Infinity Cache: achieved = measured fabric traffic (PMC) when a '
                            'matching profile exists, else the algorithmic byte rate capped at the peak; '
                            'traffic / algorithmic bytes = %s' % (None if traffic is None else round(traffic / alg_bytes, 3))})
    out['roofline_siddon'] = sid

    # air pixels (zeroed by the mask afterwards, matdecomp.py:204-205) are not iterated: count the others only
    masked = float((counts_nat[0] >= 0.95 * gmax).float().mean().item())
    out['gn_masked_fraction'] = masked
    # 28 flops + 1 exp per energy-iteration
    flops_per_pixel_iter = i0.shape[1] * (28 + 1)
    gn_flops_all = (1.0 - masked) * n_rays * args.iters * flops_per_pixel_iter
    # what the hardware issues of those: an energy only one spectrum weights takes 6 accumulations instead of 12 (17 of the
    # 29 flop), an energy no spectrum weights is dropped
    n_both = int(((i0[0] != 0) & (i0[1] != 0)).sum())
    n_one = int(((i0[0] != 0) ^ (i0[1] != 0)).sum())
    hw_share = (29.0 * n_both + 17.0 * n_one) / (29.0 * i0.shape[1])
    gstats = md.last_gn_stats()            # of the last timed step's launches (the default mode)
    two_level = bool(gstats) and gstats.get('mode') in md.SHORTCUT_MODES
    gn_name = ('gn_shortcut_kernel (start values from the table of the reference\'s fixed points + full-table steps)' if two_level
               else 'gn_refill_kernel<false>') if precision == 'f64' else 'gn_kernel<true,false>'
    main_ms = gstats['main_ms'] if gstats else gn_ms      # HIP events around the launch, on its stream, last timed step
    gn_form = two_level and gstats.get('mode') == 'one'    # the one step of the short cut is of the Gauss-Newton form: 6 of the 12 sums
    if gn_form:
        hw_share = (17.0 * n_both + 11.0 * n_one) / (29.0 * i0.shape[1])
    roof = {'kernel': gn_name, 'bound': 'valu_fp64' if precision == 'f64' else 'valu_fp32+fp64',
            'unit': 'TFLOP/s', 'peak': FP64_VALU_PEAK_TFLOPS, 'avg_launch_ms': main_ms, 'gn_ms_all_launches': gn_ms,
            'traffic': ((prof.get('gn_fetch_bytes_raw', 0) if prof.get('gn_fetch_counted_in_full') else prof.get('gn_fetch_bytes_x2_corrected', 0))
                        + prof.get('gn_write_bytes', 0)) or None,
            'traffic_source': traffic_src, 'algorithmic_bytes_per_launch': 24 * n_rays,
            'traffic_note': 'HBM-side bytes (FETCH_SIZE + WRITE_SIZE) of this kernel from the rocprofv3 --pmc passes of the same command '
                            'recorded in traffic_source (counters cannot be read from inside the timed run); algorithmic: 8 B of counts '
                            'in and 16 B of results out per pixel',
            'bound_note': 'neither HBM (24 - 41 B/pixel against >= 2e4 flops/pixel) nor MFMA (no dense contraction; FP64 MFMA and '
                          'FP64 VALU do not overlap on gfx950, DESIGN.md 4.4): bound = FP64 vector issue'}
    if gstats and gstats.get('pixel_iterations'):
        # EXECUTED work of the timed launch itself: the kernel counts the pixel-iterations it ran
        ex_flops = gstats['pixel_iterations'] * flops_per_pixel_iter
        live = max((1.0 - masked) * n_rays, 1.0)
        roof.update({'achieved': ex_flops / (main_ms * 1e-3) / 1e12, 'frac': ex_flops / (main_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                     'executed_pixel_iterations': gstats['pixel_iterations'],
                     'mean_iterations_per_unmasked_pixel': gstats['pixel_iterations'] / live,
                     'exit_saving': 1.0 - gstats['pixel_iterations'] / max((1.0 - masked) * n_rays * args.iters, 1.0),
                     'stalled_lane_steps': gstats.get('stalled_lane_steps'),
                     'hardware_fp64_flop_share': hw_share,
                     'hardware_fp64_utilisation': hw_share * ex_flops / (main_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                     'note': 'achieved = flops of the iterations the timed launch EXECUTED on the full tables (counted by the kernel; '
                             'SURVEY 8d: 28 flops + 1 exp per energy and iteration, unmasked pixels) / its time.  The 29 flop per energy '
                             'are ALGORITHMIC: of the %d energies of the union grid %d carry both spectra, %d only one '
                             '(6 accumulations instead of 12) and %d none (dropped), so the FP64 flops the hardware issues are '
                             'hardware_fp64_flop_share = %.2f of that and hardware_fp64_utilisation = share x frac; the rest of '
                             'the fully busy vector pipe is integer / move work and the per-iteration 2x2 solve '
                             '(profiles/r03_gn_isa.md).%s  exit_saving = share of the n_iters x pixels full-table iterations the '
                             'two-level solve and the exits made unnecessary - reported separately, not as throughput'
                             % (i0.shape[1], n_both, n_one, i0.shape[1] - n_both - n_one, hw_share,
                                '  The ONE step per pixel of the default short cut is of the Gauss-Newton form (the Hessian without '
                                'its (g / nu - 1) x second-derivative term - a second-order effect the tabulated kappa includes): 6 of the 12 '
                                'accumulations per energy, already taken out of hardware_fp64_flop_share; frac counts the unit of '
                                'SURVEY 8d - one Newton iteration of one pixel - at the reference\'s 29 flops per energy as for '
                                'every other mode' if gn_form else '')})
        if gn_form:
            roof['one_step_form'] = 'gauss-newton (6 of 12 sums per energy; the dropped term is second order and inside the tabulated kappa)'
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
    out['roofline'] = roof
    if precision == 'f64' and world == 1 and not args.skip_gn_full_loop:
        # ---- the reference's fixed iteration count, EXACTLY (stop_tol = 0): the same step timed the same way -> value_exact;
        # checked bit for bit against a launch that executes every iteration (DEXCT_GN_FULL_LOOP=1), and the default step's
        # results checked against it on every pixel
        a_default = a_out.clone()
        if two_level and gstats.get('mode') == 'one':
            # ---- the short cut with two steps and the tolerance rule for every pixel (round 4's form, 'start')
            gn_mode[0] = 'start'
            step(False)
            torch.cuda.synchronize()
            t_gn_2 = []
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step(True)
                torch.cuda.synchronize()
                t_gn_2.append(ev[2].elapsed_time(ev[3]))
            elapsed_2 = time.perf_counter() - t0
            st2 = md.last_gn_stats()
            a_two = a_out.clone()
            gn_mode[0] = None
        else:
            a_two = None
        if two_level:
            # ---- the default tolerance stop in ONE launch from the reference's start value (round 4's first form of the default)
            gn_mode[0] = False
            step(False)
            torch.cuda.synchronize()
            t_gn_1 = []
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step(True)
                torch.cuda.synchronize()
                t_gn_1.append(ev[2].elapsed_time(ev[3]))
            elapsed_1 = time.perf_counter() - t0
            st1 = md.last_gn_stats()
            a_single = a_out.clone()
            gn_mode[0] = None
        gn_tol[0] = 0.0
        step(False)
        torch.cuda.synchronize()
        t_gn_ex = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(True)
            torch.cuda.synchronize()
            t_gn_ex.append(ev[2].elapsed_time(ev[3]))
        elapsed_ex = time.perf_counter() - t0
        ex_stats = md.last_gn_stats()
        a_exact = a_out.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        md.gn_device(counts_nat[0], counts_nat[1], i0_d, mus_d, args.iters, 'f64', out=a_out, out_rc=out_rc, mask_max=gmax,
                     mask_frac=0.95, full_loop=True)             # DEXCT_GN_FLAG_FULL_LOOP: every iteration executed
        e1.record()
        torch.cuda.synchronize()
        full_ms = e0.elapsed_time(e1)
        exact_is_full = bool(torch.equal(a_out.view(torch.int64), a_exact.view(torch.int64)))
        diff = float(torch.nan_to_num((a_default - a_exact).abs() / a_exact.abs().clamp(min=1.0), nan=0.0).max().item())
        same_nan = bool(torch.equal(torch.isnan(a_default), torch.isnan(a_exact)))
        if not exact_is_full:
            raise SystemExit('bench.py: the exact launch (stop_tol = 0) differs from the full 50-iteration loop')
        if not (diff <= 1e-12 and same_nan):
            raise SystemExit(f'bench.py: the default mode moved a pixel by {diff:.3e} (> 1e-12) from the exact launch')
        if a_two is not None:
            diff2 = float(torch.nan_to_num((a_two - a_exact).abs() / a_exact.abs().clamp(min=1.0), nan=0.0).max().item())
            if not (diff2 <= 1e-12 and bool(torch.equal(torch.isnan(a_two), torch.isnan(a_exact)))):
                raise SystemExit(f'bench.py: the two-step short cut moved a pixel by {diff2:.3e} (> 1e-12) from the exact launch')
            g2_ms = float(np.mean(t_gn_2))
            out['value_two_step'] = integrals_per_step / (elapsed_2 / args.steps)
            out['gn_two_step'] = {
                'gn_ms': g2_ms, 'ms_per_step': 1e3 * elapsed_2 / args.steps, 'executed_pixel_iterations': st2['pixel_iterations'],
                'mean_iterations_per_unmasked_pixel': st2['pixel_iterations'] / max((1.0 - masked) * n_rays, 1.0),
                'achieved': st2['pixel_iterations'] * flops_per_pixel_iter / (g2_ms * 1e-3) / 1e12,
                'frac': st2['pixel_iterations'] * flops_per_pixel_iter / (g2_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                'max_diff_vs_exact': diff2,
                'note': "two_level='start' / DEXCT_GN_TWO_LEVEL=start: the short cut with two full-table steps and the tolerance rule for "
                        'every pixel (the default of round 4, on round 5\'s kernel): the same launch doing twice the counted work - its '
                        'frac is the kernel\'s rate with the per-pixel work (gate, start value, result) spread over two steps instead of one'}
            del a_two
        if two_level:
            diff1 = float(torch.nan_to_num((a_single - a_exact).abs() / a_exact.abs().clamp(min=1.0), nan=0.0).max().item())
            if not (diff1 <= 1e-12 and bool(torch.equal(torch.isnan(a_single), torch.isnan(a_exact)))):
                raise SystemExit(f'bench.py: the single-launch tolerance stop moved a pixel by {diff1:.3e} (> 1e-12) from the exact launch')
            g1_ms = float(np.mean(t_gn_1))
            out['value_single_launch'] = integrals_per_step / (elapsed_1 / args.steps)
            out['gn_single_launch'] = {
                'gn_ms': g1_ms, 'ms_per_step': 1e3 * elapsed_1 / args.steps, 'executed_pixel_iterations': st1['pixel_iterations'],
                'mean_iterations_per_unmasked_pixel': st1['pixel_iterations'] / max((1.0 - masked) * n_rays, 1.0),
                'achieved': st1['pixel_iterations'] * flops_per_pixel_iter / (g1_ms * 1e-3) / 1e12,
                'frac': st1['pixel_iterations'] * flops_per_pixel_iter / (g1_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                'max_diff_vs_exact': diff1,
                'note': 'two_level=False / DEXCT_GN_TWO_LEVEL=0: every pixel from the reference\'s start value 1e-6 on the full '
                        'tables, ended by the same tolerance rule (what `value` was before the short cut)'}
            del a_single
        gn_ex_ms = float(np.mean(t_gn_ex))
        out['value_exact'] = integrals_per_step / (elapsed_ex / args.steps)
        out['gn_exact'] = {'stop_tol': 0.0, 'gn_ms': gn_ex_ms, 'ms_per_step': 1e3 * elapsed_ex / args.steps,
                           'exact_bit_identical_to_full_loop': True, 'full_loop_ms': full_ms,
                           'executed_pixel_iterations': ex_stats['pixel_iterations'],
                           'mean_iterations_per_unmasked_pixel': ex_stats['pixel_iterations'] / max((1.0 - masked) * n_rays, 1.0),
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
        gn_tol[0] = None
        step(False)                                                                  # the default results are back in place
        torch.cuda.synchronize()
        assert torch.equal(a_out.view(torch.int64), a_default.view(torch.int64))
        del a_exact, a_default
    if 'frac' not in roof:
        roof.update({'achieved': None, 'frac': None, 'note': 'executed-iteration count not available in this mode'})

    # ---- opt-in mixed-precision Newton (float32 bulk + float64 polish), never part of `value`
    if precision == 'f64' and world == 1:
        a_mixed = torch.empty_like(a_out)
        md.gn_device(counts_nat[0], counts_nat[1], i0_d, mus_d, args.iters, 'mixed', out=a_mixed, out_rc=out_rc, mask_max=gmax,
                     mask_frac=0.95)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        md.gn_device(counts_nat[0], counts_nat[1], i0_d, mus_d, args.iters, 'mixed', out=a_mixed, out_rc=out_rc, mask_max=gmax,
                     mask_frac=0.95)
        e1.record()
        torch.cuda.synchronize()
        diff = ((a_mixed - a_out).abs() / a_out.abs().clamp(min=1.0))
        out['gn_mixed_precision'] = {'gn_ms': e0.elapsed_time(e1),
                                     'max_diff_vs_f64': float(torch.nan_to_num(diff, nan=0.0).max().item()),
                                     'note': 'DEXCT_GN_PRECISION=mixed: first n-4 iterations float32, last 4 float64; '
                                             'opt-in, not the reference arithmetic, not used for value'}
        del a_mixed, diff

    # ---- the opt-in reduced energy quadrature (dex-ct-sim_amd/quadrature.py): same kernel, shorter table with a verified
    # error bound; reported beside the step, never part of `value` (the step detects on the full grid)
    if not args.skip_quadrature and world == 1:
        t0 = time.perf_counter()
        _, mu_r, w_r, _ = pj.upload_tables(specs, 'reduced')
        prep_s = time.perf_counter() - t0
        qi = pj.quadrature_info
        if qi is None:
            out['siddon_reduced_quadrature'] = {'applied': False}
        else:
            c_red, l_red = torch.empty_like(counts_nat), torch.empty_like(log_nat)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            # both grids' outputs in the kernel's own layout for the comparison below (untimed; the step takes its log with the
            # transpose), then the reduced grid timed like the step's projection: counts only
            pj.project_tables(mu_d, w_d, out=counts_nat, layout=None, air=air, log_out=log_nat)
            pj.project_tables(mu_r, w_r, out=c_red, layout=None, air=air, log_out=l_red)
            e0.record()
            for _ in range(5):
                pj.project_tables(mu_r, w_r, out=c_red, layout=None)
            e1.record()
            torch.cuda.synchronize()
            ms_r = e0.elapsed_time(e1) / 5
            dev_c = max(float(((c_red[:, v0:v0 + 50].double() - counts_nat[:, v0:v0 + 50].double()).abs()
                               / counts_nat[:, v0:v0 + 50].double()).max()) for v0 in range(0, nV, 50))
            dev_l = max(float((l_red[:, v0:v0 + 50] - log_nat[:, v0:v0 + 50]).abs().max()) for v0 in range(0, nV, 50))
            if dev_c > 2e-6:
                raise SystemExit(f'bench.py: reduced quadrature {dev_c:.2e} from the full grid (bound 2e-6)')
            out['siddon_reduced_quadrature'] = {
                'applied': True, 'opt_in': "get_sino(..., quadrature='reduced') / DEXCT_QUADRATURE=reduced", 'siddon_ms': ms_r,
                'full_grid_siddon_ms': sid_ms, 'speedup': sid_ms / ms_r, 'nodes': qi['nodes'], 'full_grid_bins': qi['n_full'],
                'nodes_per_spectrum': qi['nodes_per_spectrum'], 'verified_max_rel_err_f64': qi['max_rel_err'],
                'points_verified': qi['n_validated'], 'path_bounds_cm': qi['l_max'],
                'max_rel_deviation_of_counts_all_rays': dev_c, 'max_abs_deviation_of_log_sinogram_all_rays': dev_l,
                'host_preparation_s_once_per_phantom_and_spectra': prep_s,
                'rays_per_s': n_rays / (ms_r * 1e-3),
                'note': 'positive-weight generalised Gauss quadrature on a subset of the grid (linear programme), verified in '
                        'float64 over every path length the phantom allows; deviation measured here on every ray of the '
                        'step against the full-grid launch (two float32 kernels); not used for value'}
            del c_red, l_red

    # ---- single-row (the reference's own 2-D case), ray-parallel kernel
    if not args.skip_single_row and world == 1:
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
        out['single_row'] = {'rays': args.views * args.channels, 'siddon_ms': ms1, 'kernel': 'rays_kernel (lanes = channels)',
                             'integrals_per_s': args.views * args.channels * sum(n_e_spec) / (ms1 * 1e-3)}
        # roofline of the reference's own geometry (one row, input/params.txt:12,18): vector issue.  Floor per slab and
        # lane of this formulation: fixed-point step 1, two slices 2, two range tests 2, two offsets + their selects 4,
        # two id selects 2, two counts (compare + add-with-carry) 4, id difference 2 = 17 vector instructions (the compiled
        # loop issues 20.5, profiles/r03_kernels.md), 2 byte loads; detection per ray and weighted energy 3 FMA + v_exp_f32
        # (2 slots) + 1 FMA per weighting spectrum.  Slabs from the oracle's plan of the same scan.
        geom1 = co.make_geom(ct1.N_proj, ct1.N_channels, 1, 0, n, n, 1, ph1.dx, ph1.dy, ph1.dz, ct1.SID, ct1.SDD)
        plan1 = co.plan(geom1, ct1.view_cs(), ct1.chan_cs(), 0, ct1.N_proj)
        slabs1 = float(plan1['n_slabs'].sum())
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
        # the same scan with ONE WAVEFRONT PER RAY (the north star's mapping): A/B of DESIGN.md section 4.1
        pj6 = fp.Projector(ct1, ph1, kernel=6)
        pj6.project_tables(mu_d, w_d, out=c1)
        e0.record()
        for _ in range(10):
            pj6.project_tables(mu_d, w_d, out=c1)
        e1.record()
        torch.cuda.synchronize()
        ms6 = e0.elapsed_time(e1) / 10
        out['single_row']['wave_per_ray'] = {'kernel': 'wave_ray_kernel (lanes = slabs of one ray)', 'siddon_ms': ms6,
                                             'integrals_per_s': args.views * args.channels * sum(n_e_spec) / (ms6 * 1e-3)}
        del pj6

    # ---- cone beam (true 3-D rays) on a slice of the same scan: the row-parallel kernel (what the host picks for
    # <= 3 materials) and the one-thread-per-ray kernel beside it
    if not args.skip_single_row and rows >= 8 and world == 1:
        cv = max(1, min(args.views, 100))
        ctc = dx.FanBeamGeometry(N_channels=args.channels, N_proj=cv, gamma_fan=0.8230337, SID=60.0, SDD=100.0,
                                 eid=True, detector_file=det, N_rows=rows, cone=True, h_iso=ph.dz)
        cc = torch.empty((2, cv, rows, args.channels), dtype=torch.float32, device=dev)
        res = {}
        for kk, name in ((0, 'cone_rows_kernel'), (1, 'cone_kernel')):
            pjc = fp.Projector(ctc, ph, kernel=kk)
            pjc.project_tables(mu_d, w_d, out=cc)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            pjc.project_tables(mu_d, w_d, out=cc)
            e1.record()
            torch.cuda.synchronize()
            msc = e0.elapsed_time(e1)
            res[name] = {'siddon_ms': msc, 'rays_per_s': cv * rows * args.channels / (msc * 1e-3),
                         'integrals_per_s': cv * rows * args.channels * sum(n_e_spec) / (msc * 1e-3)}
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
        geomc = co.make_geom(ctc.N_proj, ctc.N_channels, 1, 0, n, n, n, ph.dx, ph.dy, ph.dz, ctc.SID, ctc.SDD)
        planc = co.plan(geomc, ctc.view_cs(), ctc.chan_cs(), 0, ctc.N_proj)
        slabs_pair = float(planc['n_slabs'].sum())              # in-plane slabs, shared by the rows of a pair
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

    # ---- the PUBLIC boundary (NumPy in / NumPy out): get_sino x 2 + get_basismat_sinos(n_iters=50) as
    # main.py:120,153 call them, wall-clock, at configs[0]'s size (1200 x 800, one row) and at this workload's size.
    if world == 1 and not args.skip_dropin:
        out['dropin_e2e'] = dropin_e2e(args, dx, fp, md, ct, ph, specs, det, dev)

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
