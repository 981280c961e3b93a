"""The timed region of bench.py, and nothing else.

A step = one pass of the hot path over the workload, inputs resident in HBM (BASELINE configs[2] at N = 1):
  plan -> Siddon traversal + polychromatic detection of BOTH spectra (one fused traversal) -> global maximum (air mask)
       -> Newton decomposition (n_iters = 50 asked, as main.py:153), mask fused, results in the reference's order
       -> sino_raw and sino_log of both spectra in the reference's [view][row][channel] order (dexct_transpose_log)
       -> (N > 1) assembly of the raw sinograms over RCCL, started chunk by chunk, overlapped with the rest.
Everything a step needs is built ONCE by Workload(); Workload.step() only launches.  This module imports the product
(dex_ct_sim_amd -> libdexct_hip.so) and torch; no oracle, no host-side solver, no NumPy (tests/test_host.py checks its source).
"""
import ctypes as C
import os
import statistics
import time

import torch
import torch.distributed as dist

import dex_ct_sim_amd as dx
from dex_ct_sim_amd import _native, _shard, forward_project as fp, matdecomp as md, synthetic
from dex_ct_sim_amd._device import ptr, stream_ptr

from .launch import ROOT


class Workload:
    """The scan, its device-resident state and the step over it.  Knobs the callers of step() may turn between steps:
    ``gn_tol`` (None: the default of get_basismat_sinos; 0.0: the fixed count), ``gn_mode`` (two_level of gn_device), ``gather_mode``
    (N > 1), ``assemble`` (N > 1; False: the sinogram stays sharded - the step without its fabric part), ``noise_seed`` (None:
    the noise-free expectation; an int: the scan WITH quantum noise for the dose the spectra are scaled to)."""

    def __init__(self, args, world, rank, local_rank):
        self.args, self.world, self.rank = args, world, rank
        self.dev = dev = torch.device('cuda', local_rank)
        n, rows = args.n, (args.rows or args.n)
        self.n, self.rows = n, rows
        # strong: the scan is fixed (args.views angles in all), each rank takes views/N of it; weak: args.views per rank
        self.total_views = total_views = args.views if args.scaling == 'strong' else args.views * world
        self.det = os.path.join(ROOT, 'dex-ct-sim_amd', 'input', 'detector', 'eta_eid_mv.bin')
        self.ct = dx.FanBeamGeometry(N_channels=args.channels, N_proj=total_views, gamma_fan=0.8230337, SID=60.0, SDD=100.0,
                                     eid=True, detector_file=self.det, N_rows=rows)
        self.ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
        self.specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
        vb, ve = _shard.split(total_views, rank, world)
        if args.shard_of > 1:
            if world != 1 or args.scaling != 'strong':
                raise SystemExit('--shard-of is a single-process, strong-scaling rehearsal')
            vb, ve = _shard.split(total_views, args.shard_rank, args.shard_of)
        self.vb, self.ve = vb, ve
        self.pj = pj = fp.Projector(self.ct, self.ph, view_range=(vb, ve), kernel=args.kernel)
        self.E, self.mu_d, self.w_d, self.air = pj.upload_tables(self.specs)
        # the variance weights of the noisy step (w x detector signal per photon): the same energy grid as w_d
        _, mu_h, w_h, w2_h = fp.merged_tables(self.ct, self.ph, self.specs, with_variance=True)
        assert tuple(w_h.shape) == tuple(self.w_d.shape)
        self.w2_d = torch.from_numpy(w2_h).to(device=dev, dtype=torch.float32).contiguous()
        self.n_e_spec = [int((self.w_d[k] != 0).sum().item()) for k in range(2)]
        _, self.i0, self.mus = md.decomposition_tables(self.ct, self.specs[0], self.specs[1])
        self.i0_d = torch.tensor(self.i0, dtype=torch.float64, device=dev)
        self.mus_d = torch.tensor(self.mus, dtype=torch.float64, device=dev)
        self.lib = pj.lib
        self.nV = nV = pj.n_local_views
        self.n_rays = nV * rows * args.channels
        self.native = native = pj.native_layout      # 1: [view][channel][row] (row-parallel kernels), 0: [view][row][channel]
        self.nat_shape = nat_shape = (nV, args.channels, rows) if native == 1 else (nV, rows, args.channels)
        self.counts_nat = torch.empty((2,) + nat_shape, dtype=torch.float32, device=dev)
        self.log_nat = torch.empty_like(self.counts_nat)      # get_sino's second output (main.py:120-122), from the same kernel
        # results in the reference's order ([view][row][channel]) are part of the step: the sinograms by a transpose pass, the
        # decomposition directly from the Newton kernel (dexct_gn_options.out_rows / out_channels, ABI 3)
        self.counts = torch.empty((2, nV, rows, args.channels), dtype=torch.float32, device=dev) if native == 1 else self.counts_nat
        self.log_ref = torch.empty_like(self.counts) if native == 1 else self.log_nat
        air = self.air
        self.air_c = [(C.c_float * 2)(float(air[0]), float(air[1])), (C.c_float * 1)(float(air[1]))]   # host floats: both spectra / the second
        self.a_out = torch.empty((nV, rows, args.channels, 2), dtype=torch.float64, device=dev)
        self.out_rc = (rows, args.channels) if native == 1 else None
        self.gn_tol = None              # None: the default of get_basismat_sinos / dexct_gn_decompose (tolerance stop, 1e-12); 0.0: the fixed count
        self.gn_mode = None             # two_level of gn_device: None = its default (the short cut), False = one launch
        self.noise_seed = None
        self.assemble = True
        self.gmax = torch.empty((), dtype=torch.float64, device=dev)
        self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        self.precision = args.gn_precision or md.DEFAULT_PRECISION
        self.gather_mode = args.gather if args.gather != 'auto' else 'root'
        self.n_chunks = 1
        if world > 1:
            self._sharded_buffers()

    def _sharded_buffers(self):
        """N > 1: the step works chunk by chunk: per chunk a compact [2, views, channel, row] projection output (the kernel's own
        layout), its transposed copy [2, views, row, channel] (what travels), and the chunk of the results"""
        args, nV, rows, native, dev = self.args, self.nV, self.rows, self.native, self.dev
        n_chunks = args.gather_chunks or (1 if self.gather_mode == 'all' else 4)
        self.n_chunks = n_chunks = max(1, min(n_chunks, nV))
        self.cb = cb = [_shard.split(nV, j, n_chunks) for j in range(n_chunks)]
        self.cn = [torch.empty((2, e - b) + self.nat_shape[1:], dtype=torch.float32, device=dev) for b, e in cb]
        self.cl = [torch.empty_like(t) for t in self.cn]
        self.cr = [torch.empty((2, e - b, rows, args.channels), dtype=torch.float32, device=dev) if native == 1 else self.cn[j]
                   for j, (b, e) in enumerate(cb)]
        self.clr = [torch.empty_like(t) if native == 1 else self.cl[j] for j, t in enumerate(self.cr)]
        self.cmax = torch.empty(n_chunks, dtype=torch.float64, device=dev)
        # where the assembled sinogram lands: on every rank for 'direct' / 'all'; 'root' needs it on rank 0 only (the other ranks
        # keep the buffer for the per-mode comparison: gather_views ignores out= on ranks that receive nothing)
        self.full_out = torch.empty((2, self.total_views, rows, args.channels), dtype=torch.float32, device=dev)

    # ---------------------------------------------------------------------------------------------------------------------
    def _project(self, out, views=None, air=None, log_out=None):
        kw = {} if self.noise_seed is None else dict(w2_d=self.w2_d, seed=self.noise_seed)
        self.pj.project_tables(self.mu_d, self.w_d, out=out, layout=None, views=views, air=air, log_out=log_out, **kw)

    def step_sharded(self, timed):
        """N > 1.  Plan (whole shard, once); per chunk of this rank's views: projection, its maximum, transpose into the
        reference's order (sino_raw and, from the same pass, sino_log) and - point-to-point modes - the START of the chunk's
        transfer; then the global maximum (one scalar all-reduce), the Newton launches chunk by chunk, and the wait for the
        transfers.  Mode 'all': one all_gather_into_tensor per spectrum, started after the last chunk (rounds 1-4).
        ``assemble`` False: the same step with the sinogram left sharded (no transfer is started)."""
        args, lib, pj, ev, native, rows = self.args, self.lib, self.pj, self.ev, self.native, self.rows
        mode = self.gather_mode
        st = stream_ptr()
        _native.check(lib.dexct_fan_plan(C.byref(pj.geom), ptr(pj.view_cs), ptr(pj.chan_cs), self.vb, self.ve, ptr(pj.plan), st), 'plan')
        if timed:
            ev[0].record()
        finishes = []
        for j, (b, e) in enumerate(self.cb):
            # (row-parallel kernels: sino_log comes with the transpose into the reference's order, dexct_transpose_log)
            if native == 1:
                self._project(self.cn[j], views=(b, e))
            else:
                self._project(self.cn[j], views=(b, e), air=self.air, log_out=self.cl[j])
            _native.check(lib.dexct_reduce_max(ptr(self.cn[j][0]), 0, self.cn[j][0].numel(), ptr(self.cmax[j]), st), 'max')
            if mode == 'all':                    # the whole shard in one buffer [2, views, row, channel]
                for k in range(2):
                    if native == 1:
                        _native.check(lib.dexct_transpose_log(ptr(self.cn[j][k]), ptr(self.counts[k, b:e]), ptr(self.clr[j][k]), self.air_c[k],
                                                              1, e - b, args.channels, rows, st), 'transpose counts + log')
                    else:
                        self.counts[k, b:e].copy_(self.cn[j][k])
            else:
                if native == 1:
                    _native.check(lib.dexct_transpose_log(ptr(self.cn[j]), ptr(self.cr[j]), ptr(self.clr[j]), self.air_c[0], 2, e - b,
                                                          args.channels, rows, st), 'transpose counts + log')
                if self.assemble:
                    finishes.append(_shard.gather_views(self.cr[j], self.total_views, view_dim=1, async_op=True, out=self.full_out, mode=mode,
                                                        root=0, part=(j, self.n_chunks), tag='bench'))
        if mode == 'all' and self.assemble:
            finishes.append(_shard.gather_views(self.counts, self.total_views, view_dim=1, async_op=True, out=self.full_out, mode='all',
                                                tag='bench'))
        if timed:
            ev[1].record()
        self.gmax.copy_(self.cmax.max())         # NaN-propagating like np.max (torch.max returns NaN if any element is NaN)
        gm = _shard.global_max(self.gmax)
        if timed:
            ev[2].record()
        for j, (b, e) in enumerate(self.cb):
            md.gn_device(self.cn[j][0], self.cn[j][1], self.i0, self.mus, args.iters, self.precision, out=self.a_out[b:e], out_rc=self.out_rc,
                         mask_max=gm, mask_frac=0.95, stop_tol=self.gn_tol, two_level=self.gn_mode, accumulate_stats=j > 0)
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
        return full, self.a_out

    def step(self, timed):
        if self.world > 1:
            return self.step_sharded(timed)
        args, lib, pj, ev = self.args, self.lib, self.pj, self.ev
        st = stream_ptr()
        _native.check(lib.dexct_fan_plan(C.byref(pj.geom), ptr(pj.view_cs), ptr(pj.chan_cs), self.vb, self.ve, ptr(pj.plan), st), 'plan')
        if timed:
            ev[0].record()
        if self.native == 1:  # sino_log comes with the transpose into the reference's order below (one pass for both outputs)
            self._project(self.counts_nat)
        else:
            self._project(self.counts_nat, air=self.air, log_out=self.log_nat)      # sino_raw AND sino_log
        if timed:
            ev[1].record()
        _native.check(lib.dexct_reduce_max(ptr(self.counts_nat[0]), 0, self.counts_nat[0].numel(), ptr(self.gmax), st), 'max')
        gm = _shard.global_max(self.gmax)
        if timed:
            ev[2].record()
        # air mask fused into the Newton kernel: threshold = 0.95 * (all-reduced) max, read from the device scalar
        # (the tables as host arrays: gn_device keeps their device copies, and those of the short cut, by content)
        md.gn_device(self.counts_nat[0], self.counts_nat[1], self.i0, self.mus, args.iters, self.precision, out=self.a_out, out_rc=self.out_rc,
                     mask_max=gm, mask_frac=0.95, stop_tol=self.gn_tol, two_level=self.gn_mode)
        if timed:
            ev[3].record()
        if self.native == 1:  # hand the sinograms over in the reference's [view][row][channel] order
            _native.check(lib.dexct_transpose_log(ptr(self.counts_nat), ptr(self.counts), ptr(self.log_ref), self.air_c[0], 2, self.nV,
                                                  args.channels, self.rows, st), 'transpose counts + log')
        return self.counts, self.a_out

    def barrier(self):
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps(self, n_steps, n_warm):
        """n_warm untimed steps, then exactly n_steps bracketed by barrier + synchronize; the MAX over ranks of the wall time.
        Returns (seconds, [projection ms], [Newton ms], [exposed assembly ms]) - the per-step lists from HIP events."""
        for _ in range(n_warm):
            self.step(False)
        self.barrier()
        ev = self.ev
        ts, tg, tx = [], [], []
        t0 = time.perf_counter()
        for _ in range(n_steps):
            self.step(True)
            torch.cuda.synchronize()
            ts.append(ev[0].elapsed_time(ev[1]))
            tg.append(ev[2].elapsed_time(ev[3]))
            if self.world > 1:
                tx.append(ev[4].elapsed_time(ev[5]))
        self.barrier()
        el = time.perf_counter() - t0
        if self.world > 1:
            t = torch.tensor(el, dtype=torch.float64, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, ts, tg, tx


def mean(xs):
    return float(statistics.fmean(xs)) if xs else 0.0
