"""N > 1: which mode assembles the sinogram in the timed loop, and what every mode costs (dex-ct-sim_amd/_shard.py:
'root' = the north star's gather to rank 0, point-to-point, one transfer per peer link; 'direct' = the same transfers to every
rank; 'all' = one all_gather_into_tensor per spectrum).

--gather auto (the default): every mode takes one warm-up step and two timed ones BEFORE the timed loop; the mode whose
step is fastest (max over ranks, agreed by all ranks: the choice is broadcast from rank 0) runs the loop.  No hardware with
more than one GPU was available while this was written, so the first real line must explain itself: besides `value` it carries
`value_compute_only` (the same step with the sinogram left sharded: what the GPUs do without the fabric), per mode the
assembly alone / what the step still waits for / the step time, and the bytes per second into a receiving rank."""
import time

import torch
import torch.distributed as dist

from dex_ct_sim_amd import _native, _shard
from dex_ct_sim_amd._device import ptr, stream_ptr

from .step import mean


def pick_mode(step_ms_by_mode, preference=_shard.GATHER_MODES):
    """The fastest step; within 2 % the earlier of `preference` wins (root first: the literal north star)."""
    best = min(step_ms_by_mode.values())
    for m in preference:
        if m in step_ms_by_mode and step_ms_by_mode[m] <= 1.02 * best:
            return m
    return min(step_ms_by_mode, key=step_ms_by_mode.get)


def choose_gather(wl, args):
    """--gather auto: measured warm-up steps per mode -> the mode of the timed loop (the same on every rank).  Returns
    (mode, {'step_ms': {mode: ms}, 'why': text}) or (args.gather, None) for an explicit flag."""
    if args.gather != 'auto':
        return args.gather, None
    step_ms = {}
    for mode in _shard.GATHER_MODES:
        wl.gather_mode = mode
        el, _, _, _ = wl.timed_steps(2, 1)
        step_ms[mode] = 1e3 * el / 2                       # (timed_steps returns the max over ranks)
    mode = pick_mode(step_ms)
    pick = [mode]
    dist.broadcast_object_list(pick, src=0)               # one decision for all ranks (the times are all-reduced, but be explicit)
    mode = pick[0]
    wl.gather_mode = mode
    why = (f'{mode}: {step_ms[mode]:.2f} ms per step in the warm-up against '
           + ', '.join(f'{m} {v:.2f}' for m, v in step_ms.items() if m != mode) + ' (1 warm-up + 2 timed steps each; ties within 2 % go '
           'to the earlier of root, direct, all)')
    return mode, {'step_ms': step_ms, 'why': why}


def measure(wl, args, backend, chosen, auto, ms_per_step, t_sid, t_gn, t_exposed, integrals_per_step):
    """The `multi_gpu` object of the JSON line (all ranks call this; it contains collectives)."""
    world, rank, dev = wl.world, wl.rank, wl.dev
    sid_ms, gn_ms = mean(t_sid), mean(t_gn)
    # ---- the step without its fabric part: the sinogram stays sharded (all-reduce of the maximum kept: the mask needs it)
    wl.assemble = False
    n_co = min(3, max(args.steps, 1))
    el_co, _, _, _ = wl.timed_steps(n_co, 1)
    wl.assemble = True
    compute_ms = 1e3 * el_co / n_co
    # ---- every mode of the assembly: its transfers alone (nothing else on the GPU), and the step with it
    wl.pj.project_tables(wl.mu_d, wl.w_d, out=wl.counts_nat, layout=None, air=wl.air, log_out=wl.log_nat)   # the whole shard, for the statistics
    if wl.native == 1:
        _native.check(wl.lib.dexct_transpose_batched(ptr(wl.counts_nat), ptr(wl.counts), 2 * wl.nV, args.channels, wl.rows, 4, stream_ptr()),
                      'transpose counts')
    by_mode = {}
    gather_allocs = None
    recv_bytes = 2 * wl.total_views * wl.rows * args.channels * 4 * (world - 1) / world          # everybody else's views, both spectra
    for mode in _shard.GATHER_MODES:
        wl.barrier()
        _shard.gather_views(wl.counts, wl.total_views, view_dim=1, out=wl.full_out, mode=mode, tag='bench')        # (buffers of the mode exist)
        wl.barrier()
        n_alloc0 = torch.cuda.memory_stats().get('allocation.all.allocated', 0)
        g0 = time.perf_counter()
        for _ in range(3):
            _shard.gather_views(wl.counts, wl.total_views, view_dim=1, out=wl.full_out, mode=mode, tag='bench')
            torch.cuda.synchronize()
        alone_ms = 1e3 * (time.perf_counter() - g0) / 3
        allocs = (torch.cuda.memory_stats().get('allocation.all.allocated', 0) - n_alloc0) / 3
        if mode == chosen:
            gather_allocs = allocs
            exposed, step_ms = mean(t_exposed), ms_per_step
        else:
            wl.gather_mode = mode
            el, _, _, tx = wl.timed_steps(min(3, args.steps), 1)
            wl.gather_mode = chosen
            exposed, step_ms = mean(tx), 1e3 * el / min(3, args.steps)
        t = torch.tensor([alone_ms, exposed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        by_mode[mode] = {'gather_ms': float(t[0]), 'gather_exposed_ms': float(t[1]), 'ms_per_step': step_ms,
                         'received_bytes_per_receiving_rank': recv_bytes, 'receiving_ranks': 1 if mode == 'root' else world,
                         'GBps_into_a_receiving_rank': recv_bytes / (float(t[0]) * 1e-3) / 1e9}
    wl.step(False)                               # the selected mode's results are back in place
    torch.cuda.synchronize()
    per_rank = [None] * world
    dist.all_gather_object(per_rank, {'rank': rank, 'views': [wl.vb, wl.ve], 'siddon_ms': sid_ms, 'gn_ms': gn_ms,
                                      'gather_exposed_ms': mean(t_exposed)})
    return {'backend': 'nccl (RCCL)' if backend == 'nccl' else f'{backend} (REHEARSAL: ranks share devices, host-staged '
                                                                f'transfers; not an RCCL measurement)',
            'gather': chosen, 'gather_flag': args.gather, 'gather_choice': auto,
            'view_chunks_per_rank': wl.n_chunks if chosen != 'all' else 1,
            'collectives_per_step': {'root': 'gather of the raw sinograms (reference order) to rank 0: one point-to-point transfer per peer '
                                             'and chunk in one RCCL group', 'direct': 'the same transfers to every rank (all-gather as '
                                             'world-1 sends + receives per rank)', 'all': 'all_gather_into_tensor per spectrum'}[chosen]
                                    + ' + all_reduce(max) of one float64',
            'gather_ms': by_mode[chosen]['gather_ms'], 'gather_exposed_ms': by_mode[chosen]['gather_exposed_ms'],
            'implied_GBps_into_root': by_mode['root']['GBps_into_a_receiving_rank'],
            'value_compute_only': integrals_per_step / (compute_ms * 1e-3), 'ms_per_step_compute_only': compute_ms,
            'fabric_share_of_step': max(0.0, 1.0 - compute_ms / ms_per_step),
            'gather_device_allocations_per_call': gather_allocs,     # buffers are allocated once
            'by_mode': by_mode, 'per_rank': per_rank,
            'predicted': 'profiles/r06_shard_of.md (one rank\'s share measured alone, 50 GB/s per xGMI link assumed): configs[2] / [3] at 8 '
                         'GPUs 6.8x / 7.4x for the compute (value_compute_only) and 4.1x with the point-to-point assembly on rank 0 (value); '
                         '0.7x if an all-gather rings over one link',
            'note': 'value_compute_only: the same step with the sinogram left sharded (no transfer started; the scalar all-reduce kept) - '
                    'what separates compute scaling from fabric time; gather_ms: the assembly alone (whole shard, nothing else running); '
                    'gather_exposed_ms: what the step still waits for after its last kernel (transfers start chunk by chunk during the '
                    'projection and overlap the Newton launches); by_mode: the same numbers and the step time for every mode, measured in '
                    'this run (3 steps each for the modes that did not run the timed loop); gather_choice: how --gather auto decided'}
