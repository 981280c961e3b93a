"""Multi-rank path on CPU: gloo, world_size 2, 3 (ragged shards) and 8 (the node the path is built for)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, n_views, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from dex_ct_sim_amd import _shard
    full = torch.arange(2 * n_views * 3 * 5, dtype=torch.float32).reshape(2, n_views, 3, 5)
    b, e = _shard.my_views(n_views)
    owned = _shard.DEFAULT_GATHER_MODE == 'direct'
    for mode in _shard.GATHER_MODES:
        root = (world - 1) if mode == 'root' else 0            # (not rank 0: the root is an argument, not a convention)
        holds = mode != 'root' or rank == root
        got = _shard.gather_views(full[:, b:e].clone(), n_views, view_dim=1, mode=mode, root=root)
        # results of the public path are owned by the caller: a second gather of the same shape must not overwrite the first
        # (advisor finding of round 3); only reuse_out=True hands out the cached buffer
        again = _shard.gather_views((full[:, b:e] + 1.0).contiguous(), n_views, view_dim=1, mode=mode, root=root)
        if holds:
            owned = owned and bool(torch.equal(got, full) and torch.equal(again, full + 1.0) and got.data_ptr() != again.data_ptr())
        else:
            owned = owned and got is None and again is None    # gather to the root: the other ranks hold nothing
        r1 = _shard.gather_views(full[:, b:e].clone(), n_views, view_dim=1, tag='t' + mode, reuse_out=True, mode=mode, root=root)
        r2 = _shard.gather_views(full[:, b:e].clone(), n_views, view_dim=1, tag='t' + mode, reuse_out=True, mode=mode, root=root)
        if holds:
            owned = owned and r1.data_ptr() == r2.data_ptr() and bool(torch.equal(r2, full))
        # views on dimension 0, asynchronous, into a tensor of the caller's
        flat = full[0]
        dst = torch.full_like(flat, -1.0) if holds else None
        fin = _shard.gather_views(flat[b:e].clone(), n_views, view_dim=0, async_op=True, out=dst, mode=mode, root=root)
        res = fin()
        owned = owned and ((res is dst and bool(torch.equal(dst, flat))) if holds else res is None)
        # the shard in pieces (a step that projects its views in chunks sends each chunk when it exists): all pieces started,
        # then all awaited; ragged pieces included (3 pieces of 125 or 126 views)
        n_parts = 3
        fins = []
        for j in range(n_parts):
            pb, pe = _shard.part_bounds(n_views, world, (j, n_parts))[rank]
            fins.append(_shard.gather_views((full[:, pb:pe] * 2.0).contiguous(), n_views, view_dim=1, async_op=True, tag='p' + mode, reuse_out=True,
                                            mode=mode, root=root, part=(j, n_parts)))
        outs = [f() for f in fins]
        owned = owned and ((all(o is outs[0] for o in outs) and bool(torch.equal(outs[0], full * 2.0))) if holds else all(o is None for o in outs))
    try:
        _shard.gather_views(full[:, b:e].clone(), n_views, view_dim=1, part=(0, 2))     # parts need a common destination
        owned = False
    except ValueError:
        pass
    got = _shard.gather_views(full[:, b:e].clone(), n_views, view_dim=1)               # the default mode holds the result everywhere
    # the drop-in calls never gather to one rank, whatever DEXCT_GATHER says (advisor finding of round 5)
    keep = _shard.DEFAULT_GATHER_MODE
    for env_mode, want in (('root', 'direct'), ('direct', 'direct'), ('all', 'all')):
        _shard.DEFAULT_GATHER_MODE = env_mode
        owned = owned and _shard.dropin_mode() == want
        res = _shard.gather_views(full[:, b:e].clone(), n_views, view_dim=1, mode=_shard.dropin_mode())
        owned = owned and res is not None and bool(torch.equal(res, full))
    _shard.DEFAULT_GATHER_MODE = keep
    mx = _shard.global_max(torch.tensor(float(rank + 1), dtype=torch.float64))
    # np.max semantics over ranks (matdecomp.py:195-196): one rank's NaN makes the global maximum NaN on every rank;
    # -inf on a rank (an empty shard) does not disturb the others' values
    nan_mx = _shard.global_max(torch.tensor(float('nan') if rank == world - 1 else 5.0, dtype=torch.float64))
    inf_mx = _shard.global_max(torch.tensor(float('-inf') if rank == 0 else 2.5, dtype=torch.float64))
    q.put((rank, (b, e), bool(torch.equal(got, full)) and owned, float(mx), float(nan_mx), float(inf_mx)))
    dist.destroy_process_group()


# world 8 is the target (BASELINE configs[2] 1000 views, configs[3]/[4] 2000 views; 1003: ragged shards of 126 and 125)
@pytest.mark.parametrize('world,n_views', [(2, 10), (2, 7), (3, 8), (8, 1000), (8, 2000), (8, 1003)])
def test_gather_views_and_global_max(world, n_views):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() + world * 7 + n_views) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_views, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    covered = []
    for rank, (b, e), ok, mx, nan_mx, inf_mx in res:
        assert ok and mx == float(world)
        assert np.isnan(nan_mx) and inf_mx == 2.5
        covered += list(range(b, e))
    assert covered == list(range(n_views))       # contiguous, disjoint, complete


def test_split_is_balanced():
    from dex_ct_sim_amd import _shard
    for n in (1, 7, 1000, 2000):
        for w in (1, 2, 3, 8):
            parts = [_shard.split(n, r, w) for r in range(w)]
            sizes = [e - b for b, e in parts]
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
