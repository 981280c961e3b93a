"""Arguments, the plain command starting its own ranks, and the process group (nothing here is timed)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(argv=None):
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
    ap.add_argument('--gather', default='auto', choices=['auto', 'root', 'direct', 'all'],
                    help='N > 1: how the raw sinograms are assembled (dex-ct-sim_amd/_shard.py): root = the north star\'s gather to '
                         'rank 0 (point-to-point, one transfer per peer link); direct = the same transfers to every rank (an '
                         'all-gather that does not depend on RCCL\'s algorithm); all = one all_gather_into_tensor per spectrum; '
                         'auto (default) = every mode takes warm-up steps and the fastest step runs the timed loop')
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
    ap.add_argument('--skip-noisy', action='store_true', help='omit the noisy step (quantum noise: the reference\'s dose-scaled mode)')
    args = ap.parse_args(argv)
    dv, dc = {'config2': (1000, 800), 'config3': (2000, 1024)}[args.workload]
    args.views = args.views or dv
    args.channels = args.channels or dc
    return args


def launch_ranks(args, script):
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
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + sys.argv[1:], env=e,
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


def init_ranks(args):
    """(world, rank, local_rank, backend): the process group of an N > 1 run (RCCL = torch's "nccl"), probed before anything is
    timed; DEXCT_DIST_BACKEND=gloo only rehearses the N > 1 control flow on a box with fewer GPUs than ranks."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
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
    return world, rank, local_rank, backend
