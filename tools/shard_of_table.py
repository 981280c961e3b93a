"""profiles/r0N_shard_of.md: what ONE rank of a K-GPU strong-scaling run does, measured alone on one GPU
(`bench.py --shard-of K`), for K = 1, 2, 4, 8 on BASELINE configs[2] (1000 x 800) and configs[3] (2000 x 1024) - the
prediction the driver's first N-rank RCCL line is to be compared with.

    python tools/shard_of_table.py > profiles/r06_shard_of.md
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINK_GBS = 50.0          # assumed achievable per-link xGMI rate, one direction (nominal 7 x ~153 GB/s per GPU, both directions)

rows = []
for workload, views, chans in (('config2', 1000, 800), ('config3', 2000, 1024)):
    for K in (1, 2, 4, 8):
        for rank in sorted({0, K // 2}):
            cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', workload, '--shard-of', str(K), '--shard-rank',
                   str(rank), '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--skip-single-row', '--skip-quadrature', '--skip-gn-full-loop',
                   '--skip-dropin', '--skip-noisy']
            if K == 1:
                cmd = [c for c in cmd if c not in ('--shard-of', '--shard-rank')][:]
                cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', workload, '--steps', '3', '--warmup', '1',
                       '--no-cpu-baseline', '--skip-single-row', '--skip-quadrature', '--skip-gn-full-loop', '--skip-dropin', '--skip-noisy']
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            if p.returncode != 0:
                print(p.stderr[-2000:], file=sys.stderr)
                raise SystemExit(1)
            j = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][0])
            n_e = 134 + 74
            total_bytes = 2 * views * 512 * chans * 4                   # both raw sinograms of the whole scan, float32
            recv = total_bytes * (K - 1) / K
            rows.append(dict(workload=workload, K=K, rank=rank, views=j['config']['rays_per_gpu'] // (512 * chans),
                             step_ms=j['ms_per_step'], sid_ms=j['kernel_ms']['siddon_project'], gn_ms=j['kernel_ms']['gn_decompose'],
                             recv_gb=recv / 1e9, ring_ms=1e3 * recv / (LINK_GBS * 1e9),
                             direct_ms=1e3 * (total_bytes / K) / (LINK_GBS * 1e9) if K > 1 else 0.0,
                             integrals=views * 512 * chans * n_e))
            print(f'{workload} K={K} rank={rank}: step {j["ms_per_step"]:.1f} ms', file=sys.stderr, flush=True)

print('# One rank\'s share of a K-GPU strong-scaling run, measured alone on one MI355X (`bench.py --shard-of K`)\n')
print('Step = plan + fused dual-spectrum projection (sino_raw + sino_log) + global max + Newton (n_iters 50, default mode: the short cut, one step) + transposes, on the')
print('rank\'s contiguous views of the FIXED scan; no transfer runs here.  `gather` = bytes a rank RECEIVES when the raw sinograms (both')
print('spectra, float32, reference order) are assembled on it (mode direct / all: every rank; mode root: rank 0 only).  The transfers start')
print('chunk by chunk during the projection and overlap the Newton launches, so only `max(0, gather - Newton)` is exposed.  Two estimates at an')
print(f'assumed {LINK_GBS:.0f} GB/s per xGMI link and direction: `ring` = (K-1)/K x total / link (a ring is per-link bound),')
print('`direct` = one shard per peer link in parallel.  Predicted value = ray-energy integrals of the whole scan / (slowest')
print('measured rank step + exposed ring gather).\n')
print('| workload | K | rank | views | step ms | projection ms | Newton ms | gather GB / rank | ring ms | direct ms | exposed (ring) ms |')
print('|---|---|---|---|---|---|---|---|---|---|---|')
for r in rows:
    exposed = max(0.0, r['ring_ms'] - r['gn_ms'])
    print(f"| {r['workload']} | {r['K']} | {r['rank']} | {r['views']} | {r['step_ms']:.1f} | {r['sid_ms']:.2f} | {r['gn_ms']:.1f} | "
          f"{r['recv_gb']:.2f} | {r['ring_ms']:.1f} | {r['direct_ms']:.1f} | {exposed:.1f} |")
print('\nWith the Newton share at ~5 ms per rank at K = 8 (one chord step per pixel, profiles/r06_gn_chord.md) nothing hides the assembly: the')
print('step is what the transfers cost - bracketed here by the ring-bound estimate (one link: what an all-gather that RCCL runs as a ring')
print('would cost; bench.py --gather all) and the direct one (one transfer per peer link in parallel: what bench.py --gather root / direct')
print('issue, dexct_sino_gather; RCCL bus bandwidths of ~300 GB/s reported for 8-GPU xGMI meshes correspond to it).\n')
print('| workload | K | rank compute ms | step ms, ring-bound gather | integrals/s | speed-up | step ms, direct gather | integrals/s | speed-up |')
print('|---|---|---|---|---|---|---|---|---|')
base = {}
for wl in ('config2', 'config3'):
    for K in (1, 2, 4, 8):
        rr = [r for r in rows if r['workload'] == wl and r['K'] == K]
        comp = max(r['step_ms'] for r in rr)
        t_ring = max(r['step_ms'] + max(0.0, r['ring_ms'] - r['gn_ms']) for r in rr)
        t_dir = max(r['step_ms'] + max(0.0, r['direct_ms'] - r['gn_ms']) for r in rr)
        v_ring, v_dir = rr[0]['integrals'] / (t_ring * 1e-3), rr[0]['integrals'] / (t_dir * 1e-3)
        base.setdefault(wl, v_ring)
        print(f'| {wl} | {K} | {comp:.1f} | {t_ring:.1f} | {v_ring:.3e} | {v_ring / base[wl]:.2f} | {t_dir:.1f} | {v_dir:.3e} | {v_dir / base[wl]:.2f} |')
