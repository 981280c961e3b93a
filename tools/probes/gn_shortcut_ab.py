"""The short-cut kernel of round 5 (gn_shortcut_kernel: lock-step tiles + stash) against round 4's (gn_refill_kernel<4, 2>):
the same pixels through three libraries, each in its own process -
  r04      the round-4 tree (tools/probes/_build/r04: `git archive 1aa11ee` + its built library)
  libm     this tree with gn.hip compiled -DDEXCT_GN_LIBM_LOG (round 4's logarithm and division in gn_start): the results must
           be BIT-IDENTICAL to r04's (same states, same exits; only who computes them when has changed)
  r05      this tree as built (the table-driven logarithm): within 1e-12 of the exact count, timing
on (a) `views` views of the benchmark sinograms (512 rows x 800 channels, out_rc), (b) the noisy thin-ray case of the tests
(closed cells, continued pixels, walks).
    tools/probes/build_variant.sh libm -DDEXCT_GN_LIBM_LOG && gpurun -- python tools/probes/gn_shortcut_ab.py [views]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == '--child':
    name, tree, lib_path, views, out = sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), sys.argv[6]
    sys.path.insert(0, tree)
    os.environ['DEXCT_CACHE_DIR'] = 'off'
    import torch
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import _native, forward_project as fp, matdecomp as md, synthetic
    from dex_ct_sim_amd._device import ptr, stream_ptr
    if lib_path != '-':
        _native.LIB_PATH = lib_path
    n = 512
    det = os.path.join(tree, 'dex-ct-sim_amd', 'input', 'detector', 'eta_eid_mv.bin')
    ct = dx.FanBeamGeometry(N_channels=800, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=n)
    ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
    specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
    pj = fp.Projector(ct, ph)
    _, mu_d, w_d, air = pj.upload_tables(specs)
    counts = pj.project_tables(mu_d, w_d, layout=None)
    _, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
    gmax = torch.empty((), dtype=torch.float64, device='cuda')
    pj.lib.dexct_reduce_max(ptr(counts[0]), 0, counts[0].numel(), ptr(gmax), stream_ptr())
    R, C = n, 800
    res = {}
    o = torch.empty((views, R, C, 2), dtype=torch.float64, device='cuda')
    for tag, kw in (('default', {}), ('exact', dict(stop_tol=0.0, two_level=False))):
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            md.gn_device(counts[0], counts[1], i0, mus, 50, 'f64', out=o, out_rc=(R, C), mask_max=gmax, mask_frac=0.95, **kw)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        st = md.last_gn_stats()
        print(f'{name}: {tag}: {best:.2f} ms  {st}', flush=True)
        res[tag] = o.cpu().numpy().copy()
    # (b) noisy thin rays with water-like negative components, float64 counts: closed cells, continued pixels, walks
    rng = np.random.default_rng(5)
    a_true = np.stack([rng.uniform(0.0, 30.0, 200000) * (rng.random(200000) < 0.8), rng.uniform(-0.3, 6.0, 200000)], -1)
    a_true[::7] *= 1e-3
    g = np.exp(-(a_true @ mus)) @ i0.T
    g = g * (1.0 + 10.0 ** rng.uniform(-6, -1.5, g.shape) * rng.standard_normal(g.shape))
    g[::1000] = np.nan
    g_d = torch.tensor(np.ascontiguousarray(g.T), dtype=torch.float64, device='cuda')
    for tag, kw in (('noisy_default', {}), ('noisy_exact', dict(stop_tol=0.0, two_level=False))):
        a = md.gn_device(g_d[0], g_d[1], i0, mus, 50, 'f64', **kw)
        torch.cuda.synchronize()
        print(f'{name}: {tag}: {md.last_gn_stats()}', flush=True)
        res[tag] = a.cpu().numpy().copy()
    np.savez(out, **res)
    sys.exit(0)

views = int(sys.argv[1]) if len(sys.argv) > 1 else 100
runs = [('r04', os.path.join(ROOT, 'tools', 'probes', '_build', 'r04'), '-'),
        ('libm', ROOT, os.path.join(ROOT, 'tools', 'probes', '_build', 'lib_libm.so')),
        ('r05', ROOT, '-')]
outs = {}
for name, tree, lib in runs:
    out = f'/tmp/gn_ab_{name}.npz'
    subprocess.run([sys.executable, os.path.abspath(__file__), '--child', name, tree, lib, str(views), out], check=True)
    outs[name] = np.load(out)


def cmp(x, y):
    same_bits = bool(np.array_equal(x.view(np.int64), y.view(np.int64)))
    with np.errstate(all='ignore'):
        d = np.abs(x - y) / np.maximum(np.abs(y), 1.0)
    return same_bits, float(np.nanmax(np.where(np.isfinite(d), d, 0.0))), bool(np.array_equal(np.isnan(x), np.isnan(y))), int((x.view(np.int64) != y.view(np.int64)).any(-1).sum())


for key in ('default', 'noisy_default'):
    ex = 'exact' if key == 'default' else 'noisy_exact'
    print(f'[{key}] exact: r04 vs r05 bit-identical: {cmp(outs["r04"][ex], outs["r05"][ex])[0]}')
    for a, b in (('libm', 'r04'), ('r05', 'r04')):
        sb, worst, nanp, ndiff = cmp(outs[a][key], outs[b][key])
        print(f'[{key}] {a} vs {b}: bit-identical {sb} ({ndiff} pixels differ), max rel diff {worst:.3e}, same NaN pattern {nanp}')
    for a in ('r04', 'libm', 'r05'):
        sb, worst, nanp, _ = cmp(outs[a][key], outs[a][ex])
        print(f'[{key}] {a} default vs its exact count: max rel diff {worst:.3e}, same NaN pattern {nanp}')
