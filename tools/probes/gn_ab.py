"""A/B of Newton-kernel BUILD variants on the benchmark's sinograms (512^3, 1000 x 800 x 512 = 4.1e8 pixels, 50 iterations).
Each variant is a complete libdexct_hip.so built with extra -D flags (tools/probes/build_variant.sh) and loaded in its own
process; exact mode (stop_tol = 0) in the plain and the reference order, and the default tolerance stop.

    python tools/probes/gn_ab.py [lib.so ...]          (no argument: the product library)
    VIEWS=125 python tools/probes/gn_ab.py ...          (an 8-GPU share)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, ROOT)
    import torch
    from dex_ct_sim_amd import _native
    if sys.argv[2] != 'default':
        _native.LIB_PATH = os.path.abspath(sys.argv[2])
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import forward_project as fp, matdecomp as md, synthetic
    det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
    n, views, chans = 512, int(os.environ.get('VIEWS', 1000)), 800
    ct = dx.FanBeamGeometry(N_channels=chans, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True,
                            detector_file=det, N_rows=n)
    ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
    specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
    pj = fp.Projector(ct, ph)
    _, mu_d, w_d, _ = pj.upload_tables(specs)
    counts = pj.project_tables(mu_d, w_d, layout=None)
    _, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
    gmax = counts[0].max().double()
    a = torch.empty(tuple(counts[0].shape) + (2,), dtype=torch.float64, device=counts.device)

    def run(**kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        md.gn_device(counts[0], counts[1], i0, mus, 50, 'f64', out=a, mask_max=gmax, **kw)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    modes = [('exact plain', dict(stop_tol=0.0)), ('exact ref-order', dict(stop_tol=0.0, out_rc=(n, chans))),
             ('default ref-order', dict(out_rc=(n, chans)))]
    reps = 3
    if os.environ.get('AB_DEFAULT_ONLY'):          # the default (short-cut) launch alone, more repetitions
        modes, reps = modes[2:], 12
    run(stop_tol=0.0)
    res = {m: [] for m, _ in modes}
    for rep in range(reps):
        for m, kw in modes:
            res[m].append(run(**kw))
    st = md.last_gn_stats()
    print(f'{sys.argv[2]:40s} ' + '  '.join(f'{m}: ' + '/'.join('%.1f' % t for t in ts) for m, ts in res.items()) +
          f'  stalled {st["stalled_lane_steps"]}', flush=True)
    sys.exit(0)

for lib in (sys.argv[1:] or ['default']):
    subprocess.run([sys.executable, os.path.abspath(__file__), '--child', lib], check=False)
