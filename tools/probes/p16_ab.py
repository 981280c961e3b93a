"""A/B of BUILD variants of the packed stacked-fan kernel (tools/probes/build_variant.sh with SRC=siddon_packed) on the benchmark's
projection (512^3, 1000 x 800 x 512, both spectra): each library in its own process, noise-free and noisy launch.
    python tools/probes/p16_ab.py [lib.so ...]"""
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
    from dex_ct_sim_amd import forward_project as fp, synthetic
    det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
    ct = dx.FanBeamGeometry(N_channels=800, N_proj=1000, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=512)
    ph = synthetic.make_phantom(512, 512, extent=51.2, seed=1234)
    specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
    pj = fp.Projector(ct, ph)
    _, mu, w, w2 = fp.merged_tables(ct, ph, specs, with_variance=True)
    mu_d, w_d, w2_d = (torch.tensor(x, dtype=torch.float32, device='cuda').contiguous() for x in (pj.compact(mu), w, w2))
    out = pj.project_tables(mu_d, w_d, layout=None)

    def run(**kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pj.project_tables(mu_d, w_d, layout=None, out=out, **kw)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    res = {'noise-free': [], 'noisy': []}
    for _ in range(8):
        res['noise-free'].append(run())
        res['noisy'].append(run(w2_d=w2_d, seed=5))
    print(f'{sys.argv[2]:40s} ' + '  '.join(f'{m}: ' + '/'.join('%.2f' % t for t in ts) for m, ts in res.items()), flush=True)
    sys.exit(0)
for lib in (sys.argv[1:] or ['default']):
    subprocess.run([sys.executable, os.path.abspath(__file__), '--child', lib], check=False)
