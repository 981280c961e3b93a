"""Same-box A/B of the float64 Newton kernel's variants on the benchmark's sinograms (512^3, 1000 x 800 x 512):
grid caps (blocks_per_cu) and the full loop; exact mode, interleaved repetitions, results compared bit for bit with the
default.  (The history / ring / exponent / static-run variants of round 3 and the 5-waves-per-SIMD register allocation of round
4 are gone from the kernel; their measurements are in profiles/r03_gn_isa.md and r04_gn.md.  Build variants: tools/probes/gn_ab.py.)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
ref = torch.empty_like(a)


def run(out, env):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    md.gn_device(counts[0], counts[1], i0, mus, 50, 'f64', out=out, mask_max=gmax, stop_tol=0.0, **env)      # exact mode: the variants agree bit for bit
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


run(ref, {})
variants = [{}, {'blocks_per_cu': 3}, {'blocks_per_cu': 5}, {'blocks_per_cu': 8}, {'full_loop': True}]
times = {i: [] for i in range(len(variants))}
same = {}
for rep in range(3):
    for i, env in enumerate(variants):
        times[i].append(run(a, env))
        same[i] = bool(torch.equal(a.view(torch.int64), ref.view(torch.int64)))
for i, env in enumerate(variants):
    print(f'{str(env):70s} ms {" ".join("%.1f" % t for t in times[i])}   bit-identical to default: {same[i]}', flush=True)
