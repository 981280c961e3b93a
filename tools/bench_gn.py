"""Same-box A/B of the float64 Newton kernel's variants on the benchmark's sinograms (512^3, 1000 x 800 x 512):
DEXCT_GN_MINW (5: 96 VGPRs + spills, 4: 110 VGPRs, no scratch) x DEXCT_GN_IEXP (v_ldexp_f64 vs integer exponent add),
interleaved repetitions, results compared bit for bit with the default."""
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
    for k in ('DEXCT_GN_MINW', 'DEXCT_GN_IEXP', 'DEXCT_GN_FULL_LOOP', 'DEXCT_GN_HLDS', 'DEXCT_GN_HIST', 'DEXCT_GN_CHUNK', 'DEXCT_GN_RING',
              'DEXCT_GN_VAR', 'DEXCT_GN_QUEUE', 'DEXCT_GN_BLOCKS_PER_CU'):
        os.environ.pop(k, None)
    os.environ.update(env)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    md.gn_device(counts[0], counts[1], i0, mus, 50, 'f64', out=out, mask_max=gmax)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


run(ref, {})
variants = [{}, {'DEXCT_GN_RING': '0'}, {'DEXCT_GN_MINW': '4'}, {'DEXCT_GN_RING': '0', 'DEXCT_GN_MINW': '4'},
            {'DEXCT_GN_FULL_LOOP': '1'}, {'DEXCT_GN_FULL_LOOP': '1', 'DEXCT_GN_RING': '0'}]
if os.environ.get('GN_VARIANTS') == 'queue':      # run queue (default) against static runs, and pixels per lane and fetch
    variants = [{}, {'DEXCT_GN_QUEUE': '0'}] + [{'DEXCT_GN_CHUNK': c} for c in ('1', '2', '8', '16')] + [{'DEXCT_GN_FULL_LOOP': '1'}, {'DEXCT_GN_FULL_LOOP': '1', 'DEXCT_GN_QUEUE': '0'}]
if os.environ.get('GN_VARIANTS') == 'cap':        # workgroups launched per CU in run-queue mode
    variants = [{}, {'DEXCT_GN_QUEUE': '0'}] + [{'DEXCT_GN_BLOCKS_PER_CU': c} for c in ('4', '5', '6', '8', '16')] + [{'DEXCT_GN_BLOCKS_PER_CU': '5', 'DEXCT_GN_CHUNK': '4'}]
if os.environ.get('GN_VARIANTS') == 'chunk':      # pixels per lane of a wave's run (default 64 at this size)
    variants = [{}] + [{'DEXCT_GN_CHUNK': c} for c in os.environ.get('CHUNKS', '8,16,32,128,256').split(',')]
if os.environ.get('GN_VARIANTS') == 'hist':      # history length of the repeated-state exit (x occupancy)
    variants = [{}] + [{'DEXCT_GN_HIST': str(h)} for h in (4, 5, 6, 7, 10, 12)] + \
        [{'DEXCT_GN_HIST': '4', 'DEXCT_GN_MINW': '6'}, {'DEXCT_GN_HIST': '6', 'DEXCT_GN_MINW': '6'}, {'DEXCT_GN_RING': '1'},
         {'DEXCT_GN_MINW': '4'}, {'DEXCT_GN_RING': '1', 'DEXCT_GN_MINW': '4'}]
times = {i: [] for i in range(len(variants))}
same = {}
for rep in range(3):
    for i, env in enumerate(variants):
        times[i].append(run(a, env))
        same[i] = bool(torch.equal(a.view(torch.int64), ref.view(torch.int64)))
for i, env in enumerate(variants):
    print(f'{str(env):70s} ms {" ".join("%.1f" % t for t in times[i])}   bit-identical to default: {same[i]}', flush=True)
