"""HBM traffic of the Newton kernel per launch variant, round 4 (run under rocprofv3 --pmc WRITE_SIZE, then FETCH_SIZE):
    cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -- python3 $REPO/tools/probes/gn_write2.py [lib.so]
    python3 tools/probes/gn_write2.py --read $OUT/w
One launch per variant, in VARIANTS order: (environment, gn_device keywords)."""
import csv
import glob
import os
import sys

VARIANTS = [({}, dict(ref=True)), ({}, dict(ref=False)), ({'DEXCT_GN_MINW': '4'}, dict(ref=True)), ({'DEXCT_GN_MINW': '4'}, dict(ref=False)),
            ({}, dict(ref=True, exact=True)), ({'DEXCT_GN_MINW': '4'}, dict(ref=True, exact=True))]

if '--read' in sys.argv:
    d = sys.argv[sys.argv.index('--read') + 1]
    rows = []
    for f in glob.glob(d + '/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'gn_refill' in r['Kernel_Name']:
                rows.append((int(r['Dispatch_Id']), r['Counter_Name'], float(r['Counter_Value']),
                             int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    rows.sort()
    for (disp, name, val, ns), (env, kw) in zip(rows, VARIANTS):
        print(f'{str(env):28s} {str(kw):32s} {name} {val * 1024 / 1e9:8.3f} GB   {ns / 1e6:7.1f} ms (under the profiler)')
    sys.exit(0)

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from dex_ct_sim_amd import _native
if len(sys.argv) > 1:
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
import dex_ct_sim_amd as dx
from dex_ct_sim_amd import forward_project as fp, matdecomp as md, synthetic

det = os.path.join(ROOT, 'dex-ct-sim_amd/input/detector/eta_eid_mv.bin')
n, views, chans = 512, int(os.environ.get('VIEWS', 250)), 800
ct = dx.FanBeamGeometry(N_channels=chans, N_proj=views, gamma_fan=0.8230337, SID=60.0, SDD=100.0, eid=True, detector_file=det, N_rows=n)
ph = synthetic.make_phantom(n, n, extent=51.2, seed=1234)
specs = [synthetic.kramers_spectrum(140), synthetic.kramers_spectrum(80)]
pj = fp.Projector(ct, ph)
_, mu_d, w_d, _ = pj.upload_tables(specs)
counts = pj.project_tables(mu_d, w_d, layout=None)
_, i0, mus = md.decomposition_tables(ct, specs[0], specs[1])
gmax = counts[0].max().double()
a = torch.empty((views, chans, n, 2), dtype=torch.float64, device=counts.device)
for env, kw in VARIANTS:
    for k in ('DEXCT_GN_MINW',):
        os.environ.pop(k, None)
    os.environ.update(env)
    md.gn_device(counts[0], counts[1], i0, mus, 50, 'f64', out=a, mask_max=gmax, out_rc=(n, chans) if kw.get('ref') else None,
                 stop_tol=0.0 if kw.get('exact') else None)
    torch.cuda.synchronize()
print('result bytes per launch: %.3f GB, input bytes %.3f GB' % (a.numel() * 8 / 1e9, 2 * counts[0].numel() * 4 / 1e9))
