"""gpurun_out/prof_<tag>/ of tools/profile_config4.sh -> profiles/<tag>.md: kernel time (rocprofv3 and HIP events), FETCH_SIZE /
WRITE_SIZE of the projection kernel with the dword-per-lane correction calibrated in the same run (pack2_kernel reads the byte
volume once with one dword per lane, the access shape of rows16_kernel), traffic / algorithmic bytes, fraction of the HBM peak."""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
src = os.path.join('gpurun_out', f'prof_{tag}')
newest = lambda pat: sorted(glob.glob(pat), key=os.path.getmtime)[-1]
run = json.loads(open(os.path.join(src, 'run_stats.json')).read().strip().splitlines()[-1])
rows = list(csv.DictReader(open(newest(os.path.join(src, 'stats', '*', '*_kernel_stats.csv')))))


def pmc(which):
    out = {}
    for r in csv.DictReader(open(newest(os.path.join(src, f'pmc_{which}', '*', '*_counter_collection.csv')))):
        out.setdefault(r['Kernel_Name'], []).append(float(r['Counter_Value']) * 1024.0)        # KiB
    return {k: sum(v) / len(v) for k, v in out.items()}


fetch, write = pmc('fetch'), pmc('write')
def is_it(name):           # the projection kernel of the run; of rows16_kernel the NOISY instantiation (last template argument) iff asked
    head = name.split('(')[0].rstrip()
    return (run['kernel'] + '<') in head and (run['kernel'] != 'rows16_kernel' or head.endswith(', true>') == bool(run.get('noisy')))


kern = [k for k in fetch if is_it(k)][0]
k_ms = [float(r['AverageNs']) / 1e6 for r in rows if is_it(r['Name'])][0]
pack = [k for k in fetch if 'pack2_kernel' in k and 'groups' not in k]
corr = (float(run.get('n', 1024)) ** 3 / fetch[pack[0]]) if pack else 2.0
f_b, w_b = corr * fetch[kern], write[kern]
alg = run['algorithmic_bytes_per_launch']
out = [f'# {run["workload"].split(":")[0]} under rocprofv3 (`{tag}`; tools/profile_config4.sh)', '',
       run['workload'] + f': {run["rays"]:.4g} rays x {run["weighted_bins"]} weighted bins per launch; the 2-bit packed volume is '
       f'{run["packed_volume_MiB"]:.0f} MiB (Infinity Cache: 256 MiB).', '',
       '| kernel | calls | avg ms | % of GPU time |', '|---|---|---|---|']
out += [f'| `{r["Name"][:70]}` | {r["Calls"]} | {float(r["AverageNs"]) / 1e6:.3f} | {r["Percentage"]} |' for r in rows[:6]]
out += ['', f'`{run["kernel"]}`: {k_ms:.3f} ms per launch under rocprofv3, {run["projection_ms"]:.3f} ms by HIP events around the call (same run, '
        f'{run["reps"]} launches): {run["rays"] / (k_ms * 1e-3):.3g} rays/s, {run["rays"] * run["weighted_bins"] / (k_ms * 1e-3):.3g} ray-energy integrals/s.', '',
        '| | bytes per launch | GB/s at the kernel\'s time | of the 8 TB/s HBM peak |', '|---|---|---|---|',
        f'| algorithmic (SURVEY 8d: segments x {run["bytes_per_stored_voxel"]} B + 8 B out per ray and spectrum) | {alg / 1e9:.2f} GB | {alg / (k_ms * 1e-3) / 1e9:.0f} | {alg / (k_ms * 1e-3) / 8e12:.3f} |',
        f'| FETCH_SIZE (x {corr:.2f}: dword-per-lane loads, calibrated on pack2_kernel in this run) | {f_b / 1e9:.2f} GB | {f_b / (k_ms * 1e-3) / 1e9:.0f} | {f_b / (k_ms * 1e-3) / 8e12:.3f} |',
        f'| WRITE_SIZE | {w_b / 1e9:.2f} GB | {w_b / (k_ms * 1e-3) / 1e9:.0f} | {w_b / (k_ms * 1e-3) / 8e12:.3f} |',
        f'| FETCH + WRITE | {(f_b + w_b) / 1e9:.2f} GB | {(f_b + w_b) / (k_ms * 1e-3) / 1e9:.0f} | **{(f_b + w_b) / (k_ms * 1e-3) / 8e12:.3f}** |', '',
        f'traffic / algorithmic bytes = **{(f_b + w_b) / alg:.3f}** (fetched {f_b / (alg - 8 * run.get("spectra", 1) * run["rays"]):.3f} x the algorithmic voxel bytes, written '
        f'{w_b / (8 * run.get("spectra", 1) * run["rays"]):.3f} x the outputs).']
open(os.path.join('profiles', f'{tag}.md'), 'w').write('\n'.join(out) + '\n')
json.dump({'run': run, 'kernel_ms_rocprof': k_ms, 'fetch_bytes_corrected': f_b, 'fetch_correction': corr, 'write_bytes': w_b,
           'traffic_over_algorithmic': (f_b + w_b) / alg, 'frac_of_hbm_peak': (f_b + w_b) / (k_ms * 1e-3) / 8e12},
          open(os.path.join('profiles', f'{tag}.json'), 'w'), indent=1)
print('\n'.join(out))
