#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (written by tools/profile_gpu.sh) into
profiles/<tag>_kernel_stats.csv, profiles/<tag>_summary.md and profiles/<tag>_pmc_traffic.json.

HBM traffic follows MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE come from separate
--pmc passes, are reported in KiB, and FETCH_SIZE is calibrated on a kernel of known byte count with
the same access shape: transpose_xy_kernel reads the uint8 volume once with 64-B wave loads, exactly
the shape of the traversal kernels' loads (ratio printed below; 1.00 means no correction; wide 4-B+
per-lane streams such as gn_kernel's input read 0.5 and are doubled).
"""
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dex_ct_sim_amd.quadrature import GATE_CELLS      # (the size of the short cut's table: what the step-counting launches ran on)

tag = sys.argv[1]
src = os.path.join('gpurun_out', f'prof_{tag}')
os.makedirs('profiles', exist_ok=True)
def newest(pattern):
    """the most recent match (gpurun merges a call's files into gpurun_out/ beside those of earlier calls with the same tag)"""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:] if files else []


stats = newest(os.path.join(src, 'stats', '*', '*_kernel_stats.csv'))[0]
shutil.copy(stats, os.path.join('profiles', f'{tag}_kernel_stats.csv'))
rows = list(csv.DictReader(open(stats)))


def pmc(which):
    out = {}
    files = newest(os.path.join(src, f'pmc_{which}', '*', '*_counter_collection.csv'))
    if not files:
        return out
    for r in csv.DictReader(open(files[0])):
        out.setdefault(r['Kernel_Name'], []).append(float(r['Counter_Value']) * 1024.0)
    # (the step-counting instantiation of the Newton kernel is also dispatched once on the 16 641 pixels of the gate calibration:
    # the large dispatches are the ones of the steps)
    return {k: (max(v) if 'gn_refill_kernel<true>' in k else sum(v) / len(v)) for k, v in out.items()}


fetch, write = pmc('fetch'), pmc('write')


def pmc_raw(which):
    """{kernel: {counter: mean value per dispatch}} of a multi-counter pass (no unit scaling)."""
    out = {}
    files = newest(os.path.join(src, f'pmc_{which}', '*', '*_counter_collection.csv'))
    if not files:
        return out
    for r in csv.DictReader(open(files[0])):
        out.setdefault(r['Kernel_Name'], {}).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in out.items()}


sq = pmc_raw('sq')
bench = json.load(open(os.path.join(src, 'bench_stats.json')))
lines = [f'# rocprofv3 summary `{tag}`', '',
         f'command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps {bench["steps"]} --warmup {bench["warmup"]} --skip-gn-full-loop`'
         ' (PMC passes: `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` separately, 1 step)', '',
         '| kernel | calls | avg ms | % of GPU time |', '|---|---|---|---|']
for r in rows[:8]:
    lines.append(f'| `{r["Name"][:70]}` | {r["Calls"]} | {float(r["AverageNs"]) / 1e6:.3f} | {r["Percentage"]} |')
lines += ['', f'bench.py (same run) HIP-event averages: {bench["kernel_ms"]}', '']
tl = (bench.get('roofline') or {}).get('short_cut')
if tl:
    trace = os.path.join(os.path.dirname(stats), os.path.basename(stats).replace('kernel_stats', 'kernel_trace'))
    big = {}
    for r in csv.DictReader(open(trace)):
        for pat in ('gn_refill_kernel<true>', 'gn_shortcut_kernel'):
            if pat in r['Kernel_Name']:
                big.setdefault(pat, []).append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e6)
    for pat, v in big.items():
        if pat.endswith('<true>'):
            lines.append(f'`{pat}` (the step-counting instantiation): {len(v)} dispatches averaging {sum(v) / len(v):.2f} ms - the reference\'s walk on '
                         f'the gate\'s {(GATE_CELLS + 1) ** 2} cell corners and {GATE_CELLS ** 2} cell centres, once per pair of spectra '
                         f'(then from DEXCT_CACHE_DIR)')
            continue
        step = [x for x in v if x > 0.25 * max(v)]
        lines.append(f'`{pat}`: {len(v)} dispatches, of which {len(step)} are step launches averaging {sum(step) / len(step):.2f} ms'
                     + (' (the others: the gate calibration, once per pair of spectra)' if len(step) < len(v) else ''))
    lines += [f'Newton launch in bench.py (HIP events around it, last timed step; mode {tl["mode"]}): {tl["launch_ms"]:.2f} ms', '']
cal = [k for k in fetch if 'transpose_xy' in k]
traffic = {}
# the traversal kernel of the timed step: the one bench.py names; of rows16_kernel the noise-free instantiation (its NOISY twin -
# last template argument true - runs in bench.py's noisy_step and is recorded under other_kernels)
step_kernel = (bench.get('roofline_siddon') or {}).get('kernel', 'rows')


def is_step_traversal(k):
    name = k.split('(')[0]
    if 'cone_' in name or (step_kernel + '<') not in name:
        return False
    return not (step_kernel == 'rows16_kernel' and name.rstrip().endswith(', true>'))


if cal:
    # 512^3 = known bytes read by the in-plane transpose
    n = bench['config']['rays_per_gpu']
    lines.append('## HBM-side traffic (PMC)')
    vol_bytes = None
    for k in fetch:
        if 'transpose_xy' in k:
            vol_bytes = fetch[k]
    lines.append(f'calibration: transpose_xy_kernel (reads the {bench["config"]["workload"].split(" ")[0]} uint8 volume '
                 f'once, 64-B wave loads like the traversal kernels) FETCH_SIZE = {vol_bytes / 2**20:.2f} MiB')
    lines += ['', '| kernel | FETCH_SIZE GB | WRITE_SIZE GB |', '|---|---|---|']
    for k in fetch:
        if 'dexct' in k:
            lines.append(f'| `{k[:60]}` | {fetch[k] / 1e9:.3f} | {write.get(k, 0) / 1e9:.3f} |')
    for k in fetch:
        if is_step_traversal(k) and 'siddon_kernel' not in traffic:
            traffic['siddon_kernel'] = k
            # 4-B-per-lane dword loads (rows4) are tallied at half, like gn_kernel's float32 input stream
            corr = 2.0 if ('rows4' in k or 'rows16' in k) else 1.0
            traffic['siddon_fetch_correction'] = corr
            traffic['siddon_fetch_bytes'] = corr * fetch[k]
            traffic['siddon_write_bytes'] = write.get(k, 0.0)
            traffic['siddon_hbm_bytes_per_launch'] = corr * fetch[k] + write.get(k, 0.0)
        # the Newton kernel of the step: the short cut (gn_shortcut_kernel) where it ran, else the single launch
        # (gn_refill_kernel<false>); the step-counting launches of the gate calibration (<true>) are recorded beside it
        is_main = ('gn_shortcut_kernel' in k) or (('gn_refill_kernel' in k or 'gn_kernel<false' in k) and 'gn_refill_kernel<true>' not in k
                                                  and not any('gn_shortcut_kernel' in kk for kk in fetch))
        if is_main:
            traffic['gn_kernel'] = k.split('(')[0].replace('void dexct::', '')
            traffic['gn_fetch_bytes_x2_corrected'] = 2 * fetch[k]
            traffic['gn_write_bytes'] = write.get(k, 0.0)
            # round 4: in the reference-order mode (what bench.py runs on stacked fans) the kernel reads its pixels as 64-byte
            # runs, which FETCH_SIZE counts in full (tools/probes/gn_write2.py: 0.77 GB for 0.82 GB of input; the plain order's
            # dword-per-lane stream shows half): no doubling for the round-4 kernel
            traffic['gn_fetch_bytes_raw'] = fetch[k]
            traffic['gn_fetch_counted_in_full'] = 'gn_refill_kernel' in k or 'gn_shortcut_kernel' in k
        if 'gn_refill_kernel<true>' in k:
            traffic['gn_count_fetch_bytes_raw'] = fetch[k]
            traffic['gn_count_write_bytes'] = write.get(k, 0.0)
    for k, d in sq.items():
        if is_step_traversal(k) or 'gn_refill_kernel' in k or 'gn_shortcut_kernel' in k:
            tag2 = 'siddon' if is_step_traversal(k) else ('gn_count' if 'gn_refill_kernel<true>' in k else 'gn')
            traffic[f'{tag2}_valu_insts'] = d.get('SQ_INSTS_VALU')
            if d.get('GRBM_GUI_ACTIVE'):
                # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the chip's waves; GRBM_GUI_ACTIVE sums the 8 XCDs
                traffic[f'{tag2}_valu_busy'] = d.get('SQ_ACTIVE_INST_VALU', 0) * 4.0 / (d['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0)
                traffic[f'{tag2}_wait_any_share'] = (d.get('SQ_WAIT_ANY', 0) / d['SQ_WAVE_CYCLES']) if d.get('SQ_WAVE_CYCLES') else None
    # the other kernels of the bench line (single-row scan, cone beam): raw counters per launch.  FETCH_SIZE is NOT
    # corrected for these access shapes (byte loads per lane: uncalibrated, MI355X_MICROARCH.md section HBM)
    sq2 = pmc_raw('sq2')
    extra = {}
    for key, pat in (('single_row', 'rays_kernel'), ('wave_per_ray', 'wave_ray_kernel'), ('cone_rows', 'cone_rows_kernel'),
                     ('cone_rows', 'cone_cols_kernel'), ('cone_thread_per_ray', 'cone_kernel'),      # cone_rows = the row-parallel kernel the host picks (round 3: cone_cols_kernel)
                     ('noisy', 'rows16_kernel'), ('noisy_round5_path', 'rows4_kernel'), ('add_noise', 'add_noise_kernel'),
                     ('transpose_log', 'transpose_log_kernel')):
        for k in fetch:
            if key == 'noisy' and not k.split('(')[0].rstrip().endswith(', true>'):
                continue                               # (the NOISY instantiation of rows16_kernel only)
            if (pat + '<' in k or (pat + '(') in k) and 'layout' not in k:
                d, d2 = sq.get(k, {}), sq2.get(k, {})
                e = {'kernel': k.split('(')[0].replace('void dexct::', ''), 'fetch_bytes_raw': fetch[k], 'write_bytes': write.get(k, 0.0),
                     'valu_insts': d.get('SQ_INSTS_VALU'), 'vmem_rd_insts': d2.get('SQ_INSTS_VMEM_RD'),
                     'salu_insts': d2.get('SQ_INSTS_SALU'), 'lds_insts': d2.get('SQ_INSTS_LDS'), 'waves': d2.get('SQ_WAVES')}
                if d.get('GRBM_GUI_ACTIVE'):
                    e['valu_busy'] = d.get('SQ_ACTIVE_INST_VALU', 0) * 4.0 / (d['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0)
                    e['wait_any_share'] = (d.get('SQ_WAIT_ANY', 0) / d['SQ_WAVE_CYCLES']) if d.get('SQ_WAVE_CYCLES') else None
                extra[key] = e
    for r in rows:
        for key, e in extra.items():
            if r['Name'].split('(')[0].replace('void dexct::', '') == e['kernel']:
                e['avg_ms_kernel_trace'] = float(r['AverageNs']) / 1e6
    traffic['other_kernels'] = extra
    # calibration of the dword-per-lane correction IN THIS RUN: pack2_kernel reads the byte volume once with one dword per
    # lane (256 contiguous bytes per wave instruction) - the access shape of rows16_kernel / rows4_kernel
    for k in fetch:
        if 'pack2_kernel' in k and 'groups' not in k:
            nvox = float(bench['config'].get('n', 512)) ** 3
            traffic['dword_load_calibration'] = {'kernel': 'pack2_kernel', 'bytes_read': nvox, 'FETCH_SIZE': fetch[k],
                                                 'ratio': fetch[k] / nvox}
            lines += ['', f'dword-per-lane calibration: pack2_kernel reads {nvox / 1e9:.3f} GB, FETCH_SIZE reports '
                          f'{fetch[k] / 1e9:.3f} GB (ratio {fetch[k] / nvox:.3f}): such loads are doubled (`siddon_fetch_correction`)']
    if 'siddon_hbm_bytes_per_launch' in traffic:
        ms = bench['kernel_ms']['siddon_project']
        tb = traffic['siddon_hbm_bytes_per_launch'] / (ms * 1e-3) / 1e12
        alg = bench.get('roofline_siddon', {}).get('algorithmic_bytes_per_launch')
        traffic['siddon_traffic_TBps'] = tb
        traffic['siddon_traffic_frac_of_hbm_peak'] = tb / 8.0
        lines += ['', f'traversal kernel: {ms:.2f} ms, fabric traffic {traffic["siddon_hbm_bytes_per_launch"] / 1e9:.2f} GB per launch '
                      f'(fetch x {traffic["siddon_fetch_correction"]:.0f} + write) = {tb:.2f} TB/s = {tb / 8.0:.2f} of the 8 TB/s HBM peak'
                      + (f'; algorithmic bytes {alg / 1e9:.2f} GB = {alg / (ms * 1e-3) / 1e12:.2f} TB/s' if alg else '')
                      + f'; vector pipe {100 * (traffic.get("siddon_valu_busy") or 0):.0f} % busy, waves waiting '
                        f'{100 * (traffic.get("siddon_wait_any_share") or 0):.0f} %']
    traffic['transpose_xy_fetch_bytes'] = vol_bytes
    traffic['rays_per_gpu'] = bench['config']['rays_per_gpu']
    traffic['n'] = bench['config'].get('n', 512)
json.dump(traffic, open(os.path.join('profiles', f'{tag}_pmc_traffic.json'), 'w'), indent=1)
json.dump(bench, open(os.path.join('profiles', f'{tag}_bench.json'), 'w'), indent=1)
open(os.path.join('profiles', f'{tag}_summary.md'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
