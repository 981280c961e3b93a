"""bench.py as the driver calls it: the plain command with --gpus N must start its own ranks (no launcher), run the
sharded step (view shards projected in chunks, the raw sinograms assembled - gather to rank 0, point-to-point to every rank,
or all-gather -, all-reduced air-mask maximum) and print one JSON line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ['--n', '64', '--views', '48', '--channels', '96', '--iters', '10', '--steps', '1', '--warmup', '1',
         '--no-cpu-baseline', '--skip-single-row', '--skip-gn-full-loop']


def run_bench(*extra, env=None):
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *SMALL, *extra], capture_output=True, text=True,
                       timeout=900, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0]), p.stderr


def test_plain_command_single_gpu(hip):
    out, _ = run_bench()
    assert out['n_gpus'] == 1 and out['scaling'] == 'strong' and out['value'] > 0
    assert out['roofline']['kernel'].startswith('gn_shortcut_kernel') and 'roofline_siddon' in out
    for r in (out['roofline'], out['roofline_siddon']):
        assert r['frac'] is None or 0 < r['frac'] <= 1.0, r
    # round 6: frac = the FP64 flops the launch ISSUES / time / peak; SURVEY 8d's unit (which counts flops the short cut's
    # Gauss-Newton step does not issue) is kept beside it; where a PMC file matches, frac cannot exceed the issue share
    roof = out['roofline']
    assert roof['frac'] == roof['hardware_fp64_utilisation'] and roof['frac'] <= roof['frac_by_survey_unit']
    assert abs(roof['frac'] - roof['hardware_fp64_flop_share'] * roof['frac_by_survey_unit']) < 1e-12
    if 'issue' in roof:
        assert roof['frac'] <= roof['issue']['frac'] <= 1.0
    assert 'n_iters = 10 asked' in out['config']['workload'] and 'full-table steps executed per unmasked pixel' in out['config']['workload']
    ns = out['noisy_step']                       # the step WITH quantum noise (the reference's dose-scaled mode), default kernels
    assert ns['projection_ms']['gaussian'] > 0 and ns['projection_ms']['poisson'] > 0 and ns['ms_per_step'] > 0
    assert ns['gn']['default_max_diff_vs_exact'] <= 1e-12 and 0 < ns['relative_noise_of_the_sample'] < 0.1
    q = out['siddon_reduced_quadrature']         # opt-in shorter energy table: measured beside the step, bound checked on every ray
    assert q['applied'] and 3 * q['nodes'] < q['full_grid_bins'] and q['verified_max_rel_err_f64'] <= 1e-6
    assert q['max_rel_deviation_of_counts_all_rays'] <= 2e-6
    e2e = out['dropin_e2e']                      # the public NumPy boundary, timed (get_sino x 2 + get_basismat_sinos)
    assert len(e2e) == 2
    for case in e2e.values():
        assert case['cold']['ok'] and case['cold_no_disk_cache']['ok'] and case['warm']['ok'] and case['warm']['total_s'] > 0
        assert case['bytes']['d2h_per_get_sino'] == 8 * case['rays']


@pytest.mark.parametrize('scaling,gather', [('strong', None), ('weak', 'direct'), ('strong', 'all'), ('strong', 'root')])
def test_plain_command_two_ranks(hip, scaling, gather):
    """`python bench.py --gpus 2` (no torchrun): on a one-GPU box the two ranks share the device and rehearse over
    gloo; with two devices the same command runs RCCL.  Every mode of the assembly, with the other two measured beside it; the
    default (--gather auto) picks the mode of the timed loop from measured warm-up steps and says which and why; the line also
    carries the step without its fabric part (value_compute_only)."""
    out, err = run_bench('--gpus', '2', '--scaling', scaling, *(('--gather', gather) if gather else ()))
    assert out['n_gpus'] == 2 and out['scaling'] == scaling
    m = out['multi_gpu']
    assert m['gather_flag'] == (gather or 'auto') and m['gather'] in ('root', 'direct', 'all')
    if gather:
        assert m['gather'] == gather and m['gather_choice'] is None
        assert m['view_chunks_per_rank'] == (1 if gather == 'all' else 4)
    else:
        ch = m['gather_choice']
        assert sorted(ch['step_ms']) == ['all', 'direct', 'root'] and ch['why'].startswith(m['gather'])
        assert ch['step_ms'][m['gather']] <= 1.02 * min(ch['step_ms'].values())
    assert m['value_compute_only'] > 0 and m['ms_per_step_compute_only'] > 0 and 0 <= m['fabric_share_of_step'] < 1
    assert m['implied_GBps_into_root'] == m['by_mode']['root']['GBps_into_a_receiving_rank'] > 0
    assert len(m['per_rank']) == 2 and m['gather_ms'] > 0
    assert sorted(m['by_mode']) == ['all', 'direct', 'root']
    for mode, b in m['by_mode'].items():
        assert b['gather_ms'] > 0 and b['gather_exposed_ms'] >= 0 and b['ms_per_step'] > 0 and b['receiving_ranks'] == (1 if mode == 'root' else 2)
    assert m['gather_device_allocations_per_call'] == 0          # preallocated send / receive / result buffers
    views = [r['views'] for r in sorted(m['per_rank'], key=lambda r: r['rank'])]
    total = 48 if scaling == 'strong' else 96
    assert views[0][0] == 0 and views[0][1] == views[1][0] and views[1][1] == total
    assert out['config']['rays_total'] == total * 64 * 96
    one, _ = run_bench('--scaling', scaling)
    # same rays per GPU (weak) or same total (strong): value is whole-job throughput over the same definition
    assert one['config']['rays_total'] == (48 * 64 * 96)


def test_plain_command_four_ranks_ragged(hip):
    """Four ranks (gloo rehearsal on a one-GPU box) over a scan whose views do not divide evenly: 50 = 13+13+12+12."""
    out, _ = run_bench('--gpus', '4', '--views', '50', '--gather-chunks', '5', '--gather', 'root')      # chunks of 3, 3, 3, 2, 2 / 3, 3, 2, 2, 2 views
    assert out['n_gpus'] == 4 and out['multi_gpu']['view_chunks_per_rank'] == 5 and out['config']['rays_total'] == 50 * 64 * 96
    views = [r['views'] for r in sorted(out['multi_gpu']['per_rank'], key=lambda r: r['rank'])]
    assert views == [[0, 13], [13, 26], [26, 38], [38, 50]]
    assert out['multi_gpu']['gather_device_allocations_per_call'] == 0       # the ragged path too
    for r in range(4):                                                        # every rank leaves a log behind
        assert os.path.exists(os.path.join(ROOT, 'gpurun_out', f'rank{r}.log'))


def test_world_size_mismatch_is_an_error(hip):
    e = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *SMALL, '--gpus', '2'], capture_output=True,
                       text=True, timeout=600, env=e)
    assert p.returncode != 0 and 'WORLD_SIZE' in (p.stderr + p.stdout)


def test_plain_command_three_ranks_config3_labels(hip):
    """The configs[3] workload switch with ragged shards over gloo: 52 views = 18 + 17 + 17.  (The box's process guard admits
    6 processes on the card and the test runner is one of them: the 4-rank test above is as far as a rehearsal here should
    go; the 8-rank launch is the driver's, on an 8-GPU node, and the 8-rank gather itself runs on CPU in
    tests/test_shard_gloo.py.)"""
    out, _ = run_bench('--gpus', '3', '--views', '52', '--workload', 'config3')          # --gather auto: three ranks agree on one mode
    assert out['n_gpus'] == 3 and out['config']['rays_total'] == 52 * 64 * 96
    assert out['config']['baseline_config'] == 'configs[3]'
    views = [r['views'] for r in sorted(out['multi_gpu']['per_rank'], key=lambda r: r['rank'])]
    assert views == [[0, 18], [18, 35], [35, 52]]
    assert out['multi_gpu']['gather_device_allocations_per_call'] == 0
    assert out['multi_gpu']['gather_flag'] == 'auto' and out['multi_gpu']['gather'] in out['multi_gpu']['gather_choice']['step_ms']
    assert 'value_exact' not in out and out['value'] > 0          # (the exact-mode comparison runs at N = 1 only)


def test_under_the_drivers_launcher(hip):
    """The driver's launch for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` - bench.py then IS a rank (RANK / LOCAL_RANK / WORLD_SIZE from the launcher) and
    rank 0 prints the one JSON line.  On a one-GPU box the two ranks share the device over gloo (DEXCT_DIST_BACKEND, which the
    plain command sets by itself)."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    e = dict(os.environ, DEXCT_DIST_BACKEND='gloo')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(ROOT, 'bench.py'), *['--phantom-n' if a == '--n' else a for a in SMALL],
                        '--gpus', '2'],
                       capture_output=True, text=True, timeout=900, env=e)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['multi_gpu']['gather_flag'] == 'auto' and out['value'] > 0
    assert len(out['multi_gpu']['per_rank']) == 2
