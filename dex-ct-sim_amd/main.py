#!/usr/bin/env python3
"""Entry point with the reference's run sequence (main.py:74-178 of gjadick/dex-ct-sim):

for every run in the parameter file, for every dual-energy spectrum pair:
  1. per spectrum: load + dose-scale the spectrum (:64-69), forward project (:120), write
     ``<spec>_<dose>uGy/sino_raw_float32.bin`` and ``sino_log_float32.bin`` (:121-122);
  2. decompose the two raw sinograms into basis-material sinograms with 50 Newton iterations
     (:153) and write ``matdecomp_<s1>_<s2>_<d1>uGy_<d2>uGy/mat{1,2}_sino_float32.bin`` (:154-155).
  3. when ``back_project`` is set: reconstruct every log sinogram (:134 -> recon_raw/recon_HU .bin, :135-136)
     and both basis-material sinograms (:168 -> mat{1,2}_recon_float32.bin, :169).

Differences from the reference script, all on purpose: inputs are command-line options instead of
edited source lines (:80-82, :101-103); figures are off unless --show; both spectra of a pair are
projected in ONE traversal (path lengths do not depend on energy); run with
``python -m torch.distributed.run --nproc-per-node N main.py`` to shard the projection angles.
"""
import argparse
import os
import shutil
import sys
from time import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import dex_ct_sim_amd as dx  # noqa: E402
from dex_ct_sim_amd.back_project import get_recon  # noqa: E402
from dex_ct_sim_amd.forward_project import get_sinos  # noqa: E402
from dex_ct_sim_amd.matdecomp import get_basismat_sinos  # noqa: E402


def load_spectrum(ct, spec_id, dose, input_dir):
    """Spectrum at 1 mGy scaled to the target dose per view (main.py:64-69)."""
    spec = dx.xRaySpectrum(os.path.join(input_dir, 'spectrum', f'{spec_id}_1mGy_float32.bin'), spec_id)
    spec.rescale_counts(ct.A_iso * dose / ct.N_proj)
    return spec


def parse_pairs(items):
    pairs = []
    for it in items:
        s1, s2, d1, d2 = it.split(':')
        pairs.append((s1, s2, float(d1), float(d2)))
    return pairs


def show(title_a, a, title_b, b):
    import matplotlib.pyplot as plt
    fig, ax = plt.subplots(1, 2, figsize=[7, 3])
    for axi, img, ttl in ((ax[0], a, title_a), (ax[1], b, title_b)):
        m = axi.imshow(img, cmap='gray', aspect='auto')
        axi.set_title(ttl)
        fig.colorbar(m, ax=axi)
    fig.tight_layout()
    plt.show()


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--params', default=os.path.join(HERE, 'input', 'params.txt'))
    ap.add_argument('--out', default='./output/')
    ap.add_argument('--input-dir', default=os.path.join(HERE, 'input'))
    ap.add_argument('--pairs', nargs='+', default=['detunedMV:80kV:9:1'],
                    help='spec1:spec2:dose1_mGy:dose2_mGy (reference default, main.py:101)')
    ap.add_argument('--n-iters', type=int, default=50)
    ap.add_argument('--gn-exact', action='store_true',
                    help='run the fixed iteration count of matdecomp.py:114 bit for bit (stop_tol = 0) instead of ending a '
                         'pixel once the distance it still has to go is below 1e-12 relative (the default)')
    ap.add_argument('--gn-audit', default='100', metavar='PPM[,strict]',
                    help='sampled audit of the Newton short cut: that many pixels per million are solved again with the fixed '
                         'iteration count in the same call and compared (<= 1e-12, same NaN pattern); a difference warns, or '
                         'raises with ",strict"; 0 = off (the library default; this script switches it on)')
    ap.add_argument('--show', action='store_true')
    ap.add_argument('--noise', default='off', choices=['off', 'gaussian', 'poisson'],
                    help='quantum noise for the dose of each spectrum (default off: the noise-free expectation); '
                         'gaussian: compound-Poisson variance, normal sample; poisson: per-energy-bin photon counts')
    ap.add_argument('--seed', type=int, default=0, help='seed of the counter-based noise generator')
    ap.add_argument('--window', default=None, choices=['rect', 'sinc', 'cosine', 'hann', 'hamming'],
                    help='apodisation of the reconstruction ramp (default: DEXCT_FBP_WINDOW or rect)')
    args = ap.parse_args(argv)

    import torch.distributed as dist
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 and not dist.is_initialized():
        # RCCL; DEXCT_DIST_BACKEND=gloo rehearses the sharded flow on a box with fewer GPUs than ranks
        dist.init_process_group(os.environ.get('DEXCT_DIST_BACKEND', 'nccl'))
    rank = dist.get_rank() if dist.is_initialized() else 0

    # './input/...' paths of the params file resolve against the folder that holds the input directory; the
    # working directory stays where it is (relative --params / --input-dir / --out keep their meaning)
    args.params = os.path.abspath(args.params)
    args.input_dir = os.path.abspath(args.input_dir)
    all_params = dx.read_parameter_file(args.params, base_dir=os.path.dirname(args.input_dir))
    for params in all_params:
        run_id, do_fp, do_bp = params[:3]
        ct, phantom, _ = params[3:6]            # the spectrum entry is ignored, as in main.py:92
        out_dir = os.path.join(args.out, run_id)
        if rank == 0:
            os.makedirs(out_dir, exist_ok=True)
            shutil.copy(args.params, os.path.join(out_dir, 'params.txt'))
        if do_bp:
            N_matrix, FOV, ramp = params[6:9]
        for s1, s2, d1, d2 in parse_pairs(args.pairs):
            t0 = time()
            specs = [load_spectrum(ct, s1, d1, args.input_dir), load_spectrum(ct, s2, d2, args.input_dir)]
            print('Forward projecting!')
            sinos = get_sinos(ct, phantom, specs, noise={'off': False, 'gaussian': True, 'poisson': 'poisson'}[args.noise],
                              seed=args.seed)
            for (spec_id, dose), (sino_raw, sino_log) in zip(((s1, d1), (s2, d2)), sinos):
                sub_dir = os.path.join(out_dir, f'{spec_id}_{int(dose * 1000):04}uGy/')
                if rank == 0:
                    os.makedirs(sub_dir, exist_ok=True)
                    print(f'\n*** {sub_dir} ***')
                    sino_raw.astype(np.float32).tofile(sub_dir + 'sino_raw_float32.bin')
                    sino_log.astype(np.float32).tofile(sub_dir + 'sino_log_float32.bin')
                    if args.show:
                        show('Raw line integrals', sino_raw, 'Log sinogram', sino_log)
                    if do_bp:
                        print('Back projecting!')
                        spec = specs[0] if spec_id == s1 else specs[1]
                        recon_raw, recon_HU = get_recon(sino_log, ct, spec, N_matrix, FOV, ramp, window=args.window)
                        recon_raw.astype(np.float32).tofile(sub_dir + 'recon_raw_float32.bin')
                        recon_HU.astype(np.float32).tofile(sub_dir + 'recon_HU_float32.bin')
                        if args.show:
                            show('Raw reconstruction [1/cm]', recon_raw, 'Hounsfield Units', recon_HU)
            sub_dir = os.path.join(out_dir, f'matdecomp_{s1}_{s2}_{int(d1 * 1000):04}uGy_{int(d2 * 1000):04}uGy/')
            print('Decomposing into basis material sinograms!')
            matsino1, matsino2 = get_basismat_sinos(ct, sinos[0][0], sinos[1][0], specs[0], specs[1],
                                                    n_iters=args.n_iters, verbose=True,     # progress lines of :111-112
                                                    stop_tol=0.0 if args.gn_exact else None,
                                                    audit=float(args.gn_audit.split(',')[0]),
                                                    audit_strict=args.gn_audit.endswith(',strict'))
            if rank == 0:
                os.makedirs(sub_dir, exist_ok=True)
                print(f'\n*** {sub_dir} ***')
                matsino1.astype(np.float32).tofile(sub_dir + 'mat1_sino_float32.bin')
                matsino2.astype(np.float32).tofile(sub_dir + 'mat2_sino_float32.bin')
                if args.show:
                    show('Basis material 1', matsino1, 'Basis material 2', matsino2)
                if do_bp:
                    print('Back projecting basis material sinograms!')
                    for i, matsino in enumerate([matsino1, matsino2]):
                        recon_raw, _ = get_recon(matsino, ct, specs[0], N_matrix, FOV, ramp, window=args.window)    # spec is filler (:168)
                        recon_raw.astype(np.float32).tofile(sub_dir + f'mat{i + 1}_recon_float32.bin')
                print(f'matdecomp finished for {s1}-{s2} : t={time() - t0:.2f}s')


if __name__ == '__main__':
    main()
