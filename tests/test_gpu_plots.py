"""ROI / VMI measurements of the reference's analysis script (plots.py:136-158, 297-303, 386-395) on the GPU,
checked against the reference's own way of computing them (one make_vmi + NumPy mean/var per energy), restated
here in NumPy."""
import numpy as np
import pytest

from conftest import small_scan

pytestmark = pytest.mark.gpu


def _vmi_numpy(E0, M1, M2, HU=True):
    from dex_ct_sim_amd import matdecomp as md, xcompy
    E = np.array([float(E0)])
    u1, u2 = xcompy.mixatten(md.matcomp1, E), xcompy.mixatten(md.matcomp2, E)
    uw = 1.0 * xcompy.mixatten('H(11.2)O(88.8)', E)
    vmi = u1 * M1 + u2 * M2
    if HU:
        vmi = 1000 * (vmi - uw) / uw
    return vmi.astype(np.float32)


def _roi_numpy(M, roi_info):
    x0, y0, dx, dy = roi_info
    mask = np.zeros(M.shape, dtype=bool)
    mask[y0:y0 + dy, x0:x0 + dx] = 1
    roi = M[mask].astype(np.float64)
    return np.mean(roi), np.var(roi)


def test_label_moments_match_numpy(hip):
    from dex_ct_sim_amd import plots
    rng = np.random.default_rng(5)
    for shape, n_labels in (((37, 53), 3), ((512, 512), 7), ((1, 1), 1), ((300, 4097), 64)):
        a = rng.normal(1.0, 0.3, shape).astype(np.float32)
        b = rng.normal(0.2, 0.1, shape).astype(np.float32)
        lab = rng.integers(0, n_labels + 2, shape).astype(np.uint8)       # labels >= n_labels are skipped
        lab[: shape[0] // 2] = lab[0, 0] % n_labels                        # a large uniform region (register path)
        got = plots.label_moments(a, b, lab, n_labels)
        a64, b64 = a.astype(np.float64), b.astype(np.float64)
        for l in range(n_labels):
            m = lab == l
            ref = [m.sum(), a64[m].sum(), b64[m].sum(), (a64[m] ** 2).sum(), (a64[m] * b64[m]).sum(), (b64[m] ** 2).sum()]
            assert got[l, 0] == ref[0]
            assert np.allclose(got[l], ref, rtol=1e-12, atol=1e-12)
    one = plots.label_moments(a)                      # no labels, no second image
    assert one.shape == (1, 6) and one[0, 0] == a.size and one[0, 2] == 0 and one[0, 5] == 0
    assert np.isclose(one[0, 1], a.astype(np.float64).sum(), rtol=1e-13)


def test_measure_roi_matches_numpy(hip):
    from dex_ct_sim_amd import plots
    rng = np.random.default_rng(6)
    M = rng.normal(40.0, 12.0, (200, 240)).astype(np.float32)
    for roi in ([25, 30, 25, 25], [0, 0, 240, 200], [230, 190, 30, 30], [5, 7, 1, 1]):     # incl. clipped at the border
        u, v = plots.measure_roi(M, roi)
        ur, vr = _roi_numpy(M, roi)
        assert abs(u - ur) < 1e-10 * abs(ur) and abs(v - vr) <= 1e-9 * max(vr, 1e-12)
        assert np.array_equal(np.sort(plots.measure_roi(M, roi, give_roi=True)),
                              np.sort(M[roi[1]:roi[1] + roi[3], roi[0]:roi[0] + roi[2]].ravel()))
    u, v = plots.measure_roi(M, [500, 500, 10, 10])
    assert np.isnan(u) and np.isnan(v)


def test_vmi_sweeps_match_per_energy_loop(hip):
    """Closed-form sweeps against the reference's loop: one VMI (rounded to float32) per energy, then NumPy."""
    from dex_ct_sim_amd import plots, xcompy
    rng = np.random.default_rng(7)
    n = 96
    ids = np.zeros((n, n), dtype=np.uint8)
    yy, xx = np.mgrid[:n, :n]
    ids[(yy - 48) ** 2 + (xx - 48) ** 2 < 40 ** 2] = 1
    ids[(yy - 40) ** 2 + (xx - 60) ** 2 < 8 ** 2] = 2
    M1 = (np.where(ids == 1, 1.0, 0.0) + np.where(ids == 2, 0.4, 0.0) + rng.normal(0, 0.03, (n, n))).astype(np.float32)
    M2 = (np.where(ids == 2, 1.1, 0.0) + rng.normal(0, 0.02, (n, n))).astype(np.float32)
    Evals = np.arange(40, 141, 10)
    sig, bg = [54, 34, 10, 10], [30, 60, 12, 12]
    for HU in (True, False):
        sw = plots.vmi_roi_sweep(Evals, M1, M2, sig, bg, HU=HU)
        for k, E0 in enumerate(Evals):
            vmi = _vmi_numpy(E0, M1, M2, HU)
            u1, v1 = _roi_numpy(vmi, sig)
            u2, v2 = _roi_numpy(vmi, bg)
            tol = 2e-6 * max(abs(u1), abs(u2), 1.0)          # float32 rounding of the reference's VMI
            assert abs(sw['u_signal'][k] - u1) < tol and abs(sw['u_background'][k] - u2) < tol
            assert abs(sw['v_signal'][k] - v1) < 1e-4 * v1 and abs(sw['v_background'][k] - v2) < 1e-4 * v2
            assert abs(sw['cnr'][k] - (u1 - u2) / np.sqrt(v1 + v2)) < 1e-4 * abs(sw['cnr'][k])
        # ground truth: piecewise constant by region id; mask = non-air
        gt = np.stack([np.zeros(len(Evals)), 1.0 * xcompy.mixatten('H(11.2)O(88.8)', Evals.astype(float)),
                       1.92 * xcompy.mixatten('H(3.4)C(15.5)N(4.2)O(43.5)Na(0.1)Mg(0.2)P(10.3)S(0.3)Ca(22.5)',
                                              Evals.astype(float))])
        mask = ids > 0
        rm = plots.vmi_rmse_sweep(Evals, M1, M2, ids, gt, mask=mask, HU=HU)
        for k, E0 in enumerate(Evals):
            vmi = _vmi_numpy(E0, M1, M2, HU).astype(np.float64)
            truth = gt[:, k][ids]
            if HU:
                uw = xcompy.mixatten('H(11.2)O(88.8)', np.array([float(E0)]))[0]
                truth = 1000 * (truth - uw) / uw
            ref = np.sqrt(np.mean((vmi[mask] - truth[mask]) ** 2))
            assert abs(rm[k] - ref) < 1e-5 * ref


def test_image_quality_of_the_whole_chain(hip):
    """main.py's sequence on a small phantom (two spectra -> decomposition -> two reconstructions), then the
    analysis of plots.py: the 70 keV VMI reproduces the phantom's attenuation map, bone stands out of water."""
    import dex_ct_sim_amd as dx
    from dex_ct_sim_amd import plots, synthetic, xcompy
    from scipy import ndimage
    ct, ph = small_scan(n=128, n_views=360, n_channels=300)
    specs = [synthetic.kramers_spectrum(kv) for kv in (140, 80)]
    for s in specs:
        s.rescale_counts(1e6)
    raws = [dx.get_sino(ct, ph, s)[0] for s in specs]
    m1, m2 = dx.get_basismat_sinos(ct, raws[0], raws[1], specs[0], specs[1], n_iters=30)
    M1 = dx.get_recon(m1, ct, specs[0], 128, 51.2, 1.0)[0]
    M2 = dx.get_recon(m2, ct, specs[0], 128, 51.2, 1.0)[0]
    ids = ph.volume[ph.z_index]
    Evals = np.arange(50, 121, 10)
    # interior of each region only: the edges carry the reconstruction's blur, not the decomposition's error
    core = np.zeros_like(ids, dtype=bool)
    for m in (1, 2):
        core |= ndimage.binary_erosion(ids == m, iterations=3)
    rmse = plots.vmi_rmse_sweep(Evals, M1, M2, ids, ph.mu_table(Evals.astype(float)), mask=core, HU=True)
    assert rmse.shape == Evals.shape and np.all(np.isfinite(rmse))
    assert rmse[2] < 60.0, rmse                      # 70 keV, HU
    # yardstick: a mono-energetic 70 keV scan of the same phantom through the same reconstruction.  Its error is the
    # aliasing of the voxelised phantom's staircase edges (the FBP alone is flat to 1e-4 on analytic data,
    # test_fbp_oracle); projection + decomposition + VMI synthesis must not add to it.
    water_core = ndimage.binary_erosion(ids == 1, iterations=3) & ~ndimage.binary_dilation(ids == 2, iterations=6)
    rmse_w = plots.vmi_rmse_sweep(Evals, M1, M2, ids, ph.mu_table(Evals.astype(float)), mask=water_core, HU=True)
    mono = dx.xRaySpectrum.from_arrays('mono70', [70.0], [1.0e6])
    img70 = dx.get_recon(dx.get_sino(ct, ph, mono)[1], ct, mono, 128, 51.2, 1.0)[0]
    truth70 = ph.M_mono(70.0)
    uw = xcompy.mixatten(plots.WATER, np.array([70.0]))[0]
    rmse_mono = 1000.0 / uw * np.sqrt(np.mean((img70[water_core].astype(np.float64) - truth70[water_core]) ** 2))
    assert rmse_w[2] < 1.15 * rmse_mono + 1.0, (rmse_w, rmse_mono)
    vmi = plots.make_vmi(70.0, M1, M2, HU=False)
    truth = ph.M_mono(70.0)
    water = ndimage.binary_erosion(ids == 1, iterations=6)
    assert abs(vmi[water].mean() - truth[water].mean()) < 0.02 * truth[water].mean()
    comp, n_comp = ndimage.label(ndimage.binary_erosion(ids == 2, iterations=2))
    if n_comp:
        sizes = ndimage.sum(np.ones_like(comp), comp, index=np.arange(1, n_comp + 1))
        ys, xs = np.nonzero(comp == 1 + int(np.argmax(sizes)))           # the largest bone insert
        y0, x0 = int(ys.mean()) - 1, int(xs.mean()) - 1
        wy, wx = np.nonzero(water)
        k = len(wy) // 2
        sw = plots.vmi_roi_sweep(Evals, M1, M2, [x0, y0, 3, 3], [int(wx[k]) - 3, int(wy[k]) - 3, 6, 6])
        assert np.all(sw['u_signal'] > sw['u_background'] + 200)      # bone well above water in HU at every energy


def test_example_script_runs(hip):
    """examples/dual_energy_vmi.py (the reference's whole workflow on one page) runs and reports a sane RMSE curve."""
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('dual_energy_vmi', os.path.join(root, 'examples', 'dual_energy_vmi.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv, sys.argv = sys.argv, ['dual_energy_vmi.py', '--n', '96', '--views', '180', '--channels', '160']
    try:
        rmse = mod.main()
    finally:
        sys.argv = argv
    assert rmse.shape == (11,) and np.all(np.isfinite(rmse)) and rmse.min() < 400.0


def test_vmi_and_roi_against_the_reference_functions(hip):
    """make_vmi / measure_roi of the REAL reference (plots.py:136-158; the two definitions compiled from its syntax
    tree by tests/golden/make_goldens_r2.py) against dexct_vmi / dexct_label_moments."""
    import os
    from conftest import GOLDEN
    from dex_ct_sim_amd import plots
    e = np.load(os.path.join(GOLDEN, 'ref_extra.npz'))
    M1, M2 = e['vmi_M1'], e['vmi_M2']
    for k, E0 in enumerate(e['vmi_E']):
        for key, hu in (('hu', True), ('raw', False)):
            got = plots.make_vmi(float(E0), M1, M2, HU=hu)
            ref = e[f'vmi_{key}_{k}']
            assert got.dtype == np.float32 and got.shape == ref.shape
            assert np.array_equal(got, ref), (E0, key, np.abs(got - ref).max())      # same float64 ops, one rounding
    img = e['roi_img']
    for roi, (u_ref, v_ref) in zip(e['roi_info'], e['roi_mean_var']):
        u, v = plots.measure_roi(img, [int(x) for x in roi])
        # the reference's np.mean / np.var run in float32 (pairwise sums); the kernel sums in float64
        assert abs(u - u_ref) <= 2e-6 * max(abs(u_ref), 1.0), (roi, u, u_ref)
        assert abs(v - v_ref) <= 1e-5 * max(v_ref, 1e-12), (roi, v, v_ref)
    px = plots.measure_roi(img, [int(x) for x in e['roi_info'][2]], give_roi=True)
    assert np.array_equal(px, e['roi_pixels_2'])
