"""Forward projection: drop-in for the reference's ``xtomosim.forward_project.get_sino``.

``get_sino(ct, phantom, spec) -> (sino_raw, sino_log)`` is called at main.py:120 of the reference;
its source (x-tomo-sim) is not in the reference checkout, so the contract is reconstructed from the
callers: two ``[N_proj, N_channels]`` arrays, ``sino_raw`` in detected counts (matdecomp.py:30,179)
and ``sino_log`` the log-normalised sinogram fed to reconstruction (main.py:134).  Detection uses
the weighting that the reference's decomposition assumes (matdecomp.py:146-150):
``counts = sum_E I0(E) eta(E) [E if EID] dE exp(-sum_m mu_m(E) L_m)``, noise-free.

All work runs in the HIP library (dexct_fan_plan, dexct_volume_layouts, dexct_siddon_project).
"""
import ctypes as C
import os as _os

import numpy as np
import torch

from . import _native, _shard
from ._device import LazyPinnedResult, device, pinned_empty, pool_wanted, ptr, side_streams, stream_ptr, to_dev


def effective_weights(ct, spec):
    """I0 * detector response * dE on the spectrum's own grid (matdecomp.py:142,146-150)."""
    return spec.I0 * ct.detector_response(spec.E) * spec.bin_widths()


def merged_tables(ct, phantom, specs, with_variance=False):
    """One energy grid for a fused multi-spectrum traversal.

    Each spectrum keeps its own quadrature: on the merged grid its weight is non-zero only at its
    own energies (no interpolation), bins that no spectrum weights are dropped.
    Returns E [nE], mu [M, nE] (float64, 1/cm), w [S, nE] (float64) and, with ``with_variance``,
    w2 = w * (detector signal per photon: E for an energy-integrating detector, 1 for a counting one).
    """
    E = np.unique(np.concatenate([s.E for s in specs]))
    w = np.zeros((len(specs), E.size))
    for k, s in enumerate(specs):
        w[k, np.searchsorted(E, s.E)] = effective_weights(ct, s)
    keep = np.any(w != 0.0, axis=0)
    E, w = E[keep], w[:, keep]
    if with_variance:
        return E, phantom.mu_table(E), w, w * (E if ct.eid else 1.0)
    return E, phantom.mu_table(E), w


def compact_ids(present, materials):
    """Which table rows a projection needs, and the renumbering of the ids: ``present[i]`` says whether id i occurs in the
    scanned voxels.  Ids with the same (density, composition) share a row (their rows of density x mixatten(E) are equal at
    every energy), absent ids get none, row 0 stays id 0 (its path length comes from the chord); at least two rows are kept
    when the table has them (the kernels' smallest table).  Returns (mat_rows, lut): mat_rows[k] = the id whose table row
    serves compact id k; lut[i] = compact id of id i (256 uint8 values; 0 for absent ids)."""
    mat_rows, lut, seen = [0], np.zeros(256, dtype=np.uint8), {}
    key = lambda m: (float(m.density), str(m.matcomp))
    seen[key(materials[0])] = 0
    for i in range(1, len(materials)):
        if present[i]:
            k = key(materials[i])
            if k not in seen:
                seen[k] = len(mat_rows)
                mat_rows.append(i)
            lut[i] = seen[k]
    if len(mat_rows) < 2 <= len(materials):
        mat_rows.append(1)
    return mat_rows, lut


class Projector:
    """Device-resident state of one (scanner, phantom) pair: volume layouts and ray plans.

    ``view_range`` restricts the instance to a contiguous shard of projection angles (one per
    rank in a multi-GPU run).  ``kernel``: 0 choose, 1 ray-parallel, 2 row-parallel (1 row per lane),
    3 row-parallel with 4 rows per lane (<= 4 materials, Nz and z_index multiples of 4), 4 the same kernel
    run once per group of three materials (5..256 materials), 5 the 4-rows-per-lane kernel with a tile of
    neighbouring (view, channel) pairs per workgroup walking the volume in step (rows4t_kernel: opt-in, kept for A/B
    runs - kernel 0 never picks it; same bits as kernel 3), 6 one wavefront per ray (lanes over
    dominant-axis slabs, shuffle reductions, tables in LDS: the mapping BASELINE.json's north star names; <= 4
    materials), 7 the stacked fan on a 2-bit packed volume with bit-sliced counters (rows16_kernel: 16 rows per
    lane; <= 4 materials; what kernel 0 picks from 192 rows on when a pair fills 3/4 of its lane group), 8 the same
    kernel run once per group of three materials on packed group codes (2..256 materials; what kernel 0 picks for more
    than 4 materials under the same conditions).
    """

    def __init__(self, ct, phantom, view_range=None, kernel=0, dev=None):
        self.lib = _native.load()
        self.dev = dev or device()
        self.ct, self.phantom = ct, phantom
        self.kernel = kernel
        vb, ve = view_range if view_range is not None else (0, ct.N_proj)
        if not (0 <= vb < ve <= ct.N_proj):
            raise ValueError(f'bad view range {view_range}')
        self.view_begin, self.view_end = int(vb), int(ve)
        self._bounds, self.quadrature_info = None, None
        self.cone = bool(getattr(ct, 'cone', False))
        z_first = 0 if self.cone else phantom.z_index
        if not self.cone and (z_first < 0 or z_first + ct.N_rows > phantom.Nz):
            raise ValueError(f'rows {ct.N_rows} from slice {z_first} do not fit Nz={phantom.Nz}')
        half_diag = 0.5 * np.hypot(phantom.Nx * phantom.dx, phantom.Ny * phantom.dy)
        if ct.SID <= half_diag or ct.SDD - ct.SID < half_diag:
            # line integrals run through the whole grid: a source or a detector inside it would silently see
            # material "behind" itself
            raise ValueError('source and detector must lie outside the phantom grid: SID > half diagonal and '
                             'SDD - SID >= half diagonal')
        volume, nz = phantom.volume, phantom.Nz
        if not self.cone and ct.N_rows < nz:
            # a stacked fan only ever reads its own slices: upload those (one slice of a 512^3 phantom for the
            # reference's single-row scan instead of 128 MiB)
            volume, nz, z_first = volume[z_first:z_first + ct.N_rows], ct.N_rows, 0
        # ---- the table rows the kernels need.  A uint8 label map may use any of 256 ids (XCAT: input/params.txt:8-9,
        # plots.py:124) of which many share a composition and some do not occur in the slices scanned: count the ids on
        # the device, merge ids with the same (density, composition) - their rows of density x mixatten(E) are equal at
        # every energy - drop the absent ones, and renumber the uploaded copy.  Row 0 stays id 0 (its path length comes
        # from the chord).  mat_rows[k] = the phantom id whose table row serves compact id k.
        st = stream_ptr()
        vol_raw = to_dev(volume, torch.uint8, self.dev)
        cnt = torch.empty(256, dtype=torch.int64, device=self.dev)
        _native.check(self.lib.dexct_volume_ids(ptr(vol_raw), vol_raw.numel(), ptr(cnt), st), 'dexct_volume_ids')
        present = cnt.cpu().numpy() > 0
        if present[phantom.n_materials:].any():
            raise ValueError('the volume holds a material id without a table entry')
        self.mat_rows, lut = compact_ids(present, phantom.materials)
        if any(lut[i] != i for i in np.flatnonzero(present)):
            _native.check(self.lib.dexct_volume_remap(ptr(vol_raw), vol_raw.numel(), lut.ctypes.data_as(C.POINTER(C.c_uint8)), st),
                          'dexct_volume_remap')
        self.id_lut = lut
        # The 4-rows-per-lane kernels read aligned dwords along z: pad the uploaded copy with empty slices so that
        # the first imaged slice and the slice count are multiples of 4 (stacked fans only see their own slices,
        # the in-plane geometry does not change).
        packed_wanted = kernel in (3, 4, 5, 7, 8) or (kernel == 0 and ct.N_rows >= 64)
        # the 2-bit packed volume with bit-sliced counters (rows16_kernel): what kernel 0 picks where it applies
        lanes16 = -(-ct.N_rows // 16)
        group16 = 64 * (-(-lanes16 // 64)) if lanes16 > 32 else (32 if lanes16 > 16 else 16)     # lanes the pair occupies
        M = self.n_mat = len(self.mat_rows)
        shape2_ok = not self.cone and max(phantom.Nx, phantom.Ny) <= 2047
        # kernel 0 picks it when at least 3/4 of the pair's lane group carry rows (>= 192 rows)
        fills = ct.N_rows >= 192 and 4 * lanes16 >= 3 * group16
        packed2_ok = shape2_ok and 2 <= M <= 4                     # one fused pass: ids 0..3
        self.use_packed = (kernel == 7 and packed2_ok) or (kernel == 0 and packed2_ok and fills)
        if kernel == 7 and not packed2_ok:
            raise ValueError('kernel 7 (2-bit packed volume) needs a stacked fan, 2..4 materials and nx, ny <= 2047')
        # material groups (5..256 materials) on packed group codes: kernel 8 forces it, kernel 0 picks it like kernel 7
        self.grouped_packed = (kernel == 8 and shape2_ok and 2 <= M <= 256) or (kernel == 0 and shape2_ok and 4 < M <= 256 and fills)
        if kernel == 8 and not self.grouped_packed:
            raise ValueError('kernel 8 (material groups on the 2-bit packed volume) needs a stacked fan, 2..256 materials '
                             'and nx, ny <= 2047')
        align = 16 if (self.use_packed or self.grouped_packed) else 4
        if not self.cone and packed_wanted and (nz % align or z_first % align):
            lead = (-z_first) % align
            tail = (-(nz + lead)) % align
            vol_raw = torch.nn.functional.pad(vol_raw, (0, 0, 0, 0, lead, tail))      # empty slices (id 0)
            z_first, nz = z_first + lead, nz + lead + tail
        self.geom = _native.FanGeom(ct.N_proj, ct.N_channels, ct.N_rows, z_first, phantom.Nx, phantom.Ny,
                                    nz, 0, phantom.dx, phantom.dy, phantom.dz, ct.SID, ct.SDD)
        self.view_cs = to_dev(ct.view_cs(), torch.float64, self.dev)
        self.chan_cs = to_dev(ct.chan_cs(), torch.float64, self.dev)
        n_local = self.view_end - self.view_begin
        self.plan = torch.empty(n_local * ct.N_channels * _native.PLAN_BYTES, dtype=torch.uint8, device=self.dev)
        _native.check(self.lib.dexct_fan_plan(C.byref(self.geom), ptr(self.view_cs), ptr(self.chan_cs),
                                              self.view_begin, self.view_end, ptr(self.plan), st), 'dexct_fan_plan')
        self.vol_yx = vol_raw.contiguous()
        self.vol_xy = torch.empty_like(self.vol_yx)
        self.vol_zc = None
        self.cone_groups = False
        if self.cone:
            # cone beam: kernel 1 = one thread per ray (dexct_cone_project, any number of materials), 2 = the rows of a
            # (view, channel) pair as lanes (dexct_cone_project_rows for <= 3 materials; round 6: one pass per group of three
            # materials + one detection pass beyond that, dexct_cone_project_grouped); 0 picks 2 from 32 rows on
            self.cone_rows = kernel == 2 or (kernel == 0 and ct.N_rows >= 32)
            self.cone_groups = self.cone_rows and M > 3
            kernel = self.kernel = 1
            self.row_z = to_dev(ct.row_z(), torch.float64, self.dev)
            if self.cone_rows:
                nb = self.lib.dexct_cone_layout_bytes(phantom.Nx, phantom.Ny, nz)
                if self.cone_groups:
                    self.vol_zc = torch.empty(((M + 2) // 3) * nb, dtype=torch.uint8, device=self.dev)
                    _native.check(self.lib.dexct_cone_layout_groups(ptr(self.vol_yx), phantom.Nx, phantom.Ny, nz, M, ptr(self.vol_zc), st),
                                  'dexct_cone_layout_groups')
                else:
                    self.vol_zc = torch.empty(nb, dtype=torch.uint8, device=self.dev)
                    _native.check(self.lib.dexct_cone_layout(ptr(self.vol_yx), phantom.Nx, phantom.Ny, nz, ptr(self.vol_zc), st),
                                  'dexct_cone_layout')
        want_zf = kernel in (2, 3, 4, 5, 7, 8) or (kernel == 0 and ct.N_rows >= 64)
        self.vol_zf = torch.empty_like(self.vol_yx) if want_zf else None
        _native.check(self.lib.dexct_volume_layouts(ptr(self.vol_yx), phantom.Nx, phantom.Ny, nz,
                                                    ptr(self.vol_xy), ptr(self.vol_zf), st), 'dexct_volume_layouts')
        self.vol_z2 = None
        if self.use_packed:
            self.vol_z2 = torch.empty(self.vol_zf.numel() // 4, dtype=torch.uint8, device=self.dev)
            _native.check(self.lib.dexct_volume_pack2(ptr(self.vol_zf), self.vol_zf.numel(), ptr(self.vol_z2), st),
                          'dexct_volume_pack2')
        aligned = nz % 4 == 0 and z_first % 4 == 0
        self.grouped = not self.grouped_packed and (kernel == 4 or (kernel == 0 and want_zf and M > 4 and aligned))
        self.codes = None
        if self.grouped_packed:
            n_groups = (M - 1 + 2) // 3
            self.codes = torch.empty((n_groups, self.vol_zf.numel() // 4), dtype=torch.uint8, device=self.dev)
            _native.check(self.lib.dexct_volume_groups_pack2(ptr(self.vol_zf), self.vol_zf.numel(), M, ptr(self.codes), st),
                          'dexct_volume_groups_pack2')
        elif self.grouped:
            if not (2 <= M <= 256 and aligned):
                raise ValueError('kernel 4 needs 2..256 materials')
            n_groups = (M - 1 + 2) // 3
            self.codes = torch.empty((n_groups,) + tuple(self.vol_zf.shape), dtype=torch.uint8, device=self.dev)
            _native.check(self.lib.dexct_volume_groups(ptr(self.vol_zf), self.vol_zf.numel(), M, ptr(self.codes), st),
                          'dexct_volume_groups')

    @property
    def n_local_views(self):
        return self.view_end - self.view_begin

    def compact(self, mu):
        """The table rows of the compact ids the kernels see (one row per distinct composition that occurs in the scanned
        slices) out of a table with one row per phantom id."""
        return mu[self.mat_rows]

    def path_bounds(self):
        """(l_max [n_mat], c_max) in cm: no ray of this scan crosses more than l_max[k] of compact id k - the diagonal of the
        bounding box of its voxels (in the slice plane for a stacked fan, whose rays stay in their slice; in space for a cone
        beam); the grid's own diagonal for id 0 - nor more than c_max in total.  Computed once, on the device, from the
        uploaded volume (the domain quadrature.reduce_tables guarantees its error bound on)."""
        if self._bounds is None:
            ph = self.phantom
            v = self.vol_yx.view(-1, ph.Ny, ph.Nx)
            d = (ph.dz, ph.dy, ph.dx)
            axes = (0, 1, 2) if self.cone else (1, 2)
            c_max = float(np.sqrt(sum((v.shape[a] * d[a]) ** 2 for a in axes)))
            l_max = [c_max]
            for k in range(1, self.n_mat):
                m = v == k
                ext2 = 0.0
                for a in axes:
                    idx = torch.nonzero(m.any(dim=[b for b in (0, 1, 2) if b != a])).flatten()
                    if idx.numel():
                        ext2 += (float(idx[-1] - idx[0] + 1) * d[a]) ** 2
                l_max.append(min(c_max, float(np.sqrt(ext2))))
            self._bounds = (l_max, c_max)
        return self._bounds

    def upload_tables(self, specs, quadrature=None):
        """E, mu [n_mat, nE] and w [S, nE] on the device (float32) and the S unattenuated signals.  ``quadrature='reduced'``
        (or DEXCT_QUADRATURE=reduced) swaps the energy grid for the shorter one of quadrature.reduce_tables where it applies
        (<= 4 table rows, bound verified); ``self.quadrature_info`` then says what was used (None: the full grid)."""
        E, mu, w = merged_tables(self.ct, self.phantom, specs)
        mu_c, air = self.compact(mu), w.sum(axis=1)
        self.quadrature_info = None
        if _want_reduced(quadrature):
            from . import quadrature as _q
            red = _q.reduce_tables(mu_c, w, *self.path_bounds()) if self.n_mat <= _q.MAX_MATERIALS else None
            if red is not None:
                cols, w, self.quadrature_info = red
                E, mu_c = E[cols], mu_c[:, cols]
        return (E, to_dev(mu_c, torch.float32, self.dev), to_dev(w, torch.float32, self.dev), air)

    @property
    def native_layout(self):
        """1 (row fastest) when a row-parallel kernel will run, else 0 (channel fastest)."""
        if self.kernel in (2, 3, 4, 5, 7, 8):
            return 1
        return 1 if (self.kernel == 0 and self.vol_zf is not None and self.ct.N_rows >= 64) else 0

    def project_tables(self, mu_d, w_d, want_pathlen=False, out=None, layout=0, w2_d=None, seed=0, air=None,
                       log_out=None, views=None, want_variance=False):
        """Device-side call: mu_d [M, nE], w_d [S, nE] float32 tensors -> counts.

        ``air`` (S unattenuated signals, sum_e w[s][e]) asks for get_sino's second output as well, the log sinogram
        ln(air / counts) (main.py:120-122), written by the projection kernel's own detection store (dexct_log_out) into
        ``log_out`` or a new tensor of the counts' shape; the call then returns (counts[, pathlen], log).

        layout 0: counts [S, nV, rows, channels] (the reference's order); layout 1: [S, nV, channels, rows]
        (what the row-parallel kernels produce natively); ``layout=None`` returns the native one.  When
        the requested layout is not the kernel's own, the kernel writes its own layout and
        dexct_transpose_batched converts (cheaper than scattered 4-byte stores).
        ``w2_d`` (variance weights, merged_tables(with_variance=True)) switches quantum noise on (Philox, keyed by ``seed``
        and by the GLOBAL (view, row, channel), so shards reproduce the unsharded sinogram).  The default kernels - the
        packed stacked fan (<= 2 spectra) and the cone beam - sum the variance in the detection's own energy loop and draw
        the sample in registers (struct dexct_noise): no variance array, no extra pass, and the log sinogram comes from the
        same store.  The other kernels write the variance and dexct_add_noise draws the same sample from it.
        ``want_variance``: the variance of the detected signal as a further result (appended last; tests).
        ``views=(a, b)``: only the local views [a, b) of this projector's range (outputs sized for b - a views): a step that
        hands its sinogram on in chunks projects chunk by chunk (bench.py, the sharded step)."""
        # the kernels read the tables through raw pointers: dense float32, whatever view the caller built
        # (torch.tensor(w[:, keep]) keeps NumPy's column-major result of the advanced index)
        mu_d, w_d = (t.to(dtype=torch.float32).contiguous() for t in (mu_d, w_d))
        w2_d = None if w2_d is None else w2_d.to(dtype=torch.float32).contiguous()
        S, nE = w_d.shape
        if mu_d.shape[1] != nE or (w2_d is not None and tuple(w2_d.shape) != (S, nE)):
            raise ValueError(f'tables disagree: mu {tuple(mu_d.shape)}, w {tuple(w_d.shape)}' + ('' if w2_d is None else f', w2 {tuple(w2_d.shape)}'))
        sl0, sl1 = (0, self.n_local_views) if views is None else (int(views[0]), int(views[1]))
        if not 0 <= sl0 < sl1 <= self.n_local_views:
            raise ValueError(f'views={views}: a non-empty range within the {self.n_local_views} local views')
        plan_ptr = self.plan.data_ptr() + sl0 * self.ct.N_channels * _native.PLAN_BYTES
        vb, ve = self.view_begin + sl0, self.view_begin + sl1
        if mu_d.shape[0] != self.n_mat:
            if mu_d.shape[0] != self.phantom.n_materials:
                raise ValueError(f'mu has {mu_d.shape[0]} rows; expected {self.n_mat} (compact ids, Projector.compact) or '
                                 f'{self.phantom.n_materials} (one per phantom id)')
            mu_d = mu_d[torch.as_tensor(self.mat_rows, device=mu_d.device)].contiguous()
        M = self.n_mat
        ct = self.ct
        nV, nR, nC = sl1 - sl0, ct.N_rows, ct.N_channels
        native = self.native_layout
        want = native if layout is None else layout
        shape = {0: (S, nV, nR, nC), 1: (S, nV, nC, nR)}
        direct = want == native or nR == 1
        run_layout = want if direct else native
        if out is not None and direct:
            counts = out
        else:
            counts = torch.empty(shape[run_layout], dtype=torch.float32, device=self.dev)
        pathlen = None
        if want_pathlen:
            pl_shape = (nV, nR, nC, M) if run_layout == 0 else (nV, nC, nR, M)
            pathlen = torch.empty(pl_shape, dtype=torch.float32, device=self.dev)
        noisy = w2_d is not None
        # kernels that draw the noise sample themselves (ABI 6): the packed stacked fan, the cone beam, and the detection pass of
        # the material groups (<= 2 spectra, <= 48 table rows); everything else writes the variance for dexct_add_noise
        groups = self.cone_groups or (not self.cone and (self.grouped_packed or self.grouped))
        in_kernel = noisy and ((self.cone and not self.cone_groups) or (self.use_packed and S <= 2 and not groups)
                               or (groups and S <= 2 and M <= 48))
        variance = torch.empty_like(counts) if (noisy and (want_variance or not in_kernel)) else None
        nz = _native.noise(seed) if in_kernel else None
        # the log sinogram: written by the detection store when the kernel's layout is the one wanted; else together with the
        # transpose at the end (dexct_transpose_log: the counts are read once for both outputs)
        fuse_log = air is not None and not direct
        log = None
        if air is not None and not fuse_log:
            log = log_out if log_out is not None else torch.empty(shape[run_layout], dtype=torch.float32, device=self.dev)
        # fused into the detection store, except where the noisy counts only exist after dexct_add_noise
        lo = _native.log_out(ptr(log), air) if (log is not None and (not noisy or in_kernel)) else None
        if self.cone:
            max_dz = float(np.max(np.abs(self.ct.row_z() - self.ct.src_z)))

            # (quantum noise: the variance of the detected signal comes out of the same launch and the kernel draws the sample -
            # round 5 ran the kernel a second time with w2 as weights)
            if self.cone_groups:
                # more than 3 table rows: one traversal per group of three materials into M planes of path lengths, one detection
                # pass over them; many materials on a large scan go through in view chunks (the scratch stays bounded)
                per_view = M * nR * nC * 4
                n_chunk = max(1, min(nV, _GROUP_SCRATCH_BYTES // max(per_view, 1)))
                scratch = torch.empty((M, n_chunk * nR * nC), dtype=torch.float32, device=self.dev)
                for v0 in range(0, nV, n_chunk):
                    v1 = min(nV, v0 + n_chunk)
                    whole = v0 == 0 and v1 == nV
                    c_t = counts if whole else torch.empty((S, v1 - v0, nR, nC), dtype=torch.float32, device=self.dev)
                    p_t = pathlen if (whole or pathlen is None) else torch.empty((v1 - v0, nR, nC, M), dtype=torch.float32, device=self.dev)
                    v_t = variance if (whole or variance is None) else torch.empty_like(c_t)
                    l_t = log if (whole or lo is None) else torch.empty_like(c_t)
                    lo_t = lo if whole else (_native.log_out(ptr(l_t), air) if lo is not None else None)
                    _native.check(self.lib.dexct_cone_project_grouped(
                        C.byref(self.geom), plan_ptr + v0 * nC * _native.PLAN_BYTES, ptr(self.view_cs), ptr(self.chan_cs), ptr(self.row_z),
                        self.ct.src_z, max_dz, vb + v0, vb + v1, ptr(self.vol_zc), M, nE, S, ptr(mu_d), ptr(w_d), ptr(c_t), ptr(p_t),
                        ptr(scratch), lo_t, ptr(w2_d), ptr(v_t), nz, stream_ptr()), 'dexct_cone_project_grouped')
                    if not whole:
                        counts[:, v0:v1].copy_(c_t)
                        if pathlen is not None:
                            pathlen[v0:v1].copy_(p_t)
                        if variance is not None:
                            variance[:, v0:v1].copy_(v_t)
                        if lo is not None:
                            log[:, v0:v1].copy_(l_t)
            elif self.cone_rows:
                _native.check(self.lib.dexct_cone_project_rows(
                    C.byref(self.geom), plan_ptr, ptr(self.view_cs), ptr(self.chan_cs), ptr(self.row_z),
                    self.ct.src_z, max_dz, vb, ve, ptr(self.vol_zc), M, nE, S, ptr(mu_d),
                    ptr(w_d), ptr(counts), ptr(pathlen), lo, ptr(w2_d), ptr(variance), nz, stream_ptr()), 'dexct_cone_project_rows')
            else:
                _native.check(self.lib.dexct_cone_project(
                    C.byref(self.geom), plan_ptr, ptr(self.view_cs), ptr(self.chan_cs), ptr(self.row_z),
                    self.ct.src_z, max_dz, vb, ve, ptr(self.vol_yx), ptr(self.vol_xy), M, nE, S,
                    ptr(mu_d), ptr(w_d), ptr(counts), ptr(pathlen), lo, ptr(w2_d), ptr(variance), nz, stream_ptr()), 'dexct_cone_project')
        elif self.use_packed and (not noisy or in_kernel):      # (noise on > 2 spectra: the byte-volume kernel below)
            _native.check(self.lib.dexct_siddon_project_packed(
                C.byref(self.geom), plan_ptr, vb, ve, ptr(self.vol_z2), M, nE, S,
                ptr(mu_d), ptr(w_d), ptr(counts), ptr(pathlen), run_layout, lo, ptr(w2_d), ptr(variance), nz, stream_ptr()),
                'dexct_siddon_project_packed')
        elif self.grouped_packed or self.grouped:        # noise too: the detection pass carries the variance
            fn, what = ((self.lib.dexct_siddon_project_grouped_packed, 'dexct_siddon_project_grouped_packed') if self.grouped_packed
                        else (self.lib.dexct_siddon_project_grouped, 'dexct_siddon_project_grouped'))
            # the per-material accumulators of the group passes: M x rays floats.  Many materials on a large scan go through
            # in view chunks so that this scratch stays below _GROUP_SCRATCH_BYTES (the outputs of a chunk are copied into place)
            per_view = M * nR * nC * 4
            n_chunk = max(1, min(nV, _GROUP_SCRATCH_BYTES // max(per_view, 1)))
            if n_chunk >= nV:
                scratch = torch.empty((M, nV * nR * nC), dtype=torch.float32, device=self.dev)
                _native.check(fn(C.byref(self.geom), plan_ptr, vb, ve, ptr(self.codes), M, nE, S,
                                 ptr(mu_d), ptr(w_d), ptr(counts), ptr(pathlen), ptr(scratch), run_layout, ptr(w2_d), ptr(variance),
                                 lo, nz, stream_ptr()), what)
            else:
                scratch = torch.empty((M, n_chunk * nR * nC), dtype=torch.float32, device=self.dev)
                for v0 in range(0, nV, n_chunk):
                    v1 = min(nV, v0 + n_chunk)
                    sub = (S, v1 - v0) + tuple(counts.shape[2:])
                    c_t = torch.empty(sub, dtype=torch.float32, device=self.dev)
                    p_t = torch.empty((v1 - v0,) + tuple(pathlen.shape[1:]), dtype=torch.float32, device=self.dev) if pathlen is not None else None
                    v_t = torch.empty_like(c_t) if variance is not None else None
                    l_t = torch.empty_like(c_t) if lo is not None else None
                    lo_t = _native.log_out(ptr(l_t), air) if lo is not None else None
                    _native.check(fn(C.byref(self.geom), plan_ptr + v0 * nC * _native.PLAN_BYTES, vb + v0,
                                     vb + v1, ptr(self.codes), M, nE, S, ptr(mu_d), ptr(w_d), ptr(c_t), ptr(p_t),
                                     ptr(scratch), run_layout, ptr(w2_d), ptr(v_t), lo_t, nz, stream_ptr()), what)
                    counts[:, v0:v1].copy_(c_t)
                    if pathlen is not None:
                        pathlen[v0:v1].copy_(p_t)
                    if variance is not None:
                        variance[:, v0:v1].copy_(v_t)
                    if l_t is not None:
                        log[:, v0:v1].copy_(l_t)
        else:
            _native.check(self.lib.dexct_siddon_project(
                C.byref(self.geom), plan_ptr, vb, ve, ptr(self.vol_yx),
                ptr(self.vol_xy), ptr(self.vol_zf), M, nE, S, ptr(mu_d), ptr(w_d), ptr(counts), ptr(pathlen),
                3 if self.kernel in (7, 8) else self.kernel, run_layout, ptr(w2_d), ptr(variance), lo, stream_ptr()),
                'dexct_siddon_project')
        if noisy and not in_kernel:
            _native.check(self.lib.dexct_add_noise(ptr(counts), ptr(variance), S, nV, nR, nC, run_layout,
                                                   vb, int(seed) & (2 ** 64 - 1), stream_ptr()),
                          'dexct_add_noise')
            if log is not None:
                self.sino_log(counts, air, log)
        if not direct:
            dst = out if out is not None else torch.empty(shape[want], dtype=torch.float32, device=self.dev)
            r, c = (nC, nR) if run_layout == 1 else (nR, nC)
            if fuse_log:
                log = log_out if log_out is not None else torch.empty(shape[want], dtype=torch.float32, device=self.dev)
            self.transpose_log(counts, dst, log, air, r, c)
            counts = dst
            if pathlen is not None:
                pathlen = pathlen.permute(0, 2, 1, 3).contiguous()      # test-only output
            if variance is not None and want_variance:
                variance = variance.permute(0, 1, 3, 2).contiguous()    # test-only output
        res = (counts, pathlen) if want_pathlen else (counts,)
        if log is not None:
            res = res + (log,)
        if want_variance:
            res = res + (variance,)
        return res if len(res) > 1 else res[0]

    def transpose_log(self, src, dst, log_dst, air, rows, cols):
        """src [S, n, rows, cols] -> dst [S, n, cols, rows] and (``log_dst`` not None) ln(air[s] / dst[s]) in the same pass
        (dexct_transpose_log)."""
        import ctypes as _C
        S, n = int(src.shape[0]), int(src.shape[1])
        arr = (_C.c_float * S)(*[float(x) for x in air[:S]]) if log_dst is not None else None
        _native.check(self.lib.dexct_transpose_log(ptr(src), ptr(dst), ptr(log_dst), arr, S, n, rows, cols, stream_ptr()),
                      'dexct_transpose_log')

    def sino_log(self, counts, air, out=None):
        """ln(air[s] / counts[s]) as a pass of its own (dexct_sino_log): noisy and gathered sinograms."""
        import ctypes as _C
        S = counts.shape[0]
        out = torch.empty_like(counts) if out is None else out
        arr = (_C.c_float * S)(*[float(x) for x in air[:S]])
        _native.check(self.lib.dexct_sino_log(ptr(counts), arr, S, counts[0].numel(), ptr(out), stream_ptr()),
                      'dexct_sino_log')
        return out

    def project(self, specs, want_pathlen=False, layout=0, noise=False, seed=0, want_log=False, quadrature=None):
        """noise: False (expectation), True / 'gaussian' (compound-Poisson variance, normal sample) or 'poisson'
        (per-energy-bin Poisson photon counts: exact for photon-starved rays, ~5x the projection time).
        ``want_log``: also the log sinogram ln(air / counts), from the device (appended to the projection's result).
        ``quadrature``: see upload_tables (noise-free projections only: the noisy ones keep the full grid, whose bins the
        variance and the photon counts are defined on)."""
        if not noise:
            _, mu_d, w_d, air = self.upload_tables(specs, quadrature)
            return self.project_tables(mu_d, w_d, want_pathlen, layout=layout, air=air if want_log else None), air
        if noise == 'poisson':
            res, air = self._project_poisson(specs, want_pathlen, layout, seed)
            if want_log:
                res = (res if isinstance(res, tuple) else (res,))
                res = res + (self.sino_log(res[0], air),)
            return res, air
        _, mu, w, w2 = merged_tables(self.ct, self.phantom, specs, with_variance=True)
        mu_d, w_d, w2_d = (to_dev(x, torch.float32, self.dev) for x in (self.compact(mu), w, w2))
        air = w.sum(axis=1)
        return self.project_tables(mu_d, w_d, want_pathlen, layout=layout, w2_d=w2_d, seed=seed,
                                   air=air if want_log else None), air

    def _project_poisson(self, specs, want_pathlen, layout, seed):
        E, mu, w = merged_tables(self.ct, self.phantom, specs)
        gain = E if self.ct.eid else np.ones_like(E)
        mu_d, w_d = to_dev(self.compact(mu), torch.float32, self.dev), to_dev(w, torch.float32, self.dev)
        ph_d, gain_d = to_dev(w / gain, torch.float32, self.dev), to_dev(gain, torch.float32, self.dev)
        # path lengths from whichever traversal kernel applies (its noise-free counts are discarded)
        want = self.native_layout if layout is None else layout
        counts, pathlen = self.project_tables(mu_d, w_d, want_pathlen=True, layout=want)
        S, nE = w_d.shape
        nV, nR, nC = self.n_local_views, self.ct.N_rows, self.ct.N_channels
        _native.check(self.lib.dexct_poisson_detect(
            ptr(pathlen), ptr(mu_d), ptr(ph_d), ptr(gain_d), mu_d.shape[0], nE, S, nV, nR, nC, want, self.view_begin,
            int(seed) & (2 ** 64 - 1), ptr(counts), stream_ptr()), 'dexct_poisson_detect')
        return ((counts, pathlen) if want_pathlen else counts), w.sum(axis=1)

    def trace(self, rays_vrc, max_seg=None):
        """Voxel-index sequence and float32 piece lengths of selected rays (views relative to the shard)."""
        rays = to_dev(np.asarray(rays_vrc, dtype=np.int32).reshape(-1, 3), torch.int32, self.dev)
        n = rays.shape[0]
        max_seg = max_seg or 2 * max(self.phantom.Nx, self.phantom.Ny) + 4
        vox = torch.zeros((n, max_seg), dtype=torch.int32, device=self.dev)
        ln = torch.zeros((n, max_seg), dtype=torch.float32, device=self.dev)
        ns = torch.zeros(n, dtype=torch.int32, device=self.dev)
        _native.check(self.lib.dexct_siddon_trace(C.byref(self.geom), ptr(self.plan), ptr(rays), n, max_seg,
                                                  ptr(vox), ptr(ln), ptr(ns), stream_ptr()), 'dexct_siddon_trace')
        return vox.cpu().numpy(), ln.cpu().numpy(), ns.cpu().numpy()

    def plan_host(self):
        """The plan table as a NumPy structured array (for parity tests)."""
        dt = np.dtype([('V0', '<i8'), ('SV', '<i8'), ('i_first', '<i4'), ('n_slabs', '<i4'), ('kf', '<f4'),
                       ('len_per_u', '<f4'), ('chord_u', '<f4'), ('flags', '<u4')])
        return self.plan.cpu().numpy().view(dt)


def _want_reduced(quadrature):
    """None: what DEXCT_QUADRATURE says (default 'full')."""
    q = quadrature if quadrature is not None else _os.environ.get('DEXCT_QUADRATURE', 'full')
    if q not in ('full', 'reduced'):
        raise ValueError(f"quadrature must be 'full' or 'reduced', not {q!r}")
    return q == 'reduced'


# scratch of the material-group passes above which a projection goes through in view chunks (DEXCT_GROUP_SCRATCH_GB)
_GROUP_SCRATCH_BYTES = int(float(_os.environ.get('DEXCT_GROUP_SCRATCH_GB', '16')) * 2 ** 30)

_cache = {}

# True (the default): every get_sino call checksums the WHOLE volume (xxh3, 64 bits; 15 ms per 128 MiB) and compares it with
# the checksum of the bytes the device-resident state was built from - the reference rebuilds its state on every call, so
# an in-place edit of ``phantom.volume`` must never return a stale sinogram.  The checksum is computed by the calling thread
# AFTER it has queued the projection kernel and the device-to-host copies and BEFORE it waits for them (everything it queued
# is asynchronous), so it costs the call no time as long as hashing is faster than the GPU work (512^3: 15 against 75 ms;
# one 512^2 slice: 0.03 against 0.5 ms); on a mismatch the state is rebuilt and the projection redone.
# False / DEXCT_VERIFY_VOLUME=0 opts out: the O(1) key below alone (version counter + a strided sample).
verify_volume = True


def _verify_enabled():
    import os
    return verify_volume and os.environ.get('DEXCT_VERIFY_VOLUME', '1') != '0'


def _hash64(a):
    """64-bit hash of an array's bytes: xxh3 where the xxhash module is present, else blake2b."""
    a = np.ascontiguousarray(a)
    try:
        import xxhash
        return xxhash.xxh3_64_intdigest(a.data)
    except ImportError:
        import hashlib
        return int.from_bytes(hashlib.blake2b(a.data, digest_size=8).digest(), 'little')


def _sample_step(shape, n_samples=4096):
    """Stride of the O(1) sample: about size / n_samples and coprime to every dimension, so that the samples spread over
    all rows, columns and slices (a stride that is a multiple of Nx - what size / 4096 is for every power-of-two volume -
    only ever visits the x = 0 face, which is air for any centred phantom)."""
    from math import gcd
    size = int(np.prod(shape))
    step = max(1, -(-size // n_samples)) | 1
    while any(d > 1 and gcd(step, int(d)) != 1 for d in shape):
        step += 2
    return step


def _volume_key(phantom):
    """What identifies the bytes of ``phantom.volume`` without reading them all: the phantom's version counter (bumped
    whenever ``volume`` is assigned and by ``phantom.touch()``), the identity and shape of the array, and a hash of ~4096
    voxels at a stride coprime to the dimensions.  This key only decides whether the cached state is a CANDIDATE; whether
    its bytes are still the volume's is settled by the whole-volume checksum (``verify_volume``, on by default)."""
    v = phantom.volume
    flat = v.reshape(-1)
    return (getattr(phantom, 'version', 0), id(v), v.shape, _hash64(flat[::_sample_step(v.shape)]))


def _fingerprint(ct, phantom, view_range):
    """Everything the device-resident state (volume layouts, ray plans) depends on.  The reference rebuilds its
    state on every get_sino call; here the state is reused only while the scanner numbers, the voxel sizes and
    the volume (see _volume_key and verify_volume) are what they were when it was built."""
    return (id(ct), id(phantom), view_range, ct.N_proj, ct.N_channels, ct.N_rows, ct.SID, ct.SDD,
            _hash64(ct.thetas), _hash64(ct.gammas),
            ct.h_iso, bool(getattr(ct, 'cone', False)), float(getattr(ct, 'src_z', 0.0)),
            phantom.z_index, phantom.Nx, phantom.Ny, phantom.Nz, phantom.dx, phantom.dy, phantom.dz,
            tuple((float(m.density), str(m.matcomp)) for m in phantom.materials)) + _volume_key(phantom)


def invalidate():
    """Drop the cached device state (the next get_sino call rebuilds it)."""
    _cache.clear()


def _projector(ct, phantom, view_range):
    """The device-resident state for this (scanner, phantom, shard), and whether the caller has to verify it: True when
    it was found in the cache and verification is on - the caller then compares ``_hash64(phantom.volume)`` with
    ``pj.volume_hash`` before it hands results out (False for a state built just now: its checksum is of the bytes it was
    built from)."""
    key = _fingerprint(ct, phantom, view_range)
    pj = _cache.get(key)
    if pj is None or pj.ct is not ct or pj.phantom is not phantom:
        _cache.clear()                      # keep one (scanner, phantom) pair resident
        pj = _cache[key] = Projector(ct, phantom, view_range)
        pj.volume_hash = _hash64(phantom.volume) if _verify_enabled() else None      # (the layouts are still being written)
        return pj, False
    return pj, bool(_verify_enabled() and pj.volume_hash is not None)


_SINO_CHUNKS = 8                # view chunks of a large noise-free projection on one process: chunk k + 1 is projected while chunk k
                                # crosses PCIe (3.3 GB of results take 58 ms, their kernels 12)


def _get_sinos_pipelined(pj, check, ct, phantom, specs, seed, quadrature, noise=False):
    """get_sinos for a large scan on one process: the views are projected in _SINO_CHUNKS chunks (Projector.project_tables,
    views=) and every chunk's two outputs leave for host memory on a download stream while the next chunk is projected.  The same
    kernels on the same rays as the single launch: the same bits - with quantum noise too (``noise`` True: the Philox counter is
    the global view).  One host block per spectrum and output (_device.LazyPinnedResult: touched and locked chunk by chunk in
    front of the copies, unlocked before it is returned)."""
    w2_d = None
    if noise:
        _, mu, w, w2 = merged_tables(ct, phantom, specs, with_variance=True)
        mu_d, w_d, w2_d = (to_dev(x, torch.float32, pj.dev) for x in (pj.compact(mu), w, w2))
        air = w.sum(axis=1)
    else:
        _, mu_d, w_d, air = pj.upload_tables(specs, quadrature)
    S, nV, nR, nC = len(specs), pj.n_local_views, ct.N_rows, ct.N_channels
    bounds = [_shard.split(nV, k, _SINO_CHUNKS) for k in range(_SINO_CHUNKS)]
    row = nR * nC * 4
    cuts = [b * row for b, _ in bounds] + [nV * row]
    holders = [[LazyPinnedResult(pj.lib, (nV, nR, nC), np.float32, cuts, pj.dev.index or 0) for _ in range(2)] for _ in range(S)]
    main, _, down = side_streams(pj.dev)      # (kernels and downloads on queues of their own, _device.side_streams)
    main.wait_stream(torch.cuda.current_stream())
    keep = []                                  # (the chunks' device tensors live until their copies are done)
    with torch.cuda.stream(main):
        for k, (b, e) in enumerate(bounds):
            c_k, l_k = pj.project_tables(mu_d, w_d, layout=0, air=air, views=(b, e), w2_d=w2_d, seed=seed)
            keep.append((c_k, l_k))
            done = torch.cuda.Event()
            done.record(main)
            down.wait_event(done)
            for s_i in range(S):
                holders[s_i][0].download(k, c_k[s_i].data_ptr(), down)
                holders[s_i][1].download(k, l_k[s_i].data_ptr(), down)
    stale = check and _hash64(phantom.volume) != pj.volume_hash        # (computed while the GPU works, as in get_sinos)
    down.synchronize()
    out = [(h[0].finish(), h[1].finish()) for h in holders]
    del keep
    if stale:
        del out
        invalidate()
        return get_sinos(ct, phantom, specs, noise=noise, seed=seed, quadrature=quadrature)
    if nR == 1:
        out = [(r[:, 0, :], l[:, 0, :]) for r, l in out]
    return out


_DOWNLOAD_PIECE = 128 << 20     # bytes of a large result per copy (and per step of the page-locking in front of it)


def get_sinos(ct, phantom, specs, noise=False, seed=0, quadrature=None):
    """Several spectra from ONE traversal (path lengths are energy independent).

    ``noise=True`` adds quantum noise for the dose the spectra are scaled to (compound-Poisson variance,
    Gaussian sample; ``noise='poisson'`` draws per-energy-bin Poisson photon counts instead, exact for
    photon-starved rays), counter-based Philox RNG keyed by ``seed`` (csrc/noise.hip); the default is the
    noise-free expectation, which is what every parity test uses.

    Returns a list of (sino_raw, sino_log) float32 NumPy pairs, shaped [N_proj, N_channels]
    (or [N_proj, N_rows, N_channels] for N_rows > 1).  Both come from the device (the log sinogram is written by
    the projection kernel's detection store) through page-locked host memory: the arrays returned are views of a
    pinned buffer that belongs to them alone (torch's caching host allocator hands it out again once they are
    garbage).  ``quadrature='reduced'`` (opt-in; quadrature.py) evaluates the noise-free detection on a shorter energy grid
    whose relative error is verified <= 1e-6 over every path length the phantom allows.  Under torch.distributed the projection angles are sharded over the ranks and every rank returns the
    full gathered sinograms.
    """
    vb, ve = _shard.my_views(ct.N_proj)
    pj, check = _projector(ct, phantom, (vb, ve))
    sharded = _shard.world()[1] > 1
    per_array = ct.N_proj * ct.N_rows * ct.N_channels * 4
    if not sharded and noise != 'poisson' and pool_wanted(per_array) and ct.N_proj >= 4 * _SINO_CHUNKS:
        return _get_sinos_pipelined(pj, check, ct, phantom, specs, seed, quadrature, noise=bool(noise))
    res, air = pj.project(specs, noise=noise, seed=seed, want_log=not sharded, quadrature=quadrature)
    if sharded:
        counts = _shard.gather_views(res, ct.N_proj, view_dim=1, tag='get_sinos', mode=_shard.dropin_mode())
        log = pj.sino_log(counts, air)              # of the gathered sinogram: one collective instead of two
    else:
        counts, log = res
    # both results start towards page-locked host memory; the checksum of the volume is computed while the kernels and the
    # copies run; one wait for everything
    n_bytes = counts.numel() * 4
    if pool_wanted(n_bytes):
        # large results: blocks of the host pool - locked already, or (a process's first projections) locked piece by piece while
        # the copies of the pieces before run (_device.LazyPinnedResult) - instead of two page-locked allocations in front of them
        cuts = list(range(0, n_bytes, _DOWNLOAD_PIECE)) + [n_bytes]
        lz = [LazyPinnedResult(pj.lib, tuple(t.shape), np.float32, cuts, pj.dev.index or 0) for t in (counts, log)]
        cur = torch.cuda.current_stream()
        for k in range(len(cuts) - 1):
            for holder, t in zip(lz, (counts, log)):
                holder.download(k, t.data_ptr() + cuts[k], cur)
        stale = check and _hash64(phantom.volume) != pj.volume_hash
        cur.synchronize()
        raw, lg = lz[0].finish(), lz[1].finish()
        h_raw = h_lg = None
    else:
        h_raw, h_lg = pinned_empty(counts.shape, counts.dtype), pinned_empty(log.shape, log.dtype)
        h_raw.copy_(counts, non_blocking=True)
        h_lg.copy_(log, non_blocking=True)
        stale = check and _hash64(phantom.volume) != pj.volume_hash
        torch.cuda.current_stream().synchronize()
        raw, lg = h_raw.numpy(), h_lg.numpy()
    if stale:
        # phantom.volume was edited in place since the device state was built (no touch()): what was just computed is of
        # the old bytes.  Rebuild from the current ones and project again - the result is what the reference, which
        # builds its state on every call, returns.  (Under torch.distributed every rank holds the same phantom and
        # takes the same branch.)
        del raw, lg, h_raw, h_lg, counts, log, res
        invalidate()
        return get_sinos(ct, phantom, specs, noise=noise, seed=seed, quadrature=quadrature)
    if ct.N_rows == 1:
        raw, lg = raw[:, :, 0, :], lg[:, :, 0, :]
    return [(raw[k], lg[k]) for k in range(len(specs))]


def get_sino(ct, phantom, spec, noise=False, seed=0, quadrature=None):
    """Drop-in for the reference call ``sino_raw, sino_log = get_sino(ct, phantom, spec)`` (main.py:120)."""
    return get_sinos(ct, phantom, [spec], noise=noise, seed=seed, quadrature=quadrature)[0]
