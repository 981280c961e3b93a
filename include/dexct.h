/*
 * dexct.h - C ABI of the MI355X-native dual-energy CT hot path.
 *
 * The reference (gjadick/dex-ct-sim) has no FFI: its boundary for this path is two Python
 * functions, get_sino(ct, phantom, spec) (main.py:120, source in the un-vendored x-tomo-sim
 * submodule) and get_basismat_sinos(ct, sino1, sino2, spec1, spec2, n_iters, mask_thresh)
 * (matdecomp.py:167-207).  The Python shims with exactly those signatures live in
 * dex-ct-sim_amd/ and call the entry points below through ctypes; INTEGRATION.md shows the
 * binding.  Everything here is plain C: device pointers are passed as void* / typed pointers
 * obtained from any HIP allocator (the shims use torch-ROCm tensors as containers), the
 * stream is a hipStream_t passed as void*, the caller owns every buffer, no entry point
 * allocates, synchronises or throws.  Return value: 0 on success, a negative DEXCT_E* code
 * otherwise (dexct_strerror gives text).
 */
#ifndef DEXCT_H
#define DEXCT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DEXCT_ABI_VERSION 6   /* 2: log_out argument of the projection entry points, dexct_sino_log;
                                 3: struct dexct_gn_options - tolerance stop, results in the reference's order; 256 material ids;
                                 4: dexct_gn_options.pass / .iterations / .start - the Newton short cut (tabulated fixed points);
                                 5: dexct_gn_options.flags / .blocks_per_cu (what the environment used to switch per call), the
                                    two-launch form of the short cut removed, DEXCT_GN_FLAG_ONE_STEP, dexct_sino_gather (peer-to-peer
                                    assembly), dexct_transpose_log (both outputs of get_sino in one pass), dexct_host_touch / _pin /
                                    _unpin / dexct_download (the NumPy boundary of large arrays);
                                 6: quantum noise inside the projection kernels - struct dexct_noise; weights2 / variance / noise
                                    arguments of dexct_siddon_project_packed, dexct_cone_project, dexct_cone_project_rows; one Philox
                                    block per detector pixel serves all its spectra (dexct_add_noise draws the same sample);
                                    dexct_cone_layout_groups / dexct_cone_project_grouped (material groups on the row-parallel
                                    cone kernels) */

#define DEXCT_OK 0
#define DEXCT_EINVAL (-1)   /* bad argument (null pointer, non-positive size, unsupported combination) */
#define DEXCT_ERANGE (-2)   /* a size exceeds what the kernels support (see DEXCT_MAX_*) */
#define DEXCT_EHIP (-3)     /* a HIP runtime call failed; dexct_last_hip_error() has the code */
#define DEXCT_ERCCL (-4)    /* RCCL not found in the process, or ncclAllGather failed (its ncclResult_t is then
                               what dexct_last_hip_error() returns) */

#define DEXCT_MAX_MATERIALS 256 /* every id a uint8 volume can hold (ABI 3; 48 before).  Fast paths: <= 4 ids in one pass,
                                  groups of three beyond; above 48 table rows the general kernels keep their per-material sums
                                  in LDS columns of 64 lanes.  The Python host first merges ids with identical composition and
                                  drops ids the volume does not hold (forward_project.Projector). */
#define DEXCT_MAX_SPECTRA 4    /* spectra detected per traversal */
#define DEXCT_FIX_FRAC 40      /* fractional bits of the fixed-point minor-axis coordinate */

/* Fan-beam scan of a voxel grid (the reference's "fan_beam" scanner_geometry,
 * input/params.txt:18-27, with detector rows stacked along z: row r images slice z_first + r,
 * which at n_rows == 1 is the reference's z_index, params.txt:16). */
typedef struct dexct_fan_geom {
  int32_t n_views;    /* N_projections */
  int32_t n_channels; /* N_channels    */
  int32_t n_rows;     /* detector rows = z-slices imaged (1 in the reference) */
  int32_t z_first;    /* slice index of row 0 */
  int32_t nx, ny, nz; /* phantom grid */
  int32_t pad_;
  double dx, dy, dz;  /* voxel size [cm] */
  double sid, sdd;    /* source-isocentre / source-detector distance [cm] */
} dexct_fan_geom;

/* Per (view, channel) traversal plan, produced on the device by dexct_fan_plan.
 * All rows of a stacked fan share it.  The minor in-plane coordinate v is carried as a
 * signed fixed-point number with DEXCT_FIX_FRAC fractional bits: v(i) = (V0 + i*SV) / 2^40
 * at the entry face of dominant-axis slab i, so voxel indices are exact integer arithmetic
 * and identical on any device or host. */
typedef struct dexct_ray_plan {
  int64_t V0;        /* fixed-point v at u = 0 */
  int64_t SV;        /* fixed-point dv/du, |SV| <= 2^40 */
  int32_t i_first;   /* first dominant-axis slab the ray touches */
  int32_t n_slabs;   /* number of slabs (0 = ray misses the grid) */
  float kf;          /* 2^-32 * min(1/|dv/du|, 2^24): fraction -> crossing parameter */
  float len_per_u;   /* path length [cm] per unit of u */
  float chord_u;     /* length of the ray inside the grid, in units of u */
  uint32_t flags;    /* bit0: dominant axis (0: u=x, v=y; 1: u=y, v=x); bit1: SV > 0 */
} dexct_ray_plan;

/* The second output of get_sino, `sino_raw, sino_log = get_sino(ct, phantom, spec)` (main.py:120-122; sino_log is what
 * get_recon reconstructs, main.py:134): sino_log[s*n_rays + ray] = ln(air[s] / counts[s*n_rays + ray]) in float32, written
 * by the projection's own detection store in the ray order of counts.  air[s] = sum_e weights[s*n_energies + e], the
 * unattenuated signal.  Pass NULL (or a NULL sino_log) when the log sinogram is not wanted.  Not together with a
 * variance output (DEXCT_EINVAL): the log of a noisy sinogram is taken after dexct_add_noise, by dexct_sino_log. */
typedef struct dexct_log_out {
  float* sino_log;              /* device, n_spectra * n_rays float32 */
  float air[DEXCT_MAX_SPECTRA]; /* host values, copied into the launch arguments */
} dexct_log_out;

/* Quantum noise drawn by the projection kernel itself (ABI 6).  The reference scales every spectrum to a dose before it
 * projects (main.py:68, doses at main.py:101) and reads sino_raw as photon counts (matdecomp.py:30,179): the noisy scan is its
 * live mode.  With `weights2` (weights x detector signal per photon, as for dexct_siddon_project) an entry point that takes
 * this struct accumulates the variance of the detected signal, sum_e weights2 exp(-...), in the same energy loop as the signal
 * (the exponentials are shared) and
 *   sample != 0: writes counts = max(signal + sqrt(variance) z, 1e-20), z standard normal from Philox4x32-10 with counter
 *                (global view, row, channel, 0) and key `seed` - one block per detector pixel, its four words feed two
 *                Box-Muller pairs, spectrum s takes normal s - exactly the sample dexct_add_noise draws from the same signal
 *                and variance (shard-, layout- and kernel-independent).  A log_out then receives the log of the SAMPLED counts.
 *   variance != NULL (argument of the entry point): also writes the variances, [s*n_rays + ray] like counts.
 * At least one of the two; at most two spectra.  NULL / weights2 == NULL: the noise-free expectation. */
typedef struct dexct_noise {
  uint64_t seed;
  int32_t sample;
  int32_t reserved_;
} dexct_noise;

const char* dexct_strerror(int code);
int dexct_abi_version(void);
int dexct_last_hip_error(void);

/* Which ids a volume holds, and renumbering them (ABI 3): a uint8 label map may use any of 256 ids (XCAT label maps:
 * input/params.txt:8-9, plots.py:124) while only a few table rows differ - the host counts the ids
 * (dexct_volume_ids: counts256[id] = number of voxels with that id, 256 device uint64, zeroed by the call; vol 16-byte
 * aligned), merges ids of identical composition, drops the ones that do not occur, and renumbers the uploaded copy in
 * place (dexct_volume_remap: vol[i] = lut256[vol[i]], lut256 = 256 HOST bytes, copied into the launch arguments).  The
 * projection kernels then see the compact ids 0..n_materials-1. */
int dexct_volume_ids(const uint8_t* vol, int64_t n_voxels, uint64_t* counts256, void* stream);
int dexct_volume_remap(uint8_t* vol, int64_t n_voxels, const uint8_t* lut256, void* stream);

/* Volume layouts.  The phantom arrives as uint8 material ids, C order [nz][ny][nx] (x fastest).
 * dexct_volume_layouts writes the two layouts the traversal kernels read:
 *   vol_yx  [nz][ny][nx]   (a straight copy; minor axis x contiguous; used by y-dominant rays)
 *   vol_xy  [nz][nx][ny]   (in-plane transpose; minor axis y contiguous; x-dominant rays)
 *   vol_zf  [ny][nx][nz]   (z fastest; used by the row-parallel kernel)          (may be NULL)
 * Each destination holds nx*ny*nz bytes. */
int dexct_volume_layouts(const uint8_t* vol, int32_t nx, int32_t ny, int32_t nz, uint8_t* vol_xy,
                         uint8_t* vol_zf, void* stream);

/* Plans for views [view_begin, view_end): plan[(view - view_begin) * n_channels + channel].
 * view_cs[2*view + {0,1}] = cos, sin of the source angle; chan_cs[2*channel + {0,1}] = cos, sin of
 * the fan angle of the channel (float64, computed once on the host so that every
 * implementation starts from the same numbers). */
int dexct_fan_plan(const dexct_fan_geom* geom, const double* view_cs, const double* chan_cs,
                   int32_t view_begin, int32_t view_end, dexct_ray_plan* plan, void* stream);

/* Siddon forward projection + polychromatic detection: replaces get_sino (main.py:120).
 *   mu[m*n_energies + e]      linear attenuation [1/cm] of material id m at energy bin e
 *   weights[s*n_energies + e] effective spectrum of spectrum s (I0 * detector response * dE,
 *                             the weighting of matdecomp.py:146-150)
 *   counts[s*n_rays + ray]    (float32), n_rays = n_local_views*n_rows*n_channels
 *   pathlen (optional, may be NULL): [ray][n_materials] float32 path length [cm] per material
 *   layout 0: ray = ((view - view_begin)*n_rows + row)*n_channels + channel   (the reference's
 *             [N_proj, N_channels] order per row; native to the ray-parallel kernel)
 *   layout 1: ray = ((view - view_begin)*n_channels + channel)*n_rows + row   (row fastest; native to
 *             the row-parallel kernels, which then store 16 B per lane; dexct_transpose_batched
 *             converts between the two)
 *   weights2 / variance (both NULL or both given): weights2[s*n_energies + e] = weights * (detector signal
 *             per photon); variance[s*n_rays + ray] = sum_e weights2 * exp(-...) is the variance of the
 *             detected signal under per-bin Poisson statistics, input of dexct_add_noise
 * vol_yx / vol_xy / vol_zf as written by dexct_volume_layouts (vol_zf may be NULL: then the
 * ray-parallel kernel is used for every shape).
 * kernel: 0 = choose, 1 = ray-parallel (one thread per ray), 2 = row-parallel (one workgroup
 * per (view, channel), one detector row per lane), 3 = row-parallel with 4 rows per lane and packed
 * integer counts (needs vol_zf, 2..4 materials, nz and z_first multiples of 4 - the Python shim pads the
 * uploaded volume with empty slices to get there).
 * Precondition: every voxel id is < n_materials (ids outside are ignored by kernels 1 and 2 and
 * give unspecified - but memory-safe - results in kernel 3). */
int dexct_siddon_project(const dexct_fan_geom* geom, const dexct_ray_plan* plan, int32_t view_begin,
                         int32_t view_end, const uint8_t* vol_yx, const uint8_t* vol_xy,
                         const uint8_t* vol_zf, int32_t n_materials, int32_t n_energies,
                         int32_t n_spectra, const float* mu, const float* weights, float* counts,
                         float* pathlen, int32_t kernel, int32_t layout, const float* weights2, float* variance,
                         const dexct_log_out* log_out, void* stream);

/* Material groups: 5..DEXCT_MAX_MATERIALS materials (stacked fan, nz and z_first multiples of 4).
 * dexct_volume_groups: codes[g][voxel] for g < ceil((n_materials-1)/3): ids 3g+1..3g+3 of the z-fastest
 *   volume -> 1..3, all other ids -> 0 (n_groups * n_voxels bytes).
 * dexct_siddon_project_grouped: one packed-count traversal per group writes raw per-material accumulators
 *   to acc_scratch[m*n_rays + ray] (float32, caller-provided, n_materials*n_rays values), then one
 *   detection pass forms material 0 from the chord and applies the tables; outputs as dexct_siddon_project.
 *   Bit-identical path lengths to the single-pass kernels (per-material sums are independent).
 *   noise (ABI 6, the three group entry points; struct dexct_noise): with weights2 and noise->sample the detection pass sums the
 *   variance with the signal and draws the sample itself (<= 2 spectra, <= 48 materials, else DEXCT_ERANGE: use the variance
 *   output + dexct_add_noise, which draws the same sample); variance is then optional and log_out receives the log of the sampled
 *   counts.  weights2 without a sample needs the variance output (and admits no log_out). */
int dexct_volume_groups(const uint8_t* vol_zf, int64_t n_voxels, int32_t n_materials, uint8_t* codes, void* stream);
int dexct_siddon_project_grouped(const dexct_fan_geom* geom, const dexct_ray_plan* plan, int32_t view_begin,
                                 int32_t view_end, const uint8_t* codes, int32_t n_materials, int32_t n_energies,
                                 int32_t n_spectra, const float* mu, const float* weights, float* counts,
                                 float* pathlen, float* acc_scratch, int32_t layout, const float* weights2,
                                 float* variance, const dexct_log_out* log_out, const dexct_noise* noise, void* stream);

/* The stacked-fan projection on a 2-BIT PACKED volume (rows16_kernel; <= 4 materials, i.e. ids 0..3): a voxel is 2
 * bits, one dword load serves 16 detector rows, the per-row material counts are kept bit-sliced (carry-save adders).
 * dexct_volume_pack2: vol_zf [ny][nx][nz] bytes -> vol_z2 [ny][nx][nz/4] bytes (row z of a column in bits
 *   2(z%4).. of byte z/4); n_voxels = nx*ny*nz, a multiple of 4.
 * dexct_siddon_project_packed: same outputs and layouts as dexct_siddon_project; nz and z_first multiples of 16;
 *   nx, ny <= 2047.  A (view, channel) pair occupies ceil(n_rows/16) lanes of a 16-, 32- or 64-lane group (lanes past
 *   the last row idle: efficient for n_rows near 256, 512 or a multiple of 1024).  Bit-identical per-material path
 *   lengths.  weights2 / variance / noise (ABI 6): see struct dexct_noise - the noisy scan on this kernel (the variance rides
 *   in the detection rounds, the sample is drawn in registers: no variance array, no sampling pass; same bits as
 *   dexct_siddon_project(kernel 3, variance) + dexct_add_noise). */
int dexct_volume_pack2(const uint8_t* vol_zf, int64_t n_voxels, uint8_t* vol_z2, void* stream);
int dexct_siddon_project_packed(const dexct_fan_geom* geom, const dexct_ray_plan* plan, int32_t view_begin,
                                int32_t view_end, const uint8_t* vol_z2, int32_t n_materials, int32_t n_energies,
                                int32_t n_spectra, const float* mu, const float* weights, float* counts,
                                float* pathlen, int32_t layout, const dexct_log_out* log_out, const float* weights2,
                                float* variance, const dexct_noise* noise, void* stream);

/* Material groups on the packed volume (5..DEXCT_MAX_MATERIALS materials; the packed form of dexct_volume_groups /
 * dexct_siddon_project_grouped, same arguments and outputs; preconditions of dexct_siddon_project_packed).
 * dexct_volume_groups_pack2: codes2[g][n_voxels / 4]: the group codes 0..3 of dexct_volume_groups at 2 bits per voxel.
 * dexct_siddon_project_grouped_packed: one rows16_kernel pass per group (no detection) into acc_scratch, then the
 *   detection pass of dexct_siddon_project_grouped.  Bit-identical to it. */
int dexct_volume_groups_pack2(const uint8_t* vol_zf, int64_t n_voxels, int32_t n_materials, uint8_t* codes2, void* stream);
int dexct_siddon_project_grouped_packed(const dexct_fan_geom* geom, const dexct_ray_plan* plan, int32_t view_begin,
                                        int32_t view_end, const uint8_t* codes2, int32_t n_materials, int32_t n_energies,
                                        int32_t n_spectra, const float* mu, const float* weights, float* counts,
                                        float* pathlen, float* acc_scratch, int32_t layout, const float* weights2,
                                        float* variance, const dexct_log_out* log_out, const dexct_noise* noise, void* stream);

/* Cone-beam (3-D) projection, SURVEY 8f.4: the fan of dexct_fan_plan in the (x, y) plane, source at height
 * src_z, detector row r at height row_z[r] (device float64 [n_rows], cm, z = 0 at the centre of the grid;
 * geom->z_first is ignored).  max_abs_dz = max_r |row_z[r] - src_z| (the caller knows it; it bounds the z
 * slope: at most one z-plane may be crossed per dominant-axis slab, else DEXCT_ERANGE).  One thread per ray;
 * counts / pathlen in layout 0 of dexct_siddon_project; every material is accumulated directly.
 * weights2 / variance / noise (ABI 6, both cone entry points): struct dexct_noise - the noisy scan in the same launch (any
 * number of spectra); variance [s*n_rays + ray] in the order of counts. */
int dexct_cone_project(const dexct_fan_geom* geom, const dexct_ray_plan* plan, const double* view_cs,
                       const double* chan_cs, const double* row_z, double src_z, double max_abs_dz,
                       int32_t view_begin, int32_t view_end, const uint8_t* vol_yx, const uint8_t* vol_xy,
                       int32_t n_materials, int32_t n_energies, int32_t n_spectra, const float* mu,
                       const float* weights, float* counts, float* pathlen, const dexct_log_out* log_out,
                       const float* weights2, float* variance, const dexct_noise* noise, void* stream);

/* The same projection with the ROWS of one (view, channel) pair as lanes: the in-plane slab records are computed once
 * per pair and shared by all its rows, a lane carries only its z DDA.  <= 3 materials.  Round 3 (cone_cols_kernel): the
 * workgroup stages the two voxel columns of each slab of a batch in LDS and the lanes read their bytes from there
 * (volumes of up to 1024 slices; cone_rows_kernel, one byte load per lane and voxel, beyond that and with
 * DEXCT_CONE_COLS=0).  It reads the guarded z-fastest layout written by dexct_cone_layout: a column is
 * zs = ((nz + 15) & ~15) + 32 bytes, vol_zc[(y*nx + x)*zs + 16 + z] = 8 * id, every guard byte and one extra column
 * hold 24 (id 3 = "outside the grid"); dexct_cone_layout_bytes gives its size.  Same outputs, bit-identical
 * per-material path lengths.  Tuning / A-B knobs (environment, read per call): DEXCT_CONE_KB=8 (slabs per staged
 * batch, default 4), DEXCT_CONE_VIEW_TILE (views per tile of the block order, default 1). */
int64_t dexct_cone_layout_bytes(int32_t nx, int32_t ny, int32_t nz);
int dexct_cone_layout(const uint8_t* vol, int32_t nx, int32_t ny, int32_t nz, uint8_t* vol_zc, void* stream);
int dexct_cone_project_rows(const dexct_fan_geom* geom, const dexct_ray_plan* plan, const double* view_cs,
                            const double* chan_cs, const double* row_z, double src_z, double max_abs_dz,
                            int32_t view_begin, int32_t view_end, const uint8_t* vol_zc, int32_t n_materials,
                            int32_t n_energies, int32_t n_spectra, const float* mu, const float* weights,
                            float* counts, float* pathlen, const dexct_log_out* log_out, const float* weights2,
                            float* variance, const dexct_noise* noise, void* stream);

/* Material groups for the row-parallel cone kernels (ABI 6; more than 3 materials, up to DEXCT_MAX_MATERIALS; volumes of any
 * height the row kernels take): what dexct_volume_groups / dexct_siddon_project_grouped are to the stacked fan.
 * dexct_cone_layout_groups: ceil(n_materials / 3) guarded layouts of dexct_cone_layout_bytes each; layout g holds 8 * (id - 3g)
 *   for the ids 3g .. 3g + 2 and 24 ("outside": nobody's voxel) for every other id.
 * dexct_cone_project_grouped: one traversal per group (cone_cols_kernel / cone_rows_kernel) writes the path lengths [cm] of its
 *   three materials to acc_scratch[m*n_rays + ray] (float32, caller-provided, n_materials*n_rays values), then one detection pass
 *   applies the tables; outputs as dexct_cone_project.  The per-material sums are independent, so the path lengths are
 *   bit-identical to dexct_cone_project's.  weights2 / variance / noise: as dexct_siddon_project_grouped. */
int dexct_cone_layout_groups(const uint8_t* vol, int32_t nx, int32_t ny, int32_t nz, int32_t n_materials, uint8_t* vol_zcg,
                             void* stream);
int dexct_cone_project_grouped(const dexct_fan_geom* geom, const dexct_ray_plan* plan, const double* view_cs,
                               const double* chan_cs, const double* row_z, double src_z, double max_abs_dz,
                               int32_t view_begin, int32_t view_end, const uint8_t* vol_zcg, int32_t n_materials,
                               int32_t n_energies, int32_t n_spectra, const float* mu, const float* weights, float* counts,
                               float* pathlen, float* acc_scratch, const dexct_log_out* log_out, const float* weights2,
                               float* variance, const dexct_noise* noise, void* stream);

/* counts = max(counts + sqrt(variance) * z, 1e-20), z ~ N(0, 1) from Philox4x32-10 with counter (view_offset + view, row,
 * channel, 0) and key seed - one block per detector pixel, spectrum s takes the s-th normal of its two Box-Muller pairs (ABI 6;
 * struct dexct_noise: the projection kernels that sample by themselves draw exactly this) - : independent of view sharding and
 * of the layout (0 / 1 as above).  The clip keeps a log sinogram finite. */
int dexct_add_noise(float* counts, const float* variance, int32_t n_spectra, int32_t n_views, int32_t n_rows,
                    int32_t n_channels, int32_t layout, int32_t view_offset, uint64_t seed, void* stream);

/* sino_log[s*n_rays + ray] = ln(air[s] / counts[s*n_rays + ray]) as a pass of its own (air: n_spectra HOST floats): for
 * noisy sinograms, whose counts exist only after dexct_add_noise / dexct_poisson_detect, and for counts that were not
 * projected in this call.  Same arithmetic as the fused form (dexct_log_out). */
int dexct_sino_log(const float* counts, const float* air, int32_t n_spectra, int64_t n_rays, float* sino_log, void* stream);

/* Exact quantum noise: counts[s][ray] = sum_e gain[e] * Poisson(photons[s][e] * exp(-sum_m mu[m][e] * pathlen[ray][m])),
 * from the per-material path lengths a projection wrote (pathlen [ray][n_materials], cm, same ray order =
 * `layout`).  photons = I0 * eta * dE (detected photons per bin), gain = E for an energy-integrating detector,
 * 1 for a counting one.  Philox counter (view_offset + view, row, channel, spectrum/energy): shard- and
 * layout-independent.  Inversion below lambda = 30, rounded normal above. */
int dexct_poisson_detect(const float* pathlen, const float* mu, const float* photons, const float* gain,
                         int32_t n_materials, int32_t n_energies, int32_t n_spectra, int32_t n_views, int32_t n_rows,
                         int32_t n_channels, int32_t layout, int32_t view_offset, uint64_t seed, float* counts,
                         void* stream);

/* dst[b][c][r] = src[b][r][c] for b < batch: [batch][rows][cols] -> [batch][cols][rows], elements of
 * elem_bytes = 4, 8 or 16 bytes (float32 sinograms, float64, the (a0, a1) float64 pairs of the
 * decomposition).  src and dst must not overlap. */
int dexct_transpose_batched(const void* src, void* dst, int64_t batch, int32_t rows, int32_t cols,
                            int32_t elem_bytes, void* stream);

/* The two outputs of get_sino (main.py:120-122) in the reference's [view][row][channel] order from the row-parallel kernels'
 * native [view][channel][row] counts, in one pass: dst[b][c][r] = src[b][r][c] for b < n_spectra * batch_per_spectrum
 * (float32), and - log_dst not NULL - log_dst[b][c][r] = ln(air[b / batch_per_spectrum] / dst[b][c][r]) with the arithmetic
 * of dexct_log_out / dexct_sino_log (air: n_spectra HOST floats).  The counts are read once and written twice; the
 * projection then needs no log output of its own.  Any shape (16-byte accesses when rows and cols are multiples of 4 and
 * the pointers are aligned).  src must not overlap dst / log_dst. */
int dexct_transpose_log(const float* src, float* dst, float* log_dst, const float* air, int32_t n_spectra,
                        int64_t batch_per_spectrum, int32_t rows, int32_t cols, void* stream);

/* Trace of single rays for parity tests: voxel-index sequence and segment lengths.
 * For each of n_rays rays (view, row, channel given in ray_vrc[3*r + {0,1,2}], view relative to
 * the plan's view_begin) writes up to max_seg segments: seg_voxel[r*max_seg + k] = linear voxel
 * index (z*ny + y)*nx + x, seg_len[r*max_seg + k] = length in units of u (float32, exactly the
 * numbers the projection kernels accumulate), n_seg[r] = number of segments produced. */
int dexct_siddon_trace(const dexct_fan_geom* geom, const dexct_ray_plan* plan, const int32_t* ray_vrc,
                       int32_t n_rays, int32_t max_seg, int32_t* seg_voxel, float* seg_len,
                       int32_t* n_seg, void* stream);

/* Assembling the sinogram of a view-sharded scan (one all-gather over xGMI): rank r passes its
 * count_per_rank float32 values (equal on all ranks: pad the last shard) and receives the shards of all ranks in
 * rank order in gathered[world * count_per_rank]; in place when local == gathered + rank*count_per_rank.
 * rccl_comm is the caller's ncclComm_t (one process per GPU).  The library does not link RCCL; it uses the copy
 * already loaded in the process, else librccl.so.1.  The Python host reaches the same ncclAllGather through
 * torch.distributed (dex-ct-sim_amd/_shard.py). */
int dexct_sino_allgather(const float* local, float* gathered, int64_t count_per_rank, void* rccl_comm, void* stream);

/* The same assembly as point-to-point transfers (ABI 5; SURVEY section 8e: "each GPU has a direct link to every peer, so
 * gather-to-root uses 7 links concurrently"): rank r's shard is counts[r] float32 values and belongs at gathered + offsets[r]
 * (shards may differ in size: no padding).  root >= 0: the gather of BASELINE.json's north star - every other rank sends its
 * shard to `root`, which receives them in place (`gathered` may be NULL on the other ranks); root = -1: every rank sends its
 * shard to every peer and receives every peer's (an all-gather made of world-1 sends and receives per rank, each pair over its
 * own xGMI link).  One ncclGroupStart / ncclGroupEnd around all of them; the rank's own shard is copied on the device unless
 * local == gathered + offsets[rank].  The Python host reaches the same transfers through torch.distributed
 * (dex-ct-sim_amd/_shard.py, gather_views(mode='root' | 'direct')). */
int dexct_sino_gather(const float* local, float* gathered, const int64_t* counts, const int64_t* offsets, int32_t rank,
                      int32_t world, int32_t root, void* rccl_comm, void* stream);

/* The host boundary of the drop-in calls (NumPy out, main.py:153-155; ABI 5): lock `n_bytes` of the caller's host array at
 * `host` for DMA (hipHostRegister; from any thread: `device` = the process's HIP device), copy device memory into it on a
 * stream, unlock it again.  A result array is locked piece by piece while the kernels of the pieces before it run
 * (dex-ct-sim_amd/matdecomp.py, _basismat_sinos_pipelined): a first call does not wait for one large page-locked allocation. */
int dexct_host_pin(void* host, int64_t n_bytes, int32_t device);
/* make the pages of [host, host + n_bytes) resident with `threads` (1..64) threads; the bytes are left as they are */
int dexct_host_touch(void* host, int64_t n_bytes, int32_t threads);
int dexct_host_unpin(void* host, int32_t device);
int dexct_download(void* host, const void* device_src, int64_t n_bytes, void* stream);

/* Options of dexct_gn_decompose (ABI 3; pass, iterations, start: ABI 4; flags, blocks_per_cu: ABI 5).  A NULL pointer = every
 * default.
 *   stop_tol    >= 0: taken as given.  0 = the reference's fixed iteration count, bit for bit (matdecomp.py:114: `for
 *               i in range(n_iters)`).  > 0 = tolerance stop: a float64 pixel also ends, with the state after the step, when
 *               the distance it still has to go - estimated from its last two steps d_k < d_(k-1) as d_k r / (1 - r), r =
 *               d_k / d_(k-1) - is at most stop_tol / 4 * max(|a0|, |a1|, 1), the step before contracted as well (d_(k-1) <
 *               d_(k-2)) and the estimate also holds with the ratio of THAT step (the worse of the two ratios counts): a wandering
 *               pixel's accidental tiny step does not end it, a creeping one is not trusted after one good step
 *               A creeping (r near 1) or wandering pixel is not stopped and runs to n_iters as in the reference.
 *               < 0: the library default = DEXCT_GN_DEFAULT_STOP_TOL (1e-12: seven orders inside the 1e-5 the results
 *               are specified to, three inside the 1e-9 the kernel keeps to the reference's own outputs), or the value of
 *               the environment variable DEXCT_GN_STOP_TOL, or 0 when DEXCT_GN_EXACT=1 (read once per process).
 *   out_rows, out_channels   both 0: out_a[2*p + m] in the order of the pixels.  Both > 0 (n_pix a multiple of their
 *               product): the pixels are given as [..][channel][row] (row fastest - layout 1 of dexct_siddon_project, what
 *               the stacked-fan kernels write) and the results are written as [..][row][channel], the reference's order
 *               (matdecomp.py:200-201): out_a[2*((v*out_rows + r)*out_channels + c) + m] for pixel (v*out_channels + c)*
 *               out_rows + r.  The float64 shared-spectrum kernel collects 4 x 16 (channel, row) tiles in LDS and writes
 *               them as 64-byte runs; no separate transpose pass over the results.
 *   kernel      0 = choose by size; 1 = one lane per pixel (gn_refill_kernel); 2 = cooperative: the four waves of a
 *               workgroup split the energies of 64 pixels (gn_coop_kernel, for sinograms too small to fill the chip with
 *               one pixel per lane).  float64, n_bins == 1 only; ignored otherwise.
 *   pass, iterations, start   the short cut.  What the reference returns is the fixed point its walk from 1e-6 ends at, and
 *               most of its ~17 Newton steps per pixel are that walk.  The end of the walk is a function of the pixel's two
 *               counts alone; the caller tabulates it once per pair of spectra - by running THIS entry point (pass =
 *               DEXCT_GN_PASS_COUNT: the reference's iteration, with step counts) on a grid of counts - and hands the table over
 *               as `start` (the Python host: dex-ct-sim_amd/quadrature.py newton_start_grid / assemble_start,
 *               matdecomp._device_tables).
 *                 pass = DEXCT_GN_PASS_COUNT, iterations != NULL, start = NULL: the usual iteration from 1e-6; besides the
 *                   results, iterations[q] (q = the pixel's index in the RESULT order, n_pix bytes) receives the number of steps
 *                   after which the tolerance rule ended the pixel, or 255 when it ended any other way.
 *                 pass = DEXCT_GN_PASS_SHORTCUT, start != NULL, iterations = NULL (what the host runs by default): a pixel in a
 *                   cell with n_iters >= need starts from the 6 x 6 Lagrange interpolant of the corners' fixed points (the cell's
 *                   corners and two rings around them: cells within two of the grid's border must be closed) and ends by
 *                   the tolerance rule of the FULL tables (two steps; or at a repeated state), accepted only within the radius;
 *                   if any of this fails - and for every pixel in a closed cell - the pixel is solved from the reference's
 *                   start value with all n_iters steps, as a plain call does.  The call therefore returns, per pixel, either
 *                   a fixed point of the full model on the reference's branch verified to stop_tol, or the reference's own
 *                   trajectory; the table only decides how fast, never what.
 *               Layout of `start` (doubles, 16-byte aligned): [0],[1] the unattenuated signals sum_e i0[k][e]; [2] a scale s;
 *               [3] cells per axis n; [4] ln of the smallest u0 of the grid; [5] cells per unit of ln u0; [6] the smallest ratio
 *               u1 / u0 of the grid; [7] cells per unit of the ratio; [8],[9] ln of [0],[1]; [10] 1 if the kappa array is present,
 *               else 0; [11] reserved; where u_k = s ln(start[k] / g_k);
 *               then the fixed points at the (n+1)^2 cell corners as pairs (a0, a1) (row = index along ln u0); then per cell
 *               the pair (need, radius): need = the number of steps a pixel whose counts fall in the cell must be allowed for
 *               the reference's walk to be known to end by the tolerance rule (infinity: closed), radius = how far from the
 *               interpolated fixed point a result is accepted; then (optional, see [10] and DEXCT_GN_FLAG_ONE_STEP) per cell
 *               kappa: the step DEXCT_GN_FLAG_ONE_STEP takes - of the Gauss-Newton form, below - of length d1 from the
 *               interpolant lands within kappa d1^2 of the fixed point (infinity: always two steps).
 *               Both passes need stop_tol > 0 (after defaults), n_bins == 1, precision 0, n_iters <= 254, kernel != 2;
 *               DEXCT_EINVAL otherwise.  pass = 0 (default): one launch from 1e-6; iterations and start must be NULL or are
 *               not used.  (ABI 4 also had a two-launch "coarse" form of the short cut - a launch on a short quadrature of the
 *               spectra in between; it was 15 % slower than this one and left the library in ABI 5.)
 *   flags       DEXCT_GN_FLAG_FULL_LOOP: execute every iteration, no exit of any kind (stop_tol is then 0): the check that the
 *               repeated-state exit changes no bit.  DEXCT_GN_FLAG_NATURAL_ORDER: hand the tiles of a small sinogram out in
 *               their natural order instead of thick tiles first (results do not depend on it).
 *               DEXCT_GN_FLAG_ONE_STEP (DEXCT_GN_PASS_SHORTCUT only; `start` must end with the kappa array, below): a pixel whose
 *               FIRST step from the interpolated fixed point has length d1 with kappa d1^2 <= stop_tol / 4 * max(|a|, 1) ends
 *               there - kappa bounds what the step leaves of a distance d1 (from the Hessian and the third derivatives of
 *               the likelihood at the tabulated fixed points) - ; every other pixel goes on to its second step and the
 *               tolerance rule.  That first step is of the GAUSS-NEWTON form: the Hessian of matdecomp.py:123 without its
 *               (g / nu - 1) x second-derivative term (half the sums per energy); the term vanishes with the distance to a
 *               fixed point that reproduces its counts, what it leaves is second order and must be part of kappa
 *               (dex-ct-sim_amd/quadrature.py, newton_kappa(gauss_newton=True)).  All further steps are full Newton steps.
 *   blocks_per_cu   > 0: workgroups per CU of the queue kernels (0: what is resident; results do not depend on it). */
#define DEXCT_GN_DEFAULT_STOP_TOL 1e-12
#define DEXCT_GN_PASS_COUNT 1
#define DEXCT_GN_PASS_SHORTCUT 2
#define DEXCT_GN_FLAG_FULL_LOOP 1
#define DEXCT_GN_FLAG_NATURAL_ORDER 2
#define DEXCT_GN_FLAG_ONE_STEP 4
typedef struct dexct_gn_options {
  double stop_tol;
  int32_t out_rows, out_channels;
  int32_t kernel;
  int32_t pass;                /* 0, DEXCT_GN_PASS_COUNT, DEXCT_GN_PASS_SHORTCUT */
  uint8_t* iterations;         /* device, n_pix bytes; DEXCT_GN_PASS_COUNT only */
  const double* start;         /* device; DEXCT_GN_PASS_SHORTCUT only: the table of the reference's fixed points (above) */
  int32_t flags;               /* DEXCT_GN_FLAG_* (ABI 5) */
  int32_t blocks_per_cu;       /* (ABI 5) */
} dexct_gn_options;

/* Per-pixel Newton (Gauss-Newton) basis-material decomposition: replaces optimize_sino_cpu
 * (matdecomp.py:87-127).
 *   g1, g2     measured counts of the two spectra, n_pix values each; g_is_f64 selects
 *              float64 (1) or float32 (0) input
 *   i0[(k*n_bins + b)*n_energies + e]  effective spectra (float64), k = 0, 1 - the reference's
 *              [nMeas, nBins, nEnergies] layout.  n_bins = 1: one spectrum for every pixel (all that
 *              do_matdecomp_gn builds, matdecomp.py:151; the fast path: tables through the scalar cache);
 *              n_bins > 1: pixel p uses row b = (p / bin_div) % n_bins (bin_div = 1 when the channel is
 *              the fastest pixel index, = n_rows for the row-fastest layout 1)
 *   mus[m*n_energies + e]  basis mass attenuation (float64), m = 0, 1
 *   out_a[2*p + m]         density line integrals (float64), initialised to 1e-6 inside; 16-byte aligned (DEXCT_EINVAL
 *                          otherwise): results leave as 16-byte pieces of whole-line stores; order: see dexct_gn_options
 *   precision: 0 = float64 throughout (reference arithmetic);
 *              1 = float32 bulk iterations followed by float64 polish iterations; a pixel the
 *                  polish is still moving is redone in float64 from the start (n_bins == 1 only)
 *   n_polish   number of trailing float64 iterations when precision == 1
 *   mask_max   NULL, or a device float64 scalar (e.g. from dexct_reduce_max, all-reduced over ranks): pixels with
 *              g1 >= mask_frac * *mask_max are the air pixels get_basismat_sinos zeroes afterwards
 *              (matdecomp.py:195-196, :204-205); they get (0, 0) directly and their iterations are skipped
 *   options    see dexct_gn_options; NULL = defaults
 *   workspace  device scratch of dexct_gn_workspace_bytes(n_energies, n_bins) bytes (the product tables
 *              the kernel reads through the scalar cache); owned by the caller, no hidden state.  After the call
 *              the uint64 at byte offset 72 holds, as a diagnostic, the number of pixel-iterations the float64
 *              shared-spectrum kernels executed (what bench.py's executed-flop rate is computed from; 0 for the
 *              other kernels), and the uint64 at byte offset 80 the number of pixels handed to waves so far -
 *              updated tile by tile WHILE the kernel runs, so a host thread may read it (on another stream) as a progress
 *              indicator: the reference prints a line every 20 views, matdecomp.py:111-112.  The uint64 at byte offset 88
 *              is the head of the tile queue, the one at 96 counts lane-steps that had to wait for a free result slot
 *              (diagnostic); the library zeroes all four words at the start of every call
 * n_iters is the reference's iteration count.  The update is a pure function of the two doubles, so the
 * kernel stops a pixel at the first state that repeats bit for bit (fixed point or cycle of up to 9 states) and
 * returns the state the cycle holds at iteration n_iters: the result of all n_iters iterations, exactly; with a tolerance
 * stop (the default, see dexct_gn_options) a converging pixel ends a few iterations earlier still.
 * Environment (read ONCE per process, at the first call; tuning only - what a call computes is decided by its arguments):
 * DEXCT_GN_EXACT=1 / DEXCT_GN_STOP_TOL=<t> set the DEFAULT tolerance (an explicit options->stop_tol >= 0 wins; a value that is not
 * a number >= 0 is ignored); DEXCT_GN_FULL_LOOP=1 = DEXCT_GN_FLAG_FULL_LOOP on every call; DEXCT_GN_BLOCKS_PER_CU=<n> = the
 * default of options->blocks_per_cu; DEXCT_GN_COOP_BELOW=<pixels> moves the size below which the cooperative kernel runs;
 * DEXCT_GN_SORT=0 = DEXCT_GN_FLAG_NATURAL_ORDER on every call; DEXCT_GN_TILES_PER_FETCH=<n> queue positions a wave reserves per
 * atomic. */
int64_t dexct_gn_workspace_bytes(int32_t n_energies, int32_t n_bins);
int dexct_gn_decompose(const void* g1, const void* g2, int32_t g_is_f64, int64_t n_pix, const double* i0,
                       const double* mus, int32_t n_energies, int32_t n_bins, int32_t bin_div, int32_t n_iters,
                       int32_t precision, int32_t n_polish, const double* mask_max, double mask_frac, double* out_a,
                       const dexct_gn_options* options, void* workspace, void* stream);

/* The energy sums of the decomposition's forward model (matdecomp.py:116-121) at n_states states a [n][2] (device float64), for the
 * table assembly of the short cut (quadrature.py, assemble_start / validate_start; no counterpart in the reference): nu_out
 * [n][2] = sum_e i0[k][e] exp(clip(-(a0 mu0[e] + a1 mu1[e]), +-700)); g_out [n][2][2] = sum_e i0[k][e] mu[m][e] x the same
 * exponential over the energies whose exponent is not clipped; s_out [n][2][2][2] (may be NULL) = sum_e i0[k][e] mu[m][e]
 * mu[p][e] likewise.  i0 [2][n_energies], mus [2][n_energies] device float64 (one shared spectrum per measurement). */
int dexct_gn_model_sums(const double* a, int64_t n_states, const double* i0, const double* mus, int32_t n_energies,
                        double* nu_out, double* g_out, double* s_out, void* stream);

/* Air mask of get_basismat_sinos (matdecomp.py:194-205): out_a[2p], out_a[2p+1] = 0 wherever
 * g1[p] >= thresh_value (thresh_value = mask_thresh * global max, computed by the caller so that a
 * sharded run can all-reduce the max first). */
int dexct_gn_apply_mask(const void* g1, int32_t g_is_f64, int64_t n_pix, double thresh_value,
                        double* out_a, void* stream);

/* max(g1) over n_pix values into *out_max (device float64 scalar). */
int dexct_reduce_max(const void* g1, int32_t g_is_f64, int64_t n_pix, double* out_max, void* stream);

/* Fan-beam filtered back-projection: replaces get_recon (main.py:134,168; x-tomo-sim back_project.py,
 * absent; README.md:30-31).  Two steps on the log sinogram [n_lines = views*rows][n_channels]:
 *   dexct_fbp_filter:      q[line][n] = dgamma * sum_m sino[line][m] * weight[m] * taps[(n - m) + n_channels - 1]
 *                          (weight[m] = SID cos(gamma_m); taps = 2*n_channels - 1 equiangular ramp taps)
 *   dexct_fbp_backproject: image[row][iy][ix] = dbeta * sum_views q(view, row, gamma'(x, y)) / L^2, pixel
 *                          driven, linear interpolation; q is [view][row][channel]; image float32 in 1/cm.
 * (A 2 pi scan; a short scan first goes through dexct_fbp_parker.) */
/* Short scans (rotation_angle_total < 2 pi, input/params.txt:24): out = 2 w(beta, gamma) * sino with Parker's weights for a
 * scan over theta_tot (views at beta_v = theta_tot * (view_offset + v) / n_views_total, channels at gamma_c = (c - (n_channels
 * - 1) / 2) * dgamma), so that dexct_fbp_filter + dexct_fbp_backproject (dbeta = theta_tot / n_views_total) reconstruct it.
 * theta_tot must cover pi + the fan angle (DEXCT_EINVAL otherwise: data are missing) and be below 2 pi; sino and out
 * [n_views][n_rows][n_channels] float32, may be the same buffer. */
int dexct_fbp_parker(const float* sino, int32_t n_views, int32_t n_rows, int32_t n_channels, double theta_tot, double dgamma,
                     int32_t view_offset, int32_t n_views_total, float* out, void* stream);
int dexct_fbp_filter(const float* sino, const float* taps, const float* weight, int64_t n_lines,
                     int32_t n_channels, double dgamma, float* q, void* stream);
int dexct_fbp_backproject(const float* q, const double* view_cs, int32_t n_views, int32_t n_channels,
                          int32_t n_rows, double sid, double dgamma, double dbeta, int32_t n_matrix, double fov,
                          float* image, void* stream);

/* Cone-beam (Feldkamp) back-projection for the detector of dexct_cone_project: q [view][row][channel] filtered row by
 * row with dexct_fbp_filter (same weights and taps as the fan); row r at height row_z0 + r*row_dz [cm] on the
 * detector cylinder, row_weight[r] = cos(cone angle of row r) = SDD / sqrt(SDD^2 + (row_z[r] - src_z)^2);
 * image [slice][iy][ix] float32 in 1/cm, slice k at z0 + k*dz, z = 0 at the centre of the phantom grid like src_z.
 * (An extension: the reference is fan-beam only.) */
int dexct_fdk_backproject(const float* q, const double* view_cs, const float* row_weight, int32_t n_views,
                          int32_t n_channels, int32_t n_rows, double sid, double sdd, double dgamma, double dbeta,
                          double row_z0, double row_dz, double src_z, int32_t n_matrix, double fov, int32_t n_slices,
                          double z0, double dz, float* image, void* stream);

/* Virtual monoenergetic image (plots.py:136-144): out = u1*m1 + u2*m2 (basis-material images m1, m2, n pixels;
 * u1, u2 mass attenuation of the basis materials at the chosen energy), in HU against u_water when hu != 0. */
int dexct_vmi(const float* m1, const float* m2, int64_t n, double u1, double u2, double u_water, int32_t hu,
              float* out, void* stream);

/* Per-label second-order moments of one or two images: the sufficient statistics of the reference's ROI
 * measurements (measure_roi, plots.py:146-158: mean and population variance of a rectangle) and of its VMI
 * sweeps (RMSE against ground truth plots.py:297-303, CNR :371-393), which are closed forms in these sums
 * because a VMI is linear in the two basis-material images.
 *   out[l][6] = { count, S m1, S m2, S m1^2, S m1*m2, S m2^2 } over the pixels i with labels[i] == l, float64;
 *   labels == NULL: every pixel is label 0; pixels with labels[i] >= n_labels are skipped; m2 == NULL: zeros.
 * n_labels <= 64.  out is zeroed by the call (on the stream).  Summation order is not fixed (float64 atomics). */
int dexct_label_moments(const float* m1, const float* m2, const uint8_t* labels, int64_t n, int32_t n_labels,
                        double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEXCT_H */
