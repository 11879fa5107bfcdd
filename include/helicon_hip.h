/*
 * helicon_hip.h — C ABI of libhelicon_hip.so: the MI355X (gfx950) implementation of the
 * denovo3D / HILL-style (twist, rise, Csym) sweep ("Path B" of SURVEY.md).
 *
 * The reference (jianglab/helicon) is pure Python and has no FFI for this path; what a
 * maintainer would bind is the three primitives the sweep composes and the thread-pool loop
 * that drives them.  Each entry point cites the reference interface it replaces
 * (paths relative to the reference's src/helicon/):
 *
 *   hh_simulate            webApps/denovo3D/utils.py:31-189   simulate_helical_projection
 *   hh_power_spectrum      lib/transforms.py:771-820          compute_power_spectra (defaults)
 *   hh_cross_correlation   lib/analysis.py:777-799            cross_correlation_coefficient
 *   hh_cosine_similarity   lib/analysis.py:802-821            cosine_similarity
 *   hh_apply_helical_symmetry  lib/transforms.py:58-165       apply_helical_symmetry
 *   hh_affine_transform_2d lib/transforms.py:315-369          rotate_shift_image (its scipy.ndimage.affine_transform call)
 *   hh_transform_map       lib/transforms.py:168-235          transform_map (Euler ZYZ + scipy.ndimage.map_coordinates, cubic)
 *   hh_warp_affine_2d      lib/transforms.py:238-312          transform_image (its skimage.transform.warp call)
 *   hh_rescale_2d          lib/filters.py:375-412             down_scale, and the app's binning (their skimage rescale call)
 *   hh_helix_moments       lib/analysis.py:645-728            estimate_helix_rotation_center_diameter (closing + moments)
 *   hh_ssim_2d,            lib/analysis.py:487-613            ssim_score / ms_ssim_score / mutual_information_score: the non-cosine
 *   hh_joint_histogram                                        scores of lsq_reconstruct (solver_linear_regression.py:484-524)
 *   hh_set_reference +     webApps/denovo3D/app.py:2455-2523  reconstruction_task (the pool over
 *   hh_sweep[_device]                                         candidates) scoring each candidate by
 *                                                             cc(ref[mask], pwr[mask])
 *                                                             (lib/alignment.py:144-147 idiom)
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success and a
 * negative hh_status otherwise, with a message available from hh_last_error(); nothing calls
 * exit(), and no C++ exception leaves the library: every entry point with a body of its own is a function-try-block
 * that turns std::bad_alloc / std::length_error into HH_ERR_NOMEM and anything else into HH_ERR_INTERNAL (message =
 * what()); no host pointer is retained after a call returns.  Images are C-order float32, ny x nx, with the
 * helical axis along the columns (x); "N" below is the side of a square power-of-two image (the tuned kernels),
 * general sizes: hh_create2.
 * A context is bound to one device and is NOT thread-safe (serialise calls per context).
 */
#ifndef HELICON_HIP_H
#define HELICON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HH_ABI_VERSION 1
#define HH_MAX_UNITS 64

typedef enum hh_status {
  HH_OK = 0,
  HH_ERR_ARG = -1,     /* invalid argument / unsupported size                      */
  HH_ERR_HIP = -2,     /* a HIP runtime call failed                                */
  HH_ERR_STATE = -3,   /* call order (e.g. sweep before set_reference/geometry)    */
  HH_ERR_NOMEM = -4,   /* host or device allocation failed                         */
  HH_ERR_INTERNAL = -5 /* a C++ exception reached the boundary (message: what())   */
} hh_status;

typedef struct hh_ctx hh_ctx;

/* Lattice geometry shared by all candidates of a sweep: the scalar arguments of
 * simulate_helical_projection (utils.py:31-47) other than (twist, rise, csym, rot). */
typedef struct hh_geom {
  double apix;             /* Angstrom / pixel                                             */
  double helical_diameter; /* Angstrom; with n_units == 0 one unit sits at radius d/2      */
  double ball_radius;      /* Angstrom; sigma^2 = ball_radius^2 / ln 2 (utils.py:92)       */
  double tilt;             /* degrees, out-of-plane tilt   (utils.py:166-168)              */
  double psi;              /* degrees, in-plane rotation; applied as -psi like utils.py:167 */
  double dy;               /* Angstrom, shift along the image rows (utils.py:169-170)      */
  int32_t n_units;         /* 0 or 1: the deterministic single-unit branch (utils.py:145-151);
                              >1: units[] gives the asymmetric unit explicitly              */
  int32_t tail_bits;       /* Gaussian support is truncated where a term < 2^-tail_bits
                              (0 selects the default, 24)                                   */
  const double* units;     /* host, n_units x 3: (radius A, azimuth rad, axial offset A) — the
                              reference's r, angle and z draws of utils.py:139-144 before
                              `rot` is added; may be NULL when n_units <= 1                 */
} hh_geom;

/* Per-kernel device time of the sampled launches since the last hh_profile_reset: while
 * profiling is enabled with period k (hh_profile_enable(ctx, k), k >= 1) HIP events are recorded
 * on the context's stream around every launch of every k-th batch of a sweep.  Event records
 * break the back-to-back issue of kernels (about 12 % of a sweep at k = 1), hence the sampling. */
typedef struct hh_profile {
  double ms_first_pass;    /* first pass: k_first_pass / k_first_pass_table (fused: the
                              stand-alone k_column_factors of a sweep's first batch)   */
  double ms_second_pass;   /* k_second_pass, or k_fused_pass (build + row FFT + |F| +
                              log1p + masked moment reduction)                         */
  double ms_finalize;      /* Pearson from moments / several-segment contraction       */
  double ms_centres;       /* k_run_table (run tables of the shared-twist pipelines)   */
  int64_t n_first_pass;    /* launches                                                 */
  int64_t n_second_pass;
  int64_t n_finalize;
  int64_t n_centres;
  int64_t candidates;      /* candidates in the sampled launches                       */
} hh_profile;

int hh_abi_version(void);
int hh_device_count(int* count);
/* Exercises the exception barrier without a device: raises, inside a guarded entry point, an exception of the kind the
 * host code under the ABI can meet — kind 0: std::vector::resize with an absurd element count (std::length_error),
 * 1: an allocation no machine can serve (std::bad_alloc), 2: std::system_error as std::thread raises it, 3: a type
 * outside std::exception — and returns the status the barrier made of it (HH_ERR_NOMEM, HH_ERR_NOMEM, HH_ERR_INTERNAL,
 * HH_ERR_INTERNAL; message from hh_last_error(NULL)); any other kind returns HH_OK.  (The reference is Python: an
 * allocation failure there is a MemoryError the caller can catch, lib/exceptions.py:1-52 — this keeps that property.) */
int hh_selftest_exception(int kind);

/* device: HIP ordinal; n: image side; max_batch: candidates per launch (0 = default).
 * hh_create2 takes the image's rows and columns (utils.py:31-47 and transforms.py:687-704 accept any (ny, nx); the
 * helical axis runs along the columns).  Square power-of-two sides 32 ... 1024 get the tuned kernels (hh_create(n) is
 * hh_create2(n, n)); every other size in [8, 1024]^2 is served by runtime-sized kernels with the same semantics:
 * hh_simulate, hh_power_spectrum, hh_set_reference and the sweeps (candidate lists are swept run by run).  A sweep with
 * tilt / psi, or on a row length nx with a prime factor above 31 (no row-transform plan), takes the direct path: raster and
 * float64 direct transforms per candidate — exact, tens of thousands of candidates per second rather than millions. */
int hh_create(hh_ctx** out, int device, int n, int max_batch);
int hh_create2(hh_ctx** out, int device, int ny, int nx, int max_batch);
void hh_destroy(hh_ctx* ctx);
/* candidates per kernel launch this context was created with (the resolved default). */
int hh_max_batch(const hh_ctx* ctx);
/* Device memory the context holds right now (bytes; its buffers grow with the sweeps it has run and stay until
 * hh_destroy).  parts (may be NULL): {run tables, column factors, two-pass intermediate, several-segment buffers,
 * everything else}.  Typical: C2 (512^2, one 100k launch) 0.21 + 2.9 + 0.27 GB; C4 (1024^2) ~7 GB; C5 (64 segments x
 * 20k candidates) ~12 GB of masked spectra. */
int64_t hh_memory_bytes(const hh_ctx* ctx, int64_t parts[5]);
/* How hh_sweep cuts a launch of the fused pipeline into workgroups: `runs` runs of `run_len`
 * candidates, `n_kb` ky blocks, `slots` workgroups resident on the device at a time.  The first runs_a runs are cut into
 * groups_a layers of cpw_a candidates each, the others into groups_b layers of cpw_b; out = {runs_a, groups_a, cpw_a,
 * groups_b, cpw_b, layers}.  Pure host arithmetic (the reference has no counterpart: its pool takes one task per candidate,
 * app.py:2473-2476); exported so the plan can be inspected and tested without a device. */
int hh_fused_schedule(int64_t runs, int run_len, int n_kb, int slots, int32_t out[6]);
/* Which row kernel hh_sweep takes for an image that is not a power-of-two square, and its launch shape: nx = row length
 * (helical axis), rows_lds = table rows a run keeps in LDS ((2 ceil(nx apix / rise_min) + 1) n_units), kg = table rows
 * one column group may reach (<= 32).  out = {r1, r2, spectrum rows per workgroup, threads per workgroup, LDS buffers
 * for the column factors (2 = double-buffered), dynamic LDS bytes}; r1 = r2 = 0 when nx has no pair of 7-smooth factors
 * <= 32 or the shape does not fit (the Stockham kernel of any radix <= 31 runs then).  Pure host arithmetic, no device. */
int hh_general_plan(int nx, int rows_lds, int kg, int64_t out[6]);
/* ctx may be NULL: returns the message of the last failed hh_create on this thread. */
const char* hh_last_error(const hh_ctx* ctx);

/* Run all later work of this context on `hip_stream` (a hipStream_t, e.g. torch's current
 * stream handle).  NULL means the device's null stream — what torch's default stream is — so
 * work enqueued here is ordered with other work on that stream (e.g. an RCCL all-gather of the
 * scores).  hh_use_own_stream returns to the context's private non-blocking stream (the default). */
int hh_set_stream(hh_ctx* ctx, void* hip_stream);
int hh_use_own_stream(hh_ctx* ctx);

int hh_set_geometry(hh_ctx* ctx, const hh_geom* geom);

/* Experimental image(s) and mask.  images: host, S x N x N float32; mask: host, N x N bytes on
 * the fftshifted plane (DC at [N/2][N/2]), non-zero = bin takes part in the correlation;
 * log_flag selects log1p(|F|) (transforms.py:807-810).  The library transforms the images on
 * the device and keeps, per segment, the Hermitian half-plane weights and the centred
 * reference spectrum. */
int hh_set_reference(hh_ctx* ctx, const float* images, int n_segments, const uint8_t* mask, int log_flag);

/* Score G candidates.  params: G x 4 float64 (twist deg, rise A, csym, rot deg), csym-major /
 * twist / rise order is the caller's business; scores: S x G float32.  A candidate whose
 * spectrum has zero variance under the mask scores 0 (analysis.py:796-797).
 * hh_sweep takes host pointers and returns after the scores are on the host;
 * hh_sweep_device takes device pointers, enqueues on the context's stream and returns. */
int hh_sweep(hh_ctx* ctx, const double* params, int64_t n_candidates, float* scores);
int hh_sweep_device(hh_ctx* ctx, const double* d_params, int64_t n_candidates, float* d_scores);

/* hh_sweep_device for a caller that also holds the candidate list on the host (h_params, same
 * G x 4 values; may be NULL).  The library reads h_params during the call only, to see how the
 * list is ordered: when it consists of equal-length runs (>= 8) of candidates that share
 * (twist, csym, rot) — the twist-major grid of app.py:2319-2403 — and tilt = psi = 0, each run's
 * column transforms are tabulated once and no candidate is rastered or column-transformed
 * individually; where the table slice and the candidate's column factors fit in LDS the whole
 * candidate is built, row-transformed and reduced inside the compute unit (the "fused" pipeline,
 * DESIGN.md section 4), otherwise the table feeds the two-pass pipeline.  Scores agree with
 * hh_sweep_device to float32 rounding.  hh_sweep does the same analysis on its host list.
 * hh_set_table_path(ctx, mode): 0 = never, 1 = run tables + second pass only, 2 = fused where it
 * fits (default). */
int hh_sweep_device_mirrored(hh_ctx* ctx, const double* d_params, const double* h_params, int64_t n_candidates,
                             float* d_scores);
/* hh_sweep_device_mirrored with a row stride for the scores: segment s of candidate i is written to
 * d_scores[s * ld_scores + i], ld_scores >= n_candidates.  Lets a multi-GPU caller sweep straight into the
 * padded send buffer of its score all-gather (helicon_amd/distributed.py).  h_params may be NULL. */
int hh_sweep_device_strided(hh_ctx* ctx, const double* d_params, const double* h_params, int64_t n_candidates,
                            float* d_scores, int64_t ld_scores);
int hh_set_table_path(hh_ctx* ctx, int mode);
/* Which pipeline the last sweep of this context ran: 0 = per-candidate raster + column transform +
 * second pass, 1 = run tables + second pass, 2 = fused. */
int hh_last_first_pass(const hh_ctx* ctx);

/* Pre-sweep image preparation on the device.
 * hh_low_high_pass_filter: helicon.low_high_pass_filter (lib/filters.py:314-372) for one N x N float32 image
 * (host in, host out): Gaussian low pass exp(-ln2 R^2 / lp^2) and / or high pass 1 - exp(-ln2 R^2 / hp^2),
 * R = radius as a fraction of Nyquist; a fraction outside (0, 1) switches that filter off.
 * hh_threshold_data: helicon.threshold_data (lib/filters.py:283-311): out = clip(data, t, None) - t with
 * t = max(data) * thresh (use_fraction != 0) or t = thresh. */
int hh_low_high_pass_filter(hh_ctx* ctx, const float* image, double low_pass_fraction, double high_pass_fraction,
                            float* out);
int hh_threshold_data(hh_ctx* ctx, const float* data, int64_t n, int use_fraction, double thresh, float* out);

/* arg-max with ties resolved to the lowest index (np.argmax); NaN never wins. */
int hh_argmax(const float* scores, int64_t n, int64_t* index);

/* The same rule for device-resident scores: row r (a segment of a sweep's S x G output, or one rank's
 * block of an all-gathered buffer) is d_scores[r * ld .. r * ld + n), ld = 0 meaning n.  One reduction kernel on
 * the context's stream writes n_rows indices to d_index (device; NULL = a buffer of the context); when h_index
 * is not NULL they are also copied to the host and the call returns after they have arrived, otherwise the
 * call is asynchronous.  Replaces pulling S x G floats to the host for S indices. */
int hh_argmax_device(hh_ctx* ctx, const float* d_scores, int64_t n_rows, int64_t n, int64_t ld, int64_t* d_index,
                     int64_t* h_index);

/* The multi-GPU sweep's only collective for a caller without a host framework (the
 * reference has no distributed code, its candidates are pool tasks: app.py:2473-2476).  One process per GPU, one
 * context each.  RCCL is loaded with dlopen("librccl.so.1") on first use, so a single-GPU user never needs it.
 *   hh_comm_unique_id   rank 0 fills 128 bytes (ncclUniqueId) and hands them to the other ranks by any means;
 *   hh_comm_init        every rank, collectively: a communicator of `world` ranks bound to this context;
 *   hh_allgather        every rank contributes `count` floats from d_send; d_recv receives world x count floats in
 *                       rank order; enqueued on the context's stream, i.e. ordered behind the sweep that produced
 *                       d_send (hh_sweep_device_strided into a NaN-padded buffer: helicon_amd/distributed.py is the
 *                       same step with torch.distributed) and ahead of hh_argmax_device on d_recv;
 *   hh_comm_destroy     also done by hh_destroy. */
int hh_comm_unique_id(void* id128);
int hh_comm_init(hh_ctx* ctx, int rank, int world, const void* id128);
int hh_allgather(hh_ctx* ctx, const float* d_send, int64_t count, float* d_recv);
int hh_comm_destroy(hh_ctx* ctx);

/* One simulated projection (host, N x N float32) for params[4] = (twist, rise, csym, rot). */
int hh_simulate(hh_ctx* ctx, const double* params, float* image_out);

/* Amplitude spectrum of one host image: pwr_out (N x N, fftshifted, min-max normalised as
 * transforms.py:817) and phase_out (N x N, radians; may be NULL). */
int hh_power_spectrum(hh_ctx* ctx, const float* image, int log_flag, float* pwr_out, float* phase_out);

/* Pearson / cosine of two host vectors (float32 or float64), reduced on the device in float64
 * with the reference's two-pass (means, then centred sums) order of operations. */
int hh_cross_correlation(hh_ctx* ctx, const float* a, const float* b, int64_t n, double* out);
int hh_cosine_similarity(hh_ctx* ctx, const float* a, const float* b, int64_t n, double* out);
int hh_cross_correlation_f64(hh_ctx* ctx, const double* a, const double* b, int64_t n, double* out);
int hh_cosine_similarity_f64(hh_ctx* ctx, const double* a, const double* b, int64_t n, double* out);

/* The resampling step of helicon.rotate_shift_image (lib/transforms.py:315-369; used by the app's
 * auto_horizontalize, webApps/denovo3D/utils.py:410, 423): scipy.ndimage.affine_transform(data, matrix, offset,
 * order = 1, mode = "constant") of one ny x nx float32 image — out[y][x] = bilinear sample of data at
 * matrix (y, x) + offset (row-major 2 x 2 matrix, (y, x) offset), 0 where the sample point leaves [0, n - 1].
 * The caller builds matrix / offset from the angle, shifts and rotation centre exactly as the reference does
 * (helicon_amd.rotate_shift_image shows it).  Context-free: errors are read with hh_last_error(NULL). */
int hh_affine_transform_2d(int device, const float* data, int ny, int nx, const double matrix[4], const double offset[2],
                           float* out);

/* The same with order = 3 (auto_horizontalize's last step, webApps/denovo3D/utils.py:420-423:
 * rotate_shift_image(order = 3)): float64 B-spline prefilter with mirror boundaries, 4 x 4 cubic taps, 0 where the sample
 * point leaves [0, n - 1]. */
int hh_affine_transform_2d_cubic(int device, const float* data, int ny, int nx, const double matrix[4], const double offset[2],
                                 float* out);

/* ---- pre-sweep image preparation that is scikit-image in the reference -------------------------
 * scikit-image is not installed beside the reference in the build environment: these four are pinned BY DERIVATION
 * (oracle/prep.py restates scikit-image 0.25's call sequences on the installed SciPy / in NumPy; csrc/image_prep.inc).
 * `data` / `out` are host rows x cols images, float32 (is_f64 = 0) or float64 (is_f64 = 1) — scikit-image computes in the
 * image's own floating type, and so do these.  Context-free: errors are read with hh_last_error(NULL).
 *
 * hh_warp_affine_2d — skimage.transform.warp(image, inverse_map, order, mode = "constant", cval, clip) of a 2-D image
 * through its fast path, what helicon.transform_image (lib/transforms.py:238-312) calls: inverse_matrix is the row-major
 * 3 x 3 matrix that maps OUTPUT (col, row, 1) to INPUT (col, row, w) (helicon_amd.transform_image composes it as the
 * reference does); order 0 (nearest) or 1 (bilinear); a sample outside the image is cval; clip != 0 clips the result to the
 * input's range (pixels equal to a cval outside that range keep it). */
int hh_warp_affine_2d(int device, const void* data, int is_f64, int rows, int cols, const double inverse_matrix[9], int order,
                      double cval, int clip, void* out);

/* hh_rescale_2d — skimage.transform.rescale / resize(image, (out_rows, out_cols), order, mode = "reflect",
 * anti_aliasing, clip), what the app's binning (webApps/denovo3D/app.py:1911-1922) and helicon.down_scale
 * (lib/filters.py:375-412) call: a Gaussian of sigma = max(0, (rows / out_rows - 1) / 2) per axis (scipy.ndimage.gaussian_filter,
 * truncate 4, mirror boundaries) when anti_aliasing != 0, then scipy.ndimage.zoom(order, mode = "mirror", grid_mode = True)
 * — order 3 on the float64 mirror-prefiltered image, order 1 on the image itself —, then the clip to the input's range.
 * The caller computes (out_rows, out_cols) = max(round(scale * shape), 1) as rescale does. */
int hh_rescale_2d(int device, const void* data, int is_f64, int rows, int cols, int out_rows, int out_cols, int order,
                  int anti_aliasing, int clip, void* out);

/* hh_helix_moments — the numeric part of helicon.estimate_helix_rotation_center_diameter (lib/analysis.py:645-728):
 * mask = skimage.morphology.closing(data > threshold, mode = "ignore") (3 x 3 cross: a dilation, then an erosion, both
 * ignoring what lies outside the image), then _weighted_params over the mask in float64 with w = I - min_mask(I) + 1e-8.
 * out = {pixels in the mask, c_y, c_x, i_yy, i_xx, i_xy, first mask row, last mask row}; with an empty mask out[0] = 0 and
 * the rest is not meaningful. */
int hh_helix_moments(int device, const void* data, int is_f64, int rows, int cols, double threshold, double out[8]);

/* hh_ssim_2d — skimage.metrics.structural_similarity(a, b, data_range=...) with its defaults (7 x 7 uniform window, sample
 * covariance, K1 = 0.01, K2 = 0.03) of two float32 images, what helicon.ssim_score and ms_ssim_score call (lib/analysis.py:
 * 487-582; the "ssim" / "ms_ssim" / "composite" scores of lsq_reconstruct, solver:484-524): the mean of the SSIM map over the
 * image without its 3-pixel rim, float32 arithmetic in numpy's operation order, the mean in float64.  rows, cols >= 7. */
int hh_ssim_2d(int device, const float* a, const float* b, int rows, int cols, double data_range, double* out);

/* hh_joint_histogram — the counts of np.histogramdd([a, b], bins) for skimage.metrics.normalized_mutual_information
 * (helicon.mutual_information_score, lib/analysis.py:585-613): edges_a / edges_b are the caller's np.linspace(min, max,
 * bins + 1); hist: int64 [bins][bins], first index = a's bin. */
int hh_joint_histogram(int device, const float* a, const float* b, int64_t n, const double* edges_a, const double* edges_b, int bins,
                       int64_t* hist);

/* helicon.transform_map (lib/transforms.py:168-235; the reference's task function resamples the symmetrised map with
 * it when tilt / psi / dy are not zero, pipeline.py:430-432): data is host float32 [shape[0]][shape[1]][shape[2]] (z, y, x);
 * the sampling grid, centred on voxel (n // 2), is scaled, rotated by the intrinsic ZYZ Euler angles (rot, tilt, psi;
 * degrees) and shifted by (dx, dy, dz), and the volume is resampled as scipy.ndimage.map_coordinates(order = 3) does:
 * float64 B-spline prefilter with mirror boundaries, 4 x 4 x 4 cubic taps, 0 where the sample point leaves the volume.
 * out: float32, same shape.  Context-free: errors are read with hh_last_error(NULL). */
int hh_transform_map(int device, const float* data, const int32_t shape[3], double scale, double rot_degree, double tilt_degree,
                     double psi_degree, double dx, double dy, double dz, float* out);

/* Helical symmetrisation of a 3-D map (transforms.py:58-165): data is host float32
 * [in_shape[0]][in_shape[1]][in_shape[2]] (z, y, x); new_size / new_apix as in the reference (pass the
 * input's own shape / apix for "unchanged").  out receives out_shape[0..2] float32 voxels — the
 * requested size, or its even-cropped version when the reference's final slice applies
 * (transforms.py:158-164); call with out == NULL to query out_shape only.  kernel_ms (may be NULL)
 * returns the device time of the gather kernel.  Context-free: errors are read with
 * hh_last_error(NULL). */
int hh_apply_helical_symmetry(int device, const float* data, const int32_t in_shape[3], double apix,
                              double twist_degree, double rise_angstrom, int csym, double fraction,
                              const int32_t new_size[3], double new_apix, float* out, int32_t out_shape[3],
                              double* kernel_ms);

int hh_synchronize(hh_ctx* ctx);

int hh_profile_enable(hh_ctx* ctx, int on);
int hh_profile_reset(hh_ctx* ctx);
int hh_profile_get(hh_ctx* ctx, hh_profile* out);

/* Profiling aid: one launch that moves `bytes` (rounded down to whole MiB) with the sweep's
 * read shape (mode 0: contiguous 16-byte-per-lane block reads of k_second_pass) or write shape
 * (mode 1: whole 128-byte lines, 8 lanes each, one line per ky block, as k_first_pass stores them), so rocprofv3's FETCH_SIZE / WRITE_SIZE can be calibrated on a
 * known byte count for exactly these access patterns. */
int hh_calibrate_traffic(hh_ctx* ctx, int mode, int64_t bytes);

/* Algorithmic HBM bytes per candidate, B_alg(N) = 4 N^2 + 16 N (N/2 + 1). */
int64_t hh_algorithmic_bytes(int n);

/* ---------------------------------------------------------------------------------------------------------------
 * Path A, first slice: the reference's shipped scorer (sparse least squares + cosine) with a matrix-free
 * nearest-neighbour projector.  Replaces, for interpolation="nn": build_A_data_matrix
 * (webApps/denovo3D/solver_linear_regression.py:1304-1654), build_A_helical_sym_matrix (:847-1298) and the
 * scipy.sparse.linalg.lsmr calls behind scipy.optimize.lsq_linear (:258-269).  The bounded trust-region iteration,
 * the positivity rule and the score stay on the host: helicon_amd/solver.py (lsq_reconstruct, :31-547).
 * An hh_pa is one candidate's implicit system: rows = [projection rays with at least one sample in the cylinder,
 * in the reference's (symmetry operation, k, j) order] + [symmetry constraints x_i - x_j = 0]; unknowns = voxels of
 * the cylinder in C-order rank.  Vectors cross this boundary as host float64 arrays. */
typedef struct hh_pa hh_pa;
typedef struct hh_pa_params {
  double scale2d_to_3d, twist_degree, rise_pixel;
  int32_t csym;
  double tilt_degree, psi_degree, dy_pixel;
  int32_t reconstruct_diameter_2d_pixel, reconstruct_length_2d_pixel, reconstruct_diameter_3d_pixel,
      reconstruct_diameter_3d_inner_pixel, reconstruct_length_3d_pixel;
  int64_t min_projection_lines;   /* stop adding symmetry operations once the data rows exceed this (solver:1647) */
  int64_t min_sym_pairs;          /* symmetry rows wanted (solver:1275); 0 = no symmetry block */
  int32_t interpolation;          /* 0 = "nn" (solver:1511-1553, 1142-1298), 1 = "linear" (solver:1414-1503, 910-1140:
                                     trilinear weights, rays recomputed in every product, 16-entry symmetry rows) */
  int32_t fsc_mode, fsc_half;     /* half sets of lsq_reconstruct's fsc_test (split_A_b, solver:175-203): fsc_half 0 = all
                                     data rows, 1 / 2 = the rows of the first / second half of the pixel ids under
                                     fsc_mode 2 (every second id), 3 (lower / upper half), >= 4 (outer / middle thirds),
                                     1 (the ids listed in fsc_ids / the others: the caller's random split, :186-189) */
  int32_t n_fsc_ids;              /* fsc_mode 1 only: number of pixel ids (k * D2d + j) in the first half ... */
  const int32_t* fsc_ids;         /* ... and the ids, any order; read during the create call only */
} hh_pa_params;
int hh_pa_create(hh_pa** out, int device, const float* image, int ny, int nx, const hh_pa_params* params);
void hh_pa_destroy(hh_pa* pa);
const char* hh_pa_last_error(const hh_pa* pa);
/* dims = {unknowns, data rows, symmetry rows, symmetry operations used} */
int hh_pa_dims(const hh_pa* pa, int64_t dims[4]);
/* b (pixel value of every data row) and b_pid (k * D2d + j), solver:1548-1549 */
int hh_pa_get_rhs(const hh_pa* pa, float* b, int32_t* b_pid);
/* y = Op x and g = Op^T y for Op = [A diag(d); diag(root)] (d, root may be NULL: Op = A = [A_data; A_hsym]);
 * y has data rows + symmetry rows (+ unknowns when root is given) entries. */
int hh_pa_matvec(hh_pa* pa, const double* x, const double* d, const double* root, double* y);
int hh_pa_rmatvec(hh_pa* pa, const double* y, const double* d, const double* root, double* g);
/* scipy.sparse.linalg.lsmr(Op, rhs, atol, btol, conlim, maxiter) with all vectors on the device;
 * info = {istop, itn}, norms = {normr, normar} (either may be NULL). */
int hh_pa_lsmr(hh_pa* pa, const double* rhs, const double* d, const double* root, double atol, double btol, double conlim,
               int maxiter, double* x_out, int info[2], double norms[2]);


/* compute_power_spectra(data, apix, cutoff_res, output_size, log) with a Fourier-space zoom (lib/transforms.py:771-820 over
 * fft_rescale, :663-713): the image's transform at ony x onx frequencies fftfreq(on) * 2 * apix / cutoff_res, as the
 * direct sum finufft.nufft2d2(eps=1e-6) approximates, float64; (-1)^(u+v), fftshift, |F| (log1p when log_flag), min-max
 * normalisation; phase_out (may be NULL) = angle of the shifted transform.  image: ny x nx host float32; outputs ony x onx. */
int hh_power_spectrum_zoom(int device, const float* image, int ny, int nx, int ony, int onx, double apix, double cutoff_y,
                           double cutoff_x, int log_flag, float* pwr_out, float* phase_out);

/* ---------------------------------------------------------------------------------------------
 * Path A for MANY candidates at once (round 3): K candidates of one image and one reconstruction box — the K tasks the
 * reference's thread pool runs one by one (webApps/denovo3D/app.py:2473-2476 -> pipeline.py:351-404 ->
 * solver_linear_regression.py:31-547) — set up together and solved together on the device: every LSMR iteration and
 * every step of scipy.optimize.lsq_linear's trust-region-reflective loop (solver_linear_regression.py:258-269) is one
 * launch for all K; the host only reads "is anyone still iterating?".  Nearest-neighbour projector (params[i]
 * .interpolation == 0); the candidates differ in twist / rise / csym / tilt / psi / dy / fsc half and share the
 * 2-D region, the 3-D box and the row targets' meaning. */
typedef struct hh_pab hh_pab;
int hh_pab_create(hh_pab** out, int device, const float* image, int ny, int nx, const hh_pa_params* params, int count);
void hh_pab_destroy(hh_pab* pab);
const char* hh_pab_last_error(const hh_pab* pab);
/* dims = {candidates, unknowns, total data rows, total symmetry rows, device bytes held};
 * rows (may be NULL): [count][3] = {data rows, symmetry rows, symmetry operations used} */
int hh_pab_dims(const hh_pab* pab, int64_t dims[5], int64_t* rows);
/* b and pixel id (k * D2d + j) of candidate c's data rows (solver:1548-1549) */
int hh_pab_get_rhs(const hh_pab* pab, int c, float* b, int32_t* b_pid);
/* the symmetry rows of candidate c as voxel-rank pairs (row: x_i - x_j = 0; build_A_helical_sym_matrix,
 * solver:1142-1298) in the reference's row order; out: [symmetry rows][2] */
int hh_pab_get_pairs(hh_pab* pab, int c, int32_t* out);
/* lsq_linear(A_c, b_c, bounds = positive[c] ? (0, max b_c) : none, tol, max_iter, lsmr_maxiter, lsmr_tol="auto") for
 * every candidate; x -> float32; score = cosine(A_data x [clipped at 0 when clip[c]], b) (solver:484-530).
 * x_out: [count][unknowns] float32 or NULL; scores: [count]; info: [count][5] = {lsq_linear status, trust-region
 * iterations, LSMR solves, LSMR iterations, of which in the first (unconstrained) solve} or NULL. */
int hh_pab_solve(hh_pab* pab, const int32_t* positive, const int32_t* clip, double tol, int max_iter, int lsmr_maxiter,
                 float* x_out, double* scores, int32_t* info);
/* Host-only check of the trilinear products' ray arithmetic (csrc/path_a_linear.inc): the coordinates of every sample of
 * every ray under every symmetry operation of candidate *params (tilt = psi = 0), once as pa_coords computes them
 * (back_project_2d_coords_to_3d_coords + solver:1389-1394, 1576-1581) and once with the per-ray part hoisted as the product
 * kernels do; out = {samples, samples that differ, ray and sample of the first, and — with device >= 0, else -1 — the same
 * two counts by that device's arithmetic, then the samples whose cell decision differs there, -1}.  device < 0: host only. */
int hh_pab_check_ray_arithmetic(const hh_pa_params* params, int ny, int nx, int device, int64_t out[8]);
/* y = A_c x and g = A_c^T y for candidate c through the product kernels the solve uses (A_c = [A_data; A_hsym], the matrix
 * build_A_data_matrix / build_A_helical_sym_matrix return, solver:1304-1654, 847-1298; rows in the reference's order):
 * x, g: [unknowns], y: [data rows + symmetry rows], host float64.  For the parity tests; not during a solve. */
int hh_pab_matvec(hh_pab* pab, int c, const double* x, double* y);
int hh_pab_rmatvec(hh_pab* pab, int c, const double* y, double* g);
/* The scikit-learn models of solve_equations (solver_linear_regression.py:270-342: ElasticNet — the reference app's default,
 * app.py:555-558 —, Lasso, Ridge, LinearRegression, all with fit_intercept = True) for every candidate: the minimiser of
 *   (1 / 2m) || (b - mean b) - (A - 1 mu^T) w ||^2 + alpha[c] l1_ratio |w|_1 + alpha[c] (1 - l1_ratio) / 2 |w|^2,  w >= 0 where positive[c]
 * (mu = column means of A, m = rows; ridge_form: alpha[c] / m, the scaling of sklearn's Ridge) by accelerated proximal gradient
 * on the implicit operator, float64; x -> float32 and the cosine score as hh_pab_solve.  The reference's solvers visit the
 * coordinates in a random order and stop loosely, in float32: for l1_ratio < 1 (or full column rank) the minimiser is unique and
 * this is it.  tol: max |w_new - w| <= tol max |w_new|.  info: [count][3] = {iterations, converged, non-zero coefficients};
 * objective: [count] value of the function above at w; x_out, info, objective may be NULL.  Needs tilt = psi = 0. */
int hh_pab_solve_prox(hh_pab* pab, const int32_t* positive, const int32_t* clip, const double* alpha, double l1_ratio, int ridge_form,
                      double tol, int max_iter, float* x_out, double* scores, int32_t* info, double* objective);
/* counters of the last hh_pab_solve: {kernel launches, host synchronisations, LSMR iterations queued, failures of the
 * solver's self-check (every workgroup of a launch saw the per-candidate state the previous launch wrote; must be 0)} */
int hh_pab_counters(const hh_pab* pab, int64_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* HELICON_HIP_H */
