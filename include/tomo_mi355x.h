/*
 * tomo_mi355x.h -- C-ABI of libtomo_mi355x.so: the MI355X (gfx950) drop-in for the
 * ordered-subsets FISTA / ADMM hot path of ToMoBAR (parallel-beam 3D forward / back
 * projection, data-fidelity gradient, TV proximal operators and the element-wise glue).
 *
 * Conventions
 *   - every pointer named *_dev is a caller-owned DEVICE pointer to C-contiguous float32;
 *     the library owns only opaque contexts (geometry tables, grow-only scratch arena);
 *   - volume  layout [nz][n][n]   (z, y, x)   -- astra.geom_size(vol_geom),  astra_base.py:215-222,547
 *   - sinogram layout [nz][na][nu] (detY, angles, detX)                   -- dicts.py:50, astra_base.py:247-252,592
 *   - voxel (ix,iy,iz) centre at (ix-n/2+1/2, iy-n/2+1/2, iz-nz/2+1/2), unit voxels, no axis flips;
 *     detector pixel iu at signed offset cor + iu - nu/2 + 1/2 along u=(cos t, sin t, 0); rays (sin t, -cos t, 0)
 *     (tomobar/supp/funcs.py:45-65); detector row iv == volume slice iz;
 *   - every entry point is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream)
 *     unless it returns a scalar to the host (tomo_norm2, tomo_max);
 *   - return value 0 = ok; otherwise a TOMO_E_* code and tomo_last_error() (thread-local) describes it.
 *
 * Each entry point cites the reference interface it replaces (paths relative to /root/reference).
 */
#ifndef TOMO_MI355X_H
#define TOMO_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* raised whenever an entry point is added or a signature changes (tomobar_amd/_lib.py checks it at load) */
#define TOMO_ABI_VERSION 6

enum {
    TOMO_OK = 0,
    TOMO_E_INVALID = 1,  /* bad argument: maps to ValueError in the Python layer */
    TOMO_E_RUNTIME = 2,  /* HIP runtime failure */
    TOMO_E_NOMEM = 3,
    TOMO_E_NODEVICE = 4  /* no usable gfx950 device: the product path fails loudly, there is no CPU fallback */
};

/* ctx flags */
#define TOMO_FLAG_LERP8 1u /* quantise interpolation weights to 8 fractional bits (NVIDIA texture-unit emulation) */

/* data fidelity selector of tomobar/data_fidelities.py:28-39;
 * TOMO_FID_RATIO = b / max(Ax, 1e-8), the OSEM ratio of methodsIR_CuPy.py:648-650 */
enum { TOMO_FID_LS = 0, TOMO_FID_PWLS = 1, TOMO_FID_KL = 2, TOMO_FID_RATIO = 3 };

typedef struct tomo_ctx tomo_ctx;

/* per-angle geometry record (host-visible copy of the device table) */
typedef struct {
    float cs, sn;   /* (float)cos(theta), (float)sin(theta) */
    float cor;      /* horizontal CoR offset */
    float slope;    /* FP: d(interp coordinate)/d(step) */
    float inv;      /* FP: 1/sin (x-stepping) or 1/cos (y-stepping) */
    float scale;    /* FP: ray length per step */
    int32_t dirx;   /* FP: 1 = step along x / interpolate along y */
    int32_t src;    /* index into the full sinogram's angle axis */
} tomo_angle_t;

int tomo_abi_version(void);
/* "shipped" (libtomo_mi355x.so) or "dev" (libtomo_mi355x_dev.so, built with -DTOMO_DEV_VARIANTS for tests / tools) */
const char *tomo_build_flavour(void);
const char *tomo_last_error(void);
int tomo_device_count(int *count);

/* ---------------------------------------------------------------- context / geometry
 * Replaces AstraTools3D.__init__ -> AstraBase._set_vol3d_geometry (astra_base.py:215-222),
 * _set_gpu_projection3d_parallel_geometry (:244-255), _setOS_indices (:195-209) and
 * _set_projection3d_OS_parallel_geometry (:287-308); geometry vectors of supp/funcs.py:45-65.
 * angles_host [na] radians (float64).  cor_host: cor_stride==0 -> one scalar; ==1 -> [na];
 * ==2 -> [na][2] (horizontal, vertical) where the vertical component must be 0.
 * Nothing is created per call afterwards (the reference re-creates an ASTRA projector per call,
 * astra_base.py:538-545,586-604). */
int tomo_ctx_create(int device, int nz, int n, int nu, int na, const double *angles_host,
                    const double *cor_host, int cor_stride, int os_number, unsigned flags,
                    tomo_ctx **out);
int tomo_ctx_destroy(tomo_ctx *ctx);
int tomo_ctx_os_number(const tomo_ctx *ctx);
int tomo_ctx_num_bins(const tomo_ctx *ctx);                       /* AstraBase.NumbProjBins */
/* AstraBase.newInd_Vec, [os_number][num_bins] int64, zero-filled tail (astra_base.py:199-209) */
int tomo_ctx_newind_table(const tomo_ctx *ctx, int64_t *out_host);
/* subset = -1 -> all angles.  Size after the one-element trim of methodsIR_CuPy.py:454-456. */
int tomo_ctx_subset_size(const tomo_ctx *ctx, int subset);
int tomo_ctx_angle_table(const tomo_ctx *ctx, int subset, tomo_angle_t *out_host, int capacity);
/* release the context's scratch arena (cp._default_memory_pool.free_all_blocks(), methodsIR_CuPy.py:425) */
int tomo_ctx_release_scratch(tomo_ctx *ctx);
/* diagnostics: which kernel form the LAST tomo_fp3d* ("fp") / tomo_bp3d* ("bp") call on this context took, e.g.
 * "y:whole-row(1024 threads, 3 passes, 3 rows/chunk) ...".  The library also prints one stderr warning per process when
 * a slow fallback form is taken.  (No reference counterpart: ASTRA picks its kernels internally, astra_base.py:554,601.) */
const char *tomo_ctx_kernel_path(const tomo_ctx *ctx, const char *op);

/* ---------------------------------------------------------------- projector pair
 * tomo_fp3d  replaces AstraBase.runAstraProj3DCuPy  (astra_base.py:560-606, direct_FP3D :601)
 *            reached through AstraTools3D._forwprojCuPy/_forwprojOSCuPy (astra_tools3d.py:78-86).
 * tomo_bp3d  replaces AstraBase.runAstraBackproj3DCuPy (astra_base.py:518-558, direct_BP3D :554)
 *            reached through _backprojCuPy/_backprojOSCuPy (astra_tools3d.py:102-110).
 * sino_dev is [nz][subset_size][nu]; both outputs are fully overwritten.
 * Scratch: tomo_fp3d* keeps an in-plane transposed copy of the volume in the context (nz*n*n floats, tomo_ctx_release_scratch);
 * tomo_bp3d* (every form, fused epilogues included) re-lays a PLANAR sinogram quad-interleaved -- one streaming pass -- into a
 * scratch arena of this (device, stream) that is as large as the sinogram (4*ceil(nz/4)*subset_size*nu floats, grow-only, freed
 * by tomo_release_scratch): its kernel stages that layout by LDS-DMA, which is worth 7-8 x the pass.  Sinograms beyond 16 GiB,
 * or a device that cannot provide the arena (remembered, never asked again), take the planar staging instead: slower, same bits. */
int tomo_fp3d(tomo_ctx *ctx, int subset, const float *vol_dev, float *sino_dev, void *stream);
int tomo_bp3d(tomo_ctx *ctx, int subset, const float *sino_dev, float *vol_dev, void *stream);

/* ---------------------------------------------------------------- fused data-fidelity gradient
 * tomo_fp3d_residual: res = w_s (.) (A_s x - b_s)  [LS/PWLS]   or   1 - b_s / max(A_s x, 1e-8)  [KL]
 *   i.e. data_fidelities.py:28-39 with the subset gather b[:, indVec, :] of methodsIR_CuPy.py:457
 *   done by index inside the kernel.  b_dev / w_dev are FULL sinograms [nz][na][nu] by default;
 *   `gathered` bit 0 / bit 1 says that b_dev / w_dev is already the subset's [nz][subset_size][nu]
 *   (the form grad_data_term receives, data_fidelities.py:7-14).  w_dev may be NULL (LS).
 *   res_dev is [nz][subset_size][nu]. */
#define TOMO_GATHERED_B 1
#define TOMO_GATHERED_W 2
int tomo_fp3d_residual(tomo_ctx *ctx, int subset, const float *vol_dev, const float *b_dev,
                       const float *w_dev, int gathered, int fidelity, float *res_dev, void *stream);

/* ---------------------------------------------------------------- robust data terms: Huber and Student's t
 * Listed as supported data fidelities in docs/source/introduction/about.rst:38 and driven by _data_["huber_threshold"] /
 * _data_["studentst_threshold"] in Demos/methods_IR_legacy/DemoFISTA_artifacts2D.py:197,263,307,348 -- keys of the reference's
 * removed RecToolsIR class; this reference version has no implementation (supp/dicts.py:85-88), so parity is formula-level
 * (oracle/tomo_oracle.py fista(..., huber=, studentst=)), UNPINNED.  Both re-weight the (weighted) residual r before A^T:
 *   TOMO_ROBUST_HUBER     : r <- (delta / |r|) r  where |r| > delta   (gradient of the Huber function of threshold delta)
 *   TOMO_ROBUST_STUDENTST : r <- (2 / (delta^2 + r^2)) r              (gradient of log(delta^2 + r^2), [KAZ1_2017])
 * tomo_fp3d_residual_robust = tomo_fp3d_residual (LS / PWLS) with the re-weighting in the same epilogue (either residual
 * layout); tomo_sino_robust re-weights an existing residual of `count` floats in place (ring-term / vertical-CoR paths). */
enum { TOMO_ROBUST_NONE = 0, TOMO_ROBUST_HUBER = 1, TOMO_ROBUST_STUDENTST = 2 };
int tomo_fp3d_residual_robust(tomo_ctx *ctx, int subset, const float *vol_dev, const float *b_dev, const float *w_dev,
                              int gathered, int fidelity, int robust, float delta, float *res_dev, void *stream);
int tomo_sino_robust(float *res_dev, size_t count, int robust, float delta, void *stream);

/* ---------------------------------------------------------------- ring-artefact data terms (BASELINE configs[4])
 * Not in this reference version (supp/dicts.py:85-88 accepts LS / PWLS / KL only); they are documented for its removed
 * RecToolsIR class: keys ringGH_lambda / ringGH_accelerate (Group-Huber) and data fidelity "SWLS" + beta_SWLS
 * (docs/source/tutorials/real_data_recon.rst:100-151), models in docs/Kazantsev_CT_20.pdf Table III:
 *   Group-Huber  min_s 1/2 ||s - L^T r||^2 + lambda ||s||_1,  L = (I (x) 1)/sqrt(m2): one offset per detector pixel,
 *                constant over the angles; SWLS: W_s = W - W 1 (1^T W 1 + beta)^-1 1^T W per detector pixel.
 * Parity for these entry points is formula-level only (oracle/tomo_oracle.py fista(..., ring=...)).
 * tomo_fp3d_residual_ring: res = (A_s x - b_s) + ring_scale * r_x[z,u]   (r_x: [nz][nu], broadcast over the angles)
 * tomo_ring_gh_reduce    : vec[z,u] = sum_a res[z,a,u] (ascending a); r_out = r_x - l_inv*vec; then, if w_full_dev is
 *                          given, res[z,a,u] *= w_full[z, src[a], u] in place (PWLS weights after the offset step)
 * tomo_swls_apply        : res_a <- w_a res_a - w_a (sum_a w_a res_a)/(sum_a w_a + beta) per detector pixel
 * tomo_ring_gh_update    : r <- soft(r, lambda); r_x = r + beta (r - r_old); r_old <- r     (count = nz*nu)
 * src_dev: int32 [na_s], the subset's indices into the full sinogram's angle axis. */
int tomo_fp3d_residual_ring(tomo_ctx *ctx, int subset, const float *vol_dev, const float *b_full_dev,
                            const float *ring_dev, float ring_scale, float *res_dev, void *stream);
int tomo_ring_gh_reduce(float *res_dev, const float *w_full_dev, const int *src_dev, int nz, int na_s, int na_full,
                        int nu, const float *rx_dev, float l_inv, float *r_out_dev, void *stream);
int tomo_swls_apply(float *res_dev, const float *w_full_dev, const int *src_dev, int nz, int na_s, int na_full, int nu,
                    float beta, void *stream);
int tomo_ring_gh_update(float *r_dev, float *r_old_dev, float *rx_dev, float lambda, float beta, size_t count,
                        void *stream);

/* ---------------------------------------------------------------- vertical centre-of-rotation component
 * CenterRotOffset may be [angles, 2] = (horizontal, vertical) offsets (supp/funcs.py:52-55).  The context takes the
 * horizontal components; a vertical component moves the detector of angle a by shift[a] rows, which for parallel rays is a
 * per-angle 2-tap resampling of the detector rows (zero outside): A_v = R(+shift) A, A_v^T = A^T R(-shift).
 * tomo_shift_rows: out[r,a,:] = (1-w) in[r+k,a,:] + w in[r+k+1,a,:], k + w = sign*shift[a], k = floor (shift_dev: float32 [na];
 * the weight does not depend on the row, so a z-slab with ghost rows resamples exactly as the whole detector does).
 * tomo_sino_residual: the residual of data_fidelities.py:28-39 on an already projected subset (the fused
 * tomo_fp3d_residual cannot resample between projector and residual; ``gathered`` as there).  Not used when no vertical component is given. */
int tomo_shift_rows(const float *in_dev, float *out_dev, int nz, int na, int nu, const float *shift_dev, float sign,
                    void *stream);
int tomo_sino_residual(const float *ax_dev, const float *b_full_dev, const float *w_full_dev, const int *src_dev, int nz,
                       int na_s, int na_full, int nu, int gathered, int fidelity, float *res_dev, void *stream);
/* tomo_sino_add_ring: res[z,a,u] += ring_scale * ring[z,u] -- the Group-Huber offsets on a residual formed by
 * tomo_sino_residual (the pair replaces tomo_fp3d_residual_ring when a vertical component sits between projector and residual). */
int tomo_sino_add_ring(float *res_dev, const float *ring_dev, float ring_scale, int nz, int na_s, int nu, void *stream);

/* tomo_bp3d_fista: x_out = P+( x_t - l_inv * A_s^T res )            methodsIR_CuPy.py:463-468
 *   nonneg != 0 applies the max(.,0) projection.  x_out may alias x_t. */
int tomo_bp3d_fista(tomo_ctx *ctx, int subset, const float *res_dev, const float *xt_dev,
                    float *xout_dev, float l_inv, int nonneg, void *stream);
/* tomo_bp3d_fista_momentum (no proximal operator between gradient step and momentum):
 *   X = P+(X_t - l_inv*A_s^T res);  X_t <- X + beta*(X - X_old);   methodsIR_CuPy.py:463-475
 *   on entry xold_x_dev holds X_old, on exit X;  xt_dev holds X_t on entry and the new X_t on exit. */
int tomo_bp3d_fista_momentum(tomo_ctx *ctx, int subset, const float *res_dev, float *xt_dev,
                             float *xold_x_dev, float l_inv, float beta, int nonneg, void *stream);
/* tomo_bp3d_admm:  g = A_s^T res;  z <- z - tau*(g + rho*(z - x + u));  optional max(.,0);
 *   if relax_on:  z <- (1-alpha)*z_start + alpha*z;   zu_out = z + u     methodsIR_CuPy.py:545-557
 *   (z_old of the reference equals z at the start of the sub-iteration, :555).
 *   one_minus_alpha / alpha are passed pre-rounded to float32 by the caller. */
int tomo_bp3d_admm(tomo_ctx *ctx, int subset, const float *res_dev, float *z_dev, const float *x_dev,
                   const float *u_dev, float *zu_out_dev, float tau, float rho, int relax_on,
                   float one_minus_alpha, float alpha, int nonneg, void *stream);

/* Private layout of the residual BETWEEN tomo_fp3d_residual and its consumer tomo_bp3d_fista / _fista_momentum / _admm on
 * the same context (round 5).  The reference materialises the residual as a [detY, angles, detX] CuPy array between
 * grad_data_term's forward and back projection (data_fidelities.py:28-40); nobody else reads it on the plain LS / PWLS /
 * KL path, so producer and consumer may agree on the layout the back projector stages fastest:
 *   TOMO_RESIDUAL_PLANAR (default): res_dev is [nz][subset_size][nu], what every entry point documents.
 *   TOMO_RESIDUAL_ZQUAD           : res_dev is [ceil(nz/4)][subset_size][nu][4] -- the four slices of a quad interleaved
 *       (slices >= nz read as 0).  The forward projector's workgroup holds exactly these four values per (angle, pixel)
 *       and stores them as ONE 16-byte word; the back projector stages one 16-byte load per (angle, quad, sample) where
 *       the planar layout costs four dword gathers from rows nz * subset_size * nu floats apart.  Same arithmetic, bit
 *       for bit.  res_dev must be 16-byte aligned and hold tomo_ctx_residual_elems(ctx, subset) floats.
 * Only the three fused epilogue entry points and tomo_fp3d_residual follow the context's setting; tomo_fp3d / tomo_bp3d
 * always take the planar layout, and tomo_fp3d_residual_ring (whose residual is read by tomo_ring_gh_reduce) refuses to
 * run while ZQUAD is set. */
enum { TOMO_RESIDUAL_PLANAR = 0, TOMO_RESIDUAL_ZQUAD = 1 };
int tomo_ctx_set_residual_layout(tomo_ctx *ctx, int layout);
int tomo_ctx_residual_layout(const tomo_ctx *ctx);
size_t tomo_ctx_residual_elems(const tomo_ctx *ctx, int subset);

/* ---------------------------------------------------------------- element-wise glue
 * tomo_momentum : x_t = x + beta*(x - x_old)                         methodsIR_CuPy.py:475
 * tomo_admm_dual: u  += z - x                                        methodsIR_CuPy.py:566
 * tomo_axpby    : y   = a*x + b*y   (Landweber / SIRT / CGLS updates, methodsIR_CuPy.py:165,222,285-295)
 * tomo_scale    : y   = a*x                                          methodsIR_CuPy.py:337
 * tomo_clamp_min: x   = max(x, lo)                                   methodsIR_CuPy.py:468,549
 * tomo_mul      : y   = x*y ;  tomo_recip_safe: y = 1/x with nan/inf -> 1 (SIRT, :206-214)
 * tomo_norm2    : sqrt(sum x^2) -> host (cp.linalg.norm,            methodsIR_CuPy.py:336,349)
 * tomo_dot      : sum x*y -> host (cp.inner, :272,284,292);  tomo_max: max -> host (:395)
 * tomo_pwls_weights: w = max(b,1e-6) / max(max(b,1e-6))              methodsIR_CuPy.py:392-395 */
int tomo_momentum(const float *x_dev, const float *xold_dev, float *xt_dev, float beta, size_t count, void *stream);
/* the same update for a context's volume [nz][n][n] that also leaves the in-plane transposed X_t in the context: the NEXT
 * tomo_fp3d* call on xt_dev (and only that one) skips its own transpose pass.  Same value as tomo_momentum, bit for bit. */
int tomo_momentum_transposed(tomo_ctx *ctx, const float *x_dev, const float *xold_dev, float *xt_dev, float beta,
                             void *stream);
/* The transposed copy is a one-shot token keyed on (xt_dev, stream): ANY following tomo_fp3d* call on the context spends
 * it, whether it can use it or not.  tomo_ctx_invalidate drops it explicitly -- call it when the volume behind xt_dev may
 * be rewritten or freed before the next forward projection (RecToolsIRCuPy.FISTA does at entry and exit, so a buffer the
 * allocator hands out again at the same address can never meet a stale copy; methodsIR_CuPy.py:447-475 has no
 * counterpart: ASTRA re-uploads the volume on every call, astra_base.py:560-606). */
int tomo_ctx_invalidate(tomo_ctx *ctx);
int tomo_admm_dual(float *u_dev, const float *z_dev, const float *x_dev, size_t count, void *stream);
int tomo_axpby(float a, const float *x_dev, float b, float *y_dev, size_t count, void *stream);
int tomo_scale(float a, const float *x_dev, float *y_dev, size_t count, void *stream);
int tomo_clamp_min(float *x_dev, float lo, size_t count, void *stream);
int tomo_mul(const float *x_dev, float *y_dev, size_t count, void *stream);
int tomo_recip_safe(const float *x_dev, float *y_dev, size_t count, void *stream);
int tomo_fill(float *x_dev, float value, size_t count, void *stream);
int tomo_norm2(const float *x_dev, size_t count, double *out_host, void *stream);
int tomo_dot(const float *x_dev, const float *y_dev, size_t count, double *out_host, void *stream);
int tomo_max(const float *x_dev, size_t count, float *out_host, void *stream);
int tomo_pwls_weights(const float *b_dev, float *w_dev, size_t count, void *stream);
/* z-slab form of the same: the caller max-all-reduces tomo_pwls_max over the slabs and passes the result on */
int tomo_pwls_max(const float *b_dev, size_t count, float *out_host, void *stream);
int tomo_pwls_weights_scaled(const float *b_dev, float *w_dev, size_t count, float wmax, void *stream);
/* diagnostics: nin-read / nout-write streaming kernel (vec = 4: float4 accesses, 1: dword) used to calibrate the
 * achievable HBM rate for a kernel's read/write mix (tools/archive/probes/stream_probe.py) */
int tomo_diag_stream(const float *const *in_dev, int nin, float *const *out_dev, int nout, size_t count,
                     int vec, int grid, void *stream);

/* ---------------------------------------------------------------- pre/post glue
 * tomo_pad_edge   : edge-pad detX by `pad` both sides, [nz][na][nu0] -> [nz][na][nu0+2pad]
 *                   (_apply_horiz_detector_padding, supp/suppTools.py:425-459)
 * tomo_crop_center: centre-crop y,x  [nz][n][n] -> [nz][m][m]  (perform_recon_crop, suppTools.py:399-422)
 * tomo_circ_mask  : in-place disc mask of apply_circular_mask (suppTools.py:364-396)
 * tomo_permute3   : out[i][j][k] = in laid out with strides (s0,s1,s2) -- materialises the axis swap of
 *                   _data_dims_swapper (supp/funcs.py:190-206) as a contiguous array */
int tomo_pad_edge(const float *in_dev, float *out_dev, int rows, int nu0, int pad, void *stream);
int tomo_crop_center(const float *in_dev, float *out_dev, int nz, int n, int m, void *stream);
int tomo_circ_mask(float *vol_dev, int nz, int n, double radius, void *stream);
int tomo_permute3(const float *in_dev, float *out_dev, int d0, int d1, int d2,
                  int64_t s0, int64_t s1, int64_t s2, void *stream);

/* ---------------------------------------------------------------- TV proximal operators
 * tomo_pdtv replaces PD_TV_cupy's device work (regularisersCuPy.py:220-296) and the 16 kernels
 *   primal_dual_for_total_variation_{2D,3D}_{float,half}[_nonneg][_methodTV]
 *   (cuda_kernels/primal_dual_for_total_variation.cu:263-301,454-492).
 *   dims: dx fastest.  nd = 2 -> [dy][dx] (dz ignored), nd = 3 -> [dz][dy][dx].
 *   sigma,tau,lt,theta are the float32 scalars of regularisersCuPy.py:215-218 (computed by the caller).
 *   half != 0 stores the dual fields as IEEE binary16.  out_dev receives U_arrays[iters % 2].
 * tomo_roftv replaces ROF_TV_cupy's device work (regularisersCuPy.py:78-167) and
 *   divergence_kernel_* / TV_kernel_* (cuda_kernels/rudin_osher_fatemi_total_variation.cu:106-148,203-248);
 *   the two kernels are fused (the D fields never reach HBM); half != 0 rounds D through binary16.
 * `device` selects the GPU (gpu_id argument of the reference functions). */
int tomo_pdtv(int device, const float *in_dev, float *out_dev, int dx, int dy, int dz, int nd,
              float sigma, float tau, float lt, float theta, int iters, int methodTV, int nonneg,
              int half, void *stream);
int tomo_roftv(int device, const float *in_dev, float *out_dev, int dx, int dy, int dz, int nd,
               float lambda, float tau, int iters, int half, void *stream);
/* scratch bytes the TV drivers hold for a given problem (informational) and arena release */
size_t tomo_pdtv_scratch_bytes(int dx, int dy, int dz, int nd, int half);
size_t tomo_roftv_scratch_bytes(int dx, int dy, int dz, int nd);
int tomo_release_scratch(int device);
/* Placement of the TV scratch arenas (no reference counterpart: CuPy's memory pool hands out whatever block comes next).
 * On MI355X the speed of the plane-marching TV kernels depends on where in HBM their arrays lie (PD_TV launch at 1024^3:
 * 10.1 ms with the arena in one block, 9.1-9.3 ms in another of the same process; docs/kernels/placement.md).  The TV
 * operators' own arena and the tomo_placed_scratch blocks -- nothing else: FBP spectra, Fourier and reduction scratch are
 * plain allocations -- are therefore chosen, when they are >= 1 GiB, among up to `tries` candidate allocations held at
 * once, each scored by a ~7 ms z-march probe; the search stops at the first candidate 8 % above an earlier one (the
 * "fast class"), keeps the best and frees the rest.  Default 8 tries (environment TOMO_MI355X_PLACE_TRIES, at most 10),
 * 1 = plain hipMalloc.  Bounds on what the search may hold TRANSIENTLY: the candidates together never exceed 85 % of the
 * memory that was free when the search began, and a further candidate is only taken while the device keeps 4 GiB free;
 * one search runs at a time per process, outside the lock that guards the other arenas.  A caller that shares the GPU
 * with another allocator and cannot afford the transient footprint (up to tries x arena bytes) sets tries = 1.
 * tomo_placement_last reports the most recent search of this process: returns the number of candidates scored
 * (0 = none yet), *bytes the block size, *chosen the index kept, scores_GBps[i] the probe rate of candidate i;
 * tomo_placement_last_fast: 1 if the kept block cleared the 8 % rule, 0 if the tries ran out first, -1 if none ran. */
/* Allocate (and place) the TV operators' scratch arena of this (device, stream) ahead of the first call that needs it --
 * e.g. tomo_pdtv_scratch_bytes(...) at set-up time, so that the placement search (0.1-4 s) is not part of the first
 * iteration (RecToolsIRCuPy.FISTA / ADMM / OSEM do this before their loops).  Grow-only like every arena: a later call
 * that needs more re-allocates. */
int tomo_reserve_scratch(int device, size_t bytes, void *stream);
int tomo_set_placement_tries(int tries);
int tomo_placement_tries(void);
/* A placed scratch block for callers that keep their own plane-marching work arrays (the z-slab drivers hold ghosted
 * copies of U, P1..3 and Input per rank; the reference's multi-GPU demo has cupy allocate them,
 * Demos/methods_IR_legacy/MultiGPU_demo.py:144-190): `bytes` of device memory owned by the library, one block per
 * (device, stream, slot 0..7), grow-only -- asking a slot for MORE than it holds frees the old block, so earlier pointers
 * into it die --, released by tomo_release_scratch.  Placement as above. */
int tomo_placed_scratch(int device, int slot, size_t bytes, void *stream, void **out_dev);
int tomo_placement_last(size_t *bytes, int *chosen, double *scores_GBps, int capacity);
int tomo_placement_last_fast(void);

/* Slab (multi-GPU) form of one PD-TV iteration on arrays that carry ghost planes:
 *   every array pointer addresses [has_lo + nz_local + has_hi][dy][dx]; the planes at either end are
 *   the neighbours' boundary planes (read-only).  Boundary rules (zIndex>0 / last_z of
 *   primal_dual_for_total_variation.cu:188,213) apply only where has_lo/has_hi == 0. */
int tomo_pdtv_iter_slab(int device, const float *in_dev, const float *u_in_dev, float *u_out_dev,
                        const void *p_in_dev[3], void *p_out_dev[3], int dx, int dy, int nz_local,
                        int has_lo, int has_hi, float sigma, float tau, float lt, float theta,
                        int methodTV, int nonneg, int half, void *stream);
/* Two PD-TV iterations in one pass on a slab: arrays address [lo_planes + nz_local + hi_planes][dy][dx] with
 * lo_planes, hi_planes in {0, 2, 3}.  Ghost planes that must be valid on entry: U two planes either side; P two planes
 * below and the first plane above; Input the nearer plane either side.  Result = two applications of
 * tomo_pdtv_iter_slab with a ghost refresh in between, bit for bit. */
int tomo_pdtv_pair_slab(int device, const float *in_dev, const float *u_in_dev, float *u_out_dev,
                        const void *p_in_dev[3], void *p_out_dev[3], int dx, int dy, int nz_local,
                        int lo_planes, int hi_planes, float sigma, float tau, float lt, float theta,
                        int methodTV, int nonneg, int half, void *stream);
/* Same, restricted to the local output planes [z_begin, z_end) (0 <= z_begin <= z_end <= nz_local).  Lets a rank
 * compute the planes its neighbours wait for first, start the halo exchange, and compute the interior while the
 * planes travel (tomobar_amd/slab.py).  tomo_pdtv_pair_slab == the range [0, nz_local). */
int tomo_pdtv_pair_slab_range(int device, const float *in_dev, const float *u_in_dev, float *u_out_dev,
                              const void *p_in_dev[3], void *p_out_dev[3], int dx, int dy, int nz_local,
                              int lo_planes, int hi_planes, int z_begin, int z_end, float sigma, float tau,
                              float lt, float theta, int methodTV, int nonneg, int half, void *stream);
/* K iterations (k = 2 or 3) in one pass on a slab whose arrays carry lo_planes / hi_planes in {0, k..3} ghost planes.
 * Ghost planes that must be valid on entry: U and P1..3 k planes below, U k planes and P1..3 k-1 planes above, Input
 * k-1 planes either side.  Result = k applications of tomo_pdtv_iter_slab with ghost refreshes in between, bit for bit.
 * (Which k a run uses is the host's choice: tomobar_amd/slab.py asks tomo_pdtv_iters_per_launch -- 3 for both dual types in the shipped build.) */
int tomo_pdtv_multi_slab_range(int device, const float *in_dev, const float *u_in_dev, float *u_out_dev,
                               const void *p_in_dev[3], void *p_out_dev[3], int dx, int dy, int nz_local,
                               int lo_planes, int hi_planes, int z_begin, int z_end, int k, float sigma, float tau,
                               float lt, float theta, int methodTV, int nonneg, int half, void *stream);
int tomo_roftv_iter_slab(int device, const float *in_dev, const float *u_in_dev, float *u_out_dev,
                         int dx, int dy, int nz_local, int lo_planes, int hi_planes,
                         float lambda, float tau, int half, void *stream);
/* Same for the local output planes [z_begin, z_end) only (boundary planes first, exchange, then the interior). */
int tomo_roftv_iter_slab_range(int device, const float *in_dev, const float *u_in_dev, float *u_out_dev,
                               int dx, int dy, int nz_local, int lo_planes, int hi_planes, int z_begin, int z_end,
                               float lambda, float tau, int half, void *stream);

/* Halo staging for the z-slab exchange ("tomo_halo_exchange"; the reference scales by independent replicas
 * only, Demos/methods_IR_legacy/MultiGPU_demo.py:144-190, so there is no call to replace).  The transport itself stays
 * with the host (torch.distributed / RCCL in tomobar_amd/slab.py, mpi4py or cupy.cuda.nccl for a CuPy caller --
 * INTEGRATION.md section C); what the library provides is the device side: the planes a neighbour needs -- the last or
 * first k planes of U, P1, P2, P3 (each contiguous in its own array) -- gathered into ONE contiguous staging buffer and
 * scattered back, so that an exchange is one send and one receive per neighbour.  Block i occupies
 * [off_i, off_i + bytes[i]) of the staging buffer with off_0 = 0, off_{i+1} = off_i + round_up(bytes[i], 16);
 * tomo_halo_staging_bytes returns the total.  At most 8 blocks per call. */
size_t tomo_halo_staging_bytes(const size_t *bytes, int nblocks);
int tomo_halo_pack(const void *const *src_dev, const size_t *bytes, int nblocks, void *staging_dev, void *stream);
int tomo_halo_unpack(const void *staging_dev, void *const *dst_dev, const size_t *bytes, int nblocks, void *stream);
/* How many PD_TV iterations tomo_pdtv fuses into one launch for float32 (half = 0) / binary16 (half != 0) dual fields under
 * the calling thread's kernel variant: a slab driver that wants to be launch-for-launch identical to the whole-volume
 * operator cuts its iterations the same way (tomobar_amd/slab.py: pd_launch_plan) and keeps that many ghost planes. */
int tomo_pdtv_iters_per_launch(int half);

/* ---------------------------------------------------------------- FBP filter
 * tomo_fbp_filter replaces _filtersinc3D_cupy (tomobar/fourier.py:26-78) and generate_filtersinc
 * (cuda_kernels/generate_filtersync.cu:5-82): every row of `rows` x `nu` float32 values is replaced, in place, by
 * irfft( rfft(row) * f ), f = fftshift-ed half-spectrum sinc-ramp of cut-off `cutoff`, unnormalised transforms with
 * `multiplier` (= 1/angles/nu in RecToolsDIRCuPy.FBP) folded into f.  Batched hipFFT; synchronises the stream. */
int tomo_fbp_filter(int device, float *data_dev, size_t rows, int nu, float cutoff, float multiplier, void *stream);

/* ---------------------------------------------------------------- Fourier reconstruction
 * tomo_fourier_inv replaces the device side of RecToolsDIRCuPy.FOURIER_INV (tomobar/methodsDIR_CuPy.py:152-447: the
 * stages _fbp_filtering :449-545, _setup_backprojection_input :645-683, _fft_and_interpolation :701-836,
 * ifft_gathered_projections :851-897, unpad_reconstructed_data :920-967) and the kernels of
 * tomobar/cuda_kernels/fft_us_kernels.cu.
 *   data_dev  [nz][nproj][raw_n] float32, nz and raw_n even (the caller pads odd sizes as the reference does, :265-279)
 *   out_dev   [out_z][out_size][out_size] float32; out_z = nz or nz-1 (odd original height)
 *   n         detector width after horizontal padding (even), ne the oversampled filter width (:465-474)
 *   unpad_m   first reconstructed column/row relative to -n/2 (unpad_recon_m, :933)
 *   w_host    HOST array of ne/2+1 complex64 (re,im pairs): filter table x phase ramp of the rotation axis (:481-483)
 *   theta_host HOST array of nproj float32 angles as the kernels see them (= -AnglesVec, :295)
 *   m, mu     footprint half-width and Gaussian parameter (:321-322, :726-737)
 *   center_size  min(center_size, 2n) of the reference (:290): >= 192 selects the circular-support gathering inside
 *             the centre box and the square-footprint form outside it; < 192 the square-footprint form everywhere.
 * Workspace: the per-device scratch arena; hipFFT plans are cached between calls (both released by
 * tomo_release_scratch).  Processes 128 slices per chunk, synchronises the stream before returning. */
int tomo_fourier_inv(int device, const float *data_dev, float *out_dev, int nz, int out_z, int nproj, int raw_n,
                     int n, int ne, int unpad_m, int out_size, const float *w_host, const float *theta_host,
                     int m, float mu, int center_size, void *stream);

/* kernel-variant selector (per calling host thread): name in {"bp","fp","pdtv","roftv"}; variant 0 = the default of every
 * class.  The shipped library accepts exactly one other value: "pdtv" 22 = float32 duals with the rounding sequence of the
 * reference's kernels reproduced bit for bit (primal_dual_for_total_variation.cu:66-123; FMA-corrected 1 / sqrtf and
 * quotient), +5 ... +16 % per launch depending on the box.  The default runs float32 duals with relaxed arithmetic (v_rsq_f32 instead of 1 / sqrtf, a
 * host-computed 1 / (1 + lt) instead of the divide: within 1e-5 of the reference, typically 3e-7); binary16 duals
 * (half_precision) and ROF_TV reproduce the reference's roundings in every build (rudin_osher_fatemi_total_variation.cu:51-61).
 * Anything else returns TOMO_E_INVALID: the independent implementations and A/B builds used by tests/ and tools/ (bp 1/2,
 * fp 1/2/3, pdtv 1/2/3/21, roftv 1..4) and the measurement switches ("probe") exist only in libtomo_mi355x_dev.so
 * (csrc/Makefile: `make dev`). */
int tomo_set_variant(const char *kernel, int variant);

/* In-library kernel timing for bench.py's roofline object: while enabled, every launch group of a kernel class
 * ("bp","fp","pdtv","roftv") is bracketed by HIP events recorded ON THE LAUNCH STREAM.  tomo_profile_read
 * synchronises those events and returns the number of kernel launches and their summed duration. */
int tomo_profile_enable(int on);
int tomo_profile_read(const char *kernel, long long *launches, double *total_ms);

/* ---------------------------------------------------------------- host (CPU) 2D plumbing, BASELINE configs[0]
 * RecToolsDIR(..., device_projector="cpu") of the reference runs ASTRA's CPU `line` projector / `BP` algorithm
 * (tomobar/methodsDIR.py:71-175, astra_base.py:224-232,310-372).  These two entry points take HOST pointers and need no
 * GPU: img [n][n], sino [na][nu], angles in radians, scalar CoR (the reference's CPU path rejects a non-zero one,
 * astra_base.py:150-153).  Same operator model and float32 arithmetic as tomo_bp3d / tomo_fp3d. */
int tomo_host_bp2d(const float *sino_host, float *img_host, int n, int nu, int na, const double *angles_host, double cor);
int tomo_host_fp2d(const float *img_host, float *sino_host, int n, int nu, int na, const double *angles_host, double cor);

#ifdef __cplusplus
}
#endif
#endif /* TOMO_MI355X_H */
