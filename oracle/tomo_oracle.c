/*
 * oracle/tomo_oracle.c -- CPU restatement of the ToMoBAR FISTA/ADMM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under tomobar_amd/ may import, link or
 * execute this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and there only as the checker / CPU baseline.
 *
 * What it restates (all paths relative to /root/reference):
 *   orc_fp3d  : the `Ax` operator called at tomobar/astra_wrappers/astra_base.py:601
 *               (direct_FP3D).  The arithmetic lives in the un-vendored third-party
 *               astra-toolbox==2.4.* (pyproject.toml:41), so this is a restatement of
 *               its published model (Joseph ray-driven, dominant-axis stepping, linear
 *               interpolation, ray-length scaling) on the geometry the reference builds
 *               at tomobar/supp/funcs.py:45-65 and astra_base.py:215-222,244-255.
 *   orc_bp3d  : the `A^T b` operator at astra_base.py:554 (direct_BP3D): voxel-driven,
 *               2-tap linear interpolation along the detector u axis, unit scale.
 *   orc_pdtv  : tomobar/cuda_kernels/primal_dual_for_total_variation.cu:125-261 (3D)
 *               and :360-452 (2D), driven as tomobar/regularisersCuPy.py:252-296.
 *   orc_roftv : tomobar/cuda_kernels/rudin_osher_fatemi_total_variation.cu:156-238 (3D)
 *               and :66-137 (2D), driven as tomobar/regularisersCuPy.py:108-167.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - projector pair: pinned by the reference's data-free literals
 *     (tests/test_RecToolsDIRCuPy.py:691-692 ones-cube FP min/max in LERP8 mode;
 *      tests/test_RecToolsIRCuPy.py:316,573,639 power-method constants), see
 *     tests/test_oracle_known_answers.py.  Element-wise ASTRA output: unpinned
 *     (ASTRA and the upstream tests' .npz data are absent).
 *   - TV operators: checked against the reference .cu sources executed on the host
 *     (oracle/ref_tv, fixtures in tests/golden/), no upstream golden vectors exist.
 *
 * Floating-point contract: compiled with -ffp-contract=off; every fused
 * multiply-add is an explicit fmaf() so the HIP kernels can reproduce the exact
 * rounding sequence.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_FLAG_LERP8 1 /* quantise interpolation weight to 8 fractional bits (NVIDIA texture unit emulation) */

/* per-angle table entry; layout shared with include/tomo_mi355x.h (tomo_angle_t) */
typedef struct {
    float cs;    /* (float)cos(theta) */
    float sn;    /* (float)sin(theta) */
    float cor;   /* horizontal centre-of-rotation offset for this angle */
    float slope; /* FP: d(interp coordinate)/d(step index) */
    float inv;   /* FP: 1/sin (x-stepping) or 1/cos (y-stepping) */
    float scale; /* FP: ray length per step = 1/|sin| or 1/|cos| */
    int32_t dirx; /* FP: 1 = step along x, interpolate along y; 0 = step along y, interpolate along x */
    int32_t src;  /* index of this angle in the full sinogram's angle axis */
} orc_angle;

/* Geometry of supp/funcs.py:45-65: ray (sin,-cos,0), u=(cos,sin,0), det centre = cor*u. */
void orc_make_angles(const double *theta, const double *cor, int cor_stride, int na,
                     const int64_t *index, int nsel, orc_angle *out)
{
    for (int k = 0; k < nsel; ++k) {
        int64_t a = index ? index[k] : k;
        double th = theta[a];
        double c = cos(th), s = sin(th);
        orc_angle t;
        t.cs = (float)c;
        t.sn = (float)s;
        t.cor = (float)(cor_stride ? cor[a * cor_stride] : cor[0]);
        t.dirx = fabs(s) >= fabs(c);
        if (t.dirx) {
            t.slope = (float)(-c / s);
            t.inv = (float)(1.0 / s);
            t.scale = (float)(1.0 / fabs(s));
        } else {
            t.slope = (float)(-s / c);
            t.inv = (float)(1.0 / c);
            t.scale = (float)(1.0 / fabs(c));
        }
        t.src = (int32_t)a;
        out[k] = t;
    }
    (void)na;
}

static inline float lerp_weight(float f, float fl, int flags)
{
    float w = f - fl;
    if (flags & ORC_FLAG_LERP8)
        w = rintf(w * 256.0f) * (1.0f / 256.0f);
    return w;
}

/* vol [nz][n][n] -> sino [nz][na][nu];  Joseph ray-driven forward projection. */
void orc_fp3d(const float *vol, float *sino, int nz, int n, int nu, int na,
              const orc_angle *tab, int flags)
{
    const float half_n = 0.5f * (float)n - 0.5f;   /* N/2 - 1/2 */
    const float half_u = 0.5f * (float)nu - 0.5f;
#pragma omp parallel for collapse(2) schedule(static)
    for (int iz = 0; iz < nz; ++iz) {
        for (int a = 0; a < na; ++a) {
            const orc_angle t = tab[a];
            const float *slice = vol + (size_t)iz * n * n;
            float *row = sino + ((size_t)iz * na + a) * nu;
            for (int iu = 0; iu < nu; ++iu) {
                float s = ((float)iu - half_u) + t.cor;
                float offset = fmaf(s, t.inv, half_n);
                float acc = 0.0f;
                for (int k = 0; k < n; ++k) {
                    float kw = (float)k - half_n;
                    float f = fmaf(kw, t.slope, offset);
                    float fl = floorf(f);
                    float w = lerp_weight(f, fl, flags);
                    int i0 = (int)fl;
                    float v0 = 0.0f, v1 = 0.0f;
                    if (t.dirx) { /* k = ix, interpolate along y */
                        if (i0 >= 0 && i0 < n) v0 = slice[(size_t)i0 * n + k];
                        if (i0 + 1 >= 0 && i0 + 1 < n) v1 = slice[(size_t)(i0 + 1) * n + k];
                    } else { /* k = iy, interpolate along x */
                        if (i0 >= 0 && i0 < n) v0 = slice[(size_t)k * n + i0];
                        if (i0 + 1 >= 0 && i0 + 1 < n) v1 = slice[(size_t)k * n + i0 + 1];
                    }
                    acc = fmaf(1.0f - w, v0, acc);
                    acc = fmaf(w, v1, acc);
                }
                row[iu] = acc * t.scale;
            }
        }
    }
}

/* sino [nz][na][nu] -> vol [nz][n][n];  voxel-driven back projection (overwrites vol). */
void orc_bp3d(const float *sino, float *vol, int nz, int n, int nu, int na,
              const orc_angle *tab, int flags)
{
    const float half_n = 0.5f * (float)n - 0.5f;
    const float half_u = 0.5f * (float)nu - 0.5f;
#pragma omp parallel for collapse(2) schedule(static)
    for (int iz = 0; iz < nz; ++iz) {
        for (int iy = 0; iy < n; ++iy) {
            const float yw = (float)iy - half_n;
            for (int ix = 0; ix < n; ++ix) {
                const float xw = (float)ix - half_n;
                float acc = 0.0f;
                for (int a = 0; a < na; ++a) {
                    const float off = half_u - tab[a].cor;
                    float f = fmaf(xw, tab[a].cs, fmaf(yw, tab[a].sn, off));
                    float fl = floorf(f);
                    float w = lerp_weight(f, fl, flags);
                    int i0 = (int)fl;
                    const float *row = sino + ((size_t)iz * na + a) * nu;
                    float s0 = (i0 >= 0 && i0 < nu) ? row[i0] : 0.0f;
                    float s1 = (i0 + 1 >= 0 && i0 + 1 < nu) ? row[i0 + 1] : 0.0f;
                    acc = fmaf(1.0f - w, s0, acc);
                    acc = fmaf(w, s1, acc);
                }
                vol[((size_t)iz * n + iy) * n + ix] = acc;
            }
        }
    }
}

/* ---- IEEE binary16 round trip (round-to-nearest-even), the storage format of the
 *      "half_precision" dual fields (primal_dual...cu:52-56, rudin_osher...cu:44-48) ---- */
static inline uint16_t f32_to_f16_bits(float x)
{
    uint32_t b;
    memcpy(&b, &x, 4);
    uint32_t sign = (b >> 16) & 0x8000u;
    uint32_t absb = b & 0x7fffffffu;
    if (absb >= 0x7f800000u) /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((absb > 0x7f800000u) ? 0x200u : 0));
    if (absb >= 0x477ff000u) /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7c00u);
    if (absb < 0x38800000u) { /* subnormal half or zero */
        if (absb < 0x33000000u) /* < 2^-25 -> 0 (2^-25 itself ties to even = 0) */
            return (uint16_t)sign;
        int e = (int)(absb >> 23);               /* biased exp, 102..112 */
        uint32_t m = (absb & 0x7fffffu) | 0x800000u; /* 24-bit significand */
        int shift = 126 - e;                     /* 14..24 : target unit 2^-24 */
        uint32_t q = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (q & 1u)))
            q++;
        return (uint16_t)(sign | q);
    }
    uint32_t e = (absb >> 23) - 112u; /* half biased exponent 1..30 */
    uint32_t m = absb & 0x7fffffu;
    uint32_t q = (e << 10) | (m >> 13);
    uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (q & 1u)))
        q++; /* may carry into the exponent, which is the correct result */
    return (uint16_t)(sign | q);
}

static inline float f16_bits_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    uint32_t b;
    if (e == 0) {
        if (m == 0) {
            b = sign;
        } else { /* subnormal: value = m * 2^-24 */
            float v = (float)m * 5.9604644775390625e-8f;
            memcpy(&b, &v, 4);
            b |= sign;
        }
    } else if (e == 31) {
        b = sign | 0x7f800000u | (m << 13);
    } else {
        b = sign | ((e + 112u) << 23) | (m << 13);
    }
    float out;
    memcpy(&out, &b, 4);
    return out;
}

float orc_round_half(float x) { return f16_bits_to_f32(f32_to_f16_bits(x)); }

static inline float store_dual(float x, int half) { return half ? orc_round_half(x) : x; }

/* ---------------------------- PD-TV ---------------------------------------------- */
/* One voxel's dual ascent + projection (primal_dual...cu:66-123 3D, :305-358 2D).
 * g[] are forward differences; p[] in/out. nd = 2 or 3. */
static inline void pd_dual(float *p, const float *g, int nd, float sigma, int methodTV)
{
    for (int c = 0; c < nd; ++c)
        p[c] = fmaf(sigma, g[c], p[c]);
    if (!methodTV) {
        float nrm = p[0] * p[0];
        for (int c = 1; c < nd; ++c)
            nrm = fmaf(p[c], p[c], nrm);
        if (nrm > 1.0f) {
            float r = 1.0f / sqrtf(nrm);
            for (int c = 0; c < nd; ++c)
                p[c] *= r;
        }
    } else {
        for (int c = 0; c < nd; ++c) {
            float v = fabsf(p[c]);
            if (v < 1.0f) v = 1.0f;
            p[c] /= v;
        }
    }
}

/* forward difference with the reference's far-edge rule: at the last index the
 * "next" sample is the previous one (primal_dual...cu:216-220), and the previous
 * sample of index 0 is 0 (:147-160). */
static inline float pd_fwd(const float *U, size_t idx, int i, int dim, size_t stride)
{
    float u = U[idx];
    float nxt;
    if (i == dim - 1)
        nxt = (i > 0) ? U[idx - stride] : 0.0f;
    else
        nxt = U[idx + stride];
    return nxt - u;
}

/* in/out: [dz][dy][dx] (dz==1 with nd==2 -> 2D kernel on [dy][dx]).
 * Returns 0 on success. Result = U_arrays[iters % 2] as in regularisersCuPy.py:293-296. */
int orc_pdtv(const float *in, float *out, int dx, int dy, int dz, int nd,
             float sigma, float tau, float lt, float theta,
             int iters, int methodTV, int nonneg, int half)
{
    if (nd == 2) dz = 1;
    const size_t sx = 1, sy = (size_t)dx, sz = (size_t)dx * dy;
    const size_t nvox = sz * (size_t)dz;
    float *U[2], *P[3], *Pn[3];
    U[0] = (float *)malloc(nvox * sizeof(float));
    U[1] = (float *)calloc(nvox, sizeof(float));
    for (int c = 0; c < 3; ++c) {
        P[c] = (float *)calloc(nvox, sizeof(float));  /* stored duals (already rounded if half) */
        Pn[c] = (float *)calloc(nvox, sizeof(float)); /* this iteration's un-rounded duals */
    }
    memcpy(U[0], in, nvox * sizeof(float));
    const float inv_den = 1.0f + lt;
    for (int it = 0; it < iters; ++it) {
        const float *Ui = U[it & 1];
        float *Uo = U[(it + 1) & 1];
        /* phase 1: every voxel's updated dual (what each thread recomputes for itself and
         * its -x/-y/-z neighbours at primal_dual...cu:215-252) */
#pragma omp parallel for collapse(2) schedule(static)
        for (int z = 0; z < dz; ++z)
            for (int y = 0; y < dy; ++y)
                for (int x = 0; x < dx; ++x) {
                    size_t idx = (size_t)x + sy * y + sz * z;
                    float g[3] = {0.0f, 0.0f, 0.0f}, p[3] = {0.0f, 0.0f, 0.0f};
                    g[0] = pd_fwd(Ui, idx, x, dx, sx);
                    g[1] = pd_fwd(Ui, idx, y, dy, sy);
                    if (nd == 3) g[2] = pd_fwd(Ui, idx, z, dz, sz);
                    for (int c = 0; c < nd; ++c) p[c] = P[c][idx];
                    pd_dual(p, g, nd, sigma, methodTV);
                    for (int c = 0; c < nd; ++c) Pn[c][idx] = p[c];
                }
        /* phase 2: primal step from the backward-difference divergence (:116-123,:254-256) */
#pragma omp parallel for collapse(2) schedule(static)
        for (int z = 0; z < dz; ++z)
            for (int y = 0; y < dy; ++y)
                for (int x = 0; x < dx; ++x) {
                    size_t idx = (size_t)x + sy * y + sz * z;
                    float u = Ui[idx];
                    if (nonneg && u < 0.0f) u = 0.0f;
                    float pv1 = -(Pn[0][idx] - (x > 0 ? Pn[0][idx - sx] : 0.0f));
                    float pv2 = -(Pn[1][idx] - (y > 0 ? Pn[1][idx - sy] : 0.0f));
                    float div = pv1 + pv2;
                    if (nd == 3) {
                        float pv3 = -(Pn[2][idx] - (z > 0 ? Pn[2][idx - sz] : 0.0f));
                        div = div + pv3;
                    }
                    float t = fmaf(-tau, div, u);
                    t = fmaf(lt, in[idx], t);
                    float nu_ = t / inv_den;
                    Uo[idx] = fmaf(theta, nu_ - u, nu_);
                }
        for (int c = 0; c < nd; ++c) {
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < nvox; ++i)
                P[c][i] = store_dual(Pn[c][i], half);
        }
    }
    memcpy(out, U[iters & 1], nvox * sizeof(float));
    free(U[0]); free(U[1]);
    for (int c = 0; c < 3; ++c) { free(P[c]); free(Pn[c]); }
    return 0;
}

/* ---------------------------- ROF-TV --------------------------------------------- */
static inline float rof_minmod_sq(float n0, float n1)
{
    /* rudin_osher...cu:51-55; 0.5*(sign+sign) is exactly -1, 0 or +1 */
    int sg = ((n1 > 0) - (n1 < 0)) + ((n0 > 0) - (n0 < 0));
    float a = fabsf(n1), b = fabsf(n0);
    float m = (float)(0.5 * (double)sg * (double)(a < b ? a : b));
    return m * m;
}

static inline float rof_norm(float nom, float d1, float d2, float d3)
{
    /* rudin_osher...cu:57-61: float adds, then + 1.0e-8 in double, then float sqrt */
    float s = (d1 + d2) + d3;
    float den = sqrtf((float)((double)s + 1.0e-8));
    return nom / den;
}

static inline int refl_hi(int i, int dim) { return i == dim - 1 ? i - 1 : i + 1; }
static inline int refl_lo(int i) { return i == 0 ? i + 1 : i - 1; }

int orc_roftv(const float *in, float *out, int dx, int dy, int dz, int nd,
              float lambda, float tau, int iters, int half)
{
    if (nd == 2) dz = 1;
    if (dx < 2 || dy < 2 || (nd == 3 && dz < 2)) return -1; /* reflecting index leaves the array in the reference */
    const size_t sy = (size_t)dx, sz = (size_t)dx * dy;
    const size_t nvox = sz * (size_t)dz;
    float *U[2], *D[3];
    U[0] = (float *)malloc(nvox * sizeof(float));
    U[1] = (float *)calloc(nvox, sizeof(float));
    for (int c = 0; c < 3; ++c) D[c] = (float *)calloc(nvox, sizeof(float));
    memcpy(U[0], in, nvox * sizeof(float));
    for (int it = 0; it < iters; ++it) {
        const float *Ui = U[it & 1];
        float *Uo = U[(it + 1) & 1];
#pragma omp parallel for collapse(2) schedule(static)
        for (int k = 0; k < dz; ++k)
            for (int j = 0; j < dy; ++j)
                for (int i = 0; i < dx; ++i) {
                    size_t idx = (size_t)i + sy * j + sz * k;
                    float u = Ui[idx];
                    /* reference naming: "x" pairs with j (dimY), "y" with i (dimX); :183-188 */
                    float nx1 = Ui[(size_t)i + sy * refl_hi(j, dy) + sz * k] - u;
                    float ny1 = Ui[(size_t)refl_hi(i, dx) + sy * j + sz * k] - u;
                    float nx0 = u - Ui[(size_t)i + sy * refl_lo(j) + sz * k];
                    float ny0 = u - Ui[(size_t)refl_lo(i) + sy * j + sz * k];
                    float dxm = rof_minmod_sq(nx0, nx1);
                    float dym = rof_minmod_sq(ny0, ny1);
                    if (nd == 3) {
                        float nz1 = Ui[(size_t)i + sy * j + sz * refl_hi(k, dz)] - u;
                        float nz0 = u - Ui[(size_t)i + sy * j + sz * refl_lo(k)];
                        float dzm = rof_minmod_sq(nz0, nz1);
                        D[0][idx] = store_dual(rof_norm(nx1, nx1 * nx1, dym, dzm), half);
                        D[1][idx] = store_dual(rof_norm(ny1, dxm, ny1 * ny1, dzm), half);
                        D[2][idx] = store_dual(rof_norm(nz1, dxm, dym, nz1 * nz1), half);
                    } else {
                        D[0][idx] = store_dual(rof_norm(nx1, nx1 * nx1, dym, 0.0f), half);
                        D[1][idx] = store_dual(rof_norm(ny1, dxm, ny1 * ny1, 0.0f), half);
                    }
                }
#pragma omp parallel for collapse(2) schedule(static)
        for (int k = 0; k < dz; ++k)
            for (int j = 0; j < dy; ++j)
                for (int i = 0; i < dx; ++i) {
                    size_t idx = (size_t)i + sy * j + sz * k;
                    float u = Ui[idx];
                    float dv = (D[0][idx] - D[0][(size_t)i + sy * refl_lo(j) + sz * k]) +
                               (D[1][idx] - D[1][(size_t)refl_lo(i) + sy * j + sz * k]);
                    if (nd == 3)
                        dv = dv + (D[2][idx] - D[2][(size_t)i + sy * j + sz * refl_lo(k)]);
                    float t = fmaf(lambda, dv, -(u - in[idx]));
                    Uo[idx] = fmaf(tau, t, u);
                }
    }
    memcpy(out, U[iters & 1], nvox * sizeof(float));
    free(U[0]); free(U[1]);
    for (int c = 0; c < 3; ++c) free(D[c]);
    return 0;
}

/* ---------------------------- single iterations on z-slabs with ghost planes --------------------------------
 * Test infrastructure for the multi-GPU row: the same arithmetic as orc_pdtv / orc_roftv, one iteration,
 * on arrays [planes][dy][dx] whose first `out_begin` and last `planes - out_end` planes are read-only ghosts.
 * first_edge / last_edge say whether plane 0 / plane planes-1 is the GLOBAL first / last slice
 * (the zIndex > 0 and last_z tests of primal_dual...cu:188,213; the reflecting k indices of rudin_osher...cu:170-171,230). */
void orc_pdtv_step(const float *in, const float *Ui, float *Uo, const float *P1i, const float *P2i, const float *P3i,
                   float *P1o, float *P2o, float *P3o, int dx, int dy, int planes, int out_begin, int out_end,
                   int first_edge, int last_edge, float sigma, float tau, float lt, float theta, int methodTV,
                   int nonneg, int half)
{
    const size_t sy = (size_t)dx, sz = (size_t)dx * dy;
    const float *Pi[3] = {P1i, P2i, P3i};
    float *Po[3] = {P1o, P2o, P3o};
    const int zlo = out_begin > 0 ? out_begin - 1 : 0; /* duals are needed one plane below the first output plane */
    float *Pn[3];
    for (int c = 0; c < 3; ++c) Pn[c] = (float *)calloc(sz * (size_t)planes, sizeof(float));
    (void)first_edge;
    for (int z = zlo; z < out_end; ++z)
        for (int y = 0; y < dy; ++y)
            for (int x = 0; x < dx; ++x) {
                size_t idx = (size_t)x + sy * y + sz * z;
                float g[3], p[3];
                g[0] = pd_fwd(Ui, idx, x, dx, 1);
                g[1] = pd_fwd(Ui, idx, y, dy, sy);
                if (z == planes - 1 && last_edge) g[2] = ((z > 0) ? Ui[idx - sz] : 0.0f) - Ui[idx];
                else g[2] = Ui[idx + sz] - Ui[idx];
                for (int c = 0; c < 3; ++c) p[c] = Pi[c][idx];
                pd_dual(p, g, 3, sigma, methodTV);
                for (int c = 0; c < 3; ++c) Pn[c][idx] = p[c];
            }
    for (int z = out_begin; z < out_end; ++z)
        for (int y = 0; y < dy; ++y)
            for (int x = 0; x < dx; ++x) {
                size_t idx = (size_t)x + sy * y + sz * z;
                float u = Ui[idx];
                if (nonneg && u < 0.0f) u = 0.0f;
                float pv1 = -(Pn[0][idx] - (x > 0 ? Pn[0][idx - 1] : 0.0f));
                float pv2 = -(Pn[1][idx] - (y > 0 ? Pn[1][idx - sy] : 0.0f));
                float pv3 = -(Pn[2][idx] - (z > 0 ? Pn[2][idx - sz] : 0.0f));
                float div = (pv1 + pv2) + pv3;
                float t = fmaf(-tau, div, u);
                t = fmaf(lt, in[idx], t);
                float nu_ = t / (1.0f + lt);
                Uo[idx] = fmaf(theta, nu_ - u, nu_);
                for (int c = 0; c < 3; ++c) Po[c][idx] = store_dual(Pn[c][idx], half);
            }
    for (int c = 0; c < 3; ++c) free(Pn[c]);
}

void orc_roftv_step(const float *in, const float *Ui, float *Uo, int dx, int dy, int planes, int out_begin,
                    int out_end, int first_edge, int last_edge, float lambda, float tau, int half)
{
    const size_t sy = (size_t)dx, sz = (size_t)dx * dy;
    float *D[3];
    for (int c = 0; c < 3; ++c) D[c] = (float *)calloc(sz * (size_t)planes, sizeof(float));
    const int zlo = out_begin > 0 ? out_begin - 1 : 0;
    for (int k = zlo; k < out_end; ++k) {
        const int k_hi = (k == planes - 1 && last_edge) ? k - 1 : k + 1;
        const int k_lo = (k == 0 && first_edge) ? k + 1 : k - 1;
        for (int j = 0; j < dy; ++j)
            for (int i = 0; i < dx; ++i) {
                size_t idx = (size_t)i + sy * j + sz * k;
                float u = Ui[idx];
                float nx1 = Ui[(size_t)i + sy * refl_hi(j, dy) + sz * k] - u;
                float ny1 = Ui[(size_t)refl_hi(i, dx) + sy * j + sz * k] - u;
                float nx0 = u - Ui[(size_t)i + sy * refl_lo(j) + sz * k];
                float ny0 = u - Ui[(size_t)refl_lo(i) + sy * j + sz * k];
                float nz1 = Ui[(size_t)i + sy * j + sz * k_hi] - u;
                float nz0 = u - Ui[(size_t)i + sy * j + sz * k_lo];
                float dxm = rof_minmod_sq(nx0, nx1), dym = rof_minmod_sq(ny0, ny1), dzm = rof_minmod_sq(nz0, nz1);
                D[0][idx] = store_dual(rof_norm(nx1, nx1 * nx1, dym, dzm), half);
                D[1][idx] = store_dual(rof_norm(ny1, dxm, ny1 * ny1, dzm), half);
                D[2][idx] = store_dual(rof_norm(nz1, dxm, dym, nz1 * nz1), half);
            }
    }
    for (int k = out_begin; k < out_end; ++k) {
        const int k_lo = (k == 0 && first_edge) ? k + 1 : k - 1;
        for (int j = 0; j < dy; ++j)
            for (int i = 0; i < dx; ++i) {
                size_t idx = (size_t)i + sy * j + sz * k;
                float u = Ui[idx];
                float dv = (D[0][idx] - D[0][(size_t)i + sy * refl_lo(j) + sz * k]) +
                           (D[1][idx] - D[1][(size_t)refl_lo(i) + sy * j + sz * k]);
                dv = dv + (D[2][idx] - D[2][(size_t)i + sy * j + sz * k_lo]);
                float t = fmaf(lambda, dv, -(u - in[idx]));
                Uo[idx] = fmaf(tau, t, u);
            }
    }
    for (int c = 0; c < 3; ++c) free(D[c]);
}

int orc_abi_version(void) { return 1; }

/* Thread count of the OpenMP loops above.  The host may show far more logical CPUs than its cgroup quota lets the
 * process use (a 256-thread box with a 16-CPU quota): 256 spinning threads on 16 CPUs' worth of time turn a 0.3 s
 * call into 50 s.  tomo_oracle.py sets this to the usable CPU count when OMP_NUM_THREADS is not given. */
void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
