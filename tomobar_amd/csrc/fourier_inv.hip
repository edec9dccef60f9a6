// Fourier reconstruction on unequally spaced grids for gfx950.
//
// What is computed is fixed by the reference: tomobar/methodsDIR_CuPy.py:152-447 (driver) with the stages :449-545
// (oversampled FBP filter), :645-683 (two slices -> one complex slice), :701-836 (1D FFT + gathering on the 2n x 2n
// frequency grid), :851-897 (2D inverse FFT), :920-967 (unpadding x phi) and the kernels of
// tomobar/cuda_kernels/fft_us_kernels.cu.  Restated on the CPU in oracle/fourier_oracle.py (pinned to the reference's own
// output, tests/golden/fourier_golden.npz).
//
// How it is computed here (MI355X-first, not the reference's thread-per-(grid point, slice) kernels):
//   * the slice-pair index z is the FASTEST dimension of everything between the 1D and the 2D FFT: polar samples
//     g[angle][radius][z] 64 slice pairs (128 slices) per chunk.  A wave owns one
//     grid point at a time with one lane per slice pair: the geometry of the gathering (which angles pass within the
//     support radius, which radial samples, the Gaussian weights -- the reference recomputes all of it per slice) is
//     wave-uniform and computed once for 64 slice pairs (lane-parallel over candidate rays / radial samples), every
//     sample fetch is one coalesced 512-byte row shared by the 8 grid points of the wave, and the sum
//     runs in the reference's order (angles ascending, radius ascending), so the result is deterministic (the reference's
//     small-detector path accumulates with atomics);
//   * a wave leaves its 16 grid points x 64 slice pairs through an LDS transpose, as 128-byte lines of f[z][ky][kx], so
//     the 2D inverse FFT is a plain contiguous batch (the strided form cost 2.2x the gathering in hipFFT transposes);
//   * both checkerboard shifts are folded into the neighbouring kernels (store of the gathering, unpadding), the
//     slice pairing + first shift into the crop of the filter output, the second 1D shift + 4/n scale into the transpose.
// FFTs are library calls (hipFFT), like the reference's cuFFT via CuPy.  No MFMA: no dense contraction.
#include "tomo_common.h"

#include <hipfft/hipfft.h>

#include <algorithm>
#include <cmath>
#include <mutex>
#include <vector>

namespace {

#define TOMO_FFT(expr)                                                                              \
    do {                                                                                            \
        hipfftResult r_ = (expr);                                                                   \
        if (r_ != HIPFFT_SUCCESS) { rc = tomo_fail(TOMO_E_RUNTIME, "%s failed: hipfft status %d", #expr, (int)r_); goto done; } \
    } while (0)
#define TOMO_HIPG(expr)                                                                             \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) { rc = tomo_fail(TOMO_E_RUNTIME, "%s failed: %s", #expr, hipGetErrorString(e_)); goto done; } \
    } while (0)

constexpr int FZ = 64;  // slice pairs per chunk = lanes of a wave

// hipFFT plans are kept between calls (creation runs the library's run-time kernel generation and allocates work areas:
// tens of milliseconds per call otherwise).  Released together with the scratch arena (tomo_release_scratch).
struct PlanKey {
    int device, kind, len, batch;
    bool operator==(const PlanKey &o) const { return device == o.device && kind == o.kind && len == o.len && batch == o.batch; }
};
struct PlanEntry { PlanKey key; hipfftHandle h; };
std::mutex g_plan_mu;
std::vector<PlanEntry> g_plans;
enum { PLAN_C2C_1D = 2, PLAN_C2C_2D = 3 };

int get_plan(int device, int kind, int len, int batch, hipStream_t st, hipfftHandle *out)
{
    std::lock_guard<std::mutex> lk(g_plan_mu);
    const PlanKey key{device, kind, len, batch};
    for (auto &e : g_plans)
        if (e.key == key) {
            if (hipfftSetStream(e.h, st) != HIPFFT_SUCCESS) return tomo_fail(TOMO_E_RUNTIME, "hipfftSetStream failed");
            *out = e.h;
            return TOMO_OK;
        }
    hipfftHandle h = 0;
    int dims[2] = {len, len};
    hipfftResult r;
    if (kind == PLAN_C2C_1D) r = hipfftPlanMany(&h, 1, dims, nullptr, 1, len, nullptr, 1, len, HIPFFT_C2C, batch);
    else r = hipfftPlanMany(&h, 2, dims, nullptr, 1, len * len, nullptr, 1, len * len, HIPFFT_C2C, batch);
    if (r != HIPFFT_SUCCESS) return tomo_fail(TOMO_E_RUNTIME, "hipfftPlanMany(kind %d, len %d, batch %d) failed: %d", kind, len, batch, (int)r);
    if (hipfftSetStream(h, st) != HIPFFT_SUCCESS) { (void)hipfftDestroy(h); return tomo_fail(TOMO_E_RUNTIME, "hipfftSetStream failed"); }
    g_plans.push_back(PlanEntry{key, h});
    *out = h;
    return TOMO_OK;
}
constexpr float PI_F = 3.1415926535897932384626433832795f;  // fft_us_kernels.cu:2

// ---- filter stage -------------------------------------------------------------------------------------------------
// The filter of methodsDIR_CuPy.py:479-534 has a REAL impulse response (a real, even window times the phase ramp of a
// shift), so the two slices of a pair can be filtered together as one complex signal a + i b: one complex transform per
// pair instead of two real ones (hipFFT's real transforms are a half-length complex transform plus a pre/post pass that
// cost more than the transform itself here).  The spectrum of the pair is multiplied by the Hermitian extension of the
// half-spectrum table.
// cbuf[r][j] = (a[clamp(j - pad_m)], b[clamp(j - pad_m)]): edge padding (cp.pad(..., mode="edge"), :522-530) + pairing;
// pair-row r = z_local * nproj + p, a = slice 2(c0 + z_local), b = the next slice.
__global__ __launch_bounds__(256) void pad_pair_kernel(const float *__restrict__ in, float2 *__restrict__ cbuf, size_t rows,
                                                       size_t row_first, int nproj, int raw_n, int ne, int pad_m)
{
    const size_t total = rows * (size_t)ne, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t r = i / ne;
        const int j = (int)(i - r * ne);
        const size_t row = row_first + r;
        const size_t z = row / nproj, p = row - z * nproj;
        const int x = min(max(j - pad_m, 0), raw_n - 1);
        const float *pa = in + ((2 * z) * nproj + p) * raw_n + x;
        cbuf[i] = make_float2(pa[0], pa[(size_t)nproj * raw_n]);
    }
}

__global__ __launch_bounds__(256) void mul_filter_kernel(float2 *spec, const float2 *__restrict__ w, size_t rows, int ne)
{
    const size_t total = rows * (size_t)ne, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const float2 g = w[i % ne], v = spec[i];
        spec[i] = make_float2(g.x * v.x - g.y * v.y, g.x * v.y + g.y * v.x);
    }
}

// Centre crop of the filtered pair-rows, 1/ne of the unnormalised inverse transform and the first fftshift sign
// (r2c_c1dfftshift, fft_us_kernels.cu:519-548) into datac[z][p][x].
__global__ __launch_bounds__(256) void crop_pair_kernel(const float2 *__restrict__ cbuf, float2 *__restrict__ datac, size_t rows,
                                                        size_t row_first, int n, int ne, int unpad_m, float inv_ne)
{
    const size_t total = rows * (size_t)n, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t r = i / n;
        const int x = (int)(i - r * n);
        const float s = ((x & 1) ? 1.0f : -1.0f) * inv_ne;
        const float2 v = cbuf[r * ne + unpad_m + x];
        datac[(row_first + r) * n + x] = make_float2(v.x * s, v.y * s);
    }
}

// g[p][x][z] = datac[z][p][x] * (4/n) * sign(x)   (c1dfftshift, fft_us_kernels.cu:550-577), z-fastest, zero beyond zc
__global__ __launch_bounds__(256) void transpose_scale_kernel(const float2 *__restrict__ datac, float2 *__restrict__ g, int zc,
                                                              int nproj, int n, float constant)
{
    __shared__ float2 tile[FZ][65];
    const int p = blockIdx.y, x0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int z = wave; z < FZ; z += 4) {
        float2 v = make_float2(0.0f, 0.0f);
        const int x = x0 + lane;
        if (z < zc && x < n) {
            v = datac[((size_t)z * nproj + p) * n + x];
            const float sgn = (x & 1) ? 1.0f : -1.0f;
            v.x = v.x * constant * sgn;
            v.y = v.y * constant * sgn;
        }
        tile[z][lane] = v;
    }
    __syncthreads();
    for (int xx = wave; xx < 64; xx += 4) {
        const int x = x0 + xx;
        if (x < n) g[((size_t)p * n + x) * FZ + lane] = tile[lane][xx];
    }
}

// ---- gathering ----------------------------------------------------------------------------------------------------
struct GatherArgs {
    const float2 *g;       // [nproj][n][FZ]
    float2 *f;             // [zc][2n][2n]  (layout of the 2D FFT)
    const float *ct, *st;  // cos / sin of theta[j]          (original angle order)
    const float *sth;      // theta sorted ascending
    const int *order;      // original index of the k-th smallest theta
    int nproj, n, m, center_size, use_center;
    float mu;
};

__device__ __forceinline__ float clamp_half(float v) { return v >= 0.5f ? 0.5f - 1e-5f : v; }

__device__ __forceinline__ int lower_bound_f(const float *a, int n, float v)  // first index with a[i] >= v
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// One grid point in the reference's "centre" form (gather_kernel_center, fft_us_kernels.cu:376-517): circular support.
__device__ __forceinline__ void gather_center_point(const GatherArgs &a, int tx, int ty, int lane, float2 &acc)
{
    const int n = a.n;
    const float coeff0 = PI_F / a.mu, coeff1 = -PI_F * PI_F / a.mu;
    const float fs2 = (float)(4 * n) * (float)n;
    const float radius_2 = 2.0f * ((float)a.m + 0.5f) * ((float)a.m + 0.5f) / fs2;
    const float px = (float)(tx - n) / (float)(2 * n), py = (float)(n - ty) / (float)(2 * n);
    const float rho2 = px * px + py * py;

    // Sorted angles [k0, k1): the per-ray geometry (distance test, radial range) is evaluated LANE-PARALLEL, one
    // candidate ray per lane; the rays that pass are then visited in ascending order (the reference's summation order),
    // their Gaussian weights again lane-parallel (one radial sample per lane, a ray has < 64 of them), and only the
    // final multiply-accumulate over the 64 slice pairs runs once per sample: weight from a v_readlane, one 512-byte row.
    auto angle_range = [&](int k0, int k1) {
        for (int kb = k0; kb < k1; kb += 64) {
            const int k = kb + lane;
            const int pi = a.order[min(k, k1 - 1)];
            const float costheta = a.ct[pi], sintheta = a.st[pi];
            const float vx = 0.5f * costheta, vy = 0.5f * sintheta;
            const float dot = vx * px + vy * py;
            const float mx = dot * vx / 0.25f, my = dot * vy / 0.25f;
            const float d2 = (mx - px) * (mx - px) + (my - py) * (my - py);
            const bool hit = (k < k1) && (radius_2 >= d2);
            const float dti = sqrtf(fmaxf(radius_2 - d2, 0.0f));
            int rmin, rmax;
            if (fabsf(vx) > fabsf(vy)) {
                rmin = n / 2 - 1 + (int)floorf((mx - dti * vx / 0.5f) / (2.0f * vx / (float)n));
                rmax = n / 2 + 1 + (int)floorf((mx + dti * vx / 0.5f) / (2.0f * vx / (float)n));
            } else {
                rmin = n / 2 - 1 + (int)floorf((my - dti * vy / 0.5f) / (2.0f * vy / (float)n));
                rmax = n / 2 + 1 + (int)floorf((my + dti * vy / 0.5f) / (2.0f * vy / (float)n));
            }
            if (rmin > rmax) { const int t = rmax; rmax = rmin; rmin = t; }
            rmin = min(max(rmin, 0), n - 1);
            rmax = min(max(rmax, 0), n - 1);
            unsigned long long todo = __builtin_amdgcn_ballot_w64(hit);
            while (todo) {
                const int l = __builtin_ctzll(todo);
                todo &= todo - 1;
                const int pi_l = __builtin_amdgcn_readlane(pi, l);
                const int r0 = __builtin_amdgcn_readlane(rmin, l), r1 = __builtin_amdgcn_readlane(rmax, l);
                const float c_l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(costheta), l));
                const float s_l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sintheta), l));
                const float2 *row = a.g + ((size_t)pi_l * n) * FZ + lane;
                for (int rb = r0; rb < r1; rb += 64) {  // one radial sample per lane (a ray has ~10; the loop is for safety)
                    const int ri = rb + lane;
                    const float rr = (float)(ri - n / 2) / (float)n;
                    const float x0 = clamp_half(rr * c_l), y0 = clamp_half(rr * s_l);
                    const float w0 = px - x0, w1 = py - y0;
                    const float w = coeff0 * __expf(coeff1 * (w0 * w0 + w1 * w1));
                    const int cnt = min(64, r1 - rb);
                    for (int j = 0; j < cnt; ++j) {
                        const float wj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), j));
                        const float2 v = row[(size_t)(rb + j) * FZ];
                        acc.x += v.x * wj;
                        acc.y += v.y * wj;
                    }
                }
            }
        }
    };

    if (radius_2 >= rho2) {  // every ray passes within the support radius
        angle_range(0, a.nproj);
        return;
    }
    // rays within asin(R / rho) of the direction of the point, modulo pi; a small margin, the exact test is in angle_range
    const float rho = sqrtf(rho2);
    const float delta = asinf(fminf(sqrtf(radius_2) / rho, 1.0f)) + 2.0e-3f;
    if (2.0f * delta >= PI_F) {
        angle_range(0, a.nproj);
        return;
    }
    const float phi = atan2f(py, px);
    const float tmin = a.sth[0], tmax = a.sth[a.nproj - 1];
    // centres phi + j*pi, ascending j, such that [c - delta, c + delta] meets [tmin, tmax]
    int j = (int)ceilf((tmin - delta - phi) / PI_F);
    int done_until = 0;  // sorted indices below this have been visited (keeps the ranges disjoint)
    for (;; ++j) {
        const float c = phi + (float)j * PI_F;
        if (c - delta > tmax) break;
        const int k0 = max(lower_bound_f(a.sth, a.nproj, c - delta), done_until);
        const int k1 = lower_bound_f(a.sth, a.nproj, c + delta);  // exclusive
        if (k1 > k0) angle_range(k0, k1);
        done_until = max(done_until, k1);
    }
}

// The same for the GPB consecutive grid points (tx0 .. tx0+GPB-1, ty) of a wave at once.  Neighbouring points draw on
// nearly the same samples: the per-point form re-reads every 512-byte sample row once per point (PMC: 80 % L1 hits but
// 216 GB of L2 traffic per 128-slice chunk at 2048 x 1800, waves waiting 75 % of the time).  Here a ray's rows are read
// once for the union of the points' radial ranges and multiplied into GPB accumulators with per-point weights (zero
// outside a point's own range: adds exactly nothing, the per-point order angle-ascending / radius-ascending is kept).
// ``pmask`` bit q = point q takes part (inside the grid and the centre box).
constexpr int GPB = 8;

__device__ __forceinline__ void gather_center_block(const GatherArgs &a, int tx0, int ty, unsigned pmask, int lane, float2 (&acc)[GPB])
{
    const int n = a.n;
    const float coeff0 = PI_F / a.mu, coeff1 = -PI_F * PI_F / a.mu;
    const float fs2 = (float)(4 * n) * (float)n;
    const float radius_2 = 2.0f * ((float)a.m + 0.5f) * ((float)a.m + 0.5f) / fs2;
    const float py = (float)(n - ty) / (float)(2 * n);
    float pxq[GPB];
#pragma unroll
    for (int q = 0; q < GPB; ++q) pxq[q] = (float)(tx0 + q - n) / (float)(2 * n);

    auto angle_range = [&](int k0, int k1) {
        for (int kb = k0; kb < k1; kb += 64) {
            const int k = kb + lane;
            const int pi = a.order[min(k, k1 - 1)];
            const float costheta = a.ct[pi], sintheta = a.st[pi];
            const float vx = 0.5f * costheta, vy = 0.5f * sintheta;
            const bool xdom = fabsf(vx) > fabsf(vy);
            int lo_hi[GPB];          // rmin | rmax << 16 per point (n <= 16384)
            unsigned hits = 0;
            int umin = n, umax = 0;
#pragma unroll
            for (int q = 0; q < GPB; ++q) {
                const float px = pxq[q];
                const float dot = vx * px + vy * py;
                const float mx = dot * vx / 0.25f, my = dot * vy / 0.25f;
                const float d2 = (mx - px) * (mx - px) + (my - py) * (my - py);
                const bool hit = (k < k1) && ((pmask >> q) & 1u) && (radius_2 >= d2);
                const float dti = sqrtf(fmaxf(radius_2 - d2, 0.0f));
                int rmin, rmax;
                if (xdom) {
                    rmin = n / 2 - 1 + (int)floorf((mx - dti * vx / 0.5f) / (2.0f * vx / (float)n));
                    rmax = n / 2 + 1 + (int)floorf((mx + dti * vx / 0.5f) / (2.0f * vx / (float)n));
                } else {
                    rmin = n / 2 - 1 + (int)floorf((my - dti * vy / 0.5f) / (2.0f * vy / (float)n));
                    rmax = n / 2 + 1 + (int)floorf((my + dti * vy / 0.5f) / (2.0f * vy / (float)n));
                }
                if (rmin > rmax) { const int t = rmax; rmax = rmin; rmin = t; }
                rmin = min(max(rmin, 0), n - 1);
                rmax = min(max(rmax, 0), n - 1);
                if (!hit) rmax = rmin;  // empty range
                lo_hi[q] = rmin | (rmax << 16);
                if (hit && rmax > rmin) {
                    hits |= 1u << q;
                    umin = min(umin, rmin);
                    umax = max(umax, rmax);
                }
            }
            unsigned long long todo = __builtin_amdgcn_ballot_w64(hits != 0);
            while (todo) {
                const int l = __builtin_ctzll(todo);
                todo &= todo - 1;
                const int pi_l = __builtin_amdgcn_readlane(pi, l);
                const int u0 = __builtin_amdgcn_readlane(umin, l), u1 = __builtin_amdgcn_readlane(umax, l);
                const float c_l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(costheta), l));
                const float s_l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sintheta), l));
                int r_q[GPB];
#pragma unroll
                for (int q = 0; q < GPB; ++q) r_q[q] = __builtin_amdgcn_readlane(lo_hi[q], l);
                const float2 *row = a.g + ((size_t)pi_l * n) * FZ + lane;
                for (int rb = u0; rb < u1; rb += 64) {  // one radial sample per lane (the union has ~20)
                    const int ri = rb + lane;
                    const float rr = (float)(ri - n / 2) / (float)n;
                    const float x0 = clamp_half(rr * c_l), y0 = clamp_half(rr * s_l);
                    const float w1 = py - y0;
                    float w[GPB];
#pragma unroll
                    for (int q = 0; q < GPB; ++q) {
                        const float w0 = pxq[q] - x0;
                        const float wv = coeff0 * __expf(coeff1 * (w0 * w0 + w1 * w1));
                        w[q] = (ri >= (r_q[q] & 0xffff) && ri < (r_q[q] >> 16)) ? wv : 0.0f;
                    }
                    const int cnt = min(64, u1 - rb);
                    for (int j = 0; j < cnt; ++j) {
                        const float2 v = row[(size_t)(rb + j) * FZ];
#pragma unroll
                        for (int q = 0; q < GPB; ++q) {
                            const float wj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w[q]), j));
                            acc[q].x += v.x * wj;
                            acc[q].y += v.y * wj;
                        }
                    }
                }
            }
        }
    };

    // angular pruning for the whole block: a circle around its middle that contains every point's support circle
    const float pxc = 0.5f * (pxq[0] + pxq[GPB - 1]);
    const float rho = sqrtf(pxc * pxc + py * py);
    const float rblk = sqrtf(radius_2) + (float)(GPB / 2 + 1) / (float)(2 * n);
    if (rblk >= rho) {
        angle_range(0, a.nproj);
        return;
    }
    const float delta = asinf(fminf(rblk / rho, 1.0f)) + 2.0e-3f;
    if (2.0f * delta >= PI_F) {
        angle_range(0, a.nproj);
        return;
    }
    const float phi = atan2f(py, pxc);
    const float tmin = a.sth[0], tmax = a.sth[a.nproj - 1];
    int j = (int)ceilf((tmin - delta - phi) / PI_F);
    int done_until = 0;
    for (;; ++j) {
        const float c = phi + (float)j * PI_F;
        if (c - delta > tmax) break;
        const int k0 = max(lower_bound_f(a.sth, a.nproj, c - delta), done_until);
        const int k1 = lower_bound_f(a.sth, a.nproj, c + delta);  // exclusive
        if (k1 > k0) angle_range(k0, k1);
        done_until = max(done_until, k1);
    }
}

// One grid point in the reference's scatter form seen from the receiving side (gather_kernel / gather_kernel_partial,
// fft_us_kernels.cu:5-113): a sample contributes when the point lies in its (2m+1)^2 footprint (periodic wrap).
__device__ __forceinline__ void gather_square_point(const GatherArgs &a, int ix, int iy, int lane, float2 &acc)
{
    const int n = a.n, m = a.m, two_n = 2 * n;
    const float coeff0 = PI_F / a.mu, coeff1 = -PI_F * PI_F / a.mu;
    const int l0 = ix - n, l1 = iy - n;  // unwrapped cell indices in [-n, n)
    for (int pi = 0; pi < a.nproj; ++pi) {
        const float costheta = a.ct[pi], sintheta = a.st[pi];
        // dominant axis of the ray: 2n * r * c must land in [L - m, L + m + 1) for an image L = l + 2n*k of the cell
        const bool xdom = fabsf(costheta) >= fabsf(sintheta);
        const float c = xdom ? costheta : -sintheta;
        const int l = xdom ? l0 : l1;
        int prev_hi = -1;
        const float sgn = c >= 0.0f ? 1.0f : -1.0f;
        // images in the order of increasing r
        for (int kk = -1; kk <= 1; ++kk) {
            const int k = c >= 0.0f ? kk : -kk;
            const float L = (float)(l + two_n * k);
            float r_lo = (L - (float)m) / ((float)two_n * c), r_hi = (L + (float)m + 1.0f) / ((float)two_n * c);
            if (sgn < 0.0f) { const float t = r_lo; r_lo = r_hi; r_hi = t; }
            if (r_hi < -0.52f || r_lo > 0.52f) continue;
            int t_lo = (int)floorf((float)(n / 2) + (float)n * r_lo) - 2, t_hi = (int)ceilf((float)(n / 2) + (float)n * r_hi) + 2;
            t_lo = max(max(t_lo, 0), prev_hi + 1);
            t_hi = min(t_hi, n - 1);
            if (t_hi < t_lo) continue;
            prev_hi = t_hi;
            const float2 *row = a.g + ((size_t)pi * n) * FZ + lane;
            for (int tx = t_lo; tx <= t_hi; ++tx) {
                const float rr = (float)(tx - n / 2) / (float)n;
                const float x0 = clamp_half(rr * costheta), y0 = clamp_half(-rr * sintheta);
                const int e0 = (int)floorf((float)two_n * x0), e1 = (int)floorf((float)two_n * y0);
                // footprint offset that maps onto this cell modulo 2n, if any
                int d0 = (ix - e0 + m - 3 * n) % two_n, d1 = (iy - e1 + m - 3 * n) % two_n;
                if (d0 < 0) d0 += two_n;
                if (d1 < 0) d1 += two_n;
                if (d0 > 2 * m || d1 > 2 * m) continue;
                const float w0 = (float)(e0 - m + d0) / (float)two_n - x0, w1 = (float)(e1 - m + d1) / (float)two_n - y0;
                const float w = coeff0 * __expf(coeff1 * (w0 * w0 + w1 * w1));
                const float2 v = row[(size_t)tx * FZ];
                acc.x += w * v.x;
                acc.y += w * v.y;
            }
        }
    }
}

constexpr int GP = 8;   // grid points per wave: 8 consecutive kx = one 64-byte sector of the frequency grid per slice pair

__global__ __launch_bounds__(256) void gather_kernel(GatherArgs a, int zc)
{
    __shared__ float2 stage[4][GP][FZ + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int two_n = 2 * a.n;
    const int iy = blockIdx.y;
    const int x_first = (blockIdx.x * 4 + wave) * GP;
    const int chs = a.center_size / 2, base = max(0, a.n - chs);
    static_assert(GP == GPB, "one block of points per wave");
    float2 accs[GPB];
    unsigned cmask = 0;  // points gathered in the centre form
#pragma unroll
    for (int q = 0; q < GPB; ++q) {
        accs[q] = make_float2(0.0f, 0.0f);
        const int ix = x_first + q;
        const bool in_box = a.use_center && ix < two_n && ix >= base && ix < base + a.center_size && iy >= base && iy < base + a.center_size;
        if (in_box) cmask |= 1u << q;
    }
    if (cmask) gather_center_block(a, x_first, iy, cmask, lane, accs);
#pragma unroll
    for (int q = 0; q < GPB; ++q) {
        const int ix = x_first + q;
        if (!((cmask >> q) & 1u) && ix < two_n) gather_square_point(a, ix, iy, lane, accs[q]);
        // first c2dfftshift (fft_us_kernels.cu:579-603) folded into the store
        const float chk = ((ix ^ iy) & 1) ? -1.0f : 1.0f;
        stage[wave][q][lane] = make_float2(accs[q].x * chk, accs[q].y * chk);
    }
    // the wave's GP x 64 results leave in the layout of the 2D FFT, f[z][ky][kx]: 4 slice pairs x 16 kx per store
    // (same-wave LDS exchange: no barrier needed, the compiler orders the ds operations of a wave)
    __builtin_amdgcn_wave_barrier();
    const int qx = lane & (GP - 1), zq = lane / GP;
#pragma unroll 4
    for (int z0 = 0; z0 < FZ; z0 += 64 / GP) {
        const int z = z0 + zq, ix = x_first + qx;
        if (z < zc && ix < two_n) a.f[((size_t)z * two_n + iy) * two_n + ix] = stage[wave][qx][z];
    }
}

// ---- unpadding: second c2dfftshift, 1/(2n)^2 of the unnormalised inverse transform, phi, slice un-pairing -----------
// out[slice0 + 2z (+1)][ry][rx] from f[z][ky][kx]   (unpadding_mul_phi, fft_us_kernels.cu:605-658)
__global__ __launch_bounds__(256) void unpad_kernel(const float2 *__restrict__ f, float *__restrict__ out, int n, int um, int size,
                                                    int slice0, int out_z, float mu, float phi_scale, float inv_norm)
{
    const int rxu = blockIdx.x * 256 + threadIdx.x, ry = blockIdx.y, z = blockIdx.z;
    if (rxu >= size) return;
    const int two_n = 2 * n;
    const int tx = n / 2 + um + rxu, ty = n / 2 + um + ry;
    float2 v = f[((size_t)z * two_n + ty) * two_n + tx];
    const float chk = ((tx ^ ty) & 1) ? -1.0f : 1.0f;
    const float dx = -0.5f + (float)(um + rxu) * 1.0f / (float)n, dy = -0.5f + (float)(um + ry) * 1.0f / (float)n;
    const float phi = expf(mu * (float)(n * n) * (dx * dx + dy * dy)) * phi_scale;
    const int s0 = slice0 + 2 * z;
    const size_t o = ((size_t)s0 * size + ry) * size + rxu;
    if (s0 < out_z) out[o] = v.x * inv_norm * chk * phi;
    if (s0 + 1 < out_z) out[o + (size_t)size * size] = v.y * inv_norm * chk * phi;
}

}  // namespace

void tomo_fourier_cache_release(int device)
{
    std::lock_guard<std::mutex> lk(g_plan_mu);
    for (size_t i = 0; i < g_plans.size();) {
        if (g_plans[i].key.device == device) {
            (void)hipfftDestroy(g_plans[i].h);
            g_plans.erase(g_plans.begin() + (long)i);
        } else {
            ++i;
        }
    }
}

extern "C" int tomo_fourier_inv(int device, const float *data_dev, float *out_dev, int nz, int out_z, int nproj, int raw_n,
                                int n, int ne, int unpad_m, int out_size, const float *w_host, const float *theta_host,
                                int m, float mu, int center_size, void *stream)
{
    TOMO_REQUIRE(device >= 0 && data_dev && out_dev && w_host && theta_host, "NULL argument");
    TOMO_REQUIRE(nz >= 2 && nz % 2 == 0 && raw_n >= 2 && raw_n % 2 == 0 && n >= raw_n && n % 2 == 0 && ne >= n && ne % 2 == 0,
                 "detector sizes must be even (the caller pads odd sizes) and raw_n <= n <= ne");
    TOMO_REQUIRE(nproj >= 1 && m >= 1 && m <= 64 && mu > 0.0f && out_size >= 1 && unpad_m >= 0 && unpad_m + out_size <= n,
                 "bad Fourier reconstruction parameters");
    TOMO_REQUIRE(out_z >= 1 && out_z <= nz && center_size >= 0 && center_size <= 2 * n, "bad output / centre size");
    TOMO_REQUIRE(2L * n <= 32768, "detector too wide for the Fourier reconstruction grid");
    TOMO_ON_DEVICE(device);
    hipStream_t st = as_stream(stream);
    int rc = TOMO_OK;

    const int nzh = nz / 2, nh = ne / 2 + 1, two_n = 2 * n;
    const int zc_max = std::min(nzh, FZ);
    // filter sub-chunks: at most ~64 Mi floats in the padded buffer
    const size_t chunk_rows = (size_t)zc_max * nproj;  // pair-rows
    const size_t rows_sub = std::max<size_t>(1, std::min<size_t>(chunk_rows, ((size_t)64 << 20) / ne));
    const size_t bytes_buf = rows_sub * (size_t)ne * sizeof(float2);
    const size_t bytes_spec = 0;
    const size_t bytes_datac = (size_t)zc_max * nproj * n * sizeof(float2);
    const size_t bytes_g = (size_t)nproj * n * FZ * sizeof(float2);
    const size_t bytes_f = (size_t)two_n * two_n * zc_max * sizeof(float2);
    const size_t bytes_tab = ((size_t)ne * sizeof(float2) + (size_t)nproj * 4 * sizeof(float) + (size_t)nproj * sizeof(int) + 1024);
    auto al = [](size_t b) { return (b + 255) / 256 * 256; };
    char *ws = nullptr;
    hipfftHandle p_filt = 0, p_c2c = 0, p_2d = 0;
    {   // workspace: the per-device grow-only scratch arena (shared with the TV operators, released by tomo_release_scratch)
        const size_t total = al(bytes_buf) + al(bytes_spec) + al(bytes_datac) + al(bytes_g) + al(bytes_f) + al(bytes_tab);
        void *base = nullptr;
        rc = tomo_arena_get(device, st, ARENA_MAIN, total, &base);
        if (rc != TOMO_OK) return rc;
        ws = (char *)base;
    }
    {
        float2 *cbuf = (float2 *)ws;
        float2 *datac = (float2 *)(ws + al(bytes_buf) + al(bytes_spec));
        float2 *g = (float2 *)((char *)datac + al(bytes_datac));
        float2 *f = (float2 *)((char *)g + al(bytes_g));
        char *tab = (char *)f + al(bytes_f);
        float2 *w_dev = (float2 *)tab;
        float *ct = (float *)(tab + al((size_t)ne * sizeof(float2)));
        float *sn = ct + nproj, *sth = sn + nproj;
        int *order = (int *)(sth + nproj);

        // host-side tables: cos / sin per angle, angles sorted ascending (stable) with their original indices
        std::vector<float> h_ct(nproj), h_sn(nproj), h_sth(nproj);
        std::vector<int> h_order(nproj);
        for (int i = 0; i < nproj; ++i) { h_ct[i] = cosf(theta_host[i]); h_sn[i] = sinf(theta_host[i]); h_order[i] = i; }
        std::stable_sort(h_order.begin(), h_order.end(), [&](int p, int q) { return theta_host[p] < theta_host[q]; });
        for (int i = 0; i < nproj; ++i) h_sth[i] = theta_host[h_order[i]];
        // Hermitian extension of the half-spectrum filter table: W[ne - k] = conj(W[k])
        std::vector<float2> h_w(ne);
        for (int k = 0; k < nh; ++k) h_w[k] = make_float2(w_host[2 * k], w_host[2 * k + 1]);
        for (int k = nh; k < ne; ++k) h_w[k] = make_float2(w_host[2 * (ne - k)], -w_host[2 * (ne - k) + 1]);
        h_w[0].y = 0.0f;       // a real inverse transform ignores the imaginary part of the DC and Nyquist bins
        h_w[ne / 2].y = 0.0f;  // (irfft, methodsDIR_CuPy.py:533): keep the two slices of a pair separate there too
        TOMO_HIPG(hipMemcpyAsync(w_dev, h_w.data(), (size_t)ne * sizeof(float2), hipMemcpyHostToDevice, st));
        TOMO_HIPG(hipMemcpyAsync(ct, h_ct.data(), nproj * sizeof(float), hipMemcpyHostToDevice, st));
        TOMO_HIPG(hipMemcpyAsync(sn, h_sn.data(), nproj * sizeof(float), hipMemcpyHostToDevice, st));
        TOMO_HIPG(hipMemcpyAsync(sth, h_sth.data(), nproj * sizeof(float), hipMemcpyHostToDevice, st));
        TOMO_HIPG(hipMemcpyAsync(order, h_order.data(), nproj * sizeof(int), hipMemcpyHostToDevice, st));
        TOMO_HIPG(hipStreamSynchronize(st));  // host vectors go out of scope at the end of this block only; be explicit

        const int pad_m = ne / 2 - raw_n / 2, crop_m = ne / 2 - n / 2;
        const float phi_scale = (float)(1 - n % 4) / (float)nproj;
        for (int c0 = 0; c0 < nzh; c0 += FZ) {
            const int zc = std::min(FZ, nzh - c0);
            const size_t rows = (size_t)zc * nproj;  // pair-rows of this chunk
            // ---- STEP 0: filter (methodsDIR_CuPy.py:449-545), slice pairs as complex rows, output shifted into datac[z][p][x]
            const float *chunk_in = data_dev + (size_t)2 * c0 * nproj * raw_n;
            for (size_t r0 = 0; r0 < rows; r0 += rows_sub) {
                const size_t rs = std::min(rows_sub, rows - r0);
                if ((rc = get_plan(device, PLAN_C2C_1D, ne, (int)rs, st, &p_filt)) != TOMO_OK) goto done;
                pad_pair_kernel<<<2048, 256, 0, st>>>(chunk_in, cbuf, rs, r0, nproj, raw_n, ne, pad_m);
                TOMO_FFT(hipfftExecC2C(p_filt, (hipfftComplex *)cbuf, (hipfftComplex *)cbuf, HIPFFT_FORWARD));
                mul_filter_kernel<<<2048, 256, 0, st>>>(cbuf, w_dev, rs, ne);
                TOMO_FFT(hipfftExecC2C(p_filt, (hipfftComplex *)cbuf, (hipfftComplex *)cbuf, HIPFFT_BACKWARD));
                crop_pair_kernel<<<2048, 256, 0, st>>>(cbuf, datac, rs, r0, n, ne, crop_m, 1.0f / (float)ne);
            }
            // ---- STEP 1: 1D FFT along the detector (methodsDIR_CuPy.py:723-724)
            if ((rc = get_plan(device, PLAN_C2C_1D, n, zc * nproj, st, &p_c2c)) != TOMO_OK) goto done;
            if ((rc = get_plan(device, PLAN_C2C_2D, two_n, zc, st, &p_2d)) != TOMO_OK) goto done;
            TOMO_FFT(hipfftExecC2C(p_c2c, (hipfftComplex *)datac, (hipfftComplex *)datac, HIPFFT_FORWARD));
            {
                dim3 grid(ceil_div(n, 64), nproj);
                transpose_scale_kernel<<<grid, 256, 0, st>>>(datac, g, zc, nproj, n, 4.0f / (float)n);
            }
            // ---- STEP 2: gathering on the 2n x 2n frequency grid (methodsDIR_CuPy.py:760-836)
            {
                GatherArgs ga;
                ga.g = g; ga.f = f; ga.ct = ct; ga.st = sn; ga.sth = sth; ga.order = order;
                ga.nproj = nproj; ga.n = n; ga.m = m; ga.mu = mu;
                ga.center_size = center_size;
                ga.use_center = center_size >= 192 ? 1 : 0;  // _CENTER_SIZE_MIN, methodsDIR_CuPy.py:23
                dim3 grid(ceil_div(two_n, 4 * GP), two_n);
                gather_kernel<<<grid, 256, 0, st>>>(ga, zc);
            }
            // ---- STEP 3: 2D inverse FFT (methodsDIR_CuPy.py:851-897); the shifts live in the neighbouring kernels
            TOMO_FFT(hipfftExecC2C(p_2d, (hipfftComplex *)f, (hipfftComplex *)f, HIPFFT_BACKWARD));
            // ---- STEP 4: unpadding x phi (methodsDIR_CuPy.py:920-967)
            {
                dim3 grid(ceil_div(out_size, 256), out_size, zc);
                unpad_kernel<<<grid, 256, 0, st>>>(f, out_dev, n, unpad_m, out_size, 2 * c0, out_z, mu, phi_scale,
                                                   1.0f / ((float)two_n * (float)two_n));
            }
            TOMO_HIPG(hipGetLastError());
        }
        TOMO_HIPG(hipStreamSynchronize(st));  // the arena may be handed to another operator as soon as we return
    }
done:
    if (rc != TOMO_OK) (void)hipStreamSynchronize(st);
    return rc;
}
