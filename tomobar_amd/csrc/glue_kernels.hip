// Element-wise glue, reductions and pre/post kernels of the FISTA / ADMM loops (HBM-streaming, float4).
// Compiled with -ffp-contract=off: every rounding below is the one the reference's separate CuPy
// ufunc launches produce (methodsIR_CuPy.py:463-475,545-566); fmaf appears only where stated.
#include "tomo_common.h"

#include <algorithm>

namespace {

constexpr int EW_BLOCK = 256;
constexpr int EW_MAX_GRID = 256 * 8;  // 256 CUs x 8 blocks, grid-stride beyond

inline int ew_grid(size_t n4)
{
    size_t g = (n4 + EW_BLOCK - 1) / EW_BLOCK;
    if (g < 1) g = 1;
    if (g > (size_t)EW_MAX_GRID) g = EW_MAX_GRID;
    return (int)g;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- generic streaming kernels: up to 3 inputs, up to 2 outputs, functor works on scalars
template <int NIN, int NOUT, typename F>
__global__ __launch_bounds__(EW_BLOCK) void ew_vec4(const float *a, const float *b,
                                                    const float *c, float *o0, float *o1,
                                                    size_t n, F f)
{
    const size_t n4 = n >> 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 va = reinterpret_cast<const float4 *>(a)[i];
        float4 vb = NIN > 1 ? reinterpret_cast<const float4 *>(b)[i] : va;
        float4 vc = NIN > 2 ? reinterpret_cast<const float4 *>(c)[i] : va;
        float4 r0, r1;
        f(va.x, vb.x, vc.x, r0.x, r1.x);
        f(va.y, vb.y, vc.y, r0.y, r1.y);
        f(va.z, vb.z, vc.z, r0.z, r1.z);
        f(va.w, vb.w, vc.w, r0.w, r1.w);
        reinterpret_cast<float4 *>(o0)[i] = r0;
        if (NOUT > 1) reinterpret_cast<float4 *>(o1)[i] = r1;
    }
    // tail (n % 4) by the first few threads of block 0
    const size_t tail0 = n4 << 2;
    if (blockIdx.x == 0 && tail0 + threadIdx.x < n) {
        size_t i = tail0 + threadIdx.x;
        float r0, r1;
        f(a[i], NIN > 1 ? b[i] : 0.0f, NIN > 2 ? c[i] : 0.0f, r0, r1);
        o0[i] = r0;
        if (NOUT > 1) o1[i] = r1;
    }
}

template <int NIN, int NOUT, typename F>
__global__ __launch_bounds__(EW_BLOCK) void ew_scalar(const float *a, const float *b, const float *c, float *o0,
                                                      float *o1, size_t n, F f)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float r0, r1;
        f(a[i], NIN > 1 ? b[i] : 0.0f, NIN > 2 ? c[i] : 0.0f, r0, r1);
        o0[i] = r0;
        if (NOUT > 1) o1[i] = r1;
    }
}

template <int NIN, int NOUT, typename F>
int ew_launch(const float *a, const float *b, const float *c, float *o0, float *o1, size_t n, void *stream, F f)
{
    if (n == 0) return TOMO_OK;
    bool vec = aligned16(a) && aligned16(o0) && (NIN < 2 || aligned16(b)) && (NIN < 3 || aligned16(c)) &&
               (NOUT < 2 || aligned16(o1));
    if (vec)
        ew_vec4<NIN, NOUT, F><<<ew_grid(n >> 2), EW_BLOCK, 0, as_stream(stream)>>>(a, b, c, o0, o1, n, f);
    else
        ew_scalar<NIN, NOUT, F><<<ew_grid(n), EW_BLOCK, 0, as_stream(stream)>>>(a, b, c, o0, o1, n, f);
    TOMO_LAUNCH_CHECK();
    return TOMO_OK;
}

struct MomentumF {  // X_t = X + beta*(X - X_old): mul then add, no fma (two CuPy ufuncs)
    float beta;
    __device__ void operator()(float x, float xo, float, float &r0, float &) const { r0 = x + beta * (x - xo); }
};
struct AdmmDualF {  // u = u + (z - x)
    __device__ void operator()(float u, float z, float x, float &r0, float &) const { r0 = u + (z - x); }
};
struct AxpbyF {
    float a, b;
    __device__ void operator()(float x, float y, float, float &r0, float &) const { r0 = a * x + b * y; }
};
struct ScaleF {
    float a;
    __device__ void operator()(float x, float, float, float &r0, float &) const { r0 = a * x; }
};
struct ClampF {
    float lo;
    __device__ void operator()(float x, float, float, float &r0, float &) const { r0 = x < lo ? lo : x; }
};
struct MulF {
    __device__ void operator()(float x, float y, float, float &r0, float &) const { r0 = x * y; }
};
struct RecipSafeF {  // 1/x with nan, +inf, -inf -> 1   (cp.nan_to_num(..., nan=1, posinf=1, neginf=1))
    __device__ void operator()(float x, float, float, float &r0, float &) const
    {
        float r = 1.0f / x;
        r0 = (isnan(r) || isinf(r)) ? 1.0f : r;
    }
};
struct FillF {
    float v;
    __device__ void operator()(float, float, float, float &r0, float &) const { r0 = v; }
};
struct PwlsF {  // w = max(b, 1e-6) / wmax
    float wmax;
    __device__ void operator()(float b, float, float, float &r0, float &) const
    {
        float w = b < 1e-6f ? 1e-6f : b;
        r0 = w / wmax;
    }
};

// ---- reductions: per-block partials in double, finished on the host (the value is returned to the host anyway)
enum { RED_SUMSQ = 0, RED_DOT = 1, RED_MAX = 2, RED_MAX_CLAMPED = 3 };

template <int MODE>
__global__ __launch_bounds__(EW_BLOCK) void reduce_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                          size_t n, double *partial)
{
    double acc = (MODE >= RED_MAX) ? -INFINITY : 0.0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = x[i];
        if (MODE == RED_SUMSQ) acc += (double)v * (double)v;
        else if (MODE == RED_DOT) acc += (double)v * (double)y[i];
        else if (MODE == RED_MAX) acc = fmax(acc, (double)v);
        else acc = fmax(acc, (double)(v < 1e-6f ? 1e-6f : v));
    }
    for (int off = 32; off > 0; off >>= 1) {
        double o = __shfl_down(acc, off, 64);
        acc = (MODE >= RED_MAX) ? fmax(acc, o) : acc + o;
    }
    __shared__ double wave_part[EW_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = wave_part[0];
        for (int w = 1; w < EW_BLOCK / 64; ++w) r = (MODE >= RED_MAX) ? fmax(r, wave_part[w]) : r + wave_part[w];
        partial[blockIdx.x] = r;
    }
}

template <int MODE>
int reduce_host(const float *x, const float *y, size_t n, double *out, void *stream)
{
    TOMO_REQUIRE(out != nullptr, "out is NULL");
    if (n == 0) { *out = (MODE >= RED_MAX) ? -INFINITY : 0.0; return TOMO_OK; }
    int dev = 0;
    TOMO_HIP(hipGetDevice(&dev));
    const int grid = ew_grid(n);
    void *buf = nullptr;
    int rc = tomo_arena_get(dev, as_stream(stream), ARENA_REDUCE, (size_t)EW_MAX_GRID * sizeof(double), &buf);
    if (rc != TOMO_OK) return rc;
    reduce_kernel<MODE><<<grid, EW_BLOCK, 0, as_stream(stream)>>>(x, y, n, (double *)buf);
    TOMO_LAUNCH_CHECK();
    std::vector<double> host(grid);
    TOMO_HIP(hipMemcpyAsync(host.data(), buf, grid * sizeof(double), hipMemcpyDeviceToHost, as_stream(stream)));
    TOMO_HIP(hipStreamSynchronize(as_stream(stream)));
    double r = host[0];
    for (int i = 1; i < grid; ++i) r = (MODE >= RED_MAX) ? (host[i] > r ? host[i] : r) : r + host[i];
    *out = r;
    return TOMO_OK;
}

// ---- ring-artefact data terms: reductions over the angles of one (ordered-subset) residual, one thread per detector
//      pixel (z, u), angles summed in ascending order (deterministic; the oracle sums in the same order)
// Group-Huber: vec = sum_a res[z,a,u];  r = r_x - l_inv * vec;  then the PWLS weights are applied to res in place
__global__ __launch_bounds__(256) void ring_gh_reduce_kernel(float *res, const float *w_full, const int *src, int nz, int na_s,
                                                             int na_full, int nu, const float *r_x, float l_inv, float *r_out)
{
    const size_t total = (size_t)nz * nu;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t z = i / nu, u = i % nu;
        float vec = 0.0f;
        float *col = res + (z * na_s) * nu + u;
        for (int a = 0; a < na_s; ++a) {
            const float v = col[(size_t)a * nu];
            vec = vec + v;
            if (w_full) col[(size_t)a * nu] = v * w_full[(z * na_full + src[a]) * nu + u];
        }
        r_out[i] = r_x[i] - l_inv * vec;
    }
}

// Stripe-weighted least squares: res_a <- w_a res_a - w_a (sum_a w_a res_a) / (sum_a w_a + beta)
__global__ __launch_bounds__(256) void swls_kernel(float *res, const float *w_full, const int *src, int nz, int na_s, int na_full,
                                                   int nu, float beta)
{
    const size_t total = (size_t)nz * nu;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t z = i / nu, u = i % nu;
        float *col = res + (z * na_s) * nu + u;
        float wr = 0.0f, ws = 0.0f;
        for (int a = 0; a < na_s; ++a) {
            const float wa = w_full[(z * na_full + src[a]) * nu + u];
            wr = wr + wa * col[(size_t)a * nu];
            ws = ws + wa;
        }
        const float q = wr / (ws + beta);
        for (int a = 0; a < na_s; ++a) {
            const float wa = w_full[(z * na_full + src[a]) * nu + u];
            col[(size_t)a * nu] = wa * col[(size_t)a * nu] - wa * q;
        }
    }
}

// soft threshold + momentum of the offsets:  r <- sign(r) max(|r| - lambda, 0);  r_x = r + beta (r - r_old);  r_old <- r
__global__ __launch_bounds__(256) void ring_gh_update_kernel(float *r, float *r_old, float *r_x, float lambda, float beta, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = r[i];
        const float m = fmaxf(fabsf(v) - lambda, 0.0f);
        const float t = v > 0.0f ? m : (v < 0.0f ? -m : 0.0f);
        r[i] = t;
        r_x[i] = t + beta * (t - r_old[i]);
        r_old[i] = t;
    }
}

// ---- vertical centre-of-rotation component (CenterRotOffset[:, 1], supp/funcs.py:52-55): the detector of angle a sits
//      shift[a] rows higher, i.e. detector row r looks at slice r + shift[a].  Parallel rays stay inside their slice, so the
//      3D operator is the per-slice operator composed with a per-angle resampling of the detector rows (2-tap linear, zero
//      outside): forward projection = resample(+shift) after A, back projection = A^T after resample(-shift).
__global__ __launch_bounds__(256) void shift_rows_kernel(const float *__restrict__ in, float *__restrict__ out, int nz, int na,
                                                         int nu, const float *__restrict__ shift, float sign)
{
    const size_t total = (size_t)nz * na * nu;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int u = (int)(i % nu);
        const int a = (int)((i / nu) % na);
        const int r = (int)(i / ((size_t)nu * na));
        const float sh = sign * shift[a];
        const float fl = floorf(sh);
        const float w = sh - fl;  // the fractional part of the shift alone: the same for every row, whatever row a z-slab starts at
        const int r0 = (int)fminf(fmaxf((float)r + fl, -2.0f), (float)nz);  // clamped: both taps are outside the detector there
        const float s0 = (r0 >= 0 && r0 < nz) ? in[((size_t)r0 * na + a) * nu + u] : 0.0f;
        const float s1 = (r0 + 1 >= 0 && r0 + 1 < nz) ? in[((size_t)(r0 + 1) * na + a) * nu + u] : 0.0f;
        out[i] = (1.0f - w) * s0 + w * s1;
    }
}

// residual of data_fidelities.py:28-39 on an already forward-projected subset: res = w (.) (ax - b), 1 - b / max(ax, 1e-8)
// (KL) or b / max(ax, 1e-8) (OSEM ratio); b / w are full sinograms addressed through the subset's angle indices
__global__ __launch_bounds__(256) void sino_residual_kernel(const float *__restrict__ ax, const float *__restrict__ b,
                                                            const float *__restrict__ w, const int *__restrict__ src, int nz,
                                                            int na_s, int na_full, int nu, int fidelity, int gathered, float *__restrict__ out)
{
    const size_t total = (size_t)nz * na_s * nu;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int u = (int)(i % nu);
        const int a = (int)((i / nu) % na_s);
        const size_t z = i / ((size_t)nu * na_s);
        const size_t fi = (z * na_full + src[a]) * nu + u;
        float val = ax[i];
        const float bv = b[(gathered & 1) ? i : fi];
        if (fidelity == TOMO_FID_KL || fidelity == TOMO_FID_RATIO) {
            const float v = val < 1e-8f ? 1e-8f : val;
            const float q = bv / v;
            val = (fidelity == TOMO_FID_KL) ? 1.0f - q : q;
        } else {
            val = val - bv;
            if (w) val = val * w[(gathered & 2) ? i : fi];
        }
        out[i] = val;
    }
}

// ---- pre/post
__global__ void pad_edge_kernel(const float *__restrict__ in, float *__restrict__ out, int rows, int nu0, int pad)
{
    const int nu = nu0 + 2 * pad;
    const size_t total = (size_t)rows * nu;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        size_t r = i / nu;
        int u = (int)(i - r * nu) - pad;
        u = u < 0 ? 0 : (u >= nu0 ? nu0 - 1 : u);
        out[i] = in[r * nu0 + u];
    }
}

__global__ void crop_kernel(const float *__restrict__ in, float *__restrict__ out, int nz, int n, int m, int a0)
{
    const size_t total = (size_t)nz * m * m;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        size_t z = i / ((size_t)m * m);
        size_t rem = i - z * m * m;
        int y = (int)(rem / m), x = (int)(rem - (size_t)y * m);
        out[i] = in[(z * n + (y + a0)) * n + (x + a0)];
    }
}

__global__ void mask_kernel(float *vol, int nz, int n, double limit)
{
    // dist computed as sqrt of an exact integer in double, as numpy does (suppTools.py:381-385)
    const size_t total = (size_t)nz * n * n;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const int h = n / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        size_t rem = i % ((size_t)n * n);
        int y = (int)(rem / n), x = (int)(rem - (size_t)y * n);
        double d = sqrt((double)((x - h) * (x - h) + (y - h) * (y - h)));
        if (!(d <= limit)) vol[i] = vol[i] * 0.0f;  // data *= mask (keeps the sign of zero / NaN like the reference)
    }
}

__global__ void permute3_kernel(const float *__restrict__ in, float *__restrict__ out, int d0, int d1, int d2,
                                long long s0, long long s1, long long s2)
{
    const size_t total = (size_t)d0 * d1 * d2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        size_t a = i / ((size_t)d1 * d2);
        size_t rem = i - a * d1 * d2;
        size_t b = rem / d2, c = rem - b * d2;
        out[i] = in[(long long)a * s0 + (long long)b * s1 + (long long)c * s2];
    }
}


// ---- halo staging for the z-slab exchange: up to 8 plane blocks (each contiguous in its own array) <-> ONE contiguous
//      staging buffer, so a neighbour exchange is one message each way instead of one per array (RCCL point-to-point pays
//      a fixed cost per op).  Blocks are copied as dwordx4 where source, destination and length allow it.
constexpr int HALO_MAX_BLOCKS = 8;
struct HaloBlocks {
    const char *src[HALO_MAX_BLOCKS];
    char *dst[HALO_MAX_BLOCKS];
    size_t bytes[HALO_MAX_BLOCKS];
    int n;
};

__global__ __launch_bounds__(256) void halo_copy_kernel(HaloBlocks h)
{
    const int b = blockIdx.y;
    const char *src = h.src[b];
    char *dst = h.dst[b];
    const size_t bytes = h.bytes[b];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool vec = (((uintptr_t)src | (uintptr_t)dst) & 15u) == 0;
    size_t done = 0;
    if (vec) {
        const size_t n16 = bytes >> 4;
        for (size_t i = t0; i < n16; i += stride) reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
        done = n16 << 4;
    } else if ((((uintptr_t)src | (uintptr_t)dst) & 3u) == 0) {
        const size_t n4 = bytes >> 2;
        for (size_t i = t0; i < n4; i += stride) reinterpret_cast<unsigned *>(dst)[i] = reinterpret_cast<const unsigned *>(src)[i];
        done = n4 << 2;
    }
    for (size_t i = done + t0; i < bytes; i += stride) dst[i] = src[i];
}

int halo_copy(const void *const *a, const void *b_contig, const size_t *bytes, int nblocks, bool pack, void *stream)
{
    TOMO_REQUIRE(nblocks >= 0 && nblocks <= HALO_MAX_BLOCKS, "a halo exchange packs at most %d blocks (got %d)", HALO_MAX_BLOCKS, nblocks);
    if (nblocks == 0) return TOMO_OK;
    TOMO_REQUIRE(a != nullptr && b_contig != nullptr && bytes != nullptr, "NULL halo block table");
    HaloBlocks h;
    h.n = nblocks;
    size_t off = 0, largest = 0;
    for (int i = 0; i < nblocks; ++i) {
        TOMO_REQUIRE(a[i] != nullptr || bytes[i] == 0, "halo block %d is NULL", i);
        char *stage = (char *)b_contig + off;
        h.src[i] = pack ? (const char *)a[i] : stage;
        h.dst[i] = pack ? stage : (char *)a[i];
        h.bytes[i] = bytes[i];
        off += (bytes[i] + 15) & ~(size_t)15;  // every block starts 16-byte aligned in the staging buffer
        largest = std::max(largest, bytes[i]);
    }
    for (int i = nblocks; i < HALO_MAX_BLOCKS; ++i) { h.src[i] = nullptr; h.dst[i] = nullptr; h.bytes[i] = 0; }
    const int gx = (int)std::min<size_t>(std::max<size_t>((largest / 16 + 255) / 256, 1), 512);
    halo_copy_kernel<<<dim3(gx, nblocks), 256, 0, as_stream(stream)>>>(h);
    TOMO_LAUNCH_CHECK();
    return TOMO_OK;
}
}  // namespace

extern "C" int tomo_momentum(const float *x, const float *xold, float *xt, float beta, size_t count, void *stream)
{
    return ew_launch<2, 1>(x, xold, nullptr, xt, nullptr, count, stream, MomentumF{beta});
}
extern "C" int tomo_admm_dual(float *u, const float *z, const float *x, size_t count, void *stream)
{
    return ew_launch<3, 1>(u, z, x, u, nullptr, count, stream, AdmmDualF{});
}
extern "C" int tomo_axpby(float a, const float *x, float b, float *y, size_t count, void *stream)
{
    return ew_launch<2, 1>(x, y, nullptr, y, nullptr, count, stream, AxpbyF{a, b});
}
extern "C" int tomo_scale(float a, const float *x, float *y, size_t count, void *stream)
{
    return ew_launch<1, 1>(x, nullptr, nullptr, y, nullptr, count, stream, ScaleF{a});
}
extern "C" int tomo_clamp_min(float *x, float lo, size_t count, void *stream)
{
    return ew_launch<1, 1>(x, nullptr, nullptr, x, nullptr, count, stream, ClampF{lo});
}
extern "C" int tomo_mul(const float *x, float *y, size_t count, void *stream)
{
    return ew_launch<2, 1>(x, y, nullptr, y, nullptr, count, stream, MulF{});
}
extern "C" int tomo_recip_safe(const float *x, float *y, size_t count, void *stream)
{
    return ew_launch<1, 1>(x, nullptr, nullptr, y, nullptr, count, stream, RecipSafeF{});
}
extern "C" int tomo_fill(float *x, float value, size_t count, void *stream)
{
    return ew_launch<1, 1>(x, nullptr, nullptr, x, nullptr, count, stream, FillF{value});
}

extern "C" int tomo_norm2(const float *x, size_t count, double *out_host, void *stream)
{
    double s = 0.0;
    int rc = reduce_host<RED_SUMSQ>(x, nullptr, count, &s, stream);
    if (rc == TOMO_OK) *out_host = sqrt(s);
    return rc;
}
extern "C" int tomo_dot(const float *x, const float *y, size_t count, double *out_host, void *stream)
{
    return reduce_host<RED_DOT>(x, y, count, out_host, stream);
}
extern "C" int tomo_max(const float *x, size_t count, float *out_host, void *stream)
{
    double m = 0.0;
    int rc = reduce_host<RED_MAX>(x, nullptr, count, &m, stream);
    if (rc == TOMO_OK) *out_host = (float)m;
    return rc;
}
extern "C" int tomo_pwls_weights(const float *b, float *w, size_t count, void *stream)
{
    double m = 0.0;
    int rc = reduce_host<RED_MAX_CLAMPED>(b, nullptr, count, &m, stream);
    if (rc != TOMO_OK) return rc;
    return ew_launch<1, 1>(b, nullptr, nullptr, w, nullptr, count, stream, PwlsF{(float)m});
}

// slab form: the maximum is taken over all slabs by the caller (max-all-reduce) between these two calls
extern "C" int tomo_pwls_max(const float *b, size_t count, float *out_host, void *stream)
{
    double m = 0.0;
    int rc = reduce_host<RED_MAX_CLAMPED>(b, nullptr, count, &m, stream);
    if (rc == TOMO_OK) *out_host = (float)m;
    return rc;
}

// res[z, a, u] += scale * ring[z, u]: the Group-Huber offsets added to an already formed LS residual (the vertical-CoR path,
// where the row resampling sits between the projector and the residual; one rounding for the product, one for the sum, as the
// fused epilogue of tomo_fp3d_residual_ring)
__global__ __launch_bounds__(256) void sino_add_ring_kernel(float *__restrict__ res, const float *__restrict__ ring, float scale,
                                                            int nz, int na_s, int nu)
{
    const size_t total = (size_t)nz * na_s * nu;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int u = (int)(i % nu);
        const int z = (int)(i / ((size_t)nu * na_s));
        const float t = scale * ring[(size_t)z * nu + u];
        res[i] = res[i] + t;
    }
}

extern "C" int tomo_pwls_weights_scaled(const float *b, float *w, size_t count, float wmax, void *stream)
{
    return ew_launch<1, 1>(b, nullptr, nullptr, w, nullptr, count, stream, PwlsF{wmax});
}

extern "C" int tomo_shift_rows(const float *in_dev, float *out_dev, int nz, int na, int nu, const float *shift_dev, float sign,
                               void *stream)
{
    TOMO_REQUIRE(in_dev && out_dev && shift_dev && in_dev != out_dev && nz > 0 && na >= 0 && nu > 0, "bad row-shift arguments");
    const size_t total = (size_t)nz * na * nu;
    if (total == 0) return TOMO_OK;
    shift_rows_kernel<<<(unsigned)std::min<size_t>((total + 255) / 256, 8192), 256, 0, as_stream(stream)>>>(
        in_dev, out_dev, nz, na, nu, shift_dev, sign);
    TOMO_LAUNCH_CHECK();
    return TOMO_OK;
}

extern "C" int tomo_sino_residual(const float *ax_dev, const float *b_full_dev, const float *w_full_dev, const int *src_dev,
                                  int nz, int na_s, int na_full, int nu, int gathered, int fidelity, float *res_dev,
                                  void *stream)
{
    TOMO_REQUIRE(ax_dev && b_full_dev && src_dev && res_dev && nz > 0 && na_s >= 0 && nu > 0, "bad residual arguments");
    const size_t total = (size_t)nz * na_s * nu;
    if (total == 0) return TOMO_OK;
    sino_residual_kernel<<<(unsigned)std::min<size_t>((total + 255) / 256, 8192), 256, 0, as_stream(stream)>>>(
        ax_dev, b_full_dev, fidelity == TOMO_FID_PWLS ? w_full_dev : nullptr, src_dev, nz, na_s, na_full, nu, fidelity, gathered, res_dev);
    TOMO_LAUNCH_CHECK();
    return TOMO_OK;
}

extern "C" int tomo_sino_add_ring(float *res_dev, const float *ring_dev, float ring_scale, int nz, int na_s, int nu, void *stream)
{
    TOMO_REQUIRE(res_dev && ring_dev && nz > 0 && na_s >= 0 && nu > 0, "bad ring-term arguments");
    const size_t total = (size_t)nz * na_s * nu;
    if (total == 0) return TOMO_OK;
    sino_add_ring_kernel<<<(unsigned)std::min<size_t>((total + 255) / 256, 8192), 256, 0, as_stream(stream)>>>(
        res_dev, ring_dev, ring_scale, nz, na_s, nu);
    TOMO_LAUNCH_CHECK();
    return TOMO_OK;
}

// Huber / Student's-t re-weighting of a residual that was NOT formed by tomo_fp3d_residual_robust's epilogue (the ring-term and
// vertical-CoR paths); element-wise, so either residual layout works (the padding of a quad-interleaved one is 0 -> 0)
__global__ __launch_bounds__(256) void sino_robust_kernel(float *__restrict__ res, size_t count, int mode, float delta)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        float r = res[i];
        if (mode == TOMO_ROBUST_HUBER) {
            const float ar = fabsf(r);
            if (ar > delta) r = (delta / ar) * r;
        } else {
            r = (2.0f / (delta * delta + r * r)) * r;
        }
        res[i] = r;
    }
}

extern "C" int tomo_sino_robust(float *res_dev, size_t count, int robust, float delta, void *stream)
{
    TOMO_REQUIRE(res_dev != nullptr || count == 0, "NULL residual pointer");
    TOMO_REQUIRE(robust == TOMO_ROBUST_HUBER || robust == TOMO_ROBUST_STUDENTST, "unknown robust mode %d", robust);
    TOMO_REQUIRE(delta > 0.0f, "the Huber / Student's-t threshold must be positive");
    if (count == 0) return TOMO_OK;
    sino_robust_kernel<<<(unsigned)std::min<size_t>((count + 255) / 256, 8192), 256, 0, as_stream(stream)>>>(res_dev, count, robust, delta);
    TOMO_LAUNCH_CHECK();
    return TOMO_OK;
}

extern "C" int tomo_ring_gh_reduce(float *res_dev, const float *w_full_dev, const int *src_dev, int nz, int na_s, int na_full,
                                   int nu, const float *rx_dev, float l_inv, float *r_out_dev, void *stream)
{
    TOMO_REQUIRE(res_dev && rx_dev && r_out_dev && nz > 0 && na_s >= 0 && nu > 0, "bad ring-term arguments");
    TOMO_REQUIRE(w_full_dev == nullptr || src_dev != nullptr, "weights need the subset's angle index table");
    const size_t total = (size_t)nz * nu;
    ring_gh_reduce_kernel<<<(unsigned)std::min<size_t>((total + 255) / 256, 4096), 256, 0, as_stream(stream)>>>(
        res_dev, w_full_dev, src_dev, nz, na_s, na_full, nu, rx_dev, l_inv, r_out_dev);
    TOMO_LAUNCH_CHECK();
    return TOMO_OK;
}

extern "C" int tomo_swls_apply(float *res_dev, const float *w_full_dev, const int *src_dev, int nz, int na_s, int na_full, int nu,
                               float beta, void *stream)
{
    TOMO_REQUIRE(res_dev && w_full_dev && src_dev && nz > 0 && na_s >= 0 && nu > 0, "bad SWLS arguments");
    const size_t total = (size_t)nz * nu;
    swls_kernel<<<(unsigned)std::min<size_t>((total + 255) / 256, 4096), 256, 0, as_stream(stream)>>>(
        res_dev, w_full_dev, src_dev, nz, na_s, na_full, nu, beta);
    TOMO_LAUNCH_CHECK();
    return TOMO_OK;
}

extern "C" int tomo_ring_gh_update(float *r_dev, float *r_old_dev, float *rx_dev, float lambda, float beta, size_t count,
                                   void *stream)
{
    TOMO_REQUIRE(r_dev && r_old_dev && rx_dev, "NULL ring-term pointer");
    if (count == 0) return TOMO_OK;
    ring_gh_update_kernel<<<(unsigned)std::min<size_t>((count + 255) / 256, 4096), 256, 0, as_stream(stream)>>>(
        r_dev, r_old_dev, rx_dev, lambda, beta, count);
    TOMO_LAUNCH_CHECK();
    return TOMO_OK;
}

extern "C" int tomo_pad_edge(const float *in, float *out, int rows, int nu0, int pad, void *stream)
{
    TOMO_REQUIRE(rows >= 0 && nu0 > 0 && pad >= 0, "bad padding arguments");
    size_t total = (size_t)rows * (nu0 + 2 * pad);
    if (total == 0) return TOMO_OK;
    pad_edge_kernel<<<ew_grid(total), EW_BLOCK, 0, as_stream(stream)>>>(in, out, rows, nu0, pad);
    TOMO_LAUNCH_CHECK();
    return TOMO_OK;
}

extern "C" int tomo_crop_center(const float *in, float *out, int nz, int n, int m, void *stream)
{
    TOMO_REQUIRE(nz > 0 && n > 0 && m > 0 && m <= n, "bad crop arguments");
    const int a0 = (n - m) / 2;
    crop_kernel<<<ew_grid((size_t)nz * m * m), EW_BLOCK, 0, as_stream(stream)>>>(in, out, nz, n, m, a0);
    TOMO_LAUNCH_CHECK();
    return TOMO_OK;
}

extern "C" int tomo_circ_mask(float *vol, int nz, int n, double radius, void *stream)
{
    TOMO_REQUIRE(nz > 0 && n > 0, "bad mask arguments");
    // suppTools.py:387-394, evaluated in double like the reference's Python floats
    const double h = (double)(n / 2), r = radius;
    const double limit = (r <= 1.0) ? h - fabs(h - h / r) : h + fabs(h - h / r);
    mask_kernel<<<ew_grid((size_t)nz * n * n), EW_BLOCK, 0, as_stream(stream)>>>(vol, nz, n, limit);
    TOMO_LAUNCH_CHECK();
    return TOMO_OK;
}

extern "C" int tomo_permute3(const float *in, float *out, int d0, int d1, int d2, int64_t s0, int64_t s1,
                             int64_t s2, void *stream)
{
    TOMO_REQUIRE(d0 > 0 && d1 > 0 && d2 > 0, "bad permute dims");
    permute3_kernel<<<ew_grid((size_t)d0 * d1 * d2), EW_BLOCK, 0, as_stream(stream)>>>(in, out, d0, d1, d2, s0, s1, s2);
    TOMO_LAUNCH_CHECK();
    return TOMO_OK;
}

// ---- diagnostics: N-stream streaming kernel (calibrates the achievable HBM rate for a kernel's read/write mix)
namespace {
struct StreamArgs { const float *in[8]; float *out[8]; int nin, nout; size_t n; };
template <int VEC>
__global__ __launch_bounds__(256) void diag_stream_kernel(StreamArgs a)
{
    const size_t nv = a.n / VEC;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.0f;
        for (int k = 0; k < a.nin; ++k) {
            if (VEC == 4) {
                float4 t = reinterpret_cast<const float4 *>(a.in[k])[i];
                acc[0] += t.x; acc[1 % VEC] += t.y; acc[2 % VEC] += t.z; acc[3 % VEC] += t.w;
            } else {
                acc[0] += a.in[k][i];
            }
        }
        for (int k = 0; k < a.nout; ++k) {
            if (VEC == 4) reinterpret_cast<float4 *>(a.out[k])[i] = make_float4(acc[0] + k, acc[1 % VEC], acc[2 % VEC], acc[3 % VEC]);
            else a.out[k][i] = acc[0] + k;
        }
    }
}
}  // namespace

extern "C" int tomo_diag_stream(const float *const *in_dev, int nin, float *const *out_dev, int nout, size_t count,
                                int vec, int grid, void *stream)
{
    TOMO_REQUIRE(nin >= 0 && nin <= 8 && nout >= 1 && nout <= 8, "at most 8 input and 8 output streams");
    StreamArgs a;
    for (int k = 0; k < 8; ++k) { a.in[k] = k < nin ? in_dev[k] : nullptr; a.out[k] = k < nout ? out_dev[k] : nullptr; }
    a.nin = nin; a.nout = nout; a.n = count;
    if (grid <= 0) grid = EW_MAX_GRID;
    if (vec == 4) diag_stream_kernel<4><<<grid, 256, 0, as_stream(stream)>>>(a);
    else diag_stream_kernel<1><<<grid, 256, 0, as_stream(stream)>>>(a);
    TOMO_LAUNCH_CHECK();
    return TOMO_OK;
}

extern "C" size_t tomo_halo_staging_bytes(const size_t *bytes, int nblocks)
{
    size_t off = 0;
    for (int i = 0; bytes && i < nblocks; ++i) off += (bytes[i] + 15) & ~(size_t)15;
    return off;
}

extern "C" int tomo_halo_pack(const void *const *src_dev, const size_t *bytes, int nblocks, void *staging_dev, void *stream)
{
    return halo_copy(src_dev, staging_dev, bytes, nblocks, true, stream);
}

extern "C" int tomo_halo_unpack(const void *staging_dev, void *const *dst_dev, const size_t *bytes, int nblocks, void *stream)
{
    return halo_copy((const void *const *)dst_dev, staging_dev, bytes, nblocks, false, stream);
}
