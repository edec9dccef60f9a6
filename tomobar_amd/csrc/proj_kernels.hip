// Parallel-beam 3D projector pair for gfx950 with the FISTA / ADMM epilogues fused in.
//
// Model (fixed by the reference's geometry, supp/funcs.py:45-65, and by the ASTRA operators it calls at
// astra_base.py:554,601; restated in oracle/tomo_oracle.c):
//   BP  voxel-driven: vol[z,y,x] = sum_a lerp_u( sino[z,a,:], x_w cos + y_w sin - cor + nu/2 - 1/2 )
//   FP  ray-driven Joseph: step along the dominant axis, 2-tap linear interpolation along the other in-plane
//       axis, zero outside the volume, scaled by the ray length per step.
// MI355X mapping:
//   * BP variant 0 ("brick", bp_brick.inl): a 256-thread workgroup owns a 32(x) x 16(y) x 16(z) voxel brick, a
//     wave a 16 x 4 patch of it.  For a batch of 8 angles the [angle][z-quad][u] window of the sinogram that the
//     brick can touch is staged in LDS as float4 over z: one ds_read_b128 per tap serves four slices,
//     interpolation index/weights are computed once per (voxel column, angle) and reused by all 16 slices.
//     Workgroups are numbered so that each XCD's L2 sees one z-batch of sinogram rows at a time.
//   * BP variant 2 ("tiled"): the first LDS kernel (64 x 8 x 16 brick, lanes along x), kept for A/B runs.
//   * BP variant 1 ("direct"): taps straight from global memory (L1/L2), 4 slices per thread.
//   * FP (fp_tiled.inl): one lane per detector pixel, 8 angles x 4 slices per lane, sequential march over the
//     volume rows staged in LDS (bit-identical to the oracle); x-stepping angles read an in-plane transposed
//     copy of the volume so that the interpolation axis is always the contiguous one.  Variant 1: no LDS.
//   * epilogues: plain / residual (FP) and plain / FISTA step / FISTA step+momentum / ADMM z-update (BP).
// No MFMA: there is no dense contraction on this path.
#include "tomo_common.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <map>
#include <mutex>
#include <numeric>
#include <string>
#include <type_traits>

namespace {

// ------------------------------------------------------------------------------------------ shared pieces
template <bool LERP8>
__device__ __forceinline__ float lerp_w(float f, float fl)
{
    float w = f - fl;
    if (LERP8) w = rintf(w * 256.0f) * (1.0f / 256.0f);
    return w;
}

// Robust re-weighting of a data residual (the Huber and Student's-t data terms of the reference's removed RecToolsIR class:
// _data_["huber_threshold"], _data_["studentst_threshold"], Demos/methods_IR_legacy/DemoFISTA_artifacts2D.py:197,348;
// formula-level, see include/tomo_mi355x.h): Huber  r <- (delta/|r|) r where |r| > delta;  Student's t  r <- (2/(delta^2 + r^2)) r
__device__ __forceinline__ float robust_weight(float r, int mode, float delta)
{
    if (mode == TOMO_ROBUST_HUBER) {
        const float ar = fabsf(r);
        if (ar > delta) r = (delta / ar) * r;
    } else if (mode == TOMO_ROBUST_STUDENTST) {
        r = (2.0f / (delta * delta + r * r)) * r;
    }
    return r;
}

enum { EPI_PLAIN = 0, EPI_FISTA = 1, EPI_FISTA_MOM = 2, EPI_ADMM = 3 };

struct BpArgs {
    const float *sino;       // [nz][na][nu]; zquad: [ceil(nz/4)][na][nu][4] (TOMO_RESIDUAL_ZQUAD, fused epilogues only)
    int zquad;
    const tomo_angle_t *tab; // na records
    int nz, n, nu, na;
    float *vol;              // EPI_PLAIN: output volume
    // fused epilogues
    const float *xt;         // FISTA: X_t (in)         ADMM: x
    float *xout;             // FISTA: X (out)          ADMM: z (in/out)
    float *xt_out;           // FISTA_MOM: new X_t      ADMM: zu (out)
    const float *xold;       // FISTA_MOM: X_old (==xout buffer)   ADMM: u
    float s0, s1, s2, s3;    // FISTA: l_inv, beta ; ADMM: tau, rho, (1-alpha), alpha
    int nonneg, relax_on;
    int ntx, nty, nzb;       // tiled variant: tile counts
    int epi_aligned;         // every array the epilogue touches is 16-byte aligned (dwordx4 epilogue of the brick kernel)
    int device;              // host side only: the context's device (made current around the launch)
    std::string *path;       // host side only: receives the name of the kernel that ran
};

// No contraction here: these are the separate CuPy ufunc roundings of methodsIR_CuPy.py:463-475,545-557.
template <int EPI>
__device__ __forceinline__ void bp_epilogue(const BpArgs &a, size_t idx, float g)
{
    if (EPI == EPI_PLAIN) {
        a.vol[idx] = g;
    } else if (EPI == EPI_FISTA) {
        float x = a.xt[idx] - a.s0 * g;
        if (a.nonneg) x = x < 0.0f ? 0.0f : x;
        a.xout[idx] = x;
    } else if (EPI == EPI_FISTA_MOM) {
        float x = a.xt[idx] - a.s0 * g;
        if (a.nonneg) x = x < 0.0f ? 0.0f : x;
        const float xo = a.xold[idx];
        a.xout[idx] = x;
        a.xt_out[idx] = x + a.s1 * (x - xo);
    } else {
        const float z0 = a.xout[idx], u = a.xold[idx];
        const float ga = a.s1 * ((z0 - a.xt[idx]) + u);
        float z = z0 - a.s0 * (g + ga);
        if (a.nonneg) z = z < 0.0f ? 0.0f : z;
        if (a.relax_on) z = a.s2 * z0 + a.s3 * z;
        a.xout[idx] = z;
        a.xt_out[idx] = z + u;
    }
}

// the same epilogues on four consecutive voxels of a row (idx a multiple of 4, 16-byte aligned arrays): one dwordx4 access
// per array instead of four dword accesses; the arithmetic per voxel is that of bp_epilogue
template <int EPI>
__device__ __forceinline__ void bp_epilogue4(const BpArgs &a, size_t idx, float4 g)
{
    // The epilogue's volume streams are touched once (X_t read, X written: 8.6 GB through a 4 MB L2 per 1024^3 call) while the
    // sinogram windows of a z-brick (4.9 MB) are re-read by every brick of the XCD: with ordinary loads / stores the streams
    // evict the windows and the L2 re-fetches them ~33 x from the fabric (14.6 GB per call); marked NON-TEMPORAL they leave the
    // windows resident -- fetch 14.6 -> 4.9 GB (X_t + 1.8 x the subset), call 7.09 -> 6.90 ms (profiles/r5i_bp_epilogue_nontemporal_ab.txt).
#ifndef TOMO_BP_EPI_TEMPORAL   // (A/B builds only: tools/run_ab.sh)
    typedef float v4 __attribute__((ext_vector_type(4)));
    auto ld4 = [&](const float *p) {
        const v4 t = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(p + idx));
        return make_float4(t.x, t.y, t.z, t.w);
    };
    auto st4 = [&](float *p, float4 v) { __builtin_nontemporal_store(v4{v.x, v.y, v.z, v.w}, reinterpret_cast<v4 *>(p + idx)); };
#else
    auto ld4 = [&](const float *p) { return *reinterpret_cast<const float4 *>(p + idx); };
    auto st4 = [&](float *p, float4 v) { *reinterpret_cast<float4 *>(p + idx) = v; };
#endif
    const float gv[4] = {g.x, g.y, g.z, g.w};
    if (EPI == EPI_PLAIN) {
        st4(a.vol, g);
    } else if (EPI == EPI_FISTA) {
        const float4 t4 = ld4(a.xt);
        const float tv[4] = {t4.x, t4.y, t4.z, t4.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float x = tv[k] - a.s0 * gv[k];
            if (a.nonneg) x = x < 0.0f ? 0.0f : x;
            o[k] = x;
        }
        st4(a.xout, make_float4(o[0], o[1], o[2], o[3]));
    } else if (EPI == EPI_FISTA_MOM) {
        const float4 t4 = ld4(a.xt), o4 = ld4(a.xold);
        const float tv[4] = {t4.x, t4.y, t4.z, t4.w}, ov[4] = {o4.x, o4.y, o4.z, o4.w};
        float o[4], m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float x = tv[k] - a.s0 * gv[k];
            if (a.nonneg) x = x < 0.0f ? 0.0f : x;
            o[k] = x;
            m[k] = x + a.s1 * (x - ov[k]);
        }
        st4(a.xout, make_float4(o[0], o[1], o[2], o[3]));
        st4(a.xt_out, make_float4(m[0], m[1], m[2], m[3]));
    } else {
        const float4 z4 = ld4(a.xout), u4 = ld4(a.xold), t4 = ld4(a.xt);
        const float zv[4] = {z4.x, z4.y, z4.z, z4.w}, uv[4] = {u4.x, u4.y, u4.z, u4.w}, tv[4] = {t4.x, t4.y, t4.z, t4.w};
        float o[4], m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float ga = a.s1 * ((zv[k] - tv[k]) + uv[k]);
            float z = zv[k] - a.s0 * (gv[k] + ga);
            if (a.nonneg) z = z < 0.0f ? 0.0f : z;
            if (a.relax_on) z = a.s2 * zv[k] + a.s3 * z;
            o[k] = z;
            m[k] = z + uv[k];
        }
        st4(a.xout, make_float4(o[0], o[1], o[2], o[3]));
        st4(a.xt_out, make_float4(m[0], m[1], m[2], m[3]));
    }
}

// ------------------------------------------------------------------------------------------ BP variant 1
template <int EPI, bool LERP8>
__global__ __launch_bounds__(256) void bp_direct_kernel(BpArgs a)
{
    constexpr int ZB = 4;
    const int ix = blockIdx.x * 64 + (threadIdx.x & 63);
    const int iy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int z0 = blockIdx.z * ZB;
    if (ix >= a.n || iy >= a.n) return;
    const float half_n = 0.5f * (float)a.n - 0.5f, half_u = 0.5f * (float)a.nu - 0.5f;
    const float xw = (float)ix - half_n, yw = (float)iy - half_n;
    float acc[ZB] = {0.0f, 0.0f, 0.0f, 0.0f};
    const size_t zstride = (size_t)a.na * a.nu;
    for (int k = 0; k < a.na; ++k) {
        const tomo_angle_t t = a.tab[k];
        const float off = half_u - t.cor;
        const float f = fmaf(xw, t.cs, fmaf(yw, t.sn, off));
        const float fl = floorf(f);
        const float w = lerp_w<LERP8>(f, fl), omw = 1.0f - w;
        const int i0 = (int)fl;
        const bool ok0 = (i0 >= 0) && (i0 < a.nu), ok1 = (i0 + 1 >= 0) && (i0 + 1 < a.nu);
        if (a.zquad) {  // uniform: the private residual layout, four slices per 16-byte word
            const float4 *row4 = reinterpret_cast<const float4 *>(a.sino) + ((size_t)(z0 >> 2) * a.na + k) * a.nu;
            const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            const float4 q0 = ok0 ? row4[i0] : zero4, q1 = ok1 ? row4[i0 + 1] : zero4;
            const float t0[ZB] = {q0.x, q0.y, q0.z, q0.w}, t1[ZB] = {q1.x, q1.y, q1.z, q1.w};
#pragma unroll
            for (int zz = 0; zz < ZB; ++zz) {
                acc[zz] = fmaf(omw, t0[zz], acc[zz]);
                acc[zz] = fmaf(w, t1[zz], acc[zz]);
            }
            continue;
        }
        const float *row = a.sino + ((size_t)z0 * a.na + k) * a.nu;
#pragma unroll
        for (int zz = 0; zz < ZB; ++zz) {
            const bool zok = z0 + zz < a.nz;
            const float s0 = (ok0 && zok) ? row[zz * zstride + i0] : 0.0f;
            const float s1 = (ok1 && zok) ? row[zz * zstride + i0 + 1] : 0.0f;
            acc[zz] = fmaf(omw, s0, acc[zz]);
            acc[zz] = fmaf(w, s1, acc[zz]);
        }
    }
#pragma unroll
    for (int zz = 0; zz < ZB; ++zz)
        if (z0 + zz < a.nz) bp_epilogue<EPI>(a, ((size_t)(z0 + zz) * a.n + iy) * a.n + ix, acc[zz]);
}

// ------------------------------------------------------------------------------------------ BP variant 0
constexpr int BP_TX = 64, BP_TY = 8, BP_RY = 2, BP_ZQ = 4, BP_AB = 8, BP_PITCH = 72;

template <int EPI, bool LERP8>
__global__ __launch_bounds__(256) void bp_tiled_kernel(BpArgs a)
{
    __shared__ float4 tile[BP_AB][BP_ZQ][BP_PITCH];  // 36 KiB
    __shared__ int umin_s[BP_AB];

    // XCD-aware numbering: workgroup b lands on XCD b%8; give each XCD its own z-batch stream
    const int ntiles = a.ntx * a.nty;
    const int b = blockIdx.x, xcd = b & 7, q = b >> 3;
    const int per_xcd = (a.nzb * ntiles + 7) >> 3;  // a contiguous eighth of the (z-batch, tile) list per XCD
    const int wi = xcd * per_xcd + q;
    if (wi >= a.nzb * ntiles) return;  // uniform for the workgroup
    const int zb = wi / ntiles;
    const int tid = wi % ntiles;
    const int tx0 = (tid % a.ntx) * BP_TX, ty0 = (tid / a.ntx) * BP_TY;
    const int z0 = zb * (4 * BP_ZQ);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ix = tx0 + lane;
    const float half_n = 0.5f * (float)a.n - 0.5f, half_u = 0.5f * (float)a.nu - 0.5f;
    const float xw = (float)ix - half_n;
    float yw[BP_RY];
#pragma unroll
    for (int r = 0; r < BP_RY; ++r) yw[r] = (float)(ty0 + wave * BP_RY + r) - half_n;

    float acc[BP_RY][4 * BP_ZQ];
#pragma unroll
    for (int r = 0; r < BP_RY; ++r)
#pragma unroll
        for (int j = 0; j < 4 * BP_ZQ; ++j) acc[r][j] = 0.0f;

    const size_t zstride = (size_t)a.na * a.nu;
    for (int a0 = 0; a0 < a.na; a0 += BP_AB) {
        const int nb = min(BP_AB, a.na - a0);
        __syncthreads();  // previous batch fully consumed
        if ((int)threadIdx.x < nb) {
            // detector window of the brick: the coordinate is monotone in x and in y, so the four corners bound it
            const tomo_angle_t t = a.tab[a0 + threadIdx.x];
            const float off = half_u - t.cor;
            const float x0 = (float)tx0 - half_n, x1 = (float)(tx0 + BP_TX - 1) - half_n;
            const float y0 = (float)ty0 - half_n, y1 = (float)(ty0 + BP_TY - 1) - half_n;
            const float f00 = fmaf(x0, t.cs, fmaf(y0, t.sn, off)), f10 = fmaf(x1, t.cs, fmaf(y0, t.sn, off));
            const float f01 = fmaf(x0, t.cs, fmaf(y1, t.sn, off)), f11 = fmaf(x1, t.cs, fmaf(y1, t.sn, off));
            umin_s[threadIdx.x] = (int)floorf(fminf(fminf(f00, f10), fminf(f01, f11)));
        }
        __syncthreads();
        // stage [angle][z-quad][u] as float4 over z: four coalesced row reads, one 16-byte LDS write
        for (int item = threadIdx.x; item < nb * BP_ZQ * BP_PITCH; item += 256) {
            const int j = item % BP_PITCH;
            const int zq = (item / BP_PITCH) % BP_ZQ;
            const int aa = item / (BP_PITCH * BP_ZQ);
            const int u = umin_s[aa] + j;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (u >= 0 && u < a.nu) {
                const int z = z0 + zq * 4;
                const float *p = a.sino + ((size_t)z * a.na + (a0 + aa)) * a.nu + u;
                if (z + 0 < a.nz) v.x = p[0];
                if (z + 1 < a.nz) v.y = p[zstride];
                if (z + 2 < a.nz) v.z = p[2 * zstride];
                if (z + 3 < a.nz) v.w = p[3 * zstride];
            }
            tile[aa][zq][j] = v;
        }
        __syncthreads();
        for (int aa = 0; aa < nb; ++aa) {
            const tomo_angle_t t = a.tab[a0 + aa];
            const float off = half_u - t.cor;
            const int um = umin_s[aa];
#pragma unroll
            for (int r = 0; r < BP_RY; ++r) {
                const float f = fmaf(xw, t.cs, fmaf(yw[r], t.sn, off));
                const float fl = floorf(f);
                const float w = lerp_w<LERP8>(f, fl), omw = 1.0f - w;
                const int idx = (int)fl - um;
#pragma unroll
                for (int zq = 0; zq < BP_ZQ; ++zq) {
                    const float4 s0 = tile[aa][zq][idx];
                    const float4 s1 = tile[aa][zq][idx + 1];
                    float *c = &acc[r][zq * 4];
                    c[0] = fmaf(omw, s0.x, c[0]); c[0] = fmaf(w, s1.x, c[0]);
                    c[1] = fmaf(omw, s0.y, c[1]); c[1] = fmaf(w, s1.y, c[1]);
                    c[2] = fmaf(omw, s0.z, c[2]); c[2] = fmaf(w, s1.z, c[2]);
                    c[3] = fmaf(omw, s0.w, c[3]); c[3] = fmaf(w, s1.w, c[3]);
                }
            }
        }
    }
    if (ix < a.n) {
#pragma unroll
        for (int r = 0; r < BP_RY; ++r) {
            const int iy = ty0 + wave * BP_RY + r;
            if (iy < a.n) {
#pragma unroll
                for (int j = 0; j < 4 * BP_ZQ; ++j)
                    if (z0 + j < a.nz) bp_epilogue<EPI>(a, ((size_t)(z0 + j) * a.n + iy) * a.n + ix, acc[r][j]);
            }
        }
    }
}

#include "bp_brick.inl"

// planar sinogram [nz][row] (row = na * nu) -> the quad-interleaved layout [ceil(nz/4)][row][4] the brick kernel stages by
// LDS-DMA; slices past nz are zeros.  One coalesced pass: 8 B per sample moved, ~0.13 ms for a 75-angle subset at 1024^3.
__global__ __launch_bounds__(256) void sino_to_quad_kernel(const float *__restrict__ in, float4 *__restrict__ out, int nz, size_t row)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int z = 4 * (int)blockIdx.y;
    if (i >= row) return;
    const float *p = in + (size_t)z * row + i;
    float4 v;
    v.x = p[0];
    v.y = z + 1 < nz ? p[row] : 0.0f;
    v.z = z + 2 < nz ? p[2 * row] : 0.0f;
    v.w = z + 3 < nz ? p[3 * row] : 0.0f;
    out[(size_t)blockIdx.y * row + i] = v;
}

constexpr size_t BP_RELAY_MAX_BYTES = (size_t)16 << 30;  // largest sinogram tomo_bp3d* re-lays into its scratch arena
// Smallest relay scratch a DEVICE could not provide (out of memory only): requests that large go straight to the planar staging
// instead of failing a hipMalloc per call.  Not for ever: another tenant's placement search or a framework's caching allocator
// may have held the memory for a moment, so every BP_RELAY_RETRY-th refused call asks again and tomo_release_scratch(device)
// forgets the refusal (tomo_bp_relay_reset).
constexpr int BP_RELAY_DEVICES = 64, BP_RELAY_RETRY = 64;
struct bp_relay_state { std::atomic<size_t> refused{~(size_t)0}; std::atomic<unsigned> skipped{0}; };
bp_relay_state g_bp_relay[BP_RELAY_DEVICES];

template <int EPI>
int bp_launch(BpArgs a, bool lerp8, hipStream_t st)
{
    TOMO_ON_DEVICE(a.device);
    tomo_prof_scope prof(PROF_BP, st, 1);
    {
        uintptr_t bits = 0;
        if (EPI == EPI_PLAIN) bits |= (uintptr_t)a.vol;
        else bits |= (uintptr_t)a.xt | (uintptr_t)a.xout;
        if (EPI == EPI_FISTA_MOM || EPI == EPI_ADMM) bits |= (uintptr_t)a.xt_out | (uintptr_t)a.xold;
        a.epi_aligned = (bits & 15) == 0;
    }
    // the brick kernel addresses a 16-slice sinogram slab with 32-bit element offsets
    // (the quad layout is read through one buffer descriptor per z-brick and batch: 4 quads x 16 bytes < 2^31)
    if (EPI == EPI_PLAIN) a.zquad = 0;  // tomo_bp3d always takes the public layout
    if (a.zquad && (((uintptr_t)a.sino) & 15) != 0)
        return tomo_fail(TOMO_E_INVALID, "the quad-interleaved residual must be 16-byte aligned");
    // A PLANAR sinogram (tomo_bp3d: FBP, SIRT, CGLS, OSEM, the power method; the fused epilogues of the ring-term and
    // vertical-CoR paths) is re-laid quad-interleaved into this stream's scratch arena first: one streaming pass (8 B per
    // sample) buys the LDS-DMA staging of the brick kernel -- 4 workgroups per CU, no staging registers -- which is worth 7 x
    // as much per 75-angle subset and 8 x for a whole angle set (docs/kernels/bp.md).  Not for sinograms beyond 16 GiB, not
    // when the arena cannot be had (then the planar staging runs), and not with bp variant 3 (dev flavour: planar staging, A/B).
    bool relaid = false;
    if (g_variant_bp == 0 && !a.zquad && (long)a.na * a.nu < (1L << 25) && a.na > 0) {
        const size_t row = (size_t)a.na * a.nu, nq = (size_t)ceil_div(a.nz, 4), bytes = nq * row * 16;
        void *q = nullptr;
        bp_relay_state &rs = g_bp_relay[a.device >= 0 && a.device < BP_RELAY_DEVICES ? a.device : 0];
        bool have = bytes <= BP_RELAY_MAX_BYTES && nq <= 65535;
        if (have && bytes >= rs.refused.load() && (rs.skipped.fetch_add(1) + 1) % BP_RELAY_RETRY != 0) have = false;
        if (have) {
            const int rc = tomo_arena_get(a.device, st, ARENA_BPQ, bytes, &q);
            if (rc == TOMO_E_NOMEM) {   // the planar staging runs, now and for the next requests this large on this device
                rs.refused.store(std::min(rs.refused.load(), bytes));
                tomo_warn_once("bp_relay", "back projection: no memory for the quad-interleaved relay of a planar sinogram, "
                                           "running the planar staging (slower; asked again every 64th call and after tomo_release_scratch)");
                have = false;
            } else if (rc != TOMO_OK) {
                return rc;   // a caller error or a runtime failure is reported, not papered over
            } else {
                rs.refused.store(~(size_t)0);
            }
        }
        if (have) {
            sino_to_quad_kernel<<<dim3((unsigned)((row + 255) / 256), (unsigned)nq), 256, 0, st>>>(a.sino, (float4 *)q, a.nz, row);
            TOMO_LAUNCH_CHECK();
            a.sino = (const float *)q;
            a.zquad = 1;
            relaid = true;
        }
    }
    const bool brick_ok = (long)a.na * a.nu < (a.zquad ? (1L << 25) : (1L << 27));
    const int variant = g_variant_bp == 3 ? 0 : g_variant_bp;  // 3 = the default kernel on the planar layout
    if (variant == 1 || (variant == 0 && !brick_ok)) {
        if (variant == 0)
            tomo_warn_once("bp_direct", "back projection: angles x detector >= 2^27 samples per slice, falling back to "
                                        "the direct (no LDS) kernel (about 3.5x slower)");
        if (a.path) *a.path = "direct(no LDS)";
        dim3 grid(ceil_div(a.n, 64), ceil_div(a.n, 4), ceil_div(a.nz, 4));
        if (lerp8) bp_direct_kernel<EPI, true><<<grid, 256, 0, st>>>(a);
        else bp_direct_kernel<EPI, false><<<grid, 256, 0, st>>>(a);
    } else if (variant == 0) {
        if (a.path) *a.path = "brick(32x16x16)";
        a.ntx = ceil_div(a.n, BB_TX);
        a.nty = ceil_div(a.n, BB_TY);
        a.nzb = ceil_div(a.nz, 4 * BB_ZQ);
        const long blocks = 8L * (((long)a.nzb * a.ntx * a.nty + 7) / 8);
        if (blocks > 0x7fffffffL) return tomo_fail(TOMO_E_INVALID, "volume too large for one BP launch");
        if (a.zquad) {
            if (a.path) *a.path = relaid ? "brick(32x16x16, planar sinogram re-laid quad-interleaved, LDS-DMA staging)"
                                         : "brick(32x16x16, quad-interleaved residual, LDS-DMA staging)";
            if (lerp8) bp_brick_kernel<EPI, true, true><<<(unsigned)blocks, 256, 0, st>>>(a);
            else bp_brick_kernel<EPI, false, true><<<(unsigned)blocks, 256, 0, st>>>(a);
            TOMO_LAUNCH_CHECK();
            return TOMO_OK;
        }
        if (lerp8) bp_brick_kernel<EPI, true><<<(unsigned)blocks, 256, 0, st>>>(a);
        else bp_brick_kernel<EPI, false><<<(unsigned)blocks, 256, 0, st>>>(a);
    }
#if TOMO_DEV
    else {  // variant 2 (dev flavour): the round-1 tiling with lanes along x, kept for A/B measurement
        if (a.zquad) return tomo_fail(TOMO_E_INVALID, "bp variant 2 reads the planar residual layout only");
        if (a.path) *a.path = "tiled(64x8x16)";
        a.ntx = ceil_div(a.n, BP_TX);
        a.nty = ceil_div(a.n, BP_TY);
        a.nzb = ceil_div(a.nz, 4 * BP_ZQ);
        const long blocks = 8L * (((long)a.nzb * a.ntx * a.nty + 7) / 8);
        if (blocks > 0x7fffffffL) return tomo_fail(TOMO_E_INVALID, "volume too large for one BP launch");
        if (lerp8) bp_tiled_kernel<EPI, true><<<(unsigned)blocks, 256, 0, st>>>(a);
        else bp_tiled_kernel<EPI, false><<<(unsigned)blocks, 256, 0, st>>>(a);
    }
#endif
    TOMO_LAUNCH_CHECK();
    return TOMO_OK;
}

// ------------------------------------------------------------------------------------------ FP
struct FpArgs {
    const float *vol;        // [nz][n][n]   (y rows, x contiguous)
    const float *volT;       // [nz][n][n]   (x rows, y contiguous) -- may be null when no angle steps along x
    const tomo_angle_t *tab;
    int nz, n, nu, na, na_full;
    float *out;              // [nz][na][nu]
    const float *b;          // residual epilogue: full sinogram [nz][na_full][nu] (null = plain FP)
    const float *w;          // PWLS weights, full sinogram (may be null)
    const float *ring;       // Group-Huber offsets [nz][nu] (may be null)
    float ring_scale;
    int fidelity;
    int gathered;            // bit0: b is the gathered subset, bit1: w is
    int zquad;               // residual epilogue: out is [ceil(nz/4)][na][nu][4] (TOMO_RESIDUAL_ZQUAD)
    int robust;              // TOMO_ROBUST_*: re-weighting of the LS / PWLS residual (Huber, Student's t)
    float rdelta;            // its threshold
};

// The residual epilogue of every forward-projection kernel (data_fidelities.py:28-39 + the ring offsets + the robust
// re-weightings): `val` = (A_s x)[z, k_a, iu]; Args = FpArgs or FpTiledArgs (same field names).
template <class Args>
__device__ __forceinline__ float fp_residual_value(const Args &a, float val, int z, int k_a, int src, int iu)
{
    const size_t gi = ((size_t)z * a.na + k_a) * a.nu + iu;          // gathered layout
    const size_t fi = ((size_t)z * a.na_full + src) * a.nu + iu;     // full-sinogram layout
    const float bv = a.b[(a.gathered & TOMO_GATHERED_B) ? gi : fi];
    if (a.fidelity == TOMO_FID_KL || a.fidelity == TOMO_FID_RATIO) {
        const float ax = val < 1e-8f ? 1e-8f : val;
        const float q = bv / ax;
        return (a.fidelity == TOMO_FID_KL) ? 1.0f - q : q;
    }
    val = val - bv;
    if (a.ring) val = val + a.ring_scale * a.ring[(size_t)z * a.nu + iu];
    if (a.w) val = val * a.w[(a.gathered & TOMO_GATHERED_W) ? gi : fi];
    return robust_weight(val, a.robust, a.rdelta);
}

__global__ __launch_bounds__(256) void transpose_inplane_kernel(const float *__restrict__ in, float *__restrict__ out, int n)
{
    __shared__ float t[32][33];
    const size_t zoff = (size_t)blockIdx.z * n * n;
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;  // 32 x 8
    for (int r = ly; r < 32; r += 8) {
        const int x = bx + lx, y = by + r;
        t[r][lx] = (x < n && y < n) ? in[zoff + (size_t)y * n + x] : 0.0f;
    }
    __syncthreads();
    for (int r = ly; r < 32; r += 8) {
        const int y = by + lx, x = bx + r;
        if (x < n && y < n) out[zoff + (size_t)x * n + y] = t[lx][r];
    }
}

// X_t = X + beta (X - X_old) (the momentum step of methodsIR_CuPy.py:475: mul then add, no fma) written twice: as is,
// and in-plane transposed into the forward projector's scratch copy -- the next forward projection of X_t then needs no
// transpose pass of its own (one read of X_t saved per sub-iteration)
__global__ __launch_bounds__(256) void momentum_transpose_kernel(const float *__restrict__ x, const float *__restrict__ xold,
                                                                float *__restrict__ xt, float *__restrict__ xt_T, float beta, int n)
{
    __shared__ float t[32][33];
    const size_t zoff = (size_t)blockIdx.z * n * n;
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;  // 32 x 8
    for (int r = ly; r < 32; r += 8) {
        const int xx = bx + lx, yy = by + r;
        float v = 0.0f;
        if (xx < n && yy < n) {
            const size_t i = zoff + (size_t)yy * n + xx;
            const float a = x[i];
            v = a + beta * (a - xold[i]);
            xt[i] = v;
        }
        t[r][lx] = v;
    }
    __syncthreads();
    for (int r = ly; r < 32; r += 8) {
        const int yy = by + lx, xx = bx + r;
        if (xx < n && yy < n) xt_T[zoff + (size_t)xx * n + yy] = t[lx][r];
    }
}

// The same two kernels on 64 x 64 tiles with 16-byte accesses (round 5): a wave reads four whole 256-byte row segments per
// instruction instead of two 128-byte ones, and the transposed tile leaves as float4 along y.  Needs n % 4 == 0 and 16-byte
// aligned arrays (the launchers fall back to the 32 x 32 dword kernels otherwise).  MOMENTUM = false: plain in-plane transpose.
// LDS tile [64][65]: the transposed read of lane (c, r) touches bank (4c + j + r) mod 64 -- conflict-free for every j.
// Every stream is touched once per call and is far larger than the L2: non-temporal accesses.  Same-box A/B at 1024^3 (momentum
// + transposed copy, profiles/r5w_momentum_tile_ab.txt): 32 x 32 dword 3.48 ms, 64 x 64 float4 3.34, + non-temporal 3.25.
typedef float glue_v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_ld4(const float *p)
{
    const glue_v4 v = __builtin_nontemporal_load(reinterpret_cast<const glue_v4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_st4(float *p, float4 v)
{
    __builtin_nontemporal_store(glue_v4{v.x, v.y, v.z, v.w}, reinterpret_cast<glue_v4 *>(p));
}
template <bool MOMENTUM>
__global__ __launch_bounds__(256) void transpose64_kernel(const float *__restrict__ x, const float *__restrict__ xold,
                                                          float *__restrict__ xt, float *__restrict__ xt_T, float beta, int n)
{
    __shared__ float t[64][65];
    const size_t zoff = (size_t)blockIdx.z * n * n;
    const int bx = blockIdx.x * 64, by = blockIdx.y * 64;
    const int c4 = (threadIdx.x & 15) * 4, r0 = threadIdx.x >> 4;  // 16 float4 columns x 16 rows per pass
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + 16 * k;
        const int xx = bx + c4, yy = by + r;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (xx < n && yy < n) {
            const size_t i = zoff + (size_t)yy * n + xx;
            const float4 a = nt_ld4(x + i);
            if (MOMENTUM) {
                const float4 o = nt_ld4(xold + i);
                v.x = a.x + beta * (a.x - o.x); v.y = a.y + beta * (a.y - o.y);
                v.z = a.z + beta * (a.z - o.z); v.w = a.w + beta * (a.w - o.w);
                nt_st4(xt + i, v);
            } else {
                v = a;
            }
        }
        t[r][c4] = v.x; t[r][c4 + 1] = v.y; t[r][c4 + 2] = v.z; t[r][c4 + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + 16 * k;           // column of the tile = row of the transposed copy
        const int xx = bx + r, yy = by + c4;
        if (xx < n && yy < n)
            nt_st4(xt_T + zoff + (size_t)xx * n + yy, make_float4(t[c4][r], t[c4 + 1][r], t[c4 + 2][r], t[c4 + 3][r]));
    }
}

static inline bool aligned16(const void *a, const void *b, const void *c, const void *d)
{
#ifdef TOMO_GLUE32   // A/B builds only (tools/run_ab.sh): the 32 x 32 dword kernels everywhere
    return false;
#endif
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(d)) & 15) == 0;
}

template <bool LERP8, bool RESID>
__global__ __launch_bounds__(256) void fp_march_kernel(FpArgs a)
{
    constexpr int ZB = 4;
    const int iu = blockIdx.x * 256 + threadIdx.x;
    const int k_a = blockIdx.y;
    const int z0 = blockIdx.z * ZB;
    if (iu >= a.nu) return;
    const tomo_angle_t t = a.tab[k_a];
    const float half_n = 0.5f * (float)a.n - 0.5f, half_u = 0.5f * (float)a.nu - 0.5f;
    const float s = ((float)iu - half_u) + t.cor;
    const float offset = fmaf(s, t.inv, half_n);
    const float *src = t.dirx ? a.volT : a.vol;  // interpolation axis contiguous in either case
    const size_t zstride = (size_t)a.n * a.n;
    const int n = a.n;
    float acc[ZB] = {0.0f, 0.0f, 0.0f, 0.0f};
    const float *base = src + (size_t)z0 * zstride;
    bool zok[ZB];
#pragma unroll
    for (int zz = 0; zz < ZB; ++zz) zok[zz] = z0 + zz < a.nz;
#pragma unroll 2
    for (int k = 0; k < n; ++k) {
        const float kw = (float)k - half_n;
        const float f = fmaf(kw, t.slope, offset);
        const float fl = floorf(f);
        const float w = lerp_w<LERP8>(f, fl), omw = 1.0f - w;
        const int i0 = (int)fl;
        const bool ok0 = (i0 >= 0) && (i0 < n), ok1 = (i0 + 1 >= 0) && (i0 + 1 < n);
        const float *row = base + (size_t)k * n;
#pragma unroll
        for (int zz = 0; zz < ZB; ++zz) {
            const float v0 = (ok0 && zok[zz]) ? row[zz * zstride + i0] : 0.0f;
            const float v1 = (ok1 && zok[zz]) ? row[zz * zstride + i0 + 1] : 0.0f;
            acc[zz] = fmaf(omw, v0, acc[zz]);
            acc[zz] = fmaf(w, v1, acc[zz]);
        }
    }
    float v4[ZB] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int zz = 0; zz < ZB; ++zz) {
        if (!zok[zz]) continue;
        const int z = z0 + zz;
        float val = acc[zz] * t.scale;
        if (RESID) val = fp_residual_value(a, val, z, k_a, t.src, iu);
        v4[zz] = val;
        if (!(RESID && a.zquad)) a.out[((size_t)z * a.na + k_a) * a.nu + iu] = val;
    }
    if (RESID && a.zquad)
        reinterpret_cast<float4 *>(a.out)[((size_t)(z0 >> 2) * a.na + k_a) * a.nu + iu] = make_float4(v4[0], v4[1], v4[2], v4[3]);
}

#include "fp_tiled.inl"

// the context's in-plane transposed volume copy (x-stepping angles read it), grow-only
int fp_scratch(tomo_ctx *ctx)
{
    const size_t need = (size_t)ctx->nz * ctx->n * ctx->n * sizeof(float);
    if (ctx->scratch_bytes < need) {
        ctx->volT_valid = false;
        if (ctx->scratch) { TOMO_HIP(hipDeviceSynchronize()); TOMO_HIP(hipFree(ctx->scratch)); ctx->scratch = nullptr; ctx->scratch_bytes = 0; }
        TOMO_HIP(hipMalloc(&ctx->scratch, need));
        ctx->scratch_bytes = need;
    }
    return TOMO_OK;
}

int fp_run(tomo_ctx *ctx, int subset, const float *vol, const float *b, const float *w, int gathered, int fidelity,
           float *out, void *stream, const float *ring = nullptr, float ring_scale = 0.0f, int zquad = 0,
           int robust = TOMO_ROBUST_NONE, float rdelta = 0.0f)
{
    TOMO_REQUIRE(ctx != nullptr, "ctx is NULL");
    TOMO_REQUIRE(vol != nullptr && out != nullptr, "NULL data pointer");
    TOMO_REQUIRE(subset < ctx->os, "subset %d out of range (OS_number %d)", subset, ctx->os);
    tomo_subset &s = (subset < 0 || ctx->os == 1) ? ctx->subsets[0] : ctx->subsets[1 + subset];
    hipStream_t st = as_stream(stream);
    // the transposed copy tomo_momentum_transposed left behind is a ONE-SHOT token: whatever this call does with it
    // (use it, find it belongs to another volume / stream, return early), it is spent before anything else happens
    const bool ready = ctx->volT_valid && ctx->volT_of == vol && ctx->volT_stream == st && ctx->scratch != nullptr;
    ctx->volT_valid = false;
    ctx->volT_of = nullptr;
    if (s.size == 0) return TOMO_OK;
    TOMO_ON_DEVICE(ctx->device);
    FpArgs a;
    a.vol = vol;
    a.volT = nullptr;
    if (s.n_dirx > 0) {
        int rc = fp_scratch(ctx);
        if (rc != TOMO_OK) return rc;
        // tomo_momentum_transposed has just written this volume's transposed copy: use it once, skip the pass
        if (!ready) {
            if (ctx->n % 4 == 0 && aligned16(vol, ctx->scratch, nullptr, nullptr)) {
                dim3 tg(ceil_div(ctx->n, 64), ceil_div(ctx->n, 64), ctx->nz);
                transpose64_kernel<false><<<tg, 256, 0, st>>>(vol, nullptr, nullptr, (float *)ctx->scratch, 0.0f, ctx->n);
            } else {
                dim3 tg(ceil_div(ctx->n, 32), ceil_div(ctx->n, 32), ctx->nz);
                transpose_inplane_kernel<<<tg, 256, 0, st>>>(vol, (float *)ctx->scratch, ctx->n);
            }
            TOMO_LAUNCH_CHECK();
        }
        a.volT = (const float *)ctx->scratch;
    }
    a.tab = ctx->dev_table + s.table_offset;
    a.nz = ctx->nz; a.n = ctx->n; a.nu = ctx->nu; a.na = s.size; a.na_full = ctx->na;
    a.out = out; a.b = b; a.w = w; a.fidelity = fidelity; a.gathered = gathered;
    a.ring = ring; a.ring_scale = ring_scale;
    a.robust = (b != nullptr) ? robust : TOMO_ROBUST_NONE; a.rdelta = rdelta;
    a.zquad = (b != nullptr) ? zquad : 0;
    if (a.zquad && (((uintptr_t)out) & 15) != 0)
        return tomo_fail(TOMO_E_INVALID, "the quad-interleaved residual must be 16-byte aligned");
    dim3 grid(ceil_div(ctx->nu, 256), s.size, ceil_div(ctx->nz, 4));
    tomo_prof_scope prof(PROF_FP, st, 1);
    ctx->last_fp_path.clear();
    const bool l8 = (ctx->flags & TOMO_FLAG_LERP8) != 0;
    // tiled variant unless a class's LDS window would not fit (very wide angular spread on a very wide volume)
    bool tiled = (g_variant_fp != 1);
    if (tiled) {
        size_t off = s.table_offset;
        for (int c = 0; c < 4; ++c) {
            if (s.n_class[c] > 0 && s.wbound[c] < 0)
                s.wbound[c] = fp_window_bound(ctx->host_table.data() + s.table_offset, ctx->host_fp_order.data() + off,
                                              s.n_class[c], ctx->n, ctx->nu);
            if (s.n_class[c] > 0 && s.wbound[c] > 4000) tiled = false;  // one staged row must fit in 64 KiB of LDS
            off += s.n_class[c];
        }
    }
    if (tiled) {
        // ---- wide form: one 1024-thread workgroup per detector row (see fp_tiled.inl), one launch per stepping AXIS: a
        // whole-row window does not care about the sign of the detector slope, so the two classes of an axis (adjacent in
        // the order table) are merged -- 37 angles make 5 groups of 8 instead of 3 + 3.  Chosen when the 256-pixel tiles
        // would stage at least 0.9x what whole rows need (see `pays` below), and it fits in LDS.
        // measured at 384..640-wide detectors (tools/kernel_bench.py): the whole-row form is 7-37 % faster than 256-pixel
        // tiles there as well, so it is considered from 128 pixels up (it used to start at 768)
        constexpr int FP_WIDE_MIN_NU = 128;
        constexpr double FP_DENSE16_PAYS = 0.75;  // dense form taken when it stages <= 0.75x what whole rows x 8 angles would
        bool done[4] = {false, false, false, false};
        // try_wide: whole-row form for `nc` angles starting at `off_d` of the order table (one stepping class, or the two
        // sign classes of an axis merged); returns 1 if launched, 0 if the 256-pixel tiles are the better choice, < 0 on error
        auto try_wide = [&](int d, size_t off_d, int nc, int &wp_cache, double cost_tiles, const char *label) -> int {
            // tiles of equal width: 2560 pixels = 3 tiles of 896 (14 waves), not 2.5 tiles of 1024
            const int nut_w = ceil_div(a.nu, 1024);
            const int bt = std::min(1024, ceil_div(ceil_div(a.nu, nut_w), 64) * 64);
            if (wp_cache < 0)
                wp_cache = fp_window_bound(ctx->host_table.data() + s.table_offset, ctx->host_fp_order.data() + off_d, nc,
                                           ctx->n, ctx->nu, bt);
            const int wp = wp_cache;
            const double cost_rows = (double)nut_w * ceil_div(nc, FP_A) * wp;
            // measured: with equal staging volume the whole-row form is still the faster one (one workgroup per CU streams rows
            // through a deep prefetch pipeline; BASELINE configs[3], 1500 dense angles at 2048: 0.59 -> 0.51 s per ADMM
            // iteration), so it is taken whenever it does not stage clearly MORE than the 256-pixel tiles
            const bool pays = cost_tiles >= 0.9 * cost_rows;
            const int passes_w = ceil_div(wp, bt);
            // rows per chunk: 4 (or 2) double-buffered rows up to two column passes; detectors wider than 2048
            // (3-5 passes, BASELINE configs[4] is 2560 wide) keep ONE tile (two barriers per chunk) of as many
            // rows as fit in the 160 KiB of LDS next to the per-row window tables
            int kc_w = 4;
            const size_t tab_w = (size_t)a.n * 8;
            size_t smem_w = (size_t)2 * kc_w * wp * 16 + tab_w;
            if (passes_w <= 2) {
                if (smem_w > 160 * 1024) { kc_w = 2; smem_w = (size_t)2 * kc_w * wp * 16 + tab_w; }
            } else {
                kc_w = passes_w == 3 ? 2 : 1;  // 3 rows (9 float4 in flight per thread) spill at 1024 threads
                while (kc_w > 1 && (size_t)kc_w * wp * 16 + tab_w > 160 * 1024) --kc_w;
                smem_w = (size_t)kc_w * wp * 16 + tab_w;
            }
            if (!(pays && passes_w <= 5 && smem_w <= 160 * 1024)) return 0;
            // per-angle lane -> pixel multipliers (round 6), built once per context for this tile width; the un-permute of
            // the epilogue needs two LDS rows of bt float4 inside the tile area
            const int *mult_d = nullptr;
#ifndef TOMO_FP_NO_LANE_MULT   // A/B builds only (tools/run_ab.sh)
            // measured A/B (profiles/r6_fp_lane_multiplier_ab.txt): x 1.024 at 1024 pixels (configs[2]), x 1.047 at 896 (configs[4]
            // share), 0.97-0.99 at 256-640 pixels where the un-permute of the epilogue is not paid back: from 768 pixels up only
            constexpr int FP_MULT_MIN_BT = 768;
            if (bt >= FP_MULT_MIN_BT && (size_t)2 * bt * 16 <= smem_w - tab_w && g_variant_fp != 4) {
                if (ctx->dev_fp_mult == nullptr || ctx->fp_mult_bt != bt) {
                    std::vector<int> mh(ctx->host_fp_order.size(), 1);
                    for (const tomo_subset &ss : ctx->subsets)
                        for (int i = 0, ne = ss.n_class[0] + ss.n_class[1] + ss.n_class[2] + ss.n_class[3]; i < ne; ++i) {
                            const tomo_angle_t &rec = ctx->host_table[ss.table_offset + ctx->host_fp_order[ss.table_offset + i]];
                            mh[ss.table_offset + i] = fp_lane_mult(std::fabs((double)rec.inv), bt);
                        }
                    // (this lambda reports errors as -1, not as a TOMO_E_* code)
                    if (ctx->dev_fp_mult == nullptr) {
                        if (hipMalloc((void **)&ctx->dev_fp_mult, std::max<size_t>(mh.size(), 1) * sizeof(int)) != hipSuccess) return -1;
                    } else if (hipDeviceSynchronize() != hipSuccess) {   // a launch in flight may still read the old table
                        return -1;
                    }
                    if (hipMemcpy(ctx->dev_fp_mult, mh.data(), mh.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return -1;
                    ctx->fp_mult_bt = bt;
                }
                mult_d = ctx->dev_fp_mult + off_d;
            }
#endif
            FpTiledArgs t;
            t.src = d ? a.volT : a.vol;
            t.tab = a.tab;
            t.order = ctx->dev_fp_order + off_d;
            t.mult = mult_d;
            t.n_class = nc;
            t.nz = a.nz; t.n = a.n; t.nu = a.nu; t.na = a.na; t.na_full = a.na_full;
            t.out = out; t.b = b; t.w = w; t.fidelity = fidelity; t.gathered = gathered;
            t.ring = ring; t.ring_scale = ring_scale; t.zquad = a.zquad; t.robust = a.robust; t.rdelta = a.rdelta;
            t.wpitch = wp;
#if TOMO_DEV
            t.probe = g_probe;
#endif
            t.nut = nut_w;
            t.bt = bt;
            t.ngroups = ceil_div(nc, FP_A);
            t.nzb = ceil_div(a.nz, 4);
            const long blocks_w = 8L * (((long)t.nzb * t.nut * t.ngroups + 7) / 8);
            if (blocks_w > 0x7fffffffL) return -1;
#define FP_WIDE_LAUNCH(L8, RES)                                                                                        \
    do {                                                                                                               \
        auto launch = [&](auto kern) {                                                                                 \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)smem_w);                                                                    \
            kern<<<(unsigned)blocks_w, bt, smem_w, st>>>(t);                                                         \
        };                                                                                                             \
        if (passes_w == 1) launch(fp_tiled_kernel<L8, RES, 1, 4, true, 1024>);                                         \
        else if (passes_w == 2 && kc_w == 4) launch(fp_tiled_kernel<L8, RES, 2, 8, true, 1024>);                       \
        else if (passes_w == 2) launch(fp_tiled_kernel<L8, RES, 2, 4, true, 1024>);                                    \
        else if (passes_w == 3 && kc_w == 2) launch(fp_tiled_kernel<L8, RES, 3, 6, false, 1024>);                      \
        else if (passes_w == 3) launch(fp_tiled_kernel<L8, RES, 3, 3, false, 1024>);                                   \
        else if (passes_w == 4) launch(fp_tiled_kernel<L8, RES, 4, 4, false, 1024>);                                   \
        else launch(fp_tiled_kernel<L8, RES, 5, 5, false, 1024>);                                                      \
    } while (0)
            if (b) { if (l8) FP_WIDE_LAUNCH(true, true); else FP_WIDE_LAUNCH(false, true); }
            else   { if (l8) FP_WIDE_LAUNCH(true, false); else FP_WIDE_LAUNCH(false, false); }
#undef FP_WIDE_LAUNCH
            if (hipGetLastError() != hipSuccess) return -1;
            ctx->last_fp_path += label;
            ctx->last_fp_path += "whole-row(" + std::to_string(bt) + " threads, " + std::to_string(passes_w) + " passes, " +
                                 std::to_string(kc_w) + " rows/chunk) ";
            return 1;
        };
        // ---- dense form (round 4): 256 detector pixels x 16 angles per workgroup.  When neighbouring angles of the slope order
        // are a fraction of a degree apart (BASELINE configs[3]: 1500 angles, no subsets) a 16-angle group's window is hardly
        // wider than an 8-angle group's, so every staged volume row serves twice the rays: 0.12 staged columns per sample
        // against 0.18 for whole rows x 8 angles.  Three 256-thread workgroups per CU (12 waves, <= 168 registers).
        constexpr int FP_A16 = 16;
        auto dense16_cost = [&](int c, size_t off_c, int nc) -> double {
            if (nc < 2 * FP_A16) return -1.0;
            if (s.wbound16[c] < 0)
                s.wbound16[c] = fp_window_bound(ctx->host_table.data() + s.table_offset, ctx->host_fp_order.data() + off_c, nc,
                                                ctx->n, ctx->nu, 256, FP_A16);
            if (s.wbound16[c] > 512) return -1.0;  // two column passes at most
            const int kc = s.wbound16[c] <= 256 ? 4 : 2;
            if ((size_t)2 * kc * s.wbound16[c] * 16 + (size_t)a.n * 8 > 64 * 1024) return -1.0;  // three workgroups per CU
            const long wgs = (long)ceil_div(a.nz, 4) * ceil_div(a.nu, 256) * ceil_div(nc, FP_A16);
            if (8L * ((wgs + 7) / 8) > 0x7fffffffL) return -1.0;
            // a launch per sign class must still fill the chip several times over (768 resident workgroups): on BASELINE
            // configs[1] (256^3, 360 angles) the form left half the CUs idle -- 542 instead of 702 iterations/s
            if (wgs < 4 * 768 && g_variant_fp != 3) return -1.0;
            return (double)ceil_div(a.nu, 256) * ceil_div(nc, FP_A16) * s.wbound16[c];
        };
        auto launch_dense16 = [&](int c, size_t off_c, int nc) -> int {
            FpTiledArgs t;
            t.src = (c >> 1) ? a.volT : a.vol;
            t.tab = a.tab;
            t.order = ctx->dev_fp_order + off_c;
            t.mult = nullptr;
            t.n_class = nc;
            t.nz = a.nz; t.n = a.n; t.nu = a.nu; t.na = a.na; t.na_full = a.na_full;
            t.out = out; t.b = b; t.w = w; t.fidelity = fidelity; t.gathered = gathered;
            t.ring = ring; t.ring_scale = ring_scale; t.zquad = a.zquad; t.robust = a.robust; t.rdelta = a.rdelta;
            t.wpitch = s.wbound16[c];
#if TOMO_DEV
            t.probe = g_probe;
#endif
            const int passes = ceil_div(t.wpitch, 256);   // 1 or 2
            const int kc = passes == 1 ? 4 : 2;            // four float4 staging items per thread and chunk either way
            const size_t smem = (size_t)2 * kc * t.wpitch * 16 + (size_t)a.n * 8;
            t.nut = ceil_div(a.nu, 256);
            t.bt = 256;
            t.ngroups = ceil_div(nc, FP_A16);
            t.nzb = ceil_div(a.nz, 4);
            const long blocks = 8L * (((long)t.nzb * t.nut * t.ngroups + 7) / 8);  // (size limits checked in dense16_cost)
#define FP_D16_LAUNCH(L8, RES)                                                                                         \
    do {                                                                                                               \
        if (passes == 1) fp_tiled_kernel<L8, RES, 1, 4, true, 256, FP_A16><<<(unsigned)blocks, 256, smem, st>>>(t);   \
        else fp_tiled_kernel<L8, RES, 2, 4, true, 256, FP_A16><<<(unsigned)blocks, 256, smem, st>>>(t);                \
    } while (0)
            if (b) { if (l8) FP_D16_LAUNCH(true, true); else FP_D16_LAUNCH(false, true); }
            else   { if (l8) FP_D16_LAUNCH(true, false); else FP_D16_LAUNCH(false, false); }
#undef FP_D16_LAUNCH
            if (hipGetLastError() != hipSuccess) return -1;
            ctx->last_fp_path += "class" + std::to_string(c) + ":dense(256 pixels x 16 angles, " + std::to_string(passes) + " passes) ";
            return 1;
        };
        {
            size_t axis_off = s.table_offset;
            const int nut = ceil_div(a.nu, 256);
            for (int d = 0; d < 2 && (g_variant_fp == 0 || g_variant_fp == 3 || g_variant_fp == 4) && a.nu >= FP_WIDE_MIN_NU; ++d) {
                const int nc0 = s.n_class[2 * d], nc1 = s.n_class[2 * d + 1], nc = nc0 + nc1;
                const size_t off_d = axis_off;
                axis_off += nc;
                if (nc == 0) continue;
                {
                    // dense form for both sign classes of the axis when it stages clearly less than whole rows x 8 angles would
                    // (variant 3, dev flavour: whenever it is applicable -- the A/B and parity tests of this form)
                    const double c0 = nc0 ? dense16_cost(2 * d, off_d, nc0) : 0.0, c1 = nc1 ? dense16_cost(2 * d + 1, off_d + nc0, nc1) : 0.0;
                    const int nut_w = ceil_div(a.nu, 1024);
                    const int bt_w = std::min(1024, ceil_div(ceil_div(a.nu, nut_w), 64) * 64);
                    bool take = c0 >= 0.0 && c1 >= 0.0;
#ifdef TOMO_FP_NO_DENSE16   // A/B builds only (tools/run_ab.sh)
                    take = false;
#endif
                    if (take && (g_variant_fp == 0 || g_variant_fp == 4)) {
                        double rows = 0.0;
                        if (a.nu <= 1024) {
                            if (s.wbound_wide[2 * d] < 0)
                                s.wbound_wide[2 * d] = fp_window_bound(ctx->host_table.data() + s.table_offset, ctx->host_fp_order.data() + off_d,
                                                                       nc, ctx->n, ctx->nu, bt_w);
                            rows = (double)nut_w * ceil_div(nc, FP_A) * s.wbound_wide[2 * d];
                        } else {
                            for (int h = 0; h < 2; ++h) {
                                const int nch = h ? nc1 : nc0;
                                if (nch == 0) continue;
                                int &wc = s.wbound_wide[2 * d + h];
                                if (wc < 0)
                                    wc = fp_window_bound(ctx->host_table.data() + s.table_offset,
                                                         ctx->host_fp_order.data() + off_d + (h ? nc0 : 0), nch, ctx->n, ctx->nu, bt_w);
                                rows += (double)nut_w * ceil_div(nch, FP_A) * wc;
                            }
                        }
                        take = (c0 + c1) <= FP_DENSE16_PAYS * rows;
                    }
                    if (take) {
                        const int rc0 = nc0 ? launch_dense16(2 * d, off_d, nc0) : 1;
                        const int rc1 = nc1 ? launch_dense16(2 * d + 1, off_d + nc0, nc1) : 1;
                        if (rc0 < 0 || rc1 < 0) return tomo_fail(TOMO_E_INVALID, "forward-projection launch failed (or the problem is too large for one launch)");
                        done[2 * d] = done[2 * d + 1] = true;
                        continue;
                    }
                }
                if (g_variant_fp == 3) continue;
                const double ct0 = (double)nut * ceil_div(nc0, FP_A) * std::max(s.wbound[2 * d], 0);
                const double ct1 = (double)nut * ceil_div(nc1, FP_A) * std::max(s.wbound[2 * d + 1], 0);
                if (a.nu <= 1024) {
                    // one 1024-pixel tile = the whole detector row: its window does not care about the sign of the
                    // detector slope, so the two classes of an axis (adjacent in the order table) are merged --
                    // 37 angles make 5 groups of 8 instead of 3 + 3
                    const int rc = try_wide(d, off_d, nc, s.wbound_wide[2 * d], ct0 + ct1, d ? "x:" : "y:");
                    if (rc < 0) return tomo_fail(TOMO_E_INVALID, "forward-projection launch failed (or the problem is too large for one launch)");
                    if (rc > 0) done[2 * d] = done[2 * d + 1] = true;
                } else {
                    // several 1024-pixel tiles per row (2560-wide detectors, BASELINE configs[4]): a group that mixes the
                    // two signs of the slope maps a tile to BOTH ends of the volume row (window = the whole row, 2564
                    // columns for every tile); class by class a tile's window stays ~1400 columns
                    if (nc0 > 0) {
                        const int rc = try_wide(d, off_d, nc0, s.wbound_wide[2 * d], ct0, d ? "x+:" : "y+:");
                        if (rc < 0) return tomo_fail(TOMO_E_INVALID, "forward-projection launch failed (or the problem is too large for one launch)");
                        if (rc > 0) done[2 * d] = true;
                    }
                    if (nc1 > 0) {
                        const int rc = try_wide(d, off_d + nc0, nc1, s.wbound_wide[2 * d + 1], ct1, d ? "x-:" : "y-:");
                        if (rc < 0) return tomo_fail(TOMO_E_INVALID, "forward-projection launch failed (or the problem is too large for one launch)");
                        if (rc > 0) done[2 * d + 1] = true;
                    }
                }
            }
        }
        size_t order_off = s.table_offset;
        for (int c = 0; c < 4; ++c) {  // one launch per stepping class (x-stepping classes read the transposed copy)
            const int nc = s.n_class[c];
            if (nc > 0 && !done[c]) {
                FpTiledArgs t;
                t.src = (c >> 1) ? a.volT : a.vol;
                t.tab = a.tab;
                t.order = ctx->dev_fp_order + order_off;
                t.mult = nullptr;
                t.n_class = nc;
                t.nz = a.nz; t.n = a.n; t.nu = a.nu; t.na = a.na; t.na_full = a.na_full;
                t.out = out; t.b = b; t.w = w; t.fidelity = fidelity; t.gathered = gathered;
                t.ring = ring; t.ring_scale = ring_scale; t.zquad = a.zquad; t.robust = a.robust; t.rdelta = a.rdelta;
                t.wpitch = s.wbound[c];
    #if TOMO_DEV
            t.probe = g_probe;
#endif
                const int passes = ceil_div(t.wpitch, 256);           // 1..5 in the pipelined kernel
                // (items per thread, double buffer) per pass count -- keep in step with FP_TILED_LAUNCH below
                const int fp_m = passes <= 2 ? 8 : (passes == 5 ? 10 : 12);
                const bool fp_db = passes <= 2;
                const int kc = fp_m / std::max(passes, 1);
                const size_t smem = (size_t)(fp_db ? 2 : 1) * kc * t.wpitch * 16 + (size_t)a.n * 8;
                t.nut = ceil_div(a.nu, 256);
                t.bt = 256;
                t.ngroups = ceil_div(nc, FP_A);
                t.nzb = ceil_div(a.nz, 4);
                const long blocks = 8L * (((long)t.nzb * t.nut * t.ngroups + 7) / 8);
                TOMO_REQUIRE(blocks <= 0x7fffffffL, "problem too large for one FP launch");
                // Windows of up to 1280 columns run the register-prefetch pipeline (double-buffered up to 512 columns,
                // single-buffered beyond); wider ones (detectors wider than 1024 whose whole-row form does not fit in
                // LDS) and variant 2 run the synchronous form: stage, barrier, sample, barrier, with a small LDS
                // footprint so that several workgroups per CU hide the staging latency.
                if (g_variant_fp == 2 || t.wpitch > FP_MAX_WPITCH) {
                    const int kcs = std::max(1, std::min(8, 40000 / (t.wpitch * 16)));
                    const size_t sm = (size_t)kcs * t.wpitch * 16;
                    TOMO_REQUIRE(sm <= 64 * 1024, "forward-projection window does not fit in LDS");
#define FP_SYNC_LAUNCH(L8, RES) fp_tiled_sync_kernel<L8, RES, 256, FP_A><<<(unsigned)blocks, 256, sm, st>>>(t, kcs)
                    if (b) { if (l8) FP_SYNC_LAUNCH(true, true); else FP_SYNC_LAUNCH(false, true); }
                    else   { if (l8) FP_SYNC_LAUNCH(true, false); else FP_SYNC_LAUNCH(false, false); }
#undef FP_SYNC_LAUNCH
                    TOMO_LAUNCH_CHECK();
                    ctx->last_fp_path += "class" + std::to_string(c) + ":tiled-sync(window " + std::to_string(t.wpitch) + ") ";
                    order_off += nc;
                    continue;
                }
#define FP_TILED_LAUNCH(L8, RES)                                                                          \
    do {                                                                                                  \
        if (passes == 1) fp_tiled_kernel<L8, RES, 1, 8, true, 256><<<(unsigned)blocks, 256, smem, st>>>(t);        \
        else if (passes == 2) fp_tiled_kernel<L8, RES, 2, 8, true, 256><<<(unsigned)blocks, 256, smem, st>>>(t);   \
        else if (passes == 3) fp_tiled_kernel<L8, RES, 3, 12, false, 256><<<(unsigned)blocks, 256, smem, st>>>(t); \
        else if (passes == 4) fp_tiled_kernel<L8, RES, 4, 12, false, 256><<<(unsigned)blocks, 256, smem, st>>>(t); \
        else fp_tiled_kernel<L8, RES, 5, 10, false, 256><<<(unsigned)blocks, 256, smem, st>>>(t);                  \
    } while (0)
                if (b) { if (l8) FP_TILED_LAUNCH(true, true); else FP_TILED_LAUNCH(false, true); }
                else   { if (l8) FP_TILED_LAUNCH(true, false); else FP_TILED_LAUNCH(false, false); }
#undef FP_TILED_LAUNCH
                TOMO_LAUNCH_CHECK();
                ctx->last_fp_path += "class" + std::to_string(c) + ":tiled-pipelined(256 threads, " + std::to_string(passes) +
                                     " passes) ";
            }
            order_off += nc;
        }
        return TOMO_OK;
    }
    ctx->last_fp_path = "march(no LDS)";
    if (g_variant_fp != 1) tomo_warn_once("fp_march", "forward projection: a staged row window exceeds 64 KiB of LDS, "
                                          "falling back to the un-tiled march kernel (about 4x slower)");
    if (b) {
        if (l8) fp_march_kernel<true, true><<<grid, 256, 0, st>>>(a);
        else fp_march_kernel<false, true><<<grid, 256, 0, st>>>(a);
    } else {
        if (l8) fp_march_kernel<true, false><<<grid, 256, 0, st>>>(a);
        else fp_march_kernel<false, false><<<grid, 256, 0, st>>>(a);
    }
    TOMO_LAUNCH_CHECK();
    return TOMO_OK;
}

int bp_prepare(tomo_ctx *ctx, int subset, const float *sino, BpArgs &a)
{
    TOMO_REQUIRE(ctx != nullptr, "ctx is NULL");
    TOMO_REQUIRE(subset < ctx->os, "subset %d out of range (OS_number %d)", subset, ctx->os);
    const tomo_subset &s = (subset < 0 || ctx->os == 1) ? ctx->subsets[0] : ctx->subsets[1 + subset];
    TOMO_REQUIRE(sino != nullptr || s.size == 0, "NULL sinogram pointer");
    a = BpArgs();
    a.device = ctx->device;
    a.path = &ctx->last_bp_path;
    a.sino = sino;
    a.tab = ctx->dev_table + s.table_offset;
    a.nz = ctx->nz; a.n = ctx->n; a.nu = ctx->nu; a.na = s.size;
    return TOMO_OK;
}

}  // namespace

void tomo_bp_relay_reset(int device)
{
    if (device < 0 || device >= BP_RELAY_DEVICES) return;
    g_bp_relay[device].refused.store(~(size_t)0);
    g_bp_relay[device].skipped.store(0);
}

// ------------------------------------------------------------------------------------------ C-ABI
extern "C" int tomo_fp3d(tomo_ctx *ctx, int subset, const float *vol_dev, float *sino_dev, void *stream)
{
    return fp_run(ctx, subset, vol_dev, nullptr, nullptr, 0, TOMO_FID_LS, sino_dev, stream);
}

extern "C" int tomo_fp3d_residual(tomo_ctx *ctx, int subset, const float *vol_dev, const float *b_full_dev,
                                  const float *w_full_dev, int gathered, int fidelity, float *res_dev, void *stream)
{
    TOMO_REQUIRE(b_full_dev != nullptr, "projection data pointer is NULL");
    TOMO_REQUIRE(fidelity == TOMO_FID_LS || fidelity == TOMO_FID_PWLS || fidelity == TOMO_FID_KL ||
                     fidelity == TOMO_FID_RATIO,
                 "data_fidelity should be LS, PWLS or KL");
    TOMO_REQUIRE(fidelity != TOMO_FID_PWLS || w_full_dev != nullptr, "PWLS needs the weights");
    return fp_run(ctx, subset, vol_dev, b_full_dev, fidelity == TOMO_FID_PWLS ? w_full_dev : nullptr, gathered,
                  fidelity, res_dev, stream, nullptr, 0.0f, ctx ? ctx->res_layout : 0);
}

extern "C" int tomo_fp3d_residual_robust(tomo_ctx *ctx, int subset, const float *vol_dev, const float *b_full_dev,
                                         const float *w_full_dev, int gathered, int fidelity, int robust, float delta,
                                         float *res_dev, void *stream)
{
    TOMO_REQUIRE(b_full_dev != nullptr, "projection data pointer is NULL");
    TOMO_REQUIRE(fidelity == TOMO_FID_LS || fidelity == TOMO_FID_PWLS, "the Huber / Student's-t re-weighting applies to the LS and PWLS residuals");
    TOMO_REQUIRE(fidelity != TOMO_FID_PWLS || w_full_dev != nullptr, "PWLS needs the weights");
    TOMO_REQUIRE(robust == TOMO_ROBUST_NONE || robust == TOMO_ROBUST_HUBER || robust == TOMO_ROBUST_STUDENTST, "unknown robust mode %d", robust);
    TOMO_REQUIRE(robust == TOMO_ROBUST_NONE || delta > 0.0f, "the Huber / Student's-t threshold must be positive");
    return fp_run(ctx, subset, vol_dev, b_full_dev, fidelity == TOMO_FID_PWLS ? w_full_dev : nullptr, gathered,
                  fidelity, res_dev, stream, nullptr, 0.0f, ctx ? ctx->res_layout : 0, robust, delta);
}

extern "C" int tomo_ctx_set_residual_layout(tomo_ctx *ctx, int layout)
{
    TOMO_REQUIRE(ctx != nullptr, "ctx is NULL");
    TOMO_REQUIRE(layout == TOMO_RESIDUAL_PLANAR || layout == TOMO_RESIDUAL_ZQUAD, "unknown residual layout %d", layout);
    ctx->res_layout = layout;
    return TOMO_OK;
}

extern "C" int tomo_ctx_residual_layout(const tomo_ctx *ctx) { return ctx ? ctx->res_layout : -1; }

extern "C" size_t tomo_ctx_residual_elems(const tomo_ctx *ctx, int subset)
{
    if (!ctx || subset >= ctx->os) return 0;
    const tomo_subset &s = (subset < 0 || ctx->os == 1) ? ctx->subsets[0] : ctx->subsets[1 + subset];
    const size_t nzr = ctx->res_layout == TOMO_RESIDUAL_ZQUAD ? (size_t)ceil_div(ctx->nz, 4) * 4 : (size_t)ctx->nz;
    return nzr * (size_t)s.size * (size_t)ctx->nu;
}

extern "C" int tomo_momentum_transposed(tomo_ctx *ctx, const float *x_dev, const float *xold_dev, float *xt_dev, float beta,
                                        void *stream)
{
    TOMO_REQUIRE(ctx != nullptr && x_dev && xold_dev && xt_dev, "NULL argument");
    TOMO_ON_DEVICE(ctx->device);
    int rc = fp_scratch(ctx);
    if (rc != TOMO_OK) return rc;
    if (ctx->n % 4 == 0 && aligned16(x_dev, xold_dev, xt_dev, ctx->scratch)) {
        dim3 tg(ceil_div(ctx->n, 64), ceil_div(ctx->n, 64), ctx->nz);
        transpose64_kernel<true><<<tg, 256, 0, as_stream(stream)>>>(x_dev, xold_dev, xt_dev, (float *)ctx->scratch, beta, ctx->n);
    } else {
        dim3 tg(ceil_div(ctx->n, 32), ceil_div(ctx->n, 32), ctx->nz);
        momentum_transpose_kernel<<<tg, 256, 0, as_stream(stream)>>>(x_dev, xold_dev, xt_dev, (float *)ctx->scratch, beta, ctx->n);
    }
    TOMO_LAUNCH_CHECK();
    ctx->volT_of = xt_dev;
    ctx->volT_stream = as_stream(stream);
    ctx->volT_valid = true;
    return TOMO_OK;
}

extern "C" int tomo_ctx_invalidate(tomo_ctx *ctx)
{
    TOMO_REQUIRE(ctx != nullptr, "ctx is NULL");
    ctx->volT_valid = false;
    ctx->volT_of = nullptr;
    return TOMO_OK;
}

extern "C" int tomo_fp3d_residual_ring(tomo_ctx *ctx, int subset, const float *vol_dev, const float *b_full_dev,
                                       const float *ring_dev, float ring_scale, float *res_dev, void *stream)
{
    TOMO_REQUIRE(b_full_dev != nullptr && ring_dev != nullptr, "projection data / ring offset pointer is NULL");
    TOMO_REQUIRE(ctx == nullptr || ctx->res_layout == TOMO_RESIDUAL_PLANAR,
                 "the ring-term residual is read by tomo_ring_gh_reduce in the planar layout: set TOMO_RESIDUAL_PLANAR first");
    return fp_run(ctx, subset, vol_dev, b_full_dev, nullptr, 0, TOMO_FID_LS, res_dev, stream, ring_dev, ring_scale);
}

extern "C" int tomo_bp3d(tomo_ctx *ctx, int subset, const float *sino_dev, float *vol_dev, void *stream)
{
    BpArgs a;
    int rc = bp_prepare(ctx, subset, sino_dev, a);
    if (rc != TOMO_OK) return rc;
    TOMO_REQUIRE(vol_dev != nullptr, "NULL volume pointer");
    a.vol = vol_dev;
    return bp_launch<EPI_PLAIN>(a, (ctx->flags & TOMO_FLAG_LERP8) != 0, as_stream(stream));
}

extern "C" int tomo_bp3d_fista(tomo_ctx *ctx, int subset, const float *res_dev, const float *xt_dev,
                               float *xout_dev, float l_inv, int nonneg, void *stream)
{
    BpArgs a;
    int rc = bp_prepare(ctx, subset, res_dev, a);
    if (rc != TOMO_OK) return rc;
    TOMO_REQUIRE(xt_dev && xout_dev, "NULL volume pointer");
    a.xt = xt_dev; a.xout = xout_dev; a.s0 = l_inv; a.nonneg = nonneg;
    a.zquad = ctx->res_layout == TOMO_RESIDUAL_ZQUAD;
    return bp_launch<EPI_FISTA>(a, (ctx->flags & TOMO_FLAG_LERP8) != 0, as_stream(stream));
}

extern "C" int tomo_bp3d_fista_momentum(tomo_ctx *ctx, int subset, const float *res_dev, float *xt_dev,
                                        float *xold_x_dev, float l_inv, float beta, int nonneg, void *stream)
{
    BpArgs a;
    int rc = bp_prepare(ctx, subset, res_dev, a);
    if (rc != TOMO_OK) return rc;
    TOMO_REQUIRE(xt_dev && xold_x_dev, "NULL volume pointer");
    a.xt = xt_dev; a.xt_out = xt_dev; a.xold = xold_x_dev; a.xout = xold_x_dev;
    a.s0 = l_inv; a.s1 = beta; a.nonneg = nonneg;
    a.zquad = ctx->res_layout == TOMO_RESIDUAL_ZQUAD;
    return bp_launch<EPI_FISTA_MOM>(a, (ctx->flags & TOMO_FLAG_LERP8) != 0, as_stream(stream));
}

extern "C" int tomo_bp3d_admm(tomo_ctx *ctx, int subset, const float *res_dev, float *z_dev, const float *x_dev,
                              const float *u_dev, float *zu_out_dev, float tau, float rho, int relax_on,
                              float one_minus_alpha, float alpha, int nonneg, void *stream)
{
    BpArgs a;
    int rc = bp_prepare(ctx, subset, res_dev, a);
    if (rc != TOMO_OK) return rc;
    TOMO_REQUIRE(z_dev && x_dev && u_dev && zu_out_dev, "NULL volume pointer");
    a.xt = x_dev; a.xout = z_dev; a.xt_out = zu_out_dev; a.xold = u_dev;
    a.s0 = tau; a.s1 = rho; a.s2 = one_minus_alpha; a.s3 = alpha;
    a.nonneg = nonneg; a.relax_on = relax_on;
    a.zquad = ctx->res_layout == TOMO_RESIDUAL_ZQUAD;
    return bp_launch<EPI_ADMM>(a, (ctx->flags & TOMO_FLAG_LERP8) != 0, as_stream(stream));
}
