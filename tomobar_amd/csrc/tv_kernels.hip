// TV proximal operators for gfx950: Chambolle-Pock primal-dual TV (PD_TV) and explicit ROF TV.
//
// What is computed is fixed by the reference kernels
//   tomobar/cuda_kernels/primal_dual_for_total_variation.cu:125-261 (3D), :360-452 (2D)
//   tomobar/cuda_kernels/rudin_osher_fatemi_total_variation.cu:66-137 (2D), :156-238 (3D)
// How it is computed here is MI355X-first:
//   * PD_TV: pd_zmarch_xk.inl, THREE iterations per pass through HBM.  A lane owns 8 consecutive rows of one x column and
//     marches along z; +-y neighbours are other registers of the same lane, +-x neighbours come from DPP wave shifts (3
//     halo lanes either side), the z-1 duals are carried, stage s (iteration n+s -> n+s+1) runs s planes behind stage 0
//     in the same wave and the hand-over state lives in private LDS slots.  Two iterations per pass (launch remainders):
//     pd_zmarch_x2.inl; one iteration (tails, 2D, thin volumes): pd_zmarch2.inl; 2D images: pd_rows2d.inl (three
//     iterations per pass, rows in registers).  Measured history and PMC evidence: docs/kernels/pd_tv.md.
//   * ROF_TV: rof_zmarch.inl, divergence and update fused on the same z-march skeleton: the D fields never reach
//     HBM (12 B/voxel/iteration) and are evaluated once per voxel.
//   * -DTOMO_DEV_VARIANTS (libtomo_mi355x_dev.so, tests / tools only): the per-voxel kernels (variant 1: one thread per
//     voxel, neighbours' duals recomputed from global memory -- the independent implementation) and the builds with the
//     compiler's IEEE sqrt / divide (2, 21) or relaxed ROF arithmetic.
// All arithmetic is float32 (explicit fmaf, -ffp-contract=off).  What the SHIPPED defaults reproduce bit for bit: ROF_TV
// (float32 and binary16 D fields) and PD_TV with binary16 duals follow the rounding sequence of oracle/tomo_oracle.c.
// PD_TV with float32 duals -- the kernel bench.py times -- ships RELAXED arithmetic (pd_default_is_exact<float>() is false:
// v_rsq_f32, a hoisted reciprocal): within 1e-5 relative L2 of the reference, NOT bit-identical; tomo_set_variant("pdtv", 22)
// (`_regularisation_["exact_roundings"] = True` through the classes) selects the reference's roundings there too, at about
// +10 % per launch.  docs/kernels/pd_tv.md has the measurements behind that choice.
#include "tomo_common.h"
#include <algorithm>
#include <utility>

namespace {

// ------------------------------------------------------------------------------------------ helpers
// One-lane wave shifts.  The z-march kernels only ever exchange with the neighbouring lane; __shfl_up/down(v, 1, 64)
// lower to ds_bpermute_b32 (an LDS-pipe instruction with ~50+ cycles latency), the gfx9 DPP wave shifts are a single
// VALU move.  Lane 0 (wave_prev) / lane 63 (wave_next) receive 0; those lanes are halo lanes whose shifted-in value is
// never consumed.
__device__ __forceinline__ float wave_prev(float v)  // lane i <- lane i-1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138 /*wave_shr:1*/, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_next(float v)  // lane i <- lane i+1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /*wave_shl:1*/, 0xf, 0xf, true));
}
#ifndef TOMO_TV_NO_DPP
#define __shfl_up(v, d, w) wave_prev(v)
#define __shfl_down(v, d, w) wave_next(v)
#endif
template <typename T> struct DualIO;
template <> struct DualIO<float> {
    static __device__ __forceinline__ float ld(const float *p, size_t i) { return p[i]; }
    static __device__ __forceinline__ void st(float *p, size_t i, float v) { p[i] = v; }
    static __device__ __forceinline__ float rt(float v) { return v; }
};
// binary16 rounding of a float32 RESULT: the reference rounds twice (the operation to float32, then the store / conversion to
// binary16).  hipcc folds "fmaf(...) then float->half" into v_fma_mixlo_f16, which rounds the exact fused result ONCE --
// a different value in the double-rounding cases (found in round 3: ROF_TV with binary16 D fields, one voxel in 512 after
// 20 iterations).  The empty asm makes the float32 value opaque, so the conversion always starts from the rounded float.
__device__ __forceinline__ __half f32_to_half_twice_rounded(float v)
{
    asm volatile("" : "+v"(v));
    return __float2half_rn(v);
}
// two such conversions in one v_cvt_pk_f16_f32 (the pair travels as one 32-bit word), and back
typedef float tv_v2f32 __attribute__((ext_vector_type(2)));
typedef _Float16 tv_v2f16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float pack_half2_twice_rounded(float a, float b)
{
    asm volatile("" : "+v"(a), "+v"(b));
    const tv_v2f32 v = {a, b};
    const tv_v2f16 h = __builtin_convertvector(v, tv_v2f16);
    return __builtin_bit_cast(float, h);
}
__device__ __forceinline__ void unpack_half2(float w, float &a, float &b)
{
    const tv_v2f32 v = __builtin_convertvector(__builtin_bit_cast(tv_v2f16, w), tv_v2f32);
    const float x = v.x, y = v.y;
    a = x; b = y;
}
template <> struct DualIO<__half> {
    static __device__ __forceinline__ float ld(const __half *p, size_t i) { return __half2float(p[i]); }
    static __device__ __forceinline__ void st(__half *p, size_t i, float v) { p[i] = f32_to_half_twice_rounded(v); }
    static __device__ __forceinline__ float rt(float v) { return __half2float(f32_to_half_twice_rounded(v)); }
};

// dual ascent + projection onto the unit ball (isotropic) / unit cube (anisotropic)

// Plane-relative buffer addressing for the z-march kernels.  A lane keeps 32-bit BYTE offsets (float arrays) of its rows
// inside one plane; the plane's base moves with scalar instructions.  With flat pointers the compiler widens every
// offset to a 64-bit register pair and spends one 64-bit VALU add per access (143 of ~1100 VALU instructions per step and
// ~60 registers in the three-iteration PD_TV kernel); `buffer_load/store ... offen` takes the 32-bit lane offset as is.
// One descriptor per (array, plane): base = plane start, num_records = plane bytes (a whole volume may exceed the 4 GiB
// a 32-bit offset can reach).
struct PlaneIO {
    int bytes;  // size of one float plane in bytes
    __device__ __forceinline__ __amdgpu_buffer_rsrc_t rs(const void *plane, int nbytes) const
    {
        return __builtin_amdgcn_make_buffer_rsrc((void *)plane, 0, nbytes, 0x00020000);
    }
    __device__ __forceinline__ float ldf(const float *plane, unsigned boff) const
    {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs(plane, bytes), (int)boff, 0, 0));
    }
    __device__ __forceinline__ void stf(float *plane, unsigned boff, float v) const
    {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rs(plane, bytes), (int)boff, 0, 0);
    }
    // (lane column offset, wave-uniform row offset) forms: `xo` = FLOAT byte offset of the lane's column inside a row
    // (VGPR), `ro` = FLOAT byte offset of the row inside the plane (SGPR, goes into the instruction's soffset).  The
    // range check of a raw buffer covers the lane offset only, so `ro + xo` must lie inside the plane.
    __device__ __forceinline__ float ldf(const float *plane, unsigned xo, int ro) const
    {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs(plane, bytes), (int)xo, ro, 0));
    }
    __device__ __forceinline__ void stf(float *plane, unsigned xo, int ro, float v) const
    {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rs(plane, bytes), (int)xo, ro, 0);
    }
    __device__ __forceinline__ float ldd(const float *plane, unsigned xo, int ro) const { return ldf(plane, xo, ro); }
    __device__ __forceinline__ void std_(float *plane, unsigned xo, int ro, float v) const { stf(plane, xo, ro, v); }
    __device__ __forceinline__ float ldd(const __half *plane, unsigned xo, int ro) const
    {
        const unsigned short h = __builtin_amdgcn_raw_buffer_load_b16(rs(plane, bytes >> 1), (int)(xo >> 1), ro >> 1, 0);
        return __half2float(__builtin_bit_cast(__half, h));
    }
    __device__ __forceinline__ void std_(__half *plane, unsigned xo, int ro, float v) const
    {
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, f32_to_half_twice_rounded(v)), rs(plane, bytes >> 1),
                                              (int)(xo >> 1), ro >> 1, 0);
    }
    // stores through a descriptor built once per plane (a store inside a guarded row loop otherwise rebuilds it: three
    // scalar instructions per store)
    __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsf(const float *plane) const { return rs(plane, bytes); }
    __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsd(const float *plane) const { return rs(plane, bytes); }
    __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsd(const __half *plane) const { return rs(plane, bytes >> 1); }
    static __device__ __forceinline__ void stf_rs(__amdgpu_buffer_rsrc_t r, unsigned xo, int ro, float v)
    {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, (int)xo, ro, 0);
    }
    static __device__ __forceinline__ void std_rs(const float *, __amdgpu_buffer_rsrc_t r, unsigned xo, int ro, float v) { stf_rs(r, xo, ro, v); }
    static __device__ __forceinline__ void std_rs(const __half *, __amdgpu_buffer_rsrc_t r, unsigned xo, int ro, float v)
    {
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, f32_to_half_twice_rounded(v)), r, (int)(xo >> 1), ro >> 1, 0);
    }
    // dual fields: float or binary16 (boff is always the FLOAT byte offset of the voxel)
    __device__ __forceinline__ float ldd(const float *plane, unsigned boff) const { return ldf(plane, boff); }
    __device__ __forceinline__ void std_(float *plane, unsigned boff, float v) const { stf(plane, boff, v); }
    __device__ __forceinline__ float ldd(const __half *plane, unsigned boff) const
    {
        const unsigned short h = __builtin_amdgcn_raw_buffer_load_b16(rs(plane, bytes >> 1), (int)(boff >> 1), 0, 0);
        return __half2float(__builtin_bit_cast(__half, h));
    }
    __device__ __forceinline__ void std_(__half *plane, unsigned boff, float v) const
    {
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, f32_to_half_twice_rounded(v)), rs(plane, bytes >> 1),
                                              (int)(boff >> 1), 0, 0);
    }
};

template <int ND, bool ANISO>
__device__ __forceinline__ void pd_dual(float (&p)[3], const float (&g)[3], float sigma)
{
#pragma unroll
    for (int c = 0; c < ND; ++c) p[c] = fmaf(sigma, g[c], p[c]);
    if (!ANISO) {
        float nrm = p[0] * p[0];
#pragma unroll
        for (int c = 1; c < ND; ++c) nrm = fmaf(p[c], p[c], nrm);
        if (nrm > 1.0f) {
            float r = 1.0f / sqrtf(nrm);
#pragma unroll
            for (int c = 0; c < ND; ++c) p[c] *= r;
        }
    } else {
#pragma unroll
        for (int c = 0; c < ND; ++c) {
            float v = fabsf(p[c]);
            v = v < 1.0f ? 1.0f : v;
            p[c] /= v;
        }
    }
}

__device__ __forceinline__ float pd_primal(float u_in, float input, float div, float tau, float lt, float theta,
                                           bool nonneg)
{
    float u = (nonneg && u_in < 0.0f) ? 0.0f : u_in;
    float t = fmaf(-tau, div, u);
    t = fmaf(lt, input, t);
    float nu = t / (1.0f + lt);
    return fmaf(theta, nu - u, nu);
}

struct PdArgs {
    const float *in;
    const float *u_in;
    float *u_out;
    const void *p_in[3];
    void *p_out[3];
    int dx, dy;
    int planes;       // planes addressed by the pointers (ghost planes included)
    int out_begin;    // first plane that is written
    int out_end;      // one past the last plane that is written
    int first_is_edge;  // plane 0 is the global z = 0 plane
    int last_is_edge;   // plane planes-1 is the global last plane
    float sigma, tau, lt, theta;
    int zchunk;       // planes per z-chunk (zmarch)
    float inv1lt;     // 1 / (1 + lt), relaxed-arithmetic kernels only
    int p_in_zero = 0;   // pd_zmarch_xk: the input duals are all zero (first launch of a prox): do not read them
    int p_out_skip = 0;  // pd_zmarch_xk: do not store the output duals (last launch of a prox)
    float nn_thr = 0.0f; // pd_zmarch_xk, relaxed float32 build: iterates below this are clipped to 0 (0 = nonnegativity, -inf = none)
#if TOMO_DEV
    int probe = 0;       // measurement only (tools/archive/probes/pd_halo_probe.py): 1 = alias the y halo rows, 2 = the x halo lanes onto the workgroup's own tile, 4 = every plane access goes to plane 0 (cache-resident: what the kernel costs without HBM)
#else
    static constexpr int probe = 0;  // the shipped flavour carries no measurement switches
#endif
};

#if TOMO_DEV
// ------------------------------------------------------------------------------------------ PD variant 1
// forward difference with the far-edge mirror (primal_dual...cu:216-220) and zero "previous" at index 0 (:147-160)
__device__ __forceinline__ float fwd_diff(const float *U, size_t idx, int i, int dim, size_t stride, bool edge_last)
{
    float u = U[idx];
    float nxt;
    if (i == dim - 1 && edge_last) nxt = (i > 0) ? U[idx - stride] : 0.0f;
    else nxt = U[idx + stride];
    return nxt - u;
}

template <typename T, int ND, bool ANISO>
__device__ __forceinline__ void pd_dual_at(const PdArgs &a, int x, int y, int z, float (&p)[3])
{
    const size_t sy = (size_t)a.dx, sz = (size_t)a.dx * a.dy;
    const size_t idx = (size_t)x + sy * y + sz * z;
    float g[3] = {0.0f, 0.0f, 0.0f};
    g[0] = fwd_diff(a.u_in, idx, x, a.dx, 1, true);
    g[1] = fwd_diff(a.u_in, idx, y, a.dy, sy, true);
    if (ND == 3) g[2] = fwd_diff(a.u_in, idx, z, a.planes, sz, a.last_is_edge != 0);
#pragma unroll
    for (int c = 0; c < ND; ++c) p[c] = DualIO<T>::ld((const T *)a.p_in[c], idx);
    pd_dual<ND, ANISO>(p, g, a.sigma);
}

template <typename T, int ND, bool NONNEG, bool ANISO>
__global__ __launch_bounds__(256) void pd_pervoxel_kernel(PdArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int z = a.out_begin + blockIdx.z;
    if (x >= a.dx) return;
    const size_t sy = (size_t)a.dx, sz = (size_t)a.dx * a.dy;
    const size_t idx = (size_t)x + sy * y + sz * z;
    float p[3] = {0, 0, 0}, q[3] = {0, 0, 0};
    pd_dual_at<T, ND, ANISO>(a, x, y, z, p);
    float px = 0.0f, py = 0.0f, pz = 0.0f;
    if (x > 0) { pd_dual_at<T, ND, ANISO>(a, x - 1, y, z, q); px = q[0]; }
    if (y > 0) { pd_dual_at<T, ND, ANISO>(a, x, y - 1, z, q); py = q[1]; }
    if (ND == 3 && z > 0) { pd_dual_at<T, ND, ANISO>(a, x, y, z - 1, q); pz = q[2]; }
    float div = (-(p[0] - px)) + (-(p[1] - py));
    if (ND == 3) div = div + (-(p[2] - pz));
    a.u_out[idx] = pd_primal(a.u_in[idx], a.in[idx], div, a.tau, a.lt, a.theta, NONNEG);
#pragma unroll
    for (int c = 0; c < ND; ++c) DualIO<T>::st((T *)a.p_out[c], idx, p[c]);
}

#endif  // TOMO_DEV

// FAST = 0: arithmetic and rounding of the reference through the compiler's IEEE sqrt / divide (bit-identical to the oracle).
// FAST = 1: 1/(1+lt) hoisted to the host, v_rsq_f32 / v_rcp_f32 instead of IEEE sqrt + divide (<= 1e-6 relative).
// FAST = 2: the reference's roundings reproduced with FMA correction steps (Markstein; see rof_eval in rof_zmarch.inl and
//           tools/probes/markstein_probe.hip): sqrt = v_rsq + two coupled Newton steps + exact-residual correction,
//           1/q and t/(1+lt) = reciprocal + exact-residual correction.  Bit-identical to FAST = 0 for normal operands
//           (the residual underflows below ~1e-31), at about half its instruction count.
__device__ __forceinline__ float mk_sqrt(float x)
{
    const float r = __builtin_amdgcn_rsqf(x);
    float q = x * r, h = 0.5f * r;
    const float e = fmaf(-h, q, 0.5f);
    q = fmaf(q, e, q);
    h = fmaf(h, e, h);
    return fmaf(fmaf(-q, q, x), h, q);
}
// 1.0f / q correctly rounded: v_rcp_f32 (1 ulp) + ONE Newton step.  On gfx950 that equals the IEEE quotient for every one of
// the 1 677 721 601 floats in [2^-100, 2^100] (tools/probes/recip_allones_probe.hip, profiles/archive/r4d_recip_allones_probe.txt;
// round 3 used two steps).  The textbook counter-example q = 0x1.fffffep+k -- where a v_rcp that returned the 1-ulp-low
// 2^-(k+1) would leave the Newton update on an exact tie -- does not occur: this hardware's v_rcp_f32 returns RN(1/q)
// there in every binade.
__device__ __forceinline__ float mk_recip(float q)
{
    const float y = __builtin_amdgcn_rcpf(q);
    return fmaf(fmaf(-q, y, 1.0f), y, y);
}

template <bool ANISO, int FAST, int ND = 3>
__device__ __forceinline__ void pd_dual_t(float (&p)[3], const float (&g)[3], float sigma)
{
    if (FAST == 0) {
        pd_dual<ND, ANISO>(p, g, sigma);
        return;
    }
#pragma unroll
    for (int c = 0; c < ND; ++c) p[c] = fmaf(sigma, g[c], p[c]);
    if (!ANISO) {
        float nrm = p[0] * p[0];
#pragma unroll
        for (int c = 1; c < ND; ++c) nrm = fmaf(p[c], p[c], nrm);
        if (FAST == 2) {
            // the reference's branch (primal_dual...cu:196-203): if (nrm > 1) p *= 1 / sqrtf(nrm), two roundings.  Branch-free
            // (p * 1.0f is p exactly): the correction steps of neighbouring rows then pack into v_pk_fma_f32, and a wave
            // with one lane over the threshold paid for the whole sequence anyway.
            const float r = mk_recip(mk_sqrt(nrm));
            const float rr = nrm > 1.0f ? r : 1.0f;
#pragma unroll
            for (int c = 0; c < ND; ++c) p[c] *= rr;
        } else {
            const float r = nrm > 1.0f ? __builtin_amdgcn_rsqf(nrm) : 1.0f;
#pragma unroll
            for (int c = 0; c < ND; ++c) p[c] *= r;
        }
    } else {
#pragma unroll
        for (int c = 0; c < ND; ++c)
            p[c] = fabsf(p[c]) > 1.0f ? copysignf(1.0f, p[c]) : p[c];  // p / |p| is exactly +-1 in IEEE arithmetic too
    }
}

// The dual update of NB independent rows, written phase by phase across the rows (FAST = 2 only; the other levels just
// loop).  A dependent v_pk_*_f32 needs one wait state after its producer; row by row the correction chain of a row pair
// is eleven dependent packed operations and the compiler fills every gap with an s_nop (147 per step of the K = 3
// kernel).  With the chains of two row pairs interleaved in program order the other pair's operation sits in the gap.
// Same operations on the same operands as pd_dual_t: identical results.
template <bool ANISO, int FAST, int NB>
__device__ __forceinline__ void pd_dual_block(float (&p)[NB][3], const float (&g)[NB][3], float sigma, int n)
{
    if constexpr (FAST != 2 || ANISO) {
#pragma unroll
        for (int k = 0; k < NB; ++k)
            if (k < n) pd_dual_t<ANISO, FAST>(p[k], g[k], sigma);
    } else {
        float nrm[NB], q[NB], h[NB], e[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) p[k][c] = fmaf(sigma, g[k][c], p[k][c]);
#pragma unroll
        for (int k = 0; k < NB; ++k) nrm[k] = p[k][0] * p[k][0];
#pragma unroll
        for (int k = 0; k < NB; ++k) nrm[k] = fmaf(p[k][1], p[k][1], nrm[k]);
#pragma unroll
        for (int k = 0; k < NB; ++k) nrm[k] = fmaf(p[k][2], p[k][2], nrm[k]);
        // mk_sqrt
#pragma unroll
        for (int k = 0; k < NB; ++k) e[k] = __builtin_amdgcn_rsqf(nrm[k]);
#pragma unroll
        for (int k = 0; k < NB; ++k) { q[k] = nrm[k] * e[k]; h[k] = 0.5f * e[k]; }
#pragma unroll
        for (int k = 0; k < NB; ++k) e[k] = fmaf(-h[k], q[k], 0.5f);
#pragma unroll
        for (int k = 0; k < NB; ++k) { q[k] = fmaf(q[k], e[k], q[k]); h[k] = fmaf(h[k], e[k], h[k]); }
#pragma unroll
        for (int k = 0; k < NB; ++k) e[k] = fmaf(-q[k], q[k], nrm[k]);
#pragma unroll
        for (int k = 0; k < NB; ++k) q[k] = fmaf(e[k], h[k], q[k]);   // = sqrtf(nrm)
        // mk_recip
#pragma unroll
        for (int k = 0; k < NB; ++k) h[k] = __builtin_amdgcn_rcpf(q[k]);
#pragma unroll
        for (int k = 0; k < NB; ++k) e[k] = fmaf(-q[k], h[k], 1.0f);
#pragma unroll
        for (int k = 0; k < NB; ++k) h[k] = fmaf(e[k], h[k], h[k]);   // = 1.0f / sqrtf(nrm) (one Newton step: see mk_recip)
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const float rr = nrm[k] > 1.0f ? h[k] : 1.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) p[k][c] *= rr;
        }
    }
}

template <int FAST>
__device__ __forceinline__ float pd_primal_t(float u_in, float input, float div, float tau, float lt, float inv1lt,
                                             float theta, bool nonneg, float nn_thr = 0.0f)
{
    if (FAST == 0) return pd_primal(u_in, input, div, tau, lt, theta, nonneg);
    const float u = (nonneg && u_in < nn_thr) ? 0.0f : u_in;
    float t = fmaf(-tau, div, u);
    t = fmaf(lt, input, t);
    float nu = t * inv1lt;
    if (FAST == 2) nu = fmaf(fmaf(-(1.0f + lt), nu, t), inv1lt, nu);  // = t / (1 + lt) correctly rounded (inv1lt = RN(1/(1+lt)))
    return fmaf(theta, nu - u, nu);
}

// the primal update of NB independent rows, phase by phase (see pd_dual_block); same operations as pd_primal_t
template <int FAST, int NB>
__device__ __forceinline__ void pd_primal_block(float (&out)[NB], const float (&u_in)[NB], const float (&input)[NB],
                                                const float (&div)[NB], float tau, float lt, float inv1lt, float theta,
                                                bool nonneg, float nn_thr = 0.0f)
{
    if constexpr (FAST == 0) {
#pragma unroll
        for (int k = 0; k < NB; ++k) out[k] = pd_primal(u_in[k], input[k], div[k], tau, lt, theta, nonneg);
    } else {
        float u[NB], t[NB], nu[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) u[k] = (nonneg && u_in[k] < nn_thr) ? 0.0f : u_in[k];
#pragma unroll
        for (int k = 0; k < NB; ++k) t[k] = fmaf(-tau, div[k], u[k]);
#pragma unroll
        for (int k = 0; k < NB; ++k) t[k] = fmaf(lt, input[k], t[k]);
#pragma unroll
        for (int k = 0; k < NB; ++k) nu[k] = t[k] * inv1lt;
        if constexpr (FAST == 2) {  // = t / (1 + lt) correctly rounded (inv1lt = RN(1/(1+lt)))
#pragma unroll
            for (int k = 0; k < NB; ++k) t[k] = fmaf(-(1.0f + lt), nu[k], t[k]);
#pragma unroll
            for (int k = 0; k < NB; ++k) nu[k] = fmaf(t[k], inv1lt, nu[k]);
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) u[k] = nu[k] - u[k];
#pragma unroll
        for (int k = 0; k < NB; ++k) out[k] = fmaf(theta, u[k], nu[k]);
    }
}

#include "pd_zmarch2.inl"
#include "pd_zmarch_x2.inl"
#include "pd_zmarch_xk.inl"
#include "pd_rows2d.inl"

// Several iterations in one pass through HBM (3D).  `k` = iterations of this launch (2 or 3).
//   variant 0 (shipped default): float32 duals with relaxed arithmetic (FAST = 1: v_rsq_f32 instead of 1 / sqrtf, a
//              host-computed 1 / (1 + lt) instead of the divide; <= 1e-5 from the reference, typically 3e-7), binary16 duals
//              with the reference's roundings (FAST = 2; one flipped binary16 rounding is 5e-4 of a dual value, relaxed
//              arithmetic cannot hold the 1e-5 parity bar there).  k = 3 -> pd_zmarch_xk<K=3, 8 rows, 2x2 waves, LDS lag>,
//              k = 2 -> pd_zmarch_x2.
//   variant 22 (shipped, opt-in): the reference's roundings through FMA correction steps (FAST = 2) for float32 duals as
//              well, same tilings: bit-identical to the oracle.  Round 4, same-box pairs, 30-iteration prox at 1024^3: 10.5-10.7 ms
//              per three-iteration launch against 9.2 (one box, +16 %; 0.636 vs 0.717 outer iterations/s on the bench,
//              profiles/archive/r4d_bench_exact_vs_relaxed.txt) or 10.05 (another box, +4.6 %; 0.645 vs 0.668) relaxed -- the ~390
//              extra VALU instructions per plane of the correction chains (two quarter-rate transcendentals and 13
//              dependent FMAs per dual row) on a kernel that is bound by instruction issue.  That is why it is not the default.
//   dev flavour: 3 = relaxed arithmetic for both dual types; 2 = the compiler's IEEE sqrt / divide sequences, two iterations
//              per launch (pd_zmarch_x2, 2x2 waves); 21 = the same on the K = 3 tiling.  2 and 21 are bit-identical to the
//              oracle: the independent exactness check of the FMA-corrected build.
static int pd_iters_per_launch(int variant)
{
    if (variant == 1) return 1;
    return variant == 2 ? 2 : 3;
}

// arithmetic level of a shipped / relaxed / exact variant for dual type T (the dev builds 2 / 21 use FAST = 0 explicitly)
template <typename T>
constexpr bool pd_default_is_exact() { return sizeof(T) == 2; }

// three iterations per launch: 8 rows per lane, 10 of the 90 hand-over slots in registers (80 KB of LDS per workgroup = two
// workgroups per CU)
template <typename T, bool NN, bool AN>
int pd_xk3_launch(const PdArgs &a, int variant, hipStream_t st)
{
#if TOMO_DEV
    if (variant == 3) return pd_zmarch_xk_launch<T, NN, AN, 1, 3, 8, 2, 2, true, 10>(a, st);
    if constexpr (sizeof(T) == 4) {  // workgroup shapes of the shipped relaxed kernel, measurement only (tools/archive/probes/pd_time.py)
        if (variant == 31) return pd_zmarch_xk_launch<T, NN, AN, 1, 3, 8, 1, 4, true, 10>(a, st);
        if (variant == 32) return pd_zmarch_xk_launch<T, NN, AN, 1, 3, 8, 4, 1, true, 10>(a, st);
    }
    if (variant == 21) {
        if constexpr (sizeof(T) == 4) return pd_zmarch_xk_launch<T, NN, AN, 0, 3, 8, 2, 2, true, 10>(a, st);
        else return pd_zmarch_xk_launch<T, NN, AN, 0, 3, 4, 2, 2, true>(a, st);  // 4 rows per lane: the IEEE expansions need the registers
    }
#endif
    const bool first = a.p_in_zero && a.u_in == a.in;  // first launch of a prox: its own instantiation (see pd_zmarch_xk.inl, FIRST)
    if (variant == 22 || pd_default_is_exact<T>())
        return first ? pd_zmarch_xk_launch<T, NN, AN, 2, 3, 8, 2, 2, true, 10, true>(a, st) : pd_zmarch_xk_launch<T, NN, AN, 2, 3, 8, 2, 2, true, 10>(a, st);
    // relaxed float32: ONE instantiation per TV type serves both settings of `nonneg` -- the clip threshold is a kernel
    // argument (0 or -inf; "u < -inf" is never true, so the iterate passes through exactly as the code without the test
    // would leave it).  The separate no-clip instantiation of the isotropic kernel allocated 256 registers with 154 spilled
    // and ran 15 % slower than the one with the test (11.9 vs 10.4 ms per launch), and with the threshold in a scalar register
    // the compiler's schedule of the clipping kernel itself is 2.8 % faster (profiles/archive/r4y_pd_instantiations.txt).  The exact
    // builds keep their two instantiations: there the no-clip one is the faster by 3 %.
    PdArgs b = a;
    b.nn_thr = NN ? 0.0f : -__builtin_inff();
    // the first launch of a prox (zero duals, Input = iterate) has its own instantiation: 14 + 32 requests per step, not 26 + 32
    if (first) return pd_zmarch_xk_launch<T, true, AN, 1, 3, 8, 2, 2, true, 10, true>(b, st);
    return pd_zmarch_xk_launch<T, true, AN, 1, 3, 8, 2, 2, true, 10>(b, st);
}

template <typename T, bool NN, bool AN>
int pd_x2_launch(const PdArgs &a, int variant, hipStream_t st)
{
#if TOMO_DEV
    if (variant == 3) return pd_zmarch_x2_launch<T, NN, AN, 1, 4, 2, 4>(a, st);
    if (variant == 2 || variant == 21) return pd_zmarch_x2_launch<T, NN, AN, 0, 4, 2, 2>(a, st);
#endif
    if (variant == 22 || pd_default_is_exact<T>()) return pd_zmarch_x2_launch<T, NN, AN, 2, 4, 2, 2>(a, st);
    return pd_zmarch_x2_launch<T, NN, AN, 1, 4, 2, 4>(a, st);
}

template <typename T>
int pd_multi_launch(const PdArgs &a, int k, int methodTV, int nonneg, int variant, hipStream_t st)
{
#define PD_XK(NN, AN) (k == 3 ? pd_xk3_launch<T, NN, AN>(a, variant, st) : pd_x2_launch<T, NN, AN>(a, variant, st))
    int rc;
    if (!nonneg && !methodTV) rc = PD_XK(false, false);
    else if (nonneg && !methodTV) rc = PD_XK(true, false);
    else if (!nonneg && methodTV) rc = PD_XK(false, true);
    else rc = PD_XK(true, true);
#undef PD_XK
    if (rc != TOMO_OK) return rc;
    TOMO_LAUNCH_CHECK();
    return TOMO_OK;
}

// 2D images: k (1, 2 or 3) iterations per launch, rows in registers (pd_rows2d.inl); the arithmetic follows the variant
template <typename T, bool NN, bool AN, int FAST>
int pd_rows2d_k(const PdArgs &a, int k, hipStream_t st)
{
    if (k == 3) return pd_rows2d_launch<T, NN, AN, FAST, 3, 8>(a, st);
    if (k == 2) return pd_rows2d_launch<T, NN, AN, FAST, 2, 8>(a, st);
    return pd_rows2d_launch<T, NN, AN, FAST, 1, 8>(a, st);
}

template <typename T>
int pd_2d_launch(const PdArgs &a, int k, int methodTV, int nonneg, int variant, hipStream_t st)
{
    const bool exact = variant == 22 || pd_default_is_exact<T>();
#if TOMO_DEV
#define PD_2D_F(NN, AN) ((variant == 2 || variant == 21) ? pd_rows2d_k<T, NN, AN, 0>(a, k, st) : (variant == 3 || !exact) ? pd_rows2d_k<T, NN, AN, 1>(a, k, st) : pd_rows2d_k<T, NN, AN, 2>(a, k, st))
#else
#define PD_2D_F(NN, AN) (exact ? pd_rows2d_k<T, NN, AN, 2>(a, k, st) : pd_rows2d_k<T, NN, AN, 1>(a, k, st))
#endif
    int rc;
    if (!nonneg && !methodTV) rc = PD_2D_F(false, false);
    else if (nonneg && !methodTV) rc = PD_2D_F(true, false);
    else if (!nonneg && methodTV) rc = PD_2D_F(false, true);
    else rc = PD_2D_F(true, true);
#undef PD_2D_F
    if (rc != TOMO_OK) return rc;
    TOMO_LAUNCH_CHECK();
    return TOMO_OK;
}

template <typename T, int ND, bool NONNEG, bool ANISO>
int pd_launch(const PdArgs &a0, int variant, hipStream_t st)
{
    PdArgs a = a0;
    const int nout = a.out_end - a.out_begin;
    if (nout <= 0 || a.dx <= 0 || a.dy <= 0) return TOMO_OK;
    // measured on MI355X, 1024^3 f32 duals (profiles/archive/r1_pdtv_pmc.txt, docs/measurement_log.md): 4x2 waves x 8 rows, lockstep = 8.6 ms;
    // 4x4 waves x 4 rows = 9.2 ms; 4x1 x 8 rows = 8.7 ms; unsynchronised waves (1x4, 4 rows) = 12.3-14 ms.
    // The arithmetic of an iteration never depends on how a run is cut into launches (slabs cut it differently): the
    // single-iteration kernel follows the variant's arithmetic like the fused ones.
    int rc;
#if TOMO_DEV
    if (variant == 1) {
        dim3 grid(ceil_div(a.dx, 256), a.dy, nout);
        pd_pervoxel_kernel<T, ND, NONNEG, ANISO><<<grid, 256, 0, st>>>(a);
        TOMO_LAUNCH_CHECK();
        return TOMO_OK;
    }
    if (variant == 2 || variant == 21) rc = pd_zmarch2_launch<T, ND, NONNEG, ANISO, 0, 8, true, 4, 2>(a, st);
    else
    if (variant == 3) rc = pd_zmarch2_launch<T, ND, NONNEG, ANISO, 1, 8, true, 4, 2>(a, st);
    else
#endif
    if (variant == 22 || pd_default_is_exact<T>()) rc = pd_zmarch2_launch<T, ND, NONNEG, ANISO, 2, 8, true, 4, 2>(a, st);
    else rc = pd_zmarch2_launch<T, ND, NONNEG, ANISO, 1, 8, true, 4, 2>(a, st);
    if (rc != TOMO_OK) return rc;
    TOMO_LAUNCH_CHECK();
    return TOMO_OK;
}

template <typename T, int ND>
int pd_dispatch(const PdArgs &a, int methodTV, int nonneg, int variant, hipStream_t st)
{
    if (!nonneg && !methodTV) return pd_launch<T, ND, false, false>(a, variant, st);
    if (nonneg && !methodTV) return pd_launch<T, ND, true, false>(a, variant, st);
    if (!nonneg && methodTV) return pd_launch<T, ND, false, true>(a, variant, st);
    return pd_launch<T, ND, true, true>(a, variant, st);
}

int pd_iter(const PdArgs &a, int nd, int methodTV, int nonneg, int half, hipStream_t st)
{
    const int v = g_variant_pdtv;
    if (nd == 3) return half ? pd_dispatch<__half, 3>(a, methodTV, nonneg, v, st) : pd_dispatch<float, 3>(a, methodTV, nonneg, v, st);
#if TOMO_DEV
    return half ? pd_dispatch<__half, 2>(a, methodTV, nonneg, v, st) : pd_dispatch<float, 2>(a, methodTV, nonneg, v, st);  // per-voxel 2D kernel (variant 1)
#else
    return tomo_fail(TOMO_E_INVALID, "internal: 2D PD_TV runs pd_rows2d in this library");
#endif
}

// ------------------------------------------------------------------------------------------ ROF
struct RofArgs {
    const float *in;
    const float *u_in;
    float *u_out;
    int dx, dy, planes, out_begin, out_end;
    int first_is_edge, last_is_edge;
    float lambda, tau;
};

__device__ __forceinline__ float rof_mm(float n0, float n1)
{
    // (0.5*(sign(n1)+sign(n0)) * min(|n1|,|n0|))^2   (rudin_osher...cu:51-55)
    // = min(|n1|,|n0|)^2 where both have the same sign and neither is zero, else 0.  max(min(n1,n0), min(-n1,-n0)) is
    // min(|n1|,|n0|) for equal signs, <= 0 otherwise (0 when one of them is zero): two v_min, one v_max3, one multiply
    // instead of the twelve compare / select / carry operations of the sign arithmetic; (+-m)^2 rounds like m^2.
    const float m = fmaxf(fmaxf(fminf(n1, n0), fminf(-n1, -n0)), 0.0f);
    return m * m;
}

__device__ __forceinline__ float rof_norm(float nom, float d1, float d2, float d3)
{
    float s = (d1 + d2) + d3;
    float den = sqrtf((float)((double)s + 1.0e-8));  // EPS is a double literal in the reference (:7,:59)
    return nom / den;
}

#if TOMO_DEV
// D component `comp` (0: pairs with y/j, 1: with x/i, 2: with z/k) at voxel (i,j,k)
template <int ND, bool HALF>
__device__ __forceinline__ float rof_D(const RofArgs &a, int i, int j, int k, int comp)
{
    const size_t sy = (size_t)a.dx, sz = (size_t)a.dx * a.dy;
    const float *U = a.u_in;
    const int i1 = (i == a.dx - 1) ? i - 1 : i + 1, i2 = (i == 0) ? i + 1 : i - 1;
    const int j1 = (j == a.dy - 1) ? j - 1 : j + 1, j2 = (j == 0) ? j + 1 : j - 1;
    const size_t base = sz * k;
    const float u = U[base + sy * j + i];
    const float nx1 = U[base + sy * j1 + i] - u, nx0 = u - U[base + sy * j2 + i];
    const float ny1 = U[base + sy * j + i1] - u, ny0 = u - U[base + sy * j + i2];
    const float dxm = rof_mm(nx0, nx1), dym = rof_mm(ny0, ny1);
    float nz1 = 0.0f, dzm = 0.0f;
    if (ND == 3) {
        const bool k_last = (k == a.planes - 1) && a.last_is_edge;
        const bool k_first = (k == 0) && a.first_is_edge;
        const int k1 = k_last ? k - 1 : k + 1, k2 = k_first ? k + 1 : k - 1;
        nz1 = U[sz * k1 + sy * j + i] - u;
        const float nz0 = u - U[sz * k2 + sy * j + i];
        dzm = rof_mm(nz0, nz1);
    }
    float d;
    if (comp == 0) d = rof_norm(nx1, nx1 * nx1, dym, dzm);
    else if (comp == 1) d = rof_norm(ny1, dxm, ny1 * ny1, dzm);
    else d = rof_norm(nz1, dxm, dym, nz1 * nz1);
    return HALF ? DualIO<__half>::rt(d) : d;
}

template <int ND, bool HALF>
__global__ __launch_bounds__(256) void rof_pervoxel_kernel(RofArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    const int k = a.out_begin + blockIdx.z;
    if (i >= a.dx) return;
    const size_t idx = (size_t)i + (size_t)a.dx * j + (size_t)a.dx * a.dy * k;
    const int i2 = (i == 0) ? i + 1 : i - 1;
    const int j2 = (j == 0) ? j + 1 : j - 1;
    float dv = (rof_D<ND, HALF>(a, i, j, k, 0) - rof_D<ND, HALF>(a, i, j2, k, 0)) +
               (rof_D<ND, HALF>(a, i, j, k, 1) - rof_D<ND, HALF>(a, i2, j, k, 1));
    if (ND == 3) {
        const bool k_first = (k == 0) && a.first_is_edge;
        const int k2 = k_first ? k + 1 : k - 1;
        dv = dv + (rof_D<ND, HALF>(a, i, j, k, 2) - rof_D<ND, HALF>(a, i, j, k2, 2));
    }
    const float u = a.u_in[idx];
    const float t = fmaf(a.lambda, dv, -(u - a.in[idx]));
    a.u_out[idx] = fmaf(a.tau, t, u);
}

#endif  // TOMO_DEV

#include "rof_zmarch.inl"

// variant 0 (shipped, float32 and binary16 D fields): the reference's rounding sequence reproduced with FMA correction
//            steps (rof_eval FAST = 3): bit-identical to the oracle, 3.4 ms per 1024^3 iteration.
// dev flavour: 2 = the same roundings through the compiler's IEEE sqrt / divide expansions (independent check, 4.4 ms);
//              3 = relaxed arithmetic (float32 sum + v_rsq_f32, 2.85 ms) -- measurement only: the D normalisation has a gain
//              of ~1e4 on noise-dominated data, where one-ulp differences grow to 2.5e-5 .. 4.5e-5 after 60 iterations
//              (profiles/archive/r3_rof_variants.txt), beyond the 1e-5 parity bar;  4 = refined v_rsq / v_rcp (no better than 3);
//              1 = per-voxel kernel (independent implementation)
template <int ND, bool HALF>
int rof_zmarch_dispatch(const RofArgs &a, int variant, hipStream_t st)
{
    int rc;
#if TOMO_DEV
    if (variant == 4) rc = rof_zmarch_launch<ND, HALF, 2, 8, 2, 2>(a, st);
    else if (variant == 3) rc = rof_zmarch_launch<ND, HALF, 1, 8, 2, 2>(a, st);
    else if (variant == 2) rc = rof_zmarch_launch<ND, HALF, 0, 8, 2, 2>(a, st);
    else
#endif
    rc = rof_zmarch_launch<ND, HALF, 3, 8, 2, 2>(a, st);
    (void)variant;
    if (rc != TOMO_OK) return rc;
    TOMO_LAUNCH_CHECK();
    return TOMO_OK;
}

int rof_iter(const RofArgs &a, int nd, int half, hipStream_t st)
{
    const int nout = a.out_end - a.out_begin;
    if (nout <= 0) return TOMO_OK;
#if TOMO_DEV
    if (g_variant_roftv == 1) {  // per-voxel kernel
        dim3 grid(ceil_div(a.dx, 256), a.dy, nout);
        if (nd == 3) {
            if (half) rof_pervoxel_kernel<3, true><<<grid, 256, 0, st>>>(a);
            else rof_pervoxel_kernel<3, false><<<grid, 256, 0, st>>>(a);
        } else {
            if (half) rof_pervoxel_kernel<2, true><<<grid, 256, 0, st>>>(a);
            else rof_pervoxel_kernel<2, false><<<grid, 256, 0, st>>>(a);
        }
        TOMO_LAUNCH_CHECK();
        return TOMO_OK;
    }
#endif
    const int v = g_variant_roftv;
    if (nd == 3) return half ? rof_zmarch_dispatch<3, true>(a, v, st) : rof_zmarch_dispatch<3, false>(a, v, st);
    return half ? rof_zmarch_dispatch<2, true>(a, v, st) : rof_zmarch_dispatch<2, false>(a, v, st);
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

// ------------------------------------------------------------------------------------------ C-ABI
// Scratch arrays of equal power-of-two size laid end to end put the same voxel of every array on the same HBM
// channel and bank; the 9-10 concurrent streams of a TV iteration then thrash DRAM rows.  Each array is therefore
// started `tv_skew()` bytes further than the plain packing would (k-th array: k * skew).
static size_t tv_skew()
{
    return 69888;  // 68 KiB + 256 B: measured -13 % on the 1024^3 PD_TV iteration vs 0
}

extern "C" size_t tomo_pdtv_scratch_bytes(int dx, int dy, int dz, int nd, int half)
{
    if (nd == 2) dz = 1;
    const size_t nvox = (size_t)dx * dy * dz;
    const size_t ub = align_up(nvox * sizeof(float), 256);
    const size_t pb = align_up(nvox * (half ? 2 : 4), 256);
    return 2 * ub + 2 * (size_t)nd * pb + 8 * tv_skew();
}

extern "C" size_t tomo_roftv_scratch_bytes(int dx, int dy, int dz, int nd)
{
    if (nd == 2) dz = 1;
    return 2 * align_up((size_t)dx * dy * dz * sizeof(float), 256);
}

extern "C" int tomo_pdtv_iters_per_launch(int half)
{
    (void)half;
    return pd_iters_per_launch(g_variant_pdtv);
}

extern "C" int tomo_pdtv(int device, const float *in_dev, float *out_dev, int dx, int dy, int dz, int nd,
                         float sigma, float tau, float lt, float theta, int iters, int methodTV, int nonneg,
                         int half, void *stream)
{
    TOMO_REQUIRE(device >= 0, "The gpu_device must be a positive integer or zero");
    TOMO_REQUIRE(nd == 2 || nd == 3, "2D or 3D arrays must be provided only");
    if (nd == 2) dz = 1;
    TOMO_REQUIRE(dx > 0 && dy > 0 && dz > 0 && iters >= 0, "bad PD_TV dimensions / iterations");
    TOMO_REQUIRE((size_t)dx * (size_t)dy < ((size_t)1 << 29), "a plane of %d x %d exceeds the 2 GiB a buffer descriptor of the TV kernels addresses", dx, dy);
    TOMO_REQUIRE(in_dev && out_dev, "NULL data pointer");
    TOMO_ON_DEVICE(device);
    hipStream_t st = as_stream(stream);
    const size_t nvox = (size_t)dx * dy * dz;
    if (iters == 0) {
        if (out_dev != in_dev) TOMO_HIP(hipMemcpyAsync(out_dev, in_dev, nvox * sizeof(float), hipMemcpyDeviceToDevice, st));
        return TOMO_OK;
    }
    void *base = nullptr;
    int rc = tomo_arena_get(device, st, ARENA_TV, tomo_pdtv_scratch_bytes(dx, dy, dz, nd, half), &base, true);
    if (rc != TOMO_OK) return rc;
    const size_t ub = align_up(nvox * sizeof(float), 256);
    const size_t pb = align_up(nvox * (half ? 2 : 4), 256);
    char *cur = (char *)base;
    float *U[2];
    void *P[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    const size_t skew = tv_skew();
    U[0] = (float *)cur; cur += ub + skew;
    U[1] = (float *)cur; cur += ub + skew;
    for (int b = 0; b < 2; ++b)
        for (int c = 0; c < nd; ++c) { P[b][c] = cur; cur += pb + skew; }
    // U_arrays[0] = data.copy() is not materialised: iteration 0 reads the caller's buffer directly;
    // duals start at zero (regularisersCuPy.py:221-223); every U / P output buffer is fully overwritten
    // (the multi-iteration kernel pd_zmarch_xk takes "the duals are zero" as a flag instead of reading a zeroed array)
    // 3D volumes run several iterations per launch (K = 3 or 2, see pd_multi_launch) while that many remain, then
    // single iterations; variant 1 keeps one iteration per launch (independent implementation)
    const int v = g_variant_pdtv;
    // a fused launch of k iterations marches k planes ahead of its output: volumes thinner than that take fewer per launch.
    // 2D images (round 4): k iterations per launch with the rows of a tile in registers (pd_rows2d.inl), any k <= 3
    const bool rows2d = (nd == 2 && v != 1);
    const int kmax = nd == 3 ? std::min(pd_iters_per_launch(v), dz) : pd_iters_per_launch(v);
    // cut `remaining` into launches of kmax / 2 / 1 iterations with as few single-iteration launches as possible
    // (4 = 2 + 2, 7 = 3 + 2 + 2: a single iteration costs 1.7x an iteration of a fused launch)
    auto step_of = [&](int remaining) {
        if (kmax >= 3 && remaining >= 3 && remaining != 4) return 3;
        return (remaining >= 2 && kmax >= 2) ? 2 : 1;
    };
    int launches = 0;
    for (int it = 0; it < iters;) launches += 1, it += step_of(iters - it);
    tomo_prof_scope prof(PROF_PDTV, st, launches);
    int cset = 0;  // buffer set holding the current iterate (iteration 0 reads the caller's buffer instead of U[0])
    for (int it = 0; it < iters;) {
        const int step = step_of(iters - it);
        const bool pair = step >= 2;
        const int ib = cset, ob = cset ^ 1;
        PdArgs a;
        a.in = in_dev;
        a.u_in = (it == 0) ? in_dev : U[ib];
        // the last launch writes straight into the caller's output buffer (unless it aliases the input)
        const bool last = (it + step == iters);
        a.u_out = (last && out_dev != in_dev) ? out_dev : U[ob];
        for (int c = 0; c < 3; ++c) { a.p_in[c] = P[ib][c]; a.p_out[c] = P[ob][c]; }
        const bool flags = (step == 3 && nd == 3) || rows2d;  // launches that understand the two flags below
        if (it == 0) {
            if (flags) a.p_in_zero = 1;
            else for (int c = 0; c < nd; ++c) TOMO_HIP(hipMemsetAsync(P[ib][c], 0, pb, st));
        }
        if (last && flags) a.p_out_skip = 1;
        a.dx = dx; a.dy = dy; a.planes = dz; a.out_begin = 0; a.out_end = dz;
        a.first_is_edge = 1; a.last_is_edge = 1;
        a.sigma = sigma; a.tau = tau; a.lt = lt; a.theta = theta; a.zchunk = dz;
#if TOMO_DEV
        a.probe = g_probe;
#endif
        if (rows2d) rc = half ? pd_2d_launch<__half>(a, step, methodTV, nonneg, v, st) : pd_2d_launch<float>(a, step, methodTV, nonneg, v, st);
        else if (pair) rc = half ? pd_multi_launch<__half>(a, step, methodTV, nonneg, v, st) : pd_multi_launch<float>(a, step, methodTV, nonneg, v, st);
        else rc = pd_iter(a, nd, methodTV, nonneg, half, st);
        if (rc != TOMO_OK) return rc;
        cset = ob;
        it += step;
    }
    if (out_dev == in_dev)
        TOMO_HIP(hipMemcpyAsync(out_dev, U[cset], nvox * sizeof(float), hipMemcpyDeviceToDevice, st));
    return TOMO_OK;
}

extern "C" int tomo_pdtv_iter_slab(int device, const float *in_dev, const float *u_in_dev, float *u_out_dev,
                                   const void *p_in_dev[3], void *p_out_dev[3], int dx, int dy, int nz_local,
                                   int has_lo, int has_hi, float sigma, float tau, float lt, float theta,
                                   int methodTV, int nonneg, int half, void *stream)
{
    TOMO_REQUIRE(device >= 0 && dx > 0 && dy > 0 && nz_local > 0, "bad slab arguments");
    TOMO_REQUIRE((size_t)dx * (size_t)dy < ((size_t)1 << 29), "a plane of %d x %d exceeds the 2 GiB a buffer descriptor of the TV kernels addresses", dx, dy);
    TOMO_ON_DEVICE(device);
    PdArgs a;
    a.in = in_dev; a.u_in = u_in_dev; a.u_out = u_out_dev;
    for (int c = 0; c < 3; ++c) { a.p_in[c] = p_in_dev[c]; a.p_out[c] = p_out_dev[c]; }
    a.dx = dx; a.dy = dy;
    a.planes = nz_local + (has_lo ? 1 : 0) + (has_hi ? 1 : 0);
    a.out_begin = has_lo ? 1 : 0;
    a.out_end = a.out_begin + nz_local;
    a.first_is_edge = has_lo ? 0 : 1;
    a.last_is_edge = has_hi ? 0 : 1;
    a.sigma = sigma; a.tau = tau; a.lt = lt; a.theta = theta; a.zchunk = nz_local;
    tomo_prof_scope prof(PROF_PDTV, as_stream(stream), 1);
    return pd_iter(a, 3, methodTV, nonneg, half, as_stream(stream));
}

extern "C" int tomo_pdtv_pair_slab(int device, const float *in_dev, const float *u_in_dev, float *u_out_dev,
                                   const void *p_in_dev[3], void *p_out_dev[3], int dx, int dy, int nz_local,
                                   int lo_planes, int hi_planes, float sigma, float tau, float lt, float theta,
                                   int methodTV, int nonneg, int half, void *stream)
{
    return tomo_pdtv_multi_slab_range(device, in_dev, u_in_dev, u_out_dev, p_in_dev, p_out_dev, dx, dy, nz_local, lo_planes,
                                      hi_planes, 0, nz_local, 2, sigma, tau, lt, theta, methodTV, nonneg, half, stream);
}

extern "C" int tomo_pdtv_pair_slab_range(int device, const float *in_dev, const float *u_in_dev, float *u_out_dev,
                                         const void *p_in_dev[3], void *p_out_dev[3], int dx, int dy, int nz_local,
                                         int lo_planes, int hi_planes, int z_begin, int z_end, float sigma, float tau,
                                         float lt, float theta, int methodTV, int nonneg, int half, void *stream)
{
    return tomo_pdtv_multi_slab_range(device, in_dev, u_in_dev, u_out_dev, p_in_dev, p_out_dev, dx, dy, nz_local, lo_planes,
                                      hi_planes, z_begin, z_end, 2, sigma, tau, lt, theta, methodTV, nonneg, half, stream);
}

extern "C" int tomo_pdtv_multi_slab_range(int device, const float *in_dev, const float *u_in_dev, float *u_out_dev,
                                          const void *p_in_dev[3], void *p_out_dev[3], int dx, int dy, int nz_local,
                                          int lo_planes, int hi_planes, int z_begin, int z_end, int k, float sigma,
                                          float tau, float lt, float theta, int methodTV, int nonneg, int half, void *stream)
{
    TOMO_REQUIRE(k == 2 || k == 3, "a fused PD_TV launch carries 2 or 3 iterations (got %d)", k);
    TOMO_REQUIRE(device >= 0 && dx > 0 && dy > 0 && nz_local >= k, "bad slab arguments (a slab needs >= %d slices)", k);
    TOMO_REQUIRE((size_t)dx * (size_t)dy < ((size_t)1 << 29), "a plane of %d x %d exceeds the 2 GiB a buffer descriptor of the TV kernels addresses", dx, dy);
    TOMO_REQUIRE(z_begin >= 0 && z_begin <= z_end && z_end <= nz_local, "bad output plane range [%d, %d)", z_begin, z_end);
    if (z_begin == z_end) return TOMO_OK;
    TOMO_REQUIRE((lo_planes == 0 || lo_planes >= k) && (hi_planes == 0 || hi_planes >= k) && lo_planes <= 3 && hi_planes <= 3,
                 "a %d-iteration slab launch needs 0 or >= %d (at most 3) ghost planes on either side", k, k);
    TOMO_ON_DEVICE(device);
    PdArgs a;
    a.in = in_dev; a.u_in = u_in_dev; a.u_out = u_out_dev;
    for (int c = 0; c < 3; ++c) { a.p_in[c] = p_in_dev[c]; a.p_out[c] = p_out_dev[c]; }
    a.dx = dx; a.dy = dy;
    a.planes = nz_local + lo_planes + hi_planes;
    a.out_begin = lo_planes + z_begin;
    a.out_end = lo_planes + z_end;
    a.first_is_edge = lo_planes ? 0 : 1;
    a.last_is_edge = hi_planes ? 0 : 1;
    a.sigma = sigma; a.tau = tau; a.lt = lt; a.theta = theta; a.zchunk = z_end - z_begin;
    hipStream_t st = as_stream(stream);
    tomo_prof_scope prof(PROF_PDTV, st, 1);
    // fused slab launches follow the variant's arithmetic; the dev variants without a fused form of their own (1: per-voxel,
    // 2: two iterations per launch) run the compiler-IEEE build of the tiling asked for
    int v = g_variant_pdtv;
    if (v == 1 || v == 2) v = (k == 3) ? 21 : 2;
    return half ? pd_multi_launch<__half>(a, k, methodTV, nonneg, v, st) : pd_multi_launch<float>(a, k, methodTV, nonneg, v, st);
}

extern "C" int tomo_roftv(int device, const float *in_dev, float *out_dev, int dx, int dy, int dz, int nd,
                          float lambda, float tau, int iters, int half, void *stream)
{
    TOMO_REQUIRE(device >= 0, "The gpu_device must be a positive integer or zero");
    TOMO_REQUIRE(nd == 2 || nd == 3, "2D or 3D arrays must be provided only");
    if (nd == 2) dz = 1;
    TOMO_REQUIRE(dx >= 2 && dy >= 2 && (nd == 2 || dz >= 2) && iters >= 0,
                 "ROF_TV needs every dimension >= 2 (reflecting boundary)");
    TOMO_REQUIRE((size_t)dx * (size_t)dy < ((size_t)1 << 29), "a plane of %d x %d exceeds the 2 GiB a buffer descriptor of the TV kernels addresses", dx, dy);
    TOMO_REQUIRE(in_dev && out_dev, "NULL data pointer");
    TOMO_ON_DEVICE(device);
    hipStream_t st = as_stream(stream);
    const size_t nvox = (size_t)dx * dy * dz;
    if (iters == 0) {
        if (out_dev != in_dev) TOMO_HIP(hipMemcpyAsync(out_dev, in_dev, nvox * sizeof(float), hipMemcpyDeviceToDevice, st));
        return TOMO_OK;
    }
    void *base = nullptr;
    int rc = tomo_arena_get(device, st, ARENA_TV, tomo_roftv_scratch_bytes(dx, dy, dz, nd), &base, true);
    if (rc != TOMO_OK) return rc;
    const size_t ub = align_up(nvox * sizeof(float), 256);
    float *U[2] = {(float *)base, (float *)((char *)base + ub)};
    tomo_prof_scope prof(PROF_ROFTV, st, iters);
    for (int it = 0; it < iters; ++it) {
        RofArgs a;
        a.in = in_dev;
        a.u_in = (it == 0) ? in_dev : U[it & 1];  // iteration 0 reads the caller's data directly
        a.u_out = (it == iters - 1 && out_dev != in_dev) ? out_dev : U[(it + 1) & 1];
        a.dx = dx; a.dy = dy; a.planes = dz; a.out_begin = 0; a.out_end = dz;
        a.first_is_edge = 1; a.last_is_edge = 1;
        a.lambda = lambda; a.tau = tau;
        rc = rof_iter(a, nd, half, st);
        if (rc != TOMO_OK) return rc;
    }
    if (out_dev == in_dev)
        TOMO_HIP(hipMemcpyAsync(out_dev, U[iters & 1], nvox * sizeof(float), hipMemcpyDeviceToDevice, st));
    return TOMO_OK;
}

extern "C" int tomo_roftv_iter_slab(int device, const float *in_dev, const float *u_in_dev, float *u_out_dev,
                                    int dx, int dy, int nz_local, int lo_planes, int hi_planes,
                                    float lambda, float tau, int half, void *stream)
{
    return tomo_roftv_iter_slab_range(device, in_dev, u_in_dev, u_out_dev, dx, dy, nz_local, lo_planes, hi_planes, 0,
                                      nz_local, lambda, tau, half, stream);
}

extern "C" int tomo_roftv_iter_slab_range(int device, const float *in_dev, const float *u_in_dev, float *u_out_dev,
                                          int dx, int dy, int nz_local, int lo_planes, int hi_planes, int z_begin,
                                          int z_end, float lambda, float tau, int half, void *stream)
{
    TOMO_REQUIRE(device >= 0 && dx >= 2 && dy >= 2 && nz_local > 0, "bad slab arguments");
    TOMO_REQUIRE((size_t)dx * (size_t)dy < ((size_t)1 << 29), "a plane of %d x %d exceeds the 2 GiB a buffer descriptor of the TV kernels addresses", dx, dy);
    TOMO_REQUIRE(z_begin >= 0 && z_begin <= z_end && z_end <= nz_local, "bad output plane range [%d, %d)", z_begin, z_end);
    if (z_begin == z_end) return TOMO_OK;
    TOMO_REQUIRE((lo_planes == 0 || lo_planes == 2) && (hi_planes == 0 || hi_planes == 1),
                 "ROF slab needs 0 or 2 ghost planes below and 0 or 1 above");
    TOMO_ON_DEVICE(device);
    RofArgs a;
    a.in = in_dev; a.u_in = u_in_dev; a.u_out = u_out_dev;
    a.dx = dx; a.dy = dy;
    a.planes = nz_local + lo_planes + hi_planes;
    a.out_begin = lo_planes + z_begin;
    a.out_end = lo_planes + z_end;
    a.first_is_edge = lo_planes ? 0 : 1;
    a.last_is_edge = hi_planes ? 0 : 1;
    a.lambda = lambda; a.tau = tau;
    return rof_iter(a, 3, half, as_stream(stream));
}
