// Round-4 check of an advisor finding: mk_recip (v_rcp + two Newton steps) and the ROF quotient (v_rcp + one Newton step +
// exact-residual correction) would not be correctly rounded for q = 0x1.fffffep+k IF v_rcp_f32 returned the 1-ulp-low value
// 2^-(k+1) there (the Newton update then lands exactly on a tie and rounds back).  What does the hardware do?
//   (1) v_rcp_f32 of the all-ones mantissa in every binade, next to the correctly rounded 1/q;
//   (2) mk_recip against 1.0f / q for EVERY normal float q in [2^-100, 2^100];
//   (3) the ROF quotient for every all-ones q in [2^-20, 2^20] x every power-of-two and 2^20 random numerators.
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -fno-fast-math -w recip_allones_probe.hip -o recip_allones_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ float mk_recip(float q)
{
    float y = __builtin_amdgcn_rcpf(q);
    y = fmaf(fmaf(-q, y, 1.0f), y, y);
    return fmaf(fmaf(-q, y, 1.0f), y, y);
}
__device__ __forceinline__ float mk_recip1(float q)  // ONE Newton step: is it already correctly rounded on this hardware?
{
    float y = __builtin_amdgcn_rcpf(q);
    return fmaf(fmaf(-q, y, 1.0f), y, y);
}
__device__ __forceinline__ float mk_div(float nom, float q)
{
    float y = __builtin_amdgcn_rcpf(q);
    y = fmaf(fmaf(-q, y, 1.0f), y, y);
    float z = nom * y;
    return fmaf(fmaf(-q, z, nom), y, z);
}

__global__ void rcp_allones(float *raw, float *want)
{
    const int k = threadIdx.x;  // exponent field 1..254
    if (k < 1 || k > 254) return;
    const float q = __uint_as_float(((uint32_t)k << 23) | 0x7fffffu);
    raw[k] = __builtin_amdgcn_rcpf(q);
    want[k] = 1.0f / q;
}

template <int STEPS>
__global__ void recip_all(uint32_t lo, uint32_t hi, unsigned long long *bad, uint32_t *first)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += stride) {
        const float q = __uint_as_float((uint32_t)b);
        const float a = STEPS == 2 ? mk_recip(q) : mk_recip1(q), w = 1.0f / q;
        if (__float_as_uint(a) != __float_as_uint(w))
            if (atomicAdd(bad, 1ULL) < 8) first[atomicAdd(first + 8, 1u) & 7] = (uint32_t)b;
    }
}

__device__ uint32_t rng(uint64_t &s) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (uint32_t)(s >> 32); }

__global__ void div_allones(unsigned long long *bad, float *first, unsigned long long *count)
{
    // q: all-ones mantissa, exponent field 107..147 (2^-20 .. 2^20); blockIdx.y picks it
    const float q = __uint_as_float(((uint32_t)(107 + blockIdx.y) << 23) | 0x7fffffu);
    uint64_t s = 0x9E3779B97F4A7C15ULL * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1) + blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int it = 0; it < 64; ++it) {
        float nom;
        if (it == 0 && t < 200) nom = __uint_as_float((uint32_t)(27 + t) << 23);           // every power of two 2^-100 .. 2^99
        else if (it == 1 && t < 200) nom = -__uint_as_float((uint32_t)(27 + t) << 23);
        else nom = __uint_as_float((rng(s) & 0x807fffffu) | ((uint32_t)(97 + rng(s) % 60) << 23));
        const float z = mk_div(nom, q), w = nom / q;
        atomicAdd(count, 1ULL);
        if (__float_as_uint(z) != __float_as_uint(w)) {
            const unsigned long long k = atomicAdd(bad, 1ULL);
            if (k < 4) { first[2 * k] = nom; first[2 * k + 1] = q; }
        }
    }
}

int main()
{
    float *raw, *want; unsigned long long *bad, *count; uint32_t *first; float *ff;
    hipMalloc(&raw, 1024); hipMalloc(&want, 1024); hipMalloc(&bad, 8); hipMalloc(&count, 8); hipMalloc(&first, 64); hipMalloc(&ff, 64);
    rcp_allones<<<1, 256>>>(raw, want);
    float hr[256], hw[256];
    hipMemcpy(hr, raw, 1024, hipMemcpyDeviceToHost); hipMemcpy(hw, want, 1024, hipMemcpyDeviceToHost);
    int low = 0, exact = 0, other = 0;
    for (int k = 1; k <= 254; ++k) {
        uint32_t a, b; memcpy(&a, &hr[k], 4); memcpy(&b, &hw[k], 4);
        if (a == b) ++exact; else if (a + 1 == b) ++low; else ++other;
    }
    printf("v_rcp_f32(0x1.fffffep+k), 254 binades: equal to RN(1/q) in %d, one ulp low in %d, other in %d   (k=0: raw %a, RN %a)\n",
           exact, low, other, hr[127], hw[127]);
    hipMemset(bad, 0, 8); hipMemset(first, 0, 64);
    uint32_t lo, hi; float flo = 0x1p-100f, fhi = 0x1p100f; memcpy(&lo, &flo, 4); memcpy(&hi, &fhi, 4);
    unsigned long long nb, nc; uint32_t f[9];
    recip_all<2><<<8192, 256>>>(lo, hi, bad, first);
    hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(f, first, 36, hipMemcpyDeviceToHost);
    printf("mk_recip (v_rcp + 2 Newton steps): %u inputs in [2^-100, 2^100], mismatches vs 1.0f / q: %llu", hi - lo + 1, nb);
    for (int i = 0; i < 4 && (unsigned long long)i < nb; ++i) { float v; memcpy(&v, &f[i], 4); printf("  q=%a", v); }
    printf("\n");
    hipMemset(bad, 0, 8); hipMemset(first, 0, 64);
    recip_all<1><<<8192, 256>>>(lo, hi, bad, first);
    hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(f, first, 36, hipMemcpyDeviceToHost);
    printf("          (v_rcp + 1 Newton step ): %u inputs in [2^-100, 2^100], mismatches vs 1.0f / q: %llu", hi - lo + 1, nb);
    for (int i = 0; i < 4 && (unsigned long long)i < nb; ++i) { float v; memcpy(&v, &f[i], 4); printf("  q=%a", v); }
    printf("\n");
    hipMemset(bad, 0, 8); hipMemset(count, 0, 8); hipMemset(ff, 0, 64);
    div_allones<<<dim3(64, 41), 256>>>(bad, ff, count);
    float g[8];
    hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&nc, count, 8, hipMemcpyDeviceToHost); hipMemcpy(g, ff, 32, hipMemcpyDeviceToHost);
    printf("ROF quotient (v_rcp + 1 Newton step + residual correction), q = all-ones mantissa in 41 binades: %llu pairs, mismatches vs '/': %llu", nc, nb);
    for (int i = 0; i < 4 && (unsigned long long)i < nb; ++i) printf("  (%a / %a)", g[2 * i], g[2 * i + 1]);
    printf("\n");
    return 0;
}
