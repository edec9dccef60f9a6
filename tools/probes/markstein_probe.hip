// Exhaustive / randomised check of the FMA-corrected sqrt and quotient used by the shipped ROF_TV normalisation
// (csrc/rof_zmarch.inl, FAST = 3) against the compiler's correctly rounded sqrtf and '/'.
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -fno-fast-math tools/probes/markstein_probe.hip -o /tmp/mk && /tmp/mk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ float mk_sqrt(float x)
{
    const float r = __builtin_amdgcn_rsqf(x);
    float q = x * r, h = 0.5f * r;
    const float e = fmaf(-h, q, 0.5f);
    q = fmaf(q, e, q);
    h = fmaf(h, e, h);
    return fmaf(fmaf(-q, q, x), h, q);
}
__device__ __forceinline__ float mk_div(float nom, float q)
{
    float y = __builtin_amdgcn_rcpf(q);
    y = fmaf(fmaf(-q, y, 1.0f), y, y);
    float z = nom * y;
    return fmaf(fmaf(-q, z, nom), y, z);
}

__global__ void sqrt_all(uint32_t lo, uint32_t hi, unsigned long long *bad, uint32_t *first)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const float a = mk_sqrt(x), w = sqrtf(x);
        if (__float_as_uint(a) != __float_as_uint(w)) {
            if (atomicAdd(bad, 1ULL) < 8) first[atomicAdd(first + 8, 1u) & 7] = (uint32_t)b;
        }
    }
}

__device__ uint32_t rng(uint64_t &s) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (uint32_t)(s >> 32); }

// mode 0: nom random in +-[2^-30, 2), q = sqrtf(random x in [1e-8, 4));  mode 1: nom = k * 2^-24 .. small integer multiples of an
// ulp (differences of voxel values), mode 2: nom rounded through binary16-ish grids (coarse mantissas)
__global__ void div_rand(int mode, unsigned long long per_thread, unsigned long long *bad, float *first)
{
    uint64_t s = 0x9E3779B97F4A7C15ULL * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1) + mode;
    for (unsigned long long it = 0; it < per_thread; ++it) {
        const uint32_t a = rng(s), b = rng(s), c = rng(s);
        float x = __uint_as_float(0x322BCC77u + b % (0x40800000u - 0x322BCC77u));  // [1e-8, 4)
        float nom;
        if (mode == 0) nom = __uint_as_float((a & 0x80000000u) | (0x30800000u + (a & 0x7fffffffu) % (0x40000000u - 0x30800000u)));
        else if (mode == 1) nom = (float)((int)(a % 4001) - 2000) * __uint_as_float(0x33800000u) * (float)(1 << (c % 12));
        else nom = __uint_as_float(a & 0xffffe000u & 0xbfffffffu);  // 10-bit mantissa, |nom| < 2
        if (mode == 1 || mode == 2) x = fmaf(nom, nom, __uint_as_float(0x322BCC77u) + (float)(c % 1000) * 1e-7f);
        const float q = sqrtf(x);
        const float z = mk_div(nom, q), w = nom / q;
        if (__float_as_uint(z) != __float_as_uint(w) && !(z == 0.0f && w == 0.0f)) {
            const unsigned long long k = atomicAdd(bad, 1ULL);
            if (k < 4) { first[2 * k] = nom; first[2 * k + 1] = q; }
        }
    }
}

int main()
{
    unsigned long long *bad; uint32_t *first; float *ff;
    hipMalloc(&bad, 8); hipMalloc(&first, 64); hipMalloc(&ff, 64);
    hipMemset(bad, 0, 8); hipMemset(first, 0, 64);
    uint32_t lo, hi; float flo = 1e-8f, fhi = 1e6f; memcpy(&lo, &flo, 4); memcpy(&hi, &fhi, 4);
    sqrt_all<<<4096, 256>>>(lo, hi, bad, first);
    unsigned long long nb; uint32_t f[9];
    hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(f, first, 36, hipMemcpyDeviceToHost);
    printf("sqrt: %u inputs in [1e-8, 1e6], mismatches vs sqrtf: %llu", hi - lo + 1, nb);
    for (int i = 0; i < 4 && (unsigned long long)i < nb; ++i) { float v; memcpy(&v, &f[i], 4); printf("  x=%.9g(0x%08x)", v, f[i]); }
    printf("\n");
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(bad, 0, 8); hipMemset(ff, 0, 64);
        div_rand<<<2048, 256>>>(mode, 4000, bad, ff);
        float g[8];
        hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(g, ff, 32, hipMemcpyDeviceToHost);
        printf("div mode %d: %llu pairs, mismatches vs '/': %llu", mode, 2048ULL * 256 * 4000, nb);
        for (int i = 0; i < 4 && (unsigned long long)i < nb; ++i) printf("  (%.9g / %.9g)", g[2 * i], g[2 * i + 1]);
        printf("\n");
    }
    return 0;
}
