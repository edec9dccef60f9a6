// Exhaustive check of r = 1.0f / sqrtf(x) (two roundings, the reference's PD_TV projection, primal_dual...cu:196-203) computed
// (a) as shipped: v_rsq + coupled Newton + residual correction for the root, then v_rcp + Newton + residual correction;
// (b) without the v_rcp: the refined half-reciprocal-root h of the first stage doubled is the start of the reciprocal.
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -fno-fast-math tools/probes/rsqrt_chain_probe.hip -o /tmp/rc && /tmp/rc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ void chains(float x, float &ra, float &rb)
{
    const float r = __builtin_amdgcn_rsqf(x);
    float q = x * r, h = 0.5f * r;
    const float e = fmaf(-h, q, 0.5f);
    q = fmaf(q, e, q);
    h = fmaf(h, e, h);
    q = fmaf(fmaf(-q, q, x), h, q);
    float y = __builtin_amdgcn_rcpf(q);
    y = fmaf(fmaf(-q, y, 1.0f), y, y);
    ra = fmaf(fmaf(-q, y, 1.0f), y, y);
    float z = h + h;
    z = fmaf(fmaf(-q, z, 1.0f), z, z);
    rb = fmaf(fmaf(-q, z, 1.0f), z, z);
}
__global__ void all(uint32_t lo, uint32_t hi, unsigned long long *bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        float ra, rb;
        chains(x, ra, rb);
        const float w = 1.0f / sqrtf(x);
        if (__float_as_uint(ra) != __float_as_uint(w)) atomicAdd(bad, 1ULL);
        if (__float_as_uint(rb) != __float_as_uint(w)) atomicAdd(bad + 1, 1ULL);
    }
}
int main()
{
    unsigned long long *bad, nb[2];
    (void)hipMalloc(&bad, 16); (void)hipMemset(bad, 0, 16);
    uint32_t lo, hi; float flo = 1.0f, fhi = 1e12f; memcpy(&lo, &flo, 4); memcpy(&hi, &fhi, 4);
    all<<<4096, 256>>>(lo, hi, bad);
    (void)hipMemcpy(nb, bad, 16, hipMemcpyDeviceToHost);
    printf("1/sqrtf(x): %u inputs in [1, 1e12]: mismatches shipped chain %llu, chain without v_rcp %llu\n", hi - lo + 1, nb[0], nb[1]);
    return 0;
}
