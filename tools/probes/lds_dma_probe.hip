// What buffer_load_dwordx4 ... lds does with out-of-range lanes and with inactive lanes (gfx950):
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_dma_probe.hip -o /tmp/lds_dma_probe && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float4 *q, float4 *o, int n)
{
    __shared__ float4 tile[128];
    tile[threadIdx.x] = make_float4(-7.f, -7.f, -7.f, -7.f);
    tile[threadIdx.x + 64] = make_float4(-7.f, -7.f, -7.f, -7.f);
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)q, 0, n * 16, 0x00020000);
    // lane l reads element l - 4 (lanes 0..3: negative offset = out of range), elements >= n out of range too;
    // lanes 40..47 are switched off
    const int lane = threadIdx.x;
    if (lane < 40 || lane >= 48)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)tile, 16, (lane - 4) * 16, 0, 0, 0);
    __syncthreads();
    o[lane] = tile[lane];
    o[lane + 64] = tile[lane + 64];
}
int main()
{
    const int n = 50;   // lanes 54.. read beyond the 50 records
    std::vector<float4> h(64);
    for (int i = 0; i < 64; ++i) h[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f);
    float4 *q, *o;
    hipMalloc(&q, 64 * 16); hipMalloc(&o, 128 * 16);
    hipMemcpy(q, h.data(), 64 * 16, hipMemcpyHostToDevice);
    k<<<1, 64>>>(q, o, n);
    std::vector<float4> r(128);
    hipMemcpy(r.data(), o, 128 * 16, hipMemcpyDeviceToHost);
    for (int i = 0; i < 72; ++i) printf("slot %2d: %g %g %g %g\n", i, r[i].x, r[i].y, r[i].z, r[i].w);
    return 0;
}
