// Probe (round 6): buffer_load_dword ... lds (4 bytes per lane, LDS address = M0 base + 4 * lane) on gfx950 --
//   (1) does the destination base reach beyond 64 KiB of the 160 KiB LDS (M0 width)?  (2) out-of-range lanes write zeros?
//   (3) lanes = (column, slice) pairs: 16 columns x 4 slices per instruction land as 16 float4 [col][slice]?
// Both through the builtin (the compiler sets M0) and through the inline-assembly form the back projector uses.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_dma_dword_probe.hip -o tools/probes/_build/lds_dma_dword_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void *lds_ptr;

template <bool ASM>
__global__ void k(const float *vol, float *o, int n, int zstride, unsigned base_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *tile = reinterpret_cast<float *>(smem + base_bytes);
    const int lane = threadIdx.x;
    tile[lane] = -7.0f;
    __syncthreads();
    // 16 columns x 4 slices: lane = col * 4 + slice; column 3 + col of slice `slice`; columns >= n are forced out of range
    const int col = lane >> 2, slice = lane & 3, x = 3 + col;
    const int off = (x < n) ? (slice * zstride + x) * 4 : (int)0x80000000;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)vol, 0, 4 * zstride * 4, 0x00020000);
    if (ASM) {
        const unsigned dst = (unsigned)(uintptr_t)(lds_ptr)tile;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(dst), "v"(off), "s"(r) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)tile, 4, off, 0, 0, 0);
    }
    __syncthreads();
    o[lane] = tile[lane];
}

int main()
{
    const int n = 14, zstride = 32;   // columns 3..13 valid (11 of the 16), 4 slices of 32 floats
    std::vector<float> h(4 * zstride);
    for (int s = 0; s < 4; ++s)
        for (int x = 0; x < zstride; ++x) h[s * zstride + x] = 100.0f * s + x;
    float *v, *o;
    hipMalloc(&v, h.size() * 4); hipMalloc(&o, 64 * 4);
    hipMemcpy(v, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const size_t shm = 150 * 1024;
    hipFuncSetAttribute((const void *)k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipFuncSetAttribute((const void *)k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    const unsigned bases[] = {0u, 40u * 1024, 70u * 1024, 100u * 1024, 140u * 1024};
    for (int a = 0; a < 2; ++a)
        for (unsigned b : bases) {
            hipMemset(o, 0xff, 64 * 4);
            if (a) k<true><<<1, 64, shm>>>(v, o, n, zstride, b);
            else k<false><<<1, 64, shm>>>(v, o, n, zstride, b);
            std::vector<float> r(64);
            hipError_t e = hipMemcpy(r.data(), o, 64 * 4, hipMemcpyDeviceToHost);
            int ok = 0, zero = 0, bad = 0;
            for (int l = 0; l < 64; ++l) {
                const int col = l >> 2, s = l & 3, x = 3 + col;
                const float want = x < n ? 100.0f * s + x : 0.0f;
                if (r[l] == want) (x < n ? ok : zero)++; else ++bad;
            }
            printf("%s base %6u B: %s  in-range lanes right %d/44, out-of-range lanes zero %d/20, wrong %d  (lane 0..7: %g %g %g %g %g %g %g %g; lane 44..47: %g %g %g %g)\n",
                   a ? "inline asm" : "builtin   ", b, hipGetErrorString(e), ok, zero, bad, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[44], r[45], r[46], r[47]);
        }
    return 0;
}
