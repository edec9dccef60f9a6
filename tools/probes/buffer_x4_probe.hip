// Probe (round 6): buffer_load_dwordx4 / dwordx2 through a raw buffer descriptor on gfx950 --
//   (1) byte offsets that are dword- but not 16-byte aligned;  (2) a load that straddles num_records: are the in-range dwords
//   returned and the out-of-range ones zero (per-dword range check), or is the whole load dropped?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/buffer_x4_probe.hip -o tools/probes/_build/buffer_x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k(const float *row, float *o4, float *o2, int n, int first)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)row, 0, n * 4, 0x00020000);
    const int lane = threadIdx.x;
    const int x = first + 4 * lane;          // lane loads columns x .. x+3 (x negative: a huge unsigned offset)
    const v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, x * 4, 0, 0));
    o4[4 * lane + 0] = v.x; o4[4 * lane + 1] = v.y; o4[4 * lane + 2] = v.z; o4[4 * lane + 3] = v.w;
    const int x2 = first + 2 * lane;
    const v2f w = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, x2 * 4, 0, 0));
    o2[2 * lane + 0] = w.x; o2[2 * lane + 1] = w.y;
}
int main()
{
    const int n = 61;   // 61 floats: with first = 0 the x4 load of columns 60..63 straddles the end (1 in range, 3 out)
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 1000.0f + i;
    float *d, *o4, *o2;
    hipMalloc(&d, 1024 + 64); hipMalloc(&o4, 256 * 4); hipMalloc(&o2, 128 * 4);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    const int firsts[] = {0, 1, 2, 3, -2};
    for (int first : firsts) {
        k<<<1, 64>>>(d + 3, o4, o2, n, first);   // the row starts 12 bytes into the allocation: offset 0 is only dword-aligned
        std::vector<float> r4(256), r2(128);
        hipMemcpy(r4.data(), o4, 1024, hipMemcpyDeviceToHost);
        hipMemcpy(r2.data(), o2, 512, hipMemcpyDeviceToHost);
        printf("first column %2d:\n  x4:", first);
        for (int i = 0; i < 72; ++i) {
            const int x = first + i;
            printf("%s%c%g", (i % 4 == 0) ? " |" : "", ' ', r4[i] == 0.0f ? 0.0f : r4[i] - 1003.0f);   // prints the column index the value came from
            (void)x;
        }
        printf("\n  x2:");
        for (int i = 56; i < 72; ++i) printf("%s %g", (i % 2 == 0) ? " |" : "", r2[i] == 0.0f ? 0.0f : r2[i] - 1003.0f);
        printf("\n");
    }
    return 0;
}
