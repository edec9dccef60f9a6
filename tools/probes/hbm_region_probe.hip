// Where in HBM is a buffer fast?  (round 4; tools/probes/pd_stream_probe.hip "place" found the same PD_TV request stream 25 %
// faster on the arenas a process allocates LAST.)  The device memory is carved into CH-GB chunks in allocation order; on
// each: a flat float4 copy (first half -> second half), a dword 5-read / 4-write mix like PD_TV's, and a strided "rows of
// planes" read.  Printed per chunk with its virtual address.  build: hipcc --offload-arch=gfx950 -O3 -w -o _build/hbm_region_probe hbm_region_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float v4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy4(const v4 *__restrict__ a, v4 *__restrict__ b, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = a[i];
}
// 5 reads + 4 writes of dwords, nine sub-arrays of the chunk
__global__ __launch_bounds__(256) void mix9(float *base, size_t per)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= per) return;
    const float s = base[i] + base[per + i] + base[2 * per + i] + base[3 * per + i] + base[4 * per + i];
    base[5 * per + i] = s; base[6 * per + i] = s; base[7 * per + i] = s; base[8 * per + i] = s;
}
__global__ __launch_bounds__(256) void readonly4(const v4 *__restrict__ a, float *sink, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const v4 v = a[i]; if (v.x == 123.456f) sink[0] = v.y; }
}
__global__ __launch_bounds__(256) void writeonly4(v4 *__restrict__ a, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = v4{1, 2, 3, 4};
}

// a z-march over ONE array of planes of 1024 x 1024 floats: a workgroup (4 waves as 2 x 2) owns 128 columns x 16 rows (+3
// halo rows either side) and walks the planes of its z-chunk, reading 22 rows per wave-column and plane and writing 8 rows per
// wave into the second half of the chunk -- PD_TV's access shape with one input and one output stream
__global__ __launch_bounds__(256) void zmarch1(const float *__restrict__ a, float *__restrict__ b, int nz, int zchunk)
{
    const int n = 1024, gx = 8, gy = 64;
    const int tile = blockIdx.x % (gx * gy), chunk = blockIdx.x / (gx * gy);
    const int xb = tile % gx, yb = tile / gx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = xb * 128 + (wave & 1) * 64 + lane, y0 = yb * 16 + (wave >> 1) * 8;
    const int z0 = chunk * zchunk, z1 = min(z0 + zchunk, nz);
    for (int z = z0; z < z1; ++z) {
        const size_t pl = (size_t)z * n * n;
        float s = 0.0f;
#pragma unroll
        for (int r = -3; r < 11; ++r) s += a[pl + (size_t)min(max(y0 + r, 0), n - 1) * n + x];
#pragma unroll
        for (int r = 0; r < 8; ++r) b[pl + (size_t)(y0 + r) * n + x] = s;
    }
}

static hipEvent_t e0, e1;
template <typename F> static double time_ms(F &&f, int reps = 3)
{
    f(); hipDeviceSynchronize();
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char **argv)
{
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double chunk_gb = argc > 1 ? atof(argv[1]) : 9.0;
    const int maxchunks = argc > 2 ? atoi(argv[2]) : 64;
    const size_t bytes = (size_t)(chunk_gb * 1e9) / (9 * 4096) * (9 * 4096);
    std::vector<char *> ch;
    for (int k = 0; k < maxchunks; ++k) {
        char *p;
        size_t fr, tot; hipMemGetInfo(&fr, &tot);
        if (fr < bytes + (size_t)6e9) break;
        if (hipMalloc(&p, bytes) != hipSuccess) break;
        hipMemset(p, 0, bytes);
        ch.push_back(p);
    }
    size_t fr, tot; hipMemGetInfo(&fr, &tot);
    printf("%zu chunks of %.2f GB held (%.1f of %.1f GB free)\n", ch.size(), bytes / 1e9, fr / 1e9, tot / 1e9);
    float *sink; hipMalloc(&sink, 64);
    for (size_t k = 0; k < ch.size(); ++k) {
        const size_t n4 = bytes / 32;  // float4 elements per half
        const double tc = time_ms([&] { copy4<<<(unsigned)((n4 + 255) / 256), 256>>>((const v4 *)ch[k], (v4 *)(ch[k] + bytes / 2), n4); });
        const size_t per = bytes / 36;
        const double tm = time_ms([&] { mix9<<<(unsigned)((per + 255) / 256), 256>>>((float *)ch[k], per); });
        const size_t nr = bytes / 16;
        const double tr = time_ms([&] { readonly4<<<(unsigned)((nr + 255) / 256), 256>>>((const v4 *)ch[k], sink, nr); });
        const double tw = time_ms([&] { writeonly4<<<(unsigned)((nr + 255) / 256), 256>>>((v4 *)ch[k], nr); });
        const int nzp = (int)(bytes / 2 / (4u << 20));
        const double tz = time_ms([&] { zmarch1<<<512 * 16, 256>>>((const float *)ch[k], (float *)(ch[k] + bytes / 2), nzp, (nzp + 15) / 16); });
        printf("z-march %7.1f GB/s  ", 2.0 * nzp * (4u << 20) / tz / 1e6);
        printf("chunk %2zu at %p (allocated after %6.1f GB): copy %7.1f GB/s   mix 5r/4w dword %7.1f GB/s   read %7.1f GB/s   write %7.1f GB/s\n", k, (void *)ch[k],
               k * bytes / 1e9, bytes / tc / 1e6, bytes / tm / 1e6, bytes / tr / 1e6, bytes / tw / 1e6);
        fflush(stdout);
    }
    return 0;
}
