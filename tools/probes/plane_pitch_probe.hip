// Does the two-speed behaviour of large allocations (docs/kernels/placement.md: a z-march over one array of 4 MB planes runs at
// 4.87 TB/s in some 34 GB blocks and 5.3 TB/s in others) depend on the PLANE PITCH being a power of two?  Several 34 GB blocks are held
// at once; on each the library's placement probe (one input stream, one output stream, PD_TV's tile shape: 2 x 2 waves, 14 rows
// read and 8 written per wave and plane) is timed with the planes 4 MiB apart and with padded pitches.
// build: hipcc --offload-arch=gfx950 -O3 -w -o _build/plane_pitch_probe plane_pitch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void zmarch(const float *__restrict__ a, float *__restrict__ b, int nz, int zchunk, size_t pitch)
{
    const int n = 1024, gx = 8, gy = 64;
    const int tile = (int)blockIdx.x % (gx * gy), chunk = (int)blockIdx.x / (gx * gy);
    const int xb = tile % gx, yb = tile / gx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = xb * 128 + (wave & 1) * 64 + lane, y0 = yb * 16 + (wave >> 1) * 8;
    const int z1 = min((chunk + 1) * zchunk, nz);
    for (int z = chunk * zchunk; z < z1; ++z) {
        const size_t pl = (size_t)z * pitch;
        float s = 0.0f;
#pragma unroll
        for (int r = -3; r < 11; ++r) s += a[pl + (size_t)min(max(y0 + r, 0), n - 1) * n + x];
#pragma unroll
        for (int r = 0; r < 8; ++r) b[pl + (size_t)(y0 + r) * n + x] = s;
    }
}

int main()
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t bytes = (size_t)34360297472ull;
    std::vector<char *> blocks;
    for (int k = 0; k < 5; ++k) {
        char *p;
        size_t fr, tot; hipMemGetInfo(&fr, &tot);
        if (fr < bytes + ((size_t)4 << 30) || hipMalloc(&p, bytes) != hipSuccess) break;
        blocks.push_back(p);
    }
    printf("%zu blocks of %.1f GB held\n", blocks.size(), bytes / 1e9);
    const size_t plane = (size_t)1 << 20;  // floats
    // pads in floats (x 4 = bytes)
    const size_t pads[] = {0, 17472 /*0x11100 B*/, 0x1100 / 4, 0x10100 / 4, 0x11000 / 4, 0x100 / 4, 0x8880 / 4, 0x22200 / 4, 0x33300 / 4, 0x111100 / 4,
                           0x1111100 / 4, 0x5500 / 4, 0x15500 / 4, 0x10000 / 4, 0x30300 / 4, 0x11300 / 4, 0x13100 / 4, 0x7700 / 4, 0x1f00 / 4, 0x11111100 / 4};
    for (int rep = 0; rep < 1; ++rep)
        for (size_t k = 0; k < blocks.size(); ++k) {
            printf("block %zu:", k);
            for (size_t pad : pads) {
                const size_t pitch = plane + pad;
                const int nz = (int)(bytes / 2 / (pitch * 4));
                const int chunks = 16, zchunk = (nz + chunks - 1) / chunks;
                const float *a = (const float *)blocks[k];
                float *b = (float *)(blocks[k] + (size_t)nz * pitch * 4);
                zmarch<<<512 * chunks, 256>>>(a, b, nz, zchunk, pitch);
                float best = 0.0f;
                for (int r = 0; r < 2; ++r) {
                    hipEventRecord(e0);
                    zmarch<<<512 * chunks, 256>>>(a, b, nz, zchunk, pitch);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    best = std::max(best, (float)(2.0 * nz * 4194304.0 / ms / 1e6));
                }
                printf("  +0x%zx: %4.0f", pad * 4, best);
            }
            printf("  GB/s\n");
            fflush(stdout);
        }
    return 0;
}
