// PD_TV's request stream (the shipped tiling: 2 x 2 waves, 8 + 6 rows, 58 + 6 columns, five input and four output arrays of 1024^3)
// with the PLANES of the arrays 4 MiB apart -- as shipped -- or 4 MiB + pad apart (tools/probes/plane_pitch_probe.hip: a plane
// pitch that is an odd multiple of 256 B lifts a one-array z-march from 4.83 to 5.09 TB/s on a slow block).  Several arenas held
// at once so that slow and fast blocks are both seen.  "scratch only": U and P1..3 (the library's own arrays) pitched, Input
// (the caller's array) not.   build: hipcc --offload-arch=gfx950 -O3 -w -o _build/pd_pitch_probe pd_pitch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

struct Args {
    const float *in[5];
    float *out[4];
    size_t pin[5], pout[4];   // plane pitch of every array, floats
    int n, nz, gx, gy, zchunk, tiles_per_xcd;
};

__global__ __launch_bounds__(256) void march(Args a)
{
    constexpr int H = 3, NR = 14, OUTC = 58;
    const int j = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
    const int tq = xcd * a.tiles_per_xcd + (j % a.tiles_per_xcd), chunk = j / a.tiles_per_xcd;
    if (tq >= a.gx * a.gy) return;
    const int xb = tq % a.gx, yb = tq / a.gx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = a.n;
    const int x = min(max((xb * 2 + (wave & 1)) * OUTC - H + lane, 0), n - 1);
    const int y0 = (yb * 2 + (wave >> 1)) * 8;
    const int z0 = chunk * a.zchunk, z1 = min(z0 + a.zchunk, a.nz);
    const bool emit = lane >= H && lane < 64 - H;
    for (int z = max(z0 - H, 0); z < z1; ++z) {
        float s = 0.0f;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const size_t o = (size_t)min(max(y0 + r - H, 0), n - 1) * n + x;
#pragma unroll
            for (int k = 0; k < 5; ++k) s += a.in[k][(size_t)z * a.pin[k] + o];
        }
        __syncthreads();
        if (emit && z >= z0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int y = y0 + r;
                if (y < n) {
                    const size_t o = (size_t)y * n + x;
#pragma unroll
                    for (int k = 0; k < 4; ++k) a.out[k][(size_t)z * a.pout[k] + o] = s;
                }
            }
        }
    }
}

int main()
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 1024, nz = 1024;
    const size_t plane = (size_t)n * n, maxpad = 4096;
    const size_t per = (plane + maxpad) * nz * 4 + 69888;          // room for the largest pad
    std::vector<char *> arenas;
    for (int k = 0; k < 6; ++k) {
        size_t fr, tot; hipMemGetInfo(&fr, &tot);
        char *p;
        if (fr < per * 9 + ((size_t)6 << 30) || hipMalloc(&p, per * 9 + 4096) != hipSuccess) break;
        hipMemset(p, 0, per * 9);
        arenas.push_back(p);
    }
    printf("%zu arenas of %.1f GB held\n", arenas.size(), per * 9 / 1e9);
    struct Mode { const char *name; size_t pad; bool input_too; };
    const Mode modes[] = {{"planes 4 MiB apart (shipped)", 0, true}, {"+256 B, scratch only", 64, false}, {"+256 B, all nine", 64, true},
                          {"+768 B, scratch only", 192, false}, {"+69888 B, scratch only", 17472 % 4096 == 0 ? 64 : 64 + 1024, false},
                          {"+512 B, scratch only (control)", 128, false}, {"planes 4 MiB apart (shipped)", 0, true}};
    for (size_t k = 0; k < arenas.size(); ++k) {
        printf("arena %zu:\n", k);
        for (const Mode &m : modes) {
            Args a;
            a.n = n; a.nz = nz;
            a.gx = ((n + 57) / 58 + 1) / 2; a.gy = (n + 15) / 16;
            a.tiles_per_xcd = (a.gx * a.gy + 7) / 8;
            const int chunks = 15;
            a.zchunk = (nz + chunks - 1) / chunks;
            for (int q = 0; q < 5; ++q) { a.in[q] = (const float *)(arenas[k] + per * q); a.pin[q] = plane + ((q == 4 && !m.input_too) ? 0 : m.pad); }
            for (int q = 0; q < 4; ++q) { a.out[q] = (float *)(arenas[k] + per * (5 + q)); a.pout[q] = plane + m.pad; }
            const unsigned blocks = 8u * a.tiles_per_xcd * chunks;
            hipFuncSetAttribute((const void *)march, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            march<<<blocks, 256, 80 * 1024>>>(a);
            std::vector<float> t;
            for (int r = 0; r < 3; ++r) {
                hipEventRecord(e0); march<<<blocks, 256, 80 * 1024>>>(a); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms);
            }
            std::sort(t.begin(), t.end());
            printf("   %-34s %7.3f ms per launch-equivalent\n", m.name, t[1]);
            fflush(stdout);
        }
    }
    return 0;
}
