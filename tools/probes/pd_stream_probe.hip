// PD_TV's ADDRESS STREAM without its arithmetic, as a testbed for access patterns (round 4).  tools/probes/hbm_copy_probe.hip showed
// that the stream of the shipped K = 3 tiling alone -- no arithmetic -- takes as long as the kernel itself on most boxes (10.4 vs
// 10.06 ms per launch), and still 9.7 ms with every halo access aliased away (compulsory traffic only: 36 GB at 3.7-4.0 TB/s, where a
// flat 9-stream dword copy moves them at 5.9 TB/s).  So the PATTERN binds.  This probe varies what a z-march kernel can choose:
//   WX x WY waves per workgroup, RY rows per lane, V columns per lane (dword / dwordx2 / dwordx4 rows), halo on / off,
//   prefetch of the next plane before the current plane's stores (PRE), tile order inside an XCD (row-major / column-major),
//   z-chunks per launch, workgroups per CU (dynamic LDS).
// Rate = COMPULSORY bytes (36 B per voxel) / time.  build: hipcc --offload-arch=gfx950 -O3 -w -o _build/pd_stream_probe pd_stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

struct Args {
    const float *in[5];
    float *out[4];
    int n, nz, gx, gy, zchunk, tiles_per_xcd, order;
};

template <int V> struct Vec { typedef float t __attribute__((ext_vector_type(V))); };
template <> struct Vec<1> { typedef float t; };
template <int V> __device__ __forceinline__ float hsum(typename Vec<V>::t v)
{
    if constexpr (V == 1) return v;
    else if constexpr (V == 2) return v.x + v.y;
    else return v.x + v.y + v.z + v.w;
}

// HALO: K = 3 halo rows / columns either side (re-read by the neighbours) or none (compulsory traffic only)
template <int WX, int WY, int RY, int V, bool HALO, bool PRE>
__global__ __launch_bounds__(64 * WX * WY) void march(Args a)
{
    typedef typename Vec<V>::t vt;
    constexpr int H = HALO ? 3 : 0, NR = RY + 2 * H, OUTC = 64 * V - 2 * H;  // output columns per wave
    const int j = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
    const int tq = xcd * a.tiles_per_xcd + (j % a.tiles_per_xcd), chunk = j / a.tiles_per_xcd;
    if (tq >= a.gx * a.gy) return;
    const int xb = a.order ? tq / a.gy : tq % a.gx, yb = a.order ? tq % a.gy : tq / a.gx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = a.n;
    int x = (xb * WX + (wave % WX)) * OUTC - H + lane * V;
    x = min(max(x, 0), n - V);
    x &= ~(V - 1);                                   // aligned vector columns
    const int y0 = (yb * WY + (wave / WX)) * RY;
    const int z0 = chunk * a.zchunk, z1 = min(z0 + a.zchunk, a.nz);
    const bool emit = !HALO || (lane * V >= H && lane * V < 64 * V - H);
    auto load_plane = [&](int z, vt (&u)[NR], vt (&p)[3][NR], vt (&f)[NR]) {
        const size_t pl = (size_t)z * n * n;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const size_t o = pl + (size_t)min(max(y0 + r - H, 0), n - 1) * n + x;
            u[r] = *(const vt *)(a.in[0] + o);
            p[0][r] = *(const vt *)(a.in[1] + o); p[1][r] = *(const vt *)(a.in[2] + o); p[2][r] = *(const vt *)(a.in[3] + o);
            f[r] = *(const vt *)(a.in[4] + o);
        }
    };
    auto sum_plane = [&](const vt (&u)[NR], const vt (&p)[3][NR], const vt (&f)[NR]) {
        vt s = u[0];
#pragma unroll
        for (int r = 0; r < NR; ++r) s += u[r] + p[0][r] + p[1][r] + p[2][r] + f[r];
        return s;
    };
    auto store_plane = [&](int z, vt s) {
        if (!emit || z < z0) return;
        const size_t pl = (size_t)z * n * n;
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const int y = y0 + r;
            if (y < n) {
                const size_t o = pl + (size_t)y * n + x;
                *(vt *)(a.out[0] + o) = s; *(vt *)(a.out[1] + o) = s; *(vt *)(a.out[2] + o) = s; *(vt *)(a.out[3] + o) = s;
            }
        }
    };
    const int zs = max(z0 - H, 0);
    if constexpr (!PRE) {
        for (int z = zs; z < z1; ++z) {
            vt u[NR], p[3][NR], f[NR];
            load_plane(z, u, p, f);
            const vt s = sum_plane(u, p, f);
            __syncthreads();
            store_plane(z, s);
        }
    } else {
        vt u[NR], p[3][NR], f[NR];
        load_plane(zs, u, p, f);
        for (int z = zs; z < z1; ++z) {
            const vt s = sum_plane(u, p, f);          // waits for plane z
            if (z + 1 < z1) load_plane(z + 1, u, p, f);  // plane z + 1 in flight ...
            __syncthreads();
            store_plane(z, s);                         // ... while plane z is stored
        }
    }
}


// BLOCK loads: one instruction fetches VX rows x 64 columns (VX dwords per lane), where the shipped kernel fetches 1 row x 64
// columns -- same bytes per wave and plane from 1/VX the instructions.  (The kernel would then transpose VX x VX blocks
// between lanes to keep "lane = column": PAT 1 places the VX rows of a block in VX adjacent lanes (in-quad DPP transpose),
// PAT 0 in lanes 64/VX apart (v_permlane swaps).)  Halo of 3 rows, 4 columns either side (aligned vectors): 56 columns out.
template <int WX, int WY, int RY, int VX, int PAT>
__global__ __launch_bounds__(64 * WX * WY) void march_blk(Args a)
{
    typedef typename Vec<VX>::t vt;
    constexpr int H = 3, HX = 4, NR = RY + 2 * H, NI = (NR + VX - 1) / VX, NO = RY / VX, OUTC = 64 - 2 * HX;
    const int j = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
    const int tq = xcd * a.tiles_per_xcd + (j % a.tiles_per_xcd), chunk = j / a.tiles_per_xcd;
    if (tq >= a.gx * a.gy) return;
    const int xb = tq % a.gx, yb = tq / a.gx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = a.n;
    const int sub = PAT ? lane % VX : lane / (64 / VX);        // which row of the block this lane fetches
    const int cx = (PAT ? lane / VX : lane % (64 / VX)) * VX;  // first of its VX columns inside the wave's 64
    int x = (xb * WX + (wave % WX)) * OUTC - HX + cx;
    const bool emit = cx >= HX && cx < 64 - HX && x < n;
    x = min(max(x, 0), n - VX);
    const int y0 = (yb * WY + (wave / WX)) * RY;
    const int z0 = chunk * a.zchunk, z1 = min(z0 + a.zchunk, a.nz);
    for (int z = max(z0 - H, 0); z < z1; ++z) {
        const size_t pl = (size_t)z * n * n;
        vt u[NI], p[3][NI], f[NI];
#pragma unroll
        for (int r = 0; r < NI; ++r) {
            const size_t o = pl + (size_t)min(max(y0 - H + r * VX + sub, 0), n - 1) * n + x;
            u[r] = *(const vt *)(a.in[0] + o);
            p[0][r] = *(const vt *)(a.in[1] + o); p[1][r] = *(const vt *)(a.in[2] + o); p[2][r] = *(const vt *)(a.in[3] + o);
            f[r] = *(const vt *)(a.in[4] + o);
        }
        vt s = u[0];
#pragma unroll
        for (int r = 0; r < NI; ++r) s += u[r] + p[0][r] + p[1][r] + p[2][r] + f[r];
        __syncthreads();
        if (emit && z >= z0) {
#pragma unroll
            for (int r = 0; r < NO; ++r) {
                const int y = y0 + r * VX + sub;
                if (y < n) {
                    const size_t o = pl + (size_t)y * n + x;
                    *(vt *)(a.out[0] + o) = s; *(vt *)(a.out[1] + o) = s; *(vt *)(a.out[2] + o) = s; *(vt *)(a.out[3] + o) = s;
                }
            }
        }
    }
}

// Y-HALO SHARED INSIDE THE WORKGROUP: a wave fetches the 3 halo rows of a side only where that side is the workgroup's edge;
// the rows it would re-fetch from a vertical neighbour IN the workgroup come through LDS (each wave leaves its first and last
// 3 rows there; one barrier).  Requests per wave and plane: 5 arrays x (8 + 3 or 8 + 0 rows) instead of 5 x 14.
template <int WX, int WY, int RY>
__global__ __launch_bounds__(64 * WX * WY) void march_share(Args a)
{
    constexpr int H = 3, OUTC = 64 - 2 * H;
    __shared__ float edge[5][WX * WY][2 * H][64];
    const int j = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
    const int tq = xcd * a.tiles_per_xcd + (j % a.tiles_per_xcd), chunk = j / a.tiles_per_xcd;
    if (tq >= a.gx * a.gy) return;
    const int xb = tq % a.gx, yb = tq / a.gx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wy = wave / WX;
    const int n = a.n;
    int x = (xb * WX + (wave % WX)) * OUTC - H + lane;
    x = min(max(x, 0), n - 1);
    const int y0 = (yb * WY + wy) * RY;
    const int z0 = chunk * a.zchunk, z1 = min(z0 + a.zchunk, a.nz);
    const bool emit = lane >= H && lane < 64 - H;
    const bool top = (wy == 0), bot = (wy == WY - 1);
    for (int z = max(z0 - H, 0); z < z1; ++z) {
        const size_t pl = (size_t)z * n * n;
        float v[5][RY + 2 * H];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
#pragma unroll
            for (int r = 0; r < RY + 2 * H; ++r) {
                const bool mine = (r >= H && r < RY + H) || (r < H && top) || (r >= RY + H && bot);  // wave-uniform
                v[k][r] = 0.0f;
                if (mine) v[k][r] = a.in[k][pl + (size_t)min(max(y0 + r - H, 0), n - 1) * n + x];
            }
#pragma unroll
            for (int r = 0; r < H; ++r) {
                if (!top) edge[k][wave][r][lane] = v[k][H + r];             // my first rows, for the wave above
                if (!bot) edge[k][wave][H + r][lane] = v[k][RY + r];        // my last rows, for the wave below
            }
        }
        __syncthreads();
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
#pragma unroll
            for (int r = 0; r < H; ++r) {
                if (!top) v[k][r] = edge[k][wave - WX][H + r][lane];
                if (!bot) v[k][RY + H + r] = edge[k][wave + WX][r][lane];
            }
#pragma unroll
            for (int r = 0; r < RY + 2 * H; ++r) s += v[k][r];
        }
        __syncthreads();
        if (emit && z >= z0) {
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const int y = y0 + r;
                if (y < n) {
                    const size_t o = pl + (size_t)y * n + x;
                    a.out[0][o] = s; a.out[1][o] = s; a.out[2][o] = s; a.out[3][o] = s;
                }
            }
        }
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

static Args g;
static size_t g_vox;

template <int WX, int WY, int RY, int V, bool HALO, bool PRE>
static void go(int lds_kib, int chunks, int order)
{
    constexpr int H = HALO ? 3 : 0, OUTC = 64 * V - 2 * H;
    Args a = g;
    a.gx = ((a.n + OUTC - 1) / OUTC + WX - 1) / WX;
    a.gy = (a.n + WY * RY - 1) / (WY * RY);
    a.tiles_per_xcd = (a.gx * a.gy + 7) / 8;
    a.zchunk = (a.nz + chunks - 1) / chunks;
    a.order = order;
    const unsigned blocks = 8u * a.tiles_per_xcd * chunks;
    const size_t dyn = (size_t)lds_kib * 1024;
    auto kern = march<WX, WY, RY, V, HALO, PRE>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    const double ms = time_ms([&] { kern<<<blocks, 64 * WX * WY, dyn>>>(a); });
    printf("%dx%d waves, %2d rows/lane, %d col/lane, halo %d, prefetch %d, order %s, %3d KiB LDS/WG, %2d chunks: %7.3f ms  %6.1f GB/s compulsory (%.3f of 8 TB/s)\n",
           WX, WY, RY, V, (int)HALO, (int)PRE, order ? "col" : "row", lds_kib, chunks, ms, 36.0 * g_vox / ms / 1e6, 36.0 * g_vox / ms / 1e6 / 8000.0);
    fflush(stdout);
}

template <int WX, int WY, int RY, int VX, int PAT>
static void go_blk(int lds_kib, int chunks)
{
    constexpr int OUTC = 56;
    Args a = g;
    a.gx = ((a.n + OUTC - 1) / OUTC + WX - 1) / WX;
    a.gy = (a.n + WY * RY - 1) / (WY * RY);
    a.tiles_per_xcd = (a.gx * a.gy + 7) / 8;
    a.zchunk = (a.nz + chunks - 1) / chunks;
    a.order = 0;
    const unsigned blocks = 8u * a.tiles_per_xcd * chunks;
    const size_t dyn = (size_t)lds_kib * 1024;
    auto kern = march_blk<WX, WY, RY, VX, PAT>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    const double ms = time_ms([&] { kern<<<blocks, 64 * WX * WY, dyn>>>(a); });
    printf("block loads %d rows x 64 columns per instruction (%s), %dx%d waves, %2d rows/lane, halo 3 rows / 4 columns, %3d KiB LDS/WG, %2d chunks: %7.3f ms  %6.1f GB/s compulsory\n",
           VX, PAT ? "rows of a block in adjacent lanes" : "rows of a block in lane groups", WX, WY, RY, lds_kib, chunks, ms, 36.0 * g_vox / ms / 1e6);
    fflush(stdout);
}

template <int WX, int WY, int RY>
static void go_share(int lds_kib, int chunks)
{
    constexpr int OUTC = 58;
    Args a = g;
    a.gx = ((a.n + OUTC - 1) / OUTC + WX - 1) / WX;
    a.gy = (a.n + WY * RY - 1) / (WY * RY);
    a.tiles_per_xcd = (a.gx * a.gy + 7) / 8;
    a.zchunk = (a.nz + chunks - 1) / chunks;
    a.order = 0;
    const unsigned blocks = 8u * a.tiles_per_xcd * chunks;
    const size_t stat = 5 * WX * WY * 6 * 64 * 4, dyn = (size_t)lds_kib * 1024 > stat ? (size_t)lds_kib * 1024 - stat : 0;
    auto kern = march_share<WX, WY, RY>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    const double ms = time_ms([&] { kern<<<blocks, 64 * WX * WY, dyn>>>(a); });
    printf("y halo shared through LDS inside the workgroup, %dx%d waves, %2d rows/lane, %3d KiB LDS/WG, %2d chunks: %7.3f ms  %6.1f GB/s compulsory\n",
           WX, WY, RY, lds_kib, chunks, ms, 36.0 * g_vox / ms / 1e6);
    fflush(stdout);
}

int main()
{
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 1024, nz = 1024;
    g_vox = (size_t)n * n * nz;
    const size_t per = g_vox * 4 + 69888;
    char *base;
    if (hipMalloc(&base, per * 9 + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(base, 0, per * 9);
    for (int k = 0; k < 5; ++k) g.in[k] = (const float *)(base + per * k);
    for (int k = 0; k < 4; ++k) g.out[k] = (float *)(base + per * (5 + k));
    g.n = n; g.nz = nz;
    printf("# the kernel's own tiling and occupancy\n");
    go<2, 2, 8, 1, true, false>(80, 32, 0);
    go<2, 2, 8, 1, false, false>(80, 32, 0);
    printf("# the same tile (halo included) from wider instructions\n");
    go_blk<2, 2, 8, 2, 1>(80, 32);
    go_blk<2, 2, 8, 2, 0>(80, 32);
    go_blk<2, 2, 8, 4, 1>(80, 32);
    go_blk<2, 2, 8, 4, 0>(80, 32);
    go_blk<2, 2, 10, 2, 1>(80, 32);
    go_blk<2, 2, 10, 4, 1>(80, 32);
    go_blk<2, 2, 8, 2, 1>(0, 32);
    go_blk<2, 2, 8, 4, 1>(0, 32);
    printf("# y halo from the vertical neighbour in the workgroup (LDS) instead of from memory\n");
    go_share<2, 2, 8>(80, 32);
    go_share<1, 4, 8>(80, 32);
    go_share<2, 4, 8>(160, 32);
    go_share<1, 8, 8>(160, 32);
    go_share<2, 2, 8>(40, 32);
    printf("# more bytes in flight: the next plane requested before the current one is stored\n");
    go<2, 2, 8, 1, true, true>(80, 32, 0);
    go<2, 2, 8, 1, false, true>(80, 32, 0);
    go<2, 2, 8, 1, false, true>(0, 32, 0);
    printf("# wider lanes (same tile area per wave: rows/lane halves as columns/lane double), no halo\n");
    go<2, 2, 4, 2, false, false>(80, 32, 0);
    go<2, 2, 2, 4, false, false>(80, 32, 0);
    go<2, 2, 8, 4, false, false>(80, 32, 0);
    go<2, 2, 8, 4, false, false>(0, 32, 0);
    printf("# workgroup shapes and tile order, no halo\n");
    go<4, 1, 8, 1, false, false>(80, 32, 0);
    go<1, 4, 8, 1, false, false>(80, 32, 0);
    go<2, 2, 8, 1, false, false>(80, 32, 1);
    go<2, 2, 16, 1, false, false>(80, 32, 0);
    printf("# z-chunks per launch, no halo\n");
    go<2, 2, 8, 1, false, false>(80, 8, 0);
    go<2, 2, 8, 1, false, false>(80, 64, 0);
    go<2, 2, 8, 1, false, false>(80, 128, 0);
    printf("# occupancy, no halo\n");
    go<2, 2, 8, 1, false, false>(150, 32, 0);
    go<2, 2, 8, 1, false, false>(50, 32, 0);
    go<2, 2, 8, 1, false, false>(0, 32, 0);
    return 0;
}
