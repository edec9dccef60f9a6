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
#include <cstring>
#include <cstdlib>
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

// "place": does the speed of the stream depend on WHERE the arrays were allocated?  Several arenas held at once, the same
// stream (the kernel's tiling, halo included) timed on each, twice.
static int placement(int narena)
{
    const int n = 1024, nz = 1024;
    g_vox = (size_t)n * n * nz;
    const size_t per = g_vox * 4 + 69888;
    g.n = n; g.nz = nz;
    std::vector<char *> bases;
    for (int k = 0; k < narena; ++k) {
        char *b;
        if (hipMalloc(&b, per * 9 + 4096) != hipSuccess) { printf("arena %d: alloc failed\n", k); break; }
        hipMemset(b, 0, per * 9);
        bases.push_back(b);
    }
    size_t fr, tot; hipMemGetInfo(&fr, &tot);
    printf("%zu arenas of %.1f GB held, %.1f of %.1f GB free\n", bases.size(), per * 9 / 1e9, fr / 1e9, tot / 1e9);
    for (int rep = 0; rep < 2; ++rep)
        for (size_t k = 0; k < bases.size(); ++k) {
            for (int q = 0; q < 5; ++q) g.in[q] = (const float *)(bases[k] + per * q);
            for (int q = 0; q < 4; ++q) g.out[q] = (float *)(bases[k] + per * (5 + q));
            printf("arena %zu at %p: ", k, (void *)bases[k]);
            go<2, 2, 8, 1, true, false>(80, 32, 0);
        }
    return 0;
}

// "ballast B": B GB allocated first, then an arena, then the ballast is freed and a second arena allocated: which is fast?
static int ballast(double gb)
{
    const int n = 1024, nz = 1024;
    g_vox = (size_t)n * n * nz;
    const size_t per = g_vox * 4 + 69888;
    g.n = n; g.nz = nz;
    auto run = [&](const char *name, char *b) {
        for (int q = 0; q < 5; ++q) g.in[q] = (const float *)(b + per * q);
        for (int q = 0; q < 4; ++q) g.out[q] = (float *)(b + per * (5 + q));
        printf("%-44s at %p: ", name, (void *)b);
        go<2, 2, 8, 1, true, false>(80, 32, 0);
    };
    std::vector<char *> bal;
    for (double left = gb; left > 0; left -= 8.0) {  // 8 GB pieces
        char *p; if (hipMalloc(&p, (size_t)8e9) != hipSuccess) break;
        hipMemset(p, 0, (size_t)8e9); bal.push_back(p);
    }
    char *a1, *a2, *a3;
    hipMalloc(&a1, per * 9 + 4096); hipMemset(a1, 0, per * 9);
    run("arena allocated behind the ballast", a1);
    for (char *p : bal) hipFree(p);
    hipDeviceSynchronize();
    run("same arena, ballast freed", a1);
    hipMalloc(&a2, per * 9 + 4096); hipMemset(a2, 0, per * 9);
    run("second arena, allocated after the free", a2);
    hipMalloc(&a3, per * 9 + 4096); hipMemset(a3, 0, per * 9);
    run("third arena", a3);
    run("first arena again", a1);
    return 0;
}

// "skew": inside ONE allocation, the nine arrays at different relative offsets (array k starts at k * (4 GiB + skew)); twice,
// on the first and on a later arena of the process
static int skews()
{
    const int n = 1024, nz = 1024;
    g_vox = (size_t)n * n * nz;
    g.n = n; g.nz = nz;
    const size_t room = g_vox * 4 + (64u << 20);
    std::vector<char *> ar;
    for (int k = 0; k < 5; ++k) { char *b; if (hipMalloc(&b, room * 9) != hipSuccess) break; hipMemset(b, 0, room * 9); ar.push_back(b); }
    const long sk[] = {0, 256, 4096, 69888, 65536 + 4096 + 256, 1 << 20, (1 << 20) + 69888, 3 << 20, (5 << 20) + 4096 * 3 + 512, 17 * 4096 + 128, 2097152 + 69888, 33554432 + 69888};
    for (size_t w : {(size_t)0, ar.size() - 1})
        for (long s : sk) {
            const size_t per = g_vox * 4 + (size_t)s;
            for (int q = 0; q < 5; ++q) g.in[q] = (const float *)(ar[w] + per * q);
            for (int q = 0; q < 4; ++q) g.out[q] = (float *)(ar[w] + per * (5 + q));
            printf("arena %zu, skew %9ld B: ", w, s);
            go<2, 2, 8, 1, true, false>(80, 32, 0);
        }
    return 0;
}

// "mixarena": device memory carved into 4.4 GB pieces in allocation order, each scored alone with a one-array z-march (first
// half -> second half); then PD_TV's stream with its nine arrays taken from the nine FASTEST pieces, the nine SLOWEST, and nine
// spread evenly over the allocation order.  Is the speed a property of each piece that adds up?
__global__ __launch_bounds__(256) void zmarch1(const float *__restrict__ a, float *__restrict__ b, int nz, int zchunk)
{
    const int n = 1024, gx = 8, gy = 64;
    const int tile = blockIdx.x % (gx * gy), chunk = blockIdx.x / (gx * gy);
    const int xb = tile % gx, yb = tile / gx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = xb * 128 + (wave & 1) * 64 + lane, y0 = yb * 16 + (wave >> 1) * 8;
    const int z1 = min((chunk + 1) * zchunk, nz);
    for (int z = chunk * zchunk; z < z1; ++z) {
        const size_t pl = (size_t)z * n * n;
        float s = 0.0f;
#pragma unroll
        for (int r = -3; r < 11; ++r) s += a[pl + (size_t)min(max(y0 + r, 0), n - 1) * n + x];
#pragma unroll
        for (int r = 0; r < 8; ++r) b[pl + (size_t)(y0 + r) * n + x] = s;
    }
}
static int mixarena()
{
    const int n = 1024, nz = 1024;
    g_vox = (size_t)n * n * nz;
    g.n = n; g.nz = nz;
    const size_t piece = g_vox * 4 + (1u << 20);
    std::vector<char *> pc;
    for (;;) {
        size_t fr, tot; hipMemGetInfo(&fr, &tot);
        if (fr < piece + (size_t)8e9) break;
        char *p; if (hipMalloc(&p, piece) != hipSuccess) break;
        hipMemset(p, 0, piece); pc.push_back(p);
    }
    std::vector<std::pair<double, int>> sc;
    for (size_t k = 0; k < pc.size(); ++k) {
        const double ms = time_ms([&] { zmarch1<<<512 * 16, 256>>>((const float *)pc[k], (float *)(pc[k] + piece / 2), 512, 32); });
        const double gbs = 2.0 * 512 * (4u << 20) / ms / 1e6;
        // the same piece in 1 GiB quarters (128 planes in, 128 out)
        double q[4];
        for (int i = 0; i < 4; ++i) {
            const char *b = pc[k] + (size_t)i * (1u << 30);
            const double m2 = time_ms([&] { zmarch1<<<512 * 16, 256>>>((const float *)b, (float *)(b + (512u << 20)), 128, 8); });
            q[i] = 2.0 * 128 * (4u << 20) / m2 / 1e6;
        }
        printf("piece %2zu (after %5.1f GB): %6.0f GB/s   quarters %6.0f %6.0f %6.0f %6.0f\n", k, k * piece / 1e9, gbs, q[0], q[1], q[2], q[3]);
        sc.push_back({gbs, (int)k});
    }
    std::sort(sc.begin(), sc.end());
    auto run = [&](const char *name, const std::vector<int> &idx) {
        for (int q = 0; q < 5; ++q) g.in[q] = (const float *)pc[idx[q]];
        for (int q = 0; q < 4; ++q) g.out[q] = (float *)pc[idx[5 + q]];
        printf("%-28s pieces", name);
        for (int i : idx) printf(" %d", i);
        printf(": ");
        go<2, 2, 8, 1, true, false>(80, 32, 0);
    };
    const int m = (int)sc.size();
    std::vector<int> fast, slow, spread, run9;
    for (int i = 0; i < 9; ++i) { slow.push_back(sc[i].second); fast.push_back(sc[m - 1 - i].second); spread.push_back(i * (m - 1) / 8); run9.push_back(i); }
    for (int rep = 0; rep < 2; ++rep) { run("nine fastest", fast); run("nine slowest", slow); run("spread over the device", spread); run("first nine (one arena)", run9); }
    return 0;
}

// "vmm G_MB": the arena built from physical chunks of G MB (hipMemCreate) mapped into one reserved address range, in
// allocation order and in a shuffled order: does the large-scale physical layout of the block decide the speed?
static int vmm(size_t g_mb, int narena)
{
    const int n = 1024, nz = 1024;
    g_vox = (size_t)n * n * nz;
    g.n = n; g.nz = nz;
    const size_t per = g_vox * 4 + 69888;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) { printf("no VMM\n"); return 1; }
    const size_t G = (g_mb << 20) / gran * gran;
    const size_t total = (per * 9 + G - 1) / G * G;
    const size_t nch = total / G;
    printf("granularity %zu KB, chunk %zu MB, %zu chunks per arena\n", gran >> 10, G >> 20, nch);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    auto run = [&](const char *name, char *b) {
        for (int q = 0; q < 5; ++q) g.in[q] = (const float *)(b + per * q);
        for (int q = 0; q < 4; ++q) g.out[q] = (float *)(b + per * (5 + q));
        printf("%-36s: ", name);
        go<2, 2, 8, 1, true, false>(80, 32, 0);
    };
    for (int k = 0; k < narena; ++k) {
        std::vector<hipMemGenericAllocationHandle_t> h(nch);
        for (size_t i = 0; i < nch; ++i)
            if (hipMemCreate(&h[i], G, &prop, 0) != hipSuccess) { printf("hipMemCreate failed at chunk %zu\n", i); return 1; }
        for (int mode = 0; mode < 3; ++mode) {
            void *va = nullptr;
            if (hipMemAddressReserve(&va, total, 0, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); return 1; }
            std::vector<size_t> order(nch);
            for (size_t i = 0; i < nch; ++i) order[i] = i;
            if (mode == 1) { unsigned r = 12345u + k; for (size_t i = nch - 1; i > 0; --i) { r = r * 1664525u + 1013904223u; std::swap(order[i], order[(r >> 8) % (i + 1)]); } }
            if (mode == 2) { const size_t st = 9; std::vector<size_t> o2; for (size_t a = 0; a < st; ++a) for (size_t i = a; i < nch; i += st) o2.push_back(i); order = o2; }  // stride 9: neighbours in VA are far apart physically
            for (size_t i = 0; i < nch; ++i)
                if (hipMemMap((char *)va + i * G, G, 0, h[order[i]], 0) != hipSuccess) { printf("map failed\n"); return 1; }
            if (hipMemSetAccess(va, total, &acc, 1) != hipSuccess) { printf("access failed\n"); return 1; }
            hipMemset(va, 0, per * 9);
            char nm[64]; snprintf(nm, sizeof nm, "arena %d, chunks %s", k, mode == 0 ? "in order" : mode == 1 ? "shuffled" : "stride 9");
            run(nm, (char *)va);
            run(nm, (char *)va);
            hipDeviceSynchronize();
            hipMemUnmap(va, total);
            hipMemAddressFree(va, total);
        }
        // keep the handles: the next arena takes the next physical memory
    }
    return 0;
}

// "vmmcmp": inside ONE process, arenas built from 2 / 64 / 1024 MB chunks and a plain hipMalloc arena, alternately, all
// held until the end (so every arena has its own physical memory): chunk size or position?
static int vmmcmp()
{
    const int n = 1024, nz = 1024;
    g_vox = (size_t)n * n * nz;
    g.n = n; g.nz = nz;
    const size_t per = g_vox * 4 + 69888;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    auto run = [&](const char *name, char *b) {
        for (int q = 0; q < 5; ++q) g.in[q] = (const float *)(b + per * q);
        for (int q = 0; q < 4; ++q) g.out[q] = (float *)(b + per * (5 + q));
        printf("%-36s: ", name);
        go<2, 2, 8, 1, true, false>(80, 32, 0);
    };
    const size_t gs[] = {64, 0, 2, 1024, 64, 0, 2};
    int k = 0;
    for (size_t g_mb : gs) {
        char nm[64];
        size_t fr, tot; hipMemGetInfo(&fr, &tot);
        if (fr < per * 9 + (size_t)8e9) break;
        if (g_mb == 0) {
            char *b; if (hipMalloc(&b, per * 9 + 4096) != hipSuccess) break;
            hipMemset(b, 0, per * 9);
            snprintf(nm, sizeof nm, "arena %d: hipMalloc", k++);
            run(nm, b);
            continue;
        }
        const size_t G = g_mb << 20, total = (per * 9 + G - 1) / G * G, nch = total / G;
        void *va = nullptr;
        if (hipMemAddressReserve(&va, total, 0, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); return 1; }
        for (size_t i = 0; i < nch; ++i) {
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, G, &prop, 0) != hipSuccess) { printf("create failed\n"); return 1; }
            if (hipMemMap((char *)va + i * G, G, 0, h, 0) != hipSuccess) { printf("map failed\n"); return 1; }
        }
        hipMemSetAccess(va, total, &acc, 1);
        hipMemset(va, 0, per * 9);
        snprintf(nm, sizeof nm, "arena %d: %zu MB chunks", k++, g_mb);
        run(nm, (char *)va);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "vmmcmp")) { hipEventCreate(&e0); hipEventCreate(&e1); return vmmcmp(); }
    if (argc > 2 && !strcmp(argv[1], "vmm")) { hipEventCreate(&e0); hipEventCreate(&e1); return vmm((size_t)atol(argv[2]), argc > 3 ? atoi(argv[3]) : 5); }
    if (argc > 1 && !strcmp(argv[1], "mixarena")) { hipEventCreate(&e0); hipEventCreate(&e1); return mixarena(); }
    if (argc > 1 && !strcmp(argv[1], "skew")) { hipEventCreate(&e0); hipEventCreate(&e1); return skews(); }
    if (argc > 2 && !strcmp(argv[1], "ballast")) { hipEventCreate(&e0); hipEventCreate(&e1); return ballast(atof(argv[2])); }
    hipEventCreate(&e0); hipEventCreate(&e1);
    if (argc > 1 && !strcmp(argv[1], "place")) return placement(argc > 2 ? atoi(argv[2]) : 5);
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
