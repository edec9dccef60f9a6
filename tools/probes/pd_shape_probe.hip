// PD_TV request-stream probe over TILE SHAPES that were not tried in round 4 (round 5, review item 1).
// Same method as pd_stream_probe.hip: a kernel that issues exactly the loads and stores a PD_TV z-march of the given shape
// would issue (five input arrays, four output arrays of 1024^3, XCD-banded tile order, z-chunks with warm-up planes,
// occupancy pinned with dynamic LDS) and as little else as the shape allows.  ms per launch are normalised to THREE
// iterations ("launch-equivalent": a K = 4 pass is scaled by 3/4) so that every line compares with the shipped 2x2 / 8-row /
// K = 3 tiling measured in the same process.
//   base      the shipped tiling (2 x 2 waves, 8 + 6 rows, 58 + 6 columns, 70 + 32 requests per wave and plane)
//   nohalo    compulsory requests only (40 + 32)
//   align     the shipped tiling with every wave's x origin rounded down to 128 B: same requests, 2 lines per row, not 3
//   half      a wave = 2 x 32 lanes, the halves own vertically adjacent 8-row tiles (16 + 6 rows x 26 + 6 columns per wave)
//   quarter   a wave = 4 x 16 lanes (32 + 6 rows x 10 + 6 columns)
//   spw       STAGE PER WAVE: a workgroup is a pipeline of K waves, wave s runs iteration n+s on the SAME (rows x 58)
//             tile; only wave 0 loads from memory, only wave K-1 stores; U, P1..3 and Input are handed from wave s to
//             wave s+1 through LDS (b128 accesses, single buffer, two barriers per plane).  The per-wave state is ONE
//             stage, so a lane can own 12-16 rows where the shipped kernel (three stages per wave) fits 8: the y halo is
//             amortised over more rows AND over a fourth iteration.  VPR dependent FMAs per row stand in for the arithmetic.
// build: hipcc --offload-arch=gfx950 -O3 -w -o _build/pd_shape_probe pd_shape_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>

struct Args {
    const float *in[5];
    float *out[4];
    int n, nz, gx, gy, zchunk, tiles_per_xcd;
};

// lanes split into SUB groups of 64 / SUB columns; group g owns rows [g * RY, (g + 1) * RY) of the wave's tile.  Inner y
// halos come from the neighbouring group (lane permutes in a real kernel: no request), outer ones from memory.
template <int WX, int WY, int RY, int SUB, bool HALO, bool ALIGN>
__global__ __launch_bounds__(64 * WX * WY) void march(Args a)
{
    constexpr int H = HALO ? 3 : 0, LW = 64 / SUB, OUTC = LW - 2 * H, TR = RY * SUB;  // TR rows per wave
    const int j = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
    const int tq = xcd * a.tiles_per_xcd + (j % a.tiles_per_xcd), chunk = j / a.tiles_per_xcd;
    if (tq >= a.gx * a.gy) return;
    const int xb = tq % a.gx, yb = tq / a.gx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / LW, col = lane % LW;
    const int n = a.n;
    int x = (xb * WX + (wave % WX)) * OUTC - H;
    if (ALIGN) x &= ~31;
    x = min(max(x + col, 0), n - 1);
    const int y0 = (yb * WY + (wave / WX)) * TR;
    const int z0 = chunk * a.zchunk, z1 = min(z0 + a.zchunk, a.nz);
    const bool emit = !HALO || (col >= H && col < LW - H);
    // rows a lane requests: its own RY, plus the outer halo if its group is the first / last of the wave.  One instruction
    // fetches one row per group, so the wave issues max over groups = RY + H row instructions per array (SUB > 1) or RY + 2H.
    constexpr int NI = SUB == 1 ? RY + 2 * H : RY + H;
    const int r0 = SUB == 1 ? -H : (grp == 0 ? -H : grp * RY);                  // first row of this lane's run
    const int nr = SUB == 1 ? NI : ((grp == 0 || grp == SUB - 1) ? RY + H : RY);  // rows it really needs
    for (int z = max(z0 - H, 0); z < z1; ++z) {
        const size_t pl = (size_t)z * n * n;
        float s = 0.0f;
#pragma unroll
        for (int r = 0; r < NI; ++r) {
            if (r < nr) {
                const size_t o = pl + (size_t)min(max(y0 + r0 + r, 0), n - 1) * n + x;
                s += a.in[0][o] + a.in[1][o] + a.in[2][o] + a.in[3][o] + a.in[4][o];
            }
        }
        __syncthreads();
        if (emit && z >= z0) {
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const int y = y0 + grp * RY + r;
                if (y < n) {
                    const size_t o = pl + (size_t)y * n + x;
                    a.out[0][o] = s; a.out[1][o] = s; a.out[2][o] = s; a.out[3][o] = s;
                }
            }
        }
    }
}

// ---- stage per wave
constexpr int spw_cu(int K, int RY, int s) { return RY + 2 * (K - s); }          // U rows stage s consumes
constexpr int r4(int r) { return (r + 3) / 4; }                                    // rows -> float4 slots
constexpr int spw_slots(int K, int RY, int s) { return r4(spw_cu(K, RY, s)) + 3 * r4(spw_cu(K, RY, s) - 1) + r4(spw_cu(K, RY, s) - 2); }
constexpr int spw_base(int K, int RY, int s) { return s <= 1 ? 0 : spw_base(K, RY, s - 1) + spw_slots(K, RY, s - 1); }
constexpr int spw_lds_bytes(int K, int RY) { return (spw_base(K, RY, K - 1) + spw_slots(K, RY, K - 1)) * 64 * 16; }

template <int K, int RY, int S, int VPR>
__device__ __forceinline__ void spw_stage(const Args &a, float4 *lds, int lane, int x, int y0, int t, int z0, bool emit)
{
    constexpr int CU = spw_cu(K, RY, S), NV = CU + 3 * (CU - 1) + (CU - 2);
    constexpr int NS = spw_slots(K, RY, S);
    const int n = a.n;
    float v[NS * 4];
    if constexpr (S == 0) {
        // the only wave that reads memory: U(t + 1) rows, P1..3(t), Input(t)
        const size_t pl = (size_t)min(t, a.nz - 1) * n * n;
        int q = 0;
#pragma unroll
        for (int r = 0; r < CU; ++r) v[q++] = a.in[0][pl + (size_t)min(max(y0 - K + r, 0), n - 1) * n + x];
#pragma unroll
        for (int c = 1; c <= 3; ++c)
#pragma unroll
            for (int r = 0; r < CU - 1; ++r) v[q++] = a.in[c][pl + (size_t)min(max(y0 - K + r, 0), n - 1) * n + x];
#pragma unroll
        for (int r = 1; r < CU - 1; ++r) v[q++] = a.in[4][pl + (size_t)min(max(y0 - K + r, 0), n - 1) * n + x];
#pragma unroll
        for (; q < NS * 4; ++q) v[q] = 0.0f;
    } else {
        const float4 *src = lds + spw_base(K, RY, S) * 64 + lane;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const float4 t = src[q * 64];
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    }
    __syncthreads();  // barrier A: every hand-over buffer has been read
    // stand-in arithmetic: VPR dependent FMAs per produced row (rows are independent chains, like the real update)
    constexpr int CO = S + 1 < K ? spw_cu(K, RY, S + 1) : RY;  // rows this stage produces
    float o[CO];
#pragma unroll
    for (int r = 0; r < CO; ++r) {
        float t = v[r] + v[(r + CU) % NV] + v[(r + 2 * CU) % NV] + v[(r + 3 * CU) % NV] + v[(r + 4 * CU - 5) % NV];
#pragma unroll
        for (int k = 0; k < VPR; ++k) t = fmaf(t, 0.999f, v[(r + k) % NV]);
        o[r] = t;
    }
    if constexpr (S + 1 < K) {
        constexpr int NO = spw_slots(K, RY, S + 1);
        float4 *dst = lds + spw_base(K, RY, S + 1) * 64 + lane;
#pragma unroll
        for (int q = 0; q < NO; ++q) dst[q * 64] = make_float4(o[(4 * q) % CO], o[(4 * q + 1) % CO], o[(4 * q + 2) % CO], o[(4 * q + 3) % CO]);
    } else {
        const int zo = t - 2 * (K - 1);  // the plane the last stage emits at this step (skewed pipeline: stage s runs plane t - 2s)
        if (emit && zo >= z0) {
            const size_t pl = (size_t)zo * n * n;
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const int y = y0 + r;
                if (y < n) {
                    const size_t oo = pl + (size_t)y * n + x;
                    a.out[0][oo] = o[r]; a.out[1][oo] = o[r]; a.out[2][oo] = o[r]; a.out[3][oo] = o[r];
                }
            }
        }
    }
    __syncthreads();  // barrier B: every hand-over buffer has been written
}

template <int K, int RY, int VPR>
__global__ __launch_bounds__(64 * K) void spw(Args a)
{
    extern __shared__ float4 lds[];
    const int j = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
    const int tq = xcd * a.tiles_per_xcd + (j % a.tiles_per_xcd), chunk = j / a.tiles_per_xcd;
    if (tq >= a.gx * a.gy) return;
    const int xb = tq % a.gx, yb = tq / a.gx;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = a.n;
    const int x = min(max(xb * (64 - 2 * K) - K + lane, 0), n - 1);
    const int y0 = yb * RY;
    const int z0 = chunk * a.zchunk, z1 = min(z0 + a.zchunk, a.nz);
    const bool emit = lane >= K && lane < 64 - K;
    // stage s works on plane t - 2s; warm-up: K planes below z0 for stage 0, drain: 2 (K - 1) steps
    for (int t = max(z0 - K, 0); t < z1 + 2 * (K - 1); ++t) {
        if (wave == 0) spw_stage<K, RY, 0, VPR>(a, lds, lane, x, y0, t, z0, emit);
        else if (wave == 1) spw_stage<K, RY, 1, VPR>(a, lds, lane, x, y0, t, z0, emit);
        else if (wave == 2) { if constexpr (K > 2) spw_stage<K, RY, 2, VPR>(a, lds, lane, x, y0, t, z0, emit); }
        else if (wave == 3) { if constexpr (K > 3) spw_stage<K, RY, 3, VPR>(a, lds, lane, x, y0, t, z0, emit); }
        else if (wave == 4) { if constexpr (K > 4) spw_stage<K, RY, 4, VPR>(a, lds, lane, x, y0, t, z0, emit); }
        else if (wave == 5) { if constexpr (K > 5) spw_stage<K, RY, 5, VPR>(a, lds, lane, x, y0, t, z0, emit); }
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

template <int WX, int WY, int RY, int SUB, bool HALO, bool ALIGN>
static void go(const char *name, int lds_kib, int chunks)
{
    constexpr int H = HALO ? 3 : 0, OUTC = 64 / SUB - 2 * H;
    Args a = g;
    a.gx = ((a.n + OUTC - 1) / OUTC + WX - 1) / WX;
    a.gy = (a.n + WY * RY * SUB - 1) / (WY * RY * SUB);
    a.tiles_per_xcd = (a.gx * a.gy + 7) / 8;
    a.zchunk = (a.nz + chunks - 1) / chunks;
    const unsigned blocks = 8u * a.tiles_per_xcd * chunks;
    const size_t dyn = (size_t)lds_kib * 1024;
    auto kern = march<WX, WY, RY, SUB, HALO, ALIGN>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    const double ms = time_ms([&] { kern<<<blocks, 64 * WX * WY, dyn>>>(a); });
    constexpr int NI = SUB == 1 ? RY + 2 * H : RY + H;
    printf("%-8s %dx%d waves, wave = %d x %2d lanes, %2d rows x %2d columns out per wave, %3d + %2d requests per wave and plane (%.4f per output), %2d chunks: %7.3f ms per launch-equivalent\n",
           name, WX, WY, SUB, 64 / SUB, RY * SUB, OUTC, 5 * NI, 4 * RY, (5.0 * NI + 4.0 * RY) / (RY * SUB * OUTC), chunks, ms);
    fflush(stdout);
}

template <int K, int RY, int VPR>
static void go_spw(int wg_per_cu, int chunks)
{
    constexpr int OUTC = 64 - 2 * K;
    Args a = g;
    a.gx = (a.n + OUTC - 1) / OUTC;
    a.gy = (a.n + RY - 1) / RY;
    a.tiles_per_xcd = (a.gx * a.gy + 7) / 8;
    a.zchunk = (a.nz + chunks - 1) / chunks;
    const unsigned blocks = 8u * a.tiles_per_xcd * chunks;
    size_t dyn = spw_lds_bytes(K, RY);
    const size_t pin = (size_t)160 * 1024 / wg_per_cu - 1024;  // occupancy: exactly wg_per_cu workgroups fit a CU
    if (dyn > pin + 1024) { printf("spw K=%d RY=%d: %zu B of hand-over LDS do not fit %d workgroups per CU\n", K, RY, dyn, wg_per_cu); return; }
    const size_t need = dyn;
    if (dyn < pin) dyn = pin;
    auto kern = spw<K, RY, VPR>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    const double ms = time_ms([&] { kern<<<blocks, 64 * K, dyn>>>(a); });
    constexpr int CU = RY + 2 * K, NL = CU + 3 * (CU - 1) + CU - 2;
    printf("spw      K = %d stages = waves, %2d rows x %2d columns out per workgroup, %3d + %2d requests per plane (%.4f per output-iteration x 3), %2d FMAs per row, %d WG/CU (%zu KiB hand-over), %2d chunks: %7.3f ms per pass = %7.3f ms per launch-equivalent\n",
           K, RY, OUTC, NL, 4 * RY, 3.0 * (NL + 4.0 * RY) / (RY * OUTC * K), VPR, wg_per_cu, need >> 10, chunks, ms, ms * 3.0 / K);
    fflush(stdout);
}

int main(int argc, char **argv)
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
    const int reps = argc > 1 ? atoi(argv[1]) : 2;
    for (int rep = 0; rep < reps; ++rep) {
        printf("# pass %d\n", rep);
        go<2, 2, 8, 1, true, false>("base", 80, 32);
        go<2, 2, 8, 1, false, false>("nohalo", 80, 32);
        go<2, 2, 8, 1, true, true>("align", 80, 32);
        go<2, 2, 8, 2, true, false>("half", 80, 32);
        go<2, 2, 8, 2, true, true>("half+al", 80, 32);
        go<2, 2, 8, 4, true, false>("quarter", 80, 32);
        go<2, 1, 8, 2, true, false>("half", 40, 32);
        go<1, 2, 8, 2, true, false>("half", 40, 32);
        go<2, 2, 8, 1, true, false>("base", 80, 32);
        go_spw<4, 12, 0>(2, 16);
        go_spw<4, 12, 24>(2, 16);
        go_spw<4, 12, 48>(2, 16);
        go_spw<4, 12, 24>(2, 32);
        go_spw<4, 16, 0>(2, 16);
        go_spw<4, 16, 24>(2, 16);
        go_spw<3, 16, 0>(2, 16);
        go_spw<3, 16, 24>(2, 16);
        go_spw<3, 16, 24>(3, 16);
        go_spw<3, 12, 24>(3, 16);
        go_spw<5, 12, 24>(1, 16);
        go_spw<6, 8, 24>(1, 16);
        go<2, 2, 8, 1, true, false>("base", 80, 32);
    }
    return 0;
}
