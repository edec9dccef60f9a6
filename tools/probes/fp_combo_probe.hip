// Probe (round 6): the COMBINATION the round-5 review asked for -- per-angle conflict-free lane -> pixel permutation AND two
// z-quads (8 slices) per thread -- on the forward projector's sampling loop (csrc/fp_tiled.inl, the `for r / for i` body).
// Round 5 closed each lever alone: the permutation because the VALU issue (12.6 instructions per read pair) co-binds with the
// LDS pipe, the second z-quad because LDS cycles per sample stay what they are.  Each argument is the other's refutation.
//
// The probe IS the kernel's sampling loop -- same index arithmetic (fma, floor, sub, sub, med3, cvt, lshl_add), same scalar row
// origin, same two ds_read_b128 per tap pair and packed FMAs, accumulators in registers, a march of `rows` volume rows with a
// per-row window origin -- without the staging (the LDS rows are filled once).  Angles are real: theta_i = theta0 + i * dtheta
// (OS-12 subsets of 900 angles: dtheta = 2.4 deg; dense sets: 0.12 deg).
//   ZQ = z-quads per thread (1 = shipped, 2 = 8 slices share one set of index arithmetic; two LDS planes)
//   A  = angles per thread, BT = threads (= detector pixels) per workgroup, WPC = workgroups per CU
//   perm: pixel of logical lane t for angle i = (m_i * t) mod BT, odd m_i chosen per angle by the bank model below
// Output: ns per wave-level read PAIR (2 x ds_read_b128 = 4 slices x 2 taps x 64 rays) per CU.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o _build/fp_combo_probe fp_combo_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ int lane_pixel(int lane)   // fp_tiled.inl: fp_lane_pixel
{
    const int q = (lane >> 2) & 7;
    const int odd = (q ^ (q >> 1) ^ (q >> 2)) & 1;
    return (lane & 32) | (odd << 4) | ((q >> 1) << 2) | (lane & 3);
}

struct Args {
    float slope[16], inv[16];
    int mult[16];
    int rows, n, wslots;     // march length, volume width, LDS slots per plane
    const int *win_lo;       // [rows] window origin per row (device)
    float *out;
};

template <int ZQ, int A, int BT>
__global__ __launch_bounds__(BT) void probe(Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *tile = reinterpret_cast<float4 *>(smem);
    int *wl = reinterpret_cast<int *>(tile + ZQ * a.wslots);
    for (int i = threadIdx.x; i < ZQ * a.wslots; i += BT) tile[i] = make_float4(1.0f + i, 2.0f, 3.0f, 0.5f * i);
    for (int i = threadIdx.x; i < a.rows; i += BT) wl[i] = a.win_lo[i];
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63;
    const int t = (tid - lane) + lane_pixel(lane);
    const float half_n = 0.5f * (float)a.n - 0.5f, half_u = 0.5f * (float)BT - 0.5f, nf = (float)a.n;
    float offs[A], slope[A], acc[A][4 * ZQ];
#pragma unroll
    for (int i = 0; i < A; ++i) {
        const int pix = (a.mult[i] * t) & (BT - 1);
        offs[i] = fmaf((float)pix - half_u, a.inv[i], half_n);
        slope[i] = a.slope[i];
#pragma unroll
        for (int z = 0; z < 4 * ZQ; ++z) acc[i][z] = 0.0f;
    }
    const int plane = a.wslots * 16;
    for (int k = 0; k < a.rows; ++k) {
        const float kw = (float)k - half_n;
        int rb = __builtin_amdgcn_readfirstlane(-wl[k] * 16);
        asm("" : "+s"(rb));
        const char *trow = reinterpret_cast<const char *>(tile);
#pragma unroll
        for (int i = 0; i < A; ++i) {
            const float f = fmaf(kw, slope[i], offs[i]);
            const float fl = floorf(f);
            const float w = f - fl, omw = 1.0f - w;
            const int idx = (int)__builtin_amdgcn_fmed3f(fl, -2.0f, nf);
            const char *p = trow + ((idx << 4) + rb);
#pragma unroll
            for (int zq = 0; zq < ZQ; ++zq) {
                const float4 *tap = reinterpret_cast<const float4 *>(p + zq * plane);
                const float4 s0 = tap[0], s1 = tap[1];
                float *c = &acc[i][4 * zq];
                c[0] = fmaf(omw, s0.x, c[0]); c[0] = fmaf(w, s1.x, c[0]);
                c[1] = fmaf(omw, s0.y, c[1]); c[1] = fmaf(w, s1.y, c[1]);
                c[2] = fmaf(omw, s0.z, c[2]); c[2] = fmaf(w, s1.z, c[2]);
                c[3] = fmaf(omw, s0.w, c[3]); c[3] = fmaf(w, s1.w, c[3]);
            }
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
        for (int z = 0; z < 4 * ZQ; ++z) s += acc[i][z];
    a.out[(size_t)blockIdx.x * BT + tid] = s;
}


// ---- the same loop, software-pipelined by hand: the reads of unit u+P are issued before the FMAs of unit u (a unit = one
// (row, angle): 2 ZQ reads); inline-assembly reads and explicit lgkmcnt waits, because the compiler serialises each read pair
// behind its own wait once the accumulators take 64 registers.  KC rows per chunk share one read of the window origins.
typedef float v4f __attribute__((ext_vector_type(4)));
template <int N> __device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

template <int ZQ, int A, int BT, int P, int WPE>
__global__ __launch_bounds__(BT) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void probe_man(Args a)
{
    constexpr int KC = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *tile = reinterpret_cast<float4 *>(smem);
    int *wl = reinterpret_cast<int *>(tile + ZQ * a.wslots);
    for (int i = threadIdx.x; i < ZQ * a.wslots; i += BT) tile[i] = make_float4(1.0f + i, 2.0f, 3.0f, 0.5f * i);
    for (int i = threadIdx.x; i < a.rows; i += BT) wl[i] = a.win_lo[i];
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63;
    const int t = (tid - lane) + lane_pixel(lane);
    const float half_n = 0.5f * (float)a.n - 0.5f, half_u = 0.5f * (float)BT - 0.5f, nf = (float)a.n;
    float offs[A], slope[A], acc[A][4 * ZQ];
#pragma unroll
    for (int i = 0; i < A; ++i) {
        const int pix = (a.mult[i] * t) & (BT - 1);
        offs[i] = fmaf((float)pix - half_u, a.inv[i], half_n);
        slope[i] = a.slope[i];
#pragma unroll
        for (int z = 0; z < 4 * ZQ; ++z) acc[i][z] = 0.0f;
    }
    const int plane = a.wslots * 16;
    const unsigned tile_base = (unsigned)(size_t)tile;   // LDS byte address
    constexpr int U = KC * A;          // units per chunk
    v4f b0[P + 1][ZQ], b1[P + 1][ZQ];
    float wv[P + 1], omwv[P + 1];
    for (int k0 = 0; k0 < a.rows; k0 += KC) {
        const int4 lo4 = *reinterpret_cast<const int4 *>(wl + k0);
        int rb[KC] = {__builtin_amdgcn_readfirstlane(-lo4.x * 16), __builtin_amdgcn_readfirstlane(-lo4.y * 16),
                      __builtin_amdgcn_readfirstlane(-lo4.z * 16), __builtin_amdgcn_readfirstlane(-lo4.w * 16)};
        float kw[KC];
#pragma unroll
        for (int r = 0; r < KC; ++r) { kw[r] = (float)(k0 + r) - half_n; asm("" : "+s"(rb[r])); }
        auto issue = [&](int u) {
            const int r = u / A, i = u % A, s = u % (P + 1);
            const float f = fmaf(kw[r], slope[i], offs[i]);
            const float fl = floorf(f);
            wv[s] = f - fl;
            omwv[s] = 1.0f - wv[s];
            const int idx = (int)__builtin_amdgcn_fmed3f(fl, -2.0f, nf);
            const unsigned addr = tile_base + (unsigned)((idx << 4) + rb[r]);
#pragma unroll
            for (int zq = 0; zq < ZQ; ++zq) {
                const unsigned ad = addr + zq * plane;
                asm volatile("ds_read_b128 %0, %1" : "=v"(b0[s][zq]) : "v"(ad));
                asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(b1[s][zq]) : "v"(ad));
            }
        };
        auto consume = [&](int u) {
            const int i = u % A, s = u % (P + 1);
#pragma unroll
            for (int zq = 0; zq < ZQ; ++zq) {
                asm volatile("" : "+v"(b0[s][zq]), "+v"(b1[s][zq]));   // values are defined from here (after the wait)
                const v4f s0 = b0[s][zq], s1 = b1[s][zq];
                const float w = wv[s], omw = omwv[s];
                float *c = &acc[i][4 * zq];
                c[0] = fmaf(omw, s0.x, c[0]); c[0] = fmaf(w, s1.x, c[0]);
                c[1] = fmaf(omw, s0.y, c[1]); c[1] = fmaf(w, s1.y, c[1]);
                c[2] = fmaf(omw, s0.z, c[2]); c[2] = fmaf(w, s1.z, c[2]);
                c[3] = fmaf(omw, s0.w, c[3]); c[3] = fmaf(w, s1.w, c[3]);
            }
        };
#pragma unroll
        for (int u = 0; u < P; ++u) issue(u);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u + P < U) { issue(u + P); lds_wait<2 * ZQ * P>(); }
            else {
                // tail: U-1-u units issued after this one are still in flight
                constexpr int dummy = 0; (void)dummy;
                switch (U - 1 - u) {
                    case 0: lds_wait<0>(); break;
                    case 1: lds_wait<2 * ZQ * 1>(); break;
                    case 2: lds_wait<2 * ZQ * 2>(); break;
                    default: lds_wait<2 * ZQ * 3>(); break;
                }
            }
            consume(u);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
        for (int z = 0; z < 4 * ZQ; ++z) s += acc[i][z];
    a.out[(size_t)blockIdx.x * BT + tid] = s;
}

// ---- bank model: LDS cycles of one 16-lane service group (16 slots of 16 B = one bank row); lanes with the same slot share
// a broadcast, lanes with equal slot mod 16 but different slots serialise.  Mean over service groups and fractional phases.
static double model_cycles(double s, int m, int bt)
{
    double total = 0.0;
    int cnt = 0;
    for (int ph = 0; ph < 29; ++ph) {
        const double x0 = 3.0 + ph * (1.0 / 29.0) * 7.3;
        for (int t0 = 0; t0 < bt; t0 += 16) {
            int slot[16];
            for (int j = 0; j < 16; ++j) slot[j] = (int)std::floor(x0 + s * (double)((m * (t0 + j)) & (bt - 1)));
            int worst = 1;
            for (int b = 0; b < 16; ++b) {
                int distinct[16], nd = 0;
                for (int j = 0; j < 16; ++j)
                    if ((slot[j] & 15) == b) {
                        bool seen = false;
                        for (int q = 0; q < nd; ++q) seen |= distinct[q] == slot[j];
                        if (!seen) distinct[nd++] = slot[j];
                    }
                worst = std::max(worst, nd);
            }
            total += worst;
            ++cnt;
        }
    }
    return total / cnt;
}

static int best_mult(double s, int bt, double *cyc_out)
{
    int bm = 1;
    double best = 1e30;
    for (int m = 1; m < 64; m += 2) {
        const double c = model_cycles(s, m, bt);
        if (c < best - 1e-9) { best = c; bm = m; }
    }
    *cyc_out = best;
    return bm;
}

template <int ZQ, int A, int BT>
static double run(const Args &base, int wpc, hipEvent_t e0, hipEvent_t e1)
{
    Args a = base;
    const size_t shm = (size_t)ZQ * a.wslots * 16 + (size_t)a.rows * 4;
    hipFuncSetAttribute((const void *)probe<ZQ, A, BT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    const int blocks = 256 * wpc;
    probe<ZQ, A, BT><<<blocks, BT, shm>>>(a);
    hipEventRecord(e0);
    for (int rep = 0; rep < 3; ++rep) probe<ZQ, A, BT><<<blocks, BT, shm>>>(a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    if (hipGetLastError() != hipSuccess) return -1.0;
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3.0f;
    // read pairs per CU: workgroups per CU x waves x rows x angles x z-quads
    const double pairs = (double)wpc * (BT / 64) * a.rows * A * ZQ;
    return ms * 1e6 / pairs;
}

template <int ZQ, int A, int BT, int P, int WPE>
static double run_man(const Args &base, int wpc, hipEvent_t e0, hipEvent_t e1)
{
    Args a = base;
    const size_t shm = (size_t)ZQ * a.wslots * 16 + (size_t)a.rows * 4;
    if (shm * wpc > 160 * 1024) return -1.0;
    hipFuncSetAttribute((const void *)probe_man<ZQ, A, BT, P, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    const int blocks = 256 * wpc;
    probe_man<ZQ, A, BT, P, WPE><<<blocks, BT, shm>>>(a);
    hipEventRecord(e0);
    for (int rep = 0; rep < 3; ++rep) probe_man<ZQ, A, BT, P, WPE><<<blocks, BT, shm>>>(a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    if (hipGetLastError() != hipSuccess) return -1.0;
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3.0f;
    const double pairs = (double)wpc * (BT / 64) * a.rows * A * ZQ;
    return ms * 1e6 / pairs;
}

int main()
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 1024, rows = 1024;
    float *out;
    int *wl_dev;
    hipMalloc(&out, sizeof(float) * 256 * 4 * 1024);
    hipMalloc(&wl_dev, sizeof(int) * rows);
    struct Case { const char *name; double theta0, dtheta; };
    const Case cases[] = {
        {"os12 [-45,-28]", -45.0, 2.4}, {"os12 [-26,-9]", -26.0, 2.4}, {"os12 [-8,9]", -8.0, 2.4}, {"os12 [10,27]", 10.0, 2.4},
        {"os12 [28,45]", 28.0, 2.4},
        {"dense 0", 0.0, 0.12}, {"dense 10", 10.0, 0.12}, {"dense 20", 20.0, 0.12}, {"dense 30", 30.0, 0.12}, {"dense 40", 40.0, 0.12},
        {"dense 43", 43.0, 0.12},
    };
    const char *cols[] = {"W:ship", "W:ship+p", "W:man1", "W:man1+p", "W:zq2", "W:zq2+p", "D:ship16", "D:man16", "D:ship16+p",
                          "D:zq2a8", "D:zq2a8+p", "E:zq2a16", "E:zq2a16+p", "F:zq2a8", "F:zq2a8+p", "Wc:zq2", "Wc:zq2+p", "Dc:zq2a8", "Dc:zq2a8+p"};
    constexpr int NC = 19;
    printf("# ns per wave-level read PAIR (2 x ds_read_b128: 4 slices x 2 taps x 64 rays) per CU; 1024 rows of a 1024-wide volume\n");
    printf("# W = 1024 threads, 1 workgroup per CU, 8 angles (whole-row form); D = 256 threads, 3 per CU; E, F = 256 threads, 2 per CU\n");
    printf("# ship = compiler-scheduled loop as shipped (ZQ 1); man = hand-pipelined reads; zq2 = 8 slices per thread; Wc / Dc = zq2 with the compiler's own schedule; +p = per-angle multiplier\n");
    printf("# model1 / modelP = bank-model LDS cycles per 16-lane service group without / with the permutation (1024-pixel tiles)\n");
    printf("%-15s %5s %5s |", "case", "mod1", "modP");
    for (int j = 0; j < NC; ++j) printf(" %10s", cols[j]);
    printf("\n");
    double sums[NC] = {0};
    int ncase = 0;
    for (const Case &c : cases) {
        Args a;
        a.rows = rows; a.n = n; a.out = out; a.win_lo = wl_dev;
        int m1024[16], m256[16];
        double mod1 = 0, modp = 0;
        for (int i = 0; i < 16; ++i) {
            const double th = (c.theta0 + c.dtheta * i) * M_PI / 180.0;
            a.slope[i] = (float)std::tan(th);
            a.inv[i] = (float)(1.0 / std::cos(th));
            double cy;
            m1024[i] = best_mult(1.0 / std::cos(th), 1024, &cy);
            if (i < 8) { modp += cy / 8; mod1 += model_cycles(1.0 / std::cos(th), 1, 1024) / 8; }
            m256[i] = best_mult(1.0 / std::cos(th), 256, &cy);
        }
        auto windows = [&](int A, int bt, int &wslots) {   // per-row window origin over the angle group and the tile's two ends
            std::vector<int> wl(rows);
            const float half_n = 0.5f * n - 0.5f, half_u = 0.5f * bt - 0.5f;
            int wmax = 0;
            for (int k = 0; k < rows; ++k) {
                float fmin = 3e38f, fmax = -3e38f;
                for (int i = 0; i < A; ++i)
                    for (int e = 0; e < 2; ++e) {
                        const float o = std::fmaf((e ? bt - 1 : 0) - half_u, a.inv[i], half_n);
                        const float f = std::fmaf((float)k - half_n, a.slope[i], o);
                        fmin = std::min(fmin, f); fmax = std::max(fmax, f);
                    }
                const int lo = (int)std::min(std::max(std::floor(fmin), -2.0f), (float)n);
                const int hi = (int)std::min(std::max(std::floor(fmax) + 1.0f, -1.0f), (float)(n + 1));
                wl[k] = lo;
                wmax = std::max(wmax, hi - lo + 2);
            }
            hipMemcpy(wl_dev, wl.data(), sizeof(int) * rows, hipMemcpyHostToDevice);
            wslots = wmax + 2;
        };
        auto setm = [&](const int *m) { for (int i = 0; i < 16; ++i) a.mult[i] = m ? m[i] : 1; };
        // every configuration three times round-robin after a warm-up round (the first launches after the host-side model
        // run on an idle, down-clocked chip: the first version of this probe charged that to whatever ran first); minimum kept
        double r[NC];
        for (int j = 0; j < NC; ++j) r[j] = 1e30;
        for (int round = 0; round < 4; ++round) {
            double q[NC];
            windows(8, 1024, a.wslots);
            setm(nullptr);
            q[0] = run<1, 8, 1024>(a, 1, e0, e1);
            q[2] = run_man<1, 8, 1024, 3, 4>(a, 1, e0, e1);
            q[4] = run_man<2, 8, 1024, 1, 4>(a, 1, e0, e1);
            q[15] = run<2, 8, 1024>(a, 1, e0, e1);
            setm(m1024);
            q[16] = run<2, 8, 1024>(a, 1, e0, e1);
            q[1] = run<1, 8, 1024>(a, 1, e0, e1);
            q[3] = run_man<1, 8, 1024, 3, 4>(a, 1, e0, e1);
            q[5] = run_man<2, 8, 1024, 1, 4>(a, 1, e0, e1);
            windows(16, 256, a.wslots);
            setm(nullptr);
            q[6] = run<1, 16, 256>(a, 3, e0, e1);
            q[7] = run_man<1, 16, 256, 3, 3>(a, 3, e0, e1);
            q[11] = run_man<2, 16, 256, 1, 2>(a, 2, e0, e1);
            setm(m256);
            q[8] = run<1, 16, 256>(a, 3, e0, e1);
            q[12] = run_man<2, 16, 256, 1, 2>(a, 2, e0, e1);
            windows(8, 256, a.wslots);
            setm(nullptr);
            q[9] = run_man<2, 8, 256, 1, 3>(a, 3, e0, e1);
            q[13] = run_man<2, 8, 256, 2, 2>(a, 2, e0, e1);
            q[17] = run<2, 8, 256>(a, 3, e0, e1);
            setm(m256);
            q[18] = run<2, 8, 256>(a, 3, e0, e1);
            q[10] = run_man<2, 8, 256, 1, 3>(a, 3, e0, e1);
            q[14] = run_man<2, 8, 256, 2, 2>(a, 2, e0, e1);
            if (round > 0)
                for (int j = 0; j < NC; ++j) r[j] = std::min(r[j], q[j]);
        }
        printf("%-15s %5.2f %5.2f |", c.name, mod1, modp);
        for (int j = 0; j < NC; ++j) printf(" %10.2f", r[j]);
        printf("\n");
        for (int j = 0; j < NC; ++j) sums[j] += r[j];
        ++ncase;
    }
    printf("%-15s %5s %5s |", "mean", "", "");
    for (int j = 0; j < NC; ++j) printf(" %10.2f", sums[j] / ncase);
    printf("\n# gains over the shipped structures: Wc:zq2+p %.3f x (W:ship / Wc:zq2+p), D:ship16+p %.3f x, Dc:zq2a8+p %.3f x (vs D:ship16)\n",
           sums[0] / sums[16], sums[6] / sums[8], sums[6] / sums[18]);
    hipFree(out); hipFree(wl_dev);
    return 0;
}
