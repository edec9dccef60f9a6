// PD_TV, K Chambolle-Pock iterations per pass through HBM (3D only): the generalisation of pd_zmarch_x2.inl to a chain of
// K stages.  Included inside the anonymous namespace of tv_kernels.hip (uses PdArgs, DualIO, pd_dual_t, pd_primal_t).
//
// One iteration moves 36 B/voxel and the two-iteration kernel is HBM-bound once the IEEE divide / sqrt are relaxed
// (profiles/r2_b_pdtv_tile_vs_x2_pmc.txt), so the only lever left is bytes per iteration.  Stage s (iteration n+s ->
// n+s+1) works on plane t-s at step t of the z-march; what it needs from stage s-1 (U^{n+s} of planes t-s-1, t-s, t-s+1
// and P^{n+s} of plane t-s) is still in registers, so HBM sees one read of Input, U, P1..3 and one write of U, P1..3 per
// K iterations.  The price is the halo every wave re-computes: stage s evaluates its duals on rows -(K-s) .. RY+(K-s)-2
// and lanes s .. 62-s, the last stage emits rows 0 .. RY-1 and lanes K .. 63-K.  With K = 3, RY = 4: 45 row loads and 21
// dual-row evaluations per 12 output-row-iterations (K = 2: 35 and 12 per 8).
// Row slots r in [-K, RY+K) are array index r+K; every array is declared at the full height NR and the unrolled code
// only ever touches the slots a stage needs (the rest is removed by the compiler).
// Arithmetic per voxel and iteration is exactly that of the single-iteration kernels, whatever K is.
// LAG: the state a stage hands over ACROSS steps (P^{n+s}(t-s) for stages s >= 1 and the Input planes t-1 .. t-K+1) lives
// in LDS instead of registers: every thread owns private slots [slot][thread] (conflict-free, no barrier needed), read
// back exactly where it is consumed.  That is what lets K = 3 run with RY = 4 rows at two waves per SIMD (46 values per
// lane would otherwise push the kernel past 256 registers): 45 row loads / 21 dual rows per 12 output-row-iterations
// instead of 40 / 18 per 9.
// compile-time stage loop: `s` must be a constant inside the stage body.  With a run-time (unrolled) loop the code of
// "s + 1 < K" for the last stage survives as a dead, not yet unrolled loop with variable indices into the register
// arrays until after the last scalar-replacement pass, which then leaves the arrays in scratch memory.
template <typename F, int... I>
__device__ __forceinline__ void xk_static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void xk_static_for(F &&f)
{
    xk_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr int xk_dual_rows(int K, int RY, int s) { return RY + 2 * (K - s) - 1; }
constexpr int xk_p_base(int K, int RY, int s) { return s <= 1 ? 0 : xk_p_base(K, RY, s - 1) + 3 * xk_dual_rows(K, RY, s - 1); }
constexpr int xk_in_rows(int K, int RY) { return RY + 2 * (K - 2); }
constexpr int xk_lag_slots(int K, int RY) { return xk_p_base(K, RY, K) + K * xk_in_rows(K, RY); }

template <typename T, bool NONNEG, bool ANISO, int FAST, int K, int RY, int WX, int WY, bool LAG = false, int LREG = 0>
__global__ __launch_bounds__(64 * WX * WY) void pd_zmarch_xk_kernel(PdArgs a, int gx, int gy, int tiles_per_xcd)
{
    constexpr int NR = RY + 2 * K;
    constexpr int NT = 64 * WX * WY;
    constexpr int IN_BASE = xk_p_base(K, RY, K), IN_ROWS = xk_in_rows(K, RY);
    // the first LREG hand-over slots stay in registers (statically indexed): trims the LDS footprint to what lets one
    // more workgroup share the CU
    __shared__ float lag[LAG ? xk_lag_slots(K, RY) - LREG : 1][LAG ? NT : 1];
    float lagreg[LREG > 0 ? LREG : 1];
#pragma unroll
    for (int q = 0; q < (LREG > 0 ? LREG : 1); ++q) lagreg[q] = 0.0f;
    const int tid = (int)threadIdx.x;
    // every XCD owns one contiguous eighth of the row-major (yb, xb) tile list: a band of rows whose halos meet in that
    // XCD's L2, and the same number of workgroups per XCD whatever gy is (tiles_per_xcd = ceil(gx * gy / 8))
    const int j = (int)blockIdx.x >> 3;
    const int xcd = (int)blockIdx.x & 7;
    const int tq = xcd * tiles_per_xcd + (j % tiles_per_xcd);
    const int chunk = j / tiles_per_xcd;
    if (tq >= gx * gy) return;
    const int xb = tq % gx;
    const int yb = tq / gx;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int xs = xb * WX + (wave % WX);
    const int x = xs * (64 - 2 * K) - K + lane;
    const int y0 = (yb * WY + (wave / WX)) * RY;
    const int dx = a.dx, dy = a.dy, dz = a.planes;
    const int zc0 = a.out_begin + chunk * a.zchunk;
    const int zc1 = min(zc0 + a.zchunk, a.out_end);
    if (zc0 >= zc1) return;  // uniform for the workgroup

    const size_t sz = (size_t)dx * dy;
    const bool x_last = (x == dx - 1);
    const bool x_has_prev = (x > 0);
    const bool emit_lane = (lane >= K) && (lane <= 63 - K) && (x < dx);
    int xc = min(max(x, 0), dx - 1);
    if (a.probe & 2) xc = min(max(xc, xb * WX * (64 - 2 * K)), min((xb + 1) * WX * (64 - 2 * K), dx) - 1);

    unsigned off[NR];  // byte offsets of row slots -K..RY+K-1 (clamped into the volume)
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        int yy = min(max(y0 + i - K, 0), dy - 1);
        if (a.probe & 1) yy = min(max(yy, yb * WY * RY), min((yb + 1) * WY * RY, dy) - 1);
        off[i] = (unsigned)(yy * dx + xc) * 4u;
    }
    const PlaneIO io{(int)(sz * 4)};  // plane-relative buffer addressing, see tv_kernels.hip
    auto ldf = [&](const float *base, unsigned boff) { return io.ldf(base, boff); };
    auto ldd = [&](const T *base, unsigned boff) { return io.ldd(base, boff); };
    const T *P_in[3] = {(const T *)a.p_in[0], (const T *)a.p_in[1], (const T *)a.p_in[2]};
    T *P_out[3] = {(T *)a.p_out[0], (T *)a.p_out[1], (T *)a.p_out[2]};

    // ---- persistent state.  Index [s] = stage.  Slots outside a stage's row range are never touched.
    float U0c[NR];               // U^n(t)
    float Ur[K][3][NR];          // stage s >= 1: U^{n+s} of planes t-s-1, t-s, t-s+1   ([s][2] is written by stage s-1)
    float Pp[K][3][NR];          // stage s >= 1: P^{n+s}(t-s) = duals stage s-1 produced one step earlier
    float In[K][NR];             // Input(t-s)
    float c3[K][NR];             // P3^{n+s+1}(t-s-1): z-1 dual of stage s
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            Ur[s][0][i] = 0.0f; Ur[s][1][i] = 0.0f; Ur[s][2][i] = 0.0f;
            Pp[s][0][i] = 0.0f; Pp[s][1][i] = 0.0f; Pp[s][2][i] = 0.0f;
            In[s][i] = 0.0f; c3[s][i] = 0.0f;
        }

    if (LAG) {
#pragma unroll
        for (int q = 0; q < xk_lag_slots(K, RY) - LREG; ++q) lag[q][tid] = 0.0f;
    }

    // stage s starts K-s planes below the first output plane (warm-up planes rebuild the carries and the rings)
    const int zA = max(zc0 - K, 0);
    const int tEnd = min(zc1, dz) + K - 2;  // the last stage must reach plane zc1-1
    {
        const float *up = a.u_in + sz * zA;
#pragma unroll
        for (int i = 0; i < NR; ++i) U0c[i] = ldf(up, off[i]);
    }

    for (int t = zA; t <= tEnd; ++t) {
        __syncthreads();  // lockstep: lines shared with the neighbouring waves merge in L1 (see pd_zmarch2)
        float U0n[NR];
        float Pw[3][NR];
        float Pnext[K][3][NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) { U0n[i] = 0.0f; Pw[0][i] = 0.0f; Pw[1][i] = 0.0f; Pw[2][i] = 0.0f; }
        const bool act0 = (t < dz);
        if (act0) {
            const bool z_last = (t == dz - 1) && a.last_is_edge;
            const int zn = z_last ? max(t - 1, 0) : min(t + 1, dz - 1);
            const float *up = a.u_in + sz * zn;
#pragma unroll
            for (int i = 0; i < NR; ++i) U0n[i] = ldf(up, off[i]);
            if (z_last && t == 0) {
#pragma unroll
                for (int i = 0; i < NR; ++i) U0n[i] = 0.0f;
            }
            if (!a.p_in_zero) {  // uniform: the first launch of a prox starts from zero duals (nothing to read)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const T *pp = P_in[c] + sz * t;
#pragma unroll
                    for (int i = 0; i < NR - 1; ++i) Pw[c][i] = ldd(pp, off[i]);
                }
            }
            const float *ip = a.in + sz * t;
#pragma unroll
            for (int i = 1; i < NR - 1; ++i) In[0][i] = ldf(ip, off[i]);
            if (LAG) {  // Input(t) for the later stages: ring slot t mod K
                const int q = t % K;
#pragma unroll
                for (int i = 2; i < 2 + IN_ROWS; ++i) lag[IN_BASE - LREG + q * IN_ROWS + (i - 2)][tid] = In[0][i];
            }
        }
        xk_static_for<K>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            const int p = t - s;                                 // plane of this stage
            const bool act = (p >= max(zc0 - (K - s), 0)) && (p < dz) && (s > 0 || act0);
            float Vn[NR];                                        // U^{n+s+1}(p), rows -(K-s-1) .. RY+(K-s-1)-1
#pragma unroll
            for (int i = 0; i < NR; ++i) Vn[i] = 0.0f;
            if (act) {
                const bool p_last = (p == dz - 1) && a.last_is_edge;
                if constexpr (s > 0 && !LAG) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int i = 0; i < NR; ++i) Pw[c][i] = (sizeof(T) == 2) ? DualIO<T>::rt(Pp[s][c][i]) : Pp[s][c][i];
                }
                float InS[NR];  // Input(p) of this stage
#pragma unroll
                for (int i = 0; i < NR; ++i) InS[i] = In[s][i];
                if constexpr (LAG && s > 0) {
                    const int q = (t + K - s) % K;
#pragma unroll
                    for (int r = -(K - s - 1); r <= RY + (K - s - 1) - 1; ++r) InS[r + K] = lag[IN_BASE - LREG + q * IN_ROWS + (r + K - 2)][tid];
                }
                // ---------------- duals, rows -(K-s) .. RY+(K-s)-2
#pragma unroll
                for (int r = -(K - s); r <= RY + (K - s) - 2; ++r) {
                    const int i = r + K;
                    const int y = y0 + r;
                    const float u = (s == 0) ? U0c[i] : Ur[s][1][i];
                    const float ux = __shfl_down(u, 1, 64);
                    const float uxm = __shfl_up(u, 1, 64);
                    float g[3];
                    g[0] = (x_last ? (x_has_prev ? uxm : 0.0f) : ux) - u;
                    const float u_up = (s == 0) ? U0c[i > 0 ? i - 1 : 0] : Ur[s][1][i > 0 ? i - 1 : 0];
                    const float u_dn = (s == 0) ? U0c[i + 1] : Ur[s][1][i + 1];
                    const float uy_mirror = (y > 0) ? u_up : 0.0f;  // the first row slot is never the volume's last row
                    g[1] = ((y == dy - 1) ? uy_mirror : u_dn) - u;
                    float uz;
                    if (s == 0) uz = U0n[i];  // stage 0: the loaded plane is already the mirrored one at the far edge
                    else uz = p_last ? ((p > 0) ? Ur[s][0][i] : 0.0f) : Ur[s][2][i];
                    g[2] = uz - u;
                    float pv[3] = {Pw[0][i], Pw[1][i], Pw[2][i]};
                    pd_dual_t<ANISO, FAST>(pv, g, a.sigma);
                    Pw[0][i] = pv[0]; Pw[1][i] = pv[1]; Pw[2][i] = pv[2];
                }
                // ---------------- primal, rows -(K-s-1) .. RY+(K-s-1)-1
                const bool emit_plane = (s == K - 1) && (p >= zc0);
#pragma unroll
                for (int r = -(K - s - 1); r <= RY + (K - s - 1) - 1; ++r) {
                    const int i = r + K;
                    const int y = y0 + r;
                    const float p1l = __shfl_up(Pw[0][i], 1, 64);
                    const float px = x_has_prev ? p1l : 0.0f;
                    const float py = (y > 0) ? Pw[1][i - 1] : 0.0f;
                    const float pz = (p > 0) ? c3[s][i] : 0.0f;
                    float div = (-(Pw[0][i] - px)) + (-(Pw[1][i] - py));
                    div = div + (-(Pw[2][i] - pz));
                    const float u = (s == 0) ? U0c[i] : Ur[s][1][i];
                    const float uo = pd_primal_t<FAST>(u, InS[i], div, a.tau, a.lt, a.inv1lt, a.theta, NONNEG);
                    Vn[i] = uo;
                    if constexpr (s == K - 1) {
                        if (emit_plane && emit_lane && y < dy) {
                            io.stf(a.u_out + sz * p, off[i], uo);
                            if (!a.p_out_skip) {  // uniform: nobody reads the duals of the last launch of a prox
#pragma unroll
                                for (int c = 0; c < 3; ++c) io.std_(P_out[c] + sz * p, off[i], Pw[c][i]);
                            }
                        }
                    }
                }
            }
            // ---------------- hand over to stage s+1 (always: drain steps need the rotation; values produced by an
            //                  inactive stage are never consumed, see the activity windows above)
            if (act) {
#pragma unroll
                for (int i = 0; i < NR; ++i) c3[s][i] = Pw[2][i];
            }
            if constexpr (s + 1 < K) {
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    Ur[s + 1][0][i] = Ur[s + 1][1][i];
                    Ur[s + 1][1][i] = Ur[s + 1][2][i];
                    Ur[s + 1][2][i] = Vn[i];
                }
                // P^{n+s+1}(p) is stage s+1's input at the NEXT step (stage s+1 of this step reads the previous hand-over)
                if constexpr (!LAG) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int i = 0; i < NR; ++i) Pnext[s + 1][c][i] = Pw[c][i];
                } else {
                    // swap through LDS: fetch what stage s+1 needs NOW (last step's hand-over), leave this step's behind
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int r = -(K - s - 1); r <= RY + (K - s - 1) - 2; ++r) {
                            const int i = r + K;
                            const int slot = xk_p_base(K, RY, s + 1) + c * xk_dual_rows(K, RY, s + 1) + (i - (s + 1));
                            const float nxt = Pw[c][i];
                            float old;
                            if (slot < LREG) {
                                old = lagreg[slot];
                                lagreg[slot] = nxt;
                            } else {
                                old = lag[slot - LREG][tid];
                                lag[slot - LREG][tid] = nxt;
                            }
                            Pw[c][i] = (sizeof(T) == 2) ? DualIO<T>::rt(old) : old;
                        }
                }
            }
        });
        if (!LAG) {
#pragma unroll
            for (int s = 1; s < K; ++s)
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int i = 0; i < NR; ++i) Pp[s][c][i] = Pnext[s][c][i];
            // ---------------- rotate the Input ring
#pragma unroll
            for (int s = K - 1; s > 0; --s)
#pragma unroll
                for (int i = 0; i < NR; ++i) In[s][i] = In[s - 1][i];
        }
        if (act0) {
#pragma unroll
            for (int i = 0; i < NR; ++i) U0c[i] = U0n[i];
        }
    }
}

template <typename T, bool NONNEG, bool ANISO, int FAST, int K, int RY, int WX, int WY, bool LAG = false, int LREG = 0>
static int pd_zmarch_xk_launch(PdArgs a, hipStream_t st, long want_per_simd = 32, int min_chunk = 24)
{
    const int nout = a.out_end - a.out_begin;
    const int gx = ceil_div(ceil_div(a.dx, 64 - 2 * K), WX), gy = ceil_div(a.dy, WY * RY);
    const int tiles_per_xcd = ceil_div(gx * gy, 8);
    const long waves_xy = (long)gx * gy * WX * WY;
    int chunks = (int)((256L * 4 * want_per_simd + waves_xy - 1) / waves_xy);
    const int max_chunks = ceil_div(nout, min_chunk * K);  // K warm-up planes per chunk: keep chunks long
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    a.zchunk = ceil_div(nout, chunks);
    chunks = ceil_div(nout, a.zchunk);
    a.inv1lt = 1.0f / (1.0f + a.lt);
    const long blocks = 8L * tiles_per_xcd * chunks;
    if (blocks > 0x7fffffffL) return tomo_fail(TOMO_E_INVALID, "volume too large for one PD_TV launch");
    pd_zmarch_xk_kernel<T, NONNEG, ANISO, FAST, K, RY, WX, WY, LAG, LREG><<<(unsigned)blocks, 64 * WX * WY, 0, st>>>(a, gx, gy, tiles_per_xcd);
    return TOMO_OK;
}
