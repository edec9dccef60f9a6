// PD_TV, K Chambolle-Pock iterations per pass through HBM (3D only): the generalisation of pd_zmarch_x2.inl to a chain of
// K stages.  Included inside the anonymous namespace of tv_kernels.hip (uses PdArgs, DualIO, pd_dual_t, pd_primal_t).
//
// One iteration moves 36 B/voxel and the two-iteration kernel is HBM-bound once the IEEE divide / sqrt are relaxed
// (profiles/archive/r2_b_pdtv_tile_vs_x2_pmc.txt), so the only lever left is bytes per iteration.  Stage s (iteration n+s ->
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
// Shipped instantiation (round 3): K = 3, RY = 8 rows per lane, 2 x 2 waves, LAG with 10 slots in registers -- 33 dual-row
// evaluations and 65 row loads per 24 output-row-iterations; float32 duals with relaxed arithmetic (FAST = 1), binary16
// duals with the reference's roundings (FAST = 2).
// Two forms of a step (`EDGE` below): the general one with every boundary select and activity test, and a SHORT one for
// waves whose lanes and row slots lie strictly inside the slice, on planes where every stage is active and none sits on
// the first / last plane: same operands, no selects (760 instead of 1480 VALU instructions per step).  The march runs
// them in separate loops: [general | short | general].
// compile-time stage loop: `s` must be a constant inside the stage body.  With a run-time (unrolled) loop the code of
// "s + 1 < K" for the last stage survives as a dead, not yet unrolled loop with variable indices into the register
// arrays until after the last scalar-replacement pass, which then leaves the arrays in scratch memory.
#ifndef XK_DB
#define XK_DB 1  // rows per block of the dual / primal updates (see pd_dual_block; 4 measured 11 % slower: the first
                 // block then waits for four rows of every input array instead of one)
#endif
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

template <typename T, bool NONNEG, bool ANISO, int FAST, int K, int RY, int WX, int WY, bool LAG = false, int LREG = 0, bool FIRST = false>
__global__ __launch_bounds__(64 * WX * WY) __attribute__((amdgpu_waves_per_eu(2))) void pd_zmarch_xk_kernel(PdArgs a, int gx, int gy, int tiles_per_xcd)
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

    // addressing: the lane keeps ONE byte offset (its clamped column), the row of slot i is wave-uniform and goes into the
    // buffer instruction's scalar offset (14 row offsets per lane cost 14 registers and pushed the two-path kernel into
    // scratch).  y0 is uniform per wave; readfirstlane tells the compiler so.
    const unsigned xo = (unsigned)xc * 4u;
    const int wy0 = __builtin_amdgcn_readfirstlane(y0);
    const int pitch = dx * 4;
    // a wave whose lanes and row slots all lie strictly inside the slice takes, on planes where every stage is active
    // and no stage sits on the first / last plane, a step without the boundary selects and activity tests (same values:
    // each select would pick the operand the short form uses).  Wave-uniform, kept in an SGPR.
    const int x_w0 = xs * (64 - 2 * K) - K;
    const bool xy_inner = __builtin_amdgcn_readfirstlane(
        (int)(x_w0 >= 1 && x_w0 + 63 <= dx - 2 && y0 - K >= 1 && y0 + RY + K - 1 <= dy - 2)) != 0;
    const PlaneIO io{(int)(sz * 4)};  // plane-relative buffer addressing, see tv_kernels.hip
    const PlaneIO io_pin{a.p_in_zero ? 0 : (int)(sz * 4)};  // input duals of the short form (see there)
    // byte offset of row slot i (rows -K .. RY+K-1) inside a float plane; EDGE: clamped into the slice
    auto rowoff = [&](int i, auto ec) __attribute__((always_inline)) {
        if constexpr (!decltype(ec)::value) {
            int yy = wy0 + i - K;
            if (a.probe & 1) yy = min(max(yy, yb * WY * RY), min((yb + 1) * WY * RY, dy) - 1);  // measurement only (uniform)
            return yy * pitch;
        } else {
            int yy = min(max(wy0 + i - K, 0), dy - 1);
            if (a.probe & 1) yy = min(max(yy, yb * WY * RY), min((yb + 1) * WY * RY, dy) - 1);
            return yy * pitch;
        }
    };
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
        for (int i = 0; i < NR; ++i) U0c[i] = io.ldf(up, xo, rowoff(i, std::true_type{}));
    }

    auto step = [&](const int t, auto ec) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(ec)::value;  // false: the short form (interior wave, steady planes)
        float U0n[NR];
        float Pw[3][NR];
        float Pnext[K][3][NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            U0n[i] = 0.0f;
            if constexpr (EDGE) { Pw[0][i] = 0.0f; Pw[1][i] = 0.0f; Pw[2][i] = 0.0f; }
        }
        const bool act0 = !EDGE || (t < dz);
        if (act0) {
            const bool z_last = EDGE && (t == dz - 1) && a.last_is_edge;
            const int zn = z_last ? max(t - 1, 0) : min(t + 1, dz - 1);
            const float *up = a.u_in + sz * ((a.probe & 4) ? 0 : zn);
#pragma unroll
            for (int i = 0; i < NR; ++i) U0n[i] = io.ldf(up, xo, rowoff(i, ec));
            if (z_last && t == 0) {
#pragma unroll
                for (int i = 0; i < NR; ++i) U0n[i] = 0.0f;
            }
            if constexpr (FIRST) {
                // first launch of a prox as its own instantiation: the duals are zero and Input IS the iterate -- no dual
                // is requested, and Input(t) below is the plane of U this lane already holds (26 requests per step, not 65)
#pragma unroll
                for (int i = 0; i < NR; ++i) { Pw[0][i] = 0.0f; Pw[1][i] = 0.0f; Pw[2][i] = 0.0f; }
            } else if constexpr (!EDGE) {
                // short form: no test.  On the first launch of a prox (duals are zero, nothing to read) the descriptor has
                // zero records: the range check answers every load with 0 and no request leaves the CU.
                // Requests go out row by row (U of the next plane was requested above, in row order, too): loads return in
                // order, so the dual update of row i can start once ITS four values are there.
                const T *pp0 = P_in[0] + sz * ((a.probe & 4) ? 0 : t);
                const T *pp1 = P_in[1] + sz * ((a.probe & 4) ? 0 : t);
                const T *pp2 = P_in[2] + sz * ((a.probe & 4) ? 0 : t);
#pragma unroll
                for (int i = 0; i < NR - 1; ++i) {
                    Pw[0][i] = io_pin.ldd(pp0, xo, rowoff(i, ec));
                    Pw[1][i] = io_pin.ldd(pp1, xo, rowoff(i, ec));
                    Pw[2][i] = io_pin.ldd(pp2, xo, rowoff(i, ec));
                }
                Pw[0][NR - 1] = 0.0f; Pw[1][NR - 1] = 0.0f; Pw[2][NR - 1] = 0.0f;
            } else if (!a.p_in_zero) {  // uniform: the first launch of a prox starts from zero duals (nothing to read)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const T *pp = P_in[c] + sz * ((a.probe & 4) ? 0 : t);
#pragma unroll
                    for (int i = 0; i < NR - 1; ++i) Pw[c][i] = io.ldd(pp, xo, rowoff(i, ec));
                }
            }
            if constexpr (FIRST) {
#pragma unroll
                for (int i = 1; i < NR - 1; ++i) In[0][i] = U0c[i];
            } else {
                const float *ip = a.in + sz * ((a.probe & 4) ? 0 : t);
#pragma unroll
                for (int i = 1; i < NR - 1; ++i) In[0][i] = io.ldf(ip, xo, rowoff(i, ec));
            }
            if (LAG) {  // Input(t) for the later stages: ring slot t mod K
                const int q = t % K;
#pragma unroll
                for (int i = 2; i < 2 + IN_ROWS; ++i) lag[IN_BASE - LREG + q * IN_ROWS + (i - 2)][tid] = In[0][i];
            }
        }
        xk_static_for<K>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            const int p = t - s;                                 // plane of this stage
            const bool act = !EDGE || ((p >= max(zc0 - (K - s), 0)) && (p < dz) && (s > 0 || act0));
            float Vn[NR];                                        // U^{n+s+1}(p), rows -(K-s-1) .. RY+(K-s-1)-1
#pragma unroll
            for (int i = 0; i < NR; ++i) Vn[i] = 0.0f;
            if (act) {
                const bool p_last = EDGE && (p == dz - 1) && a.last_is_edge;
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
                // ---------------- duals, rows -(K-s) .. RY+(K-s)-2, four rows at a time (see pd_dual_block)
                constexpr int DB = XK_DB, D0 = -(K - s), D1 = RY + (K - s) - 2;
#pragma unroll
                for (int rb = D0; rb <= D1; rb += DB) {
                    float gg[DB][3], pv[DB][3];
#pragma unroll
                    for (int k = 0; k < DB; ++k) {
                        const int r = rb + k <= D1 ? rb + k : D1;  // (a short last block repeats its last row: unused)
                        const int i = r + K;
                        const int y = y0 + r;
                        const float u = (s == 0) ? U0c[i] : Ur[s][1][i];
                        const float ux = __shfl_down(u, 1, 64);
                        const float uxm = __shfl_up(u, 1, 64);
                        gg[k][0] = (EDGE ? (x_last ? (x_has_prev ? uxm : 0.0f) : ux) : ux) - u;
                        const float u_up = (s == 0) ? U0c[i > 0 ? i - 1 : 0] : Ur[s][1][i > 0 ? i - 1 : 0];
                        const float u_dn = (s == 0) ? U0c[i + 1] : Ur[s][1][i + 1];
                        const float uy_mirror = (y > 0) ? u_up : 0.0f;  // the first row slot is never the volume's last row
                        gg[k][1] = (EDGE ? ((y == dy - 1) ? uy_mirror : u_dn) : u_dn) - u;
                        float uz;
                        if (s == 0) uz = U0n[i];  // stage 0: the loaded plane is already the mirrored one at the far edge
                        else uz = p_last ? ((p > 0) ? Ur[s][0][i] : 0.0f) : Ur[s][2][i];
                        gg[k][2] = uz - u;
                        pv[k][0] = Pw[0][i]; pv[k][1] = Pw[1][i]; pv[k][2] = Pw[2][i];
                    }
                    constexpr int left = D1 - D0 + 1;
                    const int nrows = (rb - D0 + DB <= left) ? DB : left - (rb - D0);
                    pd_dual_block<ANISO, FAST, DB>(pv, gg, a.sigma, nrows);
#pragma unroll
                    for (int k = 0; k < DB; ++k) {
                        if (rb + k <= D1) {
                            const int i = rb + k + K;
                            Pw[0][i] = pv[k][0]; Pw[1][i] = pv[k][1]; Pw[2][i] = pv[k][2];
                        }
                    }
                }
                // ---------------- primal, rows -(K-s-1) .. RY+(K-s-1)-1, four rows at a time
                const bool emit_plane = (s == K - 1) && (!EDGE || p >= zc0);
                constexpr int Q0 = -(K - s - 1), Q1 = RY + (K - s - 1) - 1;
                // output descriptors of this plane, built once (last stage only)
                const size_t po = sz * ((a.probe & 4) ? 0 : (s == K - 1 ? p : 0));
                const __amdgpu_buffer_rsrc_t r_uo = io.rsf(a.u_out + po);
                const __amdgpu_buffer_rsrc_t r_p0 = io.rsd(P_out[0] + po), r_p1 = io.rsd(P_out[1] + po), r_p2 = io.rsd(P_out[2] + po);
#pragma unroll
                for (int rb = Q0; rb <= Q1; rb += DB) {
                    float uu[DB], in4[DB], dv[DB], uo[DB];
#pragma unroll
                    for (int k = 0; k < DB; ++k) {
                        const int r = rb + k <= Q1 ? rb + k : Q1;
                        const int i = r + K;
                        const int y = y0 + r;
                        const float p1l = __shfl_up(Pw[0][i], 1, 64);
                        const float px = (!EDGE || x_has_prev) ? p1l : 0.0f;
                        const float py = (!EDGE || y > 0) ? Pw[1][i - 1] : 0.0f;
                        const float pz = (!EDGE || p > 0) ? c3[s][i] : 0.0f;
                        float div = (-(Pw[0][i] - px)) + (-(Pw[1][i] - py));
                        dv[k] = div + (-(Pw[2][i] - pz));
                        uu[k] = (s == 0) ? U0c[i] : Ur[s][1][i];
                        in4[k] = InS[i];
                    }
                    pd_primal_block<FAST, DB>(uo, uu, in4, dv, a.tau, a.lt, a.inv1lt, a.theta, NONNEG, (FAST == 1 && sizeof(T) == 4) ? a.nn_thr : 0.0f);
#pragma unroll
                    for (int k = 0; k < DB; ++k) {
                        if (rb + k <= Q1) {
                            const int i = rb + k + K;
                            const int y = y0 + rb + k;
                            Vn[i] = uo[k];
                            if constexpr (s == K - 1) {
                                if (emit_plane && emit_lane && (!EDGE || y < dy)) {
                                    PlaneIO::stf_rs(r_uo, xo, rowoff(i, ec), uo[k]);
                                    if (!a.p_out_skip) {  // uniform: nobody reads the duals of the last launch of a prox
                                        PlaneIO::std_rs(P_out[0], r_p0, xo, rowoff(i, ec), Pw[0][i]);
                                        PlaneIO::std_rs(P_out[1], r_p1, xo, rowoff(i, ec), Pw[1][i]);
                                        PlaneIO::std_rs(P_out[2], r_p2, xo, rowoff(i, ec), Pw[2][i]);
                                    }
                                }
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
                    // plane p-1 is only read by the far-edge mirror, at least one general step after the short form ended
                    if constexpr (EDGE) Ur[s + 1][0][i] = Ur[s + 1][1][i];
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
                    if constexpr (sizeof(T) == 2) {
                        // binary16 duals: what is handed over is the binary16 rounding of the dual anyway, so two of them
                        // share one LDS word (even slot of a pair; half the ds operations, three conversions per pair
                        // instead of four)
                        constexpr int ROWS = xk_dual_rows(K, RY, s + 1), NV = 3 * ROWS, B0 = xk_p_base(K, RY, s + 1);
#pragma unroll
                        for (int m = 0; m < NV; ++m) {
                            const int slot = B0 + m, c = m / ROWS, i = m % ROWS + (s + 1);
                            const bool first = (slot % 2 == 0) && slot >= LREG && m + 1 < NV;
                            const bool second = (slot % 2 == 1) && slot - 1 >= LREG && m >= 1;
                            if (second) continue;
                            if (first) {
                                const int c1 = (m + 1) / ROWS, i1 = (m + 1) % ROWS + (s + 1);
                                const float oldw = lag[slot - LREG][tid];
                                lag[slot - LREG][tid] = pack_half2_twice_rounded(Pw[c][i], Pw[c1][i1]);
                                unpack_half2(oldw, Pw[c][i], Pw[c1][i1]);
                            } else {
                                const float nxt = Pw[c][i];
                                float old;
                                if (slot < LREG) { old = lagreg[slot]; lagreg[slot] = nxt; }
                                else { old = lag[slot - LREG][tid]; lag[slot - LREG][tid] = nxt; }
                                Pw[c][i] = DualIO<T>::rt(old);
                            }
                        }
                    } else
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
    };

    // short form: every stage active (t >= zc0 + K - 2, t >= K - 1, t < dz), no stage on plane 0 (t >= K) or on the last
    // plane (t <= dz - 2), the last stage emits (t - K + 1 >= zc0).  The march is cut into [general | short | general]
    // ranges run by SEPARATE loops (an interior wave: warm-up planes, steady planes, drain; any other wave: everything in
    // the first general range): with both forms inside one loop the loop-carried state had to sit in the same registers
    // for both and the allocator spilled inside the march.  Every wave passes the barrier once per plane either way.
    const int tS0 = xy_inner ? min(max(zc0 + K - 1, K), tEnd + 1) : tEnd + 1;
    const int tS1 = xy_inner ? max(min(dz - 2, tEnd), tS0 - 1) : tEnd;
    for (int phase = 0; phase < 2; ++phase) {  // (one copy of the general form's code for both of its ranges)
        const int e0 = phase == 0 ? zA : tS1 + 1, e1 = phase == 0 ? tS0 - 1 : tEnd;
        for (int t = e0; t <= e1; ++t) {
            __syncthreads();  // lockstep: lines shared with the neighbouring waves merge in L1 (see pd_zmarch2)
            step(t, std::true_type{});
        }
        if (phase == 0 && tS0 <= tS1) {
            for (int t = tS0; t <= tS1; ++t) {
                __syncthreads();
                step(t, std::false_type{});
            }
            // the short form does not maintain plane p-1 of the later stages; nothing reads it before the next general
            // step has rotated it in again (that step is at most dz - 1, where only stage 0 sits on the last plane)
#pragma unroll
            for (int s = 1; s < K; ++s)
#pragma unroll
                for (int i = 0; i < NR; ++i) Ur[s][0][i] = 0.0f;
        }
    }
}

template <typename T, bool NONNEG, bool ANISO, int FAST, int K, int RY, int WX, int WY, bool LAG = false, int LREG = 0, bool FIRST = false>
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
    size_t dyn = 0;
#if TOMO_DEV
    // measurement only (tools/archive/probes/pd_halo_probe.py, bit 8): reserve 72 KiB of dynamic LDS on top of the kernel's own 80 KiB, so
    // that ONE workgroup fits a CU (one wave per SIMD instead of two) -- what a tiling that spends the second workgroup's
    // LDS on a prefetch buffer would have to live with
    if (a.probe & 8) {
        dyn = 72 * 1024;
        (void)hipFuncSetAttribute((const void *)pd_zmarch_xk_kernel<T, NONNEG, ANISO, FAST, K, RY, WX, WY, LAG, LREG, FIRST>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    }
#endif
    pd_zmarch_xk_kernel<T, NONNEG, ANISO, FAST, K, RY, WX, WY, LAG, LREG, FIRST><<<(unsigned)blocks, 64 * WX * WY, dyn, st>>>(a, gx, gy, tiles_per_xcd);
    return TOMO_OK;
}
