// NOT part of the build.  Round-2 experiment (PD_TV workgroup tile with LDS row halos), measured slower than the per-wave-halo
// kernels in every shape (docs/kernels/pd_tv.md, profiles/archive/r2_b_pdtv_tile_vs_x2_pmc.txt).  To rebuild it: include this file in
// tv_kernels.hip after pd_zmarch2.inl and dispatch pd_tile_launch<T, NN, AN, FAST, 4, 1, 8> from pd_multi_launch.
// PD_TV, TWO Chambolle-Pock iterations per pass through HBM, workgroup TILE with the row halos shared through LDS
// (3D only; default).  Included inside the anonymous namespace of tv_kernels.hip (uses PdArgs, DualIO, pd_dual, pd_primal).
//
// Same pipeline as pd_zmarch_x2.inl (stage A = iteration n -> n+1 on plane t, stage B = n+1 -> n+2 on plane t-1, the
// intermediate U / P never leave the chip), but the row halos are no longer RE-COMPUTED by every wave.  In the x2 kernel
// a wave produced RY = 4 output rows from 8 loaded rows of U, 7 of P1..3, 6 of Input and evaluated 7 + 5 dual rows:
// 35 row loads and 12 dual evaluations per 8 output-row-iterations; counters: 1.4-1.5x the compulsory read traffic
// (profiles/archive/r1_pdtv_pmc.txt).  Here a workgroup is a stack of WY waves that owns a tile of WY*RY consecutive rows; a wave
// loads and updates ONLY its own RY rows and hands the one row its neighbour needs (U^n, U^{n+1}: first and last row;
// P2^{n+1}, P2^{n+2}: last row) through LDS.  Only the tile as a whole carries a two-row halo top and bottom:
//     rows computed per output row  (WY*RY) / (WY*RY - 4) = 32/28 = 1.14   (x2 kernel: 1.75 loads, 1.5 duals)
// Registers drop from 210 to < 128, so 16 waves per CU stay resident (two 8-wave workgroups) instead of 8.
// Two barriers per plane separate  [A-dual | A-primal, B-dual | B-primal];  halo slots are single-buffered except the
// U^n rows, which are written a plane ahead (by parity).  x halos stay as in the x2 kernel: 2 lanes either side of a
// wave's 60 output columns, neighbours by DPP wave shifts.
// Garbage computed in the tile's outermost cells (clamped loads / stale LDS slots) never reaches an output cell: stage A
// duals are valid on tile rows [0, L-2], U^{n+1} on [1, L-2], stage B duals on [1, L-3], outputs on [2, L-3].
template <typename T, bool NONNEG, bool ANISO, bool FAST, int RY, int WX, int WY>
__global__ __launch_bounds__(64 * WX * WY, (WX * WY <= 8 ? 2 : (WX * WY <= 12 ? 3 : 4))) void pd_tile_kernel(PdArgs a, int gx, int gy, int gy_per_xcd)
{
    constexpr int NW = WX * WY, LY = RY * WY;
    __shared__ float hU_first[2][NW][64], hU_last[2][NW][64];  // U^n rows 0 / RY-1 of a plane, by plane parity
    __shared__ float hV_first[NW][64], hV_last[NW][64];        // U^{n+1}(t) rows 0 / RY-1
    __shared__ float hPa2[NW][64], hPb2[NW][64];               // P2^{n+1}(t) / P2^{n+2}(t-1), row RY-1

    int j = (int)blockIdx.x >> 3;
    const int xcd = (int)blockIdx.x & 7;
    const int xb = j % gx;
    j /= gx;
    const int yb = xcd * gy_per_xcd + (j % gy_per_xcd);
    const int chunk = j / gy_per_xcd;
    if (yb >= gy) return;  // uniform for the workgroup

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wx = wave % WX, wy = wave / WX;
    const int w_up = wy > 0 ? wave - WX : wave;        // the wave that owns the rows above (tile edge: own slot, unused)
    const int w_dn = wy < WY - 1 ? wave + WX : wave;   // ... below
    const int xs = xb * WX + wx;
    const int x = xs * 60 - 2 + lane;
    const int jr0 = wy * RY;                           // first tile row of this wave
    const int y0 = yb * (LY - 4) - 2 + jr0;
    const int dx = a.dx, dy = a.dy, dz = a.planes;
    const int zc0 = a.out_begin + chunk * a.zchunk;
    const int zc1 = min(zc0 + a.zchunk, a.out_end);
    if (zc0 >= zc1) return;  // uniform for the workgroup

    const size_t sz = (size_t)dx * dy;
    const bool x_last = (x == dx - 1);
    const bool x_has_prev = (x > 0);
    const bool emit_lane = (lane >= 2) && (lane <= 61) && (x < dx);
    const int xc = min(max(x, 0), dx - 1);

    unsigned off[RY];  // byte offsets of the wave's rows (clamped into the volume)
#pragma unroll
    for (int r = 0; r < RY; ++r) off[r] = (unsigned)(min(max(y0 + r, 0), dy - 1) * dx + xc) * 4u;
    auto ldf = [](const float *base, unsigned boff) { return *(const float *)((const char *)base + boff); };
    auto ldd = [](const T *base, unsigned boff) {
        return DualIO<T>::ld((const T *)((const char *)base + (sizeof(T) == 2 ? (boff >> 1) : boff)), 0);
    };
    const T *P_in[3] = {(const T *)a.p_in[0], (const T *)a.p_in[1], (const T *)a.p_in[2]};
    T *P_out[3] = {(T *)a.p_out[0], (T *)a.p_out[1], (T *)a.p_out[2]};

    // ---- persistent registers (own rows only)
    float Uc[RY];             // U^n(t)
    float V0[RY], V1[RY];     // U^{n+1}(t-2), U^{n+1}(t-1)
    float PaPrev[3][RY];      // P^{n+1}(t-1)
    float InPrev[RY];         // Input(t-1)
    float carryA3[RY];        // P3^{n+1}(t-1)
    float carryB3[RY];        // P3^{n+2}(t-2)
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        V0[r] = 0.0f; V1[r] = 0.0f; carryA3[r] = 0.0f; carryB3[r] = 0.0f; InPrev[r] = 0.0f;
        PaPrev[0][r] = 0.0f; PaPrev[1][r] = 0.0f; PaPrev[2][r] = 0.0f;
    }

    const int zA = max(zc0 - 2, 0);   // first plane of stage A (warm-up planes rebuild the carries)
    const int zB = max(zc0 - 1, 0);   // first plane of stage B
    const int tEnd = min(zc1, dz);    // inclusive: stage B must reach plane zc1-1
    {
        const float *up = a.u_in + sz * zA;
#pragma unroll
        for (int r = 0; r < RY; ++r) Uc[r] = ldf(up, off[r]);
        hU_first[zA & 1][wave][lane] = Uc[0];
        hU_last[zA & 1][wave][lane] = Uc[RY - 1];
        hV_first[wave][lane] = 0.0f;
        hV_last[wave][lane] = 0.0f;
    }
    // ---- software pipeline: the operands of step t (U^n(t+1), P^n(t), Input(t)) are loaded during step t-1 and wait
    //      in registers; nothing in a step waits for HBM except its very first instructions (barriers only drain LDS)
    float Un_pf[RY], P_pf[3][RY], In_pf[RY];
    auto issue_loads = [&](int t) {
        const int tt = min(t, dz - 1);  // t == dz (drain step): a valid, unused address
        const bool z_last = (tt == dz - 1) && a.last_is_edge;
        const int zn = z_last ? max(tt - 1, 0) : min(tt + 1, dz - 1);
        const float *up = a.u_in + sz * zn;
#pragma unroll
        for (int r = 0; r < RY; ++r) Un_pf[r] = ldf(up, off[r]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const T *pp = P_in[c] + sz * tt;
#pragma unroll
            for (int r = 0; r < RY; ++r) P_pf[c][r] = ldd(pp, off[r]);
        }
        const float *ip = a.in + sz * tt;
#pragma unroll
        for (int r = 0; r < RY; ++r) In_pf[r] = ldf(ip, off[r]);
    };
    issue_loads(zA);
    __syncthreads();

    for (int t = zA; t <= tEnd; ++t) {
        const bool stageA = (t < dz);
        const int s = t - 1;
        const bool stageB = (s >= zB);
        float V2[RY];        // U^{n+1}(t)
        float Pa[3][RY];     // P^n(t) -> P^{n+1}(t)
        float InA[RY];       // Input(t)
        float Un[RY];        // U^n(t+1) (or the mirrored plane at the far z edge)
        // ================= phase 1: loads, neighbour rows of U^n(t) and U^{n+1}(t-1), stage A duals
        const float v1_up = hV_last[w_up][lane], v1_dn = hV_first[w_dn][lane];
        if (stageA) {
            const bool z_last = (t == dz - 1) && a.last_is_edge;
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                Un[r] = (z_last && t == 0) ? 0.0f : Un_pf[r];  // mirrored "previous" plane of a one-plane volume is zero
                Pa[0][r] = P_pf[0][r]; Pa[1][r] = P_pf[1][r]; Pa[2][r] = P_pf[2][r];
                InA[r] = In_pf[r];
            }
            if (t + 1 < dz && t + 1 <= tEnd) issue_loads(t + 1);  // in flight across both barriers of this step
            const float uc_up = hU_last[t & 1][w_up][lane], uc_dn = hU_first[t & 1][w_dn][lane];
            hU_first[(t + 1) & 1][wave][lane] = Un[0];        // rows of plane t+1 for the next step
            hU_last[(t + 1) & 1][wave][lane] = Un[RY - 1];
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const int y = y0 + r;
                const float u = Uc[r];
                const float ux = __shfl_down(u, 1, 64);
                const float uxm = __shfl_up(u, 1, 64);
                float g[3];
                g[0] = (x_last ? (x_has_prev ? uxm : 0.0f) : ux) - u;
                const float u_prev_row = (r == 0) ? uc_up : Uc[r > 0 ? r - 1 : 0];
                const float u_next_row = (r == RY - 1) ? uc_dn : Uc[r < RY - 1 ? r + 1 : 0];
                const float uy_mirror = (y > 0) ? u_prev_row : 0.0f;
                g[1] = ((y == dy - 1) ? uy_mirror : u_next_row) - u;
                g[2] = Un[r] - u;
                float p[3] = {Pa[0][r], Pa[1][r], Pa[2][r]};
                pd_dual_t<ANISO, FAST>(p, g, a.sigma);
                Pa[0][r] = p[0]; Pa[1][r] = p[1]; Pa[2][r] = p[2];
            }
            hPa2[wave][lane] = Pa[1][RY - 1];
        }
        __syncthreads();
        // ================= phase 2: stage A primal, stage B duals (plane s = t - 1)
        float Pb[3][RY];  // P^{n+2}(s)
        if (stageA) {
            const float pa2_up = hPa2[w_up][lane];
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const int y = y0 + r;
                const float p1l = __shfl_up(Pa[0][r], 1, 64);
                const float px = x_has_prev ? p1l : 0.0f;
                const float py = (y > 0) ? ((r == 0) ? pa2_up : Pa[1][r > 0 ? r - 1 : 0]) : 0.0f;
                const float pz = (t > 0) ? carryA3[r] : 0.0f;
                float div = (-(Pa[0][r] - px)) + (-(Pa[1][r] - py));
                div = div + (-(Pa[2][r] - pz));
                V2[r] = pd_primal_t<FAST>(Uc[r], InA[r], div, a.tau, a.lt, a.inv1lt, a.theta, NONNEG);
            }
        } else {
#pragma unroll
            for (int r = 0; r < RY; ++r) V2[r] = 0.0f;
        }
        hV_first[wave][lane] = V2[0];  // read by the neighbours in phase 1 of the next step
        hV_last[wave][lane] = V2[RY - 1];
        if (stageB) {
            const bool s_last = (s == dz - 1) && a.last_is_edge;
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const int y = y0 + r;
                const float u = V1[r];
                const float ux = __shfl_down(u, 1, 64);
                const float uxm = __shfl_up(u, 1, 64);
                float g[3];
                g[0] = (x_last ? (x_has_prev ? uxm : 0.0f) : ux) - u;
                const float u_prev_row = (r == 0) ? v1_up : V1[r > 0 ? r - 1 : 0];
                const float u_next_row = (r == RY - 1) ? v1_dn : V1[r < RY - 1 ? r + 1 : 0];
                const float uy_mirror = (y > 0) ? u_prev_row : 0.0f;
                g[1] = ((y == dy - 1) ? uy_mirror : u_next_row) - u;
                const float uz = s_last ? ((s > 0) ? V0[r] : 0.0f) : V2[r];
                g[2] = uz - u;
                float p[3] = {PaPrev[0][r], PaPrev[1][r], PaPrev[2][r]};
                if (sizeof(T) == 2) {  // P^{n+1} as the next iteration would read it back from binary16 storage
                    p[0] = DualIO<T>::rt(p[0]); p[1] = DualIO<T>::rt(p[1]); p[2] = DualIO<T>::rt(p[2]);
                }
                pd_dual_t<ANISO, FAST>(p, g, a.sigma);
                Pb[0][r] = p[0]; Pb[1][r] = p[1]; Pb[2][r] = p[2];
            }
            hPb2[wave][lane] = Pb[1][RY - 1];
        }
        __syncthreads();
        // ================= phase 3: stage B primal and stores
        if (stageB) {
            const float pb2_up = hPb2[w_up][lane];
            const bool emit_plane = (s >= zc0);
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const int y = y0 + r;
                const float p1l = __shfl_up(Pb[0][r], 1, 64);
                const float px = x_has_prev ? p1l : 0.0f;
                const float py = (y > 0) ? ((r == 0) ? pb2_up : Pb[1][r > 0 ? r - 1 : 0]) : 0.0f;
                const float pz = (s > 0) ? carryB3[r] : 0.0f;
                float div = (-(Pb[0][r] - px)) + (-(Pb[1][r] - py));
                div = div + (-(Pb[2][r] - pz));
                const float uo = pd_primal_t<FAST>(V1[r], InPrev[r], div, a.tau, a.lt, a.inv1lt, a.theta, NONNEG);
                carryB3[r] = Pb[2][r];
                const bool emit_row = (jr0 + r >= 2) && (jr0 + r < LY - 2) && (y < dy);  // wave-uniform
                if (emit_plane && emit_row && emit_lane) {
                    *(float *)((char *)(a.u_out + sz * s) + off[r]) = uo;
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        DualIO<T>::st((T *)((char *)(P_out[c] + sz * s) + (sizeof(T) == 2 ? (off[r] >> 1) : off[r])), 0,
                                      Pb[c][r]);
                }
            }
        }
        // ================= rotate the pipeline registers
        if (stageA) {
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                V0[r] = V1[r]; V1[r] = V2[r]; carryA3[r] = Pa[2][r];
                PaPrev[0][r] = Pa[0][r]; PaPrev[1][r] = Pa[1][r]; PaPrev[2][r] = Pa[2][r];
                InPrev[r] = InA[r];
                Uc[r] = Un[r];
            }
        }
    }
}

template <typename T, bool NONNEG, bool ANISO, bool FAST, int RY, int WX, int WY>
static int pd_tile_launch(PdArgs a, hipStream_t st)
{
    const int nout = a.out_end - a.out_begin;
    const int gx = ceil_div(ceil_div(a.dx, 60), WX), gy = ceil_div(a.dy, WY * RY - 4);
    const int gy_per_xcd = ceil_div(gy, 8);
    const long waves_xy = (long)gx * gy * WX * WY;
    // enough z-chunks for ~3 rounds of fully resident workgroups (16 waves per CU), but each chunk long enough to
    // amortise its two warm-up planes
    int chunks = (int)((256L * 16 * 3 + waves_xy - 1) / waves_xy);
    const int max_chunks = ceil_div(nout, 64);
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    a.zchunk = ceil_div(nout, chunks);
    chunks = ceil_div(nout, a.zchunk);
    a.inv1lt = 1.0f / (1.0f + a.lt);
    const long blocks = 8L * gx * gy_per_xcd * chunks;
    if (blocks > 0x7fffffffL) return tomo_fail(TOMO_E_INVALID, "volume too large for one PD_TV launch");
    pd_tile_kernel<T, NONNEG, ANISO, FAST, RY, WX, WY><<<(unsigned)blocks, 64 * WX * WY, 0, st>>>(a, gx, gy, gy_per_xcd);
    return TOMO_OK;
}
