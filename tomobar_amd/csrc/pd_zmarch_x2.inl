// PD_TV, TWO Chambolle-Pock iterations per pass through HBM (3D only).  Included inside the anonymous namespace of
// tv_kernels.hip (uses PdArgs, DualIO, pd_dual, pd_primal).
//
// One iteration moves 36 B/voxel (read Input, U, P1..3; write U, P1..3) and is bound by the number of 128-B / 64-B
// fabric requests (profiles/archive/r1_pdtv_pmc.txt).  Here stage A (iteration n -> n+1) runs one plane ahead of stage B
// (n+1 -> n+2) inside the same z-march, so U^{n+1} and P^{n+1} never leave the register file:
//
//   step t:  stage A on plane t    : loads U^n(t+1), P^n(t), Input(t);   P^{n+1}(t), U^{n+1}(t)      (registers only)
//            stage B on plane t-1  : uses U^{n+1}(t-1), U^{n+1}(t), P^{n+1}(t-1), Input(t-1);  stores U^{n+2}, P^{n+2}
//
// i.e. 36 B/voxel per TWO iterations.  The price is a wider halo: per wave stage A covers rows -2..RY and lanes 0..62,
// stage B rows -1..RY-1 and lanes 1..61, output rows 0..RY-1 and lanes 2..61 (60 of 64).  Same workgroup shape and
// lockstep barrier as pd_zmarch2 so that the overlapping rows / lines of neighbouring waves merge in L1.
// Arithmetic and rounding are those of two successive single iterations (bit-identical; tests/test_gpu_parity.py).
template <typename T, bool NONNEG, bool ANISO, int FAST, int RY, int WX, int WY>
__global__ __launch_bounds__(64 * WX * WY) __attribute__((amdgpu_waves_per_eu(1, sizeof(T) == 4 ? 2 : 8))) void pd_zmarch_x2_kernel(PdArgs a, int gx, int gy, int tiles_per_xcd)
{
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
    const int x = xs * 60 - 2 + lane;
    const int y0 = (yb * WY + (wave / WX)) * RY;
    const int dx = a.dx, dy = a.dy, dz = a.planes;
    const int zc0 = a.out_begin + chunk * a.zchunk;
    const int zc1 = min(zc0 + a.zchunk, a.out_end);
    if (zc0 >= zc1) return;  // uniform for the workgroup

    const size_t sz = (size_t)dx * dy;
    const bool x_last = (x == dx - 1);
    const bool x_has_prev = (x > 0);
    const bool emit_lane = (lane >= 2) && (lane <= 61) && (x < dx);
    const int xc = min(max(x, 0), dx - 1);

    // byte offsets of row slots -2..RY+1 (clamped), index r+2
    unsigned off[RY + 4];
#pragma unroll
    for (int r = -2; r <= RY + 1; ++r) off[r + 2] = (unsigned)(min(max(y0 + r, 0), dy - 1) * dx + xc) * 4u;
    const PlaneIO io{(int)(sz * 4)};  // plane-relative buffer addressing, see tv_kernels.hip
    auto ldf = [&](const float *base, unsigned boff) { return io.ldf(base, boff); };
    auto ldd = [&](const T *base, unsigned boff) { return io.ldd(base, boff); };
    const T *P_in[3] = {(const T *)a.p_in[0], (const T *)a.p_in[1], (const T *)a.p_in[2]};
    T *P_out[3] = {(T *)a.p_out[0], (T *)a.p_out[1], (T *)a.p_out[2]};

    // ---- persistent registers
    float Uc[RY + 4];            // U^n(t),   rows -2..RY+1
    float V0[RY + 2], V1[RY + 2];  // U^{n+1}(t-2), U^{n+1}(t-1), rows -1..RY (index r+1)
    float PaPrev[3][RY + 1];     // P^{n+1}(t-1), rows -1..RY-1 (index r+1)
    float InPrev[RY];            // Input(t-1), rows 0..RY-1
    float carryA3[RY + 2];       // P3^{n+1}(t-1), rows -1..RY (index r+1)
    float carryB3[RY];           // P3^{n+2}(t-2), rows 0..RY-1
#pragma unroll
    for (int r = 0; r < RY + 2; ++r) { V0[r] = 0.0f; V1[r] = 0.0f; carryA3[r] = 0.0f; }
#pragma unroll
    for (int r = 0; r < RY; ++r) { carryB3[r] = 0.0f; InPrev[r] = 0.0f; }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < RY + 1; ++r) PaPrev[c][r] = 0.0f;

    const int zA = max(zc0 - 2, 0);   // first plane of stage A (warm-up planes rebuild the carries)
    const int zB = max(zc0 - 1, 0);   // first plane of stage B
    const int tEnd = min(zc1, dz);    // inclusive: stage B must reach plane zc1-1
    {
        const float *up = a.u_in + sz * zA;
#pragma unroll
        for (int r = 0; r < RY + 4; ++r) Uc[r] = ldf(up, off[r]);
    }

    for (int t = zA; t <= tEnd; ++t) {
        __syncthreads();  // lockstep (see pd_zmarch2)
        const bool stageA = (t < dz);
        float V2[RY + 2];        // U^{n+1}(t), rows -1..RY
        float Pa[3][RY + 3];     // P^{n+1}(t), rows -2..RY (index r+2)
        float InA[RY + 2];       // Input(t), rows -1..RY
        float Un[RY + 4];
        if (stageA) {
            // ---------------- loads of step t
            const bool z_last = (t == dz - 1) && a.last_is_edge;
            const int zn = z_last ? max(t - 1, 0) : min(t + 1, dz - 1);
            {
                const float *up = a.u_in + sz * zn;
#pragma unroll
                for (int r = 0; r < RY + 4; ++r) Un[r] = ldf(up, off[r]);
                if (z_last && t == 0) {
#pragma unroll
                    for (int r = 0; r < RY + 4; ++r) Un[r] = 0.0f;
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const T *pp = P_in[c] + sz * t;
#pragma unroll
                for (int r = 0; r < RY + 3; ++r) Pa[c][r] = ldd(pp, off[r]);
            }
            {
                const float *ip = a.in + sz * t;
#pragma unroll
                for (int r = 0; r < RY + 2; ++r) InA[r] = ldf(ip, off[r + 1]);
            }
            // ---------------- stage A duals, rows -2..RY
#pragma unroll
            for (int r = -2; r <= RY; ++r) {
                const int y = y0 + r;
                const float u = Uc[r + 2];
                const float ux = __shfl_down(u, 1, 64);
                const float uxm = __shfl_up(u, 1, 64);
                float g[3];
                g[0] = (x_last ? (x_has_prev ? uxm : 0.0f) : ux) - u;
                const float uy_mirror = (y > 0) ? Uc[r + 1 >= 0 ? r + 1 : 0] : 0.0f;  // rows -2,-1 are never the last row
                g[1] = ((y == dy - 1) ? uy_mirror : Uc[r + 3]) - u;
                g[2] = Un[r + 2] - u;
                float p[3] = {Pa[0][r + 2], Pa[1][r + 2], Pa[2][r + 2]};
                pd_dual_t<ANISO, FAST>(p, g, a.sigma);
                Pa[0][r + 2] = p[0]; Pa[1][r + 2] = p[1]; Pa[2][r + 2] = p[2];
            }
            // ---------------- stage A primal U^{n+1}(t), rows -1..RY
#pragma unroll
            for (int r = -1; r <= RY; ++r) {
                const int y = y0 + r;
                const float p1l = __shfl_up(Pa[0][r + 2], 1, 64);
                const float px = x_has_prev ? p1l : 0.0f;
                const float py = (y > 0) ? Pa[1][r + 1] : 0.0f;
                const float pz = (t > 0) ? carryA3[r + 1] : 0.0f;
                float div = (-(Pa[0][r + 2] - px)) + (-(Pa[1][r + 2] - py));
                div = div + (-(Pa[2][r + 2] - pz));
                V2[r + 1] = pd_primal_t<FAST>(Uc[r + 2], InA[r + 1], div, a.tau, a.lt, a.inv1lt, a.theta, NONNEG);
            }
        } else {
#pragma unroll
            for (int r = 0; r < RY + 2; ++r) V2[r] = 0.0f;
        }
        // ---------------- stage B on plane s = t - 1
        const int s = t - 1;
        if (s >= zB) {  // uniform
            const bool s_last = (s == dz - 1) && a.last_is_edge;
            float Pb[3][RY + 1];  // P^{n+2}(s), rows -1..RY-1 (index r+1)
#pragma unroll
            for (int r = -1; r < RY; ++r) {
                const int y = y0 + r;
                const float u = V1[r + 1];
                const float ux = __shfl_down(u, 1, 64);
                const float uxm = __shfl_up(u, 1, 64);
                float g[3];
                g[0] = (x_last ? (x_has_prev ? uxm : 0.0f) : ux) - u;
                const float uy_mirror = (y > 0) ? V1[r >= 0 ? r : 0] : 0.0f;  // row -1 is never the last row
                g[1] = ((y == dy - 1) ? uy_mirror : V1[r + 2]) - u;
                const float uz = s_last ? ((s > 0) ? V0[r + 1] : 0.0f) : V2[r + 1];
                g[2] = uz - u;
                float p[3] = {PaPrev[0][r + 1], PaPrev[1][r + 1], PaPrev[2][r + 1]};
                if (sizeof(T) == 2) {  // P^{n+1} as the next iteration would read it back from binary16 storage
                    p[0] = DualIO<T>::rt(p[0]); p[1] = DualIO<T>::rt(p[1]); p[2] = DualIO<T>::rt(p[2]);
                }
                pd_dual_t<ANISO, FAST>(p, g, a.sigma);
                Pb[0][r + 1] = p[0]; Pb[1][r + 1] = p[1]; Pb[2][r + 1] = p[2];
            }
            const bool emit_plane = (s >= zc0);
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const int y = y0 + r;
                const float p1l = __shfl_up(Pb[0][r + 1], 1, 64);
                const float px = x_has_prev ? p1l : 0.0f;
                const float py = (y > 0) ? Pb[1][r] : 0.0f;
                const float pz = (s > 0) ? carryB3[r] : 0.0f;
                float div = (-(Pb[0][r + 1] - px)) + (-(Pb[1][r + 1] - py));
                div = div + (-(Pb[2][r + 1] - pz));
                const float uo = pd_primal_t<FAST>(V1[r + 1], InPrev[r], div, a.tau, a.lt, a.inv1lt, a.theta, NONNEG);
                carryB3[r] = Pb[2][r + 1];
                if (emit_plane && emit_lane && y < dy) {
                    io.stf(a.u_out + sz * s, off[r + 2], uo);
#pragma unroll
                    for (int c = 0; c < 3; ++c) io.std_(P_out[c] + sz * s, off[r + 2], Pb[c][r + 1]);
                }
            }
        }
        // ---------------- rotate the pipeline registers
        if (stageA) {
#pragma unroll
            for (int r = 0; r < RY + 2; ++r) { V0[r] = V1[r]; V1[r] = V2[r]; carryA3[r] = Pa[2][r + 1]; }
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < RY + 1; ++r) PaPrev[c][r] = Pa[c][r + 1];
#pragma unroll
            for (int r = 0; r < RY; ++r) InPrev[r] = InA[r + 1];
#pragma unroll
            for (int r = 0; r < RY + 4; ++r) Uc[r] = Un[r];
        }
    }
}

template <typename T, bool NONNEG, bool ANISO, int FAST, int RY, int WX, int WY>
static int pd_zmarch_x2_launch(PdArgs a, hipStream_t st)
{
    const int nout = a.out_end - a.out_begin;
    const int gx = ceil_div(ceil_div(a.dx, 60), WX), gy = ceil_div(a.dy, WY * RY);
    const int tiles_per_xcd = ceil_div(gx * gy, 8);
    const long waves_xy = (long)gx * gy * WX * WY;
    const long want_per_simd = 32;
    int chunks = (int)((256L * 4 * want_per_simd + waves_xy - 1) / waves_xy);
    const int max_chunks = ceil_div(nout, 48);  // two warm-up planes per chunk: keep chunks long
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    a.zchunk = ceil_div(nout, chunks);
    chunks = ceil_div(nout, a.zchunk);
    a.inv1lt = 1.0f / (1.0f + a.lt);
    const long blocks = 8L * tiles_per_xcd * chunks;
    if (blocks > 0x7fffffffL) return tomo_fail(TOMO_E_INVALID, "volume too large for one PD_TV launch");
    pd_zmarch_x2_kernel<T, NONNEG, ANISO, FAST, RY, WX, WY><<<(unsigned)blocks, 64 * WX * WY, 0, st>>>(a, gx, gy, tiles_per_xcd);
    return TOMO_OK;
}
