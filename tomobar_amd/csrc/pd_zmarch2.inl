// PD_TV "zmarch2": wave-autonomous register-blocked z-march, second generation.  Included inside the anonymous
// namespace of tv_kernels.hip (uses PdArgs, DualIO, pd_dual, pd_primal).
//
// Differences from the first generation (pd_zmarch_kernel):
//   * every load is unconditional on a clamped, always-valid address; validity is applied with selects afterwards.
//     (Guarded loads compile to one exec-masked basic block each, which serialises the memory stream.)
//   * lanes 1..62 of a wave produce output; lane 0 is the -x halo lane (its dual is consumed by lane 1) and lane 63
//     the +x halo lane (its U is consumed by lane 62), so x neighbours are pure wave shuffles.
//   * workgroups are numbered so that each XCD (workgroup id % 8) owns a contiguous band of rows: the one-row /
//     one-lane halos that neighbouring waves re-read are then served by that XCD's L2.
// WX x WY waves per workgroup: WX consecutive x segments times WY consecutive row groups.  With LOCKSTEP the waves
// of a workgroup walk z together (one barrier per plane) so that cache lines straddling two x segments and the halo
// rows are requested by both users at the same moment and merge in the CU's L1 instead of becoming two HBM requests.
template <typename T, int ND, bool NONNEG, bool ANISO, int FAST, int RY, bool LOCKSTEP, int WX, int WY>
__global__ __launch_bounds__(64 * WX * WY) void pd_zmarch2_kernel(PdArgs a, int gx, int gy, int tiles_per_xcd)
{
    // ---- XCD-aware workgroup numbering (gx, gy count workgroups)
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
    const int x = xs * 62 - 1 + lane;
    const int y0 = (yb * WY + (wave / WX)) * RY;
    const int dx = a.dx, dy = a.dy;
    if (!LOCKSTEP && (y0 >= dy || xs * 62 >= dx)) return;  // idle wave (LOCKSTEP keeps it for the barriers; it never stores)
    const int zc0 = a.out_begin + chunk * a.zchunk;
    const int zc1 = min(zc0 + a.zchunk, a.out_end);
    if (zc0 >= zc1) return;

    const size_t sz = (size_t)dx * dy;
    const bool x_last = (x == dx - 1);
    const bool x_has_prev = (x > 0);
    const bool emit_lane = (lane >= 1) && (lane <= 62) && (x < dx);
    const int xc = min(max(x, 0), dx - 1);

    // in-plane BYTE offsets (float arrays) of row slots -1..RY (clamped rows), index r+1.  Unsigned 32-bit so that
    // every access is `global_load/store v, v_off, s[plane_base]` (scalar base + 32-bit lane offset).
    unsigned off[RY + 2];
#pragma unroll
    for (int r = -1; r <= RY; ++r) off[r + 1] = (unsigned)(min(max(y0 + r, 0), dy - 1) * dx + xc) * 4u;
    const PlaneIO io{(int)(sz * 4)};  // plane-relative buffer addressing, see tv_kernels.hip
    auto ldf = [&](const float *base, unsigned boff) { return io.ldf(base, boff); };
    auto ldd = [&](const T *base, unsigned boff) { return io.ldd(base, boff); };

    const T *P_in[3] = {(const T *)a.p_in[0], (const T *)a.p_in[1], (const T *)a.p_in[2]};
    T *P_out[3] = {(T *)a.p_out[0], (T *)a.p_out[1], (T *)a.p_out[2]};

    float Uc[RY + 2], Un[RY + 2], carry3[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) carry3[r] = 0.0f;

    const int zstart = (ND == 3 && zc0 > 0) ? zc0 - 1 : zc0;  // warm-up plane builds carry3
    {
        const float *up = a.u_in + sz * zstart;
#pragma unroll
        for (int r = 0; r < RY + 2; ++r) Uc[r] = ldf(up, off[r]);
    }

    for (int z = zstart; z < zc1; ++z) {
        if (LOCKSTEP) __syncthreads();  // the four waves of a workgroup walk z together: halo rows hit L1/L2
        // ---- issue every load of this step first
        float Pl[3][RY + 1], In[RY];
        if (ND == 3) {
            const bool z_last = (z == a.planes - 1) && a.last_is_edge;
            const int zn = z_last ? max(z - 1, 0) : min(z + 1, a.planes - 1);
            const float *up = a.u_in + sz * zn;
#pragma unroll
            for (int r = 0; r < RY + 2; ++r) Un[r] = ldf(up, off[r]);
            if (z_last && z == 0) {  // mirrored "previous" plane of a one-plane volume is zero
#pragma unroll
                for (int r = 0; r < RY + 2; ++r) Un[r] = 0.0f;
            }
        }
#pragma unroll
        for (int c = 0; c < ND; ++c) {
            const T *pp = P_in[c] + sz * z;
#pragma unroll
            for (int r = 0; r < RY + 1; ++r) Pl[c][r] = ldd(pp, off[r]);
        }
        {
            const float *ip = a.in + sz * z;
#pragma unroll
            for (int r = 0; r < RY; ++r) In[r] = ldf(ip, off[r + 1]);
        }
        // ---- duals of row slots -1..RY-1 (index r+1)
        float Pn[3][RY + 1];
#pragma unroll
        for (int r = -1; r < RY; ++r) {
            const int y = y0 + r;
            const float u = Uc[r + 1];
            const float ux = __shfl_down(u, 1, 64);
            const float uxm = __shfl_up(u, 1, 64);
            float g[3] = {0.0f, 0.0f, 0.0f};
            g[0] = (x_last ? (x_has_prev ? uxm : 0.0f) : ux) - u;
            const float uy_mirror = (y > 0) ? Uc[r >= 0 ? r : 0] : 0.0f;  // row y-1 (slot -1 is never the last row)
            g[1] = ((y == dy - 1) ? uy_mirror : Uc[r + 2]) - u;
            if (ND == 3) g[2] = Un[r + 1] - u;
            float p[3] = {Pl[0][r + 1], Pl[1][r + 1], ND == 3 ? Pl[2][r + 1] : 0.0f};
            pd_dual_t<ANISO, FAST, ND>(p, g, a.sigma);
#pragma unroll
            for (int c = 0; c < ND; ++c) Pn[c][r + 1] = p[c];
        }
        // ---- primal step and stores for row slots 0..RY-1
        const bool emit_plane = (z >= zc0);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const int y = y0 + r;
            const float p1l = __shfl_up(Pn[0][r + 1], 1, 64);
            const float px = x_has_prev ? p1l : 0.0f;
            const float py = (y > 0) ? Pn[1][r] : 0.0f;
            float div = (-(Pn[0][r + 1] - px)) + (-(Pn[1][r + 1] - py));
            if (ND == 3) {
                const float pz = (z > 0) ? carry3[r] : 0.0f;
                div = div + (-(Pn[2][r + 1] - pz));
                carry3[r] = Pn[2][r + 1];
            }
            const float uo = pd_primal_t<FAST>(Uc[r + 1], In[r], div, a.tau, a.lt, a.inv1lt, a.theta, NONNEG);
            if (emit_plane && emit_lane && y < dy) {
                io.stf(a.u_out + sz * z, off[r + 1], uo);
#pragma unroll
                for (int c = 0; c < ND; ++c) io.std_(P_out[c] + sz * z, off[r + 1], Pn[c][r + 1]);
            }
        }
        if (ND == 3) {
#pragma unroll
            for (int r = 0; r < RY + 2; ++r) Uc[r] = Un[r];
        }
    }
}

template <typename T, int ND, bool NONNEG, bool ANISO, int FAST, int RY, bool LOCKSTEP, int WX = 1, int WY = 4>
static int pd_zmarch2_launch(PdArgs a, hipStream_t st)
{
    const int nout = a.out_end - a.out_begin;
    const int gx = ceil_div(ceil_div(a.dx, 62), WX), gy = ceil_div(a.dy, WY * RY);
    const int tiles_per_xcd = ceil_div(gx * gy, 8);
    // z-chunks: enough waves to fill the chip (~8 per SIMD), each long enough to amortise its warm-up plane
    int chunks = 1;
    if (ND == 3) {
        const long waves_xy = (long)gx * gy * WX * WY;
        // measured: 48 waves per SIMD's worth of z-chunks (shorter marches, better balance) beats 8-16 by ~5 %
        const long want_per_simd = 48;
        const long want = 256L * 4 * want_per_simd;
        chunks = (int)((want + waves_xy - 1) / waves_xy);
        const int max_chunks = ceil_div(nout, 32);
        if (chunks > max_chunks) chunks = max_chunks;
        if (chunks < 1) chunks = 1;
    }
    a.zchunk = ceil_div(nout, chunks);
    chunks = ceil_div(nout, a.zchunk);
    a.inv1lt = 1.0f / (1.0f + a.lt);
    const long blocks = 8L * tiles_per_xcd * chunks;
    if (blocks > 0x7fffffffL) return tomo_fail(TOMO_E_INVALID, "volume too large for one PD_TV launch");
    pd_zmarch2_kernel<T, ND, NONNEG, ANISO, FAST, RY, LOCKSTEP, WX, WY><<<(unsigned)blocks, 64 * WX * WY, 0, st>>>(a, gx, gy, tiles_per_xcd);
    return TOMO_OK;
}
