// PD_TV on a 2D image, K Chambolle-Pock iterations per pass through HBM (round 4; the reference's 2D kernels:
// primal_dual_for_total_variation.cu:360-452).  Included inside the anonymous namespace of tv_kernels.hip (uses PdArgs,
// PlaneIO, DualIO, pd_dual_t, pd_primal_t).
//
// Until round 3 a 2D input ran one iteration per launch through the z-march kernel with a single plane: 28 B/pixel per
// iteration and one launch per iteration.  Here a wave owns a tile of RY rows x (64 - 2K) columns and keeps the tile plus
// a K-deep halo (RY + 2K rows x 64 lanes of U, Input, P1, P2) in registers; iteration s of the launch evaluates the duals
// on rows -(K-s) .. RY+(K-s)-2 and the primal variable on rows -(K-s-1) .. RY+(K-s-1)-1 of that tile (lanes s .. 63-s /
// s+1 .. 62-s), so after K iterations rows 0 .. RY-1 and lanes K .. 63-K hold U^{n+K}, P^{n+K} exactly as K separate
// launches would have produced them: 28 B/pixel per K iterations plus the halo re-reads, no LDS, no barrier.
// +-y neighbours are other registers of the lane, +-x neighbours come from DPP wave shifts.  Halo rows / lanes outside
// the image are loaded from clamped addresses; every use of such a value is behind the reference's own edge rule (mirror
// at the far edge, zero "previous" dual at index 0), so it never reaches a stored pixel.
// Arithmetic per pixel and iteration is that of the single-iteration kernels (same helpers), whatever K is; binary16
// duals are rounded through binary16 between the iterations of a launch exactly where the reference stores them.
template <typename T, bool NONNEG, bool ANISO, int FAST, int K, int RY>
__global__ __launch_bounds__(256) void pd_rows2d_kernel(PdArgs a, int gx, int gy)
{
    constexpr int NR = RY + 2 * K;
    const int lane = threadIdx.x & 63;
    const int tile = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    if (tile >= gx * gy) return;  // wave-uniform
    const int xb = tile % gx, yb = tile / gx;
    const int dx = a.dx, dy = a.dy;
    const int x = xb * (64 - 2 * K) - K + lane;
    const int y0 = __builtin_amdgcn_readfirstlane(yb * RY);
    const bool x_last = (x == dx - 1);
    const bool x_has_prev = (x > 0);
    const bool emit_lane = (lane >= K) && (lane <= 63 - K) && (x < dx);
    const unsigned xo = (unsigned)min(max(x, 0), dx - 1) * 4u;
    const int pitch = dx * 4;
    const PlaneIO io{dx * dy * 4};
    auto rowoff = [&](int i) __attribute__((always_inline)) { return min(max(y0 + i - K, 0), dy - 1) * pitch; };
    const T *P_in0 = (const T *)a.p_in[0], *P_in1 = (const T *)a.p_in[1];
    T *P_out0 = (T *)a.p_out[0], *P_out1 = (T *)a.p_out[1];

    float U[NR], In[NR], P0[NR], P1[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        U[i] = io.ldf(a.u_in, xo, rowoff(i));
        In[i] = io.ldf(a.in, xo, rowoff(i));
        P0[i] = 0.0f; P1[i] = 0.0f;
    }
    if (!a.p_in_zero) {  // uniform: the first launch of a prox starts from zero duals (nothing to read)
#pragma unroll
        for (int i = 0; i < NR - 1; ++i) {
            P0[i] = io.ldd(P_in0, xo, rowoff(i));
            P1[i] = io.ldd(P_in1, xo, rowoff(i));
        }
    }
#pragma unroll
    for (int s = 0; s < K; ++s) {
        // ---- duals of iteration s, rows -(K-s) .. RY+(K-s)-2
#pragma unroll
        for (int r = -(K - s); r <= RY + (K - s) - 2; ++r) {
            const int i = r + K;
            const int y = y0 + r;
            const float u = U[i];
            const float ux = wave_next(u), uxm = wave_prev(u);
            float g[3], p[3];
            g[0] = (x_last ? (x_has_prev ? uxm : 0.0f) : ux) - u;           // forward difference, mirrored at the far edge (:385-389)
            g[1] = ((y == dy - 1) ? ((y > 0) ? U[i > 0 ? i - 1 : 0] : 0.0f) : U[i + 1]) - u;
            g[2] = 0.0f;
            p[0] = P0[i]; p[1] = P1[i]; p[2] = 0.0f;
            pd_dual_t<ANISO, FAST, 2>(p, g, a.sigma);
            P0[i] = p[0]; P1[i] = p[1];
        }
        // ---- primal variable of iteration s, rows -(K-s-1) .. RY+(K-s-1)-1 (descending: row i reads the dual of row i-1)
#pragma unroll
        for (int r = RY + (K - s - 1) - 1; r >= -(K - s - 1); --r) {
            const int i = r + K;
            const int y = y0 + r;
            const float p0l = wave_prev(P0[i]);
            const float px = x_has_prev ? p0l : 0.0f;
            const float py = (y > 0) ? P1[i - 1] : 0.0f;
            const float div = (-(P0[i] - px)) + (-(P1[i] - py));
            U[i] = pd_primal_t<FAST>(U[i], In[i], div, a.tau, a.lt, a.inv1lt, a.theta, NONNEG);
        }
        if (sizeof(T) == 2 && s + 1 < K) {  // what the next iteration reads back is the stored binary16 value
#pragma unroll
            for (int i = 0; i < NR; ++i) { P0[i] = DualIO<T>::rt(P0[i]); P1[i] = DualIO<T>::rt(P1[i]); }
        }
    }
    if (!emit_lane) return;
    const __amdgpu_buffer_rsrc_t r_u = io.rsf(a.u_out), r_p0 = io.rsd(P_out0), r_p1 = io.rsd(P_out1);
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        if (y0 + r < dy) {
            const int ro = (y0 + r) * pitch;
            PlaneIO::stf_rs(r_u, xo, ro, U[r + K]);
            if (!a.p_out_skip) {  // uniform: nobody reads the duals of the last launch of a prox
                PlaneIO::std_rs(P_out0, r_p0, xo, ro, P0[r + K]);
                PlaneIO::std_rs(P_out1, r_p1, xo, ro, P1[r + K]);
            }
        }
    }
}

template <typename T, bool NONNEG, bool ANISO, int FAST, int K, int RY>
static int pd_rows2d_launch(PdArgs a, hipStream_t st)
{
    const int gx = ceil_div(a.dx, 64 - 2 * K), gy = ceil_div(a.dy, RY);
    const long tiles = (long)gx * gy;
    if ((tiles + 3) / 4 > 0x7fffffffL) return tomo_fail(TOMO_E_INVALID, "image too large for one PD_TV launch");
    a.inv1lt = 1.0f / (1.0f + a.lt);
    pd_rows2d_kernel<T, NONNEG, ANISO, FAST, K, RY><<<(unsigned)((tiles + 3) / 4), 256, 0, st>>>(a, gx, gy);
    return TOMO_OK;
}
