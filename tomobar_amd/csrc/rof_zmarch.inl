// ROF_TV iteration as a register-blocked z-march (same skeleton as pd_zmarch2).  Included inside the anonymous
// namespace of tv_kernels.hip (uses RofArgs, rof_mm, rof_norm, DualIO).
//
// The reference runs two kernels per iteration (divergence_kernel_* writes D1..D3, TV_kernel_* reads them back:
// 40 B/voxel, rudin_osher_fatemi_total_variation.cu:106-148,203-248).  The per-voxel fused kernel needs no D arrays but
// evaluates D four times per voxel from cache.  Here a lane owns RY rows of one x column and marches z: D is evaluated
// ONCE per voxel (+ one halo row / two halo lanes), D1/D2 neighbours are registers / wave shuffles, D3 of the plane
// below is carried, so the iteration moves 12 B/voxel (read U, Input; write U).  The only look-ahead is the global
// z = 0 plane, whose backward difference reflects to D3 of plane 1 (:228-235): it is evaluated in the first step.
struct RofD { float d1, d2, d3; };

template <int ND, bool HALF, int FAST>
__device__ __forceinline__ RofD rof_eval(float u, float u_i1, float u_i2, float u_j1, float u_j2, float u_k1, float u_k2)
{
    // reference naming: "x" differences run along j (rows), "y" along i (lanes)  (rudin_osher...cu:183-188)
    const float nx1 = u_j1 - u, nx0 = u - u_j2;
    const float ny1 = u_i1 - u, ny0 = u - u_i2;
    const float dxm = rof_mm(nx0, nx1), dym = rof_mm(ny0, ny1);
    RofD d;
    // FAST: float32 sum + v_rsq_f32 instead of the reference's double-precision sum, IEEE sqrt and IEEE divide
    // FAST = 1 (relaxed): float32 sum + v_rsq_f32;  0: the compiler's IEEE sqrt + divide (denormal-safe expansions);
    // 3: the reference's roundings reproduced with fused-multiply-add correction steps (Markstein): the double-precision
    //    sum as is, sqrt = v_rsq + two coupled Newton steps + one exact-residual correction, quotient = v_rcp + one Newton
    //    step + one exact-residual correction -- correctly rounded for the operands that occur here (x >= 1e-8, normal);
    // 2: 3 without the final residual corrections (<= 1 ulp, not always correctly rounded)
    auto nrm = [](float nom, float d1, float d2, float d3) {
        if (FAST == 1) return nom * __builtin_amdgcn_rsqf(((d1 + d2) + d3) + 1.0e-8f);
        if (FAST == 0) return rof_norm(nom, d1, d2, d3);
        const float x = (float)((double)((d1 + d2) + d3) + 1.0e-8);
        const float r = __builtin_amdgcn_rsqf(x);
        float q = x * r, h = 0.5f * r;
        const float e = fmaf(-h, q, 0.5f);
        q = fmaf(q, e, q);
        h = fmaf(h, e, h);
        if (FAST == 3) q = fmaf(fmaf(-q, q, x), h, q);   // correctly rounded sqrt(x)
        float y = __builtin_amdgcn_rcpf(q);
        y = fmaf(fmaf(-q, y, 1.0f), y, y);
        float z = nom * y;
        if (FAST == 3) z = fmaf(fmaf(-q, z, nom), y, z);  // correctly rounded nom / q
        return z;
    };
    // FAST = 3, three components: the same operations as three nrm() calls, written phase by phase across the components
    // (a dependent packed / transcendental operation needs a wait state after its producer: one chain at a time the
    // compiler pads every step with an s_nop -- 214 per plane; interleaved, the other components fill the gaps)
    auto nrm3 = [](const float (&nom)[3], const float (&s)[3], float (&out)[3]) {
        float x[3], q[3], h[3], e[3], y[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] = (float)((double)s[c] + 1.0e-8);
#pragma unroll
        for (int c = 0; c < 3; ++c) e[c] = __builtin_amdgcn_rsqf(x[c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) { q[c] = x[c] * e[c]; h[c] = 0.5f * e[c]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) e[c] = fmaf(-h[c], q[c], 0.5f);
#pragma unroll
        for (int c = 0; c < 3; ++c) { q[c] = fmaf(q[c], e[c], q[c]); h[c] = fmaf(h[c], e[c], h[c]); }
#pragma unroll
        for (int c = 0; c < 3; ++c) e[c] = fmaf(-q[c], q[c], x[c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) q[c] = fmaf(e[c], h[c], q[c]);   // correctly rounded sqrt(x)
#pragma unroll
        for (int c = 0; c < 3; ++c) y[c] = __builtin_amdgcn_rcpf(q[c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) e[c] = fmaf(-q[c], y[c], 1.0f);
#pragma unroll
        for (int c = 0; c < 3; ++c) y[c] = fmaf(e[c], y[c], y[c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) h[c] = nom[c] * y[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) e[c] = fmaf(-q[c], h[c], nom[c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) out[c] = fmaf(e[c], y[c], h[c]);  // correctly rounded nom / q
    };
    if (ND == 3) {
        const float nz1 = u_k1 - u, nz0 = u - u_k2;
        const float dzm = rof_mm(nz0, nz1);
        if (FAST == 3) {
            const float nom[3] = {nx1, ny1, nz1};
            const float sm[3] = {(nx1 * nx1 + dym) + dzm, (dxm + ny1 * ny1) + dzm, (dxm + dym) + nz1 * nz1};
            float o[3];
            nrm3(nom, sm, o);
            d.d1 = o[0]; d.d2 = o[1]; d.d3 = o[2];
        } else {
            d.d1 = nrm(nx1, nx1 * nx1, dym, dzm);
            d.d2 = nrm(ny1, dxm, ny1 * ny1, dzm);
            d.d3 = nrm(nz1, dxm, dym, nz1 * nz1);
        }
    } else {
        d.d1 = nrm(nx1, nx1 * nx1, dym, 0.0f);
        d.d2 = nrm(ny1, dxm, ny1 * ny1, 0.0f);
        d.d3 = 0.0f;
    }
    if (HALF) {
        d.d1 = DualIO<__half>::rt(d.d1); d.d2 = DualIO<__half>::rt(d.d2); d.d3 = DualIO<__half>::rt(d.d3);
    }
    return d;
}

template <int ND, bool HALF, int FAST, int RY, int WX, int WY>
__global__ __launch_bounds__(64 * WX * WY) void rof_zmarch_kernel(RofArgs a, int gx, int gy, int tiles_per_xcd, int zchunk)
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
    const int zc0 = a.out_begin + chunk * zchunk;
    const int zc1 = min(zc0 + zchunk, a.out_end);
    if (zc0 >= zc1) return;

    const size_t sz = (size_t)dx * dy;
    const bool x_first = (x == 0), x_last = (x == dx - 1);
    const bool emit_lane = (lane >= 2) && (lane <= 61) && (x < dx);
    const int xc = min(max(x, 0), dx - 1);
    // addressing as in pd_zmarch_xk: one lane offset (the clamped column), wave-uniform row offsets in the scalar operand
    const unsigned xo = (unsigned)xc * 4u;
    const int wy0 = __builtin_amdgcn_readfirstlane(y0);
    const int pitch = dx * 4;
    const PlaneIO io{(int)(sz * 4)};  // plane-relative buffer addressing, see tv_kernels.hip
    // row slot index q = r + 2 (rows -2..RY); EDGE: clamped into the slice
    auto rowoff = [&](int q, auto ec) __attribute__((always_inline)) {
        if constexpr (!decltype(ec)::value) return (wy0 + q - 2) * pitch;
        else return min(max(wy0 + q - 2, 0), dy - 1) * pitch;
    };
    // interior waves on interior planes take a form without the reflection selects (same operands, see pd_zmarch_xk)
    const int x_w0 = xs * 60 - 2;
    const bool xy_inner = __builtin_amdgcn_readfirstlane((int)(x_w0 >= 1 && x_w0 + 63 <= dx - 2 && y0 - 2 >= 1 && y0 + RY <= dy - 2)) != 0;

    float Up[RY + 3], Uc[RY + 3], Un[RY + 3], carry3[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) carry3[r] = 0.0f;

    // D of one plane for rows -1..RY-1 (index r+1) from the three U planes
    auto eval_plane = [&](const float (&lo)[RY + 3], const float (&mid)[RY + 3], const float (&hi)[RY + 3], bool k_first,
                          bool k_last, RofD (&D)[RY + 1], auto ec) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(ec)::value;
#pragma unroll
        for (int r = -1; r < RY; ++r) {
            const int y = y0 + r;
            const float u = mid[r + 2];
            const float ul = __shfl_up(u, 1, 64), ur = __shfl_down(u, 1, 64);
            const float u_i1 = (EDGE && x_last) ? ul : ur;
            const float u_i2 = (EDGE && x_first) ? ur : ul;
            const float u_j1 = (EDGE && y == dy - 1) ? mid[r + 1] : mid[r + 3];
            const float u_j2 = (EDGE && y == 0) ? mid[r + 3] : mid[r + 1];
            const float u_k1 = (EDGE && k_last) ? lo[r + 2] : hi[r + 2];
            const float u_k2 = (EDGE && k_first) ? hi[r + 2] : lo[r + 2];
            D[r + 1] = rof_eval<ND, HALF, FAST>(u, u_i1, u_i2, u_j1, u_j2, u_k1, u_k2);
        }
    };

    const int zstart = (ND == 3 && zc0 > 0) ? zc0 - 1 : zc0;  // warm-up plane builds the carried D3
    {
        const float *uc = a.u_in + sz * zstart;
        const float *up = a.u_in + sz * max(zstart - 1, 0);
#pragma unroll
        for (int r = 0; r < RY + 3; ++r) {
            Uc[r] = io.ldf(uc, xo, rowoff(r, std::true_type{}));
            Up[r] = (ND == 3) ? io.ldf(up, xo, rowoff(r, std::true_type{})) : 0.0f;
        }
    }

    auto step = [&](const int t, auto ec) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(ec)::value;
        const bool k_first = EDGE && (t == 0) && a.first_is_edge;
        const bool k_last = EDGE && (t == dz - 1) && a.last_is_edge;
        float In[RY];
        if (ND == 3) {
            const float *un = a.u_in + sz * (EDGE ? min(t + 1, dz - 1) : t + 1);
#pragma unroll
            for (int r = 0; r < RY + 3; ++r) Un[r] = io.ldf(un, xo, rowoff(r, ec));
        }
        {
            const float *ip = a.in + sz * t;
#pragma unroll
            for (int r = 0; r < RY; ++r) In[r] = io.ldf(ip, xo, rowoff(r + 2, ec));
        }
        RofD D[RY + 1];
        eval_plane(Up, Uc, Un, k_first, k_last, D, ec);
        float d3_ahead[RY];
#pragma unroll
        for (int r = 0; r < RY; ++r) d3_ahead[r] = 0.0f;
        if (ND == 3 && k_first) {
            // global first plane: the backward z difference reflects to D3 of plane 1, which needs U of plane 2
            float U2[RY + 3];
            const float *u2 = a.u_in + sz * min(2, dz - 1);
#pragma unroll
            for (int r = 0; r < RY + 3; ++r) U2[r] = io.ldf(u2, xo, rowoff(r, ec));
            RofD D1p[RY + 1];
            eval_plane(Uc, Un, U2, false, (1 == dz - 1) && a.last_is_edge, D1p, ec);
#pragma unroll
            for (int r = 0; r < RY; ++r) d3_ahead[r] = D1p[r + 1].d3;
        }
        const bool emit_plane = !EDGE || (t >= zc0);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const int y = y0 + r;
            // D1 of the reflected backward row (y-1, or y+1 at y == 0); D2 of the reflected backward lane
            const float d1b = (EDGE && y == 0) ? D[r + 2 <= RY ? r + 2 : RY].d1 : D[r].d1;
            const float d2l = __shfl_up(D[r + 1].d2, 1, 64), d2r = __shfl_down(D[r + 1].d2, 1, 64);
            const float d2b = (EDGE && x_first) ? d2r : d2l;
            float dv = (D[r + 1].d1 - d1b) + (D[r + 1].d2 - d2b);
            if (ND == 3) {
                const float d3b = k_first ? d3_ahead[r] : carry3[r];
                dv = dv + (D[r + 1].d3 - d3b);
                carry3[r] = D[r + 1].d3;
            }
            const float u = Uc[r + 2];
            const float tt = fmaf(a.lambda, dv, -(u - In[r]));
            const float uo = fmaf(a.tau, tt, u);
            if (emit_plane && emit_lane && (!EDGE || y < dy)) io.stf(a.u_out + sz * t, xo, rowoff(r + 2, ec), uo);
        }
        if (ND == 3) {
#pragma unroll
            for (int r = 0; r < RY + 3; ++r) { Up[r] = Uc[r]; Uc[r] = Un[r]; }
        }
    };

    // [general | short | general] ranges in separate loops (see pd_zmarch_xk): the short form needs an emitting plane that
    // is neither the first nor the last of the volume
    const int tS0 = (ND == 3 && xy_inner) ? min(max(zc0, 1), zc1) : zc1;
    const int tS1 = (ND == 3 && xy_inner) ? max(min(zc1 - 1, dz - 2), tS0 - 1) : zc1 - 1;
    for (int phase = 0; phase < 2; ++phase) {
        const int e0 = phase == 0 ? zstart : tS1 + 1, e1 = phase == 0 ? tS0 - 1 : zc1 - 1;
        for (int t = e0; t <= e1; ++t) {
            __syncthreads();  // lockstep
            step(t, std::true_type{});
        }
        if (phase == 0) {
            for (int t = tS0; t <= tS1; ++t) {
                __syncthreads();
                step(t, std::false_type{});
            }
        }
    }
}

template <int ND, bool HALF, int FAST, int RY, int WX, int WY>
static int rof_zmarch_launch(const RofArgs &a, hipStream_t st)
{
    const int nout = a.out_end - a.out_begin;
    const int gx = ceil_div(ceil_div(a.dx, 60), WX), gy = ceil_div(a.dy, WY * RY);
    const int tiles_per_xcd = ceil_div(gx * gy, 8);
    int chunks = 1;
    if (ND == 3) {
        const long waves_xy = (long)gx * gy * WX * WY;
        chunks = (int)((256L * 4 * 32 + waves_xy - 1) / waves_xy);
        const int max_chunks = ceil_div(nout, 32);
        if (chunks > max_chunks) chunks = max_chunks;
        if (chunks < 1) chunks = 1;
    }
    const int zchunk = ceil_div(nout, chunks);
    chunks = ceil_div(nout, zchunk);
    const long blocks = 8L * tiles_per_xcd * chunks;
    if (blocks > 0x7fffffffL) return tomo_fail(TOMO_E_INVALID, "volume too large for one ROF_TV launch");
    rof_zmarch_kernel<ND, HALF, FAST, RY, WX, WY><<<(unsigned)blocks, 64 * WX * WY, 0, st>>>(a, gx, gy, tiles_per_xcd, zchunk);
    return TOMO_OK;
}
