// yk_xfin.h — f16x2 mode: a detection head in ONE launch (included by yk_exact.hip).
//
//     Conv2D 3x3 (or 1x1) + BN + LeakyReLU(0.1)  ->  Conv2D 1x1 (+bias), the network output          (models/yolonet.py:27-29, 35-38)
//
// Round 5 ran a head as THREE launches: the 3x3 conv split over K into fp32 slabs (64x64 tiles, 735 / 1120 workgroups), a finishing pass
// that added the slabs and stored the 192- / 128-channel tensor, and the 75-channel 1x1 conv that read it back: 45 / 62 MB of HBM traffic
// against 4.3 / 9.7 MB of algorithmic bytes, and - with four batches in flight - the launches that stretched most (1.4 - 1.9x,
// profiles/r06_inflight_pmc.json: the per-CU vector-memory path is the busiest unit of the step, and a 64x64 tile re-reads its operands
// from L2 more often than any other shape).  Here a workgroup owns 64 GEMM rows and ALL N = 128 / 192 channels of the 3x3 conv:
//
//   main loop   as xg_kernel (LDS-DMA ring, one s_barrier per 32-deep k-step, three MFMAs per product), over this workgroup's K slice;
//   reduction   (K split over `splitk` workgroups, z = blockIdx slice): every slice stores its accumulators write-through (sc1) and
//               takes a ticket from the tile's counter; the LAST arriver adds the slices in z order (sc1 loads: fixed order, so the result
//               does not depend on who is last) and goes on - no finishing launch, and no slab at all when splitk == 1;
//   BN + act    in registers, scaled by the per-image storage exponent the unfused path would have used (same bound, same rounding)
//               and split into (hi | lo) straight into the 1x1 conv's MFMA pixel operand in LDS (the ring's memory, now free) - the 128- /
//               192-channel tensor never exists in HBM;
//   1x1 conv    wave w multiplies the tile's 16-row block w with the 75 (80) output channels: weight fragments from global memory into
//               registers (host order = fragment order), K = N; + bias; fp32 rows of the network output.
//
// Arithmetic order per output element is the unfused path's (k-steps ascending, w_lo*x_hi, w_hi*x_lo, w_hi*x_hi; slices in z order): with
// the same slice counts (7 and 4 at 32 images: the developer build's YK_XF_SPLITK_192 / _128) the logits ARE bit-identical to the three-launch form
// (tests/test_gpu_fin.py::test_equal_slice_counts_give_bit_identical_logits); the shipped rule slices differently: 2e-6 of scale apart.
#pragma once

struct xf_args {
    xg_args c;                          // the 3x3 (or 1x1) conv: sources, geometry, weights, BN, bound; c.N = its channels (BN tile = all of them)
    // the 1x1 output conv
    const uint8_t *w2;                  // [K2 steps][nslab2][hi|lo][16][32] halfs, chunk-swizzled (pack_w order)
    int nslab2, N2;                     // 16-column blocks (5 for 75 channels), real channels
    const float *scale2, *bias2;        // scale carries 2^-s of the weight split
    float slope2, cap2;
    float *out32;                       // [M][N2] fp32
    uint32_t *ticket;                   // [M tiles] arrival counters (zero between launches: the last arriver clears its own)
    uint32_t slab_bytes;
};

// NW waves: wave (wm, wn) owns rows [wm*32, wm*32+32) x columns [wn*BN/WN, (wn+1)*BN/WN); (BM, NW) = (64, 4): 2x2 waves, (64, 8): 2x4, (128, 8): 4x2
template <int BM, int BN, int NWV>
struct xf_cfg {
    static constexpr int NW = NWV, WM = BM / 32, WN = NW / WM, NT = 64 * NW;
    static constexpr int TM = 2, TN = BN / WN / 16;
    static constexpr int RB = BM / 16;                            // 16-row blocks of the tile
    static constexpr int STAGE = (BM + BN) * 128;
    static constexpr int A2 = (BN / 32) * RB * 2048;              // the 1x1 conv's pixel operand: [k-step][row block][hi|lo][16][64 B]
    static constexpr int SMALL = 5 * BM * 4 + 64;
    static constexpr int lds(int ns) { return (ns * STAGE > A2 ? ns * STAGE : A2) + SMALL; }
};

template <int BM, int BN, int NS, int NWV>
__global__ void __launch_bounds__(64 * NWV) xf_kernel(const xf_args f) {
    typedef xf_cfg<BM, BN, NWV> C;
    constexpr int WN = C::WN, NW = C::NW, TM = C::TM, TN = C::TN, RB = C::RB, NT = C::NT;
    // A pieces: global piece p = wid*A_IT + n is (row block p >> 1, half p & 1); a wave with ONE piece fetches one half of one row block
    constexpr int A_IT = (BM / 16 * 2) / NW, B_IT = (BN / 16 * 2) / NW, L = A_IT + B_IT, AR = (A_IT + 1) / 2;
    static_assert((BM / 16 * 2) % NW == 0 && (BN / 16 * 2) % NW == 0 && (A_IT == 1 || A_IT % 2 == 0), "1 KB pieces must divide among the waves");
    static_assert(NS >= 2 && (NS - 2) * L <= 63, "vmcnt is a 6-bit counter");
    const xg_args &a = f.c;
    constexpr int BIG = C::lds(NS) - C::SMALL;
    float *s_up = reinterpret_cast<float *>(xsm + BIG), *s_resc = s_up + BM, *s_down = s_resc + BM, *s_rup = s_down + BM;
    uint32_t *s_amax = reinterpret_cast<uint32_t *>(s_rup + BM);
    uint32_t *s_flag = s_amax + BM;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    // slices of one K range run back to back on one XCD: they read the same weight slice out of that L2
    const int gx = gridDim.x;
    const int v = x_xcd_tile(blockIdx.x + gx * blockIdx.z, gx * gridDim.z);
    const int vz = v / gx, vx = v - vz * gx;
    const int m0 = vx * BM;
    const int b0 = (int)x_div((uint32_t)m0, a.fd_hw), bl = (int)x_div((uint32_t)min(a.M - 1, m0 + BM - 1), a.fd_hw);

    const int kb = a.taps * a.nc0, nk_all = kb + a.taps * a.nc1;
    const int per = (nk_all + a.splitk - 1) / a.splitk;
    const int kt0 = vz * per, nk = max(0, min(per, nk_all - kt0));
    const int lim = kt0 + nk;

    // ---- this lane's A rows (fixed over the walk): row l>>2 of the 16-row blocks wid*AR + it, fetching chunk (l&3) ^ ((row>>1)&3)
    const int lr = lane >> 2, lc = (lane & 3) ^ ((lr >> 1) & 3);
    const int G0 = a.s0.G, G1 = a.s1.G, W0 = a.s0.W;
    uint32_t P0[AR], P1[AR], rmask[AR];
    int ry[AR], rx[AR];
#pragma unroll
    for (int it = 0; it < AR; ++it) {
        const int m = m0 + ((wid * A_IT) / 2 + it) * 16 + lr;
        const bool ok = m < a.M;
        const uint32_t mm = ok ? m : 0;
        const uint32_t b = x_div(mm, a.fd_hw), rem = mm - b * a.HoWo;
        const uint32_t oy = x_div(rem, a.fd_wo), ox = rem - oy * a.Wo;
        const int ry0 = (int)oy * a.stride - a.pad_t, rx0 = (int)ox * a.stride - a.pad_l;
        ry[it] = ry0;
        rx[it] = rx0;
        if (a.up0) P0[it] = b * (uint32_t)(a.s0.H * W0 * G0 * 32) + lc * 32u;
        else P0[it] = b * (uint32_t)(a.Hi * a.Wi * G0 * 32) + (uint32_t)((ry0 * a.Wi + rx0) * G0 * 32) + lc * 32u;
        P1[it] = b * (uint32_t)(a.Hi * a.Wi * G1 * 32) + (uint32_t)((ry0 * a.Wi + rx0) * G1 * 32) + lc * 32u;
        uint32_t msk = 0;
        for (int t = 0; t < a.taps; ++t) {
            const int ky = (a.ks == 3) ? t / 3 : 0, kx = t - ky * a.ks;
            if (ok && (unsigned)(ry0 + ky) < (unsigned)a.Hi && (unsigned)(rx0 + kx) < (unsigned)a.Wi) msk |= 1u << t;
        }
        rmask[it] = msk;
    }
    const bool lastbad0 = ((a.nc0 - 1) * 4 + lc) >= G0, lastbad1 = a.nc1 > 0 && ((a.nc1 - 1) * 4 + lc) >= G1;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void *)a.s0.p, 0, a.s0.bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void *)(a.s1.p ? a.s1.p : a.s0.p), 0, a.s1.p ? a.s1.bytes : a.s0.bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);

    // walk state (uniform): segment, channel step inside the segment, tap (fastest)
    int step = kt0, seg, tap, cs;
    if (kt0 < kb) {
        seg = 0;
        cs = kt0 / a.taps;
        tap = kt0 - cs * a.taps;
    } else {
        seg = 1;
        const int r = kt0 - kb;
        cs = r / a.taps;
        tap = r - cs * a.taps;
    }
    uint32_t aoff[AR];
    auto retap = [&]() {
        const int ky = (a.ks == 3) ? (tap * 11) >> 5 : 0, kx = tap - ky * a.ks;
        const bool tlive = tap < a.taps;
#pragma unroll
        for (int it = 0; it < AR; ++it) {
            const bool ok = tlive && ((rmask[it] >> tap) & 1u);
            uint32_t o;
            if (seg) o = P1[it] + (uint32_t)((ky * a.Wi + kx) * G1 * 32);
            else if (a.up0) o = P0[it] + (uint32_t)((((ry[it] + ky) >> 1) * W0 + ((rx[it] + kx) >> 1)) * G0 * 32);
            else o = P0[it] + (uint32_t)((ky * a.Wi + kx) * G0 * 32);
            aoff[it] = ok ? o : X_OOB;
        }
    };
    retap();
    const uint32_t wbase = (uint32_t)(wid * B_IT) * 1024u + lane * 16u, wstep = (uint32_t)a.nslab * 2048u;
    auto dma = [&](int stage) {
        unsigned char *As = xsm + stage * C::STAGE, *Bs = As + BM * 128;
        const bool live = step < lim;
        const int nc = seg ? a.nc1 : a.nc0;
        const bool bad = !live || (cs == nc - 1 && (seg ? lastbad1 : lastbad0));
        const uint32_t d_cso = bad ? X_OOB : (uint32_t)cs * 128u;
        const uint32_t d_ws = live ? wbase + (uint32_t)step * wstep : X_OOB;
#pragma unroll
        for (int n = 0; n < A_IT; ++n) {
            const uint32_t o = aoff[n >> 1] + d_cso + (uint32_t)((wid * A_IT + n) & 1) * 16u;
            lds_ptr_t dsta = (lds_ptr_t)(As + (wid * A_IT + n) * 1024);
            if (seg) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dsta, 16, o, 0, 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dsta, 16, o, 0, 0, 0);
            }
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            lds_ptr_t dstb = (lds_ptr_t)(Bs + (wid * B_IT + it) * 1024);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, dstb, 16, d_ws + (uint32_t)it * 1024u, 0, 0, 0);
        }
        ++step;
        ++tap;
        if (tap >= a.taps) {                                       // uniform
            tap = 0;
            ++cs;
            if (!seg && cs >= a.nc0 && a.nc1 > 0) {
                seg = 1;
                cs = 0;
            }
        }
        if (a.taps > 1 || cs == 0) retap();
    };
    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    const int foff = fr * 64 + ((fq ^ ((fr >> 1) & 3)) * 16);
    auto compute = [&](int stage) {
        const unsigned char *As = xsm + stage * C::STAGE, *Bs = As + BM * 128;
        half8 xh[TM], xl[TM], wh[TN], wl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            xh[i] = *reinterpret_cast<const half8 *>(As + ((wm * TM + i) * 2) * 1024 + foff);
            xl[i] = *reinterpret_cast<const half8 *>(As + ((wm * TM + i) * 2 + 1) * 1024 + foff);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            wh[j] = *reinterpret_cast<const half8 *>(Bs + ((wn * TN + j) * 2) * 1024 + foff);
            wl[j] = *reinterpret_cast<const half8 *>(Bs + ((wn * TN + j) * 2 + 1) * 1024 + foff);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], xh[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xl[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xh[i], acc[i][j], 0, 0, 0);
    };
    int rowb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) rowb[i] = (int)x_div((uint32_t)min(a.M - 1, m0 + (wm * TM + i) * 16 + fr), a.fd_hw) - b0;

#pragma unroll
    for (int s = 0; s < NS - 1; ++s) dma(s);
    xg_prep<BM, NW>(a, b0, bl, s_up, s_resc, s_down, s_rup, s_amax);   // eexp_out / amax_out are null: the intermediate tensor does not exist
    bool in0 = a.nc1 > 0 && kt0 < kb;
    auto rescale = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float fs = s_resc[rowb[i]];
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] *= fs;
        }
    };
    int rd = 0, wr = NS - 1;
    for (int kt = 0; kt < nk; ++kt) {
        x_wait_vm<(NS - 2) * L>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (in0 && kt0 + kt == kb) {
            rescale();
            in0 = false;
        }
        dma(wr);
        compute(rd);
        rd = (rd + 1 == NS) ? 0 : rd + 1;
        wr = (wr + 1 == NS) ? 0 : wr + 1;
    }
    x_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (in0) rescale();

    // ---- reduction over the K slices: write-through partial sums, a ticket per tile, the last arriver adds them in z order
    if (a.splitk > 1) {
        const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void *)a.slab, 0, f.slab_bytes, 0x00020000);
        const uint32_t ntile = (uint32_t)gx, per_slice = ntile * (uint32_t)(TM * TN) * (uint32_t)NT * 16u;
        const uint32_t mine = (uint32_t)vz * per_slice + ((uint32_t)vx * (TM * TN) * (uint32_t)NT + (uint32_t)tid) * 16u;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rsl, mine + (uint32_t)(i * TN + j) * (uint32_t)(NT * 16), 0, /*sc1*/ 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's stores have reached the memory side
        __syncthreads();
        if (tid == 0) {
            const uint32_t old = __hip_atomic_fetch_add(f.ticket + vx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = old + 1u == (uint32_t)a.splitk;
            if (last) __hip_atomic_store(f.ticket + vx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            s_flag[0] = last ? 1u : 0u;
        }
        __syncthreads();
        if (!s_flag[0]) return;
        const uint32_t tile_off = ((uint32_t)vx * (TM * TN) * (uint32_t)NT + (uint32_t)tid) * 16u;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < a.splitk; ++z) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] += __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rsl, (uint32_t)z * per_slice + tile_off + (uint32_t)(i * TN + j) * (uint32_t)(NT * 16), 0, /*sc1*/ 16));
        }
    }

    // ---- BN + activation, x 2^-e of the image, split -> the 1x1 conv's pixel operand in LDS
    unsigned char *A2 = xsm;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rb = wm * TM + i;
        const int bi = rowb[i];
        const float up = s_up[bi], down = s_down[bi];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = (wn * TN + j) * 16 + (lane >> 4) * 4;
            const float4 sc = *reinterpret_cast<const float4 *>(a.scale + n), bs = *reinterpret_cast<const float4 *>(a.bias + n);
            float vv[4];
            vv[0] = x_actf(__builtin_fmaf(acc[i][j][0] * up, sc.x, bs.x), a.slope, a.cap) * down;
            vv[1] = x_actf(__builtin_fmaf(acc[i][j][1] * up, sc.y, bs.y), a.slope, a.cap) * down;
            vv[2] = x_actf(__builtin_fmaf(acc[i][j][2] * up, sc.z, bs.z), a.slope, a.cap) * down;
            vv[3] = x_actf(__builtin_fmaf(acc[i][j][3] * up, sc.w, bs.w), a.slope, a.cap) * down;
            half4 hi, lo;
            x_split4(vv, hi, lo);
            const int ks2 = n >> 5, chunk = (n >> 3) & 3;
            unsigned char *d = A2 + ((ks2 * RB + rb) * 2) * 1024 + fr * 64 + ((chunk ^ ((fr >> 1) & 3)) << 4) + (n & 7) * 2;
            *reinterpret_cast<half4 *>(d) = hi;
            *reinterpret_cast<half4 *>(d + 1024) = lo;
        }
    }
    __syncthreads();

    // ---- the 1x1 output conv: wave w owns row block w % RB and N2 blocks [g*NBW, (g+1)*NBW) with g = w / RB (one group, or two: 3 + 2 blocks); K = BN
    constexpr int NB2 = 5, K2 = BN / 32;                              // 80 columns cover the 75 = 3 * (5 + 20) of a VOC head
    constexpr int NG = NW / RB, NBW = (NB2 + NG - 1) / NG;
    const int rb2 = wid % RB, nbb = (wid / RB) * NBW;
    const uint8_t *wq = f.w2 + foff;                                 // host order = fragment order: lane (row fr, chunk fq) reads its own 16 bytes
    half8 bh[2][NBW], blo[2][NBW];
    auto loadb = [&](int ks, int buf) {
#pragma unroll
        for (int q = 0; q < NBW; ++q) {
            const int nb = min(nbb + q, NB2 - 1);
            const uint8_t *pq = wq + ((size_t)ks * f.nslab2 + nb) * 2048;
            bh[buf][q] = *reinterpret_cast<const half8 *>(pq);
            blo[buf][q] = *reinterpret_cast<const half8 *>(pq + 1024);
        }
    };
    floatx4 acc2[NBW];
#pragma unroll
    for (int q = 0; q < NBW; ++q) acc2[q] = floatx4{0.f, 0.f, 0.f, 0.f};
    loadb(0, 0);
#pragma unroll
    for (int ks = 0; ks < K2; ++ks) {
        if (ks + 1 < K2) loadb(ks + 1, (ks + 1) & 1);
        const half8 xh = *reinterpret_cast<const half8 *>(A2 + ((ks * RB + rb2) * 2) * 1024 + foff);
        const half8 xl = *reinterpret_cast<const half8 *>(A2 + ((ks * RB + rb2) * 2 + 1) * 1024 + foff);
#pragma unroll
        for (int q = 0; q < NBW; ++q) acc2[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(blo[ks & 1][q], xh, acc2[q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NBW; ++q) acc2[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[ks & 1][q], xl, acc2[q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NBW; ++q) acc2[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[ks & 1][q], xh, acc2[q], 0, 0, 0);
    }
    {
        const int r = rb2 * 16 + fr, m = m0 + r;
        const int bi = (int)x_div((uint32_t)min(a.M - 1, m), a.fd_hw) - b0;
        const float up2 = 1.0f / s_down[bi];                        // 2^e of the intermediate (a power of two: exact)
        if (m < a.M) {
            float *o = f.out32 + (size_t)m * f.N2;
#pragma unroll
            for (int q = 0; q < NBW; ++q) {
                const int nb = nbb + q;
                if (nb >= NB2) continue;
                const int n = nb * 16 + (lane >> 4) * 4;
                const float4 sc = *reinterpret_cast<const float4 *>(f.scale2 + n), bs = *reinterpret_cast<const float4 *>(f.bias2 + n);
                const float o0 = x_actf(__builtin_fmaf(acc2[q][0] * up2, sc.x, bs.x), f.slope2, f.cap2);
                const float o1 = x_actf(__builtin_fmaf(acc2[q][1] * up2, sc.y, bs.y), f.slope2, f.cap2);
                const float o2 = x_actf(__builtin_fmaf(acc2[q][2] * up2, sc.z, bs.z), f.slope2, f.cap2);
                const float o3 = x_actf(__builtin_fmaf(acc2[q][3] * up2, sc.w, bs.w), f.slope2, f.cap2);
                if (n + 0 < f.N2) o[n + 0] = o0;
                if (n + 1 < f.N2) o[n + 1] = o1;
                if (n + 2 < f.N2) o[n + 2] = o2;
                if (n + 3 < f.N2) o[n + 3] = o3;
            }
        }
    }
}
