// yk_xheads.h — f16x2 mode: the detection HEADS as ONE launch of per-image workgroup clusters (included by yk_exact.hip).
//
// yolonet.py:23-60 ends in   y1 = Conv1x1(Conv3x3(x2)),   y2 = Conv1x1(Conv3x3(Concat(Up(Conv1x1(x2)), x1)))   on 7x10 / 14x20 pixel
// images: five convs of a few hundred MFLOP per image whose K is long (6912 / 4608) and whose M x N is tiny.  As plain launches they were
// 7 kernels (two of them split-K pairs with fp32 slabs) of ~12 us fixed cost each: 144 of the step's 553 us, MFMA busy 0.15
// (profiles/r03_x2_kernel_trace_per_launch.csv).  Here an image belongs to a cluster of CW = 8 workgroups (the cluster machinery of
// yk_xpersist.h: block b runs on XCD b % 8, the members of an image are blocks 8 apart, granule barrier, XCC_ID exchange) and a conv
// is split over the members along K:
//
//   * member j takes the 32-channel chunks j, j+8, ... of the K axis (all taps of a chunk) and computes the WHOLE M x N tile grid of
//     the image for them - every operand byte is used by all 16-pixel blocks x all 16-channel blocks of the workgroup;
//   * its chunk images (all pixels x 32 channels, hi and lo planes, one pixel of zero border = Keras 'same' padding, UpSampling2D by
//     address) are copied to LDS once; the nine taps of a 3x3 conv are nine shifted fragment reads of the same image (the launch form
//     fetched every tap again from L2: 9x the bytes through the LDS-DMA path);
//   * a wave owns NRW pixel blocks x NCW channel blocks (and, WK = 2, every other k-step); the WEIGHT fragments of its channel blocks are
//     private to it and go straight from L2 to registers two steps ahead (no staging, no barrier inside the K loop at all);
//   * the partial sums (fp32, true scale) go to a per-cluster buffer in L2, ONE cluster barrier, then member j adds the partials of the
//     pixel blocks j, j+8, j+16 in slot order (deterministic), applies BatchNorm + LeakyReLU and
//       - writes the stored layout [pixel][c/8][hi|lo] (the 1x1 conv in front of UpSampling2D), or
//       - runs the network-output 1x1 conv on its own pixels right away (K = 192 / 128: operand split with a LOCAL exponent - a
//         pixel's logits depend on that pixel only, so nothing has to be agreed on with the other members) and writes fp32 logits.
//
// Exponents: the K loop accumulates in the units of the chunk's source (2^-e_src per image); where a member's chunks cross from source 0 to
// source 1 of a Concatenate the accumulators are multiplied by 2^(e0 - e1) (exact), partial sums are written multiplied by 2^e.  A
// stored output's exponent is the exponent of the usual a-priori bound (gain * amax(src) + off).
#pragma once

constexpr int XH_NT = 512, XH_CW = 8, XH_MAXPH = 6, XH_NCH = 3;

struct xh_src {
    const uint8_t *p;                  // stored tensor [B][H][W][G][hi x8 | lo x8]
    const int *eexp;
    const uint32_t *amax;
    uint32_t bytes;
    int G, H, W, up, nchunk;           // up: read through UpSampling2D(2); nchunk = ceil(G / 4)
};

struct xh_phase {
    xh_src s0, s1;                     // s1.p null without Concatenate
    int H, W, W2, PP, taps;            // output image (= input image: stride 1, 'same'), padded width, padded pixels
    yk_fastdiv fd_w, fd_w2;
    int nrb, ncb, nchunk;              // 16-pixel blocks, 16-channel blocks, 32-channel chunks of the K axis
    int variant, WR, WC, WK;
    uint32_t plane, img;               // LDS bytes of one plane / one chunk image
    const uint8_t *w;                  // [chunk * taps + tap][nslab][hi|lo][16][32] halfs (pack_w order)
    uint32_t w_bytes;
    int nslab;
    const float *scale, *bias;
    float slope, cap, gain0, gain1, off;
    uint32_t part_off;                 // this phase's region inside a cluster's partial-sum buffer
    int nslot;                         // CW partial sums per tile (the K halves of a workgroup are added in LDS)
    int pre_barrier;                   // a source was written by the phase right before this one
    uint8_t *out;                      // stored output, or null
    uint32_t out_bytes;
    int outG;
    int *eexp_out;
    uint32_t *amax_out;
    const uint8_t *tw;                 // the network-output 1x1 conv behind it, or null
    uint32_t tw_bytes;
    int t_nslab, t_ncb, t_N, t_nks;
    const float *t_scale, *t_bias;
    float t_slope, t_cap;
    float *out32;                      // [B][H*W][t_N]
};

struct xh_args {
    const xh_phase *ph;
    int n_phase, B, CW, n_cluster;
    uint8_t *part;                     // [n_cluster][part_stride] partial sums
    uint32_t part_stride, part_bytes;
    unsigned long long *gran;          // [max_batch][CW] barrier granules (xp_cluster_barrier)
    uint32_t *pxcc;                    // [max_batch][CW]
    uint32_t *err;
    uint32_t lds_misc;                 // offset of the scalars behind the largest phase's images
    long long *stamps;
    int write_through;                 // 1: never trust the placement check - every exchange goes write-through + L1-bypassing (YK_CLUSTER_WT=1; tests)
    int dbg;
};

// a buffer descriptor the compiler KNOWS to be wave-uniform (the phase table's fields reach the kernel through spilled scalars; a
// descriptor it could not prove uniform costs a readfirstlane waterfall loop around every load: 213 of them in the first build)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t xh_rsrc(const void *ptr, uint32_t bytes) {
    const unsigned long long v = (unsigned long long)ptr;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, (uint32_t)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
__device__ __forceinline__ int xh_ld_i32_sc1(const void *p) {             // a word another CU may have written during this launch
    const __amdgpu_buffer_rsrc_t r = xh_rsrc(p, 4);
    return (int)__builtin_amdgcn_raw_buffer_load_b32(r, 0, 0, 16);
}

// ---- the member's chunk images -> LDS -------------------------------------------------------------------------------------------
// image c (slot) = chunk j + c*CW: hi plane [padded pixel][4 chunks of 16 B, chunk q at position q ^ ((pixel >> 1) & 3)], lo plane behind it.
// Dead slots (no such chunk) and the border are zeros.
__device__ __forceinline__ void xh_fill(const xh_phase &P, int b, int j, int tid) {
    const int n16 = P.PP * 8;                                         // 16-byte items per image
#pragma unroll 1
    for (int c = 0; c < XH_NCH; ++c) {
        const int id = j + c * XH_CW;
        if (c * XH_CW >= P.nchunk && c > 0) break;                    // no member has a chunk in this slot
        const bool live = id < P.nchunk;
        const bool second = id >= P.s0.nchunk;
        const xh_src &S = second ? P.s1 : P.s0;
        const int g0 = (second ? id - P.s0.nchunk : id) * 4;
        const __amdgpu_buffer_rsrc_t rs = xh_rsrc(S.p, S.bytes);
        const uint32_t ibase = (uint32_t)b * (uint32_t)(S.H * S.W * S.G * 32);
        constexpr int MAXIT = 6;
        u32x4 v[MAXIT];
#pragma unroll
        for (int k = 0; k < MAXIT; ++k) {
            const int item = tid + k * XH_NT;
            const int pix = item >> 3, sub = item & 7, q = sub >> 1, hl = sub & 1;
            const int py = (int)x_div((uint32_t)pix, P.fd_w2), px = pix - py * P.W2;
            const bool in = live && item < n16 && py >= 1 && py <= P.H && px >= 1 && px <= P.W && (g0 + q) < S.G;
            int y = py - 1, x = px - 1;
            if (S.up) {
                y >>= 1;
                x >>= 1;
            }
            const uint32_t off = in ? ibase + (uint32_t)(((y * S.W + x) * S.G + g0 + q) * 32 + hl * 16) : X_OOB;
            v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));   // sc1: maybe written during this launch
        }
#pragma unroll
        for (int k = 0; k < MAXIT; ++k) {
            const int item = tid + k * XH_NT;
            const int pix = item >> 3, sub = item & 7, q = sub >> 1, hl = sub & 1;
            if (item < n16)
                *reinterpret_cast<u32x4 *>(xsm + (uint32_t)c * P.img + (uint32_t)hl * P.plane + (uint32_t)pix * 64u + (uint32_t)((q ^ ((pix >> 1) & 3)) << 4)) = v[k];
        }
    }
}

template <int I, int N, typename F>
__device__ __forceinline__ void xh_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        xh_static_for<I + 1, N>(f);
    }
}

// ---- K loop: acc[NRW][NCW] += W[chunks of this member] x image ---------------------------------------------------------------------
// NI steps per wave (fully unrolled: straight-line code keeps hipcc's wait counts exact around the rotating register sets).  Step i of
// the wave group kh is k-step s = kh + i*WK of the member's list (chunk slot s / TAPS, tap s % TAPS); steps past the list multiply
// zeros (out-of-range weight loads) with the pixels of slot 0.
template <int NRW, int NCW, int NI, int TAPS, bool TWO>
__device__ __forceinline__ void xh_kloop(const xh_phase &P, int j, int wc, int kh, int nmine, float resc, const int (&q0)[NRW], floatx4 (&acc)[NRW][NCW], int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    const uint32_t foff = (uint32_t)(fr * 64 + ((fq ^ ((fr >> 1) & 3)) * 16));
    const __amdgpu_buffer_rsrc_t rsw = xh_rsrc(P.w, P.w_bytes);
    const int nsteps = nmine * TAPS, WK = P.WK;
    half8 Bh[3][NCW], Bl[3][NCW];
    auto loadB = [&](half8 (&h)[NCW], half8 (&l)[NCW], int i) {
        const int s = kh + i * WK;
        const int c = TAPS == 1 ? s : s / TAPS, t = s - c * TAPS;
        const uint32_t wstep = (uint32_t)((j + c * XH_CW) * TAPS + t);
#pragma unroll
        for (int jj = 0; jj < NCW; ++jj) {
            const int cbg = wc * NCW + jj;
            const uint32_t off = (s < nsteps && cbg < P.ncb) ? (wstep * (uint32_t)P.nslab + (uint32_t)cbg) * 2048u + foff : X_OOB;
            h[jj] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsw, off, 0, 0));
            l[jj] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsw, off + 1024u, 0, 0));
        }
    };
    auto mma = [&](const half8 &w, const half8 &x, floatx4 &c) { c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, c, 0, 0, 0); };
    // A step's pixel blocks are worked through in groups of RG; group-iteration gi = (step gi / NGS, group gi % NGS).  The fragment reads
    // of gi+1 are issued in front of the MFMAs of gi (two register sets), the order is pinned: left to itself the scheduler hoisted
    // reads of later steps until it spilled registers inside the loop.
    constexpr int RG = NRW % 3 == 0 ? 3 : NRW, NGS = NRW / RG, NG = NI * NGS;
    half8 Ah[2][RG], Al[2][RG];
    auto readA = [&](half8 (&h)[RG], half8 (&l)[RG], int gi) {
        const int i = gi / NGS, g = (gi - i * NGS) * RG;
        const int s = kh + i * WK;
        const bool live = s < nsteps;
        const int c = TAPS == 1 ? s : s / TAPS, t = s - c * TAPS;
        const int toff = TAPS == 9 ? ((t * 11) >> 5) * P.W2 + (t - ((t * 11) >> 5) * 3) : P.W2 + 1;        // t / 3 for t < 9
        const uint32_t img = live ? (uint32_t)c * P.img : 0u;
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const int pa = q0[g + r] + toff;
            const unsigned char *at = xsm + img + (uint32_t)pa * 64u + (uint32_t)((fq ^ ((pa >> 1) & 3)) << 4);
            h[r] = *reinterpret_cast<const half8 *>(at);
            l[r] = *reinterpret_cast<const half8 *>(at + P.plane);
        }
    };
    // three products per tile as three sweeps over the accumulators: consecutive MFMAs never share one
    auto mmaG = [&](const half8 (&xh)[RG], const half8 (&xl)[RG], const half8 (&wh)[NCW], const half8 (&wl)[NCW], int g) {
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int jj = 0; jj < NCW; ++jj) mma(wl[jj], xh[r], acc[g + r][jj]);
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int jj = 0; jj < NCW; ++jj) mma(wh[jj], xl[r], acc[g + r][jj]);
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int jj = 0; jj < NCW; ++jj) mma(wh[jj], xh[r], acc[g + r][jj]);
    };
    // one group per step (NRW = 5): a second fragment set would push the wave past its registers (it spilled inside the loop); the other
    // wave of the SIMD covers the read latency there
    constexpr bool DB = NGS > 1;
    loadB(Bh[0], Bl[0], 0);
    loadB(Bh[1], Bl[1], 1);
    if constexpr (DB) readA(Ah[0], Al[0], 0);
    xh_static_for<0, NG>([&](auto ic) {                              // (a `#pragma unroll` loop of this size was left rolled: register sets indexed at run time)
        constexpr int gi = decltype(ic)::value;
        constexpr int i = gi / NGS, g = (gi - i * NGS) * RG;
        if (g == 0) {
            if constexpr (TWO) {
                if (i == TAPS && resc != 1.f) {                       // (WK == 1) the member's second chunk belongs to the other source
#pragma unroll
                    for (int r = 0; r < NRW; ++r)
#pragma unroll
                        for (int jj = 0; jj < NCW; ++jj) acc[r][jj] *= resc;
                }
            }
            loadB(Bh[(i + 2) % 3], Bl[(i + 2) % 3], i + 2);
        }
        if constexpr (DB) {
            if (gi + 1 < NG) readA(Ah[(gi + 1) & 1], Al[(gi + 1) & 1], gi + 1);
            __builtin_amdgcn_sched_barrier(0);
            mmaG(Ah[gi & 1], Al[gi & 1], Bh[i % 3], Bl[i % 3], g);
        } else {
            readA(Ah[0], Al[0], gi);
            __builtin_amdgcn_sched_barrier(0);
            mmaG(Ah[0], Al[0], Bh[i % 3], Bl[i % 3], g);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}

// ---- one conv phase of one image -----------------------------------------------------------------------------------------------------
template <int NRW, int NCW, int NI, int TAPS, bool TWO>
__device__ __forceinline__ void xh_conv(const xh_args &a, const xh_phase &P, int b, int cl, int j, bool same_xcd, uint32_t *s_max, uint32_t &arrivals, int pi) {
    // the thread id is laundered per phase: everything derived from it is recomputed here instead of being hoisted out of the image /
    // phase loops and kept alive across all three instantiations (the first build spilled ~60 such values to scratch at kernel entry)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, nl4 = (lane >> 4) * 4;
    const int npx = P.H * P.W;
    // wave -> (K parity, row group, column group)
    const int kh = wid / (P.WR * P.WC), wrc = wid - kh * (P.WR * P.WC), wr = wrc / P.WC, wc = wrc - wr * P.WC;
    const bool idle = kh >= P.WK;                                     // (a variant with fewer than 8 working waves)
    // exponents of the sources (sc1: a source may have been produced a moment ago by the other members): requested here, used behind the
    // image copy (two dependent L2 round trips in front of it otherwise)
    const int e0 = xh_ld_i32_sc1(P.s0.eexp + b), e1 = P.s1.p ? xh_ld_i32_sc1(P.s1.eexp + b) : 0;
    int nmine = 0;
#pragma unroll
    for (int c = 0; c < XH_NCH; ++c) nmine += (j + c * XH_CW) < P.nchunk ? 1 : 0;
    XP_ISTAMP(a, 8 * pi)
    xh_fill(P, b, j, tid);
    int q0[NRW];
#pragma unroll
    for (int i = 0; i < NRW; ++i) {
        const int m = min((min(wr * NRW + i, P.nrb - 1)) * 16 + fr, npx - 1);
        const int oy = (int)x_div((uint32_t)m, P.fd_w), ox = m - oy * P.W;
        q0[i] = oy * P.W2 + ox;
    }
    floatx4 acc[NRW][NCW];
#pragma unroll
    for (int i = 0; i < NRW; ++i)
#pragma unroll
        for (int jj = 0; jj < NCW; ++jj) acc[i][jj] = floatx4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    XP_ISTAMP(a, 8 * pi + 1)
    const bool sec0 = j >= P.s0.nchunk, sec1 = (j + XH_CW) >= P.s0.nchunk;         // source of chunk slot 0 / 1
    const int e1v = P.s1.p ? e1 : e0;
    const int ea = sec0 ? e1v : e0, eb = (TWO && nmine > 1) ? (sec1 ? e1v : e0) : ea;
    const float resc = x_pow2(ea - eb), up = x_pow2(TWO ? eb : ea);
    if (!idle && !X_DBG(a, 1)) xh_kloop<NRW, NCW, NI, TAPS, TWO>(P, j, wc, kh, nmine, resc, q0, acc, lane);
    XP_ISTAMP(a, 8 * pi + 2)
    // ---- partial sums -> the cluster's buffer
    const __amdgpu_buffer_rsrc_t rp = xh_rsrc(a.part, a.part_bytes);
    const uint32_t pbase = (uint32_t)cl * a.part_stride + P.part_off;
    bool writer = !idle;
    if (P.WK == 2) {                                                  // the two K halves of the workgroup meet in LDS: one partial sum per member leaves the CU
        __syncthreads();                                              // (every wave is done with the images)
#pragma unroll
        for (int i = 0; i < NRW; ++i)
#pragma unroll
            for (int jj = 0; jj < NCW; ++jj) {
                const int rb = wr * NRW + i, cbg = wc * NCW + jj;
                if (kh == 1 && rb < P.nrb && cbg < P.ncb) *reinterpret_cast<floatx4 *>(xsm + (uint32_t)((rb * P.ncb + cbg) * 1024 + lane * 16)) = acc[i][jj];
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NRW; ++i)
#pragma unroll
            for (int jj = 0; jj < NCW; ++jj) {
                const int rb = wr * NRW + i, cbg = wc * NCW + jj;
                if (kh == 0 && rb < P.nrb && cbg < P.ncb) acc[i][jj] += *reinterpret_cast<const floatx4 *>(xsm + (uint32_t)((rb * P.ncb + cbg) * 1024 + lane * 16));
            }
        writer = kh == 0;
    }
    if (writer) {
        const uint32_t slot = (uint32_t)j;
#pragma unroll
        for (int i = 0; i < NRW; ++i) {
            const int rb = wr * NRW + i;
#pragma unroll
            for (int jj = 0; jj < NCW; ++jj) {
                const int cbg = wc * NCW + jj;
                if (rb >= P.nrb || cbg >= P.ncb) continue;
                const uint32_t off = pbase + ((slot * (uint32_t)P.nrb + (uint32_t)rb) * (uint32_t)P.ncb + (uint32_t)cbg) * 1024u + (uint32_t)lane * 16u;
                const floatx4 v = acc[i][jj] * up;
                if (same_xcd) xp_store16_plain(rp, off, __builtin_bit_cast(u32x4, v));
                else xp_store16_sc1(rp, off, __builtin_bit_cast(u32x4, v));
            }
        }
    }
    // pixel blocks j, j+8, j+16 belong to this member after the barrier; its waves share their (pixel block, channel block) tiles
    int nown = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) nown += (j + XH_CW * i) < P.nrb ? 1 : 0;
    // the tail conv's weights do not depend on anybody: the wave's first tile of them is requested before the barrier
    constexpr int TKS = 6;
    const int fq = lane >> 4;
    const uint32_t foff = (uint32_t)(fr * 64 + ((fq ^ ((fr >> 1) & 3)) * 16));
    const __amdgpu_buffer_rsrc_t rt = xh_rsrc(P.tw ? P.tw : a.part, P.tw ? P.tw_bytes : 16u);
    half8 twh[TKS], twl[TKS];
    auto load_tw = [&](int cbt) {
#pragma unroll
        for (int ks = 0; ks < TKS; ++ks) {
            const uint32_t wo = ks < P.t_nks ? (uint32_t)(ks * P.t_nslab + cbt) * 2048u + foff : X_OOB;
            twh[ks] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rt, wo, 0, 0));
            twl[ks] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rt, wo + 1024u, 0, 0));
        }
    };
    if (P.tw && wid < nown * P.t_ncb) load_tw(wid % P.t_ncb);
    XP_ISTAMP(a, 8 * pi + 3)
    ++arrivals;
    xp_cluster_barrier(a.gran + (size_t)b * 2 * a.CW, a.CW, j, arrivals, 0.f, s_max, a.err);
    XP_ISTAMP(a, 8 * pi + 4)
    // ---- reduce: every partial sum of a tile in flight at once, added in slot order (deterministic); the next tile's are requested
    // before this one's are added
    const int pitch = P.ncb * 16 + 4;                                 // floats per pixel row of the tail conv's LDS tile
    float *T = reinterpret_cast<float *>(xsm);
    int eo = 0;
    if (!P.tw) {                                                      // a stored output: the exponent of its a-priori bound
        float bound = P.gain0 * x_amax_wave(P.s0.amax, b) + P.off;
        if (P.s1.p) bound += P.gain1 * x_amax_wave(P.s1.amax, b);
        eo = x_exp_of(__float_as_uint(fminf(bound, P.cap)));
    }
    const float down = x_pow2(-eo);
    const __amdgpu_buffer_rsrc_t ro = xh_rsrc(P.out ? P.out : a.part, P.out ? P.out_bytes : 16u);
    if (tid == 0) s_max[0] = 0u;
    float mx = 0.f;
    const int ntile = nown * P.ncb;
    const uint32_t sstr = (uint32_t)(P.nrb * P.ncb) * 1024u;
    constexpr int NSL = XH_CW;
    floatx4 t[2][NSL];
    auto req = [&](floatx4 (&r)[NSL], int tile) {
        const int i = tile / P.ncb, cb = tile - i * P.ncb, rb = j + XH_CW * i;
        const uint32_t o0 = pbase + ((uint32_t)rb * (uint32_t)P.ncb + (uint32_t)cb) * 1024u + (uint32_t)lane * 16u;
#pragma unroll
        for (int k = 0; k < NSL; ++k)
            r[k] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rp, (tile < ntile && k < P.nslot) ? o0 + (uint32_t)k * sstr : X_OOB, 0, 16));
    };
    auto finish = [&](const floatx4 (&r)[NSL], int tile) {
        const int i = tile / P.ncb, cb = tile - i * P.ncb, rb = j + XH_CW * i;
        floatx4 sum = r[0];
#pragma unroll
        for (int k = 1; k < NSL; ++k) sum += r[k];                    // (slots past nslot are zeros)
        const int n = cb * 16 + nl4, m = rb * 16 + fr;
        const float4 sc = *reinterpret_cast<const float4 *>(P.scale + n), bs = *reinterpret_cast<const float4 *>(P.bias + n);
        float v[4];
        v[0] = x_actf(__builtin_fmaf(sum[0], sc.x, bs.x), P.slope, P.cap);
        v[1] = x_actf(__builtin_fmaf(sum[1], sc.y, bs.y), P.slope, P.cap);
        v[2] = x_actf(__builtin_fmaf(sum[2], sc.z, bs.z), P.slope, P.cap);
        v[3] = x_actf(__builtin_fmaf(sum[3], sc.w, bs.w), P.slope, P.cap);
        if (m < npx)
#pragma unroll
            for (int k = 0; k < 4; ++k) mx = fmaxf(mx, fabsf(v[k]));
        if (P.tw) {
            *reinterpret_cast<float4 *>(T + (i * 16 + fr) * pitch + n) = float4{v[0], v[1], v[2], v[3]};
        } else if (m < npx && (n >> 3) < P.outG) {
            float vd[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) vd[k] = v[k] * down;
            half4 hi, lo;
            x_split4(vd, hi, lo);
            const uint32_t off = (uint32_t)(((b * npx + m) * P.outG + (n >> 3)) * 32 + (n & 7) * 2);
            if (same_xcd) {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), ro, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), ro, off + 16u, 0, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), ro, off, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), ro, off + 16u, 0, 16);
            }
        }
    };
    if (wid < ntile) req(t[0], wid);
#pragma unroll 1
    for (int tile = wid; tile < ntile; tile += 16) {                  // (<= 36 tiles: at most three rounds of two)
        req(t[1], tile + 8);
        finish(t[0], tile);
        if (tile + 8 < ntile) {
            req(t[0], tile + 16);
            finish(t[1], tile + 8);
        }
    }
    XP_ISTAMP(a, 8 * pi + 5)
    __syncthreads();
    x_amax_lds(s_max, 0, mx);
    __syncthreads();
    XP_ISTAMP(a, 8 * pi + 6)
    const uint32_t mbits = s_max[0];
    if (!P.tw) {
        if (tid == 0) {
            if (j == 0) {
                if (same_xcd) P.eexp_out[b] = eo;
                else __hip_atomic_store(P.eexp_out + b, eo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (mbits) x_amax_global(P.amax_out + (size_t)b * XS, mbits);
        }
        return;
    }
    // ---- the network-output 1x1 conv on this member's pixels: operand split with the local exponent
    const int el = x_exp_of(mbits);
    const float dl = x_pow2(-el), ul = x_pow2(el);
#pragma unroll 1
    for (int pidx = wid; pidx < nown * P.t_ncb; pidx += 8) {
        const int i = pidx / P.t_ncb, cbt = pidx - i * P.t_ncb, rb = j + XH_CW * i;
        if (pidx != wid) load_tw(cbt);                                // (the first tile's weights were requested before the barrier)
        const float4 tsc = *reinterpret_cast<const float4 *>(P.t_scale + cbt * 16 + nl4), tbs = *reinterpret_cast<const float4 *>(P.t_bias + cbt * 16 + nl4);
        floatx4 c2 = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < TKS; ++ks) {
            if (ks >= P.t_nks) break;
            const float *src = T + (i * 16 + fr) * pitch + ks * 32 + fq * 8;
            const float4 a0 = *reinterpret_cast<const float4 *>(src), a1 = *reinterpret_cast<const float4 *>(src + 4);
            const bool kin = ks * 32 + fq * 8 < P.ncb * 16;           // (K not a multiple of 32: the tile has no such columns)
            float vd[8] = {a0.x * dl, a0.y * dl, a0.z * dl, a0.w * dl, a1.x * dl, a1.y * dl, a1.z * dl, a1.w * dl};
            if (!kin)
#pragma unroll
                for (int k = 0; k < 8; ++k) vd[k] = 0.f;
            half8 xh, xl;
            x_split8(vd, xh, xl);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(twl[ks], xh, c2, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(twh[ks], xl, c2, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(twh[ks], xh, c2, 0, 0, 0);
        }
        const int n = cbt * 16 + nl4, m = rb * 16 + fr;
        if (m < npx) {
            float *o = P.out32 + ((size_t)b * npx + m) * P.t_N + n;
            const float sck[4] = {tsc.x, tsc.y, tsc.z, tsc.w}, bsk[4] = {tbs.x, tbs.y, tbs.z, tbs.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (n + k < P.t_N) o[k] = x_actf(__builtin_fmaf(c2[k] * ul, sck[k], bsk[k]), P.t_slope, P.t_cap);
        }
    }
}

__global__ void __launch_bounds__(XH_NT) xh_kernel(const xh_args a) {
    uint32_t *s_max = reinterpret_cast<uint32_t *>(xsm + a.lds_misc);
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int cq = slot / a.CW, j = slot - cq * a.CW;
    const int c0 = cq * 8 + xcd;                                      // cluster id: its images are c0, c0 + n_cluster, ...
    const uint32_t my_xcc = (uint32_t)__builtin_amdgcn_s_getreg(6164) & 15u;       // hwreg(HW_REG_XCC_ID, 0, 4)
    for (int b = c0; b < a.B; b += a.n_cluster) {
        uint32_t arrivals = 0;
        bool same_xcd = false;
        if (threadIdx.x == 0) __hip_atomic_store(a.pxcc + (size_t)b * a.CW + j, my_xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int pi = 0; pi < a.n_phase; ++pi) {
            const xh_phase P = a.ph[pi];                              // by value: the fields live in registers, not behind a pointer
            if (P.pre_barrier) {
                ++arrivals;
                xp_cluster_barrier(a.gran + (size_t)b * 2 * a.CW, a.CW, j, arrivals, 0.f, s_max, a.err);
            }
            if (P.variant == 0) xh_conv<9, 2, 18, 9, true>(a, P, b, c0, j, same_xcd, s_max, arrivals, pi);
            else if (P.variant == 1) xh_conv<5, 3, 14, 9, false>(a, P, b, c0, j, same_xcd, s_max, arrivals, pi);
            else xh_conv<5, 2, 2, 1, false>(a, P, b, c0, j, same_xcd, s_max, arrivals, pi);
            if (pi == 0) {                                            // the first barrier is behind us: where does everybody run?
                bool same = true;
                for (int k = 0; k < a.CW; ++k) same = same && __hip_atomic_load(a.pxcc + (size_t)b * a.CW + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_xcc + 1u;
                same_xcd = same && !a.write_through && !X_DBG(a, 32);
            }
            __syncthreads();                                          // the tail conv's LDS tile before the next phase's images
        }
        XP_ISTAMP(a, 8 * a.n_phase)
    }
}
