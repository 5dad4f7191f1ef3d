// yk_xpersist.h — f16x2 mode: the LATE backbone as ONE persistent launch (included by yk_exact.hip).
//
// From 14x20 pixels per image on, a MobileNet layer is a few hundred KB per image: as separate launches every layer paid ~6 us of
// launch + prologue, ~6 us of epilogue and, because a kernel boundary empties the L2s, a ~0.8 us first-touch round trip per pipeline
// step (profiles/r04_knockout_sweep.txt: the K loop of a 384 -> 384 pointwise conv is 10 of its 23 us and does not care whether the
// MFMAs run).  Here a chain   DepthwiseConv2D+BN+act -> Conv2D 1x1+BN+act -> DepthwiseConv2D ...   (keras_mobilenet.py:359-436 blocks
// 7-13) runs inside one kernel:
//
//   * an image belongs to a CLUSTER of CW = 8 workgroups (512 threads each); member j owns the channel slice [j*C/8, (j+1)*C/8) of every
//     tensor of its image and ALL of its pixels - a depthwise conv never needs a neighbour workgroup's data (no halo);
//   * pointwise phase: the workgroup multiplies the image's whole depthwise tensor D [pixels][C] (MFMA pixel operand, streamed from L2
//     by LDS-DMA through a 3-deep ring) with its own slice of the weights; the result y [pixels][C/8] stays in LDS as fp32;
//   * depthwise phase: 3x3 taps on that LDS image (zero border = Keras padding), BN, activation, split to (hi, lo) and stored as the NEXT
//     pointwise conv's operand D, already in MFMA tile order ([k-step][16-pixel block][hi|lo][16][64 B, chunk-swizzled]: its operand
//     pieces are linear 1 KB copies);
//   * one cluster barrier per block (after the depthwise phase): every wave drains its D stores, one lane bumps the image's arrival
//     counter and polls it; the consumers read D with sc1 (L1-bypassing) LDS-DMA loads, so no acquire fence is needed
//     (cdna_hip_programming.md guideline 16: "sc1 loads may replace the acquire").  The members of a cluster tell each other which XCD
//     they run on (s_getreg XCC_ID) through the first barrier: when they share one - how the grid is laid out, block b runs on XCD
//     b % 8 and the members of an image are blocks 8 apart - D is stored with PLAIN stores and stays in that XCD's L2, the coherence
//     point of all its CUs (0.25 us away instead of 1.5 us through the memory side); otherwise, and always before the placement is
//     known, D is stored write-through (sc1): correct for any placement, fast for the expected one.  Spins are bounded: a cluster that
//     cannot assemble (members not resident) gives up, sets the plan's sticky error word and leaves garbage rather than hanging the GPU.
//
// Exponents (header of yk_exact.hip): y never leaves the CU, so it needs none.  D's storage exponent comes from a bound of a bound
// (gain_dw * min(cap, gain_pw * M + off) + off_dw, M = the MEASURED maximum of the previous D: each workgroup publishes the maximum of
// what it wrote, the cluster barrier carries it) - two levels of over-estimate, as in the fused blocks of yk_xblock.h.
// Tensors that other kernels read (the x1 / x2 taps of yolonet.py:23-25) are written in the stored layout by STORE phases.
#pragma once

enum { XP_LOAD = 1, XP_DW = 2, XP_PW = 3, XP_STORE = 4 };
constexpr int XP_NW = 8, XP_NT = 64 * XP_NW, XP_NS = 3, XP_MAXPH = 24;
constexpr unsigned XP_MISC = 8192;                     // tail of the dynamic LDS: maxima, scalars, the next depthwise conv's parameter slice

struct xp_phase {
    int type;
    int H, W, Gs;                      // y image this phase reads or writes: pixels and channel groups (of 8) per workgroup slice
    yk_fastdiv fd_w, fd_gs;
    // XP_DW: y [H][W] -> D [Ho*Wo][C]
    int Ho, Wo, stride, pad_t, pad_l;
    yk_fastdiv fd_wo;
    const float *par;                  // [11][Cp] nine taps, scale, bias
    int Cp;
    float slope, cap, gain, off;
    int nrb_out, dbuf;                 // 16-pixel blocks of the output, which D buffer
    // XP_PW: D [nrb*16][nks*32] x W[slice] -> y
    int nrb, nks, ncb, nslab, WR, WC, ppw;
    uint32_t w_off;                    // arena offset of [nks][nslab][hi|lo][16][32] halfs
    const float *scale, *bias;
    float pslope, pcap, pgain, poff;
    // XP_LOAD / XP_STORE: a tensor in the stored layout [B][H][W][G][hi x8 | lo x8]
    const uint8_t *src;
    const int *src_eexp;
    const uint32_t *src_amax;
    uint8_t *dst;
    int *dst_eexp;
    uint32_t *dst_amax;
    int tG;                            // channel groups of that tensor
    int src_f32, dst_f32;              // XP_LOAD / XP_STORE: the tensor is stored as fp32 planes (exponent 0)
    int barrier_after;
    // the depthwise phase that follows a LOAD / PW phase: its parameter slice is requested while this phase runs
    const float *nd_par;
    int nd_Cp, nd_Gs;
    int resident;                      // XP_PW: the workgroup's whole weight slice is deposited in LDS before the K loop (1), by the phase before it (2)
    // XP_DW: the resident pointwise phase behind it - its weight slice is requested before the cluster barrier and lands under it
    uint32_t nw_off;
    int nw_nks, nw_ncb, nw_nslab;
    int adirect;                       // XP_PW: every 16-pixel block belongs to one wave (WC == 1): its D fragments go straight to registers
    int zero_border;                   // XP_PW: the y image changes shape here
};

struct xp_args {
    const xp_phase *ph;
    int n_phase, B, CW, n_cluster;
    const uint8_t *arena;              // pointwise weights + the two D buffers: one buffer descriptor
    uint32_t arena_bytes, d_off[2], d_img_stride;
    unsigned long long *gran;          // [max_batch][2 barrier parities][CW] {barrier ordinal : 32 | float bits of the member's maximum : 32}, cleared by the step's first launch
    uint32_t *pxcc;                    // [max_batch][CW] XCD id of every member + 1
    uint32_t *err;                     // sticky: a cluster barrier timed out
    long long *stamps;                 // developer builds: [workgroup][4 * XP_MAXPH + 4] wall_clock64 ticks (100 MHz), or null
    int write_through;                 // 1: never trust the placement check - every exchange goes write-through + L1-bypassing (YK_CLUSTER_WT=1; tests)
    int dbg;                           // developer builds: knock-out bits (1 no MFMA, 2 no operand DMA, 4 no D stores, 8 plain D stores, 16 no dw taps)
};

#ifdef YK_DEV
#define XP_ISTAMP(a, k) \
    if ((a).stamps && threadIdx.x == 0) (a).stamps[(size_t)blockIdx.x * (4 * XP_MAXPH + 4) + (k)] = (long long)wall_clock64();
#else
#define XP_ISTAMP(a, k)
#endif
__device__ __forceinline__ void xp_store16_sc1(const __amdgpu_buffer_rsrc_t rs, uint32_t off, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, /*sc1: write-through*/ 16);
}
__device__ __forceinline__ void xp_store16_plain(const __amdgpu_buffer_rsrc_t rs, uint32_t off, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
}

// Every workgroup of the image has finished the phase and its stores are visible.  The data is the flag (guideline 16, R2): a member
// publishes ONE 8-byte granule {ordinal of this barrier, the maximum of what it wrote}; lanes 0..CW-1 of wave 0 re-read the image's CW
// granules (one 64-byte line) until every tag has REACHED the ordinal - no counter, no returning atomic, and the maxima arrive with the
// last tag.  Two granule sets alternate by the ordinal's parity (`gran` points at the image's [2][CW] block): a fast member that is
// already publishing barrier n + 1 writes the other set, so a member that was descheduled before it saw every tag of barrier n still
// finds them - and their maxima - intact (set n & 1 is next written at barrier n + 2, which nobody reaches before everyone has passed
// n + 1, hence n).  Tags are compared with >= for the same reason.  Returns the image's maximum.
__device__ __forceinline__ float xp_cluster_barrier(unsigned long long *gran, int CW, int j, uint32_t ordinal, float wg_max, uint32_t *s_max, uint32_t *err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // every storing wave drains its stores
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        unsigned long long *set = gran + (size_t)(ordinal & 1u) * CW;
        if (lane == 0)
            __hip_atomic_store(set + j, ((unsigned long long)ordinal << 32) | __float_as_uint(wg_max), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        unsigned long long x = 0;
        for (;;) {
            x = lane < CW ? __hip_atomic_load(set + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)ordinal << 32);
            if (__all((uint32_t)(x >> 32) >= ordinal)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) {                               // ~1 s: the members are not co-resident
                if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // mapped host memory: the host polls it
                break;
            }
        }
        uint32_t m = lane < CW ? (uint32_t)x : 0u;                    // non-negative floats order like their bit patterns
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
        if (lane == 0) s_max[1] = m;
    }
    __syncthreads();
    return __uint_as_float(s_max[1]);
}

// LDS image of y: fp32, a zero border of one pixel, two planes (channels 0-3 | 4-7 of every group) so that the 16-byte reads of
// consecutive (pixel, group) items are consecutive
__device__ __forceinline__ uint32_t xp_ypl(const xp_phase &P) { return (uint32_t)((P.H + 2) * (P.W + 2) * P.Gs * 16); }

__device__ __forceinline__ void xp_zero_border(const xp_phase &P) {
    const int W2 = P.W + 2, nb = 2 * W2 + 2 * P.H;
    const uint32_t ypl = xp_ypl(P);
    for (int i = threadIdx.x; i < nb * P.Gs; i += XP_NT) {
        const int k = (int)x_div((uint32_t)i, P.fd_gs), g = i - k * P.Gs;
        int q;
        if (k < W2) q = k;                                            // top row
        else if (k < 2 * W2) q = (P.H + 1) * W2 + (k - W2);           // bottom row
        else {
            const int r = (k - 2 * W2) >> 1;
            q = (r + 1) * W2 + (((k - 2 * W2) & 1) ? P.W + 1 : 0);
        }
        *reinterpret_cast<u32x4 *>(xsm + (q * P.Gs + g) * 16) = u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4 *>(xsm + ypl + (q * P.Gs + g) * 16) = u32x4{0u, 0u, 0u, 0u};
    }
}

// The parameter slice [11][Gs*8] fp32 of a depthwise phase, requested by LDS-DMA into the misc region (behind the maxima) while the
// phase before it still runs: the depthwise pass opens with LDS reads instead of 22 dependent global loads per thread.
constexpr int XP_PAR_OFF = 256;
__device__ __forceinline__ void xp_fetch_par(const float *par, int Cp, int Gs, int j) {
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int cpr = Gs * 2, n16 = 11 * cpr;                           // 16-byte chunks per row, in all
    if (wid * 64 >= n16) return;                                      // wave-uniform
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)par, 0, (uint32_t)(11 * Cp * 4), 0x00020000);
    const int q = wid * 64 + lane, t = q / cpr, c = q - t * cpr;
    const uint32_t off = q < n16 ? (uint32_t)((t * Cp + j * Gs * 8) * 4 + c * 16) : X_OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_ptr_t)(xsm + XP_NS * 8 * 6 * 1024 + XP_PAR_OFF + wid * 1024), 16, off, 0, 0, 0);
}

// ---- XP_LOAD: the workgroup's channel slice of a stored tensor -> y ---------------------------------------------------
__device__ __forceinline__ float xp_load(const xp_phase &P, int b, int j) {
    const float up = x_pow2(P.src_eexp[b]);
    const uint32_t ypl = xp_ypl(P);
    const int n = P.H * P.W * P.Gs, W2 = P.W + 2;
    for (int i = threadIdx.x; i < n; i += XP_NT) {
        const int p = (int)x_div((uint32_t)i, P.fd_gs), g = i - p * P.Gs;
        const int py = (int)x_div((uint32_t)p, P.fd_w), px = p - py * P.W;
        const uint8_t *s = P.src + (((size_t)b * P.H * P.W + p) * P.tG + j * P.Gs + g) * 32;
        const u32x4 h = *reinterpret_cast<const u32x4 *>(s), l = *reinterpret_cast<const u32x4 *>(s + 16);
        float4 v0, v1;
        if (P.src_f32) {
            v0 = float4{__uint_as_float(h[0]) * up, __uint_as_float(h[1]) * up, __uint_as_float(h[2]) * up, __uint_as_float(h[3]) * up};
            v1 = float4{__uint_as_float(l[0]) * up, __uint_as_float(l[1]) * up, __uint_as_float(l[2]) * up, __uint_as_float(l[3]) * up};
        } else {
        v0.x = x_mix_sum_lo(h[0], l[0]) * up; v0.y = x_mix_sum_hi(h[0], l[0]) * up;
        v0.z = x_mix_sum_lo(h[1], l[1]) * up; v0.w = x_mix_sum_hi(h[1], l[1]) * up;
        v1.x = x_mix_sum_lo(h[2], l[2]) * up; v1.y = x_mix_sum_hi(h[2], l[2]) * up;
        v1.z = x_mix_sum_lo(h[3], l[3]) * up; v1.w = x_mix_sum_hi(h[3], l[3]) * up;
        }
        const int q = (py + 1) * W2 + px + 1;
        *reinterpret_cast<float4 *>(xsm + (q * P.Gs + g) * 16) = v0;
        *reinterpret_cast<float4 *>(xsm + ypl + (q * P.Gs + g) * 16) = v1;
    }
    xp_zero_border(P);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (the next phase's parameter slice)
    __syncthreads();
    return x_amax_wave(P.src_amax, b);                                // the measured maximum of |y|
}

// ---- XP_STORE: y -> the workgroup's channel slice of a stored tensor ----------------------------------------------------
__device__ __forceinline__ void xp_store(const xp_phase &P, int b, int j, float bound, uint32_t *s_max) {
    const int eo = P.dst_f32 ? 0 : x_exp_of(__float_as_uint(bound));
    const float down = x_pow2(-eo);
    const uint32_t ypl = xp_ypl(P);
    const int n = P.H * P.W * P.Gs, W2 = P.W + 2;
    float mx = 0.f;
    for (int i = threadIdx.x; i < n; i += XP_NT) {
        const int p = (int)x_div((uint32_t)i, P.fd_gs), g = i - p * P.Gs;
        const int py = (int)x_div((uint32_t)p, P.fd_w), px = p - py * P.W;
        const int q = (py + 1) * W2 + px + 1;
        const float4 v0 = *reinterpret_cast<const float4 *>(xsm + (q * P.Gs + g) * 16);
        const float4 v1 = *reinterpret_cast<const float4 *>(xsm + ypl + (q * P.Gs + g) * 16);
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        float vd[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            mx = fmaxf(mx, fabsf(v[k]));
            vd[k] = v[k] * down;
        }
        uint8_t *d = P.dst + (((size_t)b * P.H * P.W + p) * P.tG + j * P.Gs + g) * 32;
        if (P.dst_f32) {
            *reinterpret_cast<float4 *>(d) = v0;
            *reinterpret_cast<float4 *>(d + 16) = v1;
            continue;
        }
        half8 hi, lo;
        x_split8(vd, hi, lo);
        *reinterpret_cast<half8 *>(d) = hi;
        *reinterpret_cast<half8 *>(d + 16) = lo;
    }
    if (threadIdx.x == 0) {
        s_max[0] = 0u;
        P.dst_eexp[b] = eo;                                           // every member writes the same value
    }
    __syncthreads();
    x_amax_lds(s_max, 0, mx);
    __syncthreads();
    if (threadIdx.x == 0 && s_max[0]) x_amax_global(P.dst_amax + (size_t)b * XS, s_max[0]);
}

__device__ __forceinline__ void xp_preload_weights(const xp_args &a, uint32_t w_off, int nks, int ncb, int nslab, int j);

// ---- XP_DW: depthwise 3x3 on y -> D (the next pointwise conv's pixel operand, MFMA tile order, write-through) ----------------
// returns the storage exponent of D; the workgroup's maximum of |z| goes to *wg_max
__device__ __forceinline__ int xp_dw(const xp_args &a, const xp_phase &P, int b, int j, float ay, bool same_xcd, uint32_t *s_max, float *wg_max) {
    const int tid = threadIdx.x, Gs = P.Gs, s = P.stride, W2 = P.W + 2;
    const int NTd = (int)x_div((uint32_t)XP_NT, P.fd_gs) * Gs, PP = NTd / Gs;
    const float bound = fminf(P.cap, P.gain * ay + P.off);
    const int ed = x_exp_of(__float_as_uint(bound));
    const float down = x_pow2(-ed);
    const uint32_t ypl = xp_ypl(P);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.arena, 0, a.arena_bytes, 0x00020000);
    const uint32_t dimg = a.d_off[P.dbuf] + (uint32_t)b * a.d_img_stride;
    if (tid == 0) s_max[0] = 0u;
    // the next pointwise phase's weights do not depend on anybody: copied to LDS under this pass
    if (P.nw_nks) xp_preload_weights(a, P.nw_off, P.nw_nks, P.nw_ncb, P.nw_nslab, j);
    float mx = 0.f;
    if (tid < NTd) {
        const int p0 = (int)x_div((uint32_t)tid, P.fd_gs), gl = tid - p0 * Gs;
        const int gch = j * Gs + gl;                                  // channel group in the whole tensor
        const float *wp = reinterpret_cast<const float *>(xsm + XP_NS * 8 * 6 * 1024 + XP_PAR_OFF) + gl * 8;   // [11][Gs*8], fetched by xp_fetch_par
        const int rowf = Gs * 8;
        float4 w0[9], w1[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            w0[t] = *reinterpret_cast<const float4 *>(wp + t * rowf);
            w1[t] = *reinterpret_cast<const float4 *>(wp + t * rowf + 4);
        }
        const float4 sc0 = *reinterpret_cast<const float4 *>(wp + 9 * rowf), sc1 = *reinterpret_cast<const float4 *>(wp + 9 * rowf + 4);
        const float4 bs0 = *reinterpret_cast<const float4 *>(wp + 10 * rowf), bs1 = *reinterpret_cast<const float4 *>(wp + 10 * rowf + 4);
        const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
        const float bs[8] = {bs0.x, bs0.y, bs0.z, bs0.w, bs1.x, bs1.y, bs1.z, bs1.w};
        const uint32_t kso = (uint32_t)(gch >> 2) * (uint32_t)P.nrb_out * 2048u;     // [k-step][16-pixel block][hi|lo][16][64 B]
        const int cch = gch & 3;
        for (int p = p0; p < P.Ho * P.Wo; p += PP) {
            const int py = (int)x_div((uint32_t)p, P.fd_wo), px = p - py * P.Wo;
            const int q0 = (py * s + 1 - P.pad_t) * W2 + px * s + 1 - P.pad_l;
            float2v acc2[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (X_DBG(a, 16)) break;
                const int at = ((q0 + (t / 3) * W2 + (t % 3)) * Gs + gl) * 16;
                const float4 x0 = *reinterpret_cast<const float4 *>(xsm + at), x1 = *reinterpret_cast<const float4 *>(xsm + ypl + at);
                acc2[0] = __builtin_elementwise_fma(float2v{x0.x, x0.y}, float2v{w0[t].x, w0[t].y}, acc2[0]);
                acc2[1] = __builtin_elementwise_fma(float2v{x0.z, x0.w}, float2v{w0[t].z, w0[t].w}, acc2[1]);
                acc2[2] = __builtin_elementwise_fma(float2v{x1.x, x1.y}, float2v{w1[t].x, w1[t].y}, acc2[2]);
                acc2[3] = __builtin_elementwise_fma(float2v{x1.z, x1.w}, float2v{w1[t].z, w1[t].w}, acc2[3]);
            }
            const float accv[8] = {acc2[0].x, acc2[0].y, acc2[1].x, acc2[1].y, acc2[2].x, acc2[2].y, acc2[3].x, acc2[3].y};
            float vd[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float v = x_actf(__builtin_fmaf(accv[k], sc[k], bs[k]), P.slope, P.cap);
                mx = fmaxf(mx, fabsf(v));
                vd[k] = v * down;
            }
            half8 hi, lo;
            x_split8(vd, hi, lo);
            const int r = p & 15;
            const uint32_t off = dimg + kso + (uint32_t)(p >> 4) * 2048u + (uint32_t)(r * 64 + ((cch ^ ((r >> 1) & 3)) * 16));
            if (X_DBG(a, 4)) continue;
            if (same_xcd || X_DBG(a, 8)) {
                xp_store16_plain(rs, off, __builtin_bit_cast(u32x4, hi));
                xp_store16_plain(rs, off + 1024u, __builtin_bit_cast(u32x4, lo));
            } else {
                xp_store16_sc1(rs, off, __builtin_bit_cast(u32x4, hi));
                xp_store16_sc1(rs, off + 1024u, __builtin_bit_cast(u32x4, lo));
            }
        }
    }
    __syncthreads();
    x_amax_lds(s_max, 0, mx);
    __syncthreads();
    *wg_max = __uint_as_float(s_max[0]);
    return ed;
}

// ---- XP_PW: y[slice] = act(BN(W[slice] x D)) -----------------------------------------------------------------------------
// A ring stage holds 8*PPW KB: slot s < 2*nrb = piece (16-pixel block s>>1, hi|lo) of D, then 2*ncb pieces of this workgroup's weights,
// then dummies (every wave issues exactly PPW pieces per step: the counted vmcnt stays uniform); wave w fills slots w, w+8, ...
// One k-step of a wave: NR row blocks x NC column blocks (its share of the nrb x ncb tile grid), three products per tile as three
// sweeps over the accumulators (consecutive MFMAs never share one); NR, NC are wave-uniform and fixed for the phase.
template <int NR, int NC, typename F>
__device__ __forceinline__ void xp_mma(const unsigned char *S, const unsigned char *Bs, const int (&rb)[3], const int (&cb)[3], int foff, floatx4 (&acc)[3][3],
                                       F &&issue) {
    half8 xh[NR], xl[NR], wh[NC], wl[NC];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        xh[i] = *reinterpret_cast<const half8 *>(S + rb[i] * 2048 + foff);
        xl[i] = *reinterpret_cast<const half8 *>(S + rb[i] * 2048 + 1024 + foff);
    }
#pragma unroll
    for (int jj = 0; jj < NC; ++jj) {
        wh[jj] = *reinterpret_cast<const half8 *>(Bs + cb[jj] * 2048 + foff);
        wl[jj] = *reinterpret_cast<const half8 *>(Bs + cb[jj] * 2048 + 1024 + foff);
    }
    // the next stage's operand pieces are requested BETWEEN the sweeps: a piece issues in the shadow of the matrix pipe working through
    // the sweep before it (issued in one block ahead of the reads they cost the wave ~100 cycles each with the pipe idle)
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int jj = 0; jj < NC; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[jj], xh[i], acc[i][jj], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    issue(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int jj = 0; jj < NC; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[jj], xl[i], acc[i][jj], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    issue(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int jj = 0; jj < NC; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[jj], xh[i], acc[i][jj], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    issue(2);
}

template <int PPW>
__device__ __forceinline__ void xp_pw(const xp_args &a, const xp_phase &P, int b, int j, int e_in, int pi) {
    constexpr int STG = 8 * PPW * 1024;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (P.nd_par) xp_fetch_par(P.nd_par, P.nd_Cp, P.nd_Gs, j);       // lands under the K loop (older than every ring piece: no count changes)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.arena, 0, a.arena_bytes, 0x00020000);
    const uint32_t nA = 2u * P.nrb, nT = nA + 2u * P.ncb;
    const uint32_t dimg = a.d_off[P.dbuf] + (uint32_t)b * a.d_img_stride;
    // this wave's tiles: row blocks wr + WR*i, column blocks wc + WC*jj (i, jj < 3)
    const int wr = wid / P.WC, wc = wid - wr * P.WC;
    int rb[3], cb[3], nr = 0, nc = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        rb[i] = wr + P.WR * i;
        nr += rb[i] < P.nrb ? 1 : 0;
        rb[i] = rb[i] < P.nrb ? rb[i] : 0;
        cb[i] = wc + P.WC * i;
        nc += cb[i] < P.ncb ? 1 : 0;
        cb[i] = cb[i] < P.ncb ? cb[i] : 0;
    }
    const int fr = lane & 15, fq = lane >> 4, nl4 = fq * 4;
    // BatchNorm scale / bias of this lane's channels: requested first, used last
    float4 scv[3], bsv[3];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        const int n = j * P.ncb * 16 + cb[jj] * 16 + nl4;
        scv[jj] = *reinterpret_cast<const float4 *>(P.scale + n);
        bsv[jj] = *reinterpret_cast<const float4 *>(P.bias + n);
    }
    uint32_t base[PPW], kstr[PPW];
#pragma unroll
    for (int t = 0; t < PPW; ++t) {
        const uint32_t s = (uint32_t)wid + 8u * t;
        if (s < nA) {
            base[t] = dimg + s * 1024u;
            kstr[t] = nA * 1024u;
        } else if (s < nT) {
            base[t] = P.w_off + ((uint32_t)(j * P.ncb) * 2u + (s - nA)) * 1024u;
            kstr[t] = (uint32_t)P.nslab * 2048u;
        } else {
            base[t] = X_OOB;
            kstr[t] = 0u;
        }
    }
    const int nks = P.nks;
    auto dma = [&](int stage, int ks, int grp) {                      // grp < 0: every piece of the wave; else pieces t = grp, grp + 3, ...
        const bool live = ks < nks;
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            if (grp >= 0 && (t % 3) != grp) continue;
            const uint32_t off = live ? base[t] + (uint32_t)ks * kstr[t] + lane * 16u : X_OOB;
            // D pieces were written by other CUs during this launch: sc1 loads (served by the L2, never by this CU's L1)
            // sc1 loads (served by the L2, never by this CU's L1): D was written by other CUs during this launch; the weights do not care
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(xsm + stage * STG + (wid + 8 * t) * 1024), 16, off, 0, 0, 16);
        }
    };
    floatx4 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) acc[i][jj] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int foff = fr * 64 + ((fq ^ ((fr >> 1) & 3)) * 16);
    const int shape = nr * 4 + nc;
    auto step = [&](int stage, int wstage, int ks_next) {             // MFMAs of `stage`, requests for `wstage` in between
        const unsigned char *S = xsm + stage * STG, *Bs = S + nA * 1024;
        auto issue = [&](int grp) {
            if (!X_DBG(a, 2)) dma(wstage, ks_next, grp);
        };
        if (X_DBG(a, 1)) {
            issue(-1);
            return;
        }
        switch (shape) {                                              // wave-uniform
        case 4 * 3 + 3: xp_mma<3, 3>(S, Bs, rb, cb, foff, acc, issue); break;
        case 4 * 3 + 2: xp_mma<3, 2>(S, Bs, rb, cb, foff, acc, issue); break;
        case 4 * 3 + 1: xp_mma<3, 1>(S, Bs, rb, cb, foff, acc, issue); break;
        case 4 * 2 + 3: xp_mma<2, 3>(S, Bs, rb, cb, foff, acc, issue); break;
        case 4 * 2 + 2: xp_mma<2, 2>(S, Bs, rb, cb, foff, acc, issue); break;
        case 4 * 2 + 1: xp_mma<2, 1>(S, Bs, rb, cb, foff, acc, issue); break;
        case 4 * 1 + 3: xp_mma<1, 3>(S, Bs, rb, cb, foff, acc, issue); break;
        case 4 * 1 + 2: xp_mma<1, 2>(S, Bs, rb, cb, foff, acc, issue); break;
        case 4 * 1 + 1: xp_mma<1, 1>(S, Bs, rb, cb, foff, acc, issue); break;
        default: issue(-1); break;                                    // a wave without tiles still feeds the ring
        }
    };
#pragma unroll
    for (int s = 0; s < XP_NS - 1; ++s) dma(s, s, -1);
    XP_ISTAMP(a, 4 * pi + 1)
    int rd = 0, wrs = XP_NS - 1;
    for (int kt = 0; kt < nks; ++kt) {
        x_wait_vm<(XP_NS - 2) * PPW>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        step(rd, wrs, kt + XP_NS - 1);
        rd = (rd + 1 == XP_NS) ? 0 : rd + 1;
        wrs = (wrs + 1 == XP_NS) ? 0 : wrs + 1;
    }
    XP_ISTAMP(a, 4 * pi + 2)
    x_wait_vm<0>();                                                   // the dead prefetches, before the ring becomes the y image
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    XP_ISTAMP(a, 4 * pi + 3)
    // ---- epilogue: lane holds channels nl..nl+3 of pixel rb*16 + fr
    const float up = x_pow2(e_in);
    const uint32_t ypl = xp_ypl(P);
    const int npx = P.H * P.W, W2 = P.W + 2;
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        if (jj >= nc) continue;
        const int nloc = cb[jj] * 16 + nl4;
        const float4 sc = scv[jj], bs = bsv[jj];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int m = rb[i] * 16 + fr;
            if (i >= nr || m >= npx) continue;
            float4 v;
            v.x = x_actf(__builtin_fmaf(acc[i][jj][0] * up, sc.x, bs.x), P.pslope, P.pcap);
            v.y = x_actf(__builtin_fmaf(acc[i][jj][1] * up, sc.y, bs.y), P.pslope, P.pcap);
            v.z = x_actf(__builtin_fmaf(acc[i][jj][2] * up, sc.z, bs.z), P.pslope, P.pcap);
            v.w = x_actf(__builtin_fmaf(acc[i][jj][3] * up, sc.w, bs.w), P.pslope, P.pcap);
            const int py = (int)x_div((uint32_t)m, P.fd_w), px = m - py * P.W;
            const int q = (py + 1) * W2 + px + 1;
            *reinterpret_cast<float4 *>(xsm + ((nloc >> 2) & 1) * ypl + (q * P.Gs + (nloc >> 3)) * 16) = v;
        }
    }
    xp_zero_border(P);
    __syncthreads();
}


// ---- XP_PW, one operand private to each wave: it never touches LDS ---------------------------------------------------------
// In a tile grid of nrb 16-pixel blocks x ncb 16-channel blocks either every pixel block belongs to ONE wave (14x20 pixels, 48 channels
// per workgroup: 18 x 3, wave w owns pixel blocks w, w+8, w+16 and all three channel blocks) or every channel block does (7x10 pixels, 96
// channels: 5 x 6, wave w < 6 owns channel block w and all five pixel blocks).  The OWN operand's fragments are used by one wave only:
// staging them in LDS costs an LDS-DMA write and a ds_read for nothing and makes a ring stage 36 KB (measured: the K loop was bound by
// 48 LDS-DMA issues + 96 KB of LDS reads per step behind one barrier, 0.8 us per step; profiles/r04_persist_phases.txt).  Here a wave
// loads its own fragments straight into registers (both operands are stored in fragment order: the lane's 16 bytes sit at piece + foff, a
// fully coalesced 1 KB load), two steps ahead, three register sets; the SHARED operand (a few pieces per step) is staged through a
// small LDS ring.  The y image no longer shares LDS with a ring.
// No LDS-DMA inside the loop: beside one, hipcc waits vmcnt(0) before the first use of ANY register a plain load filled (it drained the
// two-steps-ahead fragment loads every step); with plain loads only it counts exactly.  The shared pieces therefore travel
// global -> register (three steps ahead) -> ds_write (one step ahead of their use) -> fragment reads.
constexpr int XP_YB_BYTES = 68 * 1024, XP_BRING = XP_YB_BYTES;        // [y image][shared-operand ring 3 x 8*PS KB] ... [misc]
template <int NO, int NSH, int PS, bool OWNPIX>
__device__ __forceinline__ void xp_pw_direct_loop(const __amdgpu_buffer_rsrc_t rs, int nks, const uint32_t (&own_base)[3], uint32_t own_kstr,
                                                  const uint32_t (&sh_base)[2], uint32_t sh_kstr, int foff, int wid, int lane, floatx4 (&acc)[NO][NSH], int dbg) {
    (void)dbg;
    constexpr int STG = 8 * PS * 1024;
    half8 Oh[3][NO], Ol[3][NO];
    u32x4 Sr[3][PS];
    auto loadO = [&](half8 (&h)[NO], half8 (&l)[NO], int ks) {
#pragma unroll
        for (int i = 0; i < NO; ++i) {
            const uint32_t off = ks < nks ? own_base[i] + (uint32_t)ks * own_kstr + (uint32_t)foff : X_OOB;
            // the pixel operand was written by other CUs during this launch: sc1 (served by the L2, never by this CU's L1)
            if constexpr (OWNPIX) {
                h[i] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));
                l[i] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 1024u, 0, 16));
            } else {
                h[i] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
                l[i] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 1024u, 0, 0));
            }
        }
    };
    auto loadS = [&](u32x4 (&r)[PS], int ks) {
#pragma unroll
        for (int t = 0; t < PS; ++t) {
            const uint32_t off = ks < nks ? sh_base[t] + (uint32_t)ks * sh_kstr + lane * 16u : X_OOB;
            if constexpr (OWNPIX) {
                r[t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            } else {
                r[t] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));
            }
        }
    };
    auto writeS = [&](const u32x4 (&r)[PS], int st) {
#pragma unroll
        for (int t = 0; t < PS; ++t) *reinterpret_cast<u32x4 *>(xsm + XP_BRING + st * STG + (wid + 8 * t) * 1024 + lane * 16) = r[t];
    };
    auto mma = [&](const half8 &w, const half8 &x, floatx4 &c) { c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, c, 0, 0, 0); };
    half8 sh[NSH], sl_[NSH];                                          // the shared operand's fragments of the CURRENT step, read one step early
    auto readS = [&](int st) {
        const unsigned char *Ss = xsm + XP_BRING + st * STG;
#pragma unroll
        for (int jj = 0; jj < NSH; ++jj) {
            sh[jj] = *reinterpret_cast<const half8 *>(Ss + jj * 2048 + foff);
            sl_[jj] = *reinterpret_cast<const half8 *>(Ss + jj * 2048 + 1024 + foff);
        }
    };
    // step ks (st = ks % 3).  On entry every operand of the step is in registers: own fragments (requested at ks-2), shared fragments
    // (read from LDS at the end of ks-1) - the matrix pipe starts right behind the barrier.  In its shadow: the shared pieces of ks+2 go
    // to LDS (they are read at the end of ks+1), requests for the shared pieces of ks+4 and the own fragments of ks+2, and at the end the
    // LDS reads for ks+1.  Three products per tile as three sweeps over the accumulators (consecutive MFMAs never share one):
    // w_lo x_hi, w_hi x_lo, w_hi x_hi.
    auto step = [&](half8 (&ch)[NO], half8 (&cl)[NO], half8 (&nh)[NO], half8 (&nl)[NO], const u32x4 (&sw)[PS], u32x4 (&sld)[PS], int ks, int st) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the fragment reads of ks, this wave's ds_writes of ks-1
        __builtin_amdgcn_s_barrier();                                 // the pieces of ks+1 are in LDS; stage (ks+2)%3 has been read by everybody
        asm volatile("" ::: "memory");
        writeS(sw, st == 0 ? 2 : st - 1);                             // pieces of ks+2
#pragma unroll
        for (int i = 0; i < NO; ++i)
#pragma unroll
            for (int jj = 0; jj < NSH; ++jj) {
                if constexpr (OWNPIX) mma(sl_[jj], ch[i], acc[i][jj]);
                else mma(cl[i], sh[jj], acc[i][jj]);
            }
        __builtin_amdgcn_sched_barrier(0);
        loadS(sld, ks + 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NO; ++i)
#pragma unroll
            for (int jj = 0; jj < NSH; ++jj) {
                if constexpr (OWNPIX) mma(sh[jj], cl[i], acc[i][jj]);
                else mma(ch[i], sl_[jj], acc[i][jj]);
            }
        __builtin_amdgcn_sched_barrier(0);
        loadO(nh, nl, ks + 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NO; ++i)
#pragma unroll
            for (int jj = 0; jj < NSH; ++jj) {
                if constexpr (OWNPIX) mma(sh[jj], ch[i], acc[i][jj]);
                else mma(ch[i], sh[jj], acc[i][jj]);
            }
        __builtin_amdgcn_sched_barrier(0);
        readS(st == 2 ? 0 : st + 1);                                  // fragments of ks+1
    };
    loadS(Sr[0], 0);
    loadS(Sr[1], 1);
    loadO(Oh[0], Ol[0], 0);
    loadS(Sr[2], 2);
    loadO(Oh[1], Ol[1], 1);
    writeS(Sr[0], 0);
    writeS(Sr[1], 1);
    loadS(Sr[0], 3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    readS(0);
    for (int kt = 0; kt < nks; kt += 3) {                             // steps past nks multiply zeros (out-of-range loads)
        // (own set of ks, own set for ks+2, pieces written = ks+2, pieces requested = ks+4 into the set ks+1 has vacated)
        step(Oh[0], Ol[0], Oh[2], Ol[2], Sr[2], Sr[1], kt, 0);
        step(Oh[1], Ol[1], Oh[0], Ol[0], Sr[0], Sr[2], kt + 1, 1);
        step(Oh[2], Ol[2], Oh[1], Ol[1], Sr[1], Sr[0], kt + 2, 2);
    }
}

// The same with the shared operand RESIDENT: when the workgroup's whole weight slice (nks x NSH x 2 KB) fits in LDS beside the y image,
// it is deposited once (LDS-DMA, before the loop) and the K loop has no staging, no barrier and no lockstep at all: a wave streams its own
// fragments two steps ahead and reads the step's weight fragments from LDS.  (Measured before: with per-step staging the loop ran at
// 0.65 us per step even with the MFMAs and the fragment loads knocked out - the staged pieces' load latency, two steps deep, set the pace.)
// NKS > 0: the k-step count is a compile-time constant and the loop is unrolled completely - in straight-line code hipcc's wait counts
// are exact; around a loop with register sets rotating through it, it put a vmcnt(0) at the head of every iteration.
template <int NO, int NSH, int NKS>
__device__ __forceinline__ void xp_pw_resident_loop(const __amdgpu_buffer_rsrc_t rs, int nks_rt, const uint32_t (&own_base)[3], uint32_t own_kstr, int foff,
                                                    floatx4 (&acc)[NO][NSH], int dbg) {
    (void)dbg;
    const int nks = NKS > 0 ? NKS : nks_rt;
    half8 Oh[3][NO], Ol[3][NO];
    auto loadO = [&](half8 (&h)[NO], half8 (&l)[NO], int ks) {
#pragma unroll
        for (int i = 0; i < NO; ++i) {
            const uint32_t off = ks < nks ? own_base[i] + (uint32_t)ks * own_kstr + (uint32_t)foff : X_OOB;
            h[i] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));
            l[i] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 1024u, 0, 16));
        }
    };
    auto mma = [&](const half8 &w, const half8 &x, floatx4 &c) { c = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, c, 0, 0, 0); };
    auto step = [&](half8 (&ch)[NO], half8 (&cl)[NO], half8 (&nh)[NO], half8 (&nl)[NO], int ks) {
        const unsigned char *Ws = xsm + XP_BRING + (ks < nks ? ks : 0) * (NSH * 2048);
        half8 wh[NSH], wl[NSH];
#pragma unroll
        for (int jj = 0; jj < NSH; ++jj) {
            wh[jj] = *reinterpret_cast<const half8 *>(Ws + jj * 2048 + foff);
            wl[jj] = *reinterpret_cast<const half8 *>(Ws + jj * 2048 + 1024 + foff);
        }
#pragma unroll
        for (int i = 0; i < NO; ++i)
#pragma unroll
            for (int jj = 0; jj < NSH; ++jj) mma(wl[jj], ch[i], acc[i][jj]);
        __builtin_amdgcn_sched_barrier(0);
        loadO(nh, nl, ks + 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NO; ++i)
#pragma unroll
            for (int jj = 0; jj < NSH; ++jj) mma(wh[jj], cl[i], acc[i][jj]);
#pragma unroll
        for (int i = 0; i < NO; ++i)
#pragma unroll
            for (int jj = 0; jj < NSH; ++jj) mma(wh[jj], ch[i], acc[i][jj]);
    };
    loadO(Oh[0], Ol[0], 0);
    loadO(Oh[1], Ol[1], 1);
    if constexpr (NKS > 0) {
#pragma unroll
        for (int kt = 0; kt < NKS; kt += 3) {
            step(Oh[0], Ol[0], Oh[2], Ol[2], kt);
            step(Oh[1], Ol[1], Oh[0], Ol[0], kt + 1);
            step(Oh[2], Ol[2], Oh[1], Ol[1], kt + 2);
        }
    } else {
        for (int kt = 0; kt < nks; kt += 3) {                         // steps past nks multiply zeros (out-of-range fragment loads)
            step(Oh[0], Ol[0], Oh[2], Ol[2], kt);
            step(Oh[1], Ol[1], Oh[0], Ol[0], kt + 1);
            step(Oh[2], Ol[2], Oh[1], Ol[1], kt + 2);
        }
    }
}

// The workgroup's weight slice of a pointwise phase -> LDS (piece p = (k-step, channel block, hi|lo) at p KB); wave w deposits
// p = w, w+8, ...  Issued by the DEPTHWISE phase before it, in inline assembly: hipcc must not know about these LDS-DMAs, or it parks a
// vmcnt(0) in front of the depthwise pass's first LDS read (it cannot tell the y image from the weight region) and the copy, instead
// of running under the pass, is waited for.  The consumer waits for it explicitly (vmcnt(0) + barrier at the head of the pointwise phase).
// M0 (the LDS-DMA destination) is compiler-reserved: saved, set and restored inside one statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void xp_glds16_asm(const uint8_t *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void xp_preload_weights(const xp_args &a, uint32_t w_off, int nks, int ncb, int nslab, int j) {
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int np = nks * ncb * 2, per = 2 * ncb;
    const uint32_t lds0 = (uint32_t)(unsigned long)(lds_ptr_t)(xsm + XP_BRING);
    for (int p = wid; p < np; p += 8) {
        const int ks = p / per, r = p - ks * per;
        const uint8_t *src = a.arena + (size_t)w_off + (size_t)ks * (size_t)nslab * 2048u + (size_t)(j * per + r) * 1024u + lane * 16u;
        xp_glds16_asm(src, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + (uint32_t)p * 1024u)));
    }
}

// OWNPIX: own blocks = pixel blocks wid + 8*i (i < NO), shared = the workgroup's NSH channel blocks.
// !OWNPIX: own blocks = channel blocks wid + 8*i, shared = the image's NSH pixel blocks.
template <int NO, int NSH, int PS, bool OWNPIX>
__device__ __forceinline__ void xp_pw_direct_body(const xp_args &a, const xp_phase &P, int b, int j, int e_in, int pi) {
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.arena, 0, a.arena_bytes, 0x00020000);
    const uint32_t dimg = a.d_off[P.dbuf] + (uint32_t)b * a.d_img_stride;
    const uint32_t wslice = P.w_off + (uint32_t)(j * P.ncb) * 2048u;  // this workgroup's channel blocks inside a k-step of the weights
    const uint32_t pix_kstr = (uint32_t)P.nrb * 2048u, w_kstr = (uint32_t)P.nslab * 2048u;
    const int nown = OWNPIX ? P.nrb : P.ncb;
    int ob[3], no = 0;
    uint32_t own_base[3], sh_base[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ob[i] = wid + 8 * i;
        no += (ob[i] < nown && i < NO) ? 1 : 0;
        ob[i] = ob[i] < nown ? ob[i] : 0;                             // (a wave without blocks works on block 0; the result is discarded)
        own_base[i] = (OWNPIX ? dimg : wslice) + (uint32_t)ob[i] * 2048u;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int s = wid + 8 * t;                                    // piece s of the shared operand's step (block s>>1, hi|lo)
        sh_base[t] = s < 2 * NSH ? (OWNPIX ? wslice : dimg) + (uint32_t)s * 1024u : X_OOB;
    }
    const int fr = lane & 15, fq = lane >> 4, nl4 = fq * 4;
    const int foff = fr * 64 + ((fq ^ ((fr >> 1) & 3)) * 16);
    floatx4 acc[NO][NSH];
#pragma unroll
    for (int i = 0; i < NO; ++i)
#pragma unroll
        for (int jj = 0; jj < NSH; ++jj) acc[i][jj] = floatx4{0.f, 0.f, 0.f, 0.f};
    // BatchNorm scale / bias of this lane's channels: requested first, used last
    constexpr int NCH = OWNPIX ? NSH : NO;
    float4 scv[NCH], bsv[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int n = j * P.ncb * 16 + (OWNPIX ? c : ob[c]) * 16 + nl4;
        scv[c] = *reinterpret_cast<const float4 *>(P.scale + n);
        bsv[c] = *reinterpret_cast<const float4 *>(P.bias + n);
    }
    if (OWNPIX && P.resident) {
        if (P.resident == 1) xp_preload_weights(a, P.w_off, P.nks, P.ncb, P.nslab, j);      // (2: the phase before it has asked for them already)
        // the BUILTIN wait, not inline asm: hipcc must know the LDS-DMA has landed, or it drains the fragment loads (vmcnt(0)) in front of
        // the first weight-fragment read of every loop iteration (it did: one full load latency per three steps)
        __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0)
        __syncthreads();
        XP_ISTAMP(a, 4 * pi + 1)
        if (P.nks == 12) xp_pw_resident_loop<NO, NSH, 12>(rs, 12, own_base, pix_kstr, foff, acc, a.dbg);
        else xp_pw_resident_loop<NO, NSH, 0>(rs, P.nks, own_base, pix_kstr, foff, acc, a.dbg);
    } else {
        XP_ISTAMP(a, 4 * pi + 1)
        xp_pw_direct_loop<NO, NSH, PS, OWNPIX>(rs, P.nks, own_base, OWNPIX ? pix_kstr : w_kstr, sh_base, OWNPIX ? w_kstr : pix_kstr, foff, wid, lane, acc, a.dbg);
    }
    XP_ISTAMP(a, 4 * pi + 2)
    // the next depthwise phase's parameter slice: requested here, not ahead of the loop (a pending LDS-DMA makes hipcc drain every
    // load in front of the loop's LDS reads); it lands under the epilogue
    if (P.nd_par) xp_fetch_par(P.nd_par, P.nd_Cp, P.nd_Gs, j);
    // ---- epilogue: lane holds channels nl..nl+3 of pixel pb*16 + fr; the y image has its own LDS (nobody reads it during this phase)
    const float up = x_pow2(e_in);
    const uint32_t ypl = xp_ypl(P);
    const int npx = P.H * P.W, W2 = P.W + 2;
#pragma unroll
    for (int i = 0; i < NO; ++i) {
        if (i >= no) continue;
#pragma unroll
        for (int jj = 0; jj < NSH; ++jj) {
            const int cbk = OWNPIX ? jj : ob[i], pbk = OWNPIX ? ob[i] : jj;
            const int nloc = cbk * 16 + nl4, m = pbk * 16 + fr;
            if (m >= npx) continue;
            const float4 sc = scv[OWNPIX ? jj : i], bs = bsv[OWNPIX ? jj : i];
            float4 v;
            v.x = x_actf(__builtin_fmaf(acc[i][jj][0] * up, sc.x, bs.x), P.pslope, P.pcap);
            v.y = x_actf(__builtin_fmaf(acc[i][jj][1] * up, sc.y, bs.y), P.pslope, P.pcap);
            v.z = x_actf(__builtin_fmaf(acc[i][jj][2] * up, sc.z, bs.z), P.pslope, P.pcap);
            v.w = x_actf(__builtin_fmaf(acc[i][jj][3] * up, sc.w, bs.w), P.pslope, P.pcap);
            const int py = (int)x_div((uint32_t)m, P.fd_w), px = m - py * P.W;
            const int q = (py + 1) * W2 + px + 1;
            *reinterpret_cast<float4 *>(xsm + ((nloc >> 2) & 1) * ypl + (q * P.Gs + (nloc >> 3)) * 16) = v;
        }
    }
    XP_ISTAMP(a, 4 * pi + 3)
    if (P.zero_border) xp_zero_border(P);
    __syncthreads();
}
// a wave's number of own blocks (wave-uniform: 18 pixel blocks over 8 waves = 3,3,2,2,2,2,2,2) picks its loop; every variant runs the same
// number of barriers
template <int NOMAX, int NSH, int PS, bool OWNPIX>
__device__ __forceinline__ void xp_pw_direct(const xp_args &a, const xp_phase &P, int b, int j, int e_in, int pi) {
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nown = OWNPIX ? P.nrb : P.ncb;
    int no = 0;
#pragma unroll
    for (int i = 0; i < NOMAX; ++i) no += (wid + 8 * i) < nown ? 1 : 0;
    if constexpr (NOMAX >= 3) {
        if (no >= 3) return xp_pw_direct_body<3, NSH, PS, OWNPIX>(a, P, b, j, e_in, pi);
    }
    if constexpr (NOMAX >= 2) {
        if (no == 2) return xp_pw_direct_body<2, NSH, PS, OWNPIX>(a, P, b, j, e_in, pi);
    }
    return xp_pw_direct_body<1, NSH, PS, OWNPIX>(a, P, b, j, e_in, pi);
}

__global__ void __launch_bounds__(XP_NT) xp_kernel(const xp_args a) {
    uint32_t *s_max = reinterpret_cast<uint32_t *>(xsm + (XP_NS * 8 * 6 * 1024));        // XP_MISC bytes behind the largest ring
    // block -> (cluster, member): blocks run on XCD (block % 8), so the members of a cluster are blocks 8 apart
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int cq = slot / a.CW, j = slot - cq * a.CW;
    const int c0 = cq * 8 + xcd;
    const uint32_t my_xcc = (uint32_t)__builtin_amdgcn_s_getreg(6164) & 15u;       // hwreg(HW_REG_XCC_ID, 0, 4)
#ifdef YK_DEV
#define XP_STAMP(k) \
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * (4 * XP_MAXPH + 4) + (k)] = (long long)wall_clock64();
#else
#define XP_STAMP(k)
#endif
    for (int b = c0; b < a.B; b += a.n_cluster) {
        float ay = 0.f, md = 0.f;
        int ed = 0;
        uint32_t arrivals = 0;
        bool same_xcd = false;                                        // not known before the first barrier
        if (threadIdx.x == 0) __hip_atomic_store(a.pxcc + (size_t)b * a.CW + j, my_xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int pi = 0; pi < a.n_phase; ++pi) {
            const xp_phase P = a.ph[pi];                              // by value: the fields live in registers, not behind a pointer
            XP_STAMP(4 * pi)                                          // that every store of the phase could have written through
            if (P.type == XP_LOAD) {
                if (P.nd_par) xp_fetch_par(P.nd_par, P.nd_Cp, P.nd_Gs, j);
                ay = xp_load(P, b, j);
            } else if (P.type == XP_DW) {
                float wg_max;
                ed = xp_dw(a, P, b, j, ay, same_xcd, s_max, &wg_max);
                ++arrivals;
                XP_STAMP(4 * pi + 1)
                md = xp_cluster_barrier(a.gran + (size_t)b * 2 * a.CW, a.CW, j, arrivals, wg_max, s_max, a.err);
                if (arrivals == 1u) {                                 // first barrier of the image: where does everybody run?
                    bool same = true;
                    for (int k = 0; k < a.CW; ++k) same = same && __hip_atomic_load(a.pxcc + (size_t)b * a.CW + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_xcc + 1u;
                    same_xcd = same && !a.write_through && !X_DBG(a, 32);
                }
            } else if (P.type == XP_PW) {
                ay = fminf(P.pcap, P.pgain * md + P.poff);
                // (own blocks per wave, shared blocks, shared pieces per wave, which operand is private) - instantiated for the shapes
                // x_build_persist accepts; anything else keeps the LDS-ring form below
                if (P.adirect == 1) xp_pw_direct<3, 3, 1, true>(a, P, b, j, ed, pi);          // <= 24 pixel blocks x 3 channel blocks
                else if (P.adirect == 2) xp_pw_direct<1, 5, 2, false>(a, P, b, j, ed, pi);    // 5 pixel blocks x <= 8 channel blocks
                else if (P.adirect == 3) xp_pw_direct<3, 2, 1, true>(a, P, b, j, ed, pi);
                else if (P.adirect == 4) xp_pw_direct<2, 5, 2, false>(a, P, b, j, ed, pi);    // 5 pixel blocks x <= 16 channel blocks
                else if (P.ppw == 3) xp_pw<3>(a, P, b, j, ed, pi);
                else if (P.ppw == 4) xp_pw<4>(a, P, b, j, ed, pi);
                else if (P.ppw == 5) xp_pw<5>(a, P, b, j, ed, pi);
                else xp_pw<6>(a, P, b, j, ed, pi);
            } else {
                xp_store(P, b, j, ay, s_max);
            }
        }
        XP_STAMP(4 * a.n_phase)
        __syncthreads();
    }
#undef XP_STAMP
}
