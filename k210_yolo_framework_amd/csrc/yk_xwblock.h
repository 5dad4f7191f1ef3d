// yk_xwblock.h — f16x2 mode: the fused DepthwiseConv2D 3x3 + BN + act -> Conv2D 1x1 + BN + act block of yk_xblock.h with its two phases
// on DIFFERENT waves (included by yk_exact.hip).  MobileNet block: keras_mobilenet.py:359-436.
//
// xb_kernel runs a 32-channel step as  dw(k) -> barrier -> mma(k) -> barrier  on the same four waves: while they run the nine taps on the VALU
// the matrix cores idle, and the other way round.  For the 12-step blocks at 14x20x384 (five launches, 30 % of the step's kernel time) that
// chain is 2.6 us per step with ONE workgroup per CU (160 workgroups) - 36 us per launch for 3.5 us of roofline work.  Here the workgroup
// has eight waves in two roles:
//
//   producers (waves 0-3)   DMA of the patch of step k+1 (two patch stages), then dw(k): nine taps, BN, activation, split -> A tile k&1
//   consumers (waves 4-7)   mma(k-1) from A tile (k-1)&1 with the weight fragments of step k-1 (global memory -> registers, requested when
//                           mma(k-2) was over: they land while the consumer waits for the producers), then request the fragments of step k
//
// ONE s_barrier per step: the VALU phase of step k and the MFMA phase of step k-1 overlap, a step costs max(dw, mma) instead of their sum.
// Same arithmetic, same order per output element as xb_kernel (bit-identical results).  Registers: the consumers' 96 accumulators + one set of
// fragments + the pixel fragments do not fit 128, so a workgroup (8 waves x <= 256 registers) has its CU to itself; LDS: two patch stages +
// two A tiles (44 KB at 14x4 pixels).
#pragma once

template <int TM, int TN>
struct xw_cfg {
    static constexpr int BM = 16 * TM, BN = 64 * TN;
    static constexpr int PARB = 2048;
    static constexpr int CPITCH = BN * 4 + 16;
    static constexpr int IPP = (TN >= 3 && TM >= 2) ? (TM + 1) / 2 : TM;
    static constexpr int lds(int n16p) {
        const int r = 2 * (n16p * 32 + PARB) + 2 * BM * 128, ct = IPP * 16 * CPITCH;
        return (r > ct ? r : ct) + 64;
    }
};

template <int TM, int TN>
__global__ void __launch_bounds__(512) xw_kernel(const xb_args a) {
    typedef xw_cfg<TM, TN> C;
    constexpr int BM = C::BM, BN = C::BN, GL = 4;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const bool producer = wid < 4;                                    // wave-uniform
    const int ptid = tid & 255, pwid = wid & 3;                       // thread / wave inside its role
    const int G = a.in.G, s = a.stride;
    const int STG = a.n16p * 32 + C::PARB;
    unsigned char *A0 = xsm + 2 * STG;
    float *sf = reinterpret_cast<float *>(xsm + a.lds_bytes - 64);   // [0] 2^e_in [1] 2^-e_mid [2] 2^e_mid [3] 2^-e_out [4] 2^e_res
    uint32_t *smax = reinterpret_cast<uint32_t *>(sf + 8);
    const int bid = x_xcd_tile(blockIdx.x, gridDim.x);
    const uint32_t b = x_div((uint32_t)bid, a.fd_tpi), tl = bid - b * (a.tiles_x * a.tiles_y);
    const uint32_t ty = x_div(tl, a.fd_tx), tx = tl - ty * a.tiles_x;
    const int oy0 = (int)ty * a.TH, ox0 = (int)tx * a.TW;
    const int iy0 = oy0 * s - a.pad_t, ix0 = ox0 * s - a.pad_l;
    const int n0 = blockIdx.y * BN;
    const int fr = lane & 15, fq = lane >> 4, nl4 = fq * 4;
    const uint32_t img = (uint32_t)a.in.H * a.in.W * G * 32u;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void *)(a.in.p + (size_t)b * img), 0, img, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc((void *)a.par, 0, (uint32_t)(11 * G * 32), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);
    // (producers) this lane's patch positions are the same for every step
    constexpr int PQ = 6;
    uint32_t poff[PQ];
#pragma unroll
    for (int i = 0; i < PQ; ++i) {
        poff[i] = X_OOB;
        if (producer && i * 256 < a.n16p) {
            const uint32_t q = (uint32_t)(i * 256 + ptid);
            const uint32_t pos = q >> 2, g4 = q & 3;
            const uint32_t r = x_div(pos, a.fd_pw), c = pos - r * a.PW;
            const int iy = iy0 + (int)r, ix = ix0 + (int)c;
            const bool ok = (int)q < a.n16 && (unsigned)iy < (unsigned)a.in.H && (unsigned)ix < (unsigned)a.in.W;
            poff[i] = ok ? (uint32_t)(((iy * a.in.W + ix) * G + (int)g4) * 32) : X_OOB;
        }
    }
    const int g4l = ptid & 3;
    const int rot = a.nk == 1 ? 0 : (int)(tl - x_div(tl, a.fd_nk) * (uint32_t)a.nk);
    auto kstep = [&](int i) {
        const int k = i + rot;
        return k >= a.nk ? k - a.nk : k;
    };
    const uint32_t wstep = (uint32_t)a.nslab * 2048u;
    auto dma_patch = [&](int it_) {                                   // (producers) patch + depthwise parameters of loop step it_ -> stage it_ & 1
        const int ks = kstep(it_);
        unsigned char *HI = xsm + (it_ & 1) * STG, *LO = HI + a.n16p * 16, *PARb = HI + a.n16p * 32;
        const bool gok = (ks * 4 + g4l) < G;
        const uint32_t koff = gok ? (uint32_t)ks * 128u : X_OOB;
#pragma unroll
        for (int i = 0; i < PQ; ++i)
            if (i * 256 + pwid * 64 < a.n16p) {
                const uint32_t oh = poff[i] + koff, ol = oh + 16u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr_t)(HI + (i * 256 + pwid * 64) * 16), 16, oh, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr_t)(LO + (i * 256 + pwid * 64) * 16), 16, ol, 0, 0, 0);
            }
        if (pwid < 2) {
            const int idx = pwid * 64 + lane, t = idx >> 3, pc = idx & 7;
            const int ch = ks * 32 + pc * 4;
            const uint32_t op = (idx < 88 && ch < G * 8) ? (uint32_t)((t * G * 8 + ch) * 4) : X_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsp, (lds_ptr_t)(PARb + pwid * 1024), 16, op, 0, 0, 0);
        }
    };
    const bool wave_live = !producer && n0 + pwid * TN * 16 < a.N;    // a consumer wave with channels to multiply
    const int foff = fr * 64 + ((fq ^ ((fr >> 1) & 3)) * 16);
    half8 wh[TN], wl[TN];
    auto load_w = [&](int it_) {                                      // (consumers) the pointwise weight fragments of loop step it_ -> registers
        const uint32_t ws = (uint32_t)(n0 >> 4) * 2048u + (uint32_t)kstep(it_) * wstep + (uint32_t)(pwid * TN) * 2048u + (uint32_t)foff;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            wh[j] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsw, ws + (uint32_t)j * 2048u, 0, 0));
            wl[j] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsw, ws + (uint32_t)j * 2048u + 1024u, 0, 0));
        }
    };
    if (producer) dma_patch(0);
    else if (wave_live) load_w(0);
    if (wid == 0) {                                                   // per-image factors (one image per workgroup)
        const float amax_in = x_amax_wave(a.in.amax, (int)b);
        const float bmid = fminf(a.dw_cap, a.dw_gain * amax_in + a.dw_off);
        float bout = fminf(a.cap, a.gain * bmid + a.off);
        float rup = 0.f;
        if (a.res.p) {
            bout += x_amax_wave(a.res.amax, (int)b);
            rup = x_pow2(a.res.eexp[b]);
        }
        const int em = x_exp_of(__float_as_uint(bmid)), eo = a.dst_f32 ? 0 : x_exp_of(__float_as_uint(bout));
        if (lane == 0) {
            sf[0] = x_pow2(a.in.eexp[b]);
            sf[1] = x_pow2(-em);
            sf[2] = x_pow2(em);
            sf[3] = x_pow2(-eo);
            sf[4] = rup;
            smax[0] = 0u;
            a.eexp_out[b] = eo;
        }
    }
    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int nk = a.nk;
    for (int it = 0; it <= nk; ++it) {
        // producers: this wave's pieces of patch(it) have landed, its A-tile writes of dw(it-1) are done; consumers: mma(it-2)... mma(it-1)'s
        // operands are complete only after the barrier.  The consumers' weight loads stay in flight across it.
        if (producer) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (producer) {
            if (it >= nk) continue;
            if (it + 1 < nk) dma_patch(it + 1);                       // stage (it+1)&1: last read by dw(it-1), which every producer finished before the barrier
            const unsigned char *HI = xsm + (it & 1) * STG, *LO = HI + a.n16p * 16;
            const unsigned char *PARB_ = HI + a.n16p * 32;
            unsigned char *A = A0 + (it & 1) * (BM * 128);           // last read by mma(it-2), finished before the barrier
            const float up = sf[0], dmid = sf[1];
            const bool f32patch = a.src_f32 != 0;
            // ---- depthwise: item = (pixel p, group q of this step)
            for (int item = ptid; item < BM * GL; item += 256) {
                const int p = item >> 2, q = item & 3;
                const int py = (int)x_div((uint32_t)p, a.fd_tw), px = p - py * a.TW;
                const bool live = py < a.TH && oy0 + py < a.Ho && ox0 + px < a.Wo;
                half8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = hi;
                if (live) {
                    const int base = ((py * s) * a.PW + px * s) * GL + q;
                    float2v d2[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int at = (base + ((t / 3) * a.PW + (t % 3)) * GL) * 16;
                        const u32x4 h = *reinterpret_cast<const u32x4 *>(HI + at), l = *reinterpret_cast<const u32x4 *>(LO + at);
                        const u32x4 w0 = *reinterpret_cast<const u32x4 *>(PARB_ + (t * 32 + q * 8) * 4), w1 = *reinterpret_cast<const u32x4 *>(PARB_ + (t * 32 + q * 8 + 4) * 4);
                        const float2v w2[4] = {{__uint_as_float(w0[0]), __uint_as_float(w0[1])}, {__uint_as_float(w0[2]), __uint_as_float(w0[3])},
                                               {__uint_as_float(w1[0]), __uint_as_float(w1[1])}, {__uint_as_float(w1[2]), __uint_as_float(w1[3])}};
                        if (f32patch) {                               // (h, l) are the fp32 planes: channels 0-3 | 4-7
                            d2[0] = __builtin_elementwise_fma(float2v{__uint_as_float(h[0]), __uint_as_float(h[1])}, w2[0], d2[0]);
                            d2[1] = __builtin_elementwise_fma(float2v{__uint_as_float(h[2]), __uint_as_float(h[3])}, w2[1], d2[1]);
                            d2[2] = __builtin_elementwise_fma(float2v{__uint_as_float(l[0]), __uint_as_float(l[1])}, w2[2], d2[2]);
                            d2[3] = __builtin_elementwise_fma(float2v{__uint_as_float(l[2]), __uint_as_float(l[3])}, w2[3], d2[3]);
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float2v x2 = {x_mix_sum_lo(h[j], l[j]), x_mix_sum_hi(h[j], l[j])};
                                d2[j] = __builtin_elementwise_fma(x2, w2[j], d2[j]);
                            }
                        }
                    }
                    const float d[8] = {d2[0].x, d2[0].y, d2[1].x, d2[1].y, d2[2].x, d2[2].y, d2[3].x, d2[3].y};
                    const u32x4 s0_ = *reinterpret_cast<const u32x4 *>(PARB_ + (9 * 32 + q * 8) * 4), s1_ = *reinterpret_cast<const u32x4 *>(PARB_ + (9 * 32 + q * 8 + 4) * 4);
                    const u32x4 b0_ = *reinterpret_cast<const u32x4 *>(PARB_ + (10 * 32 + q * 8) * 4), b1_ = *reinterpret_cast<const u32x4 *>(PARB_ + (10 * 32 + q * 8 + 4) * 4);
                    const float scd[8] = {__uint_as_float(s0_[0]), __uint_as_float(s0_[1]), __uint_as_float(s0_[2]), __uint_as_float(s0_[3]),
                                          __uint_as_float(s1_[0]), __uint_as_float(s1_[1]), __uint_as_float(s1_[2]), __uint_as_float(s1_[3])};
                    const float bsd[8] = {__uint_as_float(b0_[0]), __uint_as_float(b0_[1]), __uint_as_float(b0_[2]), __uint_as_float(b0_[3]),
                                          __uint_as_float(b1_[0]), __uint_as_float(b1_[1]), __uint_as_float(b1_[2]), __uint_as_float(b1_[3])};
                    float vd[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) vd[j] = x_actf(__builtin_fmaf(d[j] * up, scd[j], bsd[j]), a.dw_slope, a.dw_cap) * dmid;
                    x_split8(vd, hi, lo);
                }
                const int r = p & 15;
                unsigned char *dst = A + (p >> 4) * (GL * 512) + r * (GL * 16) + ((q ^ ((r >> 1) & 3)) * 16);
                *reinterpret_cast<half8 *>(dst) = hi;
                *reinterpret_cast<half8 *>(dst + GL * 256) = lo;
            }
        } else {
            if (it == 0 || !wave_live) continue;
            // ---- pointwise step it-1: three products per tile
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the fragments of step it-1 (requested one whole producer phase ago)
            const unsigned char *A = A0 + ((it - 1) & 1) * (BM * 128);
            half8 xh[TM], xl[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                xh[i] = *reinterpret_cast<const half8 *>(A + i * (GL * 512) + foff);
                xl[i] = *reinterpret_cast<const half8 *>(A + i * (GL * 512) + GL * 256 + foff);
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
            if (it < nk) {
                __builtin_amdgcn_sched_barrier(0);                    // the fragment registers are free only behind the last MFMA that reads them
                load_w(it);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                     // LDS becomes the output tile
    asm volatile("" ::: "memory");
    // ---- epilogue (consumers hold the accumulators): lane holds channels n..n+3 of pixel i*16 + fr
    unsigned char *Cs = xsm;
    const float umid = sf[2], dout = sf[3], rup = sf[4];
    float rmax = 0.f;
    constexpr int VPR = BN / 4, IPP = C::IPP;
    float4 scu[TN], bs[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (pwid * TN + j) * 16 + nl4;
        const float4 sc = *reinterpret_cast<const float4 *>(a.scale + n);
        bs[j] = *reinterpret_cast<const float4 *>(a.bias + n);
        scu[j] = float4{sc.x * umid, sc.y * umid, sc.z * umid, sc.w * umid};
    }
    if (a.dst_f32 && !a.res.p) {
        if (wave_live) {
            const uint32_t oimg = (uint32_t)a.Ho * a.Wo * a.outG * 32u;
            const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + (size_t)b * oimg), 0, oimg, 0x00020000);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int p = i * 16 + fr;
                const int py = (int)x_div((uint32_t)p, a.fd_tw), px = p - py * a.TW;
                const int oy = oy0 + py, ox = ox0 + px;
                const bool mok = py < a.TH && oy < a.Ho && ox < a.Wo;
                const uint32_t po = (uint32_t)((oy * a.Wo + ox) * a.outG) * 32u;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + (pwid * TN + j) * 16 + nl4;
                    float v[4];
                    v[0] = x_actf(__builtin_fmaf(acc[i][j][0], scu[j].x, bs[j].x), a.slope, a.cap);
                    v[1] = x_actf(__builtin_fmaf(acc[i][j][1], scu[j].y, bs[j].y), a.slope, a.cap);
                    v[2] = x_actf(__builtin_fmaf(acc[i][j][2], scu[j].z, bs[j].z), a.slope, a.cap);
                    v[3] = x_actf(__builtin_fmaf(acc[i][j][3], scu[j].w, bs[j].w), a.slope, a.cap);
                    if (mok && (n >> 3) < a.outG) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) rmax = fmaxf(rmax, fabsf(v[k]));
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, rso,
                                                               po + (uint32_t)((n >> 3) * 32 + ((n >> 2) & 1) * 16), 0, 0);
                    }
                }
            }
        }
        x_amax_lds(smax, 0, rmax);
        __syncthreads();
        if (tid == 0 && smax[0]) x_amax_global(a.amax_out + (size_t)b * XS, smax[0]);
        return;
    }
    // (staged form) the tile leaves through LDS in passes of IPP row blocks; all 512 threads copy out
    const uint32_t simg = (uint32_t)a.Ho * a.Wo * a.outG * 32u, rimg = (uint32_t)a.Ho * a.Wo * a.res.G * 32u;
    const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + (size_t)b * simg), 0, simg, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc((void *)(a.res.p ? a.res.p + (size_t)b * rimg : a.out), 0, a.res.p ? rimg : 0u, 0x00020000);
#pragma unroll
    for (int i0 = 0; i0 < TM; i0 += IPP) {
        if (i0 > 0) __syncthreads();                                  // the previous pass has been copied out
        if (!producer) {
#pragma unroll
            for (int i = i0; i < i0 + IPP && i < TM; ++i) {
                if (!wave_live && pwid != 0) break;                   // (consumer wave 0 also writes the rows' pixel indices)
                const int p = i * 16 + fr;
                const int py = (int)x_div((uint32_t)p, a.fd_tw), px = p - py * a.TW;
                const int oy = oy0 + py, ox = ox0 + px;
                const bool mok = py < a.TH && oy < a.Ho && ox < a.Wo;
                const int m = oy * a.Wo + ox;
                if (pwid == 0 && fq == 0) *reinterpret_cast<int *>(Cs + (p - i0 * 16) * C::CPITCH + BN * 4) = mok ? m : -1;
                if (!wave_live) continue;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int nl = (pwid * TN + j) * 16 + nl4, n = n0 + nl;
                    float v[4];
                    v[0] = x_actf(__builtin_fmaf(acc[i][j][0], scu[j].x, bs[j].x), a.slope, a.cap);
                    v[1] = x_actf(__builtin_fmaf(acc[i][j][1], scu[j].y, bs[j].y), a.slope, a.cap);
                    v[2] = x_actf(__builtin_fmaf(acc[i][j][2], scu[j].z, bs[j].z), a.slope, a.cap);
                    v[3] = x_actf(__builtin_fmaf(acc[i][j][3], scu[j].w, bs[j].w), a.slope, a.cap);
                    if (a.res.p && mok && (n >> 3) < a.res.G) {
                        const uint32_t q = (uint32_t)((m * a.res.G + (n >> 3)) * 32 + (n & 7) * 2), q2 = q + 16u;
                        const half4 rh = __builtin_bit_cast(half4, __builtin_amdgcn_raw_buffer_load_b64(rsr, q, 0, 0));
                        const half4 rl = __builtin_bit_cast(half4, __builtin_amdgcn_raw_buffer_load_b64(rsr, q2, 0, 0));
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] += ((float)rh[k] + (float)rl[k]) * rup;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (mok) rmax = fmaxf(rmax, fabsf(v[k]));
                    if (a.dst_f32) {
                        *reinterpret_cast<u32x4 *>(Cs + (p - i0 * 16) * C::CPITCH + (nl >> 3) * 32 + ((nl >> 2) & 1) * 16) =
                            u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                    } else {
                        half4 hi, lo;
                        float vd[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) vd[k] = v[k] * dout;
                        x_split4(vd, hi, lo);
                        unsigned char *d = Cs + (p - i0 * 16) * C::CPITCH + (nl >> 3) * 32 + (nl & 7) * 2;
                        *reinterpret_cast<half4 *>(d) = hi;
                        *reinterpret_cast<half4 *>(d + 16) = lo;
                    }
                }
            }
        }
        const bool last = i0 + IPP >= TM;
        if (last) x_amax_lds(smax, 0, rmax);
        __syncthreads();
        if (last && tid == 0 && smax[0]) x_amax_global(a.amax_out + (size_t)b * XS, smax[0]);
        const int rows = (TM - i0 < IPP ? TM - i0 : IPP) * 16;
        for (int v = tid; v < rows * VPR; v += 512) {
            const int pr = v / VPR, cv = v - pr * VPR;
            const int m = *reinterpret_cast<const int *>(Cs + pr * C::CPITCH + BN * 4), g = (n0 >> 3) + (cv >> 1);
            const uint32_t so = (m >= 0 && g < a.outG) ? (uint32_t)((m * a.outG + g) * 32 + (cv & 1) * 16) : X_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(Cs + pr * C::CPITCH + cv * 16), rss, so, 0, 0);
        }
    }
}
