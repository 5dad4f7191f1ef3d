// yk_xblock.h — f16x2 mode: DepthwiseConv2D 3x3 + BN + act -> Conv2D 1x1 + BN + act (+ residual Add) in ONE launch (included by
// yk_exact.hip).  MobileNet block: keras_mobilenet.py:359-436, keras_mobilenet_v2.py:452-481.
//
// A workgroup (4 waves) owns a TH x TW patch of output pixels of ONE image (BM = 16*TM >= TH*TW GEMM rows) and BN = 64*TN output
// channels (wave w: channels [w*16*TN, (w+1)*16*TN) of all BM pixels).  The channel axis is walked in steps of 32 (four groups of 8):
//
//   DMA(k)    one burst of `buffer_load ... lds`: the input patch of the step's four channel groups ((TH-1)s+3 x (TW-1)s+3 positions
//             x 4 groups, hi and lo halves in separate regions; pixels outside the image and groups past the tensor's last one get an
//             out-of-range offset and arrive as zeros) and the step's slice of the depthwise parameter table (nine taps, scale, bias x
//             32 channels fp32)
//   W(k)      the pointwise weight fragments of the step, global memory -> REGISTERS of the wave that multiplies them (wave w owns
//             output channels [w*16*TN, (w+1)*16*TN): a weight tile in LDS would be written once and read once by one wave; host order =
//             fragment order, one coalesced 1 KB load per 16-channel block and half).  Round 5: this took 6 - 24 KB per workgroup out of
//             LDS - a fourth workgroup per CU, and N = 384 in one workgroup
//   dw(k)     one thread = one (pixel, channel group): nine taps from LDS; hi + lo back in fp32 by one mixed-precision op per channel,
//             packed fp32 FMAs on channel pairs, BN + activation, scaled by the MIDDLE exponent and split pair-wise into (hi, lo)
//             straight into the MFMA A tile - the depthwise tensor never exists in HBM
//   mma(k)    A x B on v_mfma_f32_16x16x32_f16, three products per tile
//
// Single-buffered and phase-shifted (two s_barriers per step): the patch of step k+1 is requested when dw(k) is over and lands under
// mma(k); the weight fragments of step k are requested before dw(k).  One stage of patch + A tile is 23 - 39 KB and the small tiles are
// held to 128 registers, so FOUR workgroups share a CU and their DMA, depthwise (VALU) and MFMA phases overlap - co-residency is what
// these launches' time follows (round 5: 3 -> 4 workgroups = -8 ... -15 %; two stages, which halve it, never paid; removing the depthwise
// pass's LDS reads in a pricing build changes nothing).  The output tile leaves through LDS in passes of IPP row blocks, or - fp32
// outputs - straight from the registers through a per-image buffer descriptor.  The depthwise tensor's maximum is never measured, so its
// exponent comes from its bound (gain_dw * amax(in) + off_dw) and the pointwise bound is built on that bound; the two levels of
// over-estimate (2^3 x 2^7 in these networks) stay far inside fp16's exponent range (see the header of yk_exact.hip).  STEM: the
// depthwise input is the network's first conv, computed in the kernel from the frame window (xb_stem_patch); its patch and A tile hold
// the GL channel groups the stem has (three for 24 filters).
#pragma once

struct xb_args {
    xview in, res;                     // res.p null without a residual
    int B, Ho, Wo, N;
    int stride, pad_t, pad_l;
    const float *par;                  // [11][Cp] fp32: nine depthwise taps, scale, bias
    int nk;                            // k-steps = ceil(G / 4)
    float dw_slope, dw_cap, dw_gain, dw_off;
    const uint8_t *w;                  // pointwise weights [nk][nslab][2][16][32] halfs
    uint32_t w_bytes;
    int nslab;
    const float *scale, *bias;         // pointwise BN (scale carries 2^-s of the weight split)
    float slope, cap, gain, off;       // |pointwise output| <= gain * bound(dw) + off
    uint8_t *out;
    int outG;
    int *eexp_out;
    uint32_t *amax_out;
    // geometry, fixed at plan creation
    int TH, TW, PH, PW, tiles_x, tiles_y, n16, n16p;
    int db;                            // 1: two stages of (patch, parameters, weight tile), DMA(k+1) requested at the start of step k
    int prepass;                       // stride-1 block on a stored tensor: the patch is converted to fp32 ONCE, in place, before the taps
    // A tensor whose only consumer is the depthwise conv of another fused block is stored as FP32 ([pixel][group][ch 0-3 | ch 4-7], the same
    // 32 bytes per group as (hi | lo), exponent 0): its producer skips the split, its consumer's taps skip 72 conversions per item
    int src_f32, dst_f32;
    int lds_bytes;                     // total dynamic LDS
    yk_fastdiv fd_tpi, fd_tx, fd_tw, fd_pw, fd_nk;
    // fused stem: the block's depthwise input is the output of the network's FIRST conv (3 input channels, <= 32 filters), computed in
    // this kernel from the frames; that tensor (55 MB per batch of 32 at 224x320) is then never written nor read
    int stem;                          // 0: `in` is a stored tensor
    const void *frames;                // u8 or f32 [B][fH][fW][3] (set per run)
    const unsigned *img_max;           // YK_MAXP partial maxima per image (u8 path)
    int in_f32, fH, fW;
    int st_stride, st_pad_t, st_pad_l, st_cout;
    const yk_half *st_wf;              // [2 n-blocks][hi|lo][64 lanes][8]: w * 2^s in MFMA fragment order, k = ky*8+j | 24+ky
    const float *st_scale, *st_bias;   // [32], zero padded; scale carries 2^-s
    float st_slope, st_cap, st_bound;  // |stem output| <= st_bound (the normalised image is in [0, 1]); st_e its exponent
    int st_e;
    yk_fastdiv fd_wrow;                // division by the window row length (WC * 3)
    yk_fastdiv fd_dpr;                 // division by the window row length in dwords (u8 frames: the window stays bytes in LDS)
    // fused stem: the patch and the A tile hold the GL channel groups that EXIST (24 stem filters: three groups of 8, 48 bytes per position /
    // A row per plane instead of 64): 39 KB instead of 49 - a FOURTH workgroup per CU - and the depthwise pass has no dead items
    int GL;
    int dbg;
    long long *stamps;                 // developer builds: per-workgroup phase timestamps [wg][16] (wall_clock64), or null
};

template <int TM, int TN>
struct xb_cfg {
    static constexpr int BM = 16 * TM, BN = 64 * TN;
    static constexpr int PARB = 2048;                             // 11 x 32 floats = 88 DMA slots of 16 B, deposited by two whole waves (128 slots)
    static constexpr int CPITCH = BN * 4 + 16;
    static constexpr int IPP = (TN >= 3 && TM >= 2) ? (TM + 1) / 2 : TM;   // row blocks per output pass
    static constexpr int stage(int n16p) { return n16p * 32 + PARB; }
    static constexpr int lds(int n16p, int db) {
        const int r = (db ? 2 : 1) * stage(n16p) + BM * 128, ct = IPP * 16 * CPITCH;
        return (r > ct ? r : ct) + 64;
    }
};

// The stem conv on the positions of a depthwise patch, on the matrix cores.  K = 27 is padded to 32 in an order chosen for the loader:
// k = ky*8 + j for the first 8 of the 9 contiguous window values (3 pixels x RGB) a filter row contributes, k = 24 + ky for the ninth,
// 27..31 zero (`st_wf` holds the weights in that order, split hi | lo, as MFMA fragments).  A lane of the pixel operand (position
// fr, chunk fq) reads 8 consecutive floats of window row fq (fq < 3) or the three ninth values (fq = 3), splits them and runs three
// MFMAs per 16 output channels.  Positions outside the stem's output are the depthwise conv's zero padding, not convolutions of padded
// pixels.  Result: (hi | lo) in the patch layout [position][4 groups of 8 channels].
// u8 frames (U8): the window is BYTES in LDS (row pitch `pitch`, a multiple of 4); a lane fetches the three aligned dwords around its
// 8 bytes and shifts them into place (v_alignbyte).  A byte is exact in fp16, so the pixel operand has no low half and the division of
// `img / np.max(img)` (tools/utils.py:405) moves behind the sum: conv(img) * (scale / max) in fp32 - two MFMAs per tile instead of
// three, no per-pixel split.  f32 frames: the window is fp32, already normalised by the caller; the operand is split as everywhere.
// The storage exponent of the stem's output is a plan constant (the image is in [0, 1]), folded into scale, bias and cap.
template <bool U8, int GL>
__device__ __forceinline__ void xb_stem_patch(const xb_args &a, const void *winp, int WC, int pitch, float inv, int iy0, int ix0, unsigned char *HI, unsigned char *LO) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
    half8 wfh[2], wfl[2];
    float scv[2][4], bsv[2][4];
    const float sdown = x_pow2(-a.st_e), sfac = U8 ? sdown / inv : sdown, cap = a.st_cap * sdown;
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
        wfh[nf] = *reinterpret_cast<const half8 *>(a.st_wf + ((size_t)(nf * 2 + 0) * 64 + lane) * 8);
        wfl[nf] = *reinterpret_cast<const half8 *>(a.st_wf + ((size_t)(nf * 2 + 1) * 64 + lane) * 8);
        const float4 sc = *reinterpret_cast<const float4 *>(a.st_scale + nf * 16 + fq * 4);
        const float4 bs = *reinterpret_cast<const float4 *>(a.st_bias + nf * 16 + fq * 4);
        scv[nf][0] = sc.x * sfac, scv[nf][1] = sc.y * sfac, scv[nf][2] = sc.z * sfac, scv[nf][3] = sc.w * sfac;
        bsv[nf][0] = bs.x * sdown, bsv[nf][1] = bs.y * sdown, bsv[nf][2] = bs.z * sdown, bsv[nf][3] = bs.w * sdown;
    }
    const int st = a.st_stride, npos = a.PH * a.PW, nblk = (npos + 15) >> 4, rowf = WC * 3;
    for (int blk = wid; blk < nblk; blk += 4) {
        const int pos = blk * 16 + fr, pc = min(pos, npos - 1);
        const int r = (int)x_div((uint32_t)pc, a.fd_pw), c = pc - r * a.PW;
        const bool valid = pos < npos;
        const bool inside = valid && (unsigned)(iy0 + r) < (unsigned)a.in.H && (unsigned)(ix0 + c) < (unsigned)a.in.W;
        half8 xh, xl;
        if constexpr (U8) {
            const unsigned char *wb = reinterpret_cast<const unsigned char *>(winp);
            const int off = (r * st) * pitch + c * st * 3;
            uint32_t d0, d1 = 0u;
            if (fq < 3) {
                const int o = off + fq * pitch, sh = o & 3;
                const uint32_t *q = reinterpret_cast<const uint32_t *>(wb + (o & ~3));
                const uint32_t w0 = q[0], w1 = q[1], w2 = q[2];
                d0 = __builtin_amdgcn_alignbyte(w1, w0, sh);
                d1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
            } else {
                d0 = (uint32_t)wb[off + 8] | ((uint32_t)wb[off + pitch + 8] << 8) | ((uint32_t)wb[off + 2 * pitch + 8] << 16);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xh[j] = (yk_half)(float)((d0 >> (8 * j)) & 255u);
                xh[4 + j] = (yk_half)(float)((d1 >> (8 * j)) & 255u);
            }
        } else {
            const float *base = reinterpret_cast<const float *>(winp) + ((r * st) * WC + c * st) * 3;
            float x[8];
            if (fq < 3) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = base[fq * rowf + j];
            } else {
                x[0] = base[8];
                x[1] = base[rowf + 8];
                x[2] = base[2 * rowf + 8];
#pragma unroll
                for (int j = 3; j < 8; ++j) x[j] = 0.f;
            }
            x_split8(x, xh, xl);
        }
        floatx4 acc[2];
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfl[nf], xh, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if constexpr (!U8) {
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfh[nf], xl, acc[nf], 0, 0, 0);
        }
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) acc[nf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfh[nf], xh, acc[nf], 0, 0, 0);
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float u = __builtin_fmaf(acc[nf][k], scv[nf][k], bsv[nf][k]);
                v[k] = inside ? x_actf(u, a.st_slope, cap) : 0.f;
            }
            // the patch of a fused stem is fp32: [position][4 groups][channels 0-3] in the first plane, [..][channels 4-7] in the second (what
            // the depthwise taps multiply; round 3 stored (hi | lo) here and every tap converted them back, 72 VALU operations per item)
            const int n = nf * 16 + fq * 4;
            if (valid && (n >> 3) < GL) {
                const int at = (pos * GL + (n >> 3)) * 16;
                *reinterpret_cast<u32x4 *>(((n & 4) ? LO : HI) + at) = u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
            }
        }
    }
}

// SG: 0 = the input is a stored tensor; 2, 3, 4 = fused stem with SG channel groups.  F32IN: the fused stem reads fp32 frames (else u8) - a
// compile-time choice: with both window loaders in one kernel the stem block kept ~190 scalar registers in vector-register lanes
template <int TM, int TN, int SG, bool F32IN = false>
__global__ void __launch_bounds__(256, (TM * TN > 8 ? 2 : 4)) xb_kernel(const xb_args a) {   // 2 (3) workgroups per CU: <= 256 (168) registers; the fused-stem block stays under 128 by itself (4 per CU)
    typedef xb_cfg<TM, TN> C;
    constexpr int BM = C::BM, BN = C::BN;
    constexpr bool STEM = SG > 0;
    constexpr int GL = STEM ? SG : 4;                                 // channel groups per patch position and per A row
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int G = a.in.G, s = a.stride;
    // two stages and the in-place fp32 prepass exist for the developer sweeps only (both measured slower everywhere): constants in the shipped
    // kernels, whose code and scalar registers they otherwise cost
#ifdef YK_DEV
    const int XB_DB = a.db, XB_PREPASS = a.prepass;
#else
    constexpr int XB_DB = 0, XB_PREPASS = 0;
#endif
#ifdef YK_DEV
#define XB_STAMP(k) \
    if (a.stamps && tid == 0) a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = (long long)wall_clock64();
#else
#define XB_STAMP(k)
#endif
    XB_STAMP(0)
    const int STG = a.n16p * 32 + C::PARB;
    unsigned char *A = xsm + (XB_DB ? 2 : 1) * STG;
    float *sf = reinterpret_cast<float *>(xsm + a.lds_bytes - 64);   // [0] 2^e_in [1] 2^-e_mid [2] 2^e_mid [3] 2^-e_out [4] 2^e_res
    uint32_t *smax = reinterpret_cast<uint32_t *>(sf + 8);
    // block -> (image, tile); blockIdx.y = N slice
    const int bid = x_xcd_tile(blockIdx.x, gridDim.x);
    const uint32_t b = x_div((uint32_t)bid, a.fd_tpi), tl = bid - b * (a.tiles_x * a.tiles_y);
    const uint32_t ty = x_div(tl, a.fd_tx), tx = tl - ty * a.tiles_x;
    const int oy0 = (int)ty * a.TH, ox0 = (int)tx * a.TW;
    const int iy0 = oy0 * s - a.pad_t, ix0 = ox0 * s - a.pad_l;
    const int n0 = blockIdx.y * BN;
    const int fr = lane & 15, fq = lane >> 4, nl4 = fq * 4;
    // BatchNorm scale / bias of this lane's output channels: requested first, used last (narrow tiles only: 12 float4 pairs of a
    // 384-wide tile would cost the second workgroup per CU its registers)
    constexpr bool EARLY_SB = TN <= 1;
    float4 sc[TN], bs[TN];
    if constexpr (EARLY_SB) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wid * TN + j) * 16 + nl4;
            sc[j] = *reinterpret_cast<const float4 *>(a.scale + n);
            bs[j] = *reinterpret_cast<const float4 *>(a.bias + n);
        }
    }
    const uint32_t img = (uint32_t)a.in.H * a.in.W * G * 32u;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void *)(STEM ? (const uint8_t *)a.par : a.in.p + (size_t)b * img), 0, STEM ? 16u : img, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc((void *)a.par, 0, (uint32_t)(11 * G * 32), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);
    // this lane's patch positions are the same for every step: precompute the source offsets of its (up to PQ) DMA slots
    constexpr int PQ = 6;                                         // patch slots per lane: n16p <= PQ * 256
    uint32_t poff[PQ];
#pragma unroll
    for (int i = 0; i < PQ; ++i) {
        poff[i] = X_OOB;
        if (!STEM && i * 256 < a.n16p) {                              // (uniform: a small patch skips the slots it does not have)
            const uint32_t q = (uint32_t)(i * 256 + tid);
            const uint32_t pos = q >> 2, g4 = q & 3;
            const uint32_t r = x_div(pos, a.fd_pw), c = pos - r * a.PW;
            const int iy = iy0 + (int)r, ix = ix0 + (int)c;
            const bool ok = (int)q < a.n16 && (unsigned)iy < (unsigned)a.in.H && (unsigned)ix < (unsigned)a.in.W;
            poff[i] = ok ? (uint32_t)(((iy * a.in.W + ix) * G + (int)g4) * 32) : X_OOB;
        }
    }
    const int g4l = tid & 3;
    // Workgroups walk the channel steps from different starting points (a function of the tile's place in ITS image only, so an
    // image's arithmetic does not depend on the batch): 256 CUs asking one L2 for the same weight tile in the same microsecond
    // serialise on its banks
    const int rot = (a.nk == 1 || X_DBG(a, 32)) ? 0 : (int)(tl - x_div(tl, a.fd_nk) * (uint32_t)a.nk);
    auto kstep = [&](int i) {
        const int k = i + rot;
        return k >= a.nk ? k - a.nk : k;
    };
    const uint32_t wstep = (uint32_t)a.nslab * 2048u;
    auto dma_patch = [&](int it_) {                               // patch + depthwise parameters of loop step it_
        const int ks = kstep(it_);
        unsigned char *HI = xsm + ((XB_DB & it_) & 1) * STG, *LO = HI + a.n16p * 16, *PARb = HI + a.n16p * 32;
        const bool gok = (ks * 4 + g4l) < G;
        const uint32_t koff = gok ? (uint32_t)ks * 128u : X_OOB;
#pragma unroll
        for (int i = 0; i < PQ; ++i)
            if (!STEM && i * 256 + wid * 64 < a.n16p) {                    // wave-uniform: n16p is a multiple of 64, a wave deposits 64 slots
                const uint32_t oh = poff[i] + koff, ol = oh + 16u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr_t)(HI + (i * 256 + wid * 64) * 16), 16, oh, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr_t)(LO + (i * 256 + wid * 64) * 16), 16, ol, 0, 0, 0);
            }
        if (wid < 2) {                                            // 11 rows x 8 pieces of 16 B = 88 slots
            const int idx = wid * 64 + lane, t = idx >> 3, pc = idx & 7;
            const int ch = ks * 32 + pc * 4;
            const uint32_t op = (idx < 88 && ch < G * 8) ? (uint32_t)((t * G * 8 + ch) * 4) : X_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsp, (lds_ptr_t)(PARb + wid * 1024), 16, op, 0, 0, 0);
        }
    };
    if (!X_DBG(a, 1)) dma_patch(0);
    if constexpr (STEM) {
        // ---- the patch of the depthwise input = the stem conv's output on (PH x PW) positions, straight from the frames.
        // (1) the frame window those positions need, normalised (`img / np.max(img)`, tools/utils.py:405), as fp32 in LDS (the A tile's
        //     space: nothing else lives there before the first depthwise pass)
        const int st = a.st_stride, WR = (a.PH - 1) * st + 3, WC = (a.PW - 1) * st + 3;
        const int wy0 = iy0 * st - a.st_pad_t, wx0 = ix0 * st - a.st_pad_l;
        float *win = reinterpret_cast<float *>(A);
        float inv = 1.f;
        if (!F32IN) {
            unsigned mx = 0;
#pragma unroll
            for (int j = 0; j < 32; ++j) mx = max(mx, a.img_max[b * 32 + j]);
            inv = (float)mx;
        }
        // all of a thread's loads go out before the first one is used, and the u8 / f32 choice is made ONCE outside the unrolled loads
        // (a load inside a per-element branch makes the compiler drain the memory queue at every join).  u8 frames are fetched four
        // bytes per lane (unaligned dwords through a buffer descriptor: bytes before the first frame come back as zeros).
        const int rowf_ = WC * 3;
        if constexpr (F32IN) {
            constexpr int WQ = 16;                                    // window floats per thread: WR*WC*3 <= 16*TM*32 <= WQ*256
            const float *f = reinterpret_cast<const float *>(a.frames);
            const size_t fimg = (size_t)b * a.fH * a.fW * 3;
            float wv[WQ];
            bool wok[WQ];
#pragma unroll
            for (int q = 0; q < WQ; ++q) {
                const int i = q * 256 + tid;
                const int r = (int)x_div((uint32_t)i, a.fd_wrow), rem = i - r * rowf_, c = rem / 3, ch = rem - c * 3;
                const int fy = wy0 + r, fx = wx0 + c;
                wok[q] = i < WR * rowf_ && (unsigned)fy < (unsigned)a.fH && (unsigned)fx < (unsigned)a.fW;
                wv[q] = f[wok[q] ? fimg + ((size_t)fy * a.fW + fx) * 3 + ch : fimg];
            }
#pragma unroll
            for (int q = 0; q < WQ; ++q) {
                const int i = q * 256 + tid;
                if (i < WR * rowf_) win[i] = wok[q] ? wv[q] : 0.f;
            }
        } else {
            constexpr int DQ = 5;                                     // dwords per thread: WR * ceil(WC*3/4) <= DQ*256
            const int dpr = (rowf_ + 3) >> 2;
            const uint32_t fbytes = (uint32_t)a.B * a.fH * a.fW * 3u;
            const __amdgpu_buffer_rsrc_t rsf = __builtin_amdgcn_make_buffer_rsrc((void *)a.frames, 0, fbytes, 0x00020000);
            const int rowb = a.fW * 3;
            u32x2 d2[DQ];
            int sh[DQ];
            uint32_t msk[DQ];
#pragma unroll
            for (int q = 0; q < DQ; ++q) {
                const int i = q * 256 + tid, r = (int)x_div((uint32_t)i, a.fd_dpr), d = i - r * dpr;
                const int fy = wy0 + r;
                const bool ok = r < WR && (unsigned)fy < (unsigned)a.fH;
                // four bytes from a BYTE address (buffer loads ignore the low address bits): the aligned 8 bytes around it, shifted.  The
                // window may start before / run past its row (masked per byte); an address below the buffer start (first row of the
                // first frame, left padding) is read from 0 and shifted the other way
                const int col = wx0 * 3 + d * 4;                          // byte column of this dword's first byte in the frame row
                const int off = ((int)b * a.fH + fy) * rowb + col;
                sh[q] = off >= 0 ? (off & 3) * 8 : off * 8;
                d2[q] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsf, ok ? (uint32_t)(max(off, 0) & ~3) : X_OOB, 0, 0));
                const int lo = max(0, -col), hi = min(4, rowb - col);      // bytes [lo, hi) of the dword lie inside the frame row
                msk[q] = (ok && hi > lo) ? ((0xffffffffu << (8 * lo)) & (0xffffffffu >> (8 * (4 - hi)))) : 0u;
            }
#pragma unroll
            for (int q = 0; q < DQ; ++q) {
                const int i = q * 256 + tid;
                const unsigned long long v = ((unsigned long long)d2[q][1] << 32) | d2[q][0];
                const uint32_t dv = sh[q] >= 0 ? (uint32_t)(v >> sh[q]) : (uint32_t)(v << (-sh[q]));
                if (i < WR * dpr) reinterpret_cast<uint32_t *>(A)[i] = dv & msk[q];
            }
            // the patch builder reads whole dwords: up to 11 bytes past a row's last value -> two dwords past the window's end
            if (tid < 2) reinterpret_cast<uint32_t *>(A)[WR * dpr + tid] = 0u;
        }
        XB_STAMP(11)
        __syncthreads();
        XB_STAMP(12)
        // (2) one thread = one patch position, all st_cout channels
        unsigned char *HI = xsm, *LO = xsm + a.n16p * 16;
        if constexpr (F32IN) xb_stem_patch<false, GL>(a, A, WC, 0, 1.f, iy0, ix0, HI, LO);
        else xb_stem_patch<true, GL>(a, A, WC, ((WC * 3 + 3) >> 2) * 4, inv, iy0, ix0, HI, LO);
    }
    XB_STAMP(1)
    // per-image factors (one image per workgroup)
    if (wid == 0) {
        const float amax_in = STEM ? a.st_bound : x_amax_wave(a.in.amax, (int)b);
        const float bmid = fminf(a.dw_cap, a.dw_gain * amax_in + a.dw_off);
        float bout = fminf(a.cap, a.gain * bmid + a.off);
        float rup = 0.f;
        if (a.res.p) {
            bout += x_amax_wave(a.res.amax, (int)b);
            rup = x_pow2(a.res.eexp[b]);
        }
        const int em = x_exp_of(__float_as_uint(bmid)), eo = a.dst_f32 ? 0 : x_exp_of(__float_as_uint(bout));
        if (lane == 0) {
            sf[0] = x_pow2(STEM ? a.st_e : a.in.eexp[b]);
            sf[1] = x_pow2(-em);
            sf[2] = x_pow2(em);
            sf[3] = x_pow2(-eo);
            sf[4] = rup;
            smax[0] = 0u;
            a.eexp_out[b] = eo;
        }
    }
    XB_STAMP(2)
    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    const bool wave_live = n0 + wid * TN * 16 < a.N;                  // wave-uniform
    // fragment offsets: rows of 64 bytes with the 16-byte pieces swizzled (weight tile; A tile of a stored-tensor block), rows of GL * 16
    // bytes in a fused-stem block (48-byte rows of a 24-channel stem touch every bank once per 16 lanes without a swizzle)
    const int foff = fr * 64 + ((fq ^ ((fr >> 1) & 3)) * 16);
    // (a k group the stem does not have: the lane re-reads group 0 - finite values against weight rows that are zero)
    const int afoff = fr * (GL * 16) + (GL == 4 ? (fq ^ ((fr >> 1) & 3)) : (fq < GL ? fq : 0)) * 16;
    const int nk = X_DBG(a, 1) ? 0 : a.nk;
    for (int ks = 0; ks < nk; ++ks) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's pieces of patch(ks) have landed
        __builtin_amdgcn_s_barrier();                                 // everybody's have; mma(ks-1) is over: A and the weight tile are free
        asm volatile("" ::: "memory");
        if (ks == 0) { XB_STAMP(3) }
        if (ks == 1) { XB_STAMP(13) }
        // The pointwise weights of this step.  Wave w multiplies ITS 16 * TN output channels and nobody else's, so a weight tile in LDS
        // would be written once and read once by one wave: the fragments go from global memory straight into that wave's registers
        // (host order = fragment order: one coalesced 1 KB load per 16-channel block and half), requested here, used after the depthwise
        // pass.  No LDS stage for them: 6 - 24 KB less per workgroup, which is what lets a fourth workgroup onto a CU.
        half8 wh[TN], wl[TN];
        if (wave_live && !X_DBG(a, 16)) {
            const uint32_t ws = (uint32_t)(n0 >> 4) * 2048u + (uint32_t)kstep(ks) * wstep + (uint32_t)(wid * TN) * 2048u + (uint32_t)foff;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                wh[j] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsw, ws + (uint32_t)j * 2048u, 0, 0));
                wl[j] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsw, ws + (uint32_t)j * 2048u + 1024u, 0, 0));
            }
        }
        // two stages: the other one is free - request the patch of step ks+1 now (AFTER the weight loads: the memory counter retires in
        // order, and the weights are needed first)
        if (XB_DB && ks + 1 < nk) dma_patch(ks + 1);
        const unsigned char *HI = xsm + ((XB_DB & ks) & 1) * STG, *LO = HI + a.n16p * 16;
        // the parameter slice is read as DWORDS, like the patch: a float-typed LDS read makes the compiler wait for every LDS-DMA in
        // flight (s_waitcnt vmcnt(0): the weight tile requested a moment ago) before it, a dword-typed one does not
        const unsigned char *PARB_ = HI + a.n16p * 32;
        const float up = sf[0], dmid = sf[1];
        const bool f32patch = STEM || XB_PREPASS || a.src_f32;
        if (!STEM && XB_PREPASS && !a.src_f32) {
            // stride 1: a patch element feeds nine taps.  (hi, lo) -> fp32 once, in place (channels 0-3 over the hi plane's 16 bytes,
            // 4-7 over the lo plane's), instead of in every tap: 8 conversions per element here for 72 per item there
            for (int e = tid; e < a.n16p; e += 256) {
                const u32x4 h = *reinterpret_cast<const u32x4 *>(HI + e * 16), l = *reinterpret_cast<const u32x4 *>(LO + e * 16);
                u32x4 x0, x1;
                x0[0] = __float_as_uint(x_mix_sum_lo(h[0], l[0])); x0[1] = __float_as_uint(x_mix_sum_hi(h[0], l[0]));
                x0[2] = __float_as_uint(x_mix_sum_lo(h[1], l[1])); x0[3] = __float_as_uint(x_mix_sum_hi(h[1], l[1]));
                x1[0] = __float_as_uint(x_mix_sum_lo(h[2], l[2])); x1[1] = __float_as_uint(x_mix_sum_hi(h[2], l[2]));
                x1[2] = __float_as_uint(x_mix_sum_lo(h[3], l[3])); x1[3] = __float_as_uint(x_mix_sum_hi(h[3], l[3]));
                *reinterpret_cast<u32x4 *>(const_cast<unsigned char *>(HI) + e * 16) = x0;
                *reinterpret_cast<u32x4 *>(const_cast<unsigned char *>(LO) + e * 16) = x1;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        // ---- depthwise: item = (pixel p, group q of this step)
        if (!X_DBG(a, 2))
            for (int it = tid; it < BM * GL; it += 256) {
                int p, q;
                if constexpr (GL != 4) {
                    p = (int)((uint32_t)it / (uint32_t)GL);
                    q = it - p * GL;
                } else {
                    p = it >> 2;
                    q = it & 3;
                }
                const int py = (int)x_div((uint32_t)p, a.fd_tw), px = p - py * a.TW;
                const bool live = py < a.TH && oy0 + py < a.Ho && ox0 + px < a.Wo;
                half8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = hi;
                if (live) {
                    const int base = ((py * s) * a.PW + px * s) * GL + q;
                    // x = hi + lo back in fp32 by one mixed-precision op per channel, then packed fp32 FMAs (two channels per
                    // instruction): 12 VALU ops per tap and channel group instead of 16
                    float2v d2[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
#if defined(YK_KO) && (YK_KO & 2)                                     // pricing build: three patch reads (one per row) instead of nine
                        const int at = (base + ((t / 3) * a.PW) * GL) * 16;
#else
                        const int at = (base + ((t / 3) * a.PW + (t % 3)) * GL) * 16;
#endif
                        const u32x4 h = *reinterpret_cast<const u32x4 *>(HI + at), l = *reinterpret_cast<const u32x4 *>(LO + at);
#if defined(YK_KO) && (YK_KO & 1)                                     // pricing build (wrong results on purpose): no weight reads
                        const u32x4 w0 = {0x3f000000u, 0x3e800000u, 0x3f000000u, 0x3e800000u}, w1 = w0;
#else
                        const u32x4 w0 = *reinterpret_cast<const u32x4 *>(PARB_ + (t * 32 + q * 8) * 4), w1 = *reinterpret_cast<const u32x4 *>(PARB_ + (t * 32 + q * 8 + 4) * 4);
#endif
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
                unsigned char *dst = A + (p >> 4) * (GL * 512) + r * (GL * 16) + ((q ^ (GL == 4 ? (r >> 1) & 3 : 0)) * 16);
                *reinterpret_cast<half8 *>(dst) = hi;
                *reinterpret_cast<half8 *>(dst + GL * 256) = lo;
            }
        if (ks == 0) { XB_STAMP(4) }
        if (ks == 1) { XB_STAMP(14) }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // (the weight fragments - and, with two stages, step ks+1 - stay in flight)
        __builtin_amdgcn_s_barrier();                                 // the A tile is complete (single stage: the patch is free)
        asm volatile("" ::: "memory");
        if (ks == 0) { XB_STAMP(5) }
        if (ks == 1) { XB_STAMP(15) }
        if (!XB_DB && ks + 1 < nk) dma_patch(ks + 1);
        // ---- pointwise: three products per tile (a wave whose 16*TN channels all lie past N - the last quarter of a 48- or 96-channel
        // layer in a 64- / 128-wide tile - has nothing to multiply nor, below, to stage)
        if (wave_live && !X_DBG(a, 16)) {
            half8 xh[TM], xl[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                xh[i] = *reinterpret_cast<const half8 *>(A + i * (GL * 512) + afoff);
                xl[i] = *reinterpret_cast<const half8 *>(A + i * (GL * 512) + GL * 256 + afoff);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)                        // (a fused-stem block has ONE step: its first product starts from the zero constant)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], xh[i], STEM ? floatx4{0.f, 0.f, 0.f, 0.f} : acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xl[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xh[i], acc[i][j], 0, 0, 0);
        }
    }
    XB_STAMP(6)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                     // LDS becomes the output tile
    asm volatile("" ::: "memory");
    XB_STAMP(7)
    if (X_DBG(a, 4)) return;
    // ---- epilogue: lane holds channels n..n+3 of pixel i*16 + fr; the tile leaves through LDS in passes of IPP row blocks
    unsigned char *Cs = xsm;
    const float umid = sf[2], dout = sf[3], rup = sf[4];
    float rmax = 0.f;
    constexpr int VPR = BN / 4, IPP = C::IPP;
    if constexpr (!EARLY_SB) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wid * TN + j) * 16 + nl4;
            sc[j] = *reinterpret_cast<const float4 *>(a.scale + n);
            bs[j] = *reinterpret_cast<const float4 *>(a.bias + n);
        }
    }
    float4 scu[TN];                                                   // BN scale with the middle exponent folded in (a power of two: exact)
#pragma unroll
    for (int j = 0; j < TN; ++j) scu[j] = float4{sc[j].x * umid, sc[j].y * umid, sc[j].z * umid, sc[j].w * umid};
    if (a.dst_f32 && !a.res.p) {
        // fp32 output (its only reader is another fused block's depthwise conv): a lane's four channels are 16 contiguous bytes of the
        // tensor and the four lanes of a pixel cover 64 - stored straight from the registers.  (The (hi | lo) layout leaves 8-byte
        // pieces per lane, which is why that form is staged through LDS below; here the staging would be 33 KB written + 33 KB read on
        // the launch's busiest unit, a barrier and a copy-out loop for nothing.)
        if (wave_live) {
            // one descriptor per image (scalar), 32-bit offsets inside it: the stores need no 64-bit address arithmetic per lane
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
                    const int n = n0 + (wid * TN + j) * 16 + nl4;
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
        XB_STAMP(9)
        XB_STAMP(10)
        return;
    }
    // (staged form) one descriptor per image for the output and for the residual: a pixel is a 32-bit offset inside its image
    const uint32_t simg = (uint32_t)a.Ho * a.Wo * a.outG * 32u, rimg = (uint32_t)a.Ho * a.Wo * a.res.G * 32u;
    const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc((void *)(a.out + (size_t)b * simg), 0, simg, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc((void *)(a.res.p ? a.res.p + (size_t)b * rimg : a.out), 0, a.res.p ? rimg : 0u, 0x00020000);
#pragma unroll
    for (int i0 = 0; i0 < TM; i0 += IPP) {
        if (i0 > 0) __syncthreads();                                  // the previous pass has been copied out
#pragma unroll
        for (int i = i0; i < i0 + IPP && i < TM; ++i) {
            if (!wave_live && wid != 0) break;                        // (wave 0 also writes the rows' pixel indices)
            const int p = i * 16 + fr;
            const int py = (int)x_div((uint32_t)p, a.fd_tw), px = p - py * a.TW;
            const int oy = oy0 + py, ox = ox0 + px;
            const bool mok = py < a.TH && oy < a.Ho && ox < a.Wo;
            const int m = oy * a.Wo + ox;                             // the pixel inside its image
            // the row's pixel index (or -1) rides in the 16 pad bytes of its staging row: the copy-out below needs no division per vector
            if (wid == 0 && fq == 0) *reinterpret_cast<int *>(Cs + (p - i0 * 16) * C::CPITCH + BN * 4) = mok ? m : -1;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nl = (wid * TN + j) * 16 + nl4, n = n0 + nl;
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
                if (a.dst_f32) {                                      // fp32 planes of the group: channels 0-3 | 4-7 (exponent 0: no scaling, no split)
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
        if (i0 == 0) { XB_STAMP(8) }
        const bool last = i0 + IPP >= TM;                             // (compile-time: the loop is unrolled)
        if (last) x_amax_lds(smax, 0, rmax);                          // the tile's max rides on the barrier the staging needs anyway
        __syncthreads();
        if (last && tid == 0 && smax[0]) x_amax_global(a.amax_out + (size_t)b * XS, smax[0]);
        const int rows = (TM - i0 < IPP ? TM - i0 : IPP) * 16;
        for (int v = tid; v < rows * VPR; v += 256) {
            const int pr = v / VPR, cv = v - pr * VPR;
            const int m = *reinterpret_cast<const int *>(Cs + pr * C::CPITCH + BN * 4), g = (n0 >> 3) + (cv >> 1);
            const uint32_t so = (m >= 0 && g < a.outG) ? (uint32_t)((m * a.outG + g) * 32 + (cv & 1) * 16) : X_OOB;   // (out of range: dropped by the descriptor)
            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(Cs + pr * C::CPITCH + cv * 16), rss, so, 0, 0);
        }
    }
    XB_STAMP(9)
    XB_STAMP(10)
#undef XB_STAMP
}
