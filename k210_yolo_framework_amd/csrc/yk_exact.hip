// yk_exact.hip — the "f16x2" precision mode of the engine: fp16 MFMA arithmetic with fp32-class accuracy.
//
// Why: BASELINE.json asks for fp16-MFMA convolutions AND outputs within 1e-3 of the fp32 Keras path with exact class / box
// indices.  fp16 STORAGE alone cannot give the second (a 20-layer chain of 2^-11 roundings drifts to ~2e-3 on the scores and NMS /
// threshold decisions flip).  This mode keeps every activation in fp32 in HBM and feeds the matrix cores with
// compensated operands:
//     x * 2^-e = x_hi + x_lo   (two fp16 values, 22 significant bits; e from the tensor's per-image max so that |x * 2^-e| < 2^14)
//     w * 2^s  = w_hi + w_lo   (split once on the host, s per layer)
//     acc += w_hi*x_hi + w_hi*x_lo + w_lo*x_hi        three v_mfma_f32_16x16x32_f16 per product, fp32 accumulate
// (the dropped w_lo*x_lo term is 2^-22 of the product).  Depthwise convs, the 3-channel stem, pooling and the residual add run in
// fp32 on the VALU.  Same NetSpec, same C-ABI (yk_plan_create_ex(..., precision=1)).  Results depend only on the image itself: the
// operand exponent is taken per image, never per batch, and every tiling decision is fixed at plan creation for max_batch.
//
// Structure (round 2): one templated implicit-GEMM kernel, xconv_kernel<BN, DW> (64 x {64,128,192} tiles, BK 32, two k-steps of
// operands in flight in registers, optional depthwise 3x3 producer in the loader, 2-D pixel patches, two-phase split-K), plus small
// fp32 VALU kernels.  K2 at B=32: 915 us per batch (36 k images/s; 50 k with three batches in flight), logits 1.6e-6 of max|ref|.
//
// Reference layers: Conv2D / DepthwiseConv2D / BatchNormalization / LeakyReLU / ReLU / MaxPooling2D / UpSampling2D / Concatenate /
// Add as built by models/yolonet.py:12-260, models/keras_mobilenet.py:291-436, models/keras_mobilenet_v2.py:426-485.
#include <algorithm>
#include <math.h>
#include <string>
#include <vector>

#include "yk_conv.h"

namespace {

__device__ __forceinline__ float x_actf(float v, float slope, float cap) { return fminf(fmaxf(v, v * slope), cap); }
__device__ __forceinline__ uint32_t x_div(uint32_t n, yk_fastdiv d) { return (__umulhi(n, d.mul) + n) >> d.shift; }

// exponent e with amax * 2^-e in [2^13, 2^14); amax given as float bits (0 -> e = 0)
__device__ __forceinline__ int x_exp_of(uint32_t amax_bits) {
    const int be = (int)((amax_bits >> 23) & 0xffu);
    if (be == 0 || be == 255) return 0;
    return be - 127 - 13;
}
__device__ __forceinline__ float x_pow2(int e) { return __uint_as_float((uint32_t)(127 + max(-126, min(127, e))) << 23); }

// Per-image running maxima live in XS = 64 sub-slots per image: thousands of workgroups hitting ONE address with a device-scope atomic
// serialise at ~0.25 us each across the XCDs (measured: 3/4 of every kernel's time, and a load-then-compare guard is worse still as
// the load has to bypass L2).  A workgroup fires one no-return atomic at slot blockIdx.x % 64; a reader wave loads the 64 slots of an
// image with one coalesced access and reduces them with cross-lane shuffles.
constexpr int XS = 64;
__device__ __forceinline__ void x_amax_global(uint32_t *img_slots, uint32_t bits) { atomicMax(img_slots + (blockIdx.x & (XS - 1)), bits); }
// whole wave: max of image b's slots (every lane returns it)
__device__ __forceinline__ uint32_t x_amax_wave(const uint32_t *base, int b) {
    uint32_t v = base[(size_t)b * XS + (threadIdx.x & 63)];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}

struct xconv_args {
    const float *in0, *in1;
    int c0p, c1p, up0;
    int Hi, Wi, Ho, Wo, ks, stride, pad_t, pad_l;
    int B, M, N, K, HoWo;
    const yk_half *w_hi, *w_lo;        // [N][K], w * 2^s split
    const float *scale, *bias;         // [N padded with zeros]; scale already carries 2^-s
    float slope, cap;
    const float *res;                  // fp32 residual (pitch resp) or null
    int resp;
    float *out;
    int outp, out_exact;               // out_exact: network output (pitch = N, scalar stores)
    const uint32_t *amax_in0, *amax_in1;   // [max_batch] float bits of the per-image max |x| of the sources
    uint32_t *amax_out;                // [max_batch] or null
    yk_fastdiv fd_hw, fd_wo;
    // pixel tiling: a workgroup's 64 GEMM rows are 64 >> sp_sh consecutive SR x SC patches (SR, SC powers of two; 1 x 1 = flat order)
    int sr_sh, sc_sh, TX, TXY;
    yk_fastdiv fd_txy, fd_tx;
    // fused DepthwiseConv2D(3x3)+BN+act producing the GEMM's pixel operand (the 1x1 conv that follows it): in0 is the depthwise INPUT
    const float *dw_par;               // [11][c0p] fp32: nine taps, scale, bias
    int dw_Hi, dw_Wi, dw_stride, dw_pad_t, dw_pad_l;
    float dw_slope, dw_cap, dw_gain, dw_off;   // |dw output| <= min(dw_cap, dw_gain * amax(in) + dw_off)
    // split-K: partial accumulators go to slab[z][tile][reg][thread]; the last workgroup of a tile to arrive adds them in z order
    int splitk, phase;                 // phase 0: whole K; 1: this z's share -> slab; 2: sum the slabs + epilogue
    float *slab;
};

constexpr int XBM = 64, XBK = 32, XLD = XBK + 16;

// operand exponent of image b, computed by a whole wave
__device__ __forceinline__ int x_img_exp(const xconv_args &a, int b, bool dw) {
    const uint32_t m0 = x_amax_wave(a.amax_in0, b);
    if (dw) return x_exp_of(__float_as_uint(fminf(a.dw_cap, a.dw_gain * __uint_as_float(m0) + a.dw_off)));
    int e = x_exp_of(m0);
    if (a.in1) e = max(e, x_exp_of(x_amax_wave(a.amax_in1, b)));
    return e;
}

// wave-wide max into an LDS slot: one atomic per wave when every lane targets the same slot (the common case: one image per tile)
__device__ __forceinline__ void x_amax_lds(uint32_t *simg, int slot, float v) {
    const int s0 = __builtin_amdgcn_readfirstlane(slot);
    if (__all(slot == s0)) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
        if ((threadIdx.x & 63) == 0 && v > 0.f) atomicMax(&simg[s0], __float_as_uint(v));
    } else if (v > 0.f) {
        atomicMax(&simg[slot], __float_as_uint(v));
    }
}

// Conv2D 1x1 / 3x3 as an implicit GEMM with compensated fp16 operands; with DW the pixel operand is produced on the fly by a
// depthwise 3x3 + BN + activation in fp32 (never written to HBM).  256 threads = 2x2 waves, each a 32 x BN/2 output block.
template <int BN, bool DW>
__global__ void __launch_bounds__(256) xconv_kernel(const xconv_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
    constexpr int STG = (2 * XBM + 2 * BN) * XLD;                  // halfs per stage: A_hi, A_lo [64][XLD], B_hi, B_lo [BN][XLD]
    constexpr int NB = BN / 64, NT = BN / 32;
    yk_half *lds = reinterpret_cast<yk_half *>(xsm);
    int *spix = reinterpret_cast<int *>(xsm + (size_t)2 * STG * 2);  // [64] output pixel index of each GEMM row (-1: none)
    int *sb = spix + XBM;                                          // [64] its image
    uint32_t *simg = reinterpret_cast<uint32_t *>(sb + XBM);       // [64] per-image max of this tile's outputs
    int *sexp = reinterpret_cast<int *>(simg + XBM);               // [64] operand exponent of image sb[0] + i
    float *dwl = reinterpret_cast<float *>(sexp + XBM);            // [11][c0p] depthwise parameters (DW)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int n0 = blockIdx.y * BN;
    const int row = tid >> 2, kc = tid & 3;
    const int Ctp = a.c0p + a.c1p, taps = a.ks * a.ks;
    const int H0 = a.up0 ? (a.Hi >> 1) : a.Hi, W0 = a.up0 ? (a.Wi >> 1) : a.Wi;
    if (tid < XBM) {
        const int sp_sh = a.sr_sh + a.sc_sh;
        const uint32_t st = blockIdx.x * (XBM >> sp_sh) + (tid >> sp_sh), q = tid & ((1 << sp_sh) - 1);
        const uint32_t b = x_div(st, a.fd_txy), rem = st - b * a.TXY;
        const uint32_t ty = x_div(rem, a.fd_tx), tx = rem - ty * a.TX;
        const int oy = (int)(ty << a.sr_sh) + (int)(q >> a.sc_sh), ox = (int)(tx << a.sc_sh) + (int)(q & ((1 << a.sc_sh) - 1));
        const bool ok = (int)b < a.B && oy < a.Ho && ox < a.Wo;
        spix[tid] = ok ? ((int)b * a.Ho + oy) * a.Wo + ox : -1;
        sb[tid] = min((int)b, a.B - 1);
        simg[tid] = 0u;
    }
    if (DW)
        for (int i = tid; i < 11 * a.c0p / 4; i += 256) reinterpret_cast<float4 *>(dwl)[i] = reinterpret_cast<const float4 *>(a.dw_par)[i];
    __syncthreads();
    // this thread's A row (one output pixel) and its operand scale
    const int m = spix[row];
    const bool mok = m >= 0;
    const int rb = sb[row], b0 = sb[0];
    int riy = -(1 << 28), rix = 0;
    float sdown = 1.f;
    unsigned vmask = 0;                                            // DW: which of the nine taps fall inside the image
    const float *dwbase = a.in0;
    if (mok) {
        const uint32_t rem = m - rb * a.HoWo;
        const uint32_t oy = x_div(rem, a.fd_wo), ox = rem - oy * a.Wo;
        if (DW) {
            const int y0 = (int)oy * a.dw_stride - a.dw_pad_t, x0 = (int)ox * a.dw_stride - a.dw_pad_l;
#pragma unroll
            for (int t = 0; t < 9; ++t)
                if ((unsigned)(y0 + t / 3) < (unsigned)a.dw_Hi && (unsigned)(x0 + t % 3) < (unsigned)a.dw_Wi) vmask |= 1u << t;
            dwbase = a.in0 + ((long long)(rb * a.dw_Hi + y0) * a.dw_Wi + x0) * a.c0p;
        } else {
            riy = (int)oy * a.stride - a.pad_t;
            rix = (int)ox * a.stride - a.pad_l;
        }
    }
    const int nk = (a.K + XBK - 1) / XBK;
    const int per = (nk + a.splitk - 1) / a.splitk;
    const int kb = blockIdx.z * per, ke = a.phase == 2 ? kb : min(nk, kb + per);
    int kch = kb * XBK + kc * 8, ktap = 0;
    struct xregs {                                                 // one k-step of operands on their way from global memory to LDS
        float4 ra0, ra1;
        half8 rbh[NB], rbl[NB];
        bool aok, bok[NB];                                         // dead loads are zeroed when the registers are consumed (sstore)
    };
    xregs R0, R1;
    float4 xin[DW ? 9 : 1][2];
    int dwch = 0;
    unsigned dwmask = 0;                                           // taps of the staged depthwise step that are real
    // every load below is UNCONDITIONAL (dead ones read a safe address and are zeroed by a select afterwards): a load inside a branch
    // makes the compiler drain the memory queue (s_waitcnt vmcnt(0)) at the join, which serialises the prefetch it was meant to be
    const float4 f4z = make_float4(0.f, 0.f, 0.f, 0.f);
    const half8 h8z = half8{0, 0, 0, 0, 0, 0, 0, 0};
    auto gload = [&](xregs &R, int k0) {
        if (DW) {
            dwch = k0 + kc * 8;
            const bool cok = dwch < a.c0p;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const bool ok = cok && ((vmask >> t) & 1u);
                const float *p = ok ? dwbase + ((t / 3) * a.dw_Wi + (t % 3)) * a.c0p + dwch : a.in0;
                xin[t][0] = *reinterpret_cast<const float4 *>(p);
                xin[t][1] = *reinterpret_cast<const float4 *>(p + 4);
            }
            dwmask = cok ? vmask : 0u;
        } else {
            while (kch >= Ctp) {
                kch -= Ctp;
                ++ktap;
            }
            const int ky = (a.ks == 3) ? ktap / 3 : 0, kx = ktap - ky * a.ks;
            const int iy = riy + ky, ix = rix + kx;
            const bool ok = ktap < taps && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
            const bool first = kch < a.c0p;
            const int sy = a.up0 ? (iy >> 1) : iy, sx = a.up0 ? (ix >> 1) : ix;
            const float *p0 = a.in0 + ((long long)(rb * H0 + sy) * W0 + sx) * a.c0p + kch;
            const float *p1 = a.in1 + ((long long)(rb * a.Hi + iy) * a.Wi + ix) * a.c1p + (kch - a.c0p);
            const float *p = ok ? (first ? p0 : p1) : a.in0;
            R.ra0 = *reinterpret_cast<const float4 *>(p);
            R.ra1 = *reinterpret_cast<const float4 *>(p + 4);
            R.aok = ok;
            kch += XBK;
        }
        const int k = k0 + kc * 8;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int nrow = n0 + row + 64 * i;
            const bool ok = nrow < a.N && k < a.K;
            const size_t at = ok ? (size_t)nrow * a.K + k : 0;
            R.rbh[i] = *reinterpret_cast<const half8 *>(a.w_hi + at);
            R.rbl[i] = *reinterpret_cast<const half8 *>(a.w_lo + at);
            R.bok[i] = ok;
        }
    };
    auto sstore = [&](xregs &R, int stage) {
        yk_half *S = lds + stage * STG;
        float v[8];
        if (DW) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (dwch < a.c0p) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const bool live = (dwmask >> t) & 1u;                          // dead taps were read from a safe address
                    const float4 x0 = live ? xin[t][0] : f4z, x1 = live ? xin[t][1] : f4z;
                    const float4 w0 = *reinterpret_cast<const float4 *>(dwl + t * a.c0p + dwch);
                    const float4 w1 = *reinterpret_cast<const float4 *>(dwl + t * a.c0p + dwch + 4);
                    acc[0] = fmaf(x0.x, w0.x, acc[0]);
                    acc[1] = fmaf(x0.y, w0.y, acc[1]);
                    acc[2] = fmaf(x0.z, w0.z, acc[2]);
                    acc[3] = fmaf(x0.w, w0.w, acc[3]);
                    acc[4] = fmaf(x1.x, w1.x, acc[4]);
                    acc[5] = fmaf(x1.y, w1.y, acc[5]);
                    acc[6] = fmaf(x1.z, w1.z, acc[6]);
                    acc[7] = fmaf(x1.w, w1.w, acc[7]);
                }
                const float *sc = dwl + 9 * a.c0p + dwch, *bs = dwl + 10 * a.c0p + dwch;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = mok ? x_actf(acc[j] * sc[j] + bs[j], a.dw_slope, a.dw_cap) : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = acc[j] * sdown;
        } else {
            const float4 q0 = R.aok ? R.ra0 : f4z, q1 = R.aok ? R.ra1 : f4z;
            v[0] = q0.x * sdown; v[1] = q0.y * sdown; v[2] = q0.z * sdown; v[3] = q0.w * sdown;
            v[4] = q1.x * sdown; v[5] = q1.y * sdown; v[6] = q1.z * sdown; v[7] = q1.w * sdown;
        }
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            hi[j] = (yk_half)v[j];
            lo[j] = (yk_half)(v[j] - (float)hi[j]);
        }
        *reinterpret_cast<half8 *>(S + row * XLD + kc * 8) = hi;
        *reinterpret_cast<half8 *>(S + XBM * XLD + row * XLD + kc * 8) = lo;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            *reinterpret_cast<half8 *>(S + 2 * XBM * XLD + (row + 64 * i) * XLD + kc * 8) = R.bok[i] ? R.rbh[i] : h8z;
            *reinterpret_cast<half8 *>(S + (2 * XBM + BN) * XLD + (row + 64 * i) * XLD + kc * 8) = R.bok[i] ? R.rbl[i] : h8z;
        }
    };
    floatx4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    auto mma = [&](int stage) {
        const yk_half *S = lds + stage * STG;
        half8 xh[2], xl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            xh[i] = *reinterpret_cast<const half8 *>(S + ((wm * 2 + i) * 16 + fr) * XLD + fk);
            xl[i] = *reinterpret_cast<const half8 *>(S + XBM * XLD + ((wm * 2 + i) * 16 + fr) * XLD + fk);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const half8 wh = *reinterpret_cast<const half8 *>(S + 2 * XBM * XLD + (wn * (BN / 2) + j * 16 + fr) * XLD + fk);
            const half8 wl = *reinterpret_cast<const half8 *>(S + (2 * XBM + BN) * XLD + (wn * (BN / 2) + j * 16 + fr) * XLD + fk);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[i], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[i], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[i], acc[i][j], 0, 0, 0);
            }
        }
    };
    // a workgroup's life is a chain of memory round trips (tile map -> exponent slots -> operands -> ... -> scale / bias -> stores) and
    // with short K that chain, not the MFMAs, is the kernel's time: the first k-step's loads go out BEFORE the exponent slots are read
    if (kb < ke) gload(R0, kb * XBK);
    if (wid == 0) {                                                // operand exponent of every image this tile touches (usually one)
        const int bl = sb[XBM - 1];
        for (int b = b0; b <= bl; ++b) {
            const int e = x_img_exp(a, b, DW);
            if (lane == 0) sexp[b - b0] = e;
        }
    }
    __syncthreads();
    if (mok) sdown = x_pow2(-sexp[rb - b0]);
    if (kb < ke) sstore(R0, 0);
    __syncthreads();
    // gload() is called UNCONDITIONALLY in the loops below (a step past the end reads safe addresses and is never stored): with the
    // loads in a branch the compiler cannot count them across the join and waits for ALL outstanding loads before issuing new ones
    if (DW) {
        // the nine taps of the next k-step fly under this step's MFMAs (a second set would cost 72 registers; these layers have 1-3 steps)
        for (int kt = kb; kt < ke; ++kt) {
            gload(R0, (kt + 1) * XBK);
            mma((kt - kb) & 1);
            if (kt + 1 < ke) sstore(R0, ((kt - kb) + 1) & 1);
            __syncthreads();
        }
    } else {
        // two k-steps of global loads in flight (register sets R0 / R1): a step's operands were requested two steps before they are
        // split and stored, so the L2 / HBM round trip overlaps two rounds of MFMAs instead of one
        gload(R1, (kb + 1) * XBK);
        for (int kt = kb; kt < ke; kt += 2) {
            gload(R0, (kt + 2) * XBK);
            mma(0);
            if (kt + 1 < ke) sstore(R1, 1);
            __syncthreads();
            if (kt + 1 >= ke) break;
            gload(R1, (kt + 3) * XBK);
            mma(1);
            if (kt + 2 < ke) sstore(R0, 0);
            __syncthreads();
        }
    }
    if (a.splitk > 1) {
        // phase 1: partial sums -> slab[z][tile][reg][thread]; phase 2 (a second launch of this kernel, grid.z = 1, no k loop): add the
        // slabs in z order and finish.  (A "last workgroup reduces" scheme needs a device-scope release per workgroup, which on this
        // multi-XCD part writes back the whole L2 each time: measured slower than the second launch.)
        const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x, ntile = (size_t)gridDim.x * gridDim.y;
        if (a.phase == 1) {
            floatx4 *mine = reinterpret_cast<floatx4 *>(a.slab) + ((size_t)blockIdx.z * ntile + tile) * (2 * NT) * 256 + tid;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) mine[(i * NT + j) * 256] = acc[i][j];
            return;
        }
        for (int z = 0; z < a.splitk; ++z) {
            const floatx4 *src = reinterpret_cast<const floatx4 *>(a.slab) + ((size_t)z * ntile + tile) * (2 * NT) * 256 + tid;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] += __builtin_nontemporal_load(src + (i * NT + j) * 256);
        }
    }
    // epilogue: lane holds channels n..n+3 of GEMM row (wm*2+i)*16 + fr
    const int nl4 = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wm * 2 + i) * 16 + fr;
        const int mm = spix[r], b = sb[r];
        float rmax = 0.f;
        if (mm >= 0) {
            const float up = x_pow2(sexp[b - b0]);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + wn * (BN / 2) + j * 16 + nl4;
                if (n >= (a.out_exact ? a.N : a.outp)) continue;
                const float4 sc = *reinterpret_cast<const float4 *>(a.scale + n), bs = *reinterpret_cast<const float4 *>(a.bias + n);
                float4 v;
                v.x = x_actf(acc[i][j][0] * up * sc.x + bs.x, a.slope, a.cap);
                v.y = x_actf(acc[i][j][1] * up * sc.y + bs.y, a.slope, a.cap);
                v.z = x_actf(acc[i][j][2] * up * sc.z + bs.z, a.slope, a.cap);
                v.w = x_actf(acc[i][j][3] * up * sc.w + bs.w, a.slope, a.cap);
                if (a.out_exact) {
                    float *o = a.out + (size_t)mm * a.outp + n;
                    if (n + 0 < a.N) o[0] = v.x;
                    if (n + 1 < a.N) o[1] = v.y;
                    if (n + 2 < a.N) o[2] = v.z;
                    if (n + 3 < a.N) o[3] = v.w;
                } else {
                    if (a.res && n < a.resp) {
                        const float4 q = *reinterpret_cast<const float4 *>(a.res + (size_t)mm * a.resp + n);
                        v.x += q.x;
                        v.y += q.y;
                        v.z += q.z;
                        v.w += q.w;
                    }
                    *reinterpret_cast<float4 *>(a.out + (size_t)mm * a.outp + n) = v;
                    rmax = fmaxf(rmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                }
            }
        }
        if (a.amax_out) x_amax_lds(simg, b - b0, rmax);
    }
    if (a.amax_out) {
        __syncthreads();
        if (tid < XBM && simg[tid] && b0 + tid < a.B) x_amax_global(a.amax_out + (size_t)(b0 + tid) * XS, simg[tid]);
    }
}

// ---- depthwise 3x3, fp32, one thread = (pixel, 4 channels); grid (pixel groups of one image, image) ------------
struct xdw_args {
    const float *in;
    int B, Hi, Wi, Ho, Wo, Cp, stride, pad_t, pad_l;
    const float *w;                    // [9][Cp] fp32
    const float *scale, *bias;
    float slope, cap;
    float *out;
    uint32_t *amax_out;
    yk_fastdiv fd_g, fd_wo;            // division by Cp/4 and Wo
};
__global__ void __launch_bounds__(256) xdw_kernel(const xdw_args a) {
    __shared__ uint32_t smax;
    const int tid = threadIdx.x, b = blockIdx.y;
    if (tid == 0) smax = 0u;
    __syncthreads();
    const uint32_t G = a.Cp >> 2, idx = blockIdx.x * 256 + tid;
    float mx = 0.f;
    if (idx < (uint32_t)(a.Ho * a.Wo) * G) {
        const uint32_t pix = x_div(idx, a.fd_g), g = idx - pix * G;
        const uint32_t oy = x_div(pix, a.fd_wo), ox = pix - oy * a.Wo;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int iy0 = (int)oy * a.stride - a.pad_t, ix0 = (int)ox * a.stride - a.pad_l;
        const float *src = a.in + (size_t)b * a.Hi * a.Wi * a.Cp + g * 4;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = iy0 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ix0 + kx;
                const bool ok = (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
                const float4 x = ok ? *reinterpret_cast<const float4 *>(src + (iy * a.Wi + ix) * a.Cp) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 w = *reinterpret_cast<const float4 *>(a.w + (ky * 3 + kx) * a.Cp + g * 4);
                acc.x = fmaf(x.x, w.x, acc.x);
                acc.y = fmaf(x.y, w.y, acc.y);
                acc.z = fmaf(x.z, w.z, acc.z);
                acc.w = fmaf(x.w, w.w, acc.w);
            }
        }
        const float4 sc = *reinterpret_cast<const float4 *>(a.scale + g * 4), bs = *reinterpret_cast<const float4 *>(a.bias + g * 4);
        float4 v;
        v.x = x_actf(acc.x * sc.x + bs.x, a.slope, a.cap);
        v.y = x_actf(acc.y * sc.y + bs.y, a.slope, a.cap);
        v.z = x_actf(acc.z * sc.z + bs.z, a.slope, a.cap);
        v.w = x_actf(acc.w * sc.w + bs.w, a.slope, a.cap);
        *reinterpret_cast<float4 *>(a.out + ((size_t)b * a.Ho * a.Wo + pix) * a.Cp + g * 4) = v;
        mx = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (!a.amax_out) return;
    x_amax_lds(&smax, 0, mx);
    __syncthreads();
    if (tid == 0 && smax) x_amax_global(a.amax_out + (size_t)b * XS, smax);
}

// ---- stem conv (Cin = 3), fp32 VALU; u8 frames are normalised as float(v)/float(max) = numpy's `img / np.max(img)` rounded once
struct xstem_args {
    const void *in;
    const unsigned *img_max;           // YK_MAXP partial maxima per image (u8 path)
    int in_f32, B, Hi, Wi, Ho, Wo, stride, pad_t, pad_l, Cout, outp;
    const float *w;                    // [27][Cout]
    const float *scale, *bias;
    float slope, cap;
    float *out;
    uint32_t *amax_out;
};
template <int COUT>
__global__ void __launch_bounds__(256) xstem_kernel(const xstem_args a) {
    // the 27 x COUT weights, scale and bias are the same for every lane: read straight from the kernel-argument pointers they become
    // scalar loads feeding v_fma's SGPR operand (through LDS every FMA paid a ds_read)
    __shared__ float lut[256];
    __shared__ uint32_t smax;
    const int tid = threadIdx.x, b = blockIdx.y;
    const float *__restrict__ wl = a.w, *__restrict__ sc = a.scale, *__restrict__ bs = a.bias;
    if (tid == 0) smax = 0u;
    if (!a.in_f32) {
        unsigned mx = 0;
#pragma unroll
        for (int j = 0; j < 32; ++j) mx = max(mx, a.img_max[b * 32 + j]);
        lut[tid] = (float)tid / (float)mx;
    }
    __syncthreads();
    const int pix = blockIdx.x * 256 + tid;
    float vmax = 0.f;
    if (pix < a.Ho * a.Wo) {
        const int oy = pix / a.Wo, ox = pix - oy * a.Wo;
        float acc[COUT];
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
        const int iy0 = oy * a.stride - a.pad_t, ix0 = ox * a.stride - a.pad_l;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = iy0 + ky;
            if ((unsigned)iy >= (unsigned)a.Hi) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ix0 + kx;
                if ((unsigned)ix >= (unsigned)a.Wi) continue;
                const size_t off = ((size_t)(b * a.Hi + iy) * a.Wi + ix) * 3;
                float x[3];
                if (a.in_f32) {
                    const float *p = reinterpret_cast<const float *>(a.in) + off;
                    x[0] = p[0];
                    x[1] = p[1];
                    x[2] = p[2];
                } else {
                    const uint8_t *p = reinterpret_cast<const uint8_t *>(a.in) + off;
                    x[0] = lut[p[0]];
                    x[1] = lut[p[1]];
                    x[2] = lut[p[2]];
                }
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const float *w = wl + ((ky * 3 + kx) * 3 + ci) * COUT;
#pragma unroll
                    for (int c = 0; c < COUT; ++c) acc[c] = fmaf(x[ci], w[c], acc[c]);
                }
            }
        }
        float *o = a.out + ((size_t)b * a.Ho * a.Wo + pix) * a.outp;
#pragma unroll
        for (int c4 = 0; c4 < COUT; c4 += 4) {
            float4 v;
            v.x = x_actf(acc[c4] * sc[c4] + bs[c4], a.slope, a.cap);
            v.y = x_actf(acc[c4 + 1] * sc[c4 + 1] + bs[c4 + 1], a.slope, a.cap);
            v.z = x_actf(acc[c4 + 2] * sc[c4 + 2] + bs[c4 + 2], a.slope, a.cap);
            v.w = x_actf(acc[c4 + 3] * sc[c4 + 3] + bs[c4 + 3], a.slope, a.cap);
            *reinterpret_cast<float4 *>(o + c4) = v;
            vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    if ((tid & 63) == 0) atomicMax(&smax, __float_as_uint(vmax));
    __syncthreads();
    if (tid == 0 && smax) x_amax_global(a.amax_out + (size_t)b * XS, smax);
}

// ---- 2x2 max pool 'same' and residual add, fp32 ---------------------------------------------------------------
struct xpool_args {
    const float *in;
    int B, Hi, Wi, Ho, Wo, Cp, stride;
    float *out;
    uint32_t *amax_out;
};
__global__ void __launch_bounds__(256) xpool_kernel(const xpool_args a) {
    __shared__ uint32_t simg[256];
    const int tid = threadIdx.x;
    simg[tid] = 0u;
    __syncthreads();
    const int G = a.Cp >> 2;
    const size_t total = (size_t)a.B * a.Ho * a.Wo * G, first = (size_t)blockIdx.x * 256, idx = first + tid;
    const int b0 = (int)((first / G) / ((size_t)a.Ho * a.Wo));
    float mx = 0.f;
    int slot = 0;
    if (idx < total) {
        const int g = (int)(idx % G);
        const size_t pix = idx / G;
        const int ox = (int)(pix % a.Wo), oy = (int)((pix / a.Wo) % a.Ho), b = (int)(pix / ((size_t)a.Wo * a.Ho));
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int ky = 0; ky < 2; ++ky)
#pragma unroll
            for (int kx = 0; kx < 2; ++kx) {
                const int iy = oy * a.stride + ky, ix = ox * a.stride + kx;
                if (iy >= a.Hi || ix >= a.Wi) continue;
                const float4 x = *reinterpret_cast<const float4 *>(a.in + ((size_t)(b * a.Hi + iy) * a.Wi + ix) * a.Cp + g * 4);
                m.x = fmaxf(m.x, x.x);
                m.y = fmaxf(m.y, x.y);
                m.z = fmaxf(m.z, x.z);
                m.w = fmaxf(m.w, x.w);
            }
        *reinterpret_cast<float4 *>(a.out + pix * a.Cp + g * 4) = m;
        mx = fmaxf(fmaxf(fabsf(m.x), fabsf(m.y)), fmaxf(fabsf(m.z), fabsf(m.w)));
        slot = min(b - b0, 255);
    }
    x_amax_lds(simg, slot, mx);
    __syncthreads();
    if (simg[tid] && b0 + tid < a.B) x_amax_global(a.amax_out + (size_t)(b0 + tid) * XS, simg[tid]);
}
__global__ void __launch_bounds__(256) xadd_kernel(const float *x, const float *y, float *o, size_t n4_per_image, int B, uint32_t *amax_out) {
    __shared__ uint32_t smax;
    if (threadIdx.x == 0) smax = 0u;
    __syncthreads();
    const int b = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float mx = 0.f;
    if (i < n4_per_image) {
        const size_t k = (size_t)b * n4_per_image + i;
        const float4 p = reinterpret_cast<const float4 *>(x)[k], q = reinterpret_cast<const float4 *>(y)[k];
        const float4 r = make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w);
        reinterpret_cast<float4 *>(o)[k] = r;
        mx = fmaxf(fmaxf(fabsf(r.x), fabsf(r.y)), fmaxf(fabsf(r.z), fabsf(r.w)));
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&smax, __float_as_uint(mx));
    __syncthreads();
    if (threadIdx.x == 0 && smax) x_amax_global(amax_out + (size_t)b * XS, smax);
}

// ---- host side ------------------------------------------------------------------------------------------------
uint16_t x_f2h(float f) {   // round-to-nearest-even fp32 -> fp16 bits (normal / subnormal / overflow to inf)
    _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
float x_h2f(uint16_t u) {
    _Float16 h;
    memcpy(&h, &u, 2);
    return (float)h;
}

enum { XK_STEM = 1, XK_CONV, XK_DW, XK_POOL, XK_ADD, XK_U8MAX };
enum { XT_REAL = 0, XT_UP = 1, XT_CAT = 2 };
struct xtens {
    int h = 0, w = 0, c = 0, cp = 0, kind = XT_REAL, src0 = -1, src1 = -1;
    bool net_out = false, is_input = false;
    float *d = nullptr;
    int uses = 0;
};
struct xlaunch {
    int kind = 0;
    xconv_args c;
    xdw_args d;
    xstem_args s;
    xpool_args p;
    const float *add_a = nullptr, *add_b = nullptr;
    float *add_o = nullptr;
    uint32_t *add_amax = nullptr;
    size_t add_n4 = 0;
    int Ho = 0, Wo = 0;
    int bn = 64, dw = 0;               // xconv_kernel<bn, dw>
    unsigned lds = 0;
    std::string name;
    double flops = 0, bytes = 0;
};


// pixel patch of a GEMM row tile: minimise (pixels computed / pixels kept) x (input halo read / patch); 1 x 1 = flat pixel order
static void x_pick_patch(int Ho, int Wo, bool spatial, int *sr_sh, int *sc_sh) {
    *sr_sh = *sc_sh = 0;
    if (!spatial) return;
    const double rows = 64.0 / Wo;
    double best = Wo >= 64 ? 3.0 : (rows + 2.0) / rows;             // flat: 64 consecutive pixels touch their rows plus one above, one below
    static const int cand[][2] = {{3, 3}, {2, 4}, {4, 2}, {2, 3}, {3, 2}};
    for (auto &c : cand) {
        const int SR = 1 << c[0], SC = 1 << c[1];
        const double padded = (double)((Ho + SR - 1) / SR * SR) * ((Wo + SC - 1) / SC * SC) / ((double)Ho * Wo);
        const double cost = padded * (SR + 2.0) * (SC + 2.0) / (SR * SC);
        if (cost < best - 1e-9) {
            best = cost;
            *sr_sh = c[0];
            *sc_sh = c[1];
        }
    }
}
static int x_pick_bn(int N, bool dw) {
    // per CU: 3 workgroups of the 64-wide tile, 2 of the 128-wide, 1 of the 192-wide (LDS); a k-step's latency is hidden only by the
    // other workgroups, so the 192-wide tile is kept for the fused depthwise case where it saves deriving the depthwise tile twice
    if (N <= 64) return 64;
    if (dw && N > 128 && N <= 192) return 192;
    if (N <= 128 || N % 128 == 0) return 128;
    return 64;
}
static bool x_env_flag(const char *name, bool dflt) {
    const char *e = getenv(name);
    return (e && e[0]) ? e[0] != '0' : dflt;
}

}   // namespace

struct yk_xplan {
    int max_batch = 0, in_h = 0, in_w = 0;
    std::vector<xtens> T;
    std::vector<xlaunch> L;
    std::vector<void *> allocs;
    std::vector<int> outputs;
    unsigned *d_imgmax = nullptr;
    uint32_t *d_amax = nullptr;        // [n_tensors][max_batch][XS]
    uint32_t *d_one = nullptr;         // [max_batch][XS] bits of 1.0f (the normalised image)
};

static int x_alloc(yk_xplan *p, void **ptr, size_t bytes) {
    YK_HIP(hipMalloc(ptr, bytes));
    p->allocs.push_back(*ptr);
    YK_HIP(hipMemset(*ptr, 0, bytes));
    return YK_OK;
}
static int x_upload(yk_xplan *p, void **ptr, const void *src, size_t bytes) {
    int rc = x_alloc(p, ptr, bytes);
    if (rc) return rc;
    YK_HIP(hipMemcpy(*ptr, src, bytes, hipMemcpyHostToDevice));
    return YK_OK;
}
static int x_upload_f(yk_xplan *p, const float *src, int n, float mul, const float **d) {
    std::vector<float> v((size_t)n + 256, 0.f);
    for (int i = 0; i < n; ++i) v[i] = src[i] * mul;
    void *q;
    int rc = x_upload(p, &q, v.data(), v.size() * sizeof(float));
    *d = (const float *)q;
    return rc;
}

template <int BN, bool DW>
static int x_launch_conv(const xconv_args &g, dim3 grid, unsigned lds, hipStream_t st) {
    static unsigned allowed = 64 * 1024;                             // dynamic LDS above 64 KB has to be enabled per kernel
    if (lds > allowed) {
        YK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(xconv_kernel<BN, DW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        allowed = lds;
    }
    hipLaunchKernelGGL((xconv_kernel<BN, DW>), grid, dim3(256), lds, st, g);
    return YK_OK;
}

void yk_xplan_destroy(yk_xplan *p) {
    if (!p) return;
    for (void *q : p->allocs) (void)hipFree(q);
    delete p;
}

int yk_xplan_create(yk_xplan **out, const int32_t *ops, int n_ops, const int32_t *tensors, int n_tensors, const float *blob,
                    size_t blob_len, const int32_t *outputs, int n_outputs, int max_batch) {
    yk_xplan *p = new yk_xplan();
    p->max_batch = max_batch;
    int rc = YK_OK;
    auto fail = [&](int code) {
        yk_xplan_destroy(p);
        return code;
    };
    p->T.resize(n_tensors);
    for (int i = 0; i < n_tensors; ++i) {
        xtens &t = p->T[i];
        t.h = tensors[4 * i];
        t.w = tensors[4 * i + 1];
        t.c = tensors[4 * i + 2];
        t.cp = yk_pad8(t.c);
        t.is_input = tensors[4 * i + 3] != 0;
    }
    p->in_h = p->T[0].h;
    p->in_w = p->T[0].w;
    for (int i = 0; i < n_outputs; ++i) p->outputs.push_back(outputs[i]);
    std::vector<int> add_of(n_ops, -1);
    std::vector<char> skip(n_ops, 0);
    for (int i = 0; i < n_ops; ++i) {
        const int32_t *o = ops + (size_t)i * YK_OP_FIELDS;
        const int ty = o[YK_F_TYPE], in0 = o[YK_F_IN0], in1 = o[YK_F_IN1], ot = o[YK_F_OUT];
        if (in0 < 0 || in0 >= n_tensors || ot <= 0 || ot >= n_tensors || in1 >= n_tensors) {
            yk_set_error("yk_plan_create: op %d has a bad tensor id", i);
            return fail(YK_ERR_ARG);
        }
        p->T[in0].uses++;
        if (in1 >= 0) p->T[in1].uses++;
        if (ty == YK_OP_UPSAMPLE) {
            p->T[ot].kind = XT_UP;
            p->T[ot].src0 = in0;
        } else if (ty == YK_OP_CONCAT) {
            p->T[ot].kind = XT_CAT;
            p->T[ot].src0 = in0;
            p->T[ot].src1 = in1;
        }
        if (ty == YK_OP_CONV && (o[YK_F_FLAGS] & YK_FLAG_NET_OUTPUT)) p->T[ot].net_out = true;
    }
    for (int t : p->outputs) p->T[t].uses++;
    for (int i = 0; i + 1 < n_ops; ++i) {      // residual Add folded into the producing conv's epilogue
        const int32_t *o = ops + (size_t)i * YK_OP_FIELDS, *q = o + YK_OP_FIELDS;
        if (o[YK_F_TYPE] == YK_OP_CONV && q[YK_F_TYPE] == YK_OP_ADD && !(o[YK_F_FLAGS] & YK_FLAG_NET_OUTPUT)) {
            const int y = o[YK_F_OUT];
            const int other = (q[YK_F_IN0] == y) ? q[YK_F_IN1] : (q[YK_F_IN1] == y ? q[YK_F_IN0] : -1);
            if (other >= 0 && other != y && p->T[y].uses == 1 && p->T[other].kind == XT_REAL && !p->T[other].is_input) {
                add_of[i] = i + 1;
                skip[i + 1] = 1;
            }
        }
    }
    // DepthwiseConv2D(3x3) whose only consumer is the next op, a 1x1 stride-1 Conv2D: produced inside that conv's kernel
    std::vector<int> dw_of(n_ops, -1);
    std::vector<char> gone(n_tensors, 0);
    // measured (K2, B=32, us fused vs depthwise + 1x1 launches): 24 ch 119 vs 109, 48 ch 67 vs 73, 96 ch (stride 1) 67 vs 93,
    // 96 ch (stride 2) 50 vs 44, 192 ch 65 vs 58, 384 ch 73 vs 42: the fused workgroup has one wave per SIMD to hide the nine taps'
    // latency and re-derives the depthwise tile for every N tile, so it only pays where the depthwise tensor is big and narrow
    const int fuse_max_c = yk_dev_env("YK_X_FUSE_MAXC") ? atoi(yk_dev_env("YK_X_FUSE_MAXC")) : 96;
    const int fuse_min_c = yk_dev_env("YK_X_FUSE_MINC") ? atoi(yk_dev_env("YK_X_FUSE_MINC")) : 48;
    if (x_env_flag("YK_FUSE_DWPW", true))
        for (int i = 0; i + 1 < n_ops; ++i) {
            const int32_t *o = ops + (size_t)i * YK_OP_FIELDS, *q = o + YK_OP_FIELDS;
            const int y = o[YK_F_OUT];
            if (o[YK_F_TYPE] == YK_OP_DWCONV && q[YK_F_TYPE] == YK_OP_CONV && q[YK_F_K] == 1 && q[YK_F_STRIDE] == 1 && q[YK_F_IN0] == y &&
                p->T[y].uses == 1 && p->T[y].kind == XT_REAL && p->T[o[YK_F_IN0]].kind == XT_REAL && !p->T[o[YK_F_IN0]].is_input &&
                p->T[y].cp * o[YK_F_STRIDE] <= fuse_max_c && p->T[y].cp >= fuse_min_c) {
                dw_of[i + 1] = i;
                skip[i] = 1;
                gone[y] = 1;
            }
        }
    for (int i = 1; i < n_tensors; ++i) {
        xtens &t = p->T[i];
        if (t.kind != XT_REAL || gone[i]) continue;
        bool folded = false;
        for (int k = 0; k < n_ops; ++k)
            if (ops[(size_t)k * YK_OP_FIELDS + YK_F_OUT] == i && add_of[k] >= 0) folded = true;
        if (folded) continue;
        const size_t pitch = t.net_out ? t.c : t.cp;
        if ((rc = x_alloc(p, (void **)&t.d, ((size_t)max_batch * t.h * t.w * pitch + 64) * sizeof(float)))) return fail(rc);
    }
    if ((rc = x_alloc(p, (void **)&p->d_imgmax, sizeof(unsigned) * max_batch * 32))) return fail(rc);
    if ((rc = x_alloc(p, (void **)&p->d_amax, sizeof(uint32_t) * (size_t)n_tensors * max_batch * XS))) return fail(rc);
    {
        std::vector<uint32_t> one((size_t)max_batch * XS, 0x3f800000u);
        void *q;
        if ((rc = x_upload(p, &q, one.data(), one.size() * 4))) return fail(rc);
        p->d_one = (uint32_t *)q;
    }
    auto amax_of = [&](int tid) { return p->d_amax + (size_t)tid * max_batch * XS; };
    {
        xlaunch l;
        l.kind = XK_U8MAX;
        l.name = "u8_max";
        l.bytes = (double)p->in_h * p->in_w * 3;
        p->L.push_back(l);
    }
    for (int i = 0; i < n_ops; ++i) {
        if (skip[i]) continue;
        const int32_t *o = ops + (size_t)i * YK_OP_FIELDS;
        const int ty = o[YK_F_TYPE];
        if (ty == YK_OP_UPSAMPLE || ty == YK_OP_CONCAT) continue;
        const int xid = o[YK_F_IN0], yid = o[YK_F_OUT];
        const xtens &X = p->T[xid];
        xtens &Y = p->T[yid];
        float alpha;
        memcpy(&alpha, &o[YK_F_ALPHA], 4);
        xlaunch l;
        l.Ho = Y.h;
        l.Wo = Y.w;
        char nm[96];
        if (ty == YK_OP_CONV && X.is_input) {
            const int co = o[YK_F_COUT];
            if (o[YK_F_K] != 3 || (co != 16 && co != 24 && co != 32) || Y.net_out) {
                yk_set_error("op %d: stem conv must be 3x3 with 16/24/32 filters", i);
                return fail(YK_ERR_UNSUPPORTED);
            }
            std::vector<float> w((size_t)27 * co);
            for (int c = 0; c < co; ++c)
                for (int t = 0; t < 27; ++t) w[(size_t)t * co + c] = blob[o[YK_F_W_OFF] + (size_t)c * 27 + t];
            void *dw_;
            if ((rc = x_upload(p, &dw_, w.data(), w.size() * sizeof(float)))) return fail(rc);
            l.kind = XK_STEM;
            xstem_args &s = l.s;
            memset(&s, 0, sizeof(s));
            s.Hi = X.h; s.Wi = X.w; s.Ho = Y.h; s.Wo = Y.w;
            s.stride = o[YK_F_STRIDE]; s.pad_t = o[YK_F_PAD_T]; s.pad_l = o[YK_F_PAD_L];
            s.Cout = co; s.outp = Y.cp; s.w = (const float *)dw_;
            if ((rc = x_upload_f(p, blob + o[YK_F_SCALE_OFF], co, 1.f, &s.scale))) return fail(rc);
            if ((rc = x_upload_f(p, blob + o[YK_F_BIAS_OFF], co, 1.f, &s.bias))) return fail(rc);
            yk_act_params(o[YK_F_ACT], alpha, &s.slope, &s.cap);
            s.out = Y.d;
            s.amax_out = amax_of(yid);
            snprintf(nm, sizeof nm, "x:stem3x3s%d_%d", s.stride, co);
            l.flops = 2.0 * Y.h * Y.w * 27 * co;
            l.bytes = (double)X.h * X.w * 3 * 4 + (double)Y.h * Y.w * co * 4;
        } else if (ty == YK_OP_CONV) {
            l.kind = XK_CONV;
            xconv_args &g = l.c;
            memset(&g, 0, sizeof(g));
            int s0 = xid, s1 = -1, up0 = 0;
            const int32_t *dwo = dw_of[i] >= 0 ? ops + (size_t)dw_of[i] * YK_OP_FIELDS : nullptr;
            if (dwo) s0 = dwo[YK_F_IN0];                              // fused depthwise producer: read ITS input
            if (X.kind == XT_CAT) {
                s0 = X.src0;
                s1 = X.src1;
            }
            if (p->T[s0].kind == XT_UP) {
                up0 = 1;
                s0 = p->T[s0].src0;
            }
            const xtens &S0 = p->T[s0];
            const xtens *S1 = s1 >= 0 ? &p->T[s1] : nullptr;
            if (S0.kind != XT_REAL || (S1 && S1->kind != XT_REAL) || !S0.d || (S1 && !S1->d)) {
                yk_set_error("op %d: unsupported input view nesting", i);
                return fail(YK_ERR_UNSUPPORTED);
            }
            const int ks = o[YK_F_K], co = o[YK_F_COUT], cin = o[YK_F_CIN];
            const int c0 = S0.c, c0p = S0.cp, c1 = S1 ? S1->c : 0, c1p = S1 ? S1->cp : 0;
            if (c0 + c1 != cin || (ks != 1 && ks != 3)) {
                yk_set_error("op %d: conv shape mismatch", i);
                return fail(YK_ERR_UNSUPPORTED);
            }
            g.in0 = S0.d; g.in1 = S1 ? S1->d : nullptr;
            g.c0p = c0p; g.c1p = c1p; g.up0 = up0;
            g.Hi = X.h; g.Wi = X.w; g.Ho = Y.h; g.Wo = Y.w; g.HoWo = Y.h * Y.w;
            g.ks = ks; g.stride = o[YK_F_STRIDE]; g.pad_t = o[YK_F_PAD_T]; g.pad_l = o[YK_F_PAD_L];
            g.N = co; g.K = ks * ks * (c0p + c1p);
            // weight split: w * 2^s = hi + lo with max |w * 2^s| in [2^13, 2^14)
            float wmax = 0.f;
            const size_t nw = (size_t)co * ks * ks * cin;
            for (size_t k = 0; k < nw; ++k) wmax = std::max(wmax, fabsf(blob[o[YK_F_W_OFF] + k]));
            const int sexp = (wmax > 0.f && std::isfinite(wmax)) ? 13 - ilogbf(wmax) : 0;
            std::vector<uint16_t> wh((size_t)co * g.K, 0), wl((size_t)co * g.K, 0);
            for (int n = 0; n < co; ++n)
                for (int t = 0; t < ks * ks; ++t)
                    for (int c = 0; c < cin; ++c) {
                        const int pos = c < c0 ? c : c0p + (c - c0);
                        const float v = ldexpf(blob[o[YK_F_W_OFF] + ((size_t)n * ks * ks + t) * cin + c], sexp);
                        const uint16_t hi = x_f2h(v);
                        const size_t at = (size_t)n * g.K + (size_t)t * (c0p + c1p) + pos;
                        wh[at] = hi;
                        wl[at] = x_f2h(v - x_h2f(hi));
                    }
            void *d1, *d2;
            if ((rc = x_upload(p, &d1, wh.data(), wh.size() * 2))) return fail(rc);
            if ((rc = x_upload(p, &d2, wl.data(), wl.size() * 2))) return fail(rc);
            g.w_hi = (const yk_half *)d1;
            g.w_lo = (const yk_half *)d2;
            if ((rc = x_upload_f(p, blob + o[YK_F_SCALE_OFF], co, ldexpf(1.f, -sexp), &g.scale))) return fail(rc);
            if ((rc = x_upload_f(p, blob + o[YK_F_BIAS_OFF], co, 1.f, &g.bias))) return fail(rc);
            yk_act_params(o[YK_F_ACT], alpha, &g.slope, &g.cap);
            g.fd_hw = yk_make_fastdiv((uint32_t)(Y.h * Y.w));
            g.fd_wo = yk_make_fastdiv((uint32_t)Y.w);
            g.amax_in0 = S0.is_input ? p->d_one : amax_of(s0);
            g.amax_in1 = S1 ? amax_of(s1) : nullptr;
            if (dwo) {
                // [11][c0p]: nine taps, BN scale, BN bias of the depthwise conv; bound of its output from the input's max
                std::vector<float> par((size_t)11 * c0p, 0.f);
                float gain = 0.f, off = 0.f, dalpha;
                memcpy(&dalpha, &dwo[YK_F_ALPHA], 4);
                for (int k = 0; k < c0; ++k) {
                    float sw = 0.f;
                    for (int t = 0; t < 9; ++t) {
                        const float w = blob[dwo[YK_F_W_OFF] + (size_t)t * c0 + k];
                        par[(size_t)t * c0p + k] = w;
                        sw += fabsf(w);
                    }
                    const float sc = blob[dwo[YK_F_SCALE_OFF] + k], bs = blob[dwo[YK_F_BIAS_OFF] + k];
                    par[(size_t)9 * c0p + k] = sc;
                    par[(size_t)10 * c0p + k] = bs;
                    gain = std::max(gain, fabsf(sc) * sw);
                    off = std::max(off, fabsf(bs));
                }
                void *dp;
                if ((rc = x_upload(p, &dp, par.data(), par.size() * sizeof(float)))) return fail(rc);
                g.dw_par = (const float *)dp;
                g.dw_Hi = S0.h; g.dw_Wi = S0.w;
                g.dw_stride = dwo[YK_F_STRIDE]; g.dw_pad_t = dwo[YK_F_PAD_T]; g.dw_pad_l = dwo[YK_F_PAD_L];
                yk_act_params(dwo[YK_F_ACT], dalpha, &g.dw_slope, &g.dw_cap);
                g.dw_gain = gain * 1.0001f;
                g.dw_off = off * 1.0001f;
                l.dw = 1;
            }
            // tile shape, pixel patch and K split are fixed here, for max_batch: an image's arithmetic never depends on the batch
            l.bn = x_pick_bn(co, l.dw != 0);
            if (const char *e = yk_dev_env("YK_X_BN")) {
                const int v = atoi(e);
                if (v == 64 || v == 128 || v == 192) l.bn = v;
            }
            x_pick_patch(Y.h, Y.w, l.dw || ks == 3, &g.sr_sh, &g.sc_sh);
            if (const char *e = yk_dev_env("YK_X_PATCH"))
                if (e[0] == '0') g.sr_sh = g.sc_sh = 0;
            g.TX = (Y.w + (1 << g.sc_sh) - 1) >> g.sc_sh;
            g.TXY = g.TX * ((Y.h + (1 << g.sr_sh) - 1) >> g.sr_sh);
            g.fd_txy = yk_make_fastdiv((uint32_t)g.TXY);
            g.fd_tx = yk_make_fastdiv((uint32_t)g.TX);
            l.lds = (unsigned)((size_t)2 * (2 * XBM + 2 * l.bn) * XLD * 2 + 4 * XBM * 4 + (l.dw ? (size_t)11 * c0p * 4 : 0));
            {
                const long mt = ((long)max_batch * g.TXY + (XBM >> (g.sr_sh + g.sc_sh)) - 1) / (XBM >> (g.sr_sh + g.sc_sh));
                // a k-step is a dependent chain (loads -> split -> LDS -> MFMA), hidden only by other workgroups on the CU: small
                // problems take the narrow tile (3 workgroups per CU) and share K out until there are ~3 workgroups per CU
                if (!yk_dev_env("YK_X_BN") && mt * ((co + l.bn - 1) / l.bn) < 256) l.bn = 64;
                l.lds = (unsigned)((size_t)2 * (2 * XBM + 2 * l.bn) * XLD * 2 + 4 * XBM * 4 + (l.dw ? (size_t)11 * c0p * 4 : 0));
                const long tiles = mt * ((co + l.bn - 1) / l.bn);
                const int nk = (g.K + XBK - 1) / XBK;
                long sk = 1;
                if (tiles < 384 && nk >= 16) sk = std::min<long>(std::min<long>(8, (768 + tiles - 1) / tiles), nk / 4);
                if (const char *e = yk_dev_env("YK_X_SPLITK")) sk = std::max(1, std::min(atoi(e), nk));
                g.splitk = (int)std::max<long>(1, sk);
                if (g.splitk > 1) {
                    void *sl;
                    if ((rc = x_alloc(p, &sl, (size_t)g.splitk * tiles * (l.bn / 16) * 256 * 16))) return fail(rc);
                    g.slab = (float *)sl;
                }
            }
            xtens *dst = &Y;
            int dst_id = yid;
            if (add_of[i] >= 0) {
                const int32_t *q = ops + (size_t)add_of[i] * YK_OP_FIELDS;
                const int other = (q[YK_F_IN0] == yid) ? q[YK_F_IN1] : q[YK_F_IN0];
                g.res = p->T[other].d;
                g.resp = p->T[other].cp;
                dst_id = q[YK_F_OUT];
                dst = &p->T[dst_id];
            }
            g.out = dst->d;
            g.out_exact = dst->net_out ? 1 : 0;
            g.outp = dst->net_out ? dst->c : dst->cp;
            g.amax_out = dst->net_out ? nullptr : amax_of(dst_id);
            if (!g.out) {
                yk_set_error("op %d: output tensor not allocated", i);
                return fail(YK_ERR_UNSUPPORTED);
            }
            char tl[48];
            snprintf(tl, sizeof tl, "[64x%d%s%s]", l.bn, (g.sr_sh + g.sc_sh) ? ",patch" : "", g.splitk > 1 ? ",splitk" : "");
            if (dwo)
                snprintf(nm, sizeof nm, "x:dw3x3s%d+conv1x1_%dto%d%s%s", g.dw_stride, cin, co, g.res ? "+add" : "", tl);
            else
                snprintf(nm, sizeof nm, "x:conv%dx%ds%d_%dto%d%s%s%s", ks, ks, g.stride, cin, co, g.res ? "+add" : "",
                         S1 ? "+upcat" : (up0 ? "+up" : ""), tl);
            l.flops = 2.0 * Y.h * Y.w * ks * ks * (double)cin * co + (dwo ? 2.0 * Y.h * Y.w * 9 * cin : 0.0);
            l.bytes = ((double)S0.h * S0.w * c0 + (S1 ? (double)S1->h * S1->w * c1 : 0.0) + (double)Y.h * Y.w * co) * 4;
        } else if (ty == YK_OP_DWCONV) {
            if (X.kind != XT_REAL || X.is_input) {
                yk_set_error("op %d: depthwise conv on a view/input", i);
                return fail(YK_ERR_UNSUPPORTED);
            }
            l.kind = XK_DW;
            xdw_args &d = l.d;
            memset(&d, 0, sizeof(d));
            const int c = X.c, cp = X.cp;
            std::vector<float> w((size_t)9 * cp, 0.f);
            for (int t = 0; t < 9; ++t)
                for (int k = 0; k < c; ++k) w[(size_t)t * cp + k] = blob[o[YK_F_W_OFF] + (size_t)t * c + k];
            void *dd;
            if ((rc = x_upload(p, &dd, w.data(), w.size() * sizeof(float)))) return fail(rc);
            d.in = X.d; d.Hi = X.h; d.Wi = X.w; d.Ho = Y.h; d.Wo = Y.w; d.Cp = cp;
            d.stride = o[YK_F_STRIDE]; d.pad_t = o[YK_F_PAD_T]; d.pad_l = o[YK_F_PAD_L];
            d.w = (const float *)dd;
            if ((rc = x_upload_f(p, blob + o[YK_F_SCALE_OFF], c, 1.f, &d.scale))) return fail(rc);
            if ((rc = x_upload_f(p, blob + o[YK_F_BIAS_OFF], c, 1.f, &d.bias))) return fail(rc);
            yk_act_params(o[YK_F_ACT], alpha, &d.slope, &d.cap);
            d.out = Y.d;
            d.amax_out = amax_of(yid);
            d.fd_g = yk_make_fastdiv((uint32_t)(cp >> 2));
            d.fd_wo = yk_make_fastdiv((uint32_t)Y.w);
            snprintf(nm, sizeof nm, "x:dw3x3s%d_%d", d.stride, c);
            l.flops = 2.0 * Y.h * Y.w * 9 * c;
            l.bytes = ((double)X.h * X.w * c + (double)Y.h * Y.w * c) * 4;
        } else if (ty == YK_OP_MAXPOOL) {
            if (X.kind != XT_REAL || X.is_input) {
                yk_set_error("op %d: max pool on a view/input", i);
                return fail(YK_ERR_UNSUPPORTED);
            }
            l.kind = XK_POOL;
            xpool_args &q = l.p;
            memset(&q, 0, sizeof(q));
            q.in = X.d; q.Hi = X.h; q.Wi = X.w; q.Ho = Y.h; q.Wo = Y.w; q.Cp = X.cp; q.stride = o[YK_F_STRIDE]; q.out = Y.d;
            q.amax_out = amax_of(yid);
            snprintf(nm, sizeof nm, "x:maxpool2x2s%d_%d", q.stride, X.c);
            l.bytes = ((double)X.h * X.w * X.c + (double)Y.h * Y.w * Y.c) * 4;
        } else if (ty == YK_OP_ADD) {
            const xtens &Z = p->T[o[YK_F_IN1]];
            if (X.kind != XT_REAL || Z.kind != XT_REAL || !X.d || !Z.d || !Y.d) {
                yk_set_error("op %d: standalone Add on views", i);
                return fail(YK_ERR_UNSUPPORTED);
            }
            l.kind = XK_ADD;
            l.add_a = X.d; l.add_b = Z.d; l.add_o = Y.d;
            l.add_amax = amax_of(yid);
            l.add_n4 = (size_t)Y.h * Y.w * Y.cp / 4;
            snprintf(nm, sizeof nm, "x:add_%d", Y.c);
            l.bytes = 3.0 * Y.h * Y.w * Y.c * 4;
        } else {
            yk_set_error("op %d: unknown op type %d", i, ty);
            return fail(YK_ERR_UNSUPPORTED);
        }
        l.name = nm;
        p->L.push_back(l);
    }
    for (int t : p->outputs)
        if (!p->T[t].d || !p->T[t].net_out) {
            yk_set_error("yk_plan_create: output tensor %d is not produced by a NET_OUTPUT conv", t);
            return fail(YK_ERR_UNSUPPORTED);
        }
    (void)blob_len;
    YK_HIP(hipDeviceSynchronize());
    *out = p;
    return YK_OK;
}

int yk_xplan_run(yk_xplan *p, const void *d_in, int in_f32, int batch, hipStream_t st, hipEvent_t *ev) {
    YK_HIP(hipMemsetAsync(p->d_amax, 0, sizeof(uint32_t) * p->T.size() * p->max_batch * XS, st));
    int li = 0;
    for (xlaunch &l : p->L) {
        if (ev) YK_HIP(hipEventRecord(ev[2 * li], st));
        switch (l.kind) {
        case XK_U8MAX:
            if (!in_f32) {
                int rc = yk_launch_u8_max((const uint8_t *)d_in, (size_t)p->in_h * p->in_w * 3, batch, p->d_imgmax, st);
                if (rc) return rc;
            }
            break;
        case XK_STEM: {
            xstem_args s = l.s;
            s.in = d_in; s.in_f32 = in_f32; s.img_max = p->d_imgmax; s.B = batch;
            dim3 grid((s.Ho * s.Wo + 255) / 256, batch);
            if (s.Cout == 16) hipLaunchKernelGGL(xstem_kernel<16>, grid, dim3(256), 0, st, s);
            else if (s.Cout == 24) hipLaunchKernelGGL(xstem_kernel<24>, grid, dim3(256), 0, st, s);
            else hipLaunchKernelGGL(xstem_kernel<32>, grid, dim3(256), 0, st, s);
        } break;
        case XK_CONV: {
            xconv_args g = l.c;
            g.B = batch;
            g.M = batch * l.Ho * l.Wo;
            const int per = XBM >> (g.sr_sh + g.sc_sh);
            dim3 grid((unsigned)(((long)batch * g.TXY + per - 1) / per), (unsigned)((g.N + l.bn - 1) / l.bn), (unsigned)g.splitk);
            for (int ph = g.splitk > 1 ? 1 : 0; ph <= (g.splitk > 1 ? 2 : 0); ++ph) {
                g.phase = ph;
                if (ph == 2) grid.z = 1;
                int rc = YK_OK;
                if (l.dw) rc = l.bn == 64 ? x_launch_conv<64, true>(g, grid, l.lds, st) : l.bn == 128 ? x_launch_conv<128, true>(g, grid, l.lds, st)
                                                                                                    : x_launch_conv<192, true>(g, grid, l.lds, st);
                else rc = l.bn == 64 ? x_launch_conv<64, false>(g, grid, l.lds, st) : l.bn == 128 ? x_launch_conv<128, false>(g, grid, l.lds, st)
                                                                                                   : x_launch_conv<192, false>(g, grid, l.lds, st);
                if (rc) return rc;
            }
        } break;
        case XK_DW: {
            xdw_args d = l.d;
            d.B = batch;
            const unsigned per_image = (unsigned)d.Ho * d.Wo * (d.Cp >> 2);
            hipLaunchKernelGGL(xdw_kernel, dim3((per_image + 255) / 256, batch), dim3(256), 0, st, d);
        } break;
        case XK_POOL: {
            xpool_args q = l.p;
            q.B = batch;
            const size_t total = (size_t)batch * q.Ho * q.Wo * (q.Cp >> 2);
            hipLaunchKernelGGL(xpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, q);
        } break;
        case XK_ADD:
            hipLaunchKernelGGL(xadd_kernel, dim3((unsigned)((l.add_n4 + 255) / 256), batch), dim3(256), 0, st, l.add_a, l.add_b, l.add_o,
                               l.add_n4, batch, l.add_amax);
            break;
        }
        if (ev) YK_HIP(hipEventRecord(ev[2 * li + 1], st));
        ++li;
    }
    YK_HIP(hipGetLastError());
    return YK_OK;
}

int yk_xplan_output(yk_xplan *p, int idx, float **d_ptr, size_t *bytes, int *h, int *w, int *c) {
    if (idx < 0 || idx >= (int)p->outputs.size()) return YK_ERR_ARG;
    const xtens &t = p->T[p->outputs[idx]];
    if (d_ptr) *d_ptr = t.d;
    if (bytes) *bytes = (size_t)p->max_batch * t.h * t.w * t.c * sizeof(float);
    if (h) *h = t.h;
    if (w) *w = t.w;
    if (c) *c = t.c;
    return YK_OK;
}

int yk_xplan_read_tensor(yk_xplan *p, int tid, int batch, float *h_dst, size_t dst_elems) {
    if (tid <= 0 || tid >= (int)p->T.size()) return YK_ERR_ARG;
    const xtens &t = p->T[tid];
    const size_t n = (size_t)batch * t.h * t.w * t.c;
    if (dst_elems < n) return YK_ERR_ARG;
    if (!t.d) {
        yk_set_error("yk_debug_read_tensor: tensor %d is a view or was folded away", tid);
        return YK_ERR_UNSUPPORTED;
    }
    YK_HIP(hipDeviceSynchronize());
    const int pitch = t.net_out ? t.c : t.cp;
    std::vector<float> hbuf((size_t)batch * t.h * t.w * pitch);
    YK_HIP(hipMemcpy(hbuf.data(), t.d, hbuf.size() * 4, hipMemcpyDeviceToHost));
    const size_t pix = (size_t)batch * t.h * t.w;
    for (size_t q = 0; q < pix; ++q) memcpy(h_dst + q * t.c, hbuf.data() + q * pitch, sizeof(float) * t.c);
    return YK_OK;
}

int yk_xplan_launch_count(const yk_xplan *p) { return (int)p->L.size(); }
int yk_xplan_launch_info(const yk_xplan *p, int i, const char **name, double *flops, double *bytes) {
    if (i < 0 || i >= (int)p->L.size()) return YK_ERR_ARG;
    *name = p->L[i].name.c_str();
    *flops = p->L[i].flops;
    *bytes = p->L[i].bytes;
    return YK_OK;
}
