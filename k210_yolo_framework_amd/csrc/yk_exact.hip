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
// fp32 on the VALU.  Same NetSpec, same C-ABI (yk_plan_create_ex(..., precision=1)), no fusion: this is the accuracy mode, the fp16
// plan is the throughput mode.  Results depend only on the image itself: the operand exponent is taken per image, never per batch.
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

struct xconv_args {
    const float *in0, *in1;
    int c0p, c1p, up0;
    int Hi, Wi, Ho, Wo, ks, stride, pad_t, pad_l;
    int M, N, K, HoWo;
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
};

constexpr int XBM = 64, XBN = 64, XBK = 32, XLD = XBK + 16;

// Conv2D 1x1 / 3x3 as an implicit GEMM with compensated fp16 operands.  256 threads = 2x2 waves, each a 32x32 output block.
__global__ void __launch_bounds__(256) xconv_kernel(const xconv_args a) {
    __shared__ __attribute__((aligned(16))) yk_half lds[2 * 4 * 64 * XLD];      // 2 stages x {A_hi, A_lo, B_hi, B_lo}
    __shared__ uint32_t simg[XBM];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = blockIdx.x * XBM, n0 = blockIdx.y * XBN;
    const int row = tid >> 2, kc = tid & 3;
    const int Ctp = a.c0p + a.c1p, taps = a.ks * a.ks;
    const int H0 = a.up0 ? (a.Hi >> 1) : a.Hi, W0 = a.up0 ? (a.Wi >> 1) : a.Wi;
    if (tid < XBM) simg[tid] = 0u;

    // this thread's A row (one output pixel) and its operand scale
    const int m = m0 + row;
    const bool mok = m < a.M;
    int rb = 0, riy = -(1 << 28), rix = 0;
    float sdown = 1.f;
    if (mok) {
        const uint32_t b = x_div(m, a.fd_hw), rem = m - b * a.HoWo;
        const uint32_t oy = x_div(rem, a.fd_wo), ox = rem - oy * a.Wo;
        rb = b;
        riy = (int)oy * a.stride - a.pad_t;
        rix = (int)ox * a.stride - a.pad_l;
        int e = x_exp_of(a.amax_in0[b]);
        if (a.in1) e = max(e, x_exp_of(a.amax_in1[b]));
        sdown = x_pow2(-e);
    }
    const int nrow = n0 + row;
    int kch = kc * 8, ktap = 0;
    float4 ra0, ra1;
    half8 rbh, rbl;
    auto gload = [&](int k0) {
        while (kch >= Ctp) {
            kch -= Ctp;
            ++ktap;
        }
        const int ky = (a.ks == 3) ? ktap / 3 : 0, kx = ktap - ky * a.ks;
        const int iy = riy + ky, ix = rix + kx;
        ra0 = ra1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ktap < taps && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi) {
            const float *p;
            if (kch < a.c0p) {
                const int sy = a.up0 ? (iy >> 1) : iy, sx = a.up0 ? (ix >> 1) : ix;
                p = a.in0 + ((size_t)(rb * H0 + sy) * W0 + sx) * a.c0p + kch;
            } else {
                p = a.in1 + ((size_t)(rb * a.Hi + iy) * a.Wi + ix) * a.c1p + (kch - a.c0p);
            }
            ra0 = *reinterpret_cast<const float4 *>(p);
            ra1 = *reinterpret_cast<const float4 *>(p + 4);
        }
        rbh = rbl = half8{0, 0, 0, 0, 0, 0, 0, 0};
        const int k = k0 + kc * 8;
        if (nrow < a.N && k < a.K) {
            rbh = *reinterpret_cast<const half8 *>(a.w_hi + (size_t)nrow * a.K + k);
            rbl = *reinterpret_cast<const half8 *>(a.w_lo + (size_t)nrow * a.K + k);
        }
        kch += XBK;
    };
    auto sstore = [&](int stage) {
        yk_half *S = lds + stage * (4 * 64 * XLD);
        const float v[8] = {ra0.x * sdown, ra0.y * sdown, ra0.z * sdown, ra0.w * sdown,
                            ra1.x * sdown, ra1.y * sdown, ra1.z * sdown, ra1.w * sdown};
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            hi[j] = (yk_half)v[j];
            lo[j] = (yk_half)(v[j] - (float)hi[j]);
        }
        *reinterpret_cast<half8 *>(S + row * XLD + kc * 8) = hi;
        *reinterpret_cast<half8 *>(S + 64 * XLD + row * XLD + kc * 8) = lo;
        *reinterpret_cast<half8 *>(S + 2 * 64 * XLD + row * XLD + kc * 8) = rbh;
        *reinterpret_cast<half8 *>(S + 3 * 64 * XLD + row * XLD + kc * 8) = rbl;
    };
    floatx4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    const int nk = (a.K + XBK - 1) / XBK;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload((kt + 1) * XBK);
        const yk_half *S = lds + (kt & 1) * (4 * 64 * XLD);
        half8 xh[2], xl[2], wh[2], wl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            xh[i] = *reinterpret_cast<const half8 *>(S + ((wm * 2 + i) * 16 + fr) * XLD + fk);
            xl[i] = *reinterpret_cast<const half8 *>(S + 64 * XLD + ((wm * 2 + i) * 16 + fr) * XLD + fk);
            wh[i] = *reinterpret_cast<const half8 *>(S + 2 * 64 * XLD + ((wn * 2 + i) * 16 + fr) * XLD + fk);
            wl[i] = *reinterpret_cast<const half8 *>(S + 3 * 64 * XLD + ((wn * 2 + i) * 16 + fr) * XLD + fk);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], xh[i], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xl[i], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xh[i], acc[i][j], 0, 0, 0);
            }
        if (kt + 1 < nk) sstore((kt + 1) & 1);
        __syncthreads();
    }
    // epilogue: lane holds channels n..n+3 of pixel (wm*2+i)*16 + fr
    const int nl4 = (lane >> 4) * 4;
    const uint32_t b0 = x_div(min(m0, a.M - 1), a.fd_hw);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int mm = m0 + (wm * 2 + i) * 16 + fr;
        if (mm >= a.M) continue;
        const uint32_t b = x_div(mm, a.fd_hw);
        int e = x_exp_of(a.amax_in0[b]);
        if (a.in1) e = max(e, x_exp_of(a.amax_in1[b]));
        const float up = x_pow2(e);
        float rmax = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + (wn * 2 + j) * 16 + nl4;
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
            } else if (n < a.outp) {
                if (a.res && n < a.resp) {
                    const float4 r = *reinterpret_cast<const float4 *>(a.res + (size_t)mm * a.resp + n);
                    v.x += r.x;
                    v.y += r.y;
                    v.z += r.z;
                    v.w += r.w;
                }
                *reinterpret_cast<float4 *>(a.out + (size_t)mm * a.outp + n) = v;
                rmax = fmaxf(rmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            }
        }
        if (a.amax_out) atomicMax(&simg[b - b0], __float_as_uint(rmax));
    }
    if (a.amax_out) {
        __syncthreads();
        if (tid < XBM && simg[tid]) atomicMax(a.amax_out + b0 + tid, simg[tid]);
    }
}

// ---- depthwise 3x3, fp32, one thread = (pixel, 4 channels) -------------------------------------------------
struct xdw_args {
    const float *in;
    int B, Hi, Wi, Ho, Wo, Cp, stride, pad_t, pad_l;
    const float *w;                    // [9][Cp] fp32
    const float *scale, *bias;
    float slope, cap;
    float *out;
    uint32_t *amax_out;
};
__global__ void __launch_bounds__(256) xdw_kernel(const xdw_args a) {
    __shared__ uint32_t simg[256];
    const int tid = threadIdx.x;
    simg[tid] = 0u;
    __syncthreads();
    const int G = a.Cp >> 2;
    const size_t total = (size_t)a.B * a.Ho * a.Wo * G;
    const size_t first = (size_t)blockIdx.x * 256;
    const int b0 = (int)((first / G) / ((size_t)a.Ho * a.Wo));
    const size_t idx = first + tid;
    if (idx < total) {
        const int g = (int)(idx % G);
        const size_t pix = idx / G;
        const int ox = (int)(pix % a.Wo), oy = (int)((pix / a.Wo) % a.Ho), b = (int)(pix / ((size_t)a.Wo * a.Ho));
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int iy0 = oy * a.stride - a.pad_t, ix0 = ox * a.stride - a.pad_l;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = iy0 + ky;
            if ((unsigned)iy >= (unsigned)a.Hi) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ix0 + kx;
                if ((unsigned)ix >= (unsigned)a.Wi) continue;
                const float4 x = *reinterpret_cast<const float4 *>(a.in + ((size_t)(b * a.Hi + iy) * a.Wi + ix) * a.Cp + g * 4);
                const float4 w = *reinterpret_cast<const float4 *>(a.w + (size_t)(ky * 3 + kx) * a.Cp + g * 4);
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
        *reinterpret_cast<float4 *>(a.out + pix * a.Cp + g * 4) = v;
        const float mx = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        atomicMax(&simg[min(b - b0, 255)], __float_as_uint(mx));
    }
    __syncthreads();
    if (simg[tid] && b0 + tid < a.B) atomicMax(a.amax_out + b0 + tid, simg[tid]);
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
    __shared__ __attribute__((aligned(16))) float wl[27 * COUT];
    __shared__ float sc[COUT], bs[COUT];
    __shared__ float lut[256];
    __shared__ uint32_t smax;
    const int tid = threadIdx.x, b = blockIdx.y;
    for (int i = tid; i < 27 * COUT; i += 256) wl[i] = a.w[i];
    if (tid < COUT) {
        sc[tid] = a.scale[tid];
        bs[tid] = a.bias[tid];
    }
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
    if (tid == 0 && smax) atomicMax(a.amax_out + b, smax);
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
        const float mx = fmaxf(fmaxf(fabsf(m.x), fabsf(m.y)), fmaxf(fabsf(m.z), fabsf(m.w)));
        atomicMax(&simg[min(b - b0, 255)], __float_as_uint(mx));
    }
    __syncthreads();
    if (simg[tid] && b0 + tid < a.B) atomicMax(a.amax_out + b0 + tid, simg[tid]);
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
    if (threadIdx.x == 0 && smax) atomicMax(amax_out + b, smax);
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
    std::string name;
    double flops = 0, bytes = 0;
};

}   // namespace

struct yk_xplan {
    int max_batch = 0, in_h = 0, in_w = 0;
    std::vector<xtens> T;
    std::vector<xlaunch> L;
    std::vector<void *> allocs;
    std::vector<int> outputs;
    unsigned *d_imgmax = nullptr;
    uint32_t *d_amax = nullptr;        // [n_tensors][max_batch]
    uint32_t *d_one = nullptr;         // [max_batch] bits of 1.0f (the normalised image)
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
    for (int i = 1; i < n_tensors; ++i) {
        xtens &t = p->T[i];
        if (t.kind != XT_REAL) continue;
        bool folded = false;
        for (int k = 0; k < n_ops; ++k)
            if (ops[(size_t)k * YK_OP_FIELDS + YK_F_OUT] == i && add_of[k] >= 0) folded = true;
        if (folded) continue;
        const size_t pitch = t.net_out ? t.c : t.cp;
        if ((rc = x_alloc(p, (void **)&t.d, ((size_t)max_batch * t.h * t.w * pitch + 64) * sizeof(float)))) return fail(rc);
    }
    if ((rc = x_alloc(p, (void **)&p->d_imgmax, sizeof(unsigned) * max_batch * 32))) return fail(rc);
    if ((rc = x_alloc(p, (void **)&p->d_amax, sizeof(uint32_t) * (size_t)n_tensors * max_batch))) return fail(rc);
    {
        std::vector<uint32_t> one(max_batch, 0x3f800000u);
        void *q;
        if ((rc = x_upload(p, &q, one.data(), one.size() * 4))) return fail(rc);
        p->d_one = (uint32_t *)q;
    }
    auto amax_of = [&](int tid) { return p->d_amax + (size_t)tid * max_batch; };
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
            snprintf(nm, sizeof nm, "x:conv%dx%ds%d_%dto%d%s%s", ks, ks, g.stride, cin, co, g.res ? "+add" : "",
                     S1 ? "+upcat" : (up0 ? "+up" : ""));
            l.flops = 2.0 * Y.h * Y.w * ks * ks * (double)cin * co;
            l.bytes = ((double)X.h * X.w * cin + (double)Y.h * Y.w * co) * 4;
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
    YK_HIP(hipMemsetAsync(p->d_amax, 0, sizeof(uint32_t) * p->T.size() * p->max_batch, st));
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
            g.M = batch * l.Ho * l.Wo;
            hipLaunchKernelGGL(xconv_kernel, dim3((g.M + XBM - 1) / XBM, (g.N + XBN - 1) / XBN), dim3(256), 0, st, g);
        } break;
        case XK_DW: {
            xdw_args d = l.d;
            d.B = batch;
            const size_t total = (size_t)batch * d.Ho * d.Wo * (d.Cp >> 2);
            hipLaunchKernelGGL(xdw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d);
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
