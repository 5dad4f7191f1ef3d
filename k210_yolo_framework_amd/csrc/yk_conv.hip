// yk_conv.hip — gfx950 kernels of the conv stack (models/yolonet.py, keras_mobilenet*.py layers).
//
//  igemm_kernel   Conv2D 1x1 / 3x3 (stride 1|2, explicit top/left pad) as an implicit GEMM on
//                 v_mfma_f32_16x16x32_f16; optional virtual input = Concatenate([UpSampling2D(2)(a), b]);
//                 epilogue = folded BatchNorm (fp32 scale/bias) + LeakyReLU/ReLU/ReLU6 (+ residual Add),
//                 staged through LDS so global stores are 16 B per lane and row-contiguous.
//                 Operands are swapped (A := weights, B := pixels) so that each lane's four
//                 accumulator registers are four CONSECUTIVE output channels of one pixel.
//  first_conv     the 3-channel stem conv, reading u8 frames with Helper._process_img's
//                 `img / np.max(img)` fused in (per-image LUT), or fp32 input.
//  dw_kernel      DepthwiseConv2D 3x3, NHWC, 8 channels (16 B) per lane.
//  pool_kernel    MaxPooling2D 2x2 'same' (stride 2 and the stride-1 case of tiny_yolo).
//  u8_max_kernel  per-image max for the normalisation.
#include "yk_conv.h"

__device__ __forceinline__ float yk_act(float v, int act, float alpha) {
    switch (act) {
    case YK_ACT_RELU: return v > 0.f ? v : 0.f;
    case YK_ACT_RELU6: return v < 0.f ? 0.f : (v > 6.f ? 6.f : v);
    case YK_ACT_LEAKY: return v >= 0.f ? v : v * alpha;
    default: return v;
    }
}

// =====================================================================================
// implicit GEMM
// =====================================================================================
template <int BM, int BN, int WM, int WN, bool OUT_F32>
__global__ void __launch_bounds__(64 * WM * WN) igemm_kernel(const igemm_args a) {
    constexpr int NT = 64 * WM * WN;
    constexpr int BK = 32, LD = 40;                 // 80-byte LDS rows: 16 B aligned, spreads banks
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int A_VEC = BM * 4, B_VEC = BN * 4;   // 16-byte vectors per tile
    constexpr int A_IT = (A_VEC + NT - 1) / NT, B_IT = (B_VEC + NT - 1) / NT;
    constexpr int STAGE = (BM + BN) * LD;
    constexpr int CS_LD = BN + 8;
    constexpr int CS_HALFS = OUT_F32 ? 0 : BM * CS_LD;
    constexpr int LDS_HALFS = (2 * STAGE > CS_HALFS) ? 2 * STAGE : CS_HALFS;
    __shared__ __attribute__((aligned(16))) yk_half lds[LDS_HALFS];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kc = tid & 3;                         // this thread's 8-wide k chunk inside a BK step
    const int Ctp = a.c0p + a.c1p;
    const int taps = a.ks * a.ks;
    const int H0 = a.up0 ? (a.Hi >> 1) : a.Hi, W0 = a.up0 ? (a.Wi >> 1) : a.Wi;

    // ---- per-thread A rows (fixed over the K loop)
    int rb[A_IT], riy[A_IT], rix[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int v = tid + it * NT, row = v >> 2, m = m0 + row;
        if (v < A_VEC && m < a.M) {
            const int hw = a.Ho * a.Wo;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
            rb[it] = b;
            riy[it] = oy * a.stride - a.pad_t;
            rix[it] = ox * a.stride - a.pad_l;
        } else {
            rb[it] = 0;
            riy[it] = -(1 << 28);
            rix[it] = 0;
        }
    }
    int kch = kc * 8, ktap = 0;                     // (channel-in-tap, tap) of this thread's chunk
    while (kch >= Ctp) {
        kch -= Ctp;
        ++ktap;
    }

    half8 ra[A_IT], rbv[B_IT];
    auto gload = [&](int k0) {
        const int ky = (a.ks == 3) ? ktap / 3 : 0, kx = ktap - ky * a.ks;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            const int iy = riy[it] + ky, ix = rix[it] + kx;
            if (ktap < taps && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi) {
                const yk_half *p;
                if (kch < a.c0p) {
                    const int sy = a.up0 ? (iy >> 1) : iy, sx = a.up0 ? (ix >> 1) : ix;
                    p = a.in0 + ((size_t)(rb[it] * H0 + sy) * W0 + sx) * a.c0p + kch;
                } else {
                    p = a.in1 + ((size_t)(rb[it] * a.Hi + iy) * a.Wi + ix) * a.c1p + (kch - a.c0p);
                }
                v = *reinterpret_cast<const half8 *>(p);
            }
            ra[it] = v;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            const int vv = tid + it * NT, row = vv >> 2, n = n0 + row, k = k0 + kc * 8;
            if (vv < B_VEC && n < a.N && k < a.K) v = *reinterpret_cast<const half8 *>(a.w + (size_t)n * a.K + k);
            rbv[it] = v;
        }
        kch += BK;                                  // advance this thread's chunk to the next K step
        while (kch >= Ctp) {
            kch -= Ctp;
            ++ktap;
        }
    };
    auto sstore = [&](int stage) {
        yk_half *As = lds + stage * STAGE, *Bs = As + BM * LD;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int v = tid + it * NT;
            if (v < A_VEC) *reinterpret_cast<half8 *>(As + (v >> 2) * LD + kc * 8) = ra[it];
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int v = tid + it * NT;
            if (v < B_VEC) *reinterpret_cast<half8 *>(Bs + (v >> 2) * LD + kc * 8) = rbv[it];
        }
    };

    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nk = (a.K + BK - 1) / BK;
    gload(0);
    sstore(0);
    __syncthreads();
    int cur = 0;
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload((kt + 1) * BK);
        const yk_half *As = lds + cur * STAGE, *Bs = As + BM * LD;
        half8 wf[TN], xf[TM];
#pragma unroll
        for (int j = 0; j < TN; ++j)
            wf[j] = *reinterpret_cast<const half8 *>(Bs + ((wn * TN + j) * 16 + fr) * LD + fk);
#pragma unroll
        for (int i = 0; i < TM; ++i)
            xf[i] = *reinterpret_cast<const half8 *>(As + ((wm * TM + i) * 16 + fr) * LD + fk);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        if (kt + 1 < nk) sstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: lane holds channels n..n+3 (acc regs) of pixel m = lane&15
    const int nl4 = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = (wm * TM + i) * 16 + fr, m = m0 + ml;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = (wn * TN + j) * 16 + nl4, n = n0 + nl;
            const float4 sc = *reinterpret_cast<const float4 *>(a.scale + n);   // arrays padded past N with zeros
            const float4 bs = *reinterpret_cast<const float4 *>(a.bias + n);
            float v0 = yk_act(acc[i][j][0] * sc.x + bs.x, a.act, a.alpha);
            float v1 = yk_act(acc[i][j][1] * sc.y + bs.y, a.act, a.alpha);
            float v2 = yk_act(acc[i][j][2] * sc.z + bs.z, a.act, a.alpha);
            float v3 = yk_act(acc[i][j][3] * sc.w + bs.w, a.act, a.alpha);
            if constexpr (OUT_F32) {
                if (m < a.M) {
                    float *o = reinterpret_cast<float *>(a.out) + (size_t)m * a.outp + n;
                    if (n + 0 < a.N) o[0] = v0;
                    if (n + 1 < a.N) o[1] = v1;
                    if (n + 2 < a.N) o[2] = v2;
                    if (n + 3 < a.N) o[3] = v3;
                }
            } else {
                half4 h = {(yk_half)v0, (yk_half)v1, (yk_half)v2, (yk_half)v3};
                if (a.res && m < a.M && n < a.resp) {
                    // Add(inputs, x): the conv result is first rounded to its fp16 storage value
                    const half4 r = *reinterpret_cast<const half4 *>(a.res + (size_t)m * a.resp + n);
                    h = half4{(yk_half)((float)h[0] + (float)r[0]), (yk_half)((float)h[1] + (float)r[1]),
                              (yk_half)((float)h[2] + (float)r[2]), (yk_half)((float)h[3] + (float)r[3])};
                }
                *reinterpret_cast<half4 *>(lds + ml * CS_LD + nl) = h;
            }
        }
    }
    if constexpr (!OUT_F32) {
        __syncthreads();
        constexpr int VPR = BN / 8;
        yk_half *o = reinterpret_cast<yk_half *>(a.out);
        for (int v = tid; v < BM * VPR; v += NT) {
            const int row = v / VPR, cv = v - row * VPR, m = m0 + row, col = n0 + cv * 8;
            if (m < a.M && col < a.outp)
                *reinterpret_cast<half8 *>(o + (size_t)m * a.outp + col) =
                    *reinterpret_cast<const half8 *>(lds + row * CS_LD + cv * 8);
        }
    }
}

template <int BM, int BN, int WM, int WN, bool F32>
static int launch_cfg(const igemm_args &a, hipStream_t st) {
    dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN);
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, F32>), grid, dim3(64 * WM * WN), 0, st, a);
    return YK_OK;
}

int yk_launch_igemm(int cfg, const igemm_args &a, hipStream_t st) {
    switch (cfg) {
    case IGEMM_128x64: return launch_cfg<128, 64, 2, 2, false>(a, st);
    case IGEMM_128x48: return launch_cfg<128, 48, 4, 1, false>(a, st);
    case IGEMM_128x96: return launch_cfg<128, 96, 4, 1, false>(a, st);
    case IGEMM_128x192: return launch_cfg<128, 192, 4, 1, false>(a, st);
    case IGEMM_64x64: return launch_cfg<64, 64, 2, 2, false>(a, st);
    case IGEMM_128x128: return launch_cfg<128, 128, 2, 2, false>(a, st);
    case IGEMM_F32_128x80: return launch_cfg<128, 80, 4, 1, true>(a, st);
    case IGEMM_F32_128x64: return launch_cfg<128, 64, 2, 2, true>(a, st);
    }
    yk_set_error("yk_launch_igemm: bad config %d", cfg);
    return YK_ERR_ARG;
}

const char *yk_igemm_name(int cfg) {
    static const char *n[] = {"igemm_128x64", "igemm_128x48", "igemm_128x96", "igemm_128x192", "igemm_64x64",
                              "igemm_128x128", "igemm_f32_128x80", "igemm_f32_128x64"};
    return (cfg >= 0 && cfg < IGEMM_NUM) ? n[cfg] : "?";
}

int yk_igemm_pick(const igemm_args &a, bool out_f32) {
    if (out_f32) return a.N <= 80 ? IGEMM_F32_128x80 : IGEMM_F32_128x64;
    const long mt128 = (a.M + 127) / 128;
    if (a.N == 48) return IGEMM_128x48;
    if (a.N == 96) return IGEMM_128x96;
    if (a.N == 192 && mt128 >= 512) return IGEMM_128x192;
    if (a.N >= 128 && mt128 * ((a.N + 127) / 128) >= 512) return IGEMM_128x128;
    if (mt128 * ((a.N + 63) / 64) >= 256) return IGEMM_128x64;
    return IGEMM_64x64;
}

// =====================================================================================
// stem conv (Cin = 3)
// =====================================================================================
template <int COUT>
__global__ void __launch_bounds__(256) first_conv_kernel(const first_args a) {
    __shared__ __attribute__((aligned(16))) float wl[27 * COUT];
    __shared__ float sc[COUT], bs[COUT];
    __shared__ float lut[256];
    const int tid = threadIdx.x, b = blockIdx.y;
    for (int i = tid; i < 27 * COUT; i += 256) wl[i] = a.w[i];
    if (tid < COUT) {
        sc[tid] = a.scale[tid];
        bs[tid] = a.bias[tid];
    }
    if (!a.in_f32) lut[tid] = (float)tid / (float)a.img_max[b];     // img / np.max(img), tools/utils.py:405
    __syncthreads();
    const int pix = blockIdx.x * 256 + tid;
    if (pix >= a.Ho * a.Wo) return;
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
                for (int c = 0; c < COUT; ++c) acc[c] += x[ci] * w[c];
            }
        }
    }
    yk_half *o = a.out + ((size_t)b * a.Ho * a.Wo + pix) * a.outp;
#pragma unroll
    for (int c8 = 0; c8 < COUT; c8 += 8) {
        half8 h;
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (yk_half)yk_act(acc[c8 + j] * sc[c8 + j] + bs[c8 + j], a.act, a.alpha);
        *reinterpret_cast<half8 *>(o + c8) = h;
    }
}

int yk_launch_first(const first_args &a, hipStream_t st) {
    dim3 grid((a.Ho * a.Wo + 255) / 256, a.B);
    switch (a.Cout) {
    case 16: hipLaunchKernelGGL(first_conv_kernel<16>, grid, dim3(256), 0, st, a); break;
    case 24: hipLaunchKernelGGL(first_conv_kernel<24>, grid, dim3(256), 0, st, a); break;
    case 32: hipLaunchKernelGGL(first_conv_kernel<32>, grid, dim3(256), 0, st, a); break;
    default: yk_set_error("stem conv: Cout=%d unsupported (16/24/32)", a.Cout); return YK_ERR_UNSUPPORTED;
    }
    return YK_OK;
}

// per-image max of u8 frames -> img_max[b] (must be zeroed before the launch)
__global__ void __launch_bounds__(256) u8_max_kernel(const uint8_t *__restrict__ f, size_t per_image, int vec_ok,
                                                     unsigned *__restrict__ img_max) {
    const int b = blockIdx.y;
    const uint8_t *p = f + (size_t)b * per_image;
    unsigned m = 0;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    if (vec_ok) {
        const uint4 *q = reinterpret_cast<const uint4 *>(p);
        const size_t n16 = per_image / 16;
        for (size_t i = t; i < n16; i += nth) {
            const uint4 v = q[i];
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                m = max(m, max(max(w[j] & 0xffu, (w[j] >> 8) & 0xffu), max((w[j] >> 16) & 0xffu, w[j] >> 24)));
            }
        }
        for (size_t i = n16 * 16 + t; i < per_image; i += nth) m = max(m, (unsigned)p[i]);
    } else {
        for (size_t i = t; i < per_image; i += nth) m = max(m, (unsigned)p[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(img_max + b, m);
}

int yk_launch_u8_max(const uint8_t *frames, size_t per_image, int batch, unsigned *img_max, hipStream_t st) {
    const int vec_ok = (per_image % 16 == 0) && ((uintptr_t)frames % 16 == 0);
    int nblk = (int)((per_image / 16 + 255) / 256);
    if (nblk < 1) nblk = 1;
    if (nblk > 64) nblk = 64;
    hipLaunchKernelGGL(u8_max_kernel, dim3(nblk, batch), dim3(256), 0, st, frames, per_image, vec_ok, img_max);
    return YK_OK;
}

// =====================================================================================
// depthwise 3x3
// =====================================================================================
__global__ void __launch_bounds__(256) dw_kernel(const dw_args a) {
    const int G = a.Cp >> 3;
    const size_t total = (size_t)a.B * a.Ho * a.Wo * G;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int g = (int)(idx % G);
    const size_t pix = idx / G;
    const int ox = (int)(pix % a.Wo);
    const int oy = (int)((pix / a.Wo) % a.Ho);
    const int b = (int)(pix / ((size_t)a.Wo * a.Ho));
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int iy0 = oy * a.stride - a.pad_t, ix0 = ox * a.stride - a.pad_l;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = iy0 + ky;
        if ((unsigned)iy >= (unsigned)a.Hi) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ix0 + kx;
            if ((unsigned)ix >= (unsigned)a.Wi) continue;
            const half8 x = *reinterpret_cast<const half8 *>(a.in + ((size_t)(b * a.Hi + iy) * a.Wi + ix) * a.Cp + g * 8);
            const half8 w = *reinterpret_cast<const half8 *>(a.w + (size_t)(ky * 3 + kx) * a.Cp + g * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)x[j] * (float)w[j];
        }
    }
    half8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (yk_half)yk_act(acc[j] * a.scale[g * 8 + j] + a.bias[g * 8 + j], a.act, a.alpha);
    *reinterpret_cast<half8 *>(a.out + pix * a.Cp + g * 8) = h;
}

int yk_launch_dw(const dw_args &a, hipStream_t st) {
    const size_t total = (size_t)a.B * a.Ho * a.Wo * (a.Cp >> 3);
    hipLaunchKernelGGL(dw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
    return YK_OK;
}

// =====================================================================================
// max pool 2x2 'same'
// =====================================================================================
__global__ void __launch_bounds__(256) pool_kernel(const pool_args a) {
    const int G = a.Cp >> 3;
    const size_t total = (size_t)a.B * a.Ho * a.Wo * G;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int g = (int)(idx % G);
    const size_t pix = idx / G;
    const int ox = (int)(pix % a.Wo);
    const int oy = (int)((pix / a.Wo) % a.Ho);
    const int b = (int)(pix / ((size_t)a.Wo * a.Ho));
    half8 m;
    bool first = true;
#pragma unroll
    for (int ky = 0; ky < 2; ++ky)
#pragma unroll
        for (int kx = 0; kx < 2; ++kx) {
            const int iy = oy * a.stride + ky, ix = ox * a.stride + kx;
            if (iy >= a.Hi || ix >= a.Wi) continue;
            const half8 x = *reinterpret_cast<const half8 *>(a.in + ((size_t)(b * a.Hi + iy) * a.Wi + ix) * a.Cp + g * 8);
            if (first) {
                m = x;
                first = false;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = x[j] > m[j] ? x[j] : m[j];
            }
        }
    *reinterpret_cast<half8 *>(a.out + pix * a.Cp + g * 8) = m;
}

int yk_launch_pool(const pool_args &a, hipStream_t st) {
    const size_t total = (size_t)a.B * a.Ho * a.Wo * (a.Cp >> 3);
    hipLaunchKernelGGL(pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
    return YK_OK;
}

// =====================================================================================
// standalone residual add (fallback)
// =====================================================================================
__global__ void __launch_bounds__(256) add_kernel(const yk_half *a, const yk_half *b, yk_half *o, size_t n8) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const half8 x = reinterpret_cast<const half8 *>(a)[i], y = reinterpret_cast<const half8 *>(b)[i];
    half8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (yk_half)((float)x[j] + (float)y[j]);
    reinterpret_cast<half8 *>(o)[i] = r;
}
int yk_launch_add(const yk_half *a, const yk_half *b, yk_half *out, size_t n8, hipStream_t st) {
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, a, b, out, n8);
    return YK_OK;
}

// =====================================================================================
// fused depthwise 3x3 -> pointwise 1x1  (MobileNet block: keras_mobilenet.py:359-436,
// keras_mobilenet_v2.py:452-481).  One workgroup = BM consecutive output pixels x BN channels.
//   phase A: depthwise + BN + act for the BM x Cin tile, fp32 math, rounded to fp16 into LDS
//            (the stored value equals what the unfused pipeline would have written to HBM);
//   phase B: [BN x K] (weights, streamed L2 -> registers, double-buffered) x [K x BM] (LDS) on
//            v_mfma_f32_16x16x32_f16; epilogue as igemm_kernel.
// blockIdx.x is remapped so that consecutive pixel tiles run on the same XCD (shared halo rows
// stay in that XCD's L2).
// =====================================================================================
extern __shared__ __attribute__((aligned(16))) unsigned char yk_smem[];

template <int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(64 * WM * WN) fused_dwpw_kernel(const igemm_args a) {
    constexpr int NT = 64 * WM * WN;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int CS_LD = BN + 8;
    yk_half *As = reinterpret_cast<yk_half *>(yk_smem);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int Cp = a.c0p, Kp = (Cp + 31) & ~31, LDA = Kp + 8;

    // XCD-aware (bijective) tile remap
    const int nt = gridDim.x, bid = blockIdx.x;
    const int q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int m0 = tile * BM, n0 = blockIdx.y * BN;

    // ---------------- phase A: depthwise producer ----------------
    {
        const int G = Cp >> 3;
        const int PP = NT / G;                       // pixels per pass
        const int g = tid % G, pl = tid / G;
        if (pl < PP) {
            half8 w[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) w[t] = *reinterpret_cast<const half8 *>(a.dw_w + (size_t)t * Cp + g * 8);
            float sc[8], bs[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sc[j] = a.dw_scale[g * 8 + j];
                bs[j] = a.dw_bias[g * 8 + j];
            }
            const int hw = a.Ho * a.Wo;
            for (int p = pl; p < BM; p += PP) {
                const int m = m0 + p;
                half8 h = {0, 0, 0, 0, 0, 0, 0, 0};
                if (m < a.M) {
                    const int b = m / hw, rem = m - b * hw;
                    const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
                    const int iy0 = oy * a.dw_stride - a.dw_pad_t, ix0 = ox * a.dw_stride - a.dw_pad_l;
                    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int iy = iy0 + ky;
                        if ((unsigned)iy >= (unsigned)a.dw_Hi) continue;
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int ix = ix0 + kx;
                            if ((unsigned)ix >= (unsigned)a.dw_Wi) continue;
                            const half8 x = *reinterpret_cast<const half8 *>(
                                a.in0 + ((size_t)(b * a.dw_Hi + iy) * a.dw_Wi + ix) * Cp + g * 8);
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[j] += (float)x[j] * (float)w[ky * 3 + kx][j];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) h[j] = (yk_half)yk_act(acc[j] * sc[j] + bs[j], a.dw_act, 0.f);
                }
                *reinterpret_cast<half8 *>(As + p * LDA + g * 8) = h;
            }
        }
        // zero the K padding (Cp..Kp) once
        const int padv = (Kp - Cp) >> 3;
        for (int v = tid; v < BM * padv; v += NT) {
            const int p = v / padv, c = v - p * padv;
            *reinterpret_cast<half8 *>(As + p * LDA + Cp + c * 8) = half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    __syncthreads();

    // ---------------- phase B: GEMM, weights streamed from L2 ----------------
    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    const int nk = Kp >> 5;
    auto wload = [&](half8 (&wf)[TN], int k0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 16 + fr, k = k0 + fk;
            half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (n < a.N && k < a.K) v = *reinterpret_cast<const half8 *>(a.w + (size_t)n * a.K + k);
            wf[j] = v;
        }
    };
    half8 wcur[TN], wnext[TN];
    wload(wcur, 0);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) wload(wnext, (kt + 1) * 32);
        half8 xf[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i)
            xf[i] = *reinterpret_cast<const half8 *>(As + ((wm * TM + i) * 16 + fr) * LDA + kt * 32 + fk);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wcur[j], xf[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) wcur[j] = wnext[j];
    }
    __syncthreads();   // everyone is done with the A tile; reuse LDS for the output tile

    yk_half *Cs = reinterpret_cast<yk_half *>(yk_smem);
    const int nl4 = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = (wm * TM + i) * 16 + fr, m = m0 + ml;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = (wn * TN + j) * 16 + nl4, n = n0 + nl;
            const float4 sc = *reinterpret_cast<const float4 *>(a.scale + n);
            const float4 bs = *reinterpret_cast<const float4 *>(a.bias + n);
            half4 h = {(yk_half)yk_act(acc[i][j][0] * sc.x + bs.x, a.act, a.alpha),
                       (yk_half)yk_act(acc[i][j][1] * sc.y + bs.y, a.act, a.alpha),
                       (yk_half)yk_act(acc[i][j][2] * sc.z + bs.z, a.act, a.alpha),
                       (yk_half)yk_act(acc[i][j][3] * sc.w + bs.w, a.act, a.alpha)};
            if (a.res && m < a.M && n < a.resp) {
                const half4 rr = *reinterpret_cast<const half4 *>(a.res + (size_t)m * a.resp + n);
                h = half4{(yk_half)((float)h[0] + (float)rr[0]), (yk_half)((float)h[1] + (float)rr[1]),
                          (yk_half)((float)h[2] + (float)rr[2]), (yk_half)((float)h[3] + (float)rr[3])};
            }
            *reinterpret_cast<half4 *>(Cs + ml * CS_LD + nl) = h;
        }
    }
    __syncthreads();
    constexpr int VPR = BN / 8;
    yk_half *o = reinterpret_cast<yk_half *>(a.out);
    for (int v = tid; v < BM * VPR; v += NT) {
        const int row = v / VPR, cv = v - row * VPR, m = m0 + row, col = n0 + cv * 8;
        if (m < a.M && col < a.outp)
            *reinterpret_cast<half8 *>(o + (size_t)m * a.outp + col) = *reinterpret_cast<const half8 *>(Cs + row * CS_LD + cv * 8);
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_fused(const igemm_args &a, hipStream_t st) {
    const int Kp = (a.c0p + 31) & ~31;
    size_t lds = (size_t)BM * (Kp + 8) * 2, cs = (size_t)BM * (BN + 8) * 2;
    if (cs > lds) lds = cs;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_dwpw_kernel<BM, BN, WM, WN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN);
    hipLaunchKernelGGL((fused_dwpw_kernel<BM, BN, WM, WN>), grid, dim3(64 * WM * WN), lds, st, a);
    return YK_OK;
}

bool yk_igemm_fused_ok(int c0p, int cout) {
    // the whole K extent of the depthwise tile must fit LDS at the smallest BM (32 rows)
    const int Kp = (c0p + 31) & ~31;
    return c0p % 8 == 0 && (size_t)32 * (Kp + 8) * 2 <= 96 * 1024 && (256 / (c0p >> 3)) >= 1 && cout >= 8;
}
int yk_igemm_fused_pick(const igemm_args &a) {
    if (a.N <= 48) return FUSED_128x48;
    if (a.N <= 96) return FUSED_128x96;
    const long m64 = (a.M + 63) / 64;
    if (a.N <= 192 && m64 >= 256 && a.c0p <= 384) return FUSED_64x192;
    return FUSED_32x192;
}
const char *yk_igemm_fused_name(int cfg) {
    static const char *n[] = {"fused_128x48", "fused_128x96", "fused_64x192", "fused_32x192"};
    return (cfg >= 0 && cfg < FUSED_NUM) ? n[cfg] : "?";
}
int yk_launch_igemm_fused(int cfg, const igemm_args &a, hipStream_t st) {
    switch (cfg) {
    case FUSED_128x48: return launch_fused<128, 48, 4, 1>(a, st);
    case FUSED_128x96: return launch_fused<128, 96, 4, 1>(a, st);
    case FUSED_64x192: return launch_fused<64, 192, 2, 2>(a, st);
    case FUSED_32x192: return launch_fused<32, 192, 1, 4>(a, st);
    }
    yk_set_error("yk_launch_igemm_fused: bad config %d", cfg);
    return YK_ERR_ARG;
}
