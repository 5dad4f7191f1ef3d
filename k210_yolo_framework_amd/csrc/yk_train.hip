// yk_train.hip — fp32 building blocks of the training step (keras_train.py:73-98 —
// forward in training mode, TF autodiff, Adam) on gfx950.  The reference obtains all of this from TensorFlow
// (un-vendored); here every arithmetic op of the step is a HIP kernel behind the C-ABI, orchestrated by
// k210_yolo_framework_amd/train.py.  Training keeps fp32 storage and fp32 MFMA (v_mfma_f32_16x16x4_f32: exact
// fp32 FMA chains) — at 16 images per GPU the step is far from any roofline and parity with an fp32 autograd
// oracle is what matters first.
//
//   yk_gemm_f32            C = alpha * op(A) * op(B) + beta * C  (row-major, any shape; optional split-K)
//                          -> 1x1 conv forward / data gradient / weight gradient, and 3x3 convs through
//   yk_im2col3x3_f32 / yk_col2im3x3_f32
//   yk_dw3x3_{fwd,bwd_data,bwd_weight}_f32      DepthwiseConv2D
//   yk_bn_train_fwd_f32 / yk_bn_train_bwd_f32   BatchNormalization in training mode fused with the activation
//   yk_upsample2x_bwd_f32, yk_axpy_f32, yk_adam_f32 (Keras Adam incl. `decay`, keras_train.py:74-76)
#include "yk_common.h"
#include <algorithm>
#include <cmath>

typedef float floatx4t __attribute__((ext_vector_type(4)));

// --------------------------------------------------------------------------------------------------------
// GEMM: 64x64 tile per workgroup (2x2 waves, each 32x32 = 2x2 MFMA 16x16x4 f32 tiles), BK = 16.
// LDS holds both operands k-major (As[k][m], Bs[k][n]) so a wave's fragment read is 16 consecutive floats.
// --------------------------------------------------------------------------------------------------------
struct gemm_args {
    int M, N, K, lda, ldb, ldc, transA, transB, splitk;
    float alpha, beta;
    const float *A, *B;
    float *C, *ws;
};

__global__ void __launch_bounds__(256) gemm_f32_kernel(const gemm_args g) {
    constexpr int BM = 64, BN = 64, BK = 16, LDT = 68;
    __shared__ float As[BK][LDT], Bs[BK][LDT];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int nk = (g.K + BK - 1) / BK;
    const int per = (nk + g.splitk - 1) / g.splitk;
    const int kb = blockIdx.z * per, ke = min(nk, kb + per);
    floatx4t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = floatx4t{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = kb; kt < ke; ++kt) {
        const int k0 = kt * BK;
        // each thread brings 4 elements of A and 4 of B; the index split follows the contiguous axis
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 256;
            int mm, kk;
            if (g.transA) { kk = e >> 6; mm = e & 63; } else { mm = e >> 4; kk = e & 15; }
            const int m = m0 + mm, k = k0 + kk;
            float v = 0.f;
            if (m < g.M && k < g.K) v = g.transA ? g.A[(size_t)k * g.lda + m] : g.A[(size_t)m * g.lda + k];
            As[kk][mm] = v;
            int nn, kb2;
            if (g.transB) { nn = e >> 4; kb2 = e & 15; } else { kb2 = e >> 6; nn = e & 63; }
            const int n = n0 + nn, k2 = k0 + kb2;
            float w = 0.f;
            if (n < g.N && k2 < g.K) w = g.transB ? g.B[(size_t)n * g.ldb + k2] : g.B[(size_t)k2 * g.ldb + n];
            Bs[kb2][nn] = w;
        }
        __syncthreads();
#pragma unroll
        for (int k4 = 0; k4 < BK; k4 += 4) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[k4 + fq][wm * 32 + i * 16 + fr];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[k4 + fq][wn * 32 + j * 16 + fr];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // D layout: row = (lane>>4)*4 + r (A index = m), col = lane&15 (B index = n)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 32 + i * 16 + fq * 4 + r, n = n0 + wn * 32 + j * 16 + fr;
                if (m < g.M && n < g.N) {
                    if (g.splitk > 1) {                      // deterministic split-K: one slab per split, summed in order
                        g.ws[((size_t)blockIdx.z * g.M + m) * g.N + n] = acc[i][j][r];
                    } else {
                        float *c = g.C + (size_t)m * g.ldc + n;
                        *c = g.alpha * acc[i][j][r] + (g.beta != 0.f ? g.beta * *c : 0.f);
                    }
                }
            }
}

__global__ void __launch_bounds__(256) splitk_sum_kernel(const float *__restrict__ ws, int splits, int M, int N, float alpha, float beta,
                                                         float *__restrict__ c, int ldc) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)M * N) return;
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += ws[(size_t)z * M * N + i];
    const size_t r = i / N, q = i - r * N;
    float *o = c + r * ldc + q;
    *o = alpha * s + (beta != 0.f ? beta * *o : 0.f);
}

extern "C" int yk_gemm_f32(int transA, int transB, int M, int N, int K, float alpha, const float *A, int lda, const float *B,
                           int ldb, float beta, float *C, int ldc, void *stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) {
        yk_set_error("yk_gemm_f32: bad argument");
        return YK_ERR_ARG;
    }
    gemm_args g;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.transA = transA; g.transB = transB;
    g.alpha = alpha; g.beta = beta; g.A = A; g.B = B; g.C = C;
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    const int nk = (K + 15) / 16;
    int s = 1;
    if (tiles < 256 && nk >= 64) {            // weight gradients: tiny output, reduction over all pixels
        s = (int)std::min<long>((512 + tiles - 1) / tiles, nk / 16);
        if (s < 1) s = 1;
        if (s > 256) s = 256;
    }
    g.splitk = s;
    g.ws = nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (s > 1) {
        int dev = yk_current_device();
        if (dev < 0) return YK_ERR_NO_DEVICE;
        g.ws = (float *)yk_scratch(dev, stream, 14, sizeof(float) * (size_t)s * M * N);
        if (!g.ws) return YK_ERR_NOMEM;
    }
    hipLaunchKernelGGL(gemm_f32_kernel, dim3((M + 63) / 64, (N + 63) / 64, s), dim3(256), 0, st, g);
    if (s > 1) {
        const size_t tot = (size_t)M * N;
        hipLaunchKernelGGL(splitk_sum_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float *)g.ws, s, M, N, alpha, beta, C,
                           ldc);
    }
    YK_HIP(hipGetLastError());
    return YK_OK;
}

// --------------------------------------------------------------------------------------------------------
// im2col / col2im for 3x3 convs, NHWC.  col: [B*Ho*Wo][9*C] with k = (ky*3+kx)*C + c (matches OHWI weights).
// --------------------------------------------------------------------------------------------------------
struct conv_geom {
    int B, Hi, Wi, C, Ho, Wo, stride, pad_t, pad_l;
};

__global__ void __launch_bounds__(256) im2col3x3_kernel(conv_geom q, const float *__restrict__ x, float *__restrict__ col) {
    const size_t total = (size_t)q.B * q.Ho * q.Wo * 9 * q.C;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % q.C);
    const int t = (int)((i / q.C) % 9);
    const size_t m = i / ((size_t)9 * q.C);
    const int ox = (int)(m % q.Wo), oy = (int)((m / q.Wo) % q.Ho), b = (int)(m / ((size_t)q.Wo * q.Ho));
    const int iy = oy * q.stride - q.pad_t + t / 3, ix = ox * q.stride - q.pad_l + t % 3;
    float v = 0.f;
    if ((unsigned)iy < (unsigned)q.Hi && (unsigned)ix < (unsigned)q.Wi) v = x[(((size_t)b * q.Hi + iy) * q.Wi + ix) * q.C + c];
    col[i] = v;
}

// dx[b,iy,ix,c] = sum over the (oy,ox,tap) that read it — a gather, hence deterministic
__global__ void __launch_bounds__(256) col2im3x3_kernel(conv_geom q, const float *__restrict__ col, float *__restrict__ dx) {
    const size_t total = (size_t)q.B * q.Hi * q.Wi * q.C;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % q.C);
    const size_t p = i / q.C;
    const int ix = (int)(p % q.Wi), iy = (int)((p / q.Wi) % q.Hi), b = (int)(p / ((size_t)q.Wi * q.Hi));
    float s = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int ny = iy + q.pad_t - ky;
        if (ny < 0 || ny % q.stride) continue;
        const int oy = ny / q.stride;
        if (oy >= q.Ho) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int nx = ix + q.pad_l - kx;
            if (nx < 0 || nx % q.stride) continue;
            const int ox = nx / q.stride;
            if (ox >= q.Wo) continue;
            s += col[((((size_t)b * q.Ho + oy) * q.Wo + ox) * 9 + (ky * 3 + kx)) * q.C + c];
        }
    }
    dx[i] = s;
}

extern "C" int yk_im2col3x3_f32(const float *x, int B, int Hi, int Wi, int C, int Ho, int Wo, int stride, int pad_t, int pad_l,
                                float *col, void *stream) {
    conv_geom q = {B, Hi, Wi, C, Ho, Wo, stride, pad_t, pad_l};
    const size_t total = (size_t)B * Ho * Wo * 9 * C;
    hipLaunchKernelGGL(im2col3x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q, x, col);
    YK_HIP(hipGetLastError());
    return YK_OK;
}
extern "C" int yk_col2im3x3_f32(const float *col, int B, int Hi, int Wi, int C, int Ho, int Wo, int stride, int pad_t, int pad_l,
                                float *dx, void *stream) {
    conv_geom q = {B, Hi, Wi, C, Ho, Wo, stride, pad_t, pad_l};
    const size_t total = (size_t)B * Hi * Wi * C;
    hipLaunchKernelGGL(col2im3x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q, col, dx);
    YK_HIP(hipGetLastError());
    return YK_OK;
}

// --------------------------------------------------------------------------------------------------------
// depthwise 3x3, NHWC fp32; weights [9][C]
// --------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) dw_fwd_kernel(conv_geom q, const float *__restrict__ x, const float *__restrict__ w,
                                                     float *__restrict__ y) {
    const size_t total = (size_t)q.B * q.Ho * q.Wo * q.C;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % q.C);
    const size_t m = i / q.C;
    const int ox = (int)(m % q.Wo), oy = (int)((m / q.Wo) % q.Ho), b = (int)(m / ((size_t)q.Wo * q.Ho));
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int iy = oy * q.stride - q.pad_t + t / 3, ix = ox * q.stride - q.pad_l + t % 3;
        if ((unsigned)iy < (unsigned)q.Hi && (unsigned)ix < (unsigned)q.Wi)
            s += x[(((size_t)b * q.Hi + iy) * q.Wi + ix) * q.C + c] * w[t * q.C + c];
    }
    y[i] = s;
}

__global__ void __launch_bounds__(256) dw_bwd_data_kernel(conv_geom q, const float *__restrict__ dy, const float *__restrict__ w,
                                                          float *__restrict__ dx) {
    const size_t total = (size_t)q.B * q.Hi * q.Wi * q.C;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % q.C);
    const size_t p = i / q.C;
    const int ix = (int)(p % q.Wi), iy = (int)((p / q.Wi) % q.Hi), b = (int)(p / ((size_t)q.Wi * q.Hi));
    float s = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int ny = iy + q.pad_t - ky;
        if (ny < 0 || ny % q.stride) continue;
        const int oy = ny / q.stride;
        if (oy >= q.Ho) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int nx = ix + q.pad_l - kx;
            if (nx < 0 || nx % q.stride) continue;
            const int ox = nx / q.stride;
            if (ox >= q.Wo) continue;
            s += dy[(((size_t)b * q.Ho + oy) * q.Wo + ox) * q.C + c] * w[(ky * 3 + kx) * q.C + c];
        }
    }
    dx[i] = s;
}

// dw[t][c] = sum_{b,oy,ox} dy * x(tap t).  grid (chunks, ceil(C/64)), block (64 channels x 4 row-lanes);
// partial[chunk][9][C] then summed in chunk order by dw_bwd_weight_finish (deterministic).
__global__ void __launch_bounds__(256) dw_bwd_weight_kernel(conv_geom q, const float *__restrict__ x, const float *__restrict__ dy,
                                                            float *__restrict__ partial, int rows_per_chunk) {
    __shared__ float red[9][4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const size_t M = (size_t)q.B * q.Ho * q.Wo;
    const size_t r0 = (size_t)blockIdx.x * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    float s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (c < q.C)
        for (size_t m = r0 + rl; m < r1; m += 4) {
            const int ox = (int)(m % q.Wo), oy = (int)((m / q.Wo) % q.Ho), b = (int)(m / ((size_t)q.Wo * q.Ho));
            const float g = dy[m * q.C + c];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = oy * q.stride - q.pad_t + t / 3, ix = ox * q.stride - q.pad_l + t % 3;
                if ((unsigned)iy < (unsigned)q.Hi && (unsigned)ix < (unsigned)q.Wi)
                    s[t] += g * x[(((size_t)b * q.Hi + iy) * q.Wi + ix) * q.C + c];
            }
        }
#pragma unroll
    for (int t = 0; t < 9; ++t) red[t][rl][cl] = s[t];
    __syncthreads();
    if (rl == 0 && c < q.C)
#pragma unroll
        for (int t = 0; t < 9; ++t)
            partial[((size_t)blockIdx.x * 9 + t) * q.C + c] = red[t][0][cl] + red[t][1][cl] + red[t][2][cl] + red[t][3][cl];
}
__global__ void __launch_bounds__(256) colsum_finish_kernel(const float *__restrict__ partial, int chunks, int n, float *__restrict__ out,
                                                            float scale) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += partial[(size_t)k * n + i];
    out[i] = s * scale;
}

static int chunking(size_t M, int *rows_per_chunk) {
    int chunks = (int)std::min<size_t>(512, (M + 255) / 256);
    if (chunks < 1) chunks = 1;
    *rows_per_chunk = (int)((M + chunks - 1) / chunks);
    return (int)((M + *rows_per_chunk - 1) / *rows_per_chunk);
}

extern "C" int yk_dw3x3_fwd_f32(const float *x, const float *w, int B, int Hi, int Wi, int C, int Ho, int Wo, int stride, int pad_t,
                                int pad_l, float *y, void *stream) {
    conv_geom q = {B, Hi, Wi, C, Ho, Wo, stride, pad_t, pad_l};
    const size_t total = (size_t)B * Ho * Wo * C;
    hipLaunchKernelGGL(dw_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q, x, w, y);
    YK_HIP(hipGetLastError());
    return YK_OK;
}
extern "C" int yk_dw3x3_bwd_data_f32(const float *dy, const float *w, int B, int Hi, int Wi, int C, int Ho, int Wo, int stride,
                                     int pad_t, int pad_l, float *dx, void *stream) {
    conv_geom q = {B, Hi, Wi, C, Ho, Wo, stride, pad_t, pad_l};
    const size_t total = (size_t)B * Hi * Wi * C;
    hipLaunchKernelGGL(dw_bwd_data_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q, dy, w, dx);
    YK_HIP(hipGetLastError());
    return YK_OK;
}
extern "C" int yk_dw3x3_bwd_weight_f32(const float *x, const float *dy, int B, int Hi, int Wi, int C, int Ho, int Wo, int stride,
                                       int pad_t, int pad_l, float *dw, void *stream) {
    conv_geom q = {B, Hi, Wi, C, Ho, Wo, stride, pad_t, pad_l};
    int dev = yk_current_device();
    if (dev < 0) return YK_ERR_NO_DEVICE;
    int rpc;
    const int chunks = chunking((size_t)B * Ho * Wo, &rpc);
    float *partial = (float *)yk_scratch(dev, stream, 12, sizeof(float) * (size_t)chunks * 9 * C);
    if (!partial) return YK_ERR_NOMEM;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dw_bwd_weight_kernel, dim3(chunks, (C + 63) / 64), dim3(256), 0, st, q, x, dy, partial, rpc);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((9 * C + 255) / 256), dim3(256), 0, st, partial, chunks, 9 * C, dw, 1.f);
    YK_HIP(hipGetLastError());
    return YK_OK;
}

// --------------------------------------------------------------------------------------------------------
// BatchNormalization, training mode, fused with the activation.  z: [M][C] conv output.
//   fwd:  mean/var over M (two passes), y = act(gamma * (z - mean) * invstd + beta); saves mean, invstd
//   bwd:  g = dy * act'(.), dbeta = sum g, dgamma = sum g*xhat, dz = gamma*invstd*(g - dbeta/M - xhat*dgamma/M)
// activation codes as in yolo_hip.h (none / relu / relu6 / leaky alpha)
// --------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float t_act(float v, int act, float alpha) {
    if (act == YK_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == YK_ACT_RELU6) return v < 0.f ? 0.f : (v > 6.f ? 6.f : v);
    if (act == YK_ACT_LEAKY) return v >= 0.f ? v : v * alpha;
    return v;
}
__device__ __forceinline__ float t_act_grad(float pre, int act, float alpha) {   // derivative at the pre-activation value
    if (act == YK_ACT_RELU) return pre > 0.f ? 1.f : 0.f;
    if (act == YK_ACT_RELU6) return (pre > 0.f && pre < 6.f) ? 1.f : 0.f;
    if (act == YK_ACT_LEAKY) return pre >= 0.f ? 1.f : alpha;
    return 1.f;
}

// mode 0: sum z ; mode 1: sum (z-mean)^2 ; mode 2: sums of g and g*xhat (two outputs)
__global__ void __launch_bounds__(256) bn_colreduce_kernel(int mode, const float *__restrict__ z, const float *__restrict__ dy, size_t M,
                                                           int C, int rows_per_chunk, const float *__restrict__ mean,
                                                           const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, int act, float alpha,
                                                           float *__restrict__ partial) {
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const size_t r0 = (size_t)blockIdx.x * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        const float mu = mode ? mean[c] : 0.f;
        const float is = mode == 2 ? invstd[c] : 0.f, ga = mode == 2 ? gamma[c] : 0.f, be = mode == 2 ? beta[c] : 0.f;
        for (size_t m = r0 + rl; m < r1; m += 4) {
            const float v = z[m * C + c];
            if (mode == 0) s0 += v;
            else if (mode == 1) s0 += (v - mu) * (v - mu);
            else {
                const float xh = (v - mu) * is;
                const float g = dy[m * C + c] * t_act_grad(ga * xh + be, act, alpha);
                s0 += g;
                s1 += g * xh;
            }
        }
    }
    red[0][rl][cl] = s0;
    red[1][rl][cl] = s1;
    __syncthreads();
    if (rl == 0 && c < C) {
        partial[((size_t)blockIdx.x * 2 + 0) * C + c] = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
        partial[((size_t)blockIdx.x * 2 + 1) * C + c] = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
    }
}
// out0[c] = scale * sum_k partial[k][0][c] ; out1[c] likewise (when out1 != null); optional var -> invstd
__global__ void __launch_bounds__(256) bn_finish_kernel(const float *__restrict__ partial, int chunks, int C, float scale, float eps,
                                                        int to_invstd, float *__restrict__ out0, float *__restrict__ out1,
                                                        float *__restrict__ raw0) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, b = 0.f;
    for (int k = 0; k < chunks; ++k) {
        a += partial[((size_t)k * 2 + 0) * C + c];
        b += partial[((size_t)k * 2 + 1) * C + c];
    }
    if (raw0) raw0[c] = a * scale;                         // biased variance, for the moving statistics
    out0[c] = to_invstd ? 1.f / sqrtf(a * scale + eps) : a * scale;
    if (out1) out1[c] = b * scale;
}
__global__ void __launch_bounds__(256) bn_apply_fwd_kernel(const float *__restrict__ z, size_t total, int C, const float *__restrict__ mean,
                                                           const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, int act, float alpha, float *__restrict__ y) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    y[i] = t_act(gamma[c] * (z[i] - mean[c]) * invstd[c] + beta[c], act, alpha);
}
__global__ void __launch_bounds__(256) bn_apply_bwd_kernel(const float *__restrict__ z, const float *__restrict__ dy, size_t total, int C,
                                                           float invM, const float *__restrict__ mean, const float *__restrict__ invstd,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           const float *__restrict__ dbeta, const float *__restrict__ dgamma, int act,
                                                           float alpha, float *__restrict__ dz) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const float xh = (z[i] - mean[c]) * invstd[c];
    const float g = dy[i] * t_act_grad(gamma[c] * xh + beta[c], act, alpha);
    dz[i] = gamma[c] * invstd[c] * (g - dbeta[c] * invM - xh * dgamma[c] * invM);
}
__global__ void __launch_bounds__(256) moving_update_kernel(float *mm, float *mv, const float *mean, const float *var, int C, float mom) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    mm[c] = mm[c] * mom + mean[c] * (1.f - mom);           // keras: moving = moving*momentum + batch*(1-momentum)
    mv[c] = mv[c] * mom + var[c] * (1.f - mom);
}

extern "C" int yk_bn_train_fwd_f32(const float *z, long long M, int C, const float *gamma, const float *beta, float eps, int act,
                                   float alpha, float *y, float *save_mean, float *save_invstd, float *moving_mean,
                                   float *moving_var, float momentum, void *stream) {
    int dev = yk_current_device();
    if (dev < 0) return YK_ERR_NO_DEVICE;
    int rpc;
    const int chunks = chunking((size_t)M, &rpc);
    float *partial = (float *)yk_scratch(dev, stream, 13, sizeof(float) * ((size_t)chunks * 2 * C + C));
    if (!partial) return YK_ERR_NOMEM;
    float *var = partial + (size_t)chunks * 2 * C;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(chunks, (C + 63) / 64);
    hipLaunchKernelGGL(bn_colreduce_kernel, grid, dim3(256), 0, st, 0, z, (const float *)nullptr, (size_t)M, C, rpc, (const float *)nullptr,
                       (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, 0, 0.f, partial);
    hipLaunchKernelGGL(bn_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, st, partial, chunks, C, 1.f / (float)M, 0.f, 0, save_mean,
                       (float *)nullptr, (float *)nullptr);
    hipLaunchKernelGGL(bn_colreduce_kernel, grid, dim3(256), 0, st, 1, z, (const float *)nullptr, (size_t)M, C, rpc, (const float *)save_mean,
                       (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, 0, 0.f, partial);
    hipLaunchKernelGGL(bn_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, st, partial, chunks, C, 1.f / (float)M, eps, 1, save_invstd,
                       (float *)nullptr, var);
    const size_t total = (size_t)M * C;
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, z, total, C, (const float *)save_mean,
                       (const float *)save_invstd, gamma, beta, act, alpha, y);
    if (moving_mean && moving_var)
        hipLaunchKernelGGL(moving_update_kernel, dim3((C + 255) / 256), dim3(256), 0, st, moving_mean, moving_var, (const float *)save_mean,
                           (const float *)var, C, momentum);
    YK_HIP(hipGetLastError());
    return YK_OK;
}

extern "C" int yk_bn_train_bwd_f32(const float *z, const float *dy, long long M, int C, const float *gamma, const float *beta,
                                   const float *save_mean, const float *save_invstd, int act, float alpha, float *dz, float *dgamma,
                                   float *dbeta, void *stream) {
    int dev = yk_current_device();
    if (dev < 0) return YK_ERR_NO_DEVICE;
    int rpc;
    const int chunks = chunking((size_t)M, &rpc);
    float *partial = (float *)yk_scratch(dev, stream, 13, sizeof(float) * ((size_t)chunks * 2 * C + C));
    if (!partial) return YK_ERR_NOMEM;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_colreduce_kernel, dim3(chunks, (C + 63) / 64), dim3(256), 0, st, 2, z, dy, (size_t)M, C, rpc, save_mean, save_invstd,
                       gamma, beta, act, alpha, partial);
    hipLaunchKernelGGL(bn_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, st, partial, chunks, C, 1.f, 0.f, 0, dbeta, dgamma,
                       (float *)nullptr);
    const size_t total = (size_t)M * C;
    hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, z, dy, total, C, 1.f / (float)M, save_mean,
                       save_invstd, gamma, beta, (const float *)dbeta, (const float *)dgamma, act, alpha, dz);
    YK_HIP(hipGetLastError());
    return YK_OK;
}

// bias add (+ optional column sum of the gradient for the bias) for the two biased output convs
__global__ void __launch_bounds__(256) bias_add_kernel(float *y, size_t total, int C, const float *bias) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total) y[i] += bias[i % C];
}
extern "C" int yk_bias_add_f32(float *y, long long M, int C, const float *bias, void *stream) {
    const size_t total = (size_t)M * C;
    hipLaunchKernelGGL(bias_add_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, total, C, bias);
    YK_HIP(hipGetLastError());
    return YK_OK;
}
extern "C" int yk_colsum_f32(const float *x, long long M, int C, float *out, void *stream) {
    int dev = yk_current_device();
    if (dev < 0) return YK_ERR_NO_DEVICE;
    int rpc;
    const int chunks = chunking((size_t)M, &rpc);
    float *partial = (float *)yk_scratch(dev, stream, 13, sizeof(float) * ((size_t)chunks * 2 * C + C));
    if (!partial) return YK_ERR_NOMEM;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_colreduce_kernel, dim3(chunks, (C + 63) / 64), dim3(256), 0, st, 0, x, (const float *)nullptr, (size_t)M, C, rpc,
                       (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, 0, 0.f, partial);
    hipLaunchKernelGGL(bn_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, st, partial, chunks, C, 1.f, 0.f, 0, out, (float *)nullptr,
                       (float *)nullptr);
    YK_HIP(hipGetLastError());
    return YK_OK;
}

// --------------------------------------------------------------------------------------------------------
// small element-wise pieces
// --------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) upsample_bwd_kernel(const float *__restrict__ dy, int B, int H, int W, int C, float *__restrict__ dx) {
    const size_t total = (size_t)B * H * W * C;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const size_t p = i / C;
    const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((size_t)W * H));
    const size_t r0 = (((size_t)b * 2 * H + 2 * y) * 2 * W + 2 * x) * C + c, rs = (size_t)2 * W * C;
    dx[i] = dy[r0] + dy[r0 + C] + dy[r0 + rs] + dy[r0 + rs + C];
}
extern "C" int yk_upsample2x_bwd_f32(const float *dy, int B, int H, int W, int C, float *dx, void *stream) {
    const size_t total = (size_t)B * H * W * C;
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, B, H, W, C, dx);
    YK_HIP(hipGetLastError());
    return YK_OK;
}
__global__ void __launch_bounds__(256) axpy_kernel(size_t n, float a, const float *x, float *y) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] += a * x[i];
}
extern "C" int yk_axpy_f32(long long n, float a, const float *x, float *y, void *stream) {
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)(((size_t)n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (size_t)n, a, x, y);
    YK_HIP(hipGetLastError());
    return YK_OK;
}

// Keras Adam (keras_train.py:74-76): lr_t = lr / (1 + decay*iterations) * sqrt(1 - b2^t) / (1 - b1^t), t = iterations + 1;
// m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g^2 ; p -= lr_t * m / (sqrt(v) + eps)
__global__ void __launch_bounds__(256) adam_kernel(size_t n, float *p, const float *g, float *m, float *v, float lr_t, float b1, float b2,
                                                   float eps, float gscale) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi, vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
}
extern "C" int yk_adam_f32(long long n, float *p, const float *g, float *m, float *v, float lr, float decay, long long iterations,
                           float beta1, float beta2, float eps, float grad_scale, void *stream) {
    const double t = (double)iterations + 1.0;
    const double lr_t = (double)lr / (1.0 + (double)decay * (double)iterations) * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t));
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(((size_t)n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (size_t)n, p, g, m, v,
                       (float)lr_t, beta1, beta2, eps, grad_scale);
    YK_HIP(hipGetLastError());
    return YK_OK;
}

// --------------------------------------------------------------------------------------------------------
// MaxPool2D 2x2, padding='same' (bottom/right padded with -inf), stride 1 or 2 (tiny_yolo, yolonet.py:112-124).
// forward records the winning tap (first maximum in row-major window order); backward gathers.
// --------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const float *__restrict__ x, int B, int Hi, int Wi, int C, int Ho, int Wo, int stride,
                                                          float *__restrict__ y, uint8_t *__restrict__ arg) {
    const size_t total = (size_t)B * Ho * Wo * C;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const size_t m = i / C;
    const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), b = (int)(m / ((size_t)Wo * Ho));
    float best = -__builtin_huge_valf();
    int bt = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int iy = oy * stride + (t >> 1), ix = ox * stride + (t & 1);
        if (iy < Hi && ix < Wi) {
            const float v = x[(((size_t)b * Hi + iy) * Wi + ix) * C + c];
            if (v > best) { best = v; bt = t; }
        }
    }
    y[i] = best;
    arg[i] = (uint8_t)bt;
}
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const float *__restrict__ dy, const uint8_t *__restrict__ arg, int B, int Hi, int Wi,
                                                          int C, int Ho, int Wo, int stride, float *__restrict__ dx) {
    const size_t total = (size_t)B * Hi * Wi * C;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const size_t p = i / C;
    const int ix = (int)(p % Wi), iy = (int)((p / Wi) % Hi), b = (int)(p / ((size_t)Wi * Hi));
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ny = iy - (t >> 1), nx = ix - (t & 1);
        if (ny < 0 || nx < 0 || ny % stride || nx % stride) continue;
        const int oy = ny / stride, ox = nx / stride;
        if (oy >= Ho || ox >= Wo) continue;
        const size_t o = (((size_t)b * Ho + oy) * Wo + ox) * C + c;
        if (arg[o] == t) s += dy[o];
    }
    dx[i] = s;
}
extern "C" int yk_maxpool2_fwd_f32(const float *x, int B, int Hi, int Wi, int C, int Ho, int Wo, int stride, float *y, uint8_t *argmax,
                                   void *stream) {
    const size_t total = (size_t)B * Ho * Wo * C;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, B, Hi, Wi, C, Ho, Wo,
                       stride, y, argmax);
    YK_HIP(hipGetLastError());
    return YK_OK;
}
extern "C" int yk_maxpool2_bwd_f32(const float *dy, const uint8_t *argmax, int B, int Hi, int Wi, int C, int Ho, int Wo, int stride, float *dx,
                                   void *stream) {
    const size_t total = (size_t)B * Hi * Wi * C;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, argmax, B, Hi, Wi, C,
                       Ho, Wo, stride, dx);
    YK_HIP(hipGetLastError());
    return YK_OK;
}
