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
// Round 6 (what train.py calls now; the entry points above stay):
//   yk_gemm_bn_fwd_f32 / yk_dw3x3_bn_fwd_f32     conv + BatchNorm forward in one call, the producer of z leaves the statistics' partial sums
//   yk_gemm_f32_grouped / yk_dw3x3_bwd_weight_grouped_f32   all weight gradients of a backward pass in four launches
//   yk_conv3x3_bn_fwd_f32 / _bwd_weight_f32 / _bwd_data_f32  3x3 convs as implicit GEMMs (no column matrix)
#include "yk_common.h"
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>

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
    double *stats;                       // forward + BatchNormalization: column partials of C per m tile (yk_gemm_f32.h), or null
};

__global__ void __launch_bounds__(256) gemm_f32_kernel(const gemm_args g) {
    constexpr int BM = 64, BN = 64, BK = 16, LDT = 80;       // row stride = 16 mod 32 banks: a 4-row fragment read is conflict-free
    __shared__ float As[BK][LDT], Bs[BK][LDT];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int nk = (g.K + BK - 1) / BK;
    const int per = (nk + g.splitk - 1) / g.splitk;
    const int kb = blockIdx.z * per, ke = min(nk, kb + per);
    floatx4t acc[2][2];                                      // [n tile][m tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = floatx4t{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    // each thread brings 4 elements of A and 4 of B per k-tile; the index split follows the contiguous axis
    int a_mm[4], a_kk[4], b_nn[4], b_kk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + i * 256;
        if (g.transA) { a_kk[i] = e >> 6; a_mm[i] = e & 63; } else { a_mm[i] = e >> 4; a_kk[i] = e & 15; }
        if (g.transB) { b_nn[i] = e >> 4; b_kk[i] = e & 15; } else { b_kk[i] = e >> 6; b_nn[i] = e & 63; }
    }
    float ra[4], rb[4];
    auto fetch = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + a_mm[i], k = k0 + a_kk[i];
            ra[i] = (m < g.M && k < g.K) ? (g.transA ? g.A[(size_t)k * g.lda + m] : g.A[(size_t)m * g.lda + k]) : 0.f;
            const int n = n0 + b_nn[i], k2 = k0 + b_kk[i];
            rb[i] = (n < g.N && k2 < g.K) ? (g.transB ? g.B[(size_t)n * g.ldb + k2] : g.B[(size_t)k2 * g.ldb + n]) : 0.f;
        }
    };
    if (kb < ke) fetch(kb);
    for (int kt = kb; kt < ke; ++kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            As[a_kk[i]][a_mm[i]] = ra[i];
            Bs[b_kk[i]][b_nn[i]] = rb[i];
        }
        __syncthreads();
        if (kt + 1 < ke) fetch(kt + 1);                      // next tile's global loads fly under this tile's MFMAs
#pragma unroll
        for (int k4 = 0; k4 < BK; k4 += 4) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[k4 + fq][wm * 32 + i * 16 + fr];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[k4 + fq][wn * 32 + j * 16 + fr];
            // operand roles swapped (matrix B feeds MFMA operand A): each lane ends up with 4 CONSECUTIVE n of one m
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[i], acc[j][i], 0, 0, 0);
        }
        __syncthreads();
    }
    // D layout: row = (lane>>4)*4 + r -> n, col = lane&15 -> m
    const bool slab = g.splitk > 1;
    float *base = slab ? g.ws + (size_t)blockIdx.z * g.M * g.N : g.C;
    const int ld = slab ? g.N : g.ldc;
    const bool vec = ((ld & 3) == 0) && ((g.N & 3) == 0) && ((((uintptr_t)base) & 15) == 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + wm * 32 + i * 16 + fr, n = n0 + wn * 32 + j * 16 + fq * 4;
            if (m >= g.M || n >= g.N) continue;
            float *c = base + (size_t)m * ld + n;
            floatx4t v = acc[j][i];
            if (vec) {
                if (!slab) {
                    v *= g.alpha;
                    if (g.beta != 0.f) v += g.beta * *(const floatx4t *)c;
                }
                *(floatx4t *)c = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < g.N) c[r] = slab ? v[r] : g.alpha * v[r] + (g.beta != 0.f ? g.beta * c[r] : 0.f);
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

// Same sum for SMALL outputs with many slabs (the weight gradients of the early layers: a 32x16 result reduced over 64 slabs): the
// kernel above would be two workgroups of threads each walking 64 dependent-latency loads.  Here 16 lanes share one output: lane j
// adds slabs j, j+16, ... in order, then the 16 partial sums are added in lane order through LDS - a fixed order, so still bitwise
// reproducible.  block = 16 (slab lanes) x 16 (outputs).
__global__ void __launch_bounds__(256) splitk_sum16_kernel(const float *__restrict__ ws, int splits, int M, int N, float alpha, float beta,
                                                           float *__restrict__ c, int ldc) {
    __shared__ float red[16][17];
    const int zl = threadIdx.x >> 4, il = threadIdx.x & 15;
    const size_t tot = (size_t)M * N, i = (size_t)blockIdx.x * 16 + il;
    float s = 0.f;
    if (i < tot) {
#pragma unroll 8
        for (int z = zl; z < splits; z += 16) s += ws[(size_t)z * tot + i];
    }
    red[zl][il] = s;
    __syncthreads();
    if (zl == 0 && i < tot) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][il];
        const size_t r = i / N, q = i - r * N;
        float *o = c + r * ldc + q;
        *o = alpha * t + (beta != 0.f ? beta * *o : 0.f);
    }
}

struct conv_geom {
    int B, Hi, Wi, C, Ho, Wo, stride, pad_t, pad_l;
};
#include "yk_gemm_f32.h"

static int gemm_splits(int M, int N, int K) {
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    const int nk = (K + 15) / 16;
    int s = 1;
    if (tiles < 512 && nk >= 8) {             // small grids (weight gradients, late 7x10 / 14x20 layers): fill the 256 CUs with K slices
        s = (int)std::min<long>((1024 + tiles - 1) / tiles, nk / 4);
        if (s < 1) s = 1;
        // up to 512 slices when the result is one or two tiles (the weight gradients of the 112x160 layers: a 96x16 result over
        // K = 286 720 rows ran on 64 workgroups at 0.9 TB/s); 16 lanes per output fold the slabs in the finishing pass
        if (s > 512) s = 512;
    }
    return s;
}

extern "C" int yk_gemm_f32(int transA, int transB, int M, int N, int K, float alpha, const float *A, int lda, const float *B,
                           int ldb, float beta, float *C, int ldc, void *stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) {
        yk_set_error("yk_gemm_f32: bad argument");
        return YK_ERR_ARG;
    }
    gemm_args g;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.transA = transA; g.transB = transB;
    g.alpha = alpha; g.beta = beta; g.A = A; g.B = B; g.C = C; g.stats = nullptr;
    const int s = gemm_splits(M, N, K);
    g.splitk = s;
    g.ws = nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (s > 1) {
        int dev = yk_current_device();
        if (dev < 0) return YK_ERR_NO_DEVICE;
        g.ws = (float *)yk_scratch(dev, stream, 14, sizeof(float) * (size_t)s * M * N);
        if (!g.ws) return YK_ERR_NOMEM;
    }
    const dim3 grid((M + 63) / 64, (N + 63) / 64, s);
    if (!transA && transB) launch_gemm_v2<false, true>(g, grid, st);          // forward
    else if (!transA && !transB) launch_gemm_v2<false, false>(g, grid, st);   // data gradient
    else if (transA && !transB) launch_gemm_v2<true, false>(g, grid, st);     // weight gradient
    else hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, st, g);
    if (s > 1) {
        const size_t tot = (size_t)M * N;
        if (s >= 16 && tot <= 65536)
            hipLaunchKernelGGL(splitk_sum16_kernel, dim3((unsigned)((tot + 15) / 16)), dim3(256), 0, st, (const float *)g.ws, s, M, N, alpha,
                               beta, C, ldc);
        else
            hipLaunchKernelGGL(splitk_sum_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float *)g.ws, s, M, N, alpha, beta,
                               C, ldc);
    }
    YK_HIP(hipGetLastError());
    return YK_OK;
}

// ---- grouped GEMMs (yk_gemm_f32.h): the K-slice sums of all problems of a group in one launch too
struct sum_item {
    const float *ws;
    float *c;
    int splits, M, N, ldc, first, wide;      // wide: 16 lanes per output (splits >= 16, small result), as splitk_sum16_kernel
};
struct sum_group {
    int count;
    float alpha, beta;
    sum_item p[YK_GROUP_MAX];
};
__global__ void __launch_bounds__(256) splitk_sum_grouped_kernel(const sum_group S) {
    __shared__ float red[16][17];
    int i = 0;
    while (i + 1 < S.count && (int)blockIdx.x >= S.p[i + 1].first) ++i;
    const sum_item &q = S.p[i];
    const int local = (int)blockIdx.x - q.first;
    const size_t tot = (size_t)q.M * q.N;
    if (!q.wide) {                                           // = splitk_sum_kernel
        const size_t e = (size_t)local * 256 + threadIdx.x;
        if (e >= tot) return;
        float s = 0.f;
        for (int z = 0; z < q.splits; ++z) s += q.ws[(size_t)z * tot + e];
        const size_t r = e / q.N, c = e - r * q.N;
        float *o = q.c + r * q.ldc + c;
        *o = S.alpha * s + (S.beta != 0.f ? S.beta * *o : 0.f);
        return;
    }
    const int zl = threadIdx.x >> 4, il = threadIdx.x & 15;  // = splitk_sum16_kernel
    const size_t e = (size_t)local * 16 + il;
    float s = 0.f;
    if (e < tot) {
#pragma unroll 8
        for (int z = zl; z < q.splits; z += 16) s += q.ws[(size_t)z * tot + e];
    }
    red[zl][il] = s;
    __syncthreads();
    if (zl == 0 && e < tot) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][il];
        const size_t r = e / q.N, c = e - r * q.N;
        float *o = q.c + r * q.ldc + c;
        *o = S.alpha * t + (S.beta != 0.f ? S.beta * *o : 0.f);
    }
}

// `count` independent GEMMs of ONE layout (transA, transB) and one (alpha, beta), YK_GROUP_MAX problems per launch: the tile code of
// yk_gemm_f32 with K slices sized for the group (fixed order of additions: reproducible run to run, not bit-equal to the separate call
// when the slice counts differ).  Problems whose shapes or addresses rule out 16-byte loads go through yk_gemm_f32 one by one.
extern "C" int yk_gemm_f32_grouped(int count, int transA, int transB, const int *M, const int *N, const int *K, float alpha, const float *const *A,
                                   const int *lda, const float *const *B, const int *ldb, float beta, float *const *C, const int *ldc, void *stream) {
    if (count < 0 || (count && (!M || !N || !K || !A || !lda || !B || !ldb || !C || !ldc))) {
        yk_set_error("yk_gemm_f32_grouped: bad argument");
        return YK_ERR_ARG;
    }
    int dev = yk_current_device();
    if (dev < 0) return YK_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    std::vector<int> grouped;
    std::vector<int> splits(count, 1);
    size_t ws_floats = 0;
    for (int i = 0; i < count; ++i) {
        if (!A[i] || !B[i] || !C[i] || M[i] <= 0 || N[i] <= 0 || K[i] <= 0) {
            yk_set_error("yk_gemm_f32_grouped: bad problem %d", i);
            return YK_ERR_ARG;
        }
        const bool va = transA ? (M[i] % 4 == 0) : (K[i] % 4 == 0), vb = transB ? (K[i] % 4 == 0) : (N[i] % 4 == 0);
        const bool vec = va && vb && lda[i] % 4 == 0 && ldb[i] % 4 == 0 && (((uintptr_t)A[i] | (uintptr_t)B[i]) & 15) == 0;
        if (!vec || (transA && transB)) {
            const int rc = yk_gemm_f32(transA, transB, M[i], N[i], K[i], alpha, A[i], lda[i], B[i], ldb[i], beta, C[i], ldc[i], stream);
            if (rc != YK_OK) return rc;
            continue;
        }
        grouped.push_back(i);
    }
    if (grouped.empty()) return YK_OK;
    // K slices: alone, a problem is split until ITS tiles fill the chip (~1000 workgroups: 16 MB of slabs each, 0.56 GB written and read
    // again for the 35 weight gradients of configs[3]: r6c54).  Together the tiles of all problems fill it: slices of ~T k-steps such that the
    // group has ~8192 workgroups, never more slices than the problem would take alone.
    {
        double units = 0;
        for (int i : grouped) units += (double)((M[i] + 63) / 64) * ((N[i] + 63) / 64) * ((K[i] + 31) / 32);
        const double T = std::max(8.0, units / 8192.0);
        for (int i : grouped) {
            const int nk = (K[i] + 31) / 32;
            splits[i] = std::max(1, std::min(gemm_splits(M[i], N[i], K[i]), (int)((nk + T - 1) / T)));
            if (splits[i] > 1) ws_floats += ((size_t)splits[i] * M[i] * N[i] + 3) / 4 * 4;
        }
    }
    float *ws = nullptr;
    if (ws_floats) {
        ws = (float *)yk_scratch(dev, stream, 22, sizeof(float) * ws_floats);
        if (!ws) return YK_ERR_NOMEM;
    }
    size_t ws_off = 0;
    for (size_t g0 = 0; g0 < grouped.size(); g0 += YK_GROUP_MAX) {
        const int n = (int)std::min<size_t>(YK_GROUP_MAX, grouped.size() - g0);
        gemm_group G;
        sum_group S;
        G.count = n;
        S.count = 0;
        S.alpha = alpha;
        S.beta = beta;
        int wg = 0, swg = 0;
        for (int j = 0; j < n; ++j) {
            const int i = grouped[g0 + j];
            gemm_args &g = G.p[j];
            g.M = M[i]; g.N = N[i]; g.K = K[i]; g.lda = lda[i]; g.ldb = ldb[i]; g.ldc = ldc[i]; g.transA = transA; g.transB = transB;
            g.alpha = alpha; g.beta = beta; g.A = A[i]; g.B = B[i]; g.C = C[i]; g.stats = nullptr; g.ws = nullptr;
            g.splitk = splits[i];
            if (splits[i] > 1) {
                g.ws = ws + ws_off;
                ws_off += ((size_t)splits[i] * M[i] * N[i] + 3) / 4 * 4;
                sum_item &q = S.p[S.count++];
                const size_t tot = (size_t)M[i] * N[i];
                q.ws = g.ws; q.c = C[i]; q.splits = splits[i]; q.M = M[i]; q.N = N[i]; q.ldc = ldc[i]; q.first = swg;
                q.wide = splits[i] >= 16 && tot <= 65536;
                swg += (int)(q.wide ? (tot + 15) / 16 : (tot + 255) / 256);
            }
            G.first[j] = wg;
            wg += ((M[i] + 63) / 64) * ((N[i] + 63) / 64) * splits[i];
        }
        G.first[n] = wg;
        if (!transA && transB) hipLaunchKernelGGL((gemm_f32_grouped_kernel<false, true, true>), dim3(wg), dim3(256), 0, st, G);
        else if (!transA && !transB) hipLaunchKernelGGL((gemm_f32_grouped_kernel<false, false, true>), dim3(wg), dim3(256), 0, st, G);
        else hipLaunchKernelGGL((gemm_f32_grouped_kernel<true, false, true>), dim3(wg), dim3(256), 0, st, G);
        if (S.count) hipLaunchKernelGGL(splitk_sum_grouped_kernel, dim3(swg), dim3(256), 0, st, S);
    }
    YK_HIP(hipGetLastError());
    return YK_OK;
}

// <x, y> in two fixed-order stages (the l2 regulariser's value): out = alpha * dot + beta * out
__global__ void __launch_bounds__(256) dot_partial_kernel(size_t n, const float *__restrict__ x, const float *__restrict__ y, double *__restrict__ part) {
    __shared__ double red[256];
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += (double)x[i] * (double)y[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void __launch_bounds__(64) dot_finish_kernel(const double *__restrict__ part, int blocks, float alpha, float beta, float *__restrict__ out) {
    double s = 0;
    for (int k = threadIdx.x; k < blocks; k += 64) s += part[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) *out = alpha * (float)s + (beta != 0.f ? beta * *out : 0.f);
}
extern "C" int yk_dot_f32(long long n, const float *x, const float *y, float alpha, float beta, float *out, void *stream) {
    int dev = yk_current_device();
    if (dev < 0) return YK_ERR_NO_DEVICE;
    const int blocks = (int)std::min<long long>(512, (n + 255) / 256);
    double *part = (double *)yk_scratch(dev, stream, 15, sizeof(double) * 512);
    if (!part) return YK_ERR_NOMEM;
    hipLaunchKernelGGL(dot_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (size_t)n, x, y, part);
    hipLaunchKernelGGL(dot_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double *)part, blocks, alpha, beta, out);
    YK_HIP(hipGetLastError());
    return YK_OK;
}

// keras.regularizers.l2(weight) over SEGMENTS of one flat parameter buffer (yolonet.py:245-250: every DarknetConv2D kernel): the value
// weight * sum w^2 and / or the gradient 2 * weight * w added to the flat gradient buffer, for all segments in ONE pass (round 4: a dot
// product - two launches - and an axpy per layer: 52 launches of ~5 us on the step's critical path).  Fixed-order two-stage sum in double.
__global__ void __launch_bounds__(256) l2_seg_kernel(const float *__restrict__ P, float *__restrict__ G, const long long *__restrict__ pre,
                                                     const long long *__restrict__ off, int nseg, float two_w, int want_value, int want_grad,
                                                     double *__restrict__ part) {
    __shared__ double red[256];
    const long long total = pre[nseg];
    double s = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int lo = 0, hi = nseg;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= i) lo = mid;
            else hi = mid;
        }
        const long long j = off[lo] + (i - pre[lo]);
        const float w = P[j];
        s += (double)w * (double)w;
        if (want_grad) G[j] += two_w * w;
    }
    if (!want_value) return;
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
extern "C" int yk_l2_segments_f32(const float *params, float *grads, const long long *d_prefix, const long long *d_offset, int nseg, long long total,
                                  float weight, int want_value, int want_grad, float *out, void *stream) {
    if (!params || !d_prefix || !d_offset || nseg <= 0 || total <= 0 || (want_grad && !grads) || (want_value && !out)) {
        yk_set_error("yk_l2_segments_f32: bad argument");
        return YK_ERR_ARG;
    }
    int dev = yk_current_device();
    if (dev < 0) return YK_ERR_NO_DEVICE;
    const int blocks = (int)std::min<long long>(512, (total + 255) / 256);
    double *part = (double *)yk_scratch(dev, stream, 15, sizeof(double) * 512);
    if (!part) return YK_ERR_NOMEM;
    hipLaunchKernelGGL(l2_seg_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, d_prefix, d_offset, nseg, 2.f * weight, want_value,
                       want_grad, part);
    if (want_value) hipLaunchKernelGGL(dot_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double *)part, blocks, weight, 0.f, out);
    YK_HIP(hipGetLastError());
    return YK_OK;
}

// --------------------------------------------------------------------------------------------------------
// im2col / col2im for 3x3 convs, NHWC.  col: [B*Ho*Wo][9*C] with k = (ky*3+kx)*C + c (matches OHWI weights).
// --------------------------------------------------------------------------------------------------------
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

// four channels per thread, 32-bit index arithmetic: the element-wise kernel above spends its time in 64-bit divisions (the 14x20x704
// head conv's 113 MB column matrix took 100 us, 1.1 TB/s)
__global__ void __launch_bounds__(256) im2col3x3_v4_kernel(conv_geom q, const float *__restrict__ x, float *__restrict__ col, uint32_t total4) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total4) return;
    const uint32_t C4 = (uint32_t)q.C >> 2;
    const uint32_t r = i / C4, c4 = i - r * C4;
    const uint32_t m = r / 9u, t = r - m * 9u;
    const uint32_t my = m / (uint32_t)q.Wo, ox = m - my * (uint32_t)q.Wo;
    const uint32_t b = my / (uint32_t)q.Ho, oy = my - b * (uint32_t)q.Ho;
    const uint32_t ky = t / 3u, kx = t - ky * 3u;
    const int iy = (int)oy * q.stride - q.pad_t + (int)ky, ix = (int)ox * q.stride - q.pad_l + (int)kx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)iy < (unsigned)q.Hi && (unsigned)ix < (unsigned)q.Wi)
        v = *reinterpret_cast<const float4 *>(x + (((size_t)b * q.Hi + iy) * q.Wi + ix) * q.C + c4 * 4);
    *reinterpret_cast<float4 *>(col + (size_t)i * 4) = v;
}
extern "C" int yk_im2col3x3_f32(const float *x, int B, int Hi, int Wi, int C, int Ho, int Wo, int stride, int pad_t, int pad_l,
                                float *col, void *stream) {
    conv_geom q = {B, Hi, Wi, C, Ho, Wo, stride, pad_t, pad_l};
    const size_t total = (size_t)B * Ho * Wo * 9 * C;
    if (C % 4 == 0 && total / 4 < 0xffffff00ull && (((uintptr_t)x | (uintptr_t)col) & 15) == 0) {
        const uint32_t total4 = (uint32_t)(total / 4);
        hipLaunchKernelGGL(im2col3x3_v4_kernel, dim3((total4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, q, x, col, total4);
        YK_HIP(hipGetLastError());
        return YK_OK;
    }
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
// V consecutive channels per thread (V = 4 when C % 4 == 0: one 16-byte access instead of four 4-byte ones)
template <int V>
__device__ __forceinline__ void ldv(const float *p, float (&v)[V]) {
    if constexpr (V == 4) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int k = 0; k < V; ++k) v[k] = p[k];
    }
}
template <int V>
__device__ __forceinline__ void stv(float *p, const float (&v)[V]) {
    if constexpr (V == 4) *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else {
#pragma unroll
        for (int k = 0; k < V; ++k) p[k] = v[k];
    }
}

template <int V>
__global__ void __launch_bounds__(256) dw_fwd_kernel(conv_geom q, const float *__restrict__ x, const float *__restrict__ w,
                                                     float *__restrict__ y) {
    const int CV = q.C / V;
    const size_t total = (size_t)q.B * q.Ho * q.Wo * CV;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % CV) * V;
    const size_t m = i / CV;
    const int ox = (int)(m % q.Wo), oy = (int)((m / q.Wo) % q.Ho), b = (int)(m / ((size_t)q.Wo * q.Ho));
    float s[V];
#pragma unroll
    for (int k = 0; k < V; ++k) s[k] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int iy = oy * q.stride - q.pad_t + t / 3, ix = ox * q.stride - q.pad_l + t % 3;
        if ((unsigned)iy < (unsigned)q.Hi && (unsigned)ix < (unsigned)q.Wi) {
            float xv[V], wv[V];
            ldv<V>(x + (((size_t)b * q.Hi + iy) * q.Wi + ix) * q.C + c, xv);
            ldv<V>(w + t * q.C + c, wv);
#pragma unroll
            for (int k = 0; k < V; ++k) s[k] += xv[k] * wv[k];
        }
    }
    stv<V>(y + m * q.C + c, s);
}

template <int V>
__global__ void __launch_bounds__(256) dw_bwd_data_kernel(conv_geom q, const float *__restrict__ dy, const float *__restrict__ w,
                                                          float *__restrict__ dx) {
    const int CV = q.C / V;
    const size_t total = (size_t)q.B * q.Hi * q.Wi * CV;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % CV) * V;
    const size_t p = i / CV;
    const int ix = (int)(p % q.Wi), iy = (int)((p / q.Wi) % q.Hi), b = (int)(p / ((size_t)q.Wi * q.Hi));
    float s[V];
#pragma unroll
    for (int k = 0; k < V; ++k) s[k] = 0.f;
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
            float gv[V], wv[V];
            ldv<V>(dy + (((size_t)b * q.Ho + oy) * q.Wo + ox) * q.C + c, gv);
            ldv<V>(w + (ky * 3 + kx) * q.C + c, wv);
#pragma unroll
            for (int k = 0; k < V; ++k) s[k] += gv[k] * wv[k];
        }
    }
    stv<V>(dx + p * q.C + c, s);
}

// Column-sum finish shared by every two-stage reduction here: one wavefront per output element, lanes stride over the
// chunks and fold with a fixed xor tree — deterministic, and 64x shorter than a serial walk over 512 chunks.
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ void __launch_bounds__(256) colsum_finish_kernel(const float *__restrict__ partial, int chunks, int n, float *__restrict__ out,
                                                            float scale) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    float s = 0.f;
    for (int k = lane; k < chunks; k += 64) s += partial[(size_t)k * n + i];
    s = wave_sum(s);
    if (lane == 0) out[i] = s * scale;
}

// dw[t][c] = sum_{b,oy,ox} dy * x(tap t).  One thread = one channel x one output row: it slides a 3x3 register window along
// ox, so each step loads 3*stride new inputs instead of 9.  block = CW channel lanes x (256/CW) row lanes; grid (row chunks,
// channel groups); partial[chunk][9][C] is folded by colsum_finish_kernel.
// A thread sweeps ONE segment of an image row of one channel (shift register over x).  `segs` segments per row: whole rows gave the
// 112x160x32 layer 224 workgroups of serial 160-pixel sweeps (73 MB at 1 TB/s); four segments per row are 896 workgroups.
__device__ __forceinline__ void dw_bwd_weight_body(const conv_geom q, const float *__restrict__ x, const float *__restrict__ dy,
                                                   float *__restrict__ partial, const int rows_per_chunk, const int cw_log2, const int segs, const int wseg,
                                                   const int bx, const int by, float (*red)[256]) {
    const int CW = 1 << cw_log2, RL = 256 >> cw_log2;
    const int cl = threadIdx.x & (CW - 1), rl = threadIdx.x >> cw_log2;
    const int c = by * CW + cl;
    const int rows = q.B * q.Ho * segs;                            // virtual rows: (image row, segment)
    const int r0 = bx * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (c < q.C)
        for (int vr = r0 + rl; vr < r1; vr += RL) {
            const int r = vr / segs, sg = vr - r * segs;
            const int oxa = sg * wseg, oxb = min(q.Wo, oxa + wseg);
            const int b = r / q.Ho, oy = r - b * q.Ho;
            const float *xr[3];
            bool rv[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy * q.stride - q.pad_t + ky;
                rv[ky] = (unsigned)iy < (unsigned)q.Hi;
                xr[ky] = x + (((size_t)b * q.Hi + (rv[ky] ? iy : 0)) * q.Wi) * q.C + c;
            }
            const float *gr = dy + ((size_t)r * q.Wo) * q.C + c;
            float w[3][3];
            auto ld = [&](int ky, int ix) -> float { return (rv[ky] && (unsigned)ix < (unsigned)q.Wi) ? xr[ky][(size_t)ix * q.C] : 0.f; };
            int ix0 = oxa * q.stride - q.pad_l;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                w[ky][0] = 0.f;                       // filled by the first shift below
                w[ky][1] = ld(ky, ix0);
                w[ky][2] = ld(ky, ix0 + 1);
            }
            if (q.stride == 1) {
                for (int ox = oxa; ox < oxb; ++ox) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        w[ky][0] = w[ky][1];
                        w[ky][1] = w[ky][2];
                        w[ky][2] = ld(ky, ox - q.pad_l + 2);
                    }
                    const float g = gr[(size_t)ox * q.C];
#pragma unroll
                    for (int t = 0; t < 9; ++t) s[t] += g * w[t / 3][t % 3];
                }
            } else {
                for (int ox = oxa; ox < oxb; ++ox) {
                    const int ix = ox * q.stride - q.pad_l;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        w[ky][0] = (ox > oxa && q.stride == 2) ? w[ky][2] : ld(ky, ix);
                        w[ky][1] = ld(ky, ix + 1);
                        w[ky][2] = ld(ky, ix + 2);
                    }
                    const float g = gr[(size_t)ox * q.C];
#pragma unroll
                    for (int t = 0; t < 9; ++t) s[t] += g * w[t / 3][t % 3];
                }
            }
        }
#pragma unroll
    for (int t = 0; t < 9; ++t) red[t][threadIdx.x] = s[t];
    __syncthreads();
    if (rl == 0 && c < q.C)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float a = 0.f;
            for (int k = 0; k < RL; ++k) a += red[t][k * CW + cl];
            partial[((size_t)bx * 9 + t) * q.C + c] = a;
        }
}
__global__ void __launch_bounds__(256) dw_bwd_weight_kernel(conv_geom q, const float *__restrict__ x, const float *__restrict__ dy,
                                                            float *__restrict__ partial, int rows_per_chunk, int cw_log2, int segs, int wseg) {
    __shared__ float red[9][256];
    dw_bwd_weight_body(q, x, dy, partial, rows_per_chunk, cw_log2, segs, wseg, blockIdx.x, blockIdx.y, red);
}
// GROUPED (round 6): the depthwise weight gradients of a whole backward pass in one launch + one finishing launch (17 + 17 launches of
// configs[3] become 2); the problems ride in the kernel arguments, each computed exactly as by the separate launches (bitwise the same).
struct dww_item {
    conv_geom q;
    const float *x, *dy;
    float *partial, *dw;
    int rpc, cwl, segs, wseg, chunks, groups, first, ffirst;     // first / ffirst: first workgroup of the problem in the main / finishing launch
};
struct dww_group {
    int count;
    dww_item p[YK_GROUP_MAX];
};
static_assert(sizeof(dww_group) <= 4096, "the group rides in the kernel arguments");
__global__ void __launch_bounds__(256) dw_bwd_weight_grouped_kernel(const dww_group G) {
    __shared__ float red[9][256];
    int i = 0;
    while (i + 1 < G.count && (int)blockIdx.x >= G.p[i + 1].first) ++i;
    const dww_item it = G.p[i];
    const int local = (int)blockIdx.x - it.first;
    dw_bwd_weight_body(it.q, it.x, it.dy, it.partial, it.rpc, it.cwl, it.segs, it.wseg, local % it.chunks, local / it.chunks, red);
}
__global__ void __launch_bounds__(256) colsum_finish_grouped_kernel(const dww_group G) {
    int i = 0;
    while (i + 1 < G.count && (int)blockIdx.x >= G.p[i + 1].ffirst) ++i;
    const dww_item it = G.p[i];
    const int n = 9 * it.q.C;
    const int o = ((int)blockIdx.x - it.ffirst) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= n) return;
    float s = 0.f;
    for (int k = lane; k < it.chunks; k += 64) s += it.partial[(size_t)k * n + o];
    s = wave_sum(s);
    if (lane == 0) it.dw[o] = s;
}

static int lane_split(int C) { return C <= 16 ? 4 : (C <= 32 ? 5 : 6); }      // log2 of the channel lanes per block

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
    if (C % 4 == 0) hipLaunchKernelGGL(dw_fwd_kernel<4>, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q, x, w, y);
    else hipLaunchKernelGGL(dw_fwd_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q, x, w, y);
    YK_HIP(hipGetLastError());
    return YK_OK;
}
extern "C" int yk_dw3x3_bwd_data_f32(const float *dy, const float *w, int B, int Hi, int Wi, int C, int Ho, int Wo, int stride,
                                     int pad_t, int pad_l, float *dx, void *stream) {
    conv_geom q = {B, Hi, Wi, C, Ho, Wo, stride, pad_t, pad_l};
    const size_t total = (size_t)B * Hi * Wi * C;
    if (C % 4 == 0) hipLaunchKernelGGL(dw_bwd_data_kernel<4>, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q, dy, w, dx);
    else hipLaunchKernelGGL(dw_bwd_data_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q, dy, w, dx);
    YK_HIP(hipGetLastError());
    return YK_OK;
}
struct dww_plan {
    int cwl, groups, segs, wseg, rpc, chunks;
};
static dww_plan dww_planning(int B, int C, int Ho, int Wo) {
    dww_plan p;
    p.cwl = lane_split(C);
    const int RL = 256 >> p.cwl;
    p.groups = (C + (1 << p.cwl) - 1) >> p.cwl;
    // segments per image row: enough (row, segment) units for ~1024 workgroups of RL units each, at least 8 pixels per segment
    int segs = (1024 * RL + B * Ho * p.groups - 1) / (B * Ho * p.groups);
    segs = std::max(1, std::min(segs, std::min(8, Wo / 8)));
    p.wseg = (Wo + segs - 1) / segs;
    p.segs = (Wo + p.wseg - 1) / p.wseg;
    const int rows = B * Ho * p.segs;
    p.rpc = ((rows + 2047) / 2048 + RL - 1) / RL * RL;        // whole row-lane rounds per chunk, at most ~2048 chunks
    p.chunks = (rows + p.rpc - 1) / p.rpc;
    return p;
}
extern "C" int yk_dw3x3_bwd_weight_f32(const float *x, const float *dy, int B, int Hi, int Wi, int C, int Ho, int Wo, int stride,
                                       int pad_t, int pad_l, float *dw, void *stream) {
    conv_geom q = {B, Hi, Wi, C, Ho, Wo, stride, pad_t, pad_l};
    int dev = yk_current_device();
    if (dev < 0) return YK_ERR_NO_DEVICE;
    const dww_plan p = dww_planning(B, C, Ho, Wo);
    float *partial = (float *)yk_scratch(dev, stream, 12, sizeof(float) * (size_t)p.chunks * 9 * C);
    if (!partial) return YK_ERR_NOMEM;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dw_bwd_weight_kernel, dim3(p.chunks, p.groups), dim3(256), 0, st, q, x, dy, partial, p.rpc, p.cwl, p.segs, p.wseg);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((9 * C + 3) / 4), dim3(256), 0, st, partial, p.chunks, 9 * C, dw, 1.f);
    YK_HIP(hipGetLastError());
    return YK_OK;
}
// geom: 9 ints per problem (B, Hi, Wi, C, Ho, Wo, stride, pad_t, pad_l), as the arguments of yk_dw3x3_bwd_weight_f32
extern "C" int yk_dw3x3_bwd_weight_grouped_f32(int count, const float *const *x, const float *const *dy, const int *geom, float *const *dw, void *stream) {
    if (count < 0 || (count && (!x || !dy || !geom || !dw))) {
        yk_set_error("yk_dw3x3_bwd_weight_grouped_f32: bad argument");
        return YK_ERR_ARG;
    }
    if (!count) return YK_OK;
    int dev = yk_current_device();
    if (dev < 0) return YK_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    std::vector<dww_plan> plans(count);
    size_t floats = 0;
    for (int i = 0; i < count; ++i) {
        const int *g = geom + 9 * i;
        if (!x[i] || !dy[i] || !dw[i] || g[0] <= 0 || g[3] <= 0) {
            yk_set_error("yk_dw3x3_bwd_weight_grouped_f32: bad problem %d", i);
            return YK_ERR_ARG;
        }
        plans[i] = dww_planning(g[0], g[3], g[4], g[5]);
        floats += (size_t)plans[i].chunks * 9 * g[3];
    }
    float *partial = (float *)yk_scratch(dev, stream, 23, sizeof(float) * floats);
    if (!partial) return YK_ERR_NOMEM;
    size_t off = 0;
    for (int g0 = 0; g0 < count; g0 += YK_GROUP_MAX) {
        const int n = std::min(YK_GROUP_MAX, count - g0);
        dww_group G;
        G.count = n;
        int wg = 0, fwg = 0;
        for (int j = 0; j < n; ++j) {
            const int i = g0 + j;
            const int *g = geom + 9 * i;
            dww_item &it = G.p[j];
            it.q = conv_geom{g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8]};
            it.x = x[i]; it.dy = dy[i]; it.dw = dw[i];
            it.partial = partial + off;
            off += (size_t)plans[i].chunks * 9 * g[3];
            it.rpc = plans[i].rpc; it.cwl = plans[i].cwl; it.segs = plans[i].segs; it.wseg = plans[i].wseg;
            it.chunks = plans[i].chunks; it.groups = plans[i].groups;
            it.first = wg;
            it.ffirst = fwg;
            wg += it.chunks * it.groups;
            fwg += (9 * g[3] + 3) / 4;
        }
        hipLaunchKernelGGL(dw_bwd_weight_grouped_kernel, dim3(wg), dim3(256), 0, st, G);
        hipLaunchKernelGGL(colsum_finish_grouped_kernel, dim3(fwg), dim3(256), 0, st, G);
    }
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

// Column reductions over the M rows of z [M][C].  block = CW channel lanes x (256/CW) row lanes, grid (row chunks, channel
// groups).  STATS: sum z and sum z^2 in one pass, accumulated in double (mean and E[z^2]-mean^2 from fp32 data lose nothing at
// double precision).  BWD: sums of g and g*xhat with g = dy * act'(.) in float.
template <bool STATS>
__global__ void __launch_bounds__(256) bn_colreduce_kernel(const float *__restrict__ z, const float *__restrict__ dy, size_t M, int C,
                                                           int rows_per_chunk, int cw_log2, const float *__restrict__ mean,
                                                           const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, int act, float alpha, void *__restrict__ partial_v) {
    using acc_t = typename std::conditional<STATS, double, float>::type;
    __shared__ acc_t red[2][256];
    const int CW = 1 << cw_log2, RL = 256 >> cw_log2;
    const int cl = threadIdx.x & (CW - 1), rl = threadIdx.x >> cw_log2;
    const int c = blockIdx.y * CW + cl;
    const size_t r0 = (size_t)blockIdx.x * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    acc_t s0 = 0, s1 = 0;
    if (c < C) {
        if (STATS) {
            for (size_t m = r0 + rl; m < r1; m += RL) {
                const double v = (double)z[m * C + c];
                s0 += v;
                s1 += v * v;
            }
        } else {
            const float mu = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
            for (size_t m = r0 + rl; m < r1; m += RL) {
                const float xh = (z[m * C + c] - mu) * is;
                const float g = dy[m * C + c] * t_act_grad(ga * xh + be, act, alpha);
                s0 += g;
                s1 += g * xh;
            }
        }
    }
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    __syncthreads();
    if (rl == 0 && c < C) {
        acc_t a = 0, b = 0;
        for (int k = 0; k < RL; ++k) {
            a += red[0][k * CW + cl];
            b += red[1][k * CW + cl];
        }
        acc_t *partial = (acc_t *)partial_v;
        partial[((size_t)blockIdx.x * 2 + 0) * C + c] = a;
        partial[((size_t)blockIdx.x * 2 + 1) * C + c] = b;
    }
}
// one wavefront per channel folds the chunk partials
// WPC = waves per channel: 1 -> four channels per workgroup; 4 -> one channel per workgroup, the four wave sums added in order (the
// statistics epilogue of the GEMM leaves one partial per 64 rows: 4 480 of them for the 112x160 layers at 16 images)
template <int WPC>
__global__ void __launch_bounds__(256) bn_stats_finish_kernel(const double *__restrict__ partial, int chunks, int C, double invM, float eps,
                                                              float *__restrict__ mean, float *__restrict__ invstd,
                                                              float *__restrict__ mm, float *__restrict__ mv, float mom) {
    __shared__ double wsum[2][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = WPC == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
    if (c >= C) return;
    double a = 0, b = 0;
    for (int k = WPC == 1 ? lane : (int)threadIdx.x; k < chunks; k += 64 * WPC) {
        a += partial[((size_t)k * 2 + 0) * C + c];
        b += partial[((size_t)k * 2 + 1) * C + c];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (WPC > 1) {
        if (lane == 0) {
            wsum[0][wave] = a;
            wsum[1][wave] = b;
        }
        __syncthreads();
        if (threadIdx.x != 0) return;
        a = ((wsum[0][0] + wsum[0][1]) + wsum[0][2]) + wsum[0][3];
        b = ((wsum[1][0] + wsum[1][1]) + wsum[1][2]) + wsum[1][3];
    }
    if (lane == 0) {
        const double mu = a * invM;
        double var = b * invM - mu * mu;                 // biased (population) variance, as tf.nn.moments
        if (var < 0) var = 0;
        mean[c] = (float)mu;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (mm && mv) {                                  // keras: moving = moving*momentum + batch*(1-momentum)
            // TF 1.14 takes the fused path for NHWC 4-D inputs (tf.nn.fused_batch_norm): normalisation uses the population variance,
            // the MOVING variance is fed the Bessel-corrected one, var * M / (M - 1)
            const double Mrows = 1.0 / invM;
            const double uvar = Mrows > 1.5 ? var * Mrows / (Mrows - 1.0) : var;
            mm[c] = mm[c] * mom + (float)mu * (1.f - mom);
            mv[c] = mv[c] * mom + (float)uvar * (1.f - mom);
        }
    }
}
__global__ void __launch_bounds__(256) bn_bwd_finish_kernel(const float *__restrict__ partial, int chunks, int C, float *__restrict__ dbeta,
                                                            float *__restrict__ dgamma) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    float a = 0.f, b = 0.f;
    for (int k = lane; k < chunks; k += 64) {
        a += partial[((size_t)k * 2 + 0) * C + c];
        b += partial[((size_t)k * 2 + 1) * C + c];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) {
        dbeta[c] = a;
        if (dgamma) dgamma[c] = b;
    }
}
template <int V>
__global__ void __launch_bounds__(256) bn_apply_fwd_kernel(const float *__restrict__ z, size_t total, int C, const float *__restrict__ mean,
                                                           const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, int act, float alpha, float *__restrict__ y,
                                                           const float *__restrict__ res) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V;
    if (i >= total) return;
    const int c = (int)(i % C);
    float zv[V], mu[V], is[V], ga[V], be[V], o[V];
    ldv<V>(z + i, zv); ldv<V>(mean + c, mu); ldv<V>(invstd + c, is); ldv<V>(gamma + c, ga); ldv<V>(beta + c, be);
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = t_act(ga[k] * (zv[k] - mu[k]) * is[k] + be[k], act, alpha);
    if (res) {                                                   // keras Add()([res, this layer's output]) folded into the apply pass
        float rv[V];
        ldv<V>(res + i, rv);
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = rv[k] + o[k];
    }
    stv<V>(y + i, o);
}
template <int V>
__global__ void __launch_bounds__(256) bn_apply_bwd_kernel(const float *__restrict__ z, const float *__restrict__ dy, size_t total, int C,
                                                           float invM, const float *__restrict__ mean, const float *__restrict__ invstd,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           const float *__restrict__ dbeta, const float *__restrict__ dgamma, int act,
                                                           float alpha, float *__restrict__ dz) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V;
    if (i >= total) return;
    const int c = (int)(i % C);
    float zv[V], gy[V], mu[V], is[V], ga[V], be[V], db[V], dg[V], o[V];
    ldv<V>(z + i, zv); ldv<V>(dy + i, gy); ldv<V>(mean + c, mu); ldv<V>(invstd + c, is); ldv<V>(gamma + c, ga); ldv<V>(beta + c, be);
    ldv<V>(dbeta + c, db); ldv<V>(dgamma + c, dg);
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const float xh = (zv[k] - mu[k]) * is[k];
        const float g = gy[k] * t_act_grad(ga[k] * xh + be[k], act, alpha);
        o[k] = ga[k] * is[k] * (g - db[k] * invM - xh * dg[k] * invM);
    }
    stv<V>(dz + i, o);
}

// --------------------------------------------------------------------------------------------------------
// SMALL layers (M <= 1536 rows: the 7x10 layers at 16 images - 14 of the 55 BatchNorms of configs[3]): every launch of the
// reduce -> finish -> apply chain costs 5 us whatever it does, so ONE workgroup of 1024 threads owns 8 channels and ALL rows and does the
// whole chain without leaving the launch (round 6).  Thread (row lane of 512, half): a float4 of channels per row, R rows per thread, ALL of
// them loaded before anything is waited for and kept in registers (a first form with 256 threads sweeping the rows had 32 KB in flight per
// CU: 16 GB/s, 37 us per launch - r6c58); fixed-order LDS tree.
//   forward : fold the producer's partial sums -> statistics (+ moving statistics) -> apply (+ residual)          2 launches -> 1
//   backward: sum g, sum g*xhat over the rows -> dbeta, dgamma -> dz                                               3 launches -> 1
// --------------------------------------------------------------------------------------------------------
template <int R>
__global__ void __launch_bounds__(1024) bn_fwd_cols_kernel(const float *__restrict__ z, int M, int C, const double *__restrict__ partial, int chunks,
                                                           double invM, float eps, const float *__restrict__ gamma, const float *__restrict__ beta, int act,
                                                           float alpha, float *__restrict__ y, float *__restrict__ mean, float *__restrict__ invstd,
                                                           float *__restrict__ mm, float *__restrict__ mv, float mom, const float *__restrict__ res) {
    __shared__ double fold[2][64][8];
    __shared__ float stat[2][8];
    const int half = threadIdx.x & 1, rl = threadIdx.x >> 1;
    const int c4 = blockIdx.x * 8 + half * 4;
    const bool on = c4 < C;
    float zv[R][4], rv[R][4];
    if (on) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int m = rl + r * 512;
            if (m < M) {
                ldv<4>(z + (size_t)m * C + c4, zv[r]);
                if (res) ldv<4>(res + (size_t)m * C + c4, rv[r]);
            }
        }
    }
    {
        const int ch = threadIdx.x & 7, j = (threadIdx.x >> 3) & 1, kl = threadIdx.x >> 4;       // 64 chunk lanes
        const int c = blockIdx.x * 8 + ch;
        double a = 0;
        if (c < C)
            for (int k = kl; k < chunks; k += 64) a += partial[((size_t)k * 2 + j) * C + c];
        fold[j][kl][ch] = a;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const int ch = threadIdx.x, c = blockIdx.x * 8 + ch;
        if (c < C) {
            double a = 0, b = 0;
            for (int k = 0; k < 64; ++k) {
                a += fold[0][k][ch];
                b += fold[1][k][ch];
            }
            const double mu = a * invM;
            double var = b * invM - mu * mu;                 // biased (population) variance, as tf.nn.moments
            if (var < 0) var = 0;
            const float fm = (float)mu, fi = (float)(1.0 / sqrt(var + (double)eps));
            mean[c] = fm;
            invstd[c] = fi;
            stat[0][ch] = fm;
            stat[1][ch] = fi;
            if (mm && mv) {                                  // as bn_stats_finish_kernel
                const double Mrows = 1.0 / invM;
                const double uvar = Mrows > 1.5 ? var * Mrows / (Mrows - 1.0) : var;
                mm[c] = mm[c] * mom + (float)mu * (1.f - mom);
                mv[c] = mv[c] * mom + (float)uvar * (1.f - mom);
            }
        }
    }
    __syncthreads();
    if (!on) return;
    float mu[4], is[4], ga[4], be[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mu[k] = stat[0][half * 4 + k];
        is[k] = stat[1][half * 4 + k];
    }
    ldv<4>(gamma + c4, ga);
    ldv<4>(beta + c4, be);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int m = rl + r * 512;
        if (m < M) {
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = t_act(ga[k] * (zv[r][k] - mu[k]) * is[k] + be[k], act, alpha);
            if (res) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = rv[r][k] + o[k];
            }
            stv<4>(y + (size_t)m * C + c4, o);
        }
    }
}

template <int R>
__global__ void __launch_bounds__(1024) bn_bwd_cols_kernel(const float *__restrict__ z, const float *__restrict__ dy, int M, int C, float invM,
                                                           const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, int act, float alpha, float *__restrict__ dz,
                                                           float *__restrict__ dgamma, float *__restrict__ dbeta) {
    __shared__ float red[2][512][8];
    const int half = threadIdx.x & 1, rl = threadIdx.x >> 1;
    const int c = blockIdx.x * 8 + half * 4;
    const bool on = c < C;
    float mu[4], is[4], ga[4], be[4], s0[4], s1[4];
    float xh[R][4], g[R][4];                                  // kept for the second half: z and dy are read once
#pragma unroll
    for (int k = 0; k < 4; ++k) mu[k] = is[k] = ga[k] = be[k] = s0[k] = s1[k] = 0.f;
    if (on) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int m = rl + r * 512;
#pragma unroll
            for (int k = 0; k < 4; ++k) xh[r][k] = g[r][k] = 0.f;
            if (m < M) {
                ldv<4>(z + (size_t)m * C + c, xh[r]);
                ldv<4>(dy + (size_t)m * C + c, g[r]);
            }
        }
        ldv<4>(mean + c, mu);
        ldv<4>(invstd + c, is);
        ldv<4>(gamma + c, ga);
        ldv<4>(beta + c, be);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (rl + r * 512 < M) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float x = (xh[r][k] - mu[k]) * is[k];
                    const float gg = g[r][k] * t_act_grad(ga[k] * x + be[k], act, alpha);
                    xh[r][k] = x;
                    g[r][k] = gg;
                    s0[k] += gg;
                    s1[k] += gg * x;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[0][rl][half * 4 + k] = s0[k];
        red[1][rl][half * 4 + k] = s1[k];
    }
    __syncthreads();
    for (int o = 256; o > 0; o >>= 1) {                      // fixed-order tree over the 512 row lanes
        if (rl < o) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                red[0][rl][half * 4 + k] += red[0][rl + o][half * 4 + k];
                red[1][rl][half * 4 + k] += red[1][rl + o][half * 4 + k];
            }
        }
        __syncthreads();
    }
    if (!on) return;
    float db[4], dg[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        db[k] = red[0][0][half * 4 + k];
        dg[k] = red[1][0][half * 4 + k];
    }
    if (rl == 0) {
        stv<4>(dbeta + c, db);
        if (dgamma) stv<4>(dgamma + c, dg);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int m = rl + r * 512;
        if (m < M) {
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = ga[k] * is[k] * (g[r][k] - db[k] * invM - xh[r][k] * dg[k] * invM);     // = bn_apply_bwd_kernel
            stv<4>(dz + (size_t)m * C + c, o);
        }
    }
}
// The backward column reduction with float4 lanes (round 6): block = CW channel-quad lanes x RL row lanes (CW = C / 4, any number <= 256), a
// thread walks ~8 rows of its chunk with two 16-byte loads per row.  The scalar form above gave the 112x160 layers (C = 16-32: one channel
// group) 512 workgroups of 70-iteration threads with 4-byte loads: 40 us for 73 MB (1.8 TB/s).  partial: [chunk][2][C] floats.
__global__ void __launch_bounds__(256) bn_colreduce_bwd_v4_kernel(const float *__restrict__ z, const float *__restrict__ dy, uint32_t M, int C, int rows_per_chunk,
                                                                  int CW, int RL, const float *__restrict__ mean, const float *__restrict__ invstd,
                                                                  const float *__restrict__ gamma, const float *__restrict__ beta, int act, float alpha,
                                                                  float *__restrict__ partial) {
    __shared__ float red[2][256][4];
    const int cl = threadIdx.x % CW, rl = threadIdx.x / CW;
    const int c = (blockIdx.y * CW + cl) * 4;
    const uint32_t r0 = blockIdx.x * (uint32_t)rows_per_chunk, r1 = min(M, r0 + (uint32_t)rows_per_chunk);
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    if (rl < RL && c < C) {
        float mu[4], is[4], ga[4], be[4];
        ldv<4>(mean + c, mu);
        ldv<4>(invstd + c, is);
        ldv<4>(gamma + c, ga);
        ldv<4>(beta + c, be);
#pragma unroll 4
        for (uint32_t m = r0 + rl; m < r1; m += RL) {
            float zv[4], gy[4];
            ldv<4>(z + (size_t)m * C + c, zv);
            ldv<4>(dy + (size_t)m * C + c, gy);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = (zv[k] - mu[k]) * is[k];
                const float g = gy[k] * t_act_grad(ga[k] * xh + be[k], act, alpha);
                s0[k] += g;
                s1[k] += g * xh;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[0][threadIdx.x][k] = s0[k];
        red[1][threadIdx.x][k] = s1[k];
    }
    __syncthreads();
    if (rl == 0 && c < C) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a = 0.f, b = 0.f;
            for (int r = 0; r < RL; ++r) {
                a += red[0][r * CW + cl][k];
                b += red[1][r * CW + cl][k];
            }
            partial[((size_t)blockIdx.x * 2 + 0) * C + c + k] = a;
            partial[((size_t)blockIdx.x * 2 + 1) * C + c + k] = b;
        }
    }
}

static bool bn_cols_ok(long long M, int C, const void *a, const void *b, const void *c) {
    // R = 3 rows per thread only: at 4 480 rows (R = 9) a workgroup pulls 287 KB in 32-byte row pieces through ONE CU's L1 and takes 27 / 18 us
    // against 16 / 11 us for the separate launches that spread the same bytes over the chip (r6c58)
    return M <= 3 * 512 && C % 4 == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0;
}

static int bn_chunking(size_t M, int C, int *rows_per_chunk, int *cwl) {
    *cwl = lane_split(C);
    const int RL = 256 >> *cwl;
    size_t rpc = ((M + 511) / 512 + RL - 1) / RL * RL;
    if (rpc < (size_t)RL) rpc = RL;
    *rows_per_chunk = (int)rpc;
    return (int)((M + rpc - 1) / rpc);
}

static int bn_train_fwd(const float *z, long long M, int C, const float *gamma, const float *beta, float eps, int act, float alpha, float *y,
                        float *save_mean, float *save_invstd, float *moving_mean, float *moving_var, float momentum, const float *res, void *stream);
extern "C" int yk_bn_train_fwd_f32(const float *z, long long M, int C, const float *gamma, const float *beta, float eps, int act,
                                   float alpha, float *y, float *save_mean, float *save_invstd, float *moving_mean,
                                   float *moving_var, float momentum, void *stream) {
    return bn_train_fwd(z, M, C, gamma, beta, eps, act, alpha, y, save_mean, save_invstd, moving_mean, moving_var, momentum, nullptr, stream);
}
// the same with the residual of a following keras Add() folded in: y = res + act(BN(z))  (keras_mobilenet_v2.py:483-484, yolonet.py:203)
extern "C" int yk_bn_train_fwd_res_f32(const float *z, long long M, int C, const float *gamma, const float *beta, float eps, int act,
                                       float alpha, float *y, float *save_mean, float *save_invstd, float *moving_mean,
                                       float *moving_var, float momentum, const float *res, void *stream) {
    return bn_train_fwd(z, M, C, gamma, beta, eps, act, alpha, y, save_mean, save_invstd, moving_mean, moving_var, momentum, res, stream);
}
static int bn_train_fwd(const float *z, long long M, int C, const float *gamma, const float *beta, float eps, int act, float alpha, float *y,
                        float *save_mean, float *save_invstd, float *moving_mean, float *moving_var, float momentum, const float *res, void *stream) {
    int dev = yk_current_device();
    if (dev < 0) return YK_ERR_NO_DEVICE;
    int rpc, cwl;
    const int chunks = bn_chunking((size_t)M, C, &rpc, &cwl);
    char *ws = (char *)yk_scratch(dev, stream, 13, sizeof(double) * (size_t)chunks * 2 * C + sizeof(float) * C);
    if (!ws) return YK_ERR_NOMEM;
    double *partial = (double *)ws;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_colreduce_kernel<true>, dim3(chunks, (C + (1 << cwl) - 1) >> cwl), dim3(256), 0, st, z, (const float *)nullptr,
                       (size_t)M, C, rpc, cwl, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, 0,
                       0.f, (void *)partial);
    hipLaunchKernelGGL(bn_stats_finish_kernel<1>, dim3((C + 3) / 4), dim3(256), 0, st, (const double *)partial, chunks, C, 1.0 / (double)M, eps,
                       save_mean, save_invstd, moving_mean, moving_var, momentum);
    const size_t total = (size_t)M * C;
    if (C % 4 == 0)
        hipLaunchKernelGGL(bn_apply_fwd_kernel<4>, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, z, total, C, (const float *)save_mean,
                           (const float *)save_invstd, gamma, beta, act, alpha, y, res);
    else
        hipLaunchKernelGGL(bn_apply_fwd_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, z, total, C, (const float *)save_mean,
                           (const float *)save_invstd, gamma, beta, act, alpha, y, res);
    YK_HIP(hipGetLastError());
    return YK_OK;
}

extern "C" int yk_bn_train_bwd_f32(const float *z, const float *dy, long long M, int C, const float *gamma, const float *beta,
                                   const float *save_mean, const float *save_invstd, int act, float alpha, float *dz, float *dgamma,
                                   float *dbeta, void *stream) {
    int dev = yk_current_device();
    if (dev < 0) return YK_ERR_NO_DEVICE;
    if (bn_cols_ok(M, C, z, dy, dz) && dgamma && (((uintptr_t)dgamma | (uintptr_t)dbeta | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)save_mean | (uintptr_t)save_invstd) & 15) == 0) {
        hipLaunchKernelGGL(bn_bwd_cols_kernel<3>, dim3((C + 7) / 8), dim3(1024), 0, (hipStream_t)stream, z, dy, (int)M, C, 1.f / (float)M, save_mean,
                           save_invstd, gamma, beta, act, alpha, dz, dgamma, dbeta);
        YK_HIP(hipGetLastError());
        return YK_OK;
    }
    int rpc, cwl, chunks;
    hipStream_t st = (hipStream_t)stream;
    float *partial;
    // (only the large layers: below ~50 000 rows the scalar form's launches are at the 5-9 us floor and the 8-row threads of this one are not: r6c62)
    const bool v4 = C % 4 == 0 && M >= 50000 && M < (1ll << 31) &&
                    (((uintptr_t)z | (uintptr_t)dy | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)save_mean | (uintptr_t)save_invstd) & 15) == 0;
    if (v4) {
        const int CV = C / 4, CW = std::min(CV, 256), RL = 256 / CW, groups = (CV + CW - 1) / CW;
        chunks = (int)std::min<long long>(2048, (M + RL * 8 - 1) / (RL * 8));                   // ~8 rows per thread
        rpc = (int)(((M + chunks - 1) / chunks + RL - 1) / RL * RL);
        chunks = (int)((M + rpc - 1) / rpc);
        partial = (float *)yk_scratch(dev, stream, 13, sizeof(double) * (size_t)chunks * 2 * C + sizeof(float) * C);
        if (!partial) return YK_ERR_NOMEM;
        hipLaunchKernelGGL(bn_colreduce_bwd_v4_kernel, dim3(chunks, groups), dim3(256), 0, st, z, dy, (uint32_t)M, C, rpc, CW, RL, save_mean, save_invstd, gamma,
                           beta, act, alpha, partial);
    } else {
        chunks = bn_chunking((size_t)M, C, &rpc, &cwl);
        partial = (float *)yk_scratch(dev, stream, 13, sizeof(double) * (size_t)chunks * 2 * C + sizeof(float) * C);
        if (!partial) return YK_ERR_NOMEM;
        hipLaunchKernelGGL(bn_colreduce_kernel<false>, dim3(chunks, (C + (1 << cwl) - 1) >> cwl), dim3(256), 0, st, z, dy, (size_t)M, C, rpc, cwl,
                           save_mean, save_invstd, gamma, beta, act, alpha, (void *)partial);
    }
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, st, (const float *)partial, chunks, C, dbeta, dgamma);
    const size_t total = (size_t)M * C;
    if (C % 4 == 0)
        hipLaunchKernelGGL(bn_apply_bwd_kernel<4>, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, z, dy, total, C, 1.f / (float)M, save_mean,
                           save_invstd, gamma, beta, (const float *)dbeta, (const float *)dgamma, act, alpha, dz);
    else
        hipLaunchKernelGGL(bn_apply_bwd_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, z, dy, total, C, 1.f / (float)M, save_mean,
                           save_invstd, gamma, beta, (const float *)dbeta, (const float *)dgamma, act, alpha, dz);
    YK_HIP(hipGetLastError());
    return YK_OK;
}

// --------------------------------------------------------------------------------------------------------
// Conv + BatchNormalization forward in one call (round 6): the producer of z leaves the column partials of the batch statistics, so the
// tensor is not read a second time for them (bn_colreduce_kernel<true>: 57 launches, 0.43 ms of the configs[3] step).
//   unsplit GEMM       -> statistics epilogue per 64-row tile (yk_gemm_f32.h)
//   split-K GEMM       -> the pass that adds the K slices writes z and the partials (below)
//   depthwise 3x3      -> computed in the reduction's own layout (channel lanes x row lanes), partials from registers
// then bn_stats_finish_kernel and the apply pass as in yk_bn_train_fwd_f32.
// --------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) splitk_sum_stats_kernel(const float *__restrict__ ws, int splits, size_t M, int N, float *__restrict__ c, int ldc,
                                                               int rows_per_chunk, int cw_log2, double *__restrict__ partial) {
    __shared__ double red[2][256];
    const int CW = 1 << cw_log2, RL = 256 >> cw_log2;
    const int cl = threadIdx.x & (CW - 1), rl = threadIdx.x >> cw_log2;
    const int n = blockIdx.y * CW + cl;
    const size_t r0 = (size_t)blockIdx.x * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    const size_t tot = M * (size_t)N;
    double s0 = 0, s1 = 0;
    if (n < N)
        for (size_t m = r0 + rl; m < r1; m += RL) {
            float v = 0.f;
            for (int z = 0; z < splits; ++z) v += ws[(size_t)z * tot + m * N + n];
            c[m * ldc + n] = v;
            s0 += (double)v;
            s1 += (double)v * (double)v;
        }
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    __syncthreads();
    if (rl == 0 && n < N) {
        double a = 0, b = 0;
        for (int k = 0; k < RL; ++k) {
            a += red[0][k * CW + cl];
            b += red[1][k * CW + cl];
        }
        partial[((size_t)blockIdx.x * 2 + 0) * N + n] = a;
        partial[((size_t)blockIdx.x * 2 + 1) * N + n] = b;
    }
}

// block = CW channel-vector lanes x RL row lanes (CW = C / V when that fits one workgroup: any number, not a power of two - 36 float4 lanes x 7
// rows for the 144-channel layers); grid (row chunks, channel groups).  A thread keeps its 9 x V weights in registers and walks rows rl, rl + RL, ...
template <int V>
__global__ void __launch_bounds__(256) dw_fwd_stats_kernel(conv_geom q, const float *__restrict__ x, const float *__restrict__ w, float *__restrict__ y,
                                                           int rows_per_chunk, int CW, int RL, double *__restrict__ partial) {
    __shared__ double red[2][256][V];
    const int cl = threadIdx.x % CW, rl = threadIdx.x / CW;
    const int c = (blockIdx.y * CW + cl) * V;
    const uint32_t M = (uint32_t)q.B * q.Ho * q.Wo;
    const uint32_t r0 = blockIdx.x * (uint32_t)rows_per_chunk, r1 = min(M, r0 + (uint32_t)rows_per_chunk);
    const bool on = rl < RL && c < q.C;
    double s0[V], s1[V];
#pragma unroll
    for (int k = 0; k < V; ++k) s0[k] = s1[k] = 0;
    if (on) {
        float wv[9][V];
#pragma unroll
        for (int t = 0; t < 9; ++t) ldv<V>(w + t * q.C + c, wv[t]);
        for (uint32_t m = r0 + rl; m < r1; m += RL) {
            const uint32_t row = m / (uint32_t)q.Wo;
            const int ox = (int)(m - row * q.Wo), b = (int)(row / (uint32_t)q.Ho), oy = (int)(row - (uint32_t)b * q.Ho);
            float s[V];
#pragma unroll
            for (int k = 0; k < V; ++k) s[k] = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {                       // same tap order as dw_fwd_kernel: bitwise the same z
                const int iy = oy * q.stride - q.pad_t + t / 3, ix = ox * q.stride - q.pad_l + t % 3;
                if ((unsigned)iy < (unsigned)q.Hi && (unsigned)ix < (unsigned)q.Wi) {
                    float xv[V];
                    ldv<V>(x + (((size_t)b * q.Hi + iy) * q.Wi + ix) * q.C + c, xv);
#pragma unroll
                    for (int k = 0; k < V; ++k) s[k] += xv[k] * wv[t][k];
                }
            }
            stv<V>(y + (size_t)m * q.C + c, s);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                s0[k] += (double)s[k];
                s1[k] += (double)s[k] * (double)s[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
        red[0][threadIdx.x][k] = s0[k];
        red[1][threadIdx.x][k] = s1[k];
    }
    __syncthreads();
    if (rl == 0 && c < q.C) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            double a = 0, b = 0;
            for (int r = 0; r < RL; ++r) {
                a += red[0][r * CW + cl][k];
                b += red[1][r * CW + cl][k];
            }
            partial[((size_t)blockIdx.x * 2 + 0) * q.C + c + k] = a;
            partial[((size_t)blockIdx.x * 2 + 1) * q.C + c + k] = b;
        }
    }
}

// statistics from `chunks` partials, then the apply pass
static int bn_finish_apply(const float *z, long long M, int C, const double *partial, int chunks, const float *gamma, const float *beta, float eps, int act,
                           float alpha, float *y, float *save_mean, float *save_invstd, float *moving_mean, float *moving_var, float momentum,
                           const float *res, hipStream_t st) {
    if (bn_cols_ok(M, C, z, y, res) && (((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0) {
        hipLaunchKernelGGL(bn_fwd_cols_kernel<3>, dim3((C + 7) / 8), dim3(1024), 0, st, z, (int)M, C, partial, chunks, 1.0 / (double)M, eps, gamma, beta, act,
                           alpha, y, save_mean, save_invstd, moving_mean, moving_var, momentum, res);
        YK_HIP(hipGetLastError());
        return YK_OK;
    }
    if (chunks > 1024)
        hipLaunchKernelGGL(bn_stats_finish_kernel<4>, dim3(C), dim3(256), 0, st, partial, chunks, C, 1.0 / (double)M, eps, save_mean, save_invstd,
                           moving_mean, moving_var, momentum);
    else
        hipLaunchKernelGGL(bn_stats_finish_kernel<1>, dim3((C + 3) / 4), dim3(256), 0, st, partial, chunks, C, 1.0 / (double)M, eps, save_mean, save_invstd,
                           moving_mean, moving_var, momentum);
    const size_t total = (size_t)M * C;
    if (C % 4 == 0)
        hipLaunchKernelGGL(bn_apply_fwd_kernel<4>, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, z, total, C, (const float *)save_mean,
                           (const float *)save_invstd, gamma, beta, act, alpha, y, res);
    else
        hipLaunchKernelGGL(bn_apply_fwd_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, z, total, C, (const float *)save_mean,
                           (const float *)save_invstd, gamma, beta, act, alpha, y, res);
    YK_HIP(hipGetLastError());
    return YK_OK;
}

extern "C" int yk_gemm_bn_fwd_f32(int M, int N, int K, const float *X, int ldx, const float *W, int ldw, float *z, const float *gamma, const float *beta,
                                  float eps, int act, float alpha, float *y, float *save_mean, float *save_invstd, float *moving_mean,
                                  float *moving_var, float momentum, const float *res, void *stream) {
    if (!X || !W || !z || !y || !gamma || !beta || !save_mean || !save_invstd || M <= 0 || N <= 0 || K <= 0) {
        yk_set_error("yk_gemm_bn_fwd_f32: bad argument");
        return YK_ERR_ARG;
    }
    int dev = yk_current_device();
    if (dev < 0) return YK_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    gemm_args g;
    g.M = M; g.N = N; g.K = K; g.lda = ldx; g.ldb = ldw; g.ldc = N; g.transA = 0; g.transB = 1;
    g.alpha = 1.f; g.beta = 0.f; g.A = X; g.B = W; g.C = z; g.ws = nullptr; g.stats = nullptr;
    const int s = gemm_splits(M, N, K);
    g.splitk = s;
    const int mt = (M + 63) / 64;
    int rpc = 0, cwl = 0;
    const int chunks = s > 1 ? bn_chunking((size_t)M, N, &rpc, &cwl) : mt;
    double *partial = (double *)yk_scratch(dev, stream, 13, sizeof(double) * (size_t)chunks * 2 * N + sizeof(float) * N);
    if (!partial) return YK_ERR_NOMEM;
    const dim3 grid(mt, (N + 63) / 64, s);
    if (s > 1) {
        g.ws = (float *)yk_scratch(dev, stream, 14, sizeof(float) * (size_t)s * M * N);
        if (!g.ws) return YK_ERR_NOMEM;
        launch_gemm_v2<false, true>(g, grid, st);
        hipLaunchKernelGGL(splitk_sum_stats_kernel, dim3(chunks, (N + (1 << cwl) - 1) >> cwl), dim3(256), 0, st, (const float *)g.ws, s, (size_t)M, N, z, N,
                           rpc, cwl, partial);
    } else {
        g.stats = partial;
        launch_gemm_fwd_stats(g, grid, st);
    }
    return bn_finish_apply(z, M, N, partial, chunks, gamma, beta, eps, act, alpha, y, save_mean, save_invstd, moving_mean, moving_var, momentum, res, st);
}

extern "C" int yk_dw3x3_bn_fwd_f32(const float *x, const float *w, int B, int Hi, int Wi, int C, int Ho, int Wo, int stride, int pad_t, int pad_l,
                                   float *z, const float *gamma, const float *beta, float eps, int act, float alpha, float *y, float *save_mean,
                                   float *save_invstd, float *moving_mean, float *moving_var, float momentum, const float *res, void *stream) {
    if (!x || !w || !z || !y || !gamma || !beta || !save_mean || !save_invstd || B <= 0 || C <= 0) {
        yk_set_error("yk_dw3x3_bn_fwd_f32: bad argument");
        return YK_ERR_ARG;
    }
    int dev = yk_current_device();
    if (dev < 0) return YK_ERR_NO_DEVICE;
    conv_geom q = {B, Hi, Wi, C, Ho, Wo, stride, pad_t, pad_l};
    const long long M = (long long)B * Ho * Wo;
    if (M >= (1ll << 31)) {
        yk_set_error("yk_dw3x3_bn_fwd_f32: more than 2^31 output pixels");
        return YK_ERR_ARG;
    }
    const int V = C % 4 == 0 ? 4 : 1, CV = C / V;
    // channel lanes: at most 64 per workgroup, the groups of equal width (144 float4 lanes = 3 x 48 with 5 row lanes; one group of 144 would leave one row
    // lane and 112 idle threads)
    const int groups = (CV + 63) / 64, CW = (CV + groups - 1) / groups, RL = 256 / CW;
    // ~4 rows per thread, 2 for the small layers (a thread's rows are a serial chain of load -> multiply-add -> store: 16 rows per thread measured
    // 37-61 us per launch whatever the size, r6c44), at most 8192 chunks (past 1024 the wide finishing kernel folds them)
    const int rpt = M <= 8192 ? 2 : 4;
    int chunks = (int)std::min<long long>(8192, (M + RL * rpt - 1) / (RL * rpt));
    const int rpc = (int)(((M + chunks - 1) / chunks + RL - 1) / RL * RL);
    chunks = (int)((M + rpc - 1) / rpc);
    double *partial = (double *)yk_scratch(dev, stream, 13, sizeof(double) * (size_t)chunks * 2 * C + sizeof(float) * C);
    if (!partial) return YK_ERR_NOMEM;
    hipStream_t st = (hipStream_t)stream;
    if (V == 4) hipLaunchKernelGGL(dw_fwd_stats_kernel<4>, dim3(chunks, groups), dim3(256), 0, st, q, x, w, z, rpc, CW, RL, partial);
    else hipLaunchKernelGGL(dw_fwd_stats_kernel<1>, dim3(chunks, groups), dim3(256), 0, st, q, x, w, z, rpc, CW, RL, partial);
    return bn_finish_apply(z, M, C, partial, chunks, gamma, beta, eps, act, alpha, y, save_mean, save_invstd, moving_mean, moving_var, momentum, res, st);
}

// --------------------------------------------------------------------------------------------------------
// 3x3 convolutions as IMPLICIT GEMMs (round 6; yk_gemm_f32.h CONV = 1 / 2 / 3): no column matrix - rounds 2-5 wrote it (113 MB for the 14x20x704
// head conv), multiplied it, and for the data gradient wrote a column matrix of gradients and folded it (col2im).  Needs Cin % 4 == 0 (and
// Cout % 4 == 0, stride 1 for the data gradient); the 3-channel stem keeps im2col.
// --------------------------------------------------------------------------------------------------------
static int conv_gemm(gemm_args g, const conv_args &cv, int mode, double *stats_partial, int *chunks_out, int *rpc_out, int *cwl_out, int dev, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const int s = gemm_splits(g.M, g.N, g.K);
    g.splitk = s;
    g.ws = nullptr;
    g.stats = nullptr;
    if (s > 1) {
        g.ws = (float *)yk_scratch(dev, stream, 14, sizeof(float) * (size_t)s * g.M * g.N);
        if (!g.ws) return YK_ERR_NOMEM;
    } else if (stats_partial) {
        g.stats = stats_partial;
    }
    const dim3 grid((g.M + 63) / 64, (g.N + 63) / 64, s);
    if (mode == 1) {
        if (g.stats) hipLaunchKernelGGL((gemm_f32_conv_kernel<false, true, true, 1>), grid, dim3(256), 0, st, g, cv);
        else hipLaunchKernelGGL((gemm_f32_conv_kernel<false, true, false, 1>), grid, dim3(256), 0, st, g, cv);
    } else if (mode == 2) {
        hipLaunchKernelGGL((gemm_f32_conv_kernel<true, false, false, 2>), grid, dim3(256), 0, st, g, cv);
    } else {
        hipLaunchKernelGGL((gemm_f32_conv_kernel<false, false, false, 3>), grid, dim3(256), 0, st, g, cv);
    }
    if (s > 1) {
        const size_t tot = (size_t)g.M * g.N;
        if (stats_partial)
            hipLaunchKernelGGL(splitk_sum_stats_kernel, dim3(*chunks_out, (g.N + (1 << *cwl_out) - 1) >> *cwl_out), dim3(256), 0, st, (const float *)g.ws, s, (size_t)g.M, g.N,
                               g.C, g.ldc, *rpc_out, *cwl_out, stats_partial);
        else if (s >= 16 && tot <= 65536)
            hipLaunchKernelGGL(splitk_sum16_kernel, dim3((unsigned)((tot + 15) / 16)), dim3(256), 0, st, (const float *)g.ws, s, g.M, g.N, g.alpha, g.beta, g.C, g.ldc);
        else
            hipLaunchKernelGGL(splitk_sum_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float *)g.ws, s, g.M, g.N, g.alpha, g.beta, g.C, g.ldc);
    }
    YK_HIP(hipGetLastError());
    return YK_OK;
}
static bool conv3x3_ok(const char *who, const void *a, const void *b, const void *c, int B, int Ci, int Co, bool need_co4) {
    if (!a || !b || !c || B <= 0 || Ci <= 0 || Co <= 0) {
        yk_set_error("%s: bad argument", who);
        return false;
    }
    if (Ci % 4 || (need_co4 && Co % 4) || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15)) {
        yk_set_error("%s: needs Cin %% 4 == 0%s and 16-byte aligned tensors (use yk_im2col3x3_f32 + yk_gemm_f32 otherwise)", who, need_co4 ? ", Cout % 4 == 0" : "");
        return false;
    }
    return true;
}
// z = conv3x3(x, w) [+ BatchNormalization(training) + activation (+ residual) -> y when gamma is given, as yk_gemm_bn_fwd_f32]; w: [Co][9 * Ci]
extern "C" int yk_conv3x3_bn_fwd_f32(const float *x, const float *w, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int stride, int pad_t, int pad_l, int Co,
                                     float *z, const float *gamma, const float *beta, float eps, int act, float alpha, float *y, float *save_mean,
                                     float *save_invstd, float *moving_mean, float *moving_var, float momentum, const float *res, void *stream) {
    if (!conv3x3_ok("yk_conv3x3_bn_fwd_f32", x, w, z, B, Ci, Co, false)) return (Ci % 4) ? YK_ERR_UNSUPPORTED : YK_ERR_ARG;
    if (gamma && (!beta || !y || !save_mean || !save_invstd)) {
        yk_set_error("yk_conv3x3_bn_fwd_f32: bad BatchNormalization argument");
        return YK_ERR_ARG;
    }
    int dev = yk_current_device();
    if (dev < 0) return YK_ERR_NO_DEVICE;
    const long long Mll = (long long)B * Ho * Wo;
    if (Mll >= (1ll << 31) / 9) {
        yk_set_error("yk_conv3x3_bn_fwd_f32: too many pixels");
        return YK_ERR_ARG;
    }
    const int M = (int)Mll;
    gemm_args g;
    g.M = M; g.N = Co; g.K = 9 * Ci; g.lda = 9 * Ci; g.ldb = 9 * Ci; g.ldc = Co; g.transA = 0; g.transB = 1;
    g.alpha = 1.f; g.beta = 0.f; g.A = x; g.B = w; g.C = z;
    conv_args cv = {{B, Hi, Wi, Ci, Ho, Wo, stride, pad_t, pad_l}, Co};
    if (!gamma) return conv_gemm(g, cv, 1, nullptr, nullptr, nullptr, nullptr, dev, stream);
    const int s = gemm_splits(M, Co, 9 * Ci), mt = (M + 63) / 64;
    int rpc = 0, cwl = 0;
    int chunks = s > 1 ? bn_chunking((size_t)M, Co, &rpc, &cwl) : mt;
    double *partial = (double *)yk_scratch(dev, stream, 13, sizeof(double) * (size_t)chunks * 2 * Co + sizeof(float) * Co);
    if (!partial) return YK_ERR_NOMEM;
    const int rc = conv_gemm(g, cv, 1, partial, &chunks, &rpc, &cwl, dev, stream);
    if (rc != YK_OK) return rc;
    return bn_finish_apply(z, M, Co, partial, chunks, gamma, beta, eps, act, alpha, y, save_mean, save_invstd, moving_mean, moving_var, momentum, res,
                           (hipStream_t)stream);
}
// dw [Co][9 * Ci] = dz^T * col(x)      (tf Conv2DBackpropFilter)
extern "C" int yk_conv3x3_bwd_weight_f32(const float *x, const float *dz, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int stride, int pad_t, int pad_l,
                                         int Co, float *dw, void *stream) {
    if (!conv3x3_ok("yk_conv3x3_bwd_weight_f32", x, dz, dw, B, Ci, Co, true)) return (Ci % 4 || Co % 4) ? YK_ERR_UNSUPPORTED : YK_ERR_ARG;
    int dev = yk_current_device();
    if (dev < 0) return YK_ERR_NO_DEVICE;
    gemm_args g;
    g.M = Co; g.N = 9 * Ci; g.K = B * Ho * Wo; g.lda = Co; g.ldb = 9 * Ci; g.ldc = 9 * Ci; g.transA = 1; g.transB = 0;
    g.alpha = 1.f; g.beta = 0.f; g.A = dz; g.B = x; g.C = dw;
    conv_args cv = {{B, Hi, Wi, Ci, Ho, Wo, stride, pad_t, pad_l}, Co};
    return conv_gemm(g, cv, 2, nullptr, nullptr, nullptr, nullptr, dev, stream);
}
// dx = the transposed convolution of dz (stride 1)      (tf Conv2DBackpropInput)
extern "C" int yk_conv3x3_bwd_data_f32(const float *dz, const float *w, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int stride, int pad_t, int pad_l, int Co,
                                       float *dx, void *stream) {
    if (!conv3x3_ok("yk_conv3x3_bwd_data_f32", dz, w, dx, B, Ci, Co, true)) return (Ci % 4 || Co % 4) ? YK_ERR_UNSUPPORTED : YK_ERR_ARG;
    if (stride != 1) {
        yk_set_error("yk_conv3x3_bwd_data_f32: stride %d (only stride 1; use yk_gemm_f32 + yk_col2im3x3_f32)", stride);
        return YK_ERR_UNSUPPORTED;
    }
    int dev = yk_current_device();
    if (dev < 0) return YK_ERR_NO_DEVICE;
    gemm_args g;
    g.M = B * Hi * Wi; g.N = Ci; g.K = 9 * Co; g.lda = 9 * Co; g.ldb = Ci; g.ldc = Ci; g.transA = 0; g.transB = 0;
    g.alpha = 1.f; g.beta = 0.f; g.A = dz; g.B = w; g.C = dx;
    conv_args cv = {{B, Hi, Wi, Ci, Ho, Wo, stride, pad_t, pad_l}, Co};
    return conv_gemm(g, cv, 3, nullptr, nullptr, nullptr, nullptr, dev, stream);
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
__global__ void __launch_bounds__(256) colsum_from_stats_kernel(const double *__restrict__ partial, int chunks, int C, float *__restrict__ out) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double a = 0;
    for (int k = lane; k < chunks; k += 64) a += partial[((size_t)k * 2 + 0) * C + c];
    a = wave_sum(a);
    if (lane == 0) out[c] = (float)a;
}
extern "C" int yk_colsum_f32(const float *x, long long M, int C, float *out, void *stream) {
    int dev = yk_current_device();
    if (dev < 0) return YK_ERR_NO_DEVICE;
    int rpc, cwl;
    const int chunks = bn_chunking((size_t)M, C, &rpc, &cwl);
    double *partial = (double *)yk_scratch(dev, stream, 13, sizeof(double) * (size_t)chunks * 2 * C + sizeof(float) * C);
    if (!partial) return YK_ERR_NOMEM;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_colreduce_kernel<true>, dim3(chunks, (C + (1 << cwl) - 1) >> cwl), dim3(256), 0, st, x, (const float *)nullptr, (size_t)M,
                       C, rpc, cwl, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, 0, 0.f,
                       (void *)partial);
    hipLaunchKernelGGL(colsum_from_stats_kernel, dim3((C + 3) / 4), dim3(256), 0, st, (const double *)partial, chunks, C, out);
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
