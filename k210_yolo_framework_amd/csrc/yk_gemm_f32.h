// yk_gemm_f32.h — fp32 GEMM of the training step on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains), included by yk_train.hip.
//
//   C[M,N] = alpha * op(A) * op(B) + beta * C,  row-major; the three shapes the step uses are
//     forward          Z  = X  * W^T     (TA = 0, TB = 1: both operands k-contiguous)                keras Conv2D forward
//     data gradient    dX = dZ * W       (TA = 0, TB = 0)
//     weight gradient  dW = dZ^T * X     (TA = 1, TB = 0: both operands contiguous along m / n)      reduction over the pixels
//
// 64x64 tile, 2x2 waves, 2x2 MFMA tiles per wave, BK = 32, 16-byte global loads along whichever axis is contiguous, register
// prefetch of the next k-tile under the MFMAs, two LDS buffers -> one barrier per k-tile.  LDS layout per operand:
//   k-contiguous operand  -> tile stored m-major  [64][34]: lanes (row = lane & 15, k = lane >> 4) hit banks 2*row + k: conflict-free
//   m-contiguous operand  -> tile stored k-major  [32][80]: the read of 16 consecutive floats of row k is conflict-free (80 = 16 mod 32)
// The accumulation order over k is fixed by the tiling, so results are bitwise reproducible (no atomics; split-K writes slabs that a
// finishing pass adds in order).  Per k-tile a wave issues 32 MFMAs (32 cycles each on its SIMD) for 32 ds_read_b32: the loop is
// MFMA-bound once the loads are hidden.
#pragma once

// STATS (forward of a conv followed by BatchNormalization, unsplit K): the epilogue also leaves the column sums of its 64 rows of C -
// sum and sum of squares in double, partial[(m tile * 2 + {0,1}) * N + n] - so the batch statistics need no second pass over C.
template <bool TA, bool TB>
struct gemm_v2_lds {
    static constexpr int BK = 32, LDM = 34, LDK = 80;
    static constexpr int SA = TA ? BK * LDK : 64 * LDM, SB = TB ? 64 * LDM : BK * LDK;      // floats per operand tile
    static constexpr int FLOATS = 2 * (SA + SB);
};
// one 64x64 tile (tile indices bx, by; K slice bz) of the problem g; lds: gemm_v2_lds<TA, TB>::FLOATS floats, 16-byte aligned
// CONV (round 6): a 3x3 convolution WITHOUT the column matrix - the operand that im2col would have written is gathered by the loader
// (16-byte pieces: a float4 of channels never straddles a tap because Cin % 4 == 0):
//   1  forward          Z  = col(X) * W^T         TA = 0, TB = 1   A[m][k]:  m -> (b, oy, ox), k -> (tap, c): X at the tap's input pixel, or zeros
//   2  weight gradient  dW = dZ^T * col(X)        TA = 1, TB = 0   B[k][n]:  k -> pixel, n -> (tap, c)
//   3  data gradient    dX = col'(dZ) * W'        TA = 0, TB = 0   A[m][k']: m -> input pixel, k' -> (tap, co): dZ at the output pixel that tap
//                       (stride 1)                                 connects to it; B[k'][n] = W[co][tap][n]
// Forward and weight gradient add in the same order as im2col + GEMM (bitwise the same result).
struct conv_args {
    conv_geom q;
    int co;
};
template <bool TA, bool TB, bool VEC, bool STATS, int CONV = 0>
__device__ __forceinline__ void gemm_f32_v2_tile(const gemm_args g, const int bx, const int by, const int bz, float *__restrict__ lds,
                                                 const conv_args cv = conv_args()) {   // g BY VALUE: a reference to the kernel's argument struct put it on the stack (scratch loads in the k loop)
    static_assert(CONV == 0 || VEC, "the gathered operands are loaded 16 bytes at a time");
    constexpr int BK = 32, LDM = 34, LDK = 80;
    constexpr int SA = gemm_v2_lds<TA, TB>::SA, SB = gemm_v2_lds<TA, TB>::SB;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = bx * 64, n0 = by * 64;
    const int nk = (g.K + BK - 1) / BK;
    const int per = (nk + g.splitk - 1) / g.splitk;
    const int kb = bz * per, ke = min(nk, kb + per);
    floatx4t acc[2][2];                                      // [n tile][m tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = floatx4t{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;

    // global -> registers: two float4 per operand per thread
    floatx4t ra[2], rb[2];
    auto fetch = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * 256;                     // 512 float4 per 64x32 tile
            {
                // TA: rows are k (32), 16 float4 along m;  !TA: rows are m (64), 8 float4 along k
                const int r = TA ? e >> 4 : e >> 3, q = TA ? (e & 15) * 4 : (e & 7) * 4;
                const int m = TA ? m0 + q : m0 + r, k = TA ? k0 + r : k0 + q;
                const float *p = TA ? g.A + (size_t)k * g.lda + m : g.A + (size_t)m * g.lda + k;
                floatx4t v = {0.f, 0.f, 0.f, 0.f};
                if (CONV == 1 || CONV == 3) {
                    // m -> pixel (b, y, x): of the OUTPUT (1) / of the INPUT (3);  k -> (tap, channel)
                    const int W_ = CONV == 1 ? cv.q.Wo : cv.q.Wi, H_ = CONV == 1 ? cv.q.Ho : cv.q.Hi, CK = CONV == 1 ? cv.q.C : cv.co;
                    const int row = m / W_, px = m - row * W_, b = row / H_, py = row - b * H_;
                    const int tap = k / CK, c = k - tap * CK, ky = tap / 3, kx = tap - ky * 3;
                    if (m < g.M && k < g.K) {
                        if (CONV == 1) {
                            const int iy = py * cv.q.stride - cv.q.pad_t + ky, ix = px * cv.q.stride - cv.q.pad_l + kx;
                            if ((unsigned)iy < (unsigned)cv.q.Hi && (unsigned)ix < (unsigned)cv.q.Wi)
                                v = *reinterpret_cast<const floatx4t *>(g.A + (((size_t)b * cv.q.Hi + iy) * cv.q.Wi + ix) * cv.q.C + c);
                        } else {
                            const int oy = py + cv.q.pad_t - ky, ox = px + cv.q.pad_l - kx;
                            if ((unsigned)oy < (unsigned)cv.q.Ho && (unsigned)ox < (unsigned)cv.q.Wo)
                                v = *reinterpret_cast<const floatx4t *>(g.A + (((size_t)b * cv.q.Ho + oy) * cv.q.Wo + ox) * cv.co + c);
                        }
                    }
                } else if (VEC) {
                    if (m < g.M && k < g.K) v = *reinterpret_cast<const floatx4t *>(p);
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int mm = TA ? m + u : m, kk = TA ? k : k + u;
                        if (mm < g.M && kk < g.K) v[u] = TA ? p[u] : p[u];
                    }
                }
                ra[i] = v;
            }
            {
                // TB: B is [N][K] (rows n, float4 along k);  !TB: B is [K][N] (rows k, float4 along n)
                const int r = TB ? e >> 3 : e >> 4, q = TB ? (e & 7) * 4 : (e & 15) * 4;
                const int n = TB ? n0 + r : n0 + q, k = TB ? k0 + q : k0 + r;
                const float *p = TB ? g.B + (size_t)n * g.ldb + k : g.B + (size_t)k * g.ldb + n;
                floatx4t v = {0.f, 0.f, 0.f, 0.f};
                if (CONV == 2) {
                    // k -> output pixel (b, oy, ox);  n -> (tap, channel): X at the tap's input pixel
                    const int row = k / cv.q.Wo, ox = k - row * cv.q.Wo, b = row / cv.q.Ho, oy = row - b * cv.q.Ho;
                    const int tap = n / cv.q.C, c = n - tap * cv.q.C, ky = tap / 3, kx = tap - ky * 3;
                    const int iy = oy * cv.q.stride - cv.q.pad_t + ky, ix = ox * cv.q.stride - cv.q.pad_l + kx;
                    if (n < g.N && k < g.K && (unsigned)iy < (unsigned)cv.q.Hi && (unsigned)ix < (unsigned)cv.q.Wi)
                        v = *reinterpret_cast<const floatx4t *>(g.B + (((size_t)b * cv.q.Hi + iy) * cv.q.Wi + ix) * cv.q.C + c);
                } else if (CONV == 3) {
                    // k' -> (tap, co): row co of W [co][9 * ci], the tap's ci entries
                    const int tap = k / cv.co, cc = k - tap * cv.co;
                    if (n < g.N && k < g.K) v = *reinterpret_cast<const floatx4t *>(g.B + ((size_t)cc * 9 + tap) * cv.q.C + n);
                } else if (VEC) {
                    if (n < g.N && k < g.K) v = *reinterpret_cast<const floatx4t *>(p);
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int nn = TB ? n : n + u, kk = TB ? k + u : k;
                        if (nn < g.N && kk < g.K) v[u] = p[u];
                    }
                }
                rb[i] = v;
            }
        }
    };
    auto stash = [&](int buf) {
        float *As = lds + buf * (SA + SB), *Bs = As + SA;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * 256;
            if (TA) {
                *reinterpret_cast<floatx4t *>(As + (e >> 4) * LDK + (e & 15) * 4) = ra[i];            // [k][m]
            } else {
                float *d = As + (e >> 3) * LDM + (e & 7) * 4;                                          // [m][k], 8-byte aligned rows
                *reinterpret_cast<float2 *>(d) = make_float2(ra[i][0], ra[i][1]);
                *reinterpret_cast<float2 *>(d + 2) = make_float2(ra[i][2], ra[i][3]);
            }
            if (TB) {
                float *d = Bs + (e >> 3) * LDM + (e & 7) * 4;                                          // [n][k]
                *reinterpret_cast<float2 *>(d) = make_float2(rb[i][0], rb[i][1]);
                *reinterpret_cast<float2 *>(d + 2) = make_float2(rb[i][2], rb[i][3]);
            } else {
                *reinterpret_cast<floatx4t *>(Bs + (e >> 4) * LDK + (e & 15) * 4) = rb[i];            // [k][n]
            }
        }
    };
    if (kb < ke) {
        fetch(kb);
        stash(0);
    }
    __syncthreads();
    for (int kt = kb; kt < ke; ++kt) {
        const int buf = (kt - kb) & 1;
        if (kt + 1 < ke) fetch(kt + 1);                      // next tile's global loads fly under this tile's MFMAs
        const float *As = lds + buf * (SA + SB), *Bs = As + SA;
#pragma unroll
        for (int k4 = 0; k4 < BK; k4 += 4) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = TA ? As[(k4 + fq) * LDK + wm * 32 + i * 16 + fr] : As[(wm * 32 + i * 16 + fr) * LDM + k4 + fq];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = TB ? Bs[(wn * 32 + j * 16 + fr) * LDM + k4 + fq] : Bs[(k4 + fq) * LDK + wn * 32 + j * 16 + fr];
            // operand roles swapped (matrix B feeds MFMA operand A): each lane ends up with 4 CONSECUTIVE n of one m
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[i], acc[j][i], 0, 0, 0);
        }
        if (kt + 1 < ke) stash(buf ^ 1);
        __syncthreads();
    }
    // D layout: row = (lane>>4)*4 + r -> n, col = lane&15 -> m
    const bool slab = g.splitk > 1;
    float *base = slab ? g.ws + (size_t)bz * g.M * g.N : g.C;
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
    if constexpr (STATS) {
        // the tile through LDS (the operand buffers are free: the k loop ended on a barrier): thread (column n, row quarter q) adds 16 rows,
        // the four quarters are added in order.  Rows past M hold zeros (their A rows were loaded as zeros).
        constexpr int LT = 65;
        float *tile = lds;
        double *red = reinterpret_cast<double *>(lds + 64 * LT);
        static_assert(64 * LT % 2 == 0 && 64 * LT + 2 * 2 * 4 * 64 <= 2 * (SA + SB), "statistics staging fits the operand buffers");
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) tile[(wm * 32 + i * 16 + fr) * LT + wn * 32 + j * 16 + fq * 4 + r] = acc[j][i][r] * g.alpha;
        __syncthreads();
        const int n = tid & 63, q = tid >> 6;
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const double v = (double)tile[(q * 16 + m) * LT + n];
            s0 += v;
            s1 += v * v;
        }
        red[(0 * 4 + q) * 64 + n] = s0;
        red[(1 * 4 + q) * 64 + n] = s1;
        __syncthreads();
        if (tid < 128) {
            const int jj = tid >> 6;
            const double t = ((red[(jj * 4 + 0) * 64 + n] + red[(jj * 4 + 1) * 64 + n]) + red[(jj * 4 + 2) * 64 + n]) + red[(jj * 4 + 3) * 64 + n];
            if (n0 + n < g.N) g.stats[((size_t)bx * 2 + jj) * g.N + n0 + n] = t;
        }
    }
}

template <bool TA, bool TB, bool VEC, bool STATS = false>
__global__ void __launch_bounds__(256) gemm_f32_v2_kernel(const gemm_args g) {
    __shared__ __attribute__((aligned(16))) float lds[gemm_v2_lds<TA, TB>::FLOATS];
    gemm_f32_v2_tile<TA, TB, VEC, STATS>(g, blockIdx.x, blockIdx.y, blockIdx.z, lds);
}

template <bool TA, bool TB, bool STATS, int CONV>
__global__ void __launch_bounds__(256) gemm_f32_conv_kernel(const gemm_args g, const conv_args cv) {
    __shared__ __attribute__((aligned(16))) float lds[gemm_v2_lds<TA, TB>::FLOATS];
    gemm_f32_v2_tile<TA, TB, true, STATS, CONV>(g, blockIdx.x, blockIdx.y, blockIdx.z, lds, cv);
}

// GROUPED launch (round 6): up to YK_GROUP_MAX independent problems of one layout in ONE launch - the weight gradients of a whole backward
// pass (37 GEMMs of 1-30 tiles each, every one split over K to reach ~1000 workgroups, every one followed by its slice-adding launch).  The
// problems ride in the kernel arguments; workgroup w belongs to the problem i with first[i] <= w < first[i + 1].  Same tile code; the host sizes
// the K slices for the group as a whole (yk_gemm_f32_grouped).
#define YK_GROUP_MAX 36
struct gemm_group {
    int count;
    int first[YK_GROUP_MAX + 1];
    gemm_args p[YK_GROUP_MAX];
};
static_assert(sizeof(gemm_group) <= 4096, "the group rides in the kernel arguments");
template <bool TA, bool TB, bool VEC>
__global__ void __launch_bounds__(256) gemm_f32_grouped_kernel(const gemm_group G) {
    __shared__ __attribute__((aligned(16))) float lds[gemm_v2_lds<TA, TB>::FLOATS];
    int i = 0;
    while (i + 1 < G.count && (int)blockIdx.x >= G.first[i + 1]) ++i;
    const gemm_args g = G.p[i];
    const int local = (int)blockIdx.x - G.first[i], tm = (g.M + 63) / 64, tn = (g.N + 63) / 64;
    const int bx = local % tm, r = local / tm;
    gemm_f32_v2_tile<TA, TB, VEC, false>(g, bx, r % tn, r / tn, lds);
}

template <bool TA, bool TB>
static void launch_gemm_v2(const gemm_args &g, dim3 grid, hipStream_t st) {
    // 16-byte loads need the contiguous axis to be a whole number of float4 with 16-byte aligned rows
    const bool va = TA ? (g.M % 4 == 0) : (g.K % 4 == 0), vb = TB ? (g.K % 4 == 0) : (g.N % 4 == 0);
    const bool vec = va && vb && g.lda % 4 == 0 && g.ldb % 4 == 0 && (((uintptr_t)g.A | (uintptr_t)g.B) & 15) == 0;
    if (vec) hipLaunchKernelGGL((gemm_f32_v2_kernel<TA, TB, true>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_f32_v2_kernel<TA, TB, false>), grid, dim3(256), 0, st, g);
}

// forward with the statistics epilogue (unsplit K, alpha = 1, beta = 0)
static void launch_gemm_fwd_stats(const gemm_args &g, dim3 grid, hipStream_t st) {
    const bool vec = g.K % 4 == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0 && (((uintptr_t)g.A | (uintptr_t)g.B) & 15) == 0;
    if (vec) hipLaunchKernelGGL((gemm_f32_v2_kernel<false, true, true, true>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_f32_v2_kernel<false, true, false, true>), grid, dim3(256), 0, st, g);
}
