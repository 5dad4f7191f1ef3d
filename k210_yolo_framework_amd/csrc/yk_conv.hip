// yk_conv.hip — gfx950 kernels of the conv stack (models/yolonet.py, keras_mobilenet*.py layers).
//
//  igemm_kernel     Conv2D 1x1 / 3x3 (stride 1|2, explicit top/left pad) as an implicit GEMM on
//                   v_mfma_f32_16x16x32_f16; optional virtual input = Concatenate([UpSampling2D(2)(a), b]);
//                   epilogue = folded BatchNorm (fp32 scale/bias) + LeakyReLU/ReLU/ReLU6 (+ residual Add),
//                   staged through LDS so global stores are 16 B per lane and row-contiguous.
//                   Operands are swapped (A := weights, B := pixels) so that each lane's four
//                   accumulator registers are four CONSECUTIVE output channels of one pixel.
//                   Small-M / large-K layers (the 7x10 and 14x20 head convs at batch 32) run split-K:
//                   partial fp32 slabs + a deterministic finishing pass (splitk_reduce_kernel).
//  fused_dwpw       DepthwiseConv2D 3x3 + BN + ReLU -> Conv2D 1x1 + BN + LeakyReLU in one launch.
//  first_conv       the 3-channel stem conv, reading u8 frames with Helper._process_img's
//                   `img / np.max(img)` fused in (per-image LUT), or fp32 input.
//  dw_kernel        standalone DepthwiseConv2D 3x3, NHWC, 8 channels (16 B) per lane.
//  pool_kernel      MaxPooling2D 2x2 'same' (stride 2 and the stride-1 case of tiny_yolo).
//  u8_max_kernel    per-image max for the normalisation (one workgroup per image, no atomics).
#include <stdlib.h>

#include "yk_conv.h"

// y = min(max(v, v*slope), cap): branch-free LeakyReLU / ReLU / ReLU6 / identity (yk_act_params)
__device__ __forceinline__ float yk_actf(float v, float slope, float cap) { return fminf(fmaxf(v, v * slope), cap); }

__device__ __forceinline__ uint32_t yk_div(uint32_t n, yk_fastdiv d) { return (__umulhi(n, d.mul) + n) >> d.shift; }

// acc += f16(lo|hi half of a) * f16(lo|hi half of b): one VALU op, fp32 accumulate, no conversions
__device__ __forceinline__ void fma_mix_lo(float &acc, uint32_t a, uint32_t b) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void fma_mix_hi(float &acc, uint32_t a, uint32_t b) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(a), "v"(b));
}

// bijective XCD-aware remap of a 1-D tile index: consecutive tiles land on the same XCD (block b runs
// on XCD b%8), so halo rows / weight panels shared by neighbouring tiles hit that XCD's L2.
__device__ __forceinline__ int yk_xcd_tile(int bid, int nt) {
    const int q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

#define YK_MAXP 32  /* partial maxima per image (u8_max_kernel -> first_conv_kernel) */
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) unsigned char yk_smem[];

// ---- depthwise 3x3 work item = (output pixel, 8 channels) --------------------------------------
// Taps are fetched with raw buffer loads: out-of-image taps get an offset >= the buffer size and the
// hardware returns zeros (Keras zero padding) — no per-tap branches, one 32-bit add per tap.
#define YK_OOB 0x40000000u
__device__ __forceinline__ void dw_issue(const igemm_args &a, const __amdgpu_buffer_rsrc_t rs, uint32_t m, uint32_t coff,
                                         u32x4 (&x)[9]) {
    const uint32_t hw = a.Ho * a.Wo, px = a.c0p * 2u, rowb = a.dw_Wi * px;
    const uint32_t b = yk_div(m, a.fd_hw), rem = m - b * hw;
    const uint32_t oy = yk_div(rem, a.fd_wo), ox = rem - oy * a.Wo;
    const int iy0 = (int)oy * a.dw_stride - a.dw_pad_t, ix0 = (int)ox * a.dw_stride - a.dw_pad_l;
    const uint32_t boff = b * (a.dw_Hi * rowb) + coff;
    uint32_t ro[3], co[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        ro[k] = ((unsigned)(iy0 + k) < (unsigned)a.dw_Hi) ? boff + (uint32_t)(iy0 + k) * rowb : YK_OOB;
        co[k] = ((unsigned)(ix0 + k) < (unsigned)a.dw_Wi) ? (uint32_t)(ix0 + k) * px : YK_OOB;
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) x[ky * 3 + kx] = __builtin_amdgcn_raw_buffer_load_b128(rs, ro[ky] + co[kx], 0, 0);
}
template <bool CAPPED>
__device__ __forceinline__ float yk_act2(float v, float slope, float cap) {
    v = fmaxf(v, v * slope);
    return CAPPED ? fminf(v, cap) : v;
}
// wl: this octet's depthwise weights in LDS, tap stride Cp halfs
template <bool CAPPED>
__device__ __forceinline__ half8 dw_finish(const u32x4 (&x)[9], const yk_half *wl, int Cp, const float4 sc0, const float4 sc1,
                                           const float4 bs0, const float4 bs1, float slope, float cap) {
    float d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const u32x4 w = *reinterpret_cast<const u32x4 *>(wl + (size_t)t * Cp);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            fma_mix_lo(d[2 * j], x[t][j], w[j]);
            fma_mix_hi(d[2 * j + 1], x[t][j], w[j]);
        }
    }
    half8 h;
    h[0] = (yk_half)yk_act2<CAPPED>(d[0] * sc0.x + bs0.x, slope, cap);
    h[1] = (yk_half)yk_act2<CAPPED>(d[1] * sc0.y + bs0.y, slope, cap);
    h[2] = (yk_half)yk_act2<CAPPED>(d[2] * sc0.z + bs0.z, slope, cap);
    h[3] = (yk_half)yk_act2<CAPPED>(d[3] * sc0.w + bs0.w, slope, cap);
    h[4] = (yk_half)yk_act2<CAPPED>(d[4] * sc1.x + bs1.x, slope, cap);
    h[5] = (yk_half)yk_act2<CAPPED>(d[5] * sc1.y + bs1.y, slope, cap);
    h[6] = (yk_half)yk_act2<CAPPED>(d[6] * sc1.z + bs1.z, slope, cap);
    h[7] = (yk_half)yk_act2<CAPPED>(d[7] * sc1.w + bs1.w, slope, cap);
    return h;
}
// conv epilogue for 4 consecutive channels of one pixel -> packed fp16
template <bool CAPPED>
__device__ __forceinline__ half4 epi4(const floatx4 c, const float4 sc, const float4 bs, float slope, float cap) {
    return half4{(yk_half)yk_act2<CAPPED>(c[0] * sc.x + bs.x, slope, cap), (yk_half)yk_act2<CAPPED>(c[1] * sc.y + bs.y, slope, cap),
                 (yk_half)yk_act2<CAPPED>(c[2] * sc.z + bs.z, slope, cap), (yk_half)yk_act2<CAPPED>(c[3] * sc.w + bs.w, slope, cap)};
}

template <int BM, int BN, int WM, int WN, int OUT, int TM, int TN>
__device__ __forceinline__ void igemm_epilogue(const igemm_args &a, floatx4 (&acc)[TM][TN], yk_half *lds, int m0, int n0, int tid,
                                               int lane, int wm, int wn, int zsplit = -1) {
    constexpr int NT = 64 * WM * WN;
    constexpr int CS_LD = BN + 8;
    const int fr = lane & 15;
    // ---- epilogue: lane holds channels n..n+3 (acc regs) of pixel m = lane&15
    const int nl4 = (lane >> 4) * 4;
    if constexpr (OUT == 2) {
        float *slab = a.slab + (size_t)(zsplit < 0 ? (int)blockIdx.z : zsplit) * a.M * a.ldn;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + (wm * TM + i) * 16 + fr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + (wn * TN + j) * 16 + nl4;
                if (m < a.M && n < a.ldn)
                    *reinterpret_cast<float4 *>(slab + (size_t)m * a.ldn + n) =
                        make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
        return;
    } else {
        float4 sc[TN], bs[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 16 + nl4;
            sc[j] = *reinterpret_cast<const float4 *>(a.scale + n);   // arrays padded past N with zeros
            bs[j] = *reinterpret_cast<const float4 *>(a.bias + n);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int ml = (wm * TM + i) * 16 + fr, m = m0 + ml;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nl = (wn * TN + j) * 16 + nl4, n = n0 + nl;
                const float v0 = yk_actf(acc[i][j][0] * sc[j].x + bs[j].x, a.slope, a.cap);
                const float v1 = yk_actf(acc[i][j][1] * sc[j].y + bs[j].y, a.slope, a.cap);
                const float v2 = yk_actf(acc[i][j][2] * sc[j].z + bs[j].z, a.slope, a.cap);
                const float v3 = yk_actf(acc[i][j][3] * sc[j].w + bs[j].w, a.slope, a.cap);
                if constexpr (OUT == 1) {
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
        if constexpr (OUT == 0) {
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
}

// LDS row pitch of the igemm operand tiles = BK + YK_LDPAD halfs.  ds_read_b128 is serviced in four groups of 16 lanes
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ... - MI355X_MICROARCH.md, LDS table) over 64 four-byte banks; with lane = (row&15,
// k-chunk = lane>>4) a pitch of 10 or 6 sixteen-byte units (BK 64 / 32, pad 16) makes every group hit 16 distinct 16-byte
// slots.  The earlier pad of 8 (pitch 9 / 5 units) made 7 of the 8 rows 4-11 collide with rows 0-3 / 12-15: 2-way conflicts
// on every fragment read.
#ifndef YK_LDPAD
#define YK_LDPAD 16
#endif

// =====================================================================================
// implicit GEMM.  OUT: 0 = fp16 through LDS, 1 = fp32 direct (network outputs), 2 = split-K slab
// =====================================================================================
template <int BM, int BN, int WM, int WN, int BK, int OUT, bool UNI>
__global__ void __launch_bounds__(64 * WM * WN) igemm_kernel(const igemm_args a) {
    constexpr int NT = 64 * WM * WN;
    constexpr int LD = BK + YK_LDPAD;               // LDS row pitch (halfs), see YK_LDPAD
    constexpr int CPR = BK / 8;                     // 16-byte chunks per tile row
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int A_VEC = BM * CPR, B_VEC = BN * CPR;
    constexpr int A_IT = (A_VEC + NT - 1) / NT, B_IT = (B_VEC + NT - 1) / NT;
    constexpr int STAGE = (BM + BN) * LD;
    yk_half *lds = reinterpret_cast<yk_half *>(yk_smem);   // dynamic: max(2 stages, output tile), see launch_cfg

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int m0 = yk_xcd_tile(blockIdx.x, gridDim.x) * BM, n0 = blockIdx.y * BN;
    const int kc = tid % CPR;                       // this thread's 8-wide k chunk inside a BK step
    const int Ctp = a.c0p + a.c1p;
    const int taps = a.ks * a.ks;
    const int H0 = a.up0 ? (a.Hi >> 1) : a.Hi, W0 = a.up0 ? (a.Wi >> 1) : a.Wi;

    // K range of this split
    const int nk_all = (a.K + BK - 1) / BK;
    const int per = (nk_all + a.split_k - 1) / a.split_k;
    const int kt0 = blockIdx.z * per;
    const int nk = min(per, nk_all - kt0);

    if constexpr (UNI) {
        // ------------------------------------------------------------------------------------------------
        // Uniform-tap fast path: (c0p+c1p) % BK == 0 and c0p % BK == 0, so within one k-step every thread is in the
        // SAME filter tap and the SAME concat source.  Tap / source / channel offset become scalars, the per-row work
        // per k-step shrinks to ~10 VALU, and every load is an unconditional raw buffer load (out-of-image taps,
        // rows >= M, columns >= N get an offset past num_records and return zeros): a branch-free loop body in which
        // hipcc can keep TWO k-steps of loads in flight behind counted s_waitcnt vmcnt(N).
        // ------------------------------------------------------------------------------------------------
        int ry0[A_IT], rx0[A_IT];
        uint32_t rbo0[A_IT], rbo1[A_IT], rmask[A_IT], wro[B_IT];
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int v = tid + it * NT, row = v / CPR, m = m0 + row;
            const bool ok = v < A_VEC && m < a.M;
            const uint32_t mm = ok ? m : 0;
            const uint32_t b = yk_div(mm, a.fd_hw), rem = mm - b * (a.Ho * a.Wo);
            const uint32_t oy = yk_div(rem, a.fd_wo), ox = rem - oy * a.Wo;
            ry0[it] = (int)oy * a.stride - a.pad_t;
            rx0[it] = (int)ox * a.stride - a.pad_l;
            rbo0[it] = b * (uint32_t)(H0 * W0 * a.c0p * 2) + kc * 16u;
            rbo1[it] = b * (uint32_t)(a.Hi * a.Wi * a.c1p * 2) + kc * 16u;
            uint32_t msk = 0;
            for (int t = 0; t < taps; ++t) {
                const int ky = (a.ks == 3) ? t / 3 : 0, kx = t - ky * a.ks;
                if (ok && (unsigned)(ry0[it] + ky) < (unsigned)a.Hi && (unsigned)(rx0[it] + kx) < (unsigned)a.Wi) msk |= 1u << t;
            }
            rmask[it] = msk;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int vv = tid + it * NT, row = vv / CPR, n = n0 + row;
            wro[it] = (vv < B_VEC && n < a.N) ? (uint32_t)(n * a.K) * 2u + kc * 16u : YK_OOB;
        }
        const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void *)a.in0, 0, a.in0_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void *)(a.in1 ? a.in1 : a.in0), 0, a.in1_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);
        u32x4 ra0[A_IT], rb0[B_IT], ra1[A_IT], rb1[B_IT];
        // Incremental addressing: (tap, cin) are loop-carried SCALARS; the per-row tap offsets `aoff` are recomputed
        // only when the segment (tap, concat source) changes, i.e. every Ctp/BK (8-12) k-steps; a k-step then costs
        // one vector add per load.  `lim` = first k-step that belongs to the next split (loads past it return zeros).
        const int lim = kt0 + nk;
        int step = kt0;
        uint32_t tap = yk_div((uint32_t)kt0 * BK, a.fd_ctp);
        int cin = kt0 * BK - (int)tap * Ctp;
        uint32_t aoff[A_IT];
        bool src1 = false;
        auto retap = [&]() {
            src1 = cin >= a.c0p;
            const int ky = (a.ks == 3) ? (int)tap / 3 : 0, kx = (int)tap - ky * a.ks;
            const bool tlive = (int)tap < taps;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const int iy = ry0[it] + ky, ix = rx0[it] + kx;
                const bool inb = tlive && ((rmask[it] >> tap) & 1u);
                uint32_t o;
                if (src1) {
                    o = rbo1[it] + (uint32_t)((iy * a.Wi + ix) * a.c1p - a.c0p) * 2u;     // + cin*2 added per step
                } else {
                    const int sy = a.up0 ? (iy >> 1) : iy, sx = a.up0 ? (ix >> 1) : ix;
                    o = rbo0[it] + (uint32_t)((sy * W0 + sx) * a.c0p) * 2u;
                }
                aoff[it] = inb ? o : YK_OOB;
            }
        };
        retap();
        auto gload = [&](u32x4 (&ra)[A_IT], u32x4 (&rbv)[B_IT]) {
            const bool live = step < lim;
            const uint32_t cs = (uint32_t)cin * 2u, ws = (uint32_t)step * (BK * 2u);
            if (src1) {
#pragma unroll
                for (int it = 0; it < A_IT; ++it)
                    ra[it] = __builtin_amdgcn_raw_buffer_load_b128(rs1, live ? aoff[it] + cs : YK_OOB, 0, 0);
            } else {
#pragma unroll
                for (int it = 0; it < A_IT; ++it)
                    ra[it] = __builtin_amdgcn_raw_buffer_load_b128(rs0, live ? aoff[it] + cs : YK_OOB, 0, 0);
            }
#pragma unroll
            for (int it = 0; it < B_IT; ++it) rbv[it] = __builtin_amdgcn_raw_buffer_load_b128(rsw, live ? wro[it] + ws : YK_OOB, 0, 0);
            ++step;
            cin += BK;
            if (cin == Ctp) {
                cin = 0;
                ++tap;
            }
            if (cin == 0 || cin == a.c0p) retap();       // segment change (uniform, rare)
        };
        auto sstore = [&](const u32x4 (&ra)[A_IT], const u32x4 (&rbv)[B_IT], int stage) {
            yk_half *As = lds + stage * STAGE, *Bs = As + BM * LD;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const int v = tid + it * NT;
                if (v < A_VEC) *reinterpret_cast<u32x4 *>(As + (v / CPR) * LD + kc * 8) = ra[it];
            }
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                const int v = tid + it * NT;
                if (v < B_VEC) *reinterpret_cast<u32x4 *>(Bs + (v / CPR) * LD + kc * 8) = rbv[it];
            }
        };
        floatx4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        const int fr = lane & 15, fk = (lane >> 4) * 8;
        auto compute = [&](int stage) {
            const yk_half *As = lds + stage * STAGE, *Bs = As + BM * LD;
#pragma unroll
            for (int ks = 0; ks < BK / 32; ++ks) {
                half8 wf[TN], xf[TM];
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    wf[j] = *reinterpret_cast<const half8 *>(Bs + ((wn * TN + j) * 16 + fr) * LD + ks * 32 + fk);
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    xf[i] = *reinterpret_cast<const half8 *>(As + ((wm * TM + i) * 16 + fr) * LD + ks * 32 + fk);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
            }
        };
        if (nk > 0) {
            gload(ra0, rb0);                           // step kt0
            sstore(ra0, rb0, 0);
            __syncthreads();
            gload(ra0, rb0);                           // S0 <- step kt0+1
            gload(ra1, rb1);                           // S1 <- step kt0+2
            for (int kt = 0; kt < nk; kt += 2) {      // steps past the end of this split load zeros
                compute(0);
                sstore(ra0, rb0, 1);
                gload(ra0, rb0);                       // kt+3
                __syncthreads();
                compute(1);
                sstore(ra1, rb1, 0);
                gload(ra1, rb1);                       // kt+4
                __syncthreads();
            }
        }
        igemm_epilogue<BM, BN, WM, WN, OUT, TM, TN>(a, acc, lds, m0, n0, tid, lane, wm, wn);
        return;
    }

    // ---- per-thread A rows (fixed over the K loop)
    int rb[A_IT], riy[A_IT], rix[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int v = tid + it * NT, row = v / CPR, m = m0 + row;
        if (v < A_VEC && m < a.M) {
            const uint32_t b = yk_div(m, a.fd_hw), rem = m - b * (a.Ho * a.Wo);
            const uint32_t oy = yk_div(rem, a.fd_wo), ox = rem - oy * a.Wo;
            rb[it] = b;
            riy[it] = (int)oy * a.stride - a.pad_t;
            rix[it] = (int)ox * a.stride - a.pad_l;
        } else {
            rb[it] = 0;
            riy[it] = -(1 << 28);
            rix[it] = 0;
        }
    }
    int kch = kt0 * BK + kc * 8, ktap = 0;          // (channel-in-tap, tap) of this thread's chunk
    if (kch >= Ctp) {
        ktap = kch / Ctp;
        kch -= ktap * Ctp;
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
            const int vv = tid + it * NT, row = vv / CPR, n = n0 + row, k = k0 + kc * 8;
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
            if (v < A_VEC) *reinterpret_cast<half8 *>(As + (v / CPR) * LD + kc * 8) = ra[it];
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int v = tid + it * NT;
            if (v < B_VEC) *reinterpret_cast<half8 *>(Bs + (v / CPR) * LD + kc * 8) = rbv[it];
        }
    };

    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fk = (lane >> 4) * 8;
    if (nk > 0) {
        gload(kt0 * BK);
        sstore(0);
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) gload((kt0 + kt + 1) * BK);
            const yk_half *As = lds + cur * STAGE, *Bs = As + BM * LD;
#pragma unroll
            for (int ks = 0; ks < BK / 32; ++ks) {
                half8 wf[TN], xf[TM];
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    wf[j] = *reinterpret_cast<const half8 *>(Bs + ((wn * TN + j) * 16 + fr) * LD + ks * 32 + fk);
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    xf[i] = *reinterpret_cast<const half8 *>(As + ((wm * TM + i) * 16 + fr) * LD + ks * 32 + fk);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
            }
            if (kt + 1 < nk) sstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }

    igemm_epilogue<BM, BN, WM, WN, OUT, TM, TN>(a, acc, lds, m0, n0, tid, lane, wm, wn);
}

// finishing pass of split-K: sum the slabs in a fixed order (deterministic), then the conv epilogue
template <bool OUT_F32>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const igemm_args a) {
    const int n4 = a.ldn >> 2;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)a.M * n4) return;
    const int m = (int)(idx / n4), n = (int)(idx - (size_t)m * n4) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *p = a.slab + (size_t)m * a.ldn + n;
    const size_t zs = (size_t)a.M * a.ldn;
    for (int z = 0; z < a.split_k; ++z) {
        const float4 v = *reinterpret_cast<const float4 *>(p + z * zs);
        s.x += v.x;
        s.y += v.y;
        s.z += v.z;
        s.w += v.w;
    }
    const float4 sc = *reinterpret_cast<const float4 *>(a.scale + n), bs = *reinterpret_cast<const float4 *>(a.bias + n);
    const float v0 = yk_actf(s.x * sc.x + bs.x, a.slope, a.cap), v1 = yk_actf(s.y * sc.y + bs.y, a.slope, a.cap);
    const float v2 = yk_actf(s.z * sc.z + bs.z, a.slope, a.cap), v3 = yk_actf(s.w * sc.w + bs.w, a.slope, a.cap);
    if constexpr (OUT_F32) {
        float *o = reinterpret_cast<float *>(a.out) + (size_t)m * a.outp + n;
        if (n + 0 < a.N) o[0] = v0;
        if (n + 1 < a.N) o[1] = v1;
        if (n + 2 < a.N) o[2] = v2;
        if (n + 3 < a.N) o[3] = v3;
    } else {
        if (n >= a.outp) return;
        half4 h = {(yk_half)v0, (yk_half)v1, (yk_half)v2, (yk_half)v3};
        if (a.res && n < a.resp) {
            const half4 r = *reinterpret_cast<const half4 *>(a.res + (size_t)m * a.resp + n);
            h = half4{(yk_half)((float)h[0] + (float)r[0]), (yk_half)((float)h[1] + (float)r[1]),
                      (yk_half)((float)h[2] + (float)r[2]), (yk_half)((float)h[3] + (float)r[3])};
        }
        *reinterpret_cast<half4 *>(reinterpret_cast<yk_half *>(a.out) + (size_t)m * a.outp + n) = h;
    }
}

// Split-K finishing pass fused with the 1x1 fp32-output conv that follows it (the y1 / y2 heads: Conv3x3+BN+Leaky ->
// Conv1x1(+bias), yolonet.py:27-29,37-38).  A workgroup owns 16 pixels: it sums their slabs exactly like
// splitk_reduce_kernel (same order, same epilogue, the fp16 tensor is still written - it is a plan tensor), keeps the 16 x C
// fp16 tile in LDS and runs the 1x1 conv on it with the MFMA sequence of the igemm kernels (k ascending in steps of 32), so
// the results are bit-identical to the two separate launches it replaces.  `a` = the split conv, `b` = the 1x1 conv.
__global__ void __launch_bounds__(256) reduce_pw_kernel(const igemm_args a, const igemm_args b) {
    yk_half *Xs = reinterpret_cast<yk_half *>(yk_smem);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int Kp = (b.c0p + 31) & ~31, LDX = Kp + YK_LDPAD;
    const int m0 = blockIdx.x * 16;
    const int n4 = Kp >> 2;
    const size_t zs = (size_t)a.M * a.ldn;
    // the 1x1 conv's weight fragments are requested first, so their latency hides behind the slab reduction
    const int fr = lane & 15, fk = (lane >> 4) * 8, nl4 = (lane >> 4) * 4;
    const int ntiles = (b.N + 15) >> 4;
    half8 wf[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int nrow = (wid + 4 * q) * 16 + fr;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            wf[q][k] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            if (wid + 4 * q < ntiles && nrow < b.N && k * 32 + fk < b.K) wf[q][k] = *reinterpret_cast<const half8 *>(b.w + (size_t)nrow * b.K + k * 32 + fk);
        }
    }
    for (int item = tid; item < 16 * n4; item += 256) {
        const int ml = item / n4, n = (item - ml * n4) * 4, m = m0 + ml;
        half4 h = {0, 0, 0, 0};
        if (m < a.M && n < a.ldn) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            const float *p = a.slab + (size_t)m * a.ldn + n;
            for (int z = 0; z < a.split_k; ++z) {
                const float4 v = *reinterpret_cast<const float4 *>(p + z * zs);
                s.x += v.x;
                s.y += v.y;
                s.z += v.z;
                s.w += v.w;
            }
            const float4 sc = *reinterpret_cast<const float4 *>(a.scale + n), bs = *reinterpret_cast<const float4 *>(a.bias + n);
            const float v0 = yk_actf(s.x * sc.x + bs.x, a.slope, a.cap), v1 = yk_actf(s.y * sc.y + bs.y, a.slope, a.cap);
            const float v2 = yk_actf(s.z * sc.z + bs.z, a.slope, a.cap), v3 = yk_actf(s.w * sc.w + bs.w, a.slope, a.cap);
            if (n < a.outp) {
                h = half4{(yk_half)v0, (yk_half)v1, (yk_half)v2, (yk_half)v3};
                *reinterpret_cast<half4 *>(reinterpret_cast<yk_half *>(a.out) + (size_t)m * a.outp + n) = h;
            }
        }
        *reinterpret_cast<half4 *>(Xs + ml * LDX + n) = h;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int t = wid + 4 * q;
        if (t >= ntiles) break;
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k * 32 < Kp) {
                const half8 xf = *reinterpret_cast<const half8 *>(Xs + fr * LDX + k * 32 + fk);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[q][k], xf, acc, 0, 0, 0);
            }
        }
        const int n = t * 16 + nl4, m = m0 + fr;
        const float4 sc = *reinterpret_cast<const float4 *>(b.scale + n), bs = *reinterpret_cast<const float4 *>(b.bias + n);
        const float v0 = yk_actf(acc[0] * sc.x + bs.x, b.slope, b.cap), v1 = yk_actf(acc[1] * sc.y + bs.y, b.slope, b.cap);
        const float v2 = yk_actf(acc[2] * sc.z + bs.z, b.slope, b.cap), v3 = yk_actf(acc[3] * sc.w + bs.w, b.slope, b.cap);
        if (m < b.M) {
            float *o = reinterpret_cast<float *>(b.out) + (size_t)m * b.outp + n;
            if (n + 0 < b.N) o[0] = v0;
            if (n + 1 < b.N) o[1] = v1;
            if (n + 2 < b.N) o[2] = v2;
            if (n + 3 < b.N) o[3] = v3;
        }
    }
}

// can the split conv `a` (fp16 out) and the conv `b` that reads its output be finished by reduce_pw_kernel?
bool yk_reduce_pw_ok(const igemm_args &a, const igemm_args &b, bool b_out_f32) {
    return a.split_k > 1 && !a.res && b_out_f32 && b.split_k <= 1 && b.ks == 1 && b.stride == 1 && !b.in1 && !b.up0 && !b.dw_w && !b.res &&
           b.in0 == reinterpret_cast<const yk_half *>(a.out) && b.c0p == a.outp && b.K == b.c0p && b.c0p % 8 == 0 && b.c0p <= 256 &&
           b.N <= 128;
}
int yk_launch_reduce_pw(const igemm_args &a, const igemm_args &b, hipStream_t st) {
    const int Kp = (b.c0p + 31) & ~31;
    const size_t lds = (size_t)16 * (Kp + YK_LDPAD) * 2;
    hipLaunchKernelGGL(reduce_pw_kernel, dim3((a.M + 15) / 16), dim3(256), lds, st, a, b);
    return YK_OK;
}

// =====================================================================================
// igemm_lin_kernel: the uniform-tap path again, with the K loop as ONE basic block.
// For an input that is not read through an upsample the tap offset is linear, off = P_row(src) + tapoff(tap, src) + cin*2,
// so a k-step needs no per-row recomputation: the per-row part is two precomputed registers (one per concat source) and
// everything tap-dependent is scalar.  No branch inside the loop means no PHI copies of the 64-128 accumulator registers
// (the branchy version spent ~3 v_accvgpr_mov/read/write per MFMA on them) and lets the scheduler interleave MFMA, LDS
// and buffer loads freely.  Loads past the split's last step or on a dead tap get the OOB offset and return zeros.
// =====================================================================================
template <int BM, int BN, int WM, int WN, int BK, int OUT, int PF>
__global__ void __launch_bounds__(64 * WM * WN) igemm_lin_kernel(const igemm_args a) {
    static_assert(PF == 2 || PF == 4, "register prefetch depth (k-steps in flight); even so the two LDS stages keep their parity");
    constexpr int NT = 64 * WM * WN;
    constexpr int LD = BK + YK_LDPAD;
    constexpr int CPR = BK / 8;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int A_VEC = BM * CPR, B_VEC = BN * CPR;
    static_assert(A_VEC % NT == 0 && B_VEC % NT == 0, "tile must be a whole number of 16-byte vectors per thread");
    constexpr int A_IT = A_VEC / NT, B_IT = B_VEC / NT;
    constexpr int STAGE = (BM + BN) * LD;
    yk_half *lds = reinterpret_cast<yk_half *>(yk_smem);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int m0 = yk_xcd_tile(blockIdx.x, gridDim.x) * BM, n0 = blockIdx.y * BN;
    const int kc = tid % CPR;
    const int Ctp = a.c0p + a.c1p;
    const int taps = a.ks * a.ks;
    const int nk_all = (a.K + BK - 1) / BK;
    const int per = (nk_all + a.split_k - 1) / a.split_k;
    const int kt0 = blockIdx.z * per;
    const int nk = min(per, nk_all - kt0);

    uint32_t P0[A_IT], P1[A_IT], rmask[A_IT], wro[B_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int v = tid + it * NT, row = v / CPR, m = m0 + row;
        const bool ok = m < a.M;
        const uint32_t mm = ok ? m : 0;
        const uint32_t b = yk_div(mm, a.fd_hw), rem = mm - b * (a.Ho * a.Wo);
        const uint32_t oy = yk_div(rem, a.fd_wo), ox = rem - oy * a.Wo;
        const int ry0 = (int)oy * a.stride - a.pad_t, rx0 = (int)ox * a.stride - a.pad_l;
        // may be "negative" (wraps) for halo rows; those taps are masked, and the sum with the tap offset is exact mod 2^32
        P0[it] = b * (uint32_t)(a.Hi * a.Wi * a.c0p * 2) + (uint32_t)((ry0 * a.Wi + rx0) * a.c0p) * 2u + kc * 16u;
        P1[it] = b * (uint32_t)(a.Hi * a.Wi * a.c1p * 2) + (uint32_t)((ry0 * a.Wi + rx0) * a.c1p - a.c0p) * 2u + kc * 16u;
        uint32_t msk = 0;
        for (int t = 0; t < taps; ++t) {
            const int ky = (a.ks == 3) ? t / 3 : 0, kx = t - ky * a.ks;
            if (ok && (unsigned)(ry0 + ky) < (unsigned)a.Hi && (unsigned)(rx0 + kx) < (unsigned)a.Wi) msk |= 1u << t;
        }
        rmask[it] = msk;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int n = n0 + (tid + it * NT) / CPR;
        wro[it] = (n < a.N) ? (uint32_t)(n * a.K) * 2u + kc * 16u : YK_OOB;
    }
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void *)a.in0, 0, a.in0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void *)(a.in1 ? a.in1 : a.in0), 0, a.in1 ? a.in1_bytes : a.in0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);

    const int lim = kt0 + nk;
    int step = kt0;
    int tap = (int)yk_div((uint32_t)kt0 * BK, a.fd_ctp);
    int cin = kt0 * BK - tap * Ctp;
    u32x4 ra[PF][A_IT], rb[PF][B_IT];
    auto gload = [&](u32x4 (&ra)[A_IT], u32x4 (&rbv)[B_IT]) {
        const bool src1 = cin >= a.c0p;
        const int ky = (a.ks == 3) ? (tap * 11) >> 5 : 0, kx = tap - ky * a.ks;       // tap / 3 for tap < 9+
        const bool live = (step < lim) && (tap < taps);
        const uint32_t toff = (uint32_t)((ky * a.Wi + kx) * (src1 ? a.c1p : a.c0p)) * 2u + (uint32_t)cin * 2u;
        const uint32_t soff = live ? toff : YK_OOB;                                     // P < 2^30, so P + YK_OOB is out of range
        const uint32_t ws = live ? (uint32_t)step * (BK * 2u) : YK_OOB;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const uint32_t o = (src1 ? P1[it] : P0[it]) + soff;
            const uint32_t off = ((rmask[it] >> tap) & 1u) ? o : YK_OOB;
            ra[it] = src1 ? __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) rbv[it] = __builtin_amdgcn_raw_buffer_load_b128(rsw, wro[it] + ws, 0, 0);
        ++step;
        cin += BK;
        const bool wrap = cin >= Ctp;
        cin = wrap ? 0 : cin;
        tap += wrap ? 1 : 0;
    };
    auto sstore = [&](const u32x4 (&ra)[A_IT], const u32x4 (&rbv)[B_IT], int stage) {
        yk_half *As = lds + stage * STAGE, *Bs = As + BM * LD;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) *reinterpret_cast<u32x4 *>(As + ((tid + it * NT) / CPR) * LD + kc * 8) = ra[it];
#pragma unroll
        for (int it = 0; it < B_IT; ++it) *reinterpret_cast<u32x4 *>(Bs + ((tid + it * NT) / CPR) * LD + kc * 8) = rbv[it];
    };
    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    auto compute = [&](int stage) {
        const yk_half *As = lds + stage * STAGE, *Bs = As + BM * LD;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            half8 wf[TN], xf[TM];
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const half8 *>(Bs + ((wn * TN + j) * 16 + fr) * LD + ks * 32 + fk);
#pragma unroll
            for (int i = 0; i < TM; ++i) xf[i] = *reinterpret_cast<const half8 *>(As + ((wm * TM + i) * 16 + fr) * LD + ks * 32 + fk);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
    };
    if (nk > 0) {
        gload(ra[0], rb[0]);
        sstore(ra[0], rb[0], 0);
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PF; ++p) gload(ra[p], rb[p]);        // steps kt0+1 .. kt0+PF in flight
        for (int kt = 0; kt < nk; kt += PF) {                    // loads past the end of this split return zeros
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                if (p && kt + p >= nk) break;
                compute(p & 1);
                sstore(ra[p], rb[p], (p + 1) & 1);
                gload(ra[p], rb[p]);
                __syncthreads();
            }
        }
    }
    igemm_epilogue<BM, BN, WM, WN, OUT, TM, TN>(a, acc, lds, m0, n0, tid, lane, wm, wn);
}

#include "yk_igemm_pipe.h"
#ifdef YK_DEV
#include "yk_igemm_lc.h"                                           // loader / consumer waves: measured slower in the whole networks, developer builds only
#endif

int yk_launch_splitk_reduce(const igemm_args &a, bool out_f32, hipStream_t st) {
    const size_t total = (size_t)a.M * (a.ldn >> 2);
    dim3 grid((unsigned)((total + 255) / 256));
    if (out_f32) hipLaunchKernelGGL(splitk_reduce_kernel<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(splitk_reduce_kernel<false>, grid, dim3(256), 0, st, a);
    return YK_OK;
}

template <int BM, int BN, int WM, int WN, int BK, bool F32, bool UNI_OK = false>
static int launch_cfg(const igemm_args &a, hipStream_t st) {
    dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN, a.split_k > 1 ? a.split_k : 1);
    static const bool uni_on = yk_dev_env("YK_UNI") ? yk_dev_env("YK_UNI")[0] != '0' : true;
    // "dead" taps / rows are addressed at P + YK_OOB where P may be (mod 2^32) up to one image row + one pixel BELOW zero (the halo of the
    // first output row): the whole tensor plus that margin must stay under YK_OOB for the sum to be out of the descriptor's range
    const uint64_t margin = (uint64_t)(a.Wi + 2) * (uint64_t)std::max(a.c0p, a.c1p) * 2u + (uint64_t)a.c0p * 2u;
    const bool uni = uni_on && UNI_OK && ((a.c0p + a.c1p) % BK == 0) && (a.c0p % BK == 0) && (uint64_t)a.in0_bytes + margin < YK_OOB &&
                     (uint64_t)a.in1_bytes + margin < YK_OOB;
    constexpr size_t stages = (size_t)2 * (BM + BN) * (BK + YK_LDPAD) * 2, ctile = F32 ? 0 : (size_t)BM * (BN + 8) * 2;
    constexpr size_t lds = stages > ctile ? stages : ctile;
    auto go = [&](auto kern) {
        if (lds > 64 * 1024) {
            static bool done = false;
            if (!done) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                done = true;
            }
        }
        hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, st, a);
    };
    static const bool dma_on = yk_dev_env("YK_DMA") ? yk_dev_env("YK_DMA")[0] != '0' : true;      // LDS-DMA operand tiles (YK_DMA=0: register staging)
    if constexpr (UNI_OK && BK == 64 && !F32 && (BM / 8) % (WM * WN) == 0 && (BN / 8) % (WM * WN) == 0) {
        static const bool pipe_on = yk_dev_env("YK_PIPE") ? yk_dev_env("YK_PIPE")[0] != '0' : true;       // multi-stage LDS-DMA ring (yk_igemm_pipe.h)
        // ring depth: 2 stages wherever the grid fills the CUs several times over (3-4 stages cost occupancy and lose 10-20 % there:
        // 52x52 128->256: 2 stages 551, 3 stages 457 TF/s); a grid of at most ~2 workgroups per CU is latency-bound instead - each
        // k-step waits for the DMA issued one step earlier - and takes 4 stages (three k-steps of loads in flight) where two such
        // workgroups still fit a CU's LDS: the 64x64 tile (7x10 768->192 3x3: 26.3 -> 20.3 us; with the 64x128 tile, one workgroup
        // per CU, the upsample+concat conv went 27 -> 38 us)
        static const int ns_env = yk_dev_env("YK_NS") ? atoi(yk_dev_env("YK_NS")) : 0;                    // dev sweep
        const long wgs = (long)grid.x * grid.y * grid.z;
        const int ns = ns_env ? ns_env : (wgs <= 640 && (BM + BN) <= 128 ? 4 : 2);
        if constexpr (BM + BN <= 128)
            if (dma_on && pipe_on && uni && ns == 4) return launch_pipe<BM, BN, WM, WN, 4>(a, st);
#ifdef YK_DEV
        if (dma_on && pipe_on && uni && ns == 3) return launch_pipe<BM, BN, WM, WN, 3>(a, st);
#endif
        if (dma_on && pipe_on && uni) return launch_pipe<BM, BN, WM, WN, 2>(a, st);
    }
    static const bool lin_on = yk_dev_env("YK_LIN") ? yk_dev_env("YK_LIN")[0] != '0' : true;
    constexpr bool LIN_OK = UNI_OK && ((BM * (BK / 8)) % (64 * WM * WN) == 0) && ((BN * (BK / 8)) % (64 * WM * WN) == 0);
    if constexpr (LIN_OK) {
        if (uni && lin_on && !a.up0) {
            // prefetch depth: 4 k-steps in flight where a step's staging registers are cheap (<= 16 VGPRs)
            constexpr int PF = ((BM + BN) * (BK / 8) / (64 * WM * WN) <= 4) ? 4 : 2;
            static const int pf_env = yk_dev_env("YK_PF") ? atoi(yk_dev_env("YK_PF")) : 0;
            if (PF == 4 && pf_env != 2) {
                if (a.split_k > 1) go(igemm_lin_kernel<BM, BN, WM, WN, BK, 2, PF>);
                else go(igemm_lin_kernel<BM, BN, WM, WN, BK, F32 ? 1 : 0, PF>);
            } else {
                if (a.split_k > 1) go(igemm_lin_kernel<BM, BN, WM, WN, BK, 2, 2>);
                else go(igemm_lin_kernel<BM, BN, WM, WN, BK, F32 ? 1 : 0, 2>);
            }
            return YK_OK;
        }
    }
    if (uni) {
        if (a.split_k > 1) go(igemm_kernel<BM, BN, WM, WN, BK, 2, UNI_OK>);
        else go(igemm_kernel<BM, BN, WM, WN, BK, F32 ? 1 : 0, UNI_OK>);
    } else {
        if (a.split_k > 1) go(igemm_kernel<BM, BN, WM, WN, BK, 2, false>);
        else go(igemm_kernel<BM, BN, WM, WN, BK, F32 ? 1 : 0, false>);
    }
    return YK_OK;
}

struct igemm_cfg_info {
    int bm, bn, bk;
    const char *name;
};
static const igemm_cfg_info g_cfg[IGEMM_NUM] = {
    {128, 64, 32, "igemm_128x64"}, {128, 48, 32, "igemm_128x48"}, {128, 96, 32, "igemm_128x96"},
    {128, 192, 64, "igemm_128x192k64"}, {64, 64, 64, "igemm_64x64k64"}, {128, 128, 64, "igemm_128x128k64"},
    {64, 80, 64, "igemm_f32_64x80k64"}, {128, 64, 32, "igemm_f32_128x64"}, {128, 64, 64, "igemm_128x64k64"},
    {64, 128, 64, "igemm_64x128k64"}, {64, 192, 64, "igemm_64x192k64"}, {256, 128, 64, "igemm_256x128k64"},
    {256, 128, 64, "igemm_256x128k64w4"}, {128, 256, 64, "igemm_128x256k64"}, {128, 128, 64, "igemm_128x128k64r"}, {256, 256, 64, "igemm_256x256k64"},
    {256, 128, 64, "igemm_lc_256x128"}, {128, 128, 64, "igemm_lc_128x128"}, {128, 256, 64, "igemm_lc_128x256"}};

static int yk_big_ring_depth(int cfg) { return cfg == IGEMM_256x128 ? 3 : 2; }

int yk_launch_igemm(int cfg, const igemm_args &a, hipStream_t st) {
    switch (cfg) {
    case IGEMM_128x64: return launch_cfg<128, 64, 2, 2, 32, false, true>(a, st);
    case IGEMM_128x48: return launch_cfg<128, 48, 4, 1, 32, false>(a, st);
    case IGEMM_128x96: return launch_cfg<128, 96, 4, 1, 32, false>(a, st);
    case IGEMM_128x192: return launch_cfg<128, 192, 2, 2, 64, false, true>(a, st);
    case IGEMM_64x64: return launch_cfg<64, 64, 2, 2, 64, false, true>(a, st);
    case IGEMM_128x128: return launch_cfg<128, 128, 2, 2, 64, false, true>(a, st);
    case IGEMM_F32_64x80: return launch_cfg<64, 80, 4, 1, 64, true, true>(a, st);
    case IGEMM_F32_128x64: return launch_cfg<128, 64, 2, 2, 32, true>(a, st);
    case IGEMM_128x64K64: return launch_cfg<128, 64, 2, 2, 64, false, true>(a, st);
    case IGEMM_64x128: return launch_cfg<64, 128, 2, 2, 64, false, true>(a, st);
    case IGEMM_64x192: return launch_cfg<64, 192, 2, 2, 64, false, true>(a, st);
    // large ring tiles (MFMA-bound layers at large M: Darknet-53 / tiny-YOLO 3x3 convs); LDS-DMA preconditions only, no register-staged fallback
    case IGEMM_256x128:
    case IGEMM_256x128W4:
    case IGEMM_128x256:
    case IGEMM_128x128R:
    case IGEMM_256x256:
    case IGEMM_LC_256x128:
    case IGEMM_LC_128x128:
    case IGEMM_LC_128x256: {
        const uint64_t margin = (uint64_t)(a.Wi + 2) * (uint64_t)std::max(a.c0p, a.c1p) * 2u + (uint64_t)a.c0p * 2u;
        if ((a.c0p + a.c1p) % 64 || a.c0p % 64 || (uint64_t)a.in0_bytes + margin >= YK_OOB || (uint64_t)a.in1_bytes + margin >= YK_OOB) break;
        const int ns_env = yk_dev_env("YK_NS") ? atoi(yk_dev_env("YK_NS")) : 0;     // (developer build: read per launch, tools/r05_igemm_sweep.py)
        const int ns = ns_env ? ns_env : yk_big_ring_depth(cfg);
        if (cfg == IGEMM_256x128) return ns >= 3 ? launch_pipe<256, 128, 4, 2, 3>(a, st) : launch_pipe<256, 128, 4, 2, 2>(a, st);
        if (cfg == IGEMM_256x128W4) return ns >= 3 ? launch_pipe<256, 128, 2, 2, 3>(a, st) : launch_pipe<256, 128, 2, 2, 2>(a, st);
        if (cfg >= IGEMM_LC_256x128) {                             // loader / consumer form (yk_igemm_lc.h, developer builds): no upsampled source
#ifdef YK_DEV
            if (a.up0) break;
            if (cfg == IGEMM_LC_256x128) return launch_lc<256, 128, 2, 2, 4, 3>(a, st);
            if (cfg == IGEMM_LC_128x256) return launch_lc<128, 256, 2, 2, 4, 3>(a, st);
            return ns >= 4 ? launch_lc<128, 128, 2, 2, 4, 4>(a, st) : launch_lc<128, 128, 2, 2, 4, 3>(a, st);
#else
            break;
#endif
        }
        if (cfg == IGEMM_256x256) return launch_pipe<256, 256, 4, 2, 2>(a, st);
        if (cfg == IGEMM_128x256) return ns >= 3 ? launch_pipe<128, 256, 2, 2, 3>(a, st) : launch_pipe<128, 256, 2, 2, 2>(a, st);
        return ns >= 4 ? launch_pipe<128, 128, 2, 2, 4>(a, st) : (ns == 3 ? launch_pipe<128, 128, 2, 2, 3>(a, st) : launch_pipe<128, 128, 2, 2, 2>(a, st));
    }
    }
    yk_set_error("yk_launch_igemm: bad config %d", cfg);
    return YK_ERR_ARG;
}

int yk_fused_pad() {
    static const int p = yk_dev_env("YK_FPAD") ? atoi(yk_dev_env("YK_FPAD")) : 16;
    return (p == 8 || p == 16 || p == 24) ? p : 16;
}

const char *yk_igemm_name(int cfg) { return (cfg >= 0 && cfg < IGEMM_NUM) ? g_cfg[cfg].name : "?"; }

int yk_igemm_pick(const igemm_args &a, bool out_f32) {
    if (out_f32) return a.N <= 80 ? IGEMM_F32_64x80 : IGEMM_F32_128x64;
    if (a.K >= (yk_dev_env("YK_FORCE_MINK") ? atoi(yk_dev_env("YK_FORCE_MINK")) : 1024)) {   // tuning sweep hook (tools/igemm_sweep.py)
        const char *f = yk_dev_env("YK_IGEMM_FORCE");
        if (f && f[0]) {
            const int c = atoi(f);
            const bool ring_only = c >= IGEMM_256x128;                 // no register-staged fallback: only where the LDS-DMA ring applies
            const bool ring_ok = ((a.c0p + a.c1p) % 64 == 0) && (a.c0p % 64 == 0);
            if (c >= 0 && c < IGEMM_NUM && c != IGEMM_F32_64x80 && c != IGEMM_F32_128x64 && a.N % g_cfg[c].bn == 0 && (!ring_only || ring_ok))
                return c;
        }
    }
    const long mt128 = (a.M + 127) / 128;
    if (a.N == 48) return IGEMM_128x48;
    if (a.N == 96) return IGEMM_128x96;
    if (a.N == 192 && mt128 >= 512) return IGEMM_128x192;
    // Measured on Darknet-53 / tiny-YOLO shapes at B=16 (tools/igemm_shapes.py, TF/s): every config sits between 300 and 470 -
    // no unit is saturated (MFMA busy 17 %, LDS 39 %, PMC run in profiles/), a workgroup's k-step is a dependent chain
    // (loads -> LDS -> barrier -> fragments -> MFMA), so MORE, SMALLER workgroups win: 64x64x64 beats 128x64x32 by 10-15 % and
    // 128x128x64 (1 wave/SIMD) is the slowest everywhere (104x104 64->128: 294 vs 416).  Long reductions therefore take the
    // small tile; short ones keep 128x64, whose per-workgroup fixed cost is amortised over more output.
    // with LDS-DMA tiles the 64x128x64 config is the fastest wherever it applies (52x52 128->256: 519 vs 460 TFLOP/s for 64x64x64,
    // 13x13 512->1024: 480 vs 447); the register-staged fallbacks (upsampled input, channel pitch not a multiple of 64) keep 64x64
    const bool dma_ok = ((a.c0p + a.c1p) % 64 == 0) && (a.c0p % 64 == 0);     // upsampled sources included (yk_igemm_pipe.h)
    // round 5, tools/r05_igemm_sweep.py at 32 / 64 images: from ~150 k GEMM rows on the 128x128 ring tile (two stages, two workgroups per CU)
    // is 5-8 % ahead on the 3x3 layers (52x52 128->256 at 64 images: 718 vs 663 TFLOP/s) and 17-22 % on the wide 1x1 layers
    // (52x52 256->128: 308 vs 263); below that the two tie and the smaller tile fills the chip better
    {
        const uint64_t margin = (uint64_t)(a.Wi + 2) * (uint64_t)std::max(a.c0p, a.c1p) * 2u + (uint64_t)a.c0p * 2u;
        const bool ring_fits = (uint64_t)a.in0_bytes + margin < YK_OOB && (uint64_t)a.in1_bytes + margin < YK_OOB;   // (the ring kernel has no fallback form)
        // loader + consumer waves (yk_igemm_lc.h), 128x256 tile.  Alone, on a layer without residual and without split-K, it is ahead on the
        // long-K wide-N 3x3 layers (tools/r05_igemm_sweep.py, TFLOP/s at 32 / 64 images: 26x26 256->512 686 / 878 vs 658 / 745, 13x13 512->1024
        // 780 / 800 vs 665 / 708); picked for the whole of Darknet-53 it LOSES (6644 vs 7552 images/s at 32 images, 8328 vs 8708 at 64: its
        // 128x256 tiles need split-K slabs + a finishing pass where the 64x128 tile fills the chip without) - developer builds only, YK_IGEMM_LC=1
        if (yk_dev_env("YK_IGEMM_LC") && yk_dev_env("YK_IGEMM_LC")[0] == '1' && dma_ok && ring_fits && !a.up0 && !a.in1 && a.N % 256 == 0 && a.K >= 1024 && a.M >= 4096) return IGEMM_LC_128x256;
        if (dma_ok && ring_fits && !a.up0 && a.N % 128 == 0 && a.M >= 150000 && a.K >= 256) return IGEMM_128x128R;
    }
    if (a.K >= 512) return (a.N % 128 == 0 && dma_ok) ? IGEMM_64x128 : IGEMM_64x64;
    if (mt128 * ((a.N + 63) / 64) >= 256) return IGEMM_128x64;
    return IGEMM_64x64;
}

// split K when the (M,N) tiling alone cannot fill 256 CUs and K is long enough to share out
int yk_igemm_split(int cfg, const igemm_args &a) {
    const igemm_cfg_info &c = g_cfg[cfg];
    const long tiles = (long)((a.M + c.bm - 1) / c.bm) * ((a.N + c.bn - 1) / c.bn);
    const int nk = (a.K + c.bk - 1) / c.bk;
    if (a.K >= 1024) {
        const char *f = yk_dev_env("YK_SPLIT_FORCE");
        if (f && f[0]) return std::max(1, std::min(atoi(f), nk));
    }
    if (tiles >= 384 || nk < 6) return 1;
    const long target = ((long)c.bm * c.bn >= 128 * 128) ? 512 : 1024;
    long s = (target + tiles - 1) / tiles;
    if (s > nk / 3) s = nk / 3;
    // the slabs cost HBM traffic twice (written here, read by the finishing pass): measured on the 7x10 / 14x20 head convs the conv
    // itself takes the same time for 2..10 splits (it is bound by operand traffic through L2, not by parallelism), the finishing
    // pass grows from 10 to 16 us.  Four splits are enough to have ~3 workgroups per CU.
    if (s > 4) s = 4;
    return s < 2 ? 1 : (int)s;
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
    if (!a.in_f32) {                                                 // img / np.max(img), tools/utils.py:405
        unsigned mx = 0;
#pragma unroll
        for (int j = 0; j < YK_MAXP; ++j) mx = max(mx, a.img_max[b * YK_MAXP + j]);
        lut[tid] = (float)(yk_half)((float)tid / (float)mx);   // the normalised image is an fp16 tensor like every other activation
    }
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
                x[0] = (float)(yk_half)p[0];
                x[1] = (float)(yk_half)p[1];
                x[2] = (float)(yk_half)p[2];
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
        for (int j = 0; j < 8; ++j) h[j] = (yk_half)yk_actf(acc[c8 + j] * sc[c8 + j] + bs[c8 + j], a.slope, a.cap);
        *reinterpret_cast<half8 *>(o + c8) = h;
    }
}

// Stem as MFMA (u8 frames).  K = 27 is padded to 32 in an order chosen for the loader, not the math: k = ky*8 + j for the first
// 8 of the 9 contiguous bytes (3 pixels x RGB) a tap row contributes, k = 24 + ky for the ninth, 27..31 zero (`wm` holds the
// weights in that order).  A lane of the pixel operand (pixel = lane&15, chunk q = lane>>4) therefore needs ONE unaligned 8-byte
// load (q < 3: row q) or three byte loads (q = 3); left padding falls out of a 64-bit shift, right padding / row overrun out of
// a shift right, invalid rows out of the buffer bounds check.  Bytes become fp16 through a 256-entry LUT of half(v / max) in LDS.
// 16 pixels x 32 channels per two MFMAs; the VALU version spends ~985 instructions per wave of 64 pixels on the same work.
__global__ void __launch_bounds__(256) stem_mfma_kernel(const first_args a) {
    __shared__ yk_half lut[256];
    __shared__ unsigned smx;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int P = a.Ho * a.Wo, wg_per_img = P >> 8;
    const int b = blockIdx.x / wg_per_img, pix0 = (blockIdx.x - b * wg_per_img) << 8;
    if (tid < 64) {
        unsigned m = tid < YK_MAXP ? a.img_max[b * YK_MAXP + tid] : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
        if (tid == 0) smx = m;
    }
    __syncthreads();
    lut[tid] = (yk_half)((float)tid / (float)smx);
    const int fr = lane & 15, q = lane >> 4, nl4 = q * 4;
    half8 wf[2];
    float4 sc[2], bs[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        wf[j] = *reinterpret_cast<const half8 *>(a.wm + (j * 16 + fr) * 32 + q * 8);
        sc[j] = *reinterpret_cast<const float4 *>(a.scale + j * 16 + nl4);
        bs[j] = *reinterpret_cast<const float4 *>(a.bias + j * 16 + nl4);
    }
    const uint32_t img_bytes = (uint32_t)a.Hi * a.Wi * 3u, rowb = (uint32_t)a.Wi * 3u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)((const uint8_t *)a.in + (size_t)b * img_bytes), 0, img_bytes, 0x00020000);
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 d8[4];
    uint32_t b9[4][3], shl[4], nval[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int p = pix0 + (wid * 4 + t) * 16 + fr;
        const int oy = p / a.Wo, ox = p - oy * a.Wo;
        const int iy0 = oy * a.stride - a.pad_t, s = (ox * a.stride - a.pad_l) * 3;      // s: first byte of the 9 in a tap row
        // the 8-byte load is clamped into the row: left of it (padding) a shift left brings zeros in, at the row's end a shift
        // right drops the bytes that would belong to the next row - and the load never runs past the image (bounds check)
        const int ld = min(max(s, 0), (int)rowb - 8);
        shl[t] = (uint32_t)((ld - s) * 8);                                                 // > 0: shift left, < 0 (as int): shift right
        nval[t] = (uint32_t)max(min((int)rowb - s, 9), 0);                                // bytes of the 9 that lie inside the row
        const int iy = iy0 + q;
        const bool rowok = q < 3 && (unsigned)iy < (unsigned)a.Hi;
        d8[t] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, rowok ? (uint32_t)iy * rowb + (uint32_t)ld : YK_OOB, 0, 0));
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iyr = iy0 + r;
            const bool ok = q == 3 && (unsigned)iyr < (unsigned)a.Hi && s + 8 >= 0 && nval[t] == 9u;
            b9[t][r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rs, ok ? (uint32_t)iyr * rowb + (uint32_t)(s + 8) : YK_OOB, 0, 0);
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        unsigned long long v = ((unsigned long long)d8[t][1] << 32) | d8[t][0];
        const int sh = (int)shl[t];
        v = sh >= 0 ? (v << sh) : (v >> (-sh));                                            // zero padding on either side
        if (q == 3) v = (unsigned long long)(b9[t][0] | (b9[t][1] << 8) | (b9[t][2] << 16));
        half8 xf;
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[j] = lut[(unsigned)(v >> (8 * j)) & 255u];
        if (q == 3) {                                                                      // k = 27..31 are padding, not LUT[0]
#pragma unroll
            for (int j = 3; j < 8; ++j) xf[j] = (yk_half)0.f;
        }
        const int p = pix0 + (wid * 4 + t) * 16 + fr;
        yk_half *o = a.out + ((size_t)b * P + p) * a.outp;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf, acc, 0, 0, 0);
            const int n = j * 16 + nl4;
            if (n < a.outp) {
                const half4 h = {(yk_half)yk_actf(acc[0] * sc[j].x + bs[j].x, a.slope, a.cap), (yk_half)yk_actf(acc[1] * sc[j].y + bs[j].y, a.slope, a.cap),
                                 (yk_half)yk_actf(acc[2] * sc[j].z + bs[j].z, a.slope, a.cap), (yk_half)yk_actf(acc[3] * sc[j].w + bs[j].w, a.slope, a.cap)};
                *reinterpret_cast<half4 *>(o + n) = h;
            }
        }
    }
}

int yk_launch_first(const first_args &a, hipStream_t st) {
    static const bool mfma_on = yk_dev_env("YK_STEM_MFMA") ? yk_dev_env("YK_STEM_MFMA")[0] != '0' : true;
    if (mfma_on && !a.in_f32 && a.wm && a.Cout <= 32 && a.outp % 4 == 0 && (a.Ho * a.Wo) % 256 == 0 &&
        (size_t)a.Hi * a.Wi * 3 < YK_OOB) {
        hipLaunchKernelGGL(stem_mfma_kernel, dim3((unsigned)(a.B * ((a.Ho * a.Wo) >> 8))), dim3(256), 0, st, a);
        return YK_OK;
    }
    dim3 grid((a.Ho * a.Wo + 255) / 256, a.B);
    switch (a.Cout) {
    case 16: hipLaunchKernelGGL(first_conv_kernel<16>, grid, dim3(256), 0, st, a); break;
    case 24: hipLaunchKernelGGL(first_conv_kernel<24>, grid, dim3(256), 0, st, a); break;
    case 32: hipLaunchKernelGGL(first_conv_kernel<32>, grid, dim3(256), 0, st, a); break;
    default: yk_set_error("stem conv: Cout=%d unsupported (16/24/32)", a.Cout); return YK_ERR_UNSUPPORTED;
    }
    return YK_OK;
}

// per-image max of u8 frames: YK_MAXP workgroups per image write partial maxima img_max[b*YK_MAXP + j]
// (plain stores, no atomics, no memset); the stem conv folds the partials when it builds its LUT.
// `zero` (may be null): a word array this first launch of a step also clears - the f16x2 plan's per-image running maxima, which every
// later launch of the step accumulates into (a separate hipMemsetAsync was a 4 us fill launch per step).
__global__ void __launch_bounds__(256) u8_max_kernel(const uint8_t *__restrict__ f, size_t per_image, int vec_ok,
                                                     unsigned *__restrict__ img_max, uint32_t *__restrict__ zero, size_t zero_words) {
    __shared__ unsigned part[4];
    const int b = blockIdx.y, j = blockIdx.x, tid = threadIdx.x;
    if (zero) {
        const size_t nthr = (size_t)gridDim.x * gridDim.y * 256;
        for (size_t i = ((size_t)b * gridDim.x + j) * 256 + tid; i < zero_words; i += nthr) zero[i] = 0u;
    }
    const uint8_t *p = f + (size_t)b * per_image;
    unsigned m = 0;
    const size_t t = (size_t)j * 256 + tid, nth = (size_t)gridDim.x * 256;
    if (vec_ok) {
        const uint4 *q = reinterpret_cast<const uint4 *>(p);
        const size_t n16 = per_image / 16;
        size_t i = t;
        for (; i + 3 * nth < n16; i += 4 * nth) {          // 4 independent 16-byte loads in flight
            const uint4 v[4] = {q[i], q[i + nth], q[i + 2 * nth], q[i + 3 * nth]};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    m = max(m, max(max(w[k] & 0xffu, (w[k] >> 8) & 0xffu), max((w[k] >> 16) & 0xffu, w[k] >> 24)));
            }
        }
        for (; i < n16; i += nth) {
            const uint4 v = q[i];
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                m = max(m, max(max(w[k] & 0xffu, (w[k] >> 8) & 0xffu), max((w[k] >> 16) & 0xffu, w[k] >> 24)));
        }
        for (size_t r = n16 * 16 + t; r < per_image; r += nth) m = max(m, (unsigned)p[r]);
    } else {
        for (size_t i = t; i < per_image; i += nth) m = max(m, (unsigned)p[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((tid & 63) == 0) part[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) img_max[b * YK_MAXP + j] = max(max(part[0], part[1]), max(part[2], part[3]));
}

int yk_launch_u8_max(const uint8_t *frames, size_t per_image, int batch, unsigned *img_max, hipStream_t st, uint32_t *zero, size_t zero_words) {
    const int vec_ok = (per_image % 16 == 0) && ((uintptr_t)frames % 16 == 0);
    static const int parts = yk_dev_env("YK_MAXP_RT") ? std::max(1, std::min(YK_MAXP, atoi(yk_dev_env("YK_MAXP_RT")))) : YK_MAXP;   // unused slots stay 0
    hipLaunchKernelGGL(u8_max_kernel, dim3(parts, batch), dim3(256), 0, st, frames, per_image, vec_ok, img_max, zero, zero_words);
    return YK_OK;
}

// =====================================================================================
// depthwise 3x3 (standalone)
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
            const u32x4 x = *reinterpret_cast<const u32x4 *>(a.in + ((size_t)(b * a.Hi + iy) * a.Wi + ix) * a.Cp + g * 8);
            const u32x4 w = *reinterpret_cast<const u32x4 *>(a.w + (size_t)(ky * 3 + kx) * a.Cp + g * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                fma_mix_lo(acc[2 * j], x[j], w[j]);
                fma_mix_hi(acc[2 * j + 1], x[j], w[j]);
            }
        }
    }
    half8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (yk_half)yk_actf(acc[j] * a.scale[g * 8 + j] + a.bias[g * 8 + j], a.slope, a.cap);
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
//   phase A: depthwise + BN + act for the BM x Cin tile: v_fma_mix_f32 (fp16 operands, fp32
//            accumulate, no conversions), two pixels per thread in flight, result rounded to fp16
//            into LDS (= what the unfused pipeline would have written to HBM);
//   phase B: [BN x K] (weights, streamed L2 -> registers, WPF k-steps ahead, first ones issued
//            BEFORE phase A so they overlap it) x [K x BM] (LDS) on v_mfma_f32_16x16x32_f16;
//            epilogue as igemm_kernel.
// =====================================================================================
template <int BM, int BN, int WM, int WN, int IT>
__global__ void __launch_bounds__(64 * WM * WN) fused_dwpw_kernel(const igemm_args a) {
    constexpr int NT = 64 * WM * WN;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int CS_LD = BN + 8;
    constexpr int WPF = 2;                           // weight prefetch depth (k-steps of 32)
    yk_half *As = reinterpret_cast<yk_half *>(yk_smem);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int Cp = a.c0p, Kp = (Cp + 31) & ~31, LDA = Kp + a.lda_pad;
    const int m0 = yk_xcd_tile(blockIdx.x, gridDim.x) * BM, n0 = blockIdx.y * BN;
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    const int nk = Kp >> 5;
#define YK_STAMP(k)                                                                                  \
    if (a.dbg && tid == 0) a.dbg[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = (long long)wall_clock64();
    YK_STAMP(0)

    // weight fragments: issue the first WPF k-steps now, they land while phase A runs
    auto wload = [&](half8 (&wf)[TN], int k0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 16 + fr, k = k0 + fk;
            half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (n < a.N && k < a.K) v = *reinterpret_cast<const half8 *>(a.w + (size_t)n * a.K + k);
            wf[j] = v;
        }
    };
    half8 wq[WPF][TN];
#pragma unroll
    for (int s = 0; s < WPF; ++s)
        if (s < nk) wload(wq[s], s * 32);

    YK_STAMP(1)
    // ---------------- phase A: depthwise producer ----------------
    {
        const int G = Cp >> 3;
        const int PP = NT / G;                       // pixels per pass
        const int g = tid % G, pl = tid / G;
        if (pl < PP) {
            u32x4 w[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) w[t] = *reinterpret_cast<const u32x4 *>(a.dw_w + (size_t)t * Cp + g * 8);
            const float4 sc0 = *reinterpret_cast<const float4 *>(a.dw_scale + g * 8);
            const float4 sc1 = *reinterpret_cast<const float4 *>(a.dw_scale + g * 8 + 4);
            const float4 bs0 = *reinterpret_cast<const float4 *>(a.dw_bias + g * 8);
            const float4 bs1 = *reinterpret_cast<const float4 *>(a.dw_bias + g * 8 + 4);
            const int hw = a.Ho * a.Wo;
            const yk_half *inb = a.in0 + g * 8;
            for (int p = pl; p < BM; p += IT * PP) {
                // IT output pixels per iteration: 9*IT independent 16-byte loads in flight
                u32x4 x[IT][9];
                bool live[IT];
#pragma unroll
                for (int u = 0; u < IT; ++u) {
                    const int pp = p + u * PP, m = m0 + pp;
                    live[u] = (pp < BM);
                    const bool valid = live[u] && m < a.M;
                    const uint32_t b = yk_div(valid ? m : 0, a.fd_hw), rem = (valid ? m : 0) - b * hw;
                    const uint32_t oy = yk_div(rem, a.fd_wo), ox = rem - oy * a.Wo;
                    const int iy0 = (int)oy * a.dw_stride - a.dw_pad_t, ix0 = (int)ox * a.dw_stride - a.dw_pad_l;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int iy = iy0 + ky, ix = ix0 + kx;
                            u32x4 v = {0, 0, 0, 0};
                            if (valid && (unsigned)iy < (unsigned)a.dw_Hi && (unsigned)ix < (unsigned)a.dw_Wi)
                                v = *reinterpret_cast<const u32x4 *>(inb + ((size_t)(b * a.dw_Hi + iy) * a.dw_Wi + ix) * Cp);
                            x[u][ky * 3 + kx] = v;
                        }
                }
#pragma unroll
                for (int u = 0; u < IT; ++u) {
                    if (!live[u]) continue;
                    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int t = 0; t < 9; ++t)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            fma_mix_lo(acc[2 * j], x[u][t][j], w[t][j]);
                            fma_mix_hi(acc[2 * j + 1], x[u][t][j], w[t][j]);
                        }
                    half8 h;
                    h[0] = (yk_half)yk_actf(acc[0] * sc0.x + bs0.x, a.dw_slope, a.dw_cap);
                    h[1] = (yk_half)yk_actf(acc[1] * sc0.y + bs0.y, a.dw_slope, a.dw_cap);
                    h[2] = (yk_half)yk_actf(acc[2] * sc0.z + bs0.z, a.dw_slope, a.dw_cap);
                    h[3] = (yk_half)yk_actf(acc[3] * sc0.w + bs0.w, a.dw_slope, a.dw_cap);
                    h[4] = (yk_half)yk_actf(acc[4] * sc1.x + bs1.x, a.dw_slope, a.dw_cap);
                    h[5] = (yk_half)yk_actf(acc[5] * sc1.y + bs1.y, a.dw_slope, a.dw_cap);
                    h[6] = (yk_half)yk_actf(acc[6] * sc1.z + bs1.z, a.dw_slope, a.dw_cap);
                    h[7] = (yk_half)yk_actf(acc[7] * sc1.w + bs1.w, a.dw_slope, a.dw_cap);
                    *reinterpret_cast<half8 *>(As + (p + u * PP) * LDA + g * 8) = h;
                }
            }
        }
        // zero the K padding (Cp..Kp) once
        const int padv = (Kp - Cp) >> 3;
        for (int v = tid; v < BM * padv; v += NT) {
            const int p = v / padv, c = v - p * padv;
            *reinterpret_cast<half8 *>(As + p * LDA + Cp + c * 8) = half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    YK_STAMP(2)
    __syncthreads();
    YK_STAMP(3)

    // ---------------- phase B: GEMM, weights streamed from L2 ----------------
    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; kt += WPF) {
#pragma unroll
        for (int s = 0; s < WPF; ++s) {
            if (kt + s < nk) {
                half8 xf[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    xf[i] = *reinterpret_cast<const half8 *>(As + ((wm * TM + i) * 16 + fr) * LDA + (kt + s) * 32 + fk);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[s][j], xf[i], acc[i][j], 0, 0, 0);
                if (kt + s + WPF < nk) wload(wq[s], (kt + s + WPF) * 32);
            }
        }
    }
    YK_STAMP(4)
    __syncthreads();   // everyone is done with the A tile; reuse LDS for the output tile
    YK_STAMP(5)

    yk_half *Cs = reinterpret_cast<yk_half *>(yk_smem);
    const int nl4 = (lane >> 4) * 4;
    float4 sc[TN], bs[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 16 + nl4;
        sc[j] = *reinterpret_cast<const float4 *>(a.scale + n);
        bs[j] = *reinterpret_cast<const float4 *>(a.bias + n);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = (wm * TM + i) * 16 + fr, m = m0 + ml;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = (wn * TN + j) * 16 + nl4, n = n0 + nl;
            half4 h = {(yk_half)yk_actf(acc[i][j][0] * sc[j].x + bs[j].x, a.slope, a.cap),
                       (yk_half)yk_actf(acc[i][j][1] * sc[j].y + bs[j].y, a.slope, a.cap),
                       (yk_half)yk_actf(acc[i][j][2] * sc[j].z + bs[j].z, a.slope, a.cap),
                       (yk_half)yk_actf(acc[i][j][3] * sc[j].w + bs[j].w, a.slope, a.cap)};
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
    YK_STAMP(6)
    if (a.dbg && tid == 0)
        a.dbg[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + 7] =
            ((long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);
#undef YK_STAMP
}


// -------------------------------------------------------------------------------------
// "wide" variant: 12 wavefronts (768 threads) per workgroup.  Built for memory-level parallelism at
// batch 32, where every MobileNet layer is one or two workgroup rounds and a workgroup's life is a
// chain of memory round trips:
//   * wave (wm, wn) owns 16 output channels x (16*TM) pixels and PRELOADS its whole weight panel
//     (16 x K, up to WPF k-steps of 32) into registers before anything else — no weight load sits
//     on the dependent path; deeper K re-fills the ring right after each use;
//   * the depthwise tile (BM x Cin) is produced by all 768 threads with one or two pixels per thread
//     and all 9/18 tap loads in flight at once; depthwise weights are read from LDS (broadcast-free,
//     consecutive lanes -> consecutive 16-byte slots) instead of living in 36 VGPRs;
//   * phase B touches only LDS + MFMA.
// Requires 768 % (Cin_pitch/8) == 0.
// -------------------------------------------------------------------------------------
#ifndef YK_WIDE_ONE
#define YK_WIDE_ONE 1   /* one depthwise item in flight per thread: 168 -> 116/140 VGPRs.  Measured: +5 % alone (86.7 vs 82 k images/s)
                           and +7 % with three batches in flight (121 vs 113 k) - the freed registers let other waves co-reside */
#endif
template <int WM, int WN, int TM, int WPF>
__global__ void __launch_bounds__(768) fused_wide_kernel(const igemm_args a) {
    constexpr int NT = 768;
    constexpr int BM = WM * 16 * TM, BN = WN * 16;
    constexpr int CS_LD = BN + 8;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int Cp = a.c0p, Kp = (Cp + 31) & ~31, LDA = Kp + a.lda_pad;
    yk_half *As = reinterpret_cast<yk_half *>(yk_smem);
    yk_half *Ws = As + (size_t)BM * LDA;             // depthwise weights [9][Cp]
    const int m0 = yk_xcd_tile(blockIdx.x, gridDim.x) * BM, n0 = blockIdx.y * BN;
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    const int nk = Kp >> 5;
#define YK_STAMP(k)                                                                                  \
    if (a.dbg && tid == 0) a.dbg[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = (long long)wall_clock64();
    YK_STAMP(0)

    // (1) this wave's pointwise weight panel -> registers
    const int nrow = n0 + wn * 16 + fr;
    const yk_half *wrow = a.w + (size_t)nrow * a.K + fk;
    half8 wq[WPF];
#pragma unroll
    for (int s = 0; s < WPF; ++s) {
        half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (s < nk && nrow < a.N && s * 32 + fk < a.K) v = *reinterpret_cast<const half8 *>(wrow + s * 32);
        wq[s] = v;
    }
    // (2) depthwise weights -> LDS
    for (int v = tid; v < 9 * (Cp >> 3); v += NT)
        *reinterpret_cast<half8 *>(Ws + v * 8) = *reinterpret_cast<const half8 *>(a.dw_w + (size_t)v * 8);

    // (3) depthwise tap loads: item j of this thread is pixel pl + j*PP, channel group g
    const int G = Cp >> 3;
    const int pl = (int)yk_div(tid, a.fd_g), g = tid - pl * G;
    const int PP = NT / G;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.in0, 0, a.in0_bytes, 0x00020000);
    const float4 sc0 = *reinterpret_cast<const float4 *>(a.dw_scale + g * 8);
    const float4 sc1 = *reinterpret_cast<const float4 *>(a.dw_scale + g * 8 + 4);
    const float4 bs0 = *reinterpret_cast<const float4 *>(a.dw_bias + g * 8);
    const float4 bs1 = *reinterpret_cast<const float4 *>(a.dw_bias + g * 8 + 4);
    const bool dcap = a.dw_cap < 3.0e38f;
    // zero the K padding (Cp..Kp)
    {
        const int padv = (Kp - Cp) >> 3;
        for (int v = tid; v < BM * padv; v += NT) {
            const int p = v / padv, c = v - p * padv;
            *reinterpret_cast<half8 *>(As + p * LDA + Cp + c * 8) = half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    __syncthreads();                                 // Ws visible
    YK_STAMP(1)
    for (int p = pl; p < BM; p += (YK_WIDE_ONE ? 1 : 2) * PP) {
        u32x4 x0[9], x1[9];
        const bool two = !YK_WIDE_ONE && (p + PP) < BM;
        dw_issue(a, rs, (uint32_t)min(m0 + p, a.M - 1), (uint32_t)g * 16u, x0);
        if (two) dw_issue(a, rs, (uint32_t)min(m0 + p + PP, a.M - 1), (uint32_t)g * 16u, x1);
        const yk_half *wl = Ws + g * 8;
        *reinterpret_cast<half8 *>(As + p * LDA + g * 8) =
            dcap ? dw_finish<true>(x0, wl, Cp, sc0, sc1, bs0, bs1, a.dw_slope, a.dw_cap)
                 : dw_finish<false>(x0, wl, Cp, sc0, sc1, bs0, bs1, a.dw_slope, a.dw_cap);
        if (two)
            *reinterpret_cast<half8 *>(As + (p + PP) * LDA + g * 8) =
                dcap ? dw_finish<true>(x1, wl, Cp, sc0, sc1, bs0, bs1, a.dw_slope, a.dw_cap)
                     : dw_finish<false>(x1, wl, Cp, sc0, sc1, bs0, bs1, a.dw_slope, a.dw_cap);
    }
    YK_STAMP(2)
    __syncthreads();
    YK_STAMP(3)

    // (4) GEMM from LDS + registers
    floatx4 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; kt += WPF) {
#pragma unroll
        for (int s = 0; s < WPF; ++s) {
            if (kt + s < nk) {
                half8 xf[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    xf[i] = *reinterpret_cast<const half8 *>(As + ((wm * TM + i) * 16 + fr) * LDA + (kt + s) * 32 + fk);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[s], xf[i], acc[i], 0, 0, 0);
                if (kt + s + WPF < nk) {
                    half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (nrow < a.N && (kt + s + WPF) * 32 + fk < a.K)
                        v = *reinterpret_cast<const half8 *>(wrow + (kt + s + WPF) * 32);
                    wq[s] = v;
                }
            }
        }
    }
    YK_STAMP(4)
    __syncthreads();
    YK_STAMP(5)

    // (5) epilogue through LDS
    yk_half *Cs = reinterpret_cast<yk_half *>(yk_smem);
    const int nl = wn * 16 + (lane >> 4) * 4, n = n0 + nl;
    const float4 sc = *reinterpret_cast<const float4 *>(a.scale + n);
    const float4 bs = *reinterpret_cast<const float4 *>(a.bias + n);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = (wm * TM + i) * 16 + fr, m = m0 + ml;
        half4 h = {(yk_half)yk_actf(acc[i][0] * sc.x + bs.x, a.slope, a.cap), (yk_half)yk_actf(acc[i][1] * sc.y + bs.y, a.slope, a.cap),
                   (yk_half)yk_actf(acc[i][2] * sc.z + bs.z, a.slope, a.cap), (yk_half)yk_actf(acc[i][3] * sc.w + bs.w, a.slope, a.cap)};
        if (a.res && m < a.M && n < a.resp) {
            const half4 rr = *reinterpret_cast<const half4 *>(a.res + (size_t)m * a.resp + n);
            h = half4{(yk_half)((float)h[0] + (float)rr[0]), (yk_half)((float)h[1] + (float)rr[1]),
                      (yk_half)((float)h[2] + (float)rr[2]), (yk_half)((float)h[3] + (float)rr[3])};
        }
        *reinterpret_cast<half4 *>(Cs + ml * CS_LD + nl) = h;
    }
    __syncthreads();
    constexpr int VPR = BN / 8;
    yk_half *o = reinterpret_cast<yk_half *>(a.out);
    for (int v = tid; v < BM * VPR; v += NT) {
        const int row = v / VPR, cv = v - row * VPR, m = m0 + row, col = n0 + cv * 8;
        if (m < a.M && col < a.outp)
            *reinterpret_cast<half8 *>(o + (size_t)m * a.outp + col) = *reinterpret_cast<const half8 *>(Cs + row * CS_LD + cv * 8);
    }
    YK_STAMP(6)
    if (a.dbg && tid == 0)
        a.dbg[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + 7] =
            ((long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);
#undef YK_STAMP
}

// -------------------------------------------------------------------------------------
// "lr" (low-register, high-residency) variant for the large-spatial / few-channel blocks (Cin <= 128):
// these layers stream tens of MB with almost no reuse, so what matters is how many independent
// loads the CU keeps in flight.  256-thread workgroups, <= 64..80 VGPRs -> 6-8 workgroups per CU
// running out of phase; the pointwise GEMM walks N in passes of 48 columns so the accumulator stays
// 12*TM registers; depthwise weights come from LDS.
// -------------------------------------------------------------------------------------
template <int TM, int NP, int KS>                                    // NP: bound of the 48-column passes (N <= 48 * NP); KS: of the k-steps (Cin <= 32 * KS)
__global__ void __launch_bounds__(256, (TM == 2 && NP == 4) ? 4 : 5) fused_lr_kernel(const igemm_args a) {   // <2,4,4> would spill at 5
    constexpr int NT = 256, BM = 64 * TM, TN = 3;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int Cp = a.c0p, Kp = (Cp + 31) & ~31, LDA = Kp + a.lda_pad;
    const int npass = (a.N + 47) / 48, CS_LD = npass * 48 + 8;
    yk_half *As = reinterpret_cast<yk_half *>(yk_smem);
    yk_half *Ws = As + (size_t)BM * LDA;
    const int nsl = (a.N + 15) >> 4;                                 // 16-channel slices of the pointwise weights
    yk_half *Wp = Ws + (size_t)9 * Cp;                               // pointwise weights, MFMA fragment order [slice][k-step][lane][8]
    float *Sb = reinterpret_cast<float *>(Wp + (size_t)nsl * (Kp >> 5) * 512);   // pointwise BatchNorm scale | bias, [2][npass*48]
    yk_half *Cs = As;                                                // output tile: reuses the depthwise tile / weights after the GEMM
    const int m0 = yk_xcd_tile(blockIdx.x, gridDim.x) * BM;
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    const int nk = Kp >> 5;

    // the whole pointwise weight matrix (<= 18 KB) is requested into LDS by DMA before anything else: it lands under the depthwise
    // phase.  (Phase stamps of the previous version: the GEMM phase was 1.8 us of a 5.4 us workgroup lifetime for SIX MFMAs - it
    // opened with a round trip to L2 for the weight fragments.)
    {
        typedef __attribute__((address_space(3))) void *lds_ptr_t;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);
        const int w16 = nsl * nk * 64;
        for (int q0 = wid * 64; q0 < w16; q0 += NT) {
            const uint32_t offw = (uint32_t)(q0 + lane) * 16u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(Wp + (size_t)q0 * 8), 16, offw, 0, 0, 0);
        }
    }
    for (int v = tid; v < 9 * (Cp >> 3); v += NT)
        *reinterpret_cast<half8 *>(Ws + v * 8) = *reinterpret_cast<const half8 *>(a.dw_w + (size_t)v * 8);
    for (int v = tid; v < npass * 48; v += NT) {                     // arrays are zero-padded past N (yk_engine.hip upload_sb)
        Sb[v] = a.scale[v];
        Sb[npass * 48 + v] = a.bias[v];
    }
    {
        const int padv = (Kp - Cp) >> 3;
        for (int v = tid; v < BM * padv; v += NT) {
            const int p = v / padv, c = v - p * padv;
            *reinterpret_cast<half8 *>(As + p * LDA + Cp + c * 8) = half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    __syncthreads();

    // ---- phase A: depthwise, one pixel-octet per iteration (rows past M reuse the last pixel; never stored)
    {
        const int G = Cp >> 3;
        const int pl = (int)yk_div(tid, a.fd_g), g = tid - pl * G;
        const int PP = NT / G;
        if (pl < PP) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.in0, 0, a.in0_bytes, 0x00020000);
            const float4 sc0 = *reinterpret_cast<const float4 *>(a.dw_scale + g * 8);
            const float4 sc1 = *reinterpret_cast<const float4 *>(a.dw_scale + g * 8 + 4);
            const float4 bs0 = *reinterpret_cast<const float4 *>(a.dw_bias + g * 8);
            const float4 bs1 = *reinterpret_cast<const float4 *>(a.dw_bias + g * 8 + 4);
            const yk_half *wl = Ws + g * 8;
            const bool capped = a.dw_cap < 3.0e38f;
            for (int p = pl; p < BM; p += PP) {
                u32x4 x[9];
                dw_issue(a, rs, (uint32_t)min(m0 + p, a.M - 1), (uint32_t)g * 16u, x);
                const half8 h = capped ? dw_finish<true>(x, wl, Cp, sc0, sc1, bs0, bs1, a.dw_slope, a.dw_cap)
                                       : dw_finish<false>(x, wl, Cp, sc0, sc1, bs0, bs1, a.dw_slope, a.dw_cap);
                *reinterpret_cast<half8 *>(As + p * LDA + g * 8) = h;
            }
        }
    }
    __syncthreads();

    // ---- phase B: N in passes of 48 columns; wave w owns rows [w*16*TM, (w+1)*16*TM).  The results of all passes stay in registers
    // (packed fp16) until every wave is done reading the depthwise tile and the weights: the output tile then takes their place in
    // LDS, which keeps the workgroup's footprint at max(inputs, outputs) instead of their sum - more workgroups per CU.
    const int nl4 = (lane >> 4) * 4;
    const bool capped_o = a.cap < 3.0e38f;
    half8 xf[TM][KS];                                                // c0p <= 128: at most four k-steps of 32
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int kt = 0; kt < KS; ++kt)
            xf[i][kt] = *reinterpret_cast<const half8 *>(As + ((wid * TM + i) * 16 + fr) * LDA + min(kt, nk - 1) * 32 + fk);
    half4 outv[NP][TM][TN];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
        if (ps < npass) {
            floatx4 acc[TM][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < KS; ++kt) {
                if (kt < nk) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int sl = min(ps * TN + j, nsl - 1);    // slices past N: a valid address, the result is never stored
                        const half8 wf = *reinterpret_cast<const half8 *>(Wp + (((size_t)sl * nk + kt) * 64 + lane) * 8);
#pragma unroll
                        for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[i][kt], acc[i][j], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nl = ps * 48 + j * 16 + nl4;
                const float4 sc = *reinterpret_cast<const float4 *>(Sb + nl);
                const float4 bs = *reinterpret_cast<const float4 *>(Sb + npass * 48 + nl);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int ml = (wid * TM + i) * 16 + fr, m = m0 + ml;
                    half4 h = capped_o ? epi4<true>(acc[i][j], sc, bs, a.slope, a.cap) : epi4<false>(acc[i][j], sc, bs, a.slope, a.cap);
                    if (a.res && m < a.M && nl < a.resp) {
                        const half4 rr = *reinterpret_cast<const half4 *>(a.res + (size_t)m * a.resp + nl);
                        h = half4{(yk_half)((float)h[0] + (float)rr[0]), (yk_half)((float)h[1] + (float)rr[1]),
                                  (yk_half)((float)h[2] + (float)rr[2]), (yk_half)((float)h[3] + (float)rr[3])};
                    }
                    outv[ps][i][j] = h;
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < NP; ++ps)
        if (ps < npass)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    *reinterpret_cast<half4 *>(Cs + ((wid * TM + i) * 16 + fr) * CS_LD + ps * 48 + j * 16 + nl4) = outv[ps][i][j];
    __syncthreads();
    const int VPR = a.outp >> 3;
    yk_half *o = reinterpret_cast<yk_half *>(a.out);
    for (int v = tid; v < BM * VPR; v += NT) {
        const int row = (int)yk_div(v, a.fd_vpr), cv = v - row * VPR, m = m0 + row;
        if (m < a.M)
            *reinterpret_cast<half8 *>(o + (size_t)m * a.outp + cv * 8) = *reinterpret_cast<const half8 *>(Cs + row * CS_LD + cv * 8);
    }
}

template <int TM>
static int launch_lr(const igemm_args &a, hipStream_t st) {
    constexpr int BM = 64 * TM;
    const int Kp = (a.c0p + 31) & ~31, npass = (a.N + 47) / 48;
    const size_t lds_in = ((size_t)BM * (Kp + a.lda_pad) + (size_t)9 * a.c0p + (size_t)((a.N + 15) / 16) * (Kp / 32) * 512) * 2 + (size_t)2 * npass * 48 * 4;
    const size_t lds = std::max(lds_in, (size_t)BM * (npass * 48 + 8) * 2);   // the output tile reuses the input side
    // the results of every pass and the pixel fragments of every k-step stay in registers: instantiating for the real counts (1, 2 or
    // up to 4 each) keeps them at 12+8 / 24+16 / 48+32 registers per 16 rows instead of always the maximum, which spilled at the
    // 5-waves-per-SIMD budget (24->48: 30.9 -> 28.9 us).  Six waves per SIMD (80 registers, one spill) measured slower: 29.5 us.
    auto go = [&](auto kern) {
        static size_t attr_lds = 64 * 1024;
        if (lds > attr_lds) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_lds = lds;
        }
        hipLaunchKernelGGL(kern, dim3((a.M + BM - 1) / BM), dim3(256), lds, st, a);
    };
    const int nk = Kp / 32;
    if (npass <= 1 && nk <= 1) go(fused_lr_kernel<TM, 1, 1>);
    else if (npass <= 2 && nk <= 2) go(fused_lr_kernel<TM, 2, 2>);
    else go(fused_lr_kernel<TM, 4, 4>);
    return YK_OK;
}

template <int WM, int WN, int TM, int WPF>
static int launch_wide(const igemm_args &a, hipStream_t st) {
    constexpr int BM = WM * 16 * TM, BN = WN * 16;
    const int Kp = (a.c0p + 31) & ~31;
    size_t lds = (size_t)BM * (Kp + a.lda_pad) * 2 + (size_t)9 * a.c0p * 2, cs = (size_t)BM * (BN + 8) * 2;
    if (cs > lds) lds = cs;
    static size_t attr_lds = 64 * 1024;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_wide_kernel<WM, WN, TM, WPF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN);
    hipLaunchKernelGGL((fused_wide_kernel<WM, WN, TM, WPF>), grid, dim3(768), lds, st, a);
    return YK_OK;
}

template <int BM, int BN, int WM, int WN, int IT>
static int launch_fused(const igemm_args &a, hipStream_t st) {
    const int Kp = (a.c0p + 31) & ~31;
    size_t lds = (size_t)BM * (Kp + a.lda_pad) * 2, cs = (size_t)BM * (BN + 8) * 2;
    if (cs > lds) lds = cs;
    static size_t attr_lds = 64 * 1024;   // opt in to > 64 KiB dynamic LDS only when a layer needs it
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_dwpw_kernel<BM, BN, WM, WN, IT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN);
    hipLaunchKernelGGL((fused_dwpw_kernel<BM, BN, WM, WN, IT>), grid, dim3(64 * WM * WN), lds, st, a);
    return YK_OK;
}

#include "yk_fused_dma.h"

bool yk_igemm_fused_ok(int c0p, int cout) {
    // the whole K extent of the depthwise tile must fit LDS at the smallest BM (32 rows)
    const int Kp = (c0p + 31) & ~31;
    return c0p % 8 == 0 && (size_t)32 * (Kp + yk_fused_pad()) * 2 <= 96 * 1024 && (256 / (c0p >> 3)) >= 1 && cout >= 8;
}
int yk_igemm_fused_pick(const igemm_args &a) {
    const int G = a.c0p >> 3;
    // small-spatial blocks (<= 28x40 at batch 32): the LDS-DMA staged kernel (yk_fused_dma.h)
    static const bool fdma_on = yk_dev_env("YK_FDMA") ? yk_dev_env("YK_FDMA")[0] != '0' : true;
    if (fdma_on && a.c0p >= 96 && a.M <= 65536) {
        // measured against fused_wide at batch 32: 96->192 s2 21.3 -> 18.5 us, 384->384 17.7 -> 15.7, 384->768 s2 17.7 -> 14.4; the
        // 192->192 block (448 patches = two rounds of one 768-thread workgroup per CU) 18.1 -> 20.9: keeps the old kernel
        const fdma_plan pl = yk_fdma_plan(a);
        if (pl.npw && (pl.rounds == 1 || a.c0p <= 96)) return FUSED_DMA;
    }
    static const bool wide = yk_dev_env("YK_WIDE") ? yk_dev_env("YK_WIDE")[0] != '0' : true;
    static const bool lr = yk_dev_env("YK_LR") ? yk_dev_env("YK_LR")[0] != '0' : true;
    // (an LDS-DMA staged variant of this kernel was built and measured: 34.5 vs 33.0 us on the 24->48 block - these layers are bound by
    // VALU issue, ~650 vector instructions per wave of which 108 are the depthwise MACs, not by how the taps are fetched; dropped)
    if (lr && a.c0p <= 128 && a.N <= 192 && (long)a.M * a.N >= (1l << 22)) {
        if (const char *e = yk_dev_env("YK_LR_TM")) return atoi(e) == 2 ? LR_T2 : LR_T1;            // dev sweep
        return (a.c0p <= 32) ? LR_T2 : LR_T1;   // measured (us, 128- vs 64-pixel tile): 24 ch 28.0 / 33.0, 48 ch 23.5 / 22.3, 96 ch 28.3 / 26.2
    }
    if (wide && 768 % G == 0 && a.c0p <= 768) {
        // aim at one or two depthwise items per thread: BM * G ~ 768..1536
        if (a.N <= 48) return WIDE_4x3_T4;                                  // BM 256
        if (a.N <= 96) return (G <= 6) ? WIDE_2x6_T4 : WIDE_2x6_T2;         // BM 128 / 64
        // N >= 192: waves split N (12 x 16 columns per slice).  Fit the grid to ONE round of 256 CUs:
        // a workgroup's life here is a fixed chain of memory round trips, so 3 rounds cost 3x.
        const int ns = (a.N + 191) / 192;
        const int tiles_m = 256 / ns > 0 ? 256 / ns : 1;
        const int tm = ((a.M + tiles_m - 1) / tiles_m + 15) / 16;
        if (tm <= 2) return WIDE_1x12_T2;
        if (tm <= 4) return WIDE_1x12_T4;
        if (tm <= 5) return WIDE_1x12_T5;
        return WIDE_1x12_T9;
    }
    if (a.N <= 48) return FUSED_128x48;
    if (a.N <= 96) return FUSED_128x96;
    const long m64 = (a.M + 63) / 64;
    if (a.N <= 192 && m64 >= 256 && a.c0p <= 384) return FUSED_64x192;
    return FUSED_32x192;
}
const char *yk_igemm_fused_name(int cfg) {
    static const char *n[] = {"fused_128x48", "fused_128x96", "fused_64x192", "fused_32x192", "wide_256x48", "wide_128x96",
                              "wide_64x96", "wide_64x192", "wide_32x192", "lr_64", "lr_128", "wide_80x192", "wide_144x192", "fdma"};
    return (cfg >= 0 && cfg < FUSED_NUM) ? n[cfg] : "?";
}
int yk_launch_igemm_fused(int cfg, const igemm_args &a, hipStream_t st) {
    switch (cfg) {
    case FUSED_128x48: return launch_fused<128, 48, 4, 1, 1>(a, st);
    case FUSED_128x96: return launch_fused<128, 96, 4, 1, 1>(a, st);
    case FUSED_64x192: return launch_fused<64, 192, 2, 2, 2>(a, st);
    case FUSED_32x192: return launch_fused<32, 192, 1, 4, 2>(a, st);
    case WIDE_4x3_T4: return launch_wide<4, 3, 4, 6>(a, st);
    case WIDE_2x6_T4: return launch_wide<2, 6, 4, 6>(a, st);
    case WIDE_2x6_T2: return launch_wide<2, 6, 2, 6>(a, st);
    case WIDE_1x12_T4: return launch_wide<1, 12, 4, 6>(a, st);
    case WIDE_1x12_T2: return launch_wide<1, 12, 2, 6>(a, st);
    case WIDE_1x12_T5: return launch_wide<1, 12, 5, 6>(a, st);
    case WIDE_1x12_T9: return launch_wide<1, 12, 9, 6>(a, st);
    case LR_T1: return launch_lr<1>(a, st);
    case LR_T2: return launch_lr<2>(a, st);
    case FUSED_DMA: return yk_launch_fdma(a, st);
    }
    yk_set_error("yk_launch_igemm_fused: bad config %d", cfg);
    return YK_ERR_ARG;
}
