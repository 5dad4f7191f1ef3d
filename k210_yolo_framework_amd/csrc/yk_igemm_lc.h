// yk_igemm_lc.h — the implicit-GEMM conv with DEDICATED LOADER WAVES (round 5; included by yk_conv.hip behind yk_igemm_pipe.h).
//
// Same math, operand layout, swizzle and ring as igemm_pipe_kernel (Conv2D 1x1 / 3x3 over [src0, src1] as C[M,N] = A[M,K] * W[N,K]^T on
// v_mfma_f32_16x16x32_f16, k-steps of 64, NS-deep LDS ring filled by `buffer_load_dwordx4 ... lds`), different division of labour:
//
//   waves 0 .. WM*WN-1   CONSUMERS: fragment reads + MFMAs, nothing else - they never execute a vector-memory instruction
//   waves WM*WN ..       LOADERS  : issue every DMA piece of the prefetch step, wait for their own pieces (vmcnt), nothing else
//
// Why (profiles/r05_igemm_phase.txt): issuing one 1 KB piece blocks the issuing wave ~100 cycles wherever the instruction sits
// in its stream, and a CU's vector-memory path takes 64 B per clock.  In igemm_pipe_kernel every wave does both jobs, a barrier releases all
// of them at once, so they queue on that path together with the matrix pipe idle, then compete for the matrix pipe together: 550-700 TFLOP/s
// for every tile shape.  Here a SIMD holds one consumer and one loader (a workgroup's waves w and w + 4 share a SIMD): the loader's blocked
// issue slots cost the consumer nothing, the consumer's MFMAs run under the loader's pieces.  One s_barrier per k-step for all eight waves
// (the loaders arrive when their pieces of the NEXT step have landed, the consumers when their MFMAs of this step are issued).
#pragma once

template <int BM, int BN, int WM, int WN, int NLW, int NS, int OUT>
__global__ void __launch_bounds__(64 * (WM * WN + NLW)) igemm_lc_kernel(const igemm_args a) {
    constexpr int NCW = WM * WN, BK = 64, NT = 64 * (NCW + NLW);
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    static_assert((BM / 8) % NLW == 0 && (BN / 8) % NLW == 0, "8-row DMA groups must divide among the loader waves");
    constexpr int A_IT = BM / 8 / NLW, B_IT = BN / 8 / NLW, L = A_IT + B_IT;     // pieces per LOADER wave and k-step
    constexpr int STAGE = (BM + BN) * BK;                                         // halfs
    static_assert(NS >= 3 && (NS - 2) * L <= 63, "the loaders run one step ahead of the barrier; vmcnt is a 6-bit counter");
    yk_half *lds = reinterpret_cast<yk_half *>(yk_smem);
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gx = gridDim.x, gy = gridDim.y;
    const int L0 = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int v = yk_xcd_tile(L0, gx * gy * gridDim.z);
    const int vz = v / (gx * gy), vr = v - vz * (gx * gy), vy = vr / gx, vx = vr - vy * gx;
    const int m0 = vx * BM, n0 = vy * BN;
    const int Ctp = a.c0p + a.c1p;
    const int taps = a.ks * a.ks;
    const int nk_all = (a.K + BK - 1) / BK;
    const int per = (nk_all + a.split_k - 1) / a.split_k;
    const int kt0 = vz * per;
    const int nk = max(0, min(per, nk_all - kt0));
    typedef __attribute__((address_space(3))) void *lds_ptr_t;

    if (wid >= NCW) {
        // ------------------------------------------------------------------ LOADER
        const int lw = wid - NCW;
        const int rr = lane >> 3, gc = (lane & 7) ^ rr;           // row inside the 8-row group, global chunk this lane fetches
        uint32_t P0[A_IT], P1[A_IT], rmask[A_IT], wro[B_IT];
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int m = m0 + (lw + it * NLW) * 8 + rr;
            const bool ok = m < a.M;
            const uint32_t mm = ok ? m : 0;
            const uint32_t b = yk_div(mm, a.fd_hw), rem = mm - b * (a.Ho * a.Wo);
            const uint32_t oy = yk_div(rem, a.fd_wo), ox = rem - oy * a.Wo;
            const int ry0 = (int)oy * a.stride - a.pad_t, rx0 = (int)ox * a.stride - a.pad_l;
            P0[it] = b * (uint32_t)(a.Hi * a.Wi * a.c0p * 2) + (uint32_t)((ry0 * a.Wi + rx0) * a.c0p) * 2u + gc * 16u;
            P1[it] = b * (uint32_t)(a.Hi * a.Wi * a.c1p * 2) + (uint32_t)((ry0 * a.Wi + rx0) * a.c1p - a.c0p) * 2u + gc * 16u;
            uint32_t msk = 0;
            for (int t = 0; t < taps; ++t) {
                const int ky = (a.ks == 3) ? t / 3 : 0, kx = t - ky * a.ks;
                if (ok && (unsigned)(ry0 + ky) < (unsigned)a.Hi && (unsigned)(rx0 + kx) < (unsigned)a.Wi) msk |= 1u << t;
            }
            rmask[it] = msk;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int n = n0 + (lw + it * NLW) * 8 + rr;
            wro[it] = (n < a.N) ? (uint32_t)(n * a.K) * 2u + gc * 16u : YK_OOB;
        }
        const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void *)a.in0, 0, a.in0_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void *)(a.in1 ? a.in1 : a.in0), 0, a.in1 ? a.in1_bytes : a.in0_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);
        const int lim = kt0 + nk;
        int step = kt0;
        int tap = (int)yk_div((uint32_t)kt0 * BK, a.fd_ctp);
        int cin = kt0 * BK - tap * Ctp;
        uint32_t aoff0[A_IT], aoff1[A_IT];
        auto retap = [&]() {
            const int ky = (a.ks == 3) ? (tap * 11) >> 5 : 0, kx = tap - ky * a.ks;
            const uint32_t t0 = (uint32_t)((ky * a.Wi + kx) * a.c0p) * 2u, t1 = (uint32_t)((ky * a.Wi + kx) * a.c1p) * 2u;
            const bool tlive = tap < taps;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const bool ok = tlive && ((rmask[it] >> tap) & 1u);
                aoff0[it] = ok ? P0[it] + t0 : YK_OOB;
                aoff1[it] = ok ? P1[it] + t1 : YK_OOB;
            }
        };
        retap();
        auto dma = [&](int stage) {
            const bool live = step < lim;
            const uint32_t cs = live ? (uint32_t)cin * 2u : YK_OOB;                  // dead steps (past the split's end) deposit zeros
            const uint32_t ws = live ? (uint32_t)step * (BK * 2u) : YK_OOB;
            yk_half *As = lds + stage * STAGE, *Bs = As + BM * BK;
            if (cin >= a.c0p) {
#pragma unroll
                for (int it = 0; it < A_IT; ++it) {
                    const uint32_t off = aoff1[it] + cs;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr_t)(As + (lw + it * NLW) * 8 * BK), 16, off, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int it = 0; it < A_IT; ++it) {
                    const uint32_t off = aoff0[it] + cs;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr_t)(As + (lw + it * NLW) * 8 * BK), 16, off, 0, 0, 0);
                }
            }
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                lds_ptr_t dstb = (lds_ptr_t)(Bs + (lw + it * NLW) * 8 * BK);
                const uint32_t offb = wro[it] + ws;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, dstb, 16, offb, 0, 0, 0);
            }
            ++step;
            cin += BK;
            if (cin >= Ctp) {                                                         // uniform, every Ctp/64 steps
                cin = 0;
                ++tap;
                retap();
            }
        };
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) dma(s);
        int wr = NS - 1;
        for (int kt = 0; kt < nk; ++kt) {
            yk_wait_vm_lgkm0<(NS - 2) * L>();                      // this wave's pieces of step kt have landed
            __builtin_amdgcn_s_barrier();                          // everybody's have; the consumers are done with stage (kt - 1) % NS
            asm volatile("" ::: "memory");
            dma(wr);
            wr = (wr + 1 == NS) ? 0 : wr + 1;
        }
        yk_wait_vm_lgkm0<0>();                                     // the dead prefetches, before LDS becomes the output tile
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (OUT == 0) {                                  // the output tile leaves through LDS: the loaders help copying it out
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            constexpr int VPR = BN / 8, CS_LD = BN + 8;
            yk_half *o = reinterpret_cast<yk_half *>(a.out);
            for (int q = tid; q < BM * VPR; q += NT) {
                const int row = q / VPR, cv = q - row * VPR, m = m0 + row, col = n0 + cv * 8;
                if (m < a.M && col < a.outp)
                    *reinterpret_cast<half8 *>(o + (size_t)m * a.outp + col) = *reinterpret_cast<const half8 *>(lds + row * CS_LD + cv * 8);
            }
        }
        return;
    }
    // ---------------------------------------------------------------------- CONSUMER
    const int wm = wid / WN, wn = wid % WN;
    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, sw = fr & 7, fq = lane >> 4;
    half8 wf0[TN], xf0[TM], wf1[TN], xf1[TM];
    auto read_frags = [&](int stage, int ks, half8 (&wf)[TN], half8 (&xf)[TM]) {
        const yk_half *As = lds + stage * STAGE, *Bs = As + BM * BK;
        const int ch = ((ks * 4 + fq) ^ sw) * 8;
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const half8 *>(Bs + ((wn * TN + j) * 16 + fr) * BK + ch);
#pragma unroll
        for (int i = 0; i < TM; ++i) xf[i] = *reinterpret_cast<const half8 *>(As + ((wm * TM + i) * 16 + fr) * BK + ch);
    };
    int rd = 0;
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_barrier();                              // stage rd is complete (the loaders waited for it before arriving)
        asm volatile("" ::: "memory");
        read_frags(rd, 0, wf0, xf0);
        read_frags(rd, 1, wf1, xf1);                               // the second half-step's fragments fly under the first one's MFMAs
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0[j], xf0[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[j], xf1[i], acc[i][j], 0, 0, 0);
        rd = (rd + 1 == NS) ? 0 : rd + 1;
    }
    __builtin_amdgcn_s_barrier();                                  // (pairs with the loaders' final barrier: LDS is free)
    asm volatile("" ::: "memory");
    // ---- epilogue of the consumers (igemm_epilogue's arithmetic; the copy-out loop runs on all NT threads)
    const int nl4 = (lane >> 4) * 4;
    if constexpr (OUT == 2) {
        float *slab = a.slab + (size_t)vz * a.M * a.ldn;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + (wm * TM + i) * 16 + fr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + (wn * TN + j) * 16 + nl4;
                if (m < a.M && n < a.ldn)
                    *reinterpret_cast<float4 *>(slab + (size_t)m * a.ldn + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
    } else {
        constexpr int CS_LD = BN + 8;
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
                const float v0 = yk_actf(acc[i][j][0] * sc[j].x + bs[j].x, a.slope, a.cap);
                const float v1 = yk_actf(acc[i][j][1] * sc[j].y + bs[j].y, a.slope, a.cap);
                const float v2 = yk_actf(acc[i][j][2] * sc[j].z + bs[j].z, a.slope, a.cap);
                const float v3 = yk_actf(acc[i][j][3] * sc[j].w + bs[j].w, a.slope, a.cap);
                half4 h = {(yk_half)v0, (yk_half)v1, (yk_half)v2, (yk_half)v3};
                if (a.res && m < a.M && n < a.resp) {
                    const half4 r = *reinterpret_cast<const half4 *>(a.res + (size_t)m * a.resp + n);
                    h = half4{(yk_half)((float)h[0] + (float)r[0]), (yk_half)((float)h[1] + (float)r[1]),
                              (yk_half)((float)h[2] + (float)r[2]), (yk_half)((float)h[3] + (float)r[3])};
                }
                *reinterpret_cast<half4 *>(lds + ml * CS_LD + nl) = h;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // (pairs with the loaders' copy-out barrier)
        asm volatile("" ::: "memory");
        constexpr int VPR = BN / 8;
        yk_half *o = reinterpret_cast<yk_half *>(a.out);
        for (int q = tid; q < BM * VPR; q += NT) {
            const int row = q / VPR, cv = q - row * VPR, m = m0 + row, col = n0 + cv * 8;
            if (m < a.M && col < a.outp)
                *reinterpret_cast<half8 *>(o + (size_t)m * a.outp + col) = *reinterpret_cast<const half8 *>(lds + row * CS_LD + cv * 8);
        }
    }
}

template <int BM, int BN, int WM, int WN, int NLW, int NS>
static int launch_lc(const igemm_args &a, hipStream_t st) {
    constexpr size_t ring = (size_t)NS * (BM + BN) * 64 * 2, ct = (size_t)BM * (BN + 8) * 2;
    constexpr size_t ldsd = ring > ct ? ring : ct;
    dim3 g2((a.M + BM - 1) / BM, (a.N + BN - 1) / BN, a.split_k > 1 ? a.split_k : 1);
    auto go = [&](auto kern) {
        static bool done = false;
        if (!done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsd);
            done = true;
        }
        hipLaunchKernelGGL(kern, g2, dim3(64 * (WM * WN + NLW)), ldsd, st, a);
    };
    if (a.split_k > 1) go(igemm_lc_kernel<BM, BN, WM, WN, NLW, NS, 2>);
    else go(igemm_lc_kernel<BM, BN, WM, WN, NLW, NS, 0>);
    return YK_OK;
}
