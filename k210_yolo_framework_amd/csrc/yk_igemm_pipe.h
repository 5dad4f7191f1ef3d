// yk_igemm_pipe.h — the implicit-GEMM conv with a multi-stage LDS-DMA operand pipeline (included by yk_conv.hip).
//
// Conv2D 1x1 / 3x3 (stride 1|2) over [up2(src0), src1] as C[M,N] = A[M,K] * W[N,K]^T on v_mfma_f32_16x16x32_f16, K walked in
// steps of 64.  Both operand tiles are deposited by `buffer_load_dwordx4 ... lds` (1 KB per wave-instruction = 8 tile rows of
// 128 B, no staging registers, no ds_write pass) into an NS-deep ring of LDS stages:
//
//   prologue   DMA of steps 0 .. NS-2
//   step kt    s_waitcnt vmcnt((NS-2)*L)      this wave's pieces of step kt have landed (L = DMA instructions per wave and step)
//              s_barrier                      everybody's have, and everybody is done reading stage (kt-1) % NS
//              DMA of step kt+NS-1  ->  stage (kt-1) % NS
//              MFMAs on stage kt % NS
//
// so NS-1 steps of loads are in flight under the MFMAs, there is ONE barrier per step and the load queue is never drained inside
// the loop (a `__syncthreads()` would: hipcc puts vmcnt(0) in front of it while an LDS-DMA is pending - cdna_hip_programming.md,
// "Pipelining across barriers").  Bank conflicts of the fragment reads are avoided by the XOR swizzle chunk ^= row & 7 applied to
// the SOURCE address of each lane and to the fragment read (rule 21 of the guide; tools/lds_sim.py, tests/test_lds_model.py).
// Out-of-image taps, rows >= M, columns >= N and steps past the end of a K split get an offset beyond the descriptor's
// num_records: the hardware writes zeros, the loop body has no branches.
// UP: src0 is read through UpSampling2D(2) (yolonet.py:31-38 head, Darknet FPN :167-172): its tap offset is not linear in the tap,
// the row keeps (y0, x0) instead of a precomputed pointer.
#pragma once

#include <mutex>
#include <set>
#include <utility>

template <int N>
__device__ __forceinline__ void yk_wait_vm_lgkm0() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
#ifdef YK_DEV
#include "yk_igemm_br.h"                                           // (developer builds: the variant with the weight fragments in registers)
#endif

// IL (round 5): the DMA pieces of the step being prefetched are issued BETWEEN the MFMAs of the step being computed, one piece per few MFMAs
// (an LDS-DMA instruction costs its wave ~150 cycles of issue when a whole step's pieces go out in one block ahead of the fragment reads,
// ~60 in the shadow of the matrix pipe - MI355X_MICROARCH.md "LDS-DMA piece issue cost"; tools/r05_igemm_sweep.py: without it every tile
// shape from 64x128 to 256x128 sits at 550-700 TFLOP/s at B=32, i.e. the loop is bound by DMA issue, not by the tile).
// PS (round 5, phase split): half of the waves of a SIMD issue the step's DMA pieces BEFORE their MFMAs, the other half AFTER (needs NS >= 3:
// a piece issued at the end of step k is waited for at the start of step k + 2).  A barrier releases every wave at once; unsplit, all of them
// then sit in the vector-memory issue queue together (~100 cycles per 1 KB piece, the matrix pipe idle) and afterwards compete for the matrix
// pipe together (tools/r05_igemm_phase.py: 870 cycles of DMA issue + 1450 of MFMAs + 1200 at the barrier per k-step for 1024 cycles of MFMA).
// PS 1: by wave index (8-wave workgroups: waves w and w + 4 share a SIMD); PS 2: by the hardware wave slot's parity (two 4-wave workgroups per CU).
template <int BM, int BN, int WM, int WN, int NS, int OUT, bool UP, bool IL = false, int PS = 0>
__global__ void __launch_bounds__(64 * WM * WN) igemm_pipe_kernel(const igemm_args a) {
    constexpr int NW = WM * WN, BK = 64;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "8-row DMA groups must divide among the waves");
    constexpr int A_IT = BM / 8 / NW, B_IT = BN / 8 / NW, L = A_IT + B_IT;
    constexpr int STAGE = (BM + BN) * BK;                         // halfs
    static_assert(NS >= 2 && (NS - 2) * L <= 63, "vmcnt is a 6-bit counter");
    yk_half *lds = reinterpret_cast<yk_half *>(yk_smem);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    // XCD-aware walk of the whole 3-D grid: workgroup L (dispatch order, x fastest) runs on XCD L % 8; renumber so that an XCD owns a
    // contiguous run of (K-split, N-tile, M-tile) triples with M fastest - neighbouring M tiles share halo rows and the same weight
    // panel, and the run's working set (one weight slice + a band of the input) stays inside that XCD's 4 MB L2.
    // (Tried: cutting M into 8 bands with (K-split, N-tile) as the outer loop inside a band, so that an XCD keeps ONE band of the input:
    // single layers +2-3 %, the Darknet-53 stack -5 %, the 7x10 head conv much worse - dropped.)
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
    const int nk = min(per, nk_all - kt0);
    const int rr = lane >> 3, gc = (lane & 7) ^ rr;               // row inside the 8-row group, global chunk this lane fetches
    const int W0 = a.Wi >> 1;                                     // UP: width of the un-upsampled source

    uint32_t P0[A_IT], P1[A_IT], rmask[A_IT], wro[B_IT];
    int ry[A_IT], rx[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int m = m0 + (wid + it * NW) * 8 + rr;
        const bool ok = m < a.M;
        const uint32_t mm = ok ? m : 0;
        const uint32_t b = yk_div(mm, a.fd_hw), rem = mm - b * (a.Ho * a.Wo);
        const uint32_t oy = yk_div(rem, a.fd_wo), ox = rem - oy * a.Wo;
        const int ry0 = (int)oy * a.stride - a.pad_t, rx0 = (int)ox * a.stride - a.pad_l;
        ry[it] = ry0;
        rx[it] = rx0;
        if constexpr (UP) P0[it] = b * (uint32_t)((a.Hi >> 1) * W0 * a.c0p * 2) + gc * 16u;
        else P0[it] = b * (uint32_t)(a.Hi * a.Wi * a.c0p * 2) + (uint32_t)((ry0 * a.Wi + rx0) * a.c0p) * 2u + gc * 16u;
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
        const int n = n0 + (wid + it * NW) * 8 + rr;
        wro[it] = (n < a.N) ? (uint32_t)(n * a.K) * 2u + gc * 16u : YK_OOB;
    }
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void *)a.in0, 0, a.in0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void *)(a.in1 ? a.in1 : a.in0), 0, a.in1 ? a.in1_bytes : a.in0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void *lds_ptr_t;

    // Per-step address work is ONE add per DMA instruction: everything that depends on the filter tap (row validity, the tap's byte
    // offset in either source, the upsampled coordinates) is folded into aoff0/aoff1 when the tap CHANGES, i.e. every Ctp/64 steps
    // (counters of the previous version: 85 vector + 65 scalar instructions per 16 MFMAs, most of them this arithmetic).
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
            uint32_t o0;
            if constexpr (UP) o0 = P0[it] + (uint32_t)((((ry[it] + ky) >> 1) * W0 + ((rx[it] + kx) >> 1)) * a.c0p) * 2u;
            else o0 = P0[it] + t0;
            aoff0[it] = ok ? o0 : YK_OOB;
            aoff1[it] = ok ? P1[it] + t1 : YK_OOB;
        }
    };
    retap();
    // one step's DMA = prepare (uniform offsets) + A_IT + B_IT pieces + advance (walk state)
    uint32_t d_cs = 0, d_ws = 0;
    bool d_second = false;
    auto dma_prepare = [&]() {
        const bool live = step < lim;
        d_cs = live ? (uint32_t)cin * 2u : YK_OOB;                                 // dead steps (past the split's end) deposit zeros
        d_ws = live ? (uint32_t)step * (BK * 2u) : YK_OOB;
        d_second = cin >= a.c0p;
    };
    auto dma_piece = [&](int stage, int n) {                                      // n: a compile-time constant after unrolling
        yk_half *As = lds + stage * STAGE, *Bs = As + BM * BK;
        if (n < A_IT) {
            lds_ptr_t dsta = (lds_ptr_t)(As + (wid + n * NW) * 8 * BK);
            if (d_second) {
                const uint32_t off = aoff1[n < A_IT ? n : 0] + d_cs;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dsta, 16, off, 0, 0, 0);
            } else {
                const uint32_t off = aoff0[n < A_IT ? n : 0] + d_cs;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dsta, 16, off, 0, 0, 0);
            }
        } else {
            const int it = n - A_IT;
            lds_ptr_t dstb = (lds_ptr_t)(Bs + (wid + it * NW) * 8 * BK);
            const uint32_t offb = wro[it < B_IT ? (it < 0 ? 0 : it) : 0] + d_ws;     // a named local (see the host-pass note in yk_igemm_pipe.h's first version)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, dstb, 16, offb, 0, 0, 0);
        }
    };
    auto dma_advance = [&]() {
        ++step;
        cin += BK;
        if (cin >= Ctp) {                                                         // uniform, every Ctp/64 steps
            cin = 0;
            ++tap;
            retap();
        }
    };
    auto dma = [&](int stage) {
        dma_prepare();
#pragma unroll
        for (int n = 0; n < L; ++n) {
            dma_piece(stage, n);
        }
        dma_advance();
    };
    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, sw = fr & 7, fq = lane >> 4;
    auto compute = [&](int stage) {
        const yk_half *As = lds + stage * STAGE, *Bs = As + BM * BK;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = ((ks * 4 + fq) ^ sw) * 8;             // swizzled 16-byte chunk of this lane's fragment
            half8 wf[TN], xf[TM];
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const half8 *>(Bs + ((wn * TN + j) * 16 + fr) * BK + ch);
#pragma unroll
            for (int i = 0; i < TM; ++i) xf[i] = *reinterpret_cast<const half8 *>(As + ((wm * TM + i) * 16 + fr) * BK + ch);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
    };
    // IL: the software-pipelined step (round 5).  Counters of the plain loop on a Darknet 52x52 128->256 3x3 at 32 images (profiles/r05_igemm_pmc.txt):
    // matrix pipe busy 0.28, waves 34-40 % parked at s_waitcnt / s_barrier, 2600 cycles per wave and k-step for 512 cycles of MFMA - a wave's
    // step was a serial chain  barrier -> 8 fragment reads (~250 cycles exposed) -> 16 MFMAs -> 8 reads -> 16 MFMAs, with the step's DMA pieces
    // issued in one block in front.  Here the two half-steps' fragments live in two register sets: the reads of the second half go out before
    // the first half's MFMAs, the reads of the NEXT step's first half right behind the barrier that ends this step (under the tail of its
    // MFMAs), and the prefetch step's DMA pieces are spread between the MFMAs, one per NM / L of them.
    half8 wf0[TN], xf0[TM], wf1[TN], xf1[TM];
    auto read_frags = [&](int stage, int ks, half8 (&wf)[TN], half8 (&xf)[TM]) {
        const yk_half *As = lds + stage * STAGE, *Bs = As + BM * BK;
        const int ch = ((ks * 4 + fq) ^ sw) * 8;
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const half8 *>(Bs + ((wn * TN + j) * 16 + fr) * BK + ch);
#pragma unroll
        for (int i = 0; i < TM; ++i) xf[i] = *reinterpret_cast<const half8 *>(As + ((wm * TM + i) * 16 + fr) * BK + ch);
    };
    auto mma_half = [&](int ks, const half8 (&wf)[TN], const half8 (&xf)[TM], int wstage) {
        constexpr int NM = 2 * TM * TN;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
                const int idx = (ks * TM + i) * TN + j;                           // MFMA number inside the step
                const int p0 = idx * L / NM, p1 = (idx + 1) * L / NM;             // pieces [p0, p1) go out here
                if (p1 > p0) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int pp = 0; pp < (L + NM - 1) / NM + 1; ++pp)
                        if (p0 + pp < p1) {
                            dma_piece(wstage, p0 + pp);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    };
#ifdef YK_DEV
    // developer build: where a wave's loop time goes (cycles of wave 0, summed over the k-steps) -> a.dbg[workgroup][8] (tools/r05_igemm_phase.py)
    long long c_wait = 0, c_bar = 0, c_dma = 0, c_mma = 0, c_t = 0;
    const long long wall0 = a.dbg ? (long long)wall_clock64() : 0;
#define PIPE_CLK(v) if (a.dbg) { const long long n_ = (long long)__builtin_readcyclecounter(); v += n_ - c_t; c_t = n_; }
#define PIPE_CLK0() if (a.dbg) c_t = (long long)__builtin_readcyclecounter();
#else
#define PIPE_CLK(v)
#define PIPE_CLK0()
#endif
    static_assert(PS == 0 || NS >= 3, "a late DMA needs a step of slack");
    bool late_dma = false;
    if constexpr (PS == 1) late_dma = wid >= NW / 2;
    if constexpr (PS == 2) late_dma = (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1) != 0;    // hwreg(HW_REG_HW_ID, 0, 4): wave slot inside its SIMD
    if (nk > 0) {
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) dma(s);                   // steps past `lim` deposit zeros and keep the vmcnt arithmetic uniform
        int rd = 0, wr = NS - 1;                                   // stage read this step / stage refilled this step
        PIPE_CLK0()
        if constexpr (IL) {
            yk_wait_vm_lgkm0<(NS - 2) * L>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            read_frags(0, 0, wf0, xf0);
            for (int kt = 0; kt < nk; ++kt) {
                dma_prepare();
                read_frags(rd, 1, wf1, xf1);
                __builtin_amdgcn_sched_barrier(0);
                mma_half(0, wf0, xf0, wr);
                mma_half(1, wf1, xf1, wr);
                dma_advance();
                rd = (rd + 1 == NS) ? 0 : rd + 1;
                wr = (wr + 1 == NS) ? 0 : wr + 1;
                PIPE_CLK(c_mma)
                yk_wait_vm_lgkm0<(NS - 2) * L>();                  // the next step's pieces have landed (this wave's) ...
                PIPE_CLK(c_wait)
                __builtin_amdgcn_s_barrier();                      // ... everybody's have, and everybody has read the stage refilled next
                asm volatile("" ::: "memory");
                PIPE_CLK(c_bar)
                read_frags(rd, 0, wf0, xf0);                       // under the tail of this step's MFMAs (after the last step: a dead stage)
            }
            yk_wait_vm_lgkm0<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            for (int kt = 0; kt < nk; ++kt) {
                yk_wait_vm_lgkm0<(NS - 2) * L>();
                PIPE_CLK(c_wait)
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                PIPE_CLK(c_bar)
                if (PS != 0 && late_dma) {
                    compute(rd);
                    PIPE_CLK(c_mma)
                    dma(wr);
                    PIPE_CLK(c_dma)
                } else {
                    dma(wr);
                    PIPE_CLK(c_dma)
                    compute(rd);
                }
                rd = (rd + 1 == NS) ? 0 : rd + 1;
                wr = (wr + 1 == NS) ? 0 : wr + 1;
                PIPE_CLK(c_mma)
            }
            yk_wait_vm_lgkm0<0>();                                 // drain the dead prefetches before LDS is reused by the epilogue
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
#ifdef YK_DEV
    const long long wall1 = a.dbg ? (long long)wall_clock64() : 0;
#endif
    igemm_epilogue<BM, BN, WM, WN, OUT, TM, TN>(a, acc, lds, m0, n0, tid, lane, wm, wn, vz);
#ifdef YK_DEV
    if (a.dbg && tid == 0) {
        long long *d = a.dbg + (size_t)L0 * 8;
        d[0] = wall0; d[1] = c_wait; d[2] = c_bar; d[3] = c_dma; d[4] = c_mma; d[5] = wall1; d[6] = (long long)wall_clock64(); d[7] = nk;
    }
#endif
#undef PIPE_CLK
#undef PIPE_CLK0
}

// The round-5 loop forms (IL: DMA pieces between the MFMAs; PS: phase-split waves; BR: weight fragments in registers, yk_igemm_br.h) were all
// measured equal or slower in the whole networks ("measured and rejected"): they exist in developer builds only (-DYK_DEV).
#ifdef YK_DEV
static bool yk_pipe_interleave() {
    const char *e = getenv("YK_PIPE_IL");
    return e && e[0] == '1';
}
static int yk_pipe_phase_split() {
    const char *e = getenv("YK_PIPE_PS");
    return (e && e[0] != '0') ? 1 : 0;
}
#endif

// more than 64 KB of dynamic LDS needs the attribute on EVERY kernel (instantiation) that is launched with it, on every device
static inline void yk_allow_lds(const void *kern, size_t bytes) {
    if (bytes <= 64 * 1024) return;
    static std::mutex mu;
    static std::set<std::pair<int, const void *>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (done.insert({dev, kern}).second) (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int BM, int BN, int WM, int WN, int NS>
static int launch_pipe(const igemm_args &a, hipStream_t st) {
#ifdef YK_DEV
    if constexpr (NS == 2 && BM * BN <= 128 * 128)
        if (!a.up0 && a.wfrag && yk_pipe_br()) return launch_br<BM, BN, WM, WN, NS>(a, st);     // weight fragments in registers (yk_igemm_br.h)
#endif
    constexpr size_t ring = (size_t)NS * (BM + BN) * 64 * 2, ct = (size_t)BM * (BN + 8) * 2;
    constexpr size_t ldsd = ring > ct ? ring : ct;
    dim3 g2((a.M + BM - 1) / BM, (a.N + BN - 1) / BN, a.split_k > 1 ? a.split_k : 1);
    auto go = [&](auto kern) {
        yk_allow_lds(reinterpret_cast<const void *>(kern), ldsd);
        hipLaunchKernelGGL(kern, g2, dim3(64 * WM * WN), ldsd, st, a);
    };
    if (a.up0) {
        if (a.split_k > 1) go(igemm_pipe_kernel<BM, BN, WM, WN, NS, 2, true>);
        else go(igemm_pipe_kernel<BM, BN, WM, WN, NS, 0, true>);
        return YK_OK;
    }
#ifdef YK_DEV
    const bool il = yk_pipe_interleave();
    const int ps = NS >= 3 ? yk_pipe_phase_split() : 0;
    if (ps && NS >= 3) {
        if constexpr (NS >= 3) {
            if (WM * WN >= 8) {
                if (a.split_k > 1) go(igemm_pipe_kernel<BM, BN, WM, WN, NS, 2, false, false, 1>);
                else go(igemm_pipe_kernel<BM, BN, WM, WN, NS, 0, false, false, 1>);
            } else {
                if (a.split_k > 1) go(igemm_pipe_kernel<BM, BN, WM, WN, NS, 2, false, false, 2>);
                else go(igemm_pipe_kernel<BM, BN, WM, WN, NS, 0, false, false, 2>);
            }
        }
        return YK_OK;
    }
    if (il) {
        if (a.split_k > 1) go(igemm_pipe_kernel<BM, BN, WM, WN, NS, 2, false, true>);
        else go(igemm_pipe_kernel<BM, BN, WM, WN, NS, 0, false, true>);
        return YK_OK;
    }
#endif
    if (a.split_k > 1) go(igemm_pipe_kernel<BM, BN, WM, WN, NS, 2, false>);
    else go(igemm_pipe_kernel<BM, BN, WM, WN, NS, 0, false>);
    return YK_OK;
}
