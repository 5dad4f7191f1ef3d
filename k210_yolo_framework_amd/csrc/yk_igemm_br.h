// yk_igemm_br.h — the ring implicit-GEMM conv of yk_igemm_pipe.h with the WEIGHT fragments in registers (included by yk_conv.hip).
//
// Round 5 measured two things about the ring kernel: a wave is held ~100 cycles by the issue of every 1 KB LDS-DMA piece (its loop is bound
// by that issue, not by the tile - DESIGN.md 3d), and the fused blocks of the f16x2 mode gained 8-15 % when their weight tile left LDS
// (a fourth workgroup per CU).  The weight tile of a 64x128 step is 4 of the 6 pieces a wave issues and 8 of its 12 fragment reads, and it is
// shared by only WM waves.  Here each wave loads the fragments of ITS 16*TN output channels straight from global memory, one k-step ahead, into
// one of two register sets (the loop is unrolled by two so that the sets are named, not indexed); the ring holds the A tile only.  The weights
// are read from a copy in FRAGMENT order (igemm_args::wfrag: one contiguous KB per 16-channel block and half-step) - from the row-major
// W[N][K] a load instruction touches 16 rows x 64 bytes and the kernel runs at a third of the ring kernel's rate (233 vs 630 TFLOP/s):
//
//   step kt   s_waitcnt vmcnt((NS-2) * (A_IT + 2 TN)) ; s_barrier
//             weight fragments of step kt+1 -> the other register set ; A pieces of step kt+NS-1 -> stage (kt-1) % NS
//             MFMAs on stage kt % NS with this step's register set
//
// LDS per workgroup: NS x BM x 128 bytes (64x128, two stages: 16 KB instead of 48), so the register file sets the co-residency (96 VGPRs:
// five workgroups per CU).
//
// MEASURED (Darknet-53 f16, 32 images, tools/darknet_layers.py; parity green with it on: tests/test_gpu_layers.py, test_gpu_net.py):
// 6855 images/s of kernels against 7352 for the ring kernel, the 3x3 layers 545-557 against 601-607 TFLOP/s.  Every wave loading its own
// weights doubles the weight bytes through the CU's vector-memory path (40 KB per workgroup and k-step instead of 24): with five workgroups
// per CU that path (64 B/clk) is full.  What pays in the fused blocks (one wave per weight fragment, nothing shared) does not pay where
// two waves share a fragment.  OFF unless YK_PIPE_BR=1 (the plan then also builds the fragment-order copy).
#pragma once



template <int BM, int BN, int WM, int WN, int NS, int OUT>
__global__ void __launch_bounds__(64 * WM * WN) igemm_br_kernel(const igemm_args a) {
    constexpr int NW = WM * WN, BK = 64;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    static_assert((BM / 8) % NW == 0, "8-row DMA groups must divide among the waves");
    constexpr int A_IT = BM / 8 / NW, L = A_IT + 2 * TN;          // vector-memory instructions per wave and step
    constexpr int STAGE = BM * BK;                                // halfs: the A tile only
    static_assert(NS == 2, "two stages (a deeper ring needs its own prologue wait: the weight loads sit in front of the A pieces)");
    yk_half *lds = reinterpret_cast<yk_half *>(yk_smem);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
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
    const int rr = lane >> 3, gc = (lane & 7) ^ rr;
    const int fr = lane & 15, sw = fr & 7, fq = lane >> 4;

    uint32_t P0[A_IT], P1[A_IT], rmask[A_IT], wfo[TN];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int m = m0 + (wid + it * NW) * 8 + rr;
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
    // this lane's piece of the weight fragment of n-block j: row n0 + (wn*TN + j)*16 + fr of W[N][K], 16 bytes at chunk fq of a half-step
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nb = (n0 >> 4) + wn * TN + j;                   // 16-channel block; fragment order: 1 KB per block and half-step, lane-major
        wfo[j] = (nb < a.nb16) ? (uint32_t)nb * 2048u + (uint32_t)lane * 16u : YK_OOB;
    }
    const uint32_t wstep = (uint32_t)a.nb16 * 2048u;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void *)a.in0, 0, a.in0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void *)(a.in1 ? a.in1 : a.in0), 0, a.in1 ? a.in1_bytes : a.in0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void *)a.wfrag, 0, a.wfrag_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void *lds_ptr_t;

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
    auto dma = [&](int stage) {                                   // the A pieces of walk step `step` (dead steps deposit zeros)
        const bool live = step < lim;
        const uint32_t cs = live ? (uint32_t)cin * 2u : YK_OOB;
        const bool second = cin >= a.c0p;
        yk_half *As = lds + stage * STAGE;
#pragma unroll
        for (int n = 0; n < A_IT; ++n) {
            lds_ptr_t dsta = (lds_ptr_t)(As + (wid + n * NW) * 8 * BK);
            // (named locals: with the sum written inline hipcc's HOST pass silently drops the kernel's stub)
            const uint32_t off1 = aoff1[n] + cs, off0 = aoff0[n] + cs;
            if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dsta, 16, off1, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dsta, 16, off0, 0, 0, 0);
        }
        ++step;
        cin += BK;
        if (cin >= Ctp) {
            cin = 0;
            ++tap;
            retap();
        }
    };
    int bstep = kt0;                                               // the k-step whose weight fragments are loaded next
    auto load_w = [&](half8 (&w)[2][TN]) {
        const uint32_t so = (bstep < lim) ? (uint32_t)bstep * wstep : YK_OOB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const uint32_t offw = wfo[j] + so + (uint32_t)ks * 1024u;
                w[ks][j] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsw, offw, 0, 0));
            }
        ++bstep;
    };
    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int stage, const half8 (&w)[2][TN]) {
        const yk_half *As = lds + stage * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = ((ks * 4 + fq) ^ sw) * 8;
            half8 xf[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) xf[i] = *reinterpret_cast<const half8 *>(As + ((wm * TM + i) * 16 + fr) * BK + ch);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ks][j], xf[i], acc[i][j], 0, 0, 0);
        }
    };
    half8 wa[2][TN], wb[2][TN];
    if (nk > 0) {
        load_w(wa);                                               // step 0's fragments, then the A pieces of steps 0 .. NS-2
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) dma(s);
        int rd = 0, wr = NS - 1;
        const int nk2 = (nk + 1) & ~1;                            // an odd walk ends on a dead step: zeros against zeros
        for (int kt = 0; kt < nk2; kt += 2) {
            yk_wait_vm_lgkm0<(NS - 2) * L>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            load_w(wb);
            dma(wr);
            compute(rd, wa);
            rd = (rd + 1 == NS) ? 0 : rd + 1;
            wr = (wr + 1 == NS) ? 0 : wr + 1;
            yk_wait_vm_lgkm0<(NS - 2) * L>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            load_w(wa);
            dma(wr);
            compute(rd, wb);
            rd = (rd + 1 == NS) ? 0 : rd + 1;
            wr = (wr + 1 == NS) ? 0 : wr + 1;
        }
        yk_wait_vm_lgkm0<0>();                                     // drain the dead prefetches before LDS is reused by the epilogue
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    igemm_epilogue<BM, BN, WM, WN, OUT, TM, TN>(a, acc, lds, m0, n0, tid, lane, wm, wn, vz);
}

// off unless YK_PIPE_BR=1 (until measured)
static bool yk_pipe_br() {
#ifdef YK_DEV
    const char *e = getenv("YK_PIPE_BR");
    return e && e[0] == '1';
#else
    static const bool on = yk_env_flag("YK_PIPE_BR", false);
    return on;
#endif
}

template <int BM, int BN, int WM, int WN, int NS>
static int launch_br(const igemm_args &a, hipStream_t st) {
    constexpr size_t ring = (size_t)NS * BM * 64 * 2, ct = (size_t)BM * (BN + 8) * 2;
    constexpr size_t ldsd = ring > ct ? ring : ct;
    dim3 g2((a.M + BM - 1) / BM, (a.N + BN - 1) / BN, a.split_k > 1 ? a.split_k : 1);
    if (a.split_k > 1) hipLaunchKernelGGL((igemm_br_kernel<BM, BN, WM, WN, NS, 2>), g2, dim3(64 * WM * WN), ldsd, st, a);
    else hipLaunchKernelGGL((igemm_br_kernel<BM, BN, WM, WN, NS, 0>), g2, dim3(64 * WM * WN), ldsd, st, a);
    return YK_OK;
}
