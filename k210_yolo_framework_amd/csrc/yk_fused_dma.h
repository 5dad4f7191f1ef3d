// yk_fused_dma.h — DepthwiseConv2D 3x3 + BN + act -> Conv2D 1x1 + BN + act in one launch, input tile staged by LDS-DMA
// (included by yk_conv.hip).  MobileNet block: keras_mobilenet.py:359-436, keras_mobilenet_v2.py:452-481.
//
// What was wrong with the register-staged fused kernels on the small-spatial layers (14x20 .. 28x40 at batch 32: one or two
// workgroup rounds on 256 CUs): the depthwise phase was a CHAIN of L2 round trips - a thread issued the 9 tap loads of one item,
// waited ~1 us, computed, issued the next item's ... five times in a row (phase stamps: 5 us of a 15 us launch), and a 192-column
// split made two workgroups compute every depthwise pixel twice.  Here:
//   * a workgroup owns a TR x TC patch of output pixels of ONE image and ALL output channels;
//   * the patch's input halo ((TR-1)*s+3 rows x (TC-1)*s+3 columns x Cin) is fetched in ONE burst of `buffer_load ... lds`
//     (every lane computes the source address of the 16 bytes that belong at its linear LDS position; pixels outside the image get
//     an out-of-range offset and arrive as zeros - Keras zero padding costs nothing, the depthwise loop has no edge cases);
//   * the pointwise weight panel of each wave (NPW 16-channel slices x the whole K) is requested into registers BEFORE that burst is
//     waited for, so both L2 streams fly together and nothing else on the workgroup's path touches global memory until the stores;
//   * depthwise from LDS (v_fma_mix_f32, fp32 accumulate, result rounded to fp16 exactly like the unfused pipeline's stored tensor),
//     pointwise on v_mfma_f32_16x16x32_f16 from LDS + registers, epilogue through LDS with 16-byte row-contiguous stores.
// 12 waves: wave w owns output channels [w*NPW*16, (w+1)*NPW*16).
#pragma once

struct fdma_geom {
    int TR, TC, tiles_x, tiles_y, in_rows, in_cols;      // patch geometry (set by the launcher)
    yk_fastdiv fd_tpi, fd_tx, fd_tc, fd_rowu, fd_g;      // / tiles per image, / tiles_x, / TC, / (in_cols*G), / G
};

template <int NPW, int KS, int MT>
__global__ void __launch_bounds__(768) fused_dma_kernel(const igemm_args a, const fdma_geom t) {
    constexpr int NT = 768;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int Cp = a.c0p, G = Cp >> 3, Kp = (Cp + 31) & ~31, LDA = Kp + 16;
    const int nk = Kp >> 5;
    const int s = a.dw_stride;
    const int in16 = t.in_rows * t.in_cols * G;                              // 16-byte pieces of the input patch
    const int in16p = (in16 + 63) & ~63;
    yk_half *IN = reinterpret_cast<yk_half *>(yk_smem);
    yk_half *Wd = IN + (size_t)in16p * 8;                                     // depthwise weights [9][Cp]
    yk_half *As = Wd + 9 * Cp;                                                // depthwise result [MT*16][LDA]
    const int BM = t.TR * t.TC;

    const int n_sl0 = blockIdx.y * 12 * NPW;                                  // first 16-channel slice of this workgroup
    const int tile = yk_xcd_tile(blockIdx.x, gridDim.x);
    const int b = (int)yk_div(tile, t.fd_tpi), tl = tile - b * (t.tiles_x * t.tiles_y);
    const int ty = (int)yk_div(tl, t.fd_tx), tx = tl - ty * t.tiles_x;
    const int oy0 = ty * t.TR, ox0 = tx * t.TC;
    const int iy0 = oy0 * s - a.dw_pad_t, ix0 = ox0 * s - a.dw_pad_l;
    const int fr = lane & 15, fk = (lane >> 4) * 8;
#define YK_STAMP(k) \
    if (a.dbg && tid == 0) a.dbg[(size_t)blockIdx.x * 8 + (k)] = (long long)wall_clock64();
    YK_STAMP(0)

    // (2) depthwise weights (registers for now) and the input patch -> LDS by DMA, one burst.  These are requested FIRST: the
    //     depthwise phase needs only them, and loads retire in order, so `vmcnt(NPW*KS)` below means "patch landed" while the
    //     pointwise panel (requested next) keeps streaming under the depthwise arithmetic.
    half8 dwv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (tid < 9 * G) dwv = *reinterpret_cast<const half8 *>(a.dw_w + (size_t)tid * 8);
    {
        const uint32_t img_bytes = (uint32_t)a.dw_Hi * a.dw_Wi * Cp * 2u;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(a.in0 + (size_t)b * a.dw_Hi * a.dw_Wi * Cp), 0, img_bytes, 0x00020000);
        typedef __attribute__((address_space(3))) void *lds_ptr_t;
        const int rowu = t.in_cols * G;
        for (int q0 = wid * 64; q0 < in16p; q0 += NT) {
            const int q = q0 + lane;
            const uint32_t r = yk_div(q, t.fd_rowu), rem = q - r * rowu;
            const uint32_t c = yk_div(rem, t.fd_g), g = rem - c * G;
            const int iy = iy0 + (int)r, ix = ix0 + (int)c;
            const bool ok = q < in16 && (unsigned)iy < (unsigned)a.dw_Hi && (unsigned)ix < (unsigned)a.dw_Wi;
            const uint32_t off = ok ? (uint32_t)((iy * a.dw_Wi + ix) * Cp + (int)g * 8) * 2u : YK_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(IN + (size_t)q0 * 8), 16, off, 0, 0, 0);
        }
    }
    // (2b) pointwise weight panel -> registers (in flight under the depthwise phase).  The panel is stored in
    //     MFMA fragment order (yk_engine.hip): a wave-load is 1 KB contiguous.  Workgroups start at different k-steps so that 256 CUs
    //     do not hammer the same L2 lines in the same microsecond.
    half8 wq[NPW][KS];
    const int ns_total = (a.N + 15) >> 4;
    const int rot = tl % nk;                                                  // a function of the patch's place in ITS image only
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        const int sl = min(n_sl0 + wid * NPW + j, ns_total - 1);         // unconditional loads (a branch per load makes hipcc wait per
#pragma unroll                                                            // load): slices past N and k-steps past nk fetch a valid
        for (int k = 0; k < KS; ++k) {                                    // address whose value is never used
            int kk = min(k, nk - 1) + rot;
            kk = kk >= nk ? kk - nk : kk;
            wq[j][k] = *reinterpret_cast<const half8 *>(a.w + (((size_t)sl * nk + kk) * 64 + lane) * 8);
        }
    }
    // (3) depthwise weights -> LDS (the register copy is written out after the big wait: a store here would drain the load queue),
    //     zero the K padding of the A tile
    {
        const int padv = (Kp - Cp) >> 3;
        for (int v = tid; v < MT * 16 * padv; v += NT) {
            const int p = v / padv, c = v - p * padv;
            *reinterpret_cast<half8 *>(As + p * LDA + Cp + c * 8) = half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    // this thread's depthwise channel octet is the same for all of its items (768 % G == 0)
    const int g = tid % G, p0 = tid / G, PP = NT / G;
    const float4 sc0 = *reinterpret_cast<const float4 *>(a.dw_scale + g * 8), sc1 = *reinterpret_cast<const float4 *>(a.dw_scale + g * 8 + 4);
    const float4 bs0 = *reinterpret_cast<const float4 *>(a.dw_bias + g * 8), bs1 = *reinterpret_cast<const float4 *>(a.dw_bias + g * 8 + 4);
    const bool dcap = a.dw_cap < 3.0e38f;
    YK_STAMP(1)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW * KS) : "memory");          // patch + depthwise weights landed; panel still in flight
    if (tid < 9 * G) *reinterpret_cast<half8 *>(Wd + tid * 8) = dwv;
    __syncthreads();
    YK_STAMP(2)

    // (4) depthwise from LDS
    for (int p = p0; p < MT * 16; p += PP) {
        half8 h = {0, 0, 0, 0, 0, 0, 0, 0};
        if (p < BM) {
            const int py = (int)yk_div(p, t.fd_tc), px = p - py * t.TC;
            const yk_half *src = IN + ((size_t)(py * s * t.in_cols + px * s) * G + g) * 8;
            // one filter row at a time: 3 taps + 3 weight vectors live (the pointwise panel already holds up to 96 registers)
            float d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                u32x4 x[3], w[3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    x[kx] = *reinterpret_cast<const u32x4 *>(src + (size_t)(ky * t.in_cols + kx) * Cp);
                    w[kx] = *reinterpret_cast<const u32x4 *>(Wd + (size_t)(ky * 3 + kx) * Cp + g * 8);
                }
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        fma_mix_lo(d[2 * j], x[kx][j], w[kx][j]);
                        fma_mix_hi(d[2 * j + 1], x[kx][j], w[kx][j]);
                    }
            }
            const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w}, bs[8] = {bs0.x, bs0.y, bs0.z, bs0.w, bs1.x, bs1.y, bs1.z, bs1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = d[e] * sc[e] + bs[e];
                h[e] = (yk_half)(dcap ? yk_act2<true>(v, a.dw_slope, a.dw_cap) : yk_act2<false>(v, a.dw_slope, a.dw_cap));
            }
        }
        *reinterpret_cast<half8 *>(As + p * LDA + g * 8) = h;
    }
    __syncthreads();
    YK_STAMP(3)

    // (5) pointwise GEMM: A fragments from LDS, weights from registers
    floatx4 acc[MT][NPW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NPW; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KS; ++k) {
        if (k < nk) {
            half8 xf[MT];
            int kk = k + rot;
            kk = kk >= nk ? kk - nk : kk;
#pragma unroll
            for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const half8 *>(As + (i * 16 + fr) * LDA + kk * 32 + fk);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NPW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[j][k], xf[i], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();                                                          // A tile no longer needed: LDS becomes the output tile
    YK_STAMP(4)

    // (6) epilogue through LDS: C[BM][N] fp16, then 16-byte row-contiguous stores
    yk_half *Cs = reinterpret_cast<yk_half *>(yk_smem);
    const int ncols = min(a.outp - n_sl0 * 16, 12 * NPW * 16);                  // output channels of this workgroup
    const int CS_LD = ncols + 8;
    const int nl4 = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        const int nloc = (wid * NPW + j) * 16 + nl4, n = n_sl0 * 16 + nloc;
        if (n >= a.outp) continue;
        const float4 sc = *reinterpret_cast<const float4 *>(a.scale + n), bs = *reinterpret_cast<const float4 *>(a.bias + n);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int p = i * 16 + fr;
            const half4 h = {(yk_half)yk_actf(acc[i][j][0] * sc.x + bs.x, a.slope, a.cap), (yk_half)yk_actf(acc[i][j][1] * sc.y + bs.y, a.slope, a.cap),
                             (yk_half)yk_actf(acc[i][j][2] * sc.z + bs.z, a.slope, a.cap), (yk_half)yk_actf(acc[i][j][3] * sc.w + bs.w, a.slope, a.cap)};
            *reinterpret_cast<half4 *>(Cs + p * CS_LD + nloc) = h;
        }
    }
    __syncthreads();
    YK_STAMP(5)
    const int VPR = ncols >> 3;
    yk_half *o = reinterpret_cast<yk_half *>(a.out) + n_sl0 * 16;
    const yk_half *resp = a.res ? a.res + n_sl0 * 16 : nullptr;
    for (int v = tid; v < BM * VPR; v += NT) {
        const int p = v / VPR, cv = v - p * VPR;
        const int py = (int)yk_div(p, t.fd_tc), px = p - py * t.TC;
        const int oy = oy0 + py, ox = ox0 + px;
        if (oy < a.Ho && ox < a.Wo) {
            const size_t m = ((size_t)b * a.Ho + oy) * a.Wo + ox;
            half8 hv = *reinterpret_cast<const half8 *>(Cs + p * CS_LD + cv * 8);
            if (resp) {
                const half8 r = *reinterpret_cast<const half8 *>(resp + m * a.resp + cv * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = (yk_half)((float)hv[e] + (float)r[e]);
            }
            *reinterpret_cast<half8 *>(o + m * a.outp + cv * 8) = hv;
        }
    }
    YK_STAMP(6)
    if (a.dbg && tid == 0)
        a.dbg[(size_t)blockIdx.x * 8 + 7] = ((long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);
#undef YK_STAMP
}

struct fdma_plan {
    int npw, ks, mt, TR, TC, nsplit;
    size_t lds;
    long rounds;
};
// Patch geometry for a layer, or npw = 0 when this kernel does not apply.  Candidates: patch TR x TC (<= 80 pixels = 5 MFMA row tiles)
// and a split of the output channels over 1, 2 or 4 workgroups (which then each repeat the depthwise work of the patch).  The cost
// model is the measured phase structure of the kernel (tools/phase.py): operand bytes through L2 at ~24 B/clk per CU with every CU
// pulling the same weight panel, ~1800 cycles per depthwise pass of 768 items, the GEMM and a fixed epilogue, times the number of
// workgroup rounds on 256 CUs.
static fdma_plan yk_fdma_plan(const igemm_args &a) {
    fdma_plan r = {0, 0, 0, 0, 0, 1, 0, 0};
    const int Cp = a.c0p, G = Cp >> 3, Kp = (Cp + 31) & ~31;
    if (G <= 0 || 768 % G != 0 || a.outp % 8 != 0 || a.in0_bytes >= YK_OOB) return r;
    const int nt = (a.N + 15) / 16, nk = Kp / 32;
    if (nk > 12 || (a.resp && a.resp != a.outp)) return r;
    const int B = a.M / (a.Ho * a.Wo);
    long best_cost = -1;
    for (int ns = 1; ns <= 4; ns *= 2) {
        const int spw = (nt + ns - 1) / ns;                  // 16-channel slices per workgroup
        const int npw = (spw + 11) / 12;
        if (npw > 2 || (ns > 1 && (spw * 16) % 8 != 0)) continue;
        for (int tc = std::min(a.Wo, 80); tc >= 8; --tc) {
            if (a.Wo % tc != 0) continue;
            for (int tr = 1; tr * tc <= 80 && tr <= a.Ho; ++tr) {
                const int bm = tr * tc, mt = (bm + 15) / 16;
                if (mt > 5 || (npw == 2 && mt > 3)) continue;
                const int in_rows = (tr - 1) * a.dw_stride + 3, in_cols = (tc - 1) * a.dw_stride + 3;
                const size_t in16p = ((size_t)in_rows * in_cols * G + 63) & ~(size_t)63;
                const size_t lds = in16p * 16 + (size_t)9 * Cp * 2 + (size_t)mt * 16 * (Kp + 16) * 2;
                const size_t cs = (size_t)(mt <= 3 ? 48 : 80) * (spw * 16 + 8) * 2;
                if (std::max(lds, cs) > 150 * 1024) continue;
                const long tiles = (long)B * ((a.Ho + tr - 1) / tr) * (a.Wo / tc) * ns;
                const long rounds = (tiles + 255) / 256;
                const long bytes = (long)spw * 16 * Kp * 2 + (long)in16p * 16;
                const long cyc = bytes / 24 + ((long)mt * 16 * G + 767) / 768 * 1800 + (long)nk * mt * npw * 20 + 2500;
                const long cost = rounds * cyc;
                if (best_cost < 0 || cost < best_cost) {
                    best_cost = cost;
                    r = {npw, nk <= 6 ? 6 : 12, mt <= 3 ? 3 : 5, tr, tc, ns, std::max(lds, cs), rounds};
                }
            }
        }
    }
    return r;
}

template <int NPW, int KS, int MT>
static int launch_fdma_t(const igemm_args &a, const fdma_plan &pl, hipStream_t st) {
    fdma_geom t;
    t.TR = pl.TR;
    t.TC = pl.TC;
    t.tiles_x = (a.Wo + pl.TC - 1) / pl.TC;
    t.tiles_y = (a.Ho + pl.TR - 1) / pl.TR;
    t.in_rows = (pl.TR - 1) * a.dw_stride + 3;
    t.in_cols = (pl.TC - 1) * a.dw_stride + 3;
    const int G = a.c0p >> 3;
    t.fd_tpi = yk_make_fastdiv((uint32_t)(t.tiles_x * t.tiles_y));
    t.fd_tx = yk_make_fastdiv((uint32_t)t.tiles_x);
    t.fd_tc = yk_make_fastdiv((uint32_t)pl.TC);
    t.fd_rowu = yk_make_fastdiv((uint32_t)(t.in_cols * G));
    t.fd_g = yk_make_fastdiv((uint32_t)G);
    const int B = a.M / (a.Ho * a.Wo);
    static size_t attr_lds = 64 * 1024;
    if (pl.lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fused_dma_kernel<NPW, KS, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        attr_lds = 160 * 1024;
    }
    hipLaunchKernelGGL((fused_dma_kernel<NPW, KS, MT>), dim3((unsigned)(B * t.tiles_x * t.tiles_y), pl.nsplit), dim3(768), pl.lds, st, a, t);
    return YK_OK;
}

void yk_fdma_fill(igemm_args &a) {
    const fdma_plan pl = yk_fdma_plan(a);
    a.fp_npw = pl.npw; a.fp_ks = pl.ks; a.fp_mt = pl.mt; a.fp_tr = pl.TR; a.fp_tc = pl.TC; a.fp_ns = pl.nsplit;
    a.fp_lds = (unsigned)pl.lds;
}

static int yk_launch_fdma(const igemm_args &a, hipStream_t st) {
    const fdma_plan pl = {a.fp_npw, a.fp_ks, a.fp_mt, a.fp_tr, a.fp_tc, a.fp_ns, a.fp_lds, 0};
    if (!pl.npw) {
        yk_set_error("fused_dma: layer does not fit");
        return YK_ERR_UNSUPPORTED;
    }
    if (pl.npw == 1 && pl.ks == 6 && pl.mt == 3) return launch_fdma_t<1, 6, 3>(a, pl, st);
    if (pl.npw == 1 && pl.ks == 6 && pl.mt == 5) return launch_fdma_t<1, 6, 5>(a, pl, st);
    if (pl.npw == 1 && pl.ks == 12 && pl.mt == 3) return launch_fdma_t<1, 12, 3>(a, pl, st);
    if (pl.npw == 1 && pl.ks == 12 && pl.mt == 5) return launch_fdma_t<1, 12, 5>(a, pl, st);
    if (pl.npw == 2 && pl.ks == 6) return launch_fdma_t<2, 6, 3>(a, pl, st);
    return launch_fdma_t<2, 12, 3>(a, pl, st);
}

