// Dev probe (not part of the product): what the LDS-DMA operand path of one CU delivers, by access pattern, ring depth and workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/dma_probe.hip -o tools/dma_probe.bin && tools/dma_probe.bin
// A "piece" = one `buffer_load_dwordx4 ... lds` wave-instruction (1 KB).  Every wave issues 4 pieces per step into an NS-deep ring and waits
// like xg_kernel does (counted vmcnt, one s_barrier per step); nothing is computed.
//   pattern 0  linear: a piece is 1 KB contiguous (the weights' form; an activation tensor stored in MFMA tile order would look like this)
//   pattern 1  round 3's pixel operand: 16 rows x 64 B per piece, the 64 B as four 16-byte chunks at stride 32 (hi OR lo halves), row pitch 1536 B
//   pattern 2  8 rows x 128 B (whole lines), row pitch 1536 B
//   pattern 3  16 rows x 64 B contiguous (what a [pixel][hi x32 | lo x32] layout would give)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void *lds_ptr_t;
extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

template <int PAT, int NS, int NWAVE = 4, bool BAR = true>
__global__ void __launch_bounds__(64 * NWAVE) probe(const uint8_t *src, uint32_t bytes, int rows, int steps, int ksteps) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, bytes, 0x00020000);
    const uint32_t pitch = (uint32_t)ksteps * 128u;
    const uint32_t row0 = (uint32_t)((blockIdx.x * 64) % (rows - 256)) + wid * 16;          // this wave's 16 rows
    uint32_t o[4];
    if (PAT == 0) {
        for (int i = 0; i < 4; ++i) o[i] = (row0 * pitch) + i * 1024u + lane * 16u;         // + step * 4096
    } else if (PAT == 1) {
        const uint32_t r = lane >> 2, c = lane & 3;
        for (int i = 0; i < 4; ++i) o[i] = (row0 + r) * pitch + c * 32u + (i & 1) * 16u + (i >> 1) * 0u;
    } else if (PAT == 2) {
        const uint32_t r = (lane >> 3) + 0, c = lane & 7;
        for (int i = 0; i < 4; ++i) o[i] = (row0 + (i & 1) * 8 + r) * pitch + c * 16u;
    } else {
        const uint32_t r = lane >> 2, c = lane & 3;
        for (int i = 0; i < 4; ++i) o[i] = (row0 + r) * pitch + (i & 1) * 64u + c * 16u;
    }
    auto dma = [&](int stage, int step) {
        unsigned char *S = sm + stage * (4096 * NWAVE) + wid * 4096;
        const int k = step % ksteps;
        const uint32_t ko = (PAT == 0) ? (uint32_t)k * 4096u : (uint32_t)k * 128u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // patterns 1-3 fetch the step's 128-byte line of 16 rows twice over (two pieces per 16 rows = A operand), then the same again as "B"
            const uint32_t off = o[i] + ko + ((PAT != 0 && i >= 2) ? 8u * pitch * 0u : 0u);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(S + i * 1024), 16, off, 0, 0, 0);
        }
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) dma(s, s);
    int wr = NS - 1;
    for (int t = 0; t < steps; ++t) {
        wait_vm<(NS - 2) * 4>();
        if (BAR) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        dma(wr, t + NS - 1);
        wr = (wr + 1 == NS) ? 0 : wr + 1;
    }
    wait_vm<0>();
}

template <int PAT, int NS, int NWAVE = 4, bool BAR = true>
static void run(const uint8_t *d, uint32_t bytes, int rows, int ksteps, int wg_per_cu) {
    const int steps = 240, grid = 256 * wg_per_cu;
    const unsigned lds = NS * 4096 * NWAVE;
    if (lds * wg_per_cu > 160 * 1024) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(probe<PAT, NS, NWAVE, BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<PAT, NS, NWAVE, BAR>), dim3(grid), dim3(64 * NWAVE), lds, 0, d, bytes, rows, steps, ksteps);
    hipEventRecord(e0, 0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((probe<PAT, NS, NWAVE, BAR>), dim3(grid), dim3(64 * NWAVE), lds, 0, d, bytes, rows, steps, ksteps);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, kb = (double)grid * NWAVE * 4 * (steps + NS - 1);
    printf("waves %d bar %d  pat %d  ring %d  wg/cu %d : %7.1f us  %6.2f us/step  %6.1f GB/s/CU  %5.2f TB/s chip\n", NWAVE, (int)BAR, PAT, NS, wg_per_cu, us, us / steps, kb * 1024 / 256 / us / 1e3,
           kb * 1024 / us / 1e6);
}

int main() {
    const int rows = 8960, ksteps = 12;                      // the 14x20x384 tensor of 32 images: 13.8 MB
    const uint32_t bytes = (uint32_t)rows * ksteps * 128;
    uint8_t *d;
    hipMalloc(&d, bytes + 65536);
    hipMemset(d, 1, bytes);
#define ROW(P)                                   \
    for (int w = 1; w <= 4; w *= 2) {            \
        run<P, 2>(d, bytes, rows, ksteps, w);    \
        run<P, 3>(d, bytes, rows, ksteps, w);    \
        run<P, 4>(d, bytes, rows, ksteps, w);    \
        run<P, 6>(d, bytes, rows, ksteps, w);    \
        run<P, 8>(d, bytes, rows, ksteps, w);    \
    }
    ROW(1)
    // more waves in ONE workgroup per CU, and the same without the per-step barrier
    run<1, 3, 8, true>(d, bytes, rows, ksteps, 1);
    run<1, 3, 16, true>(d, bytes, rows, ksteps, 1);
    run<1, 3, 4, false>(d, bytes, rows, ksteps, 1);
    run<1, 3, 4, false>(d, bytes, rows, ksteps, 2);
    run<1, 3, 4, false>(d, bytes, rows, ksteps, 4);
    run<1, 3, 8, false>(d, bytes, rows, ksteps, 1);
    run<1, 3, 2, true>(d, bytes, rows, ksteps, 1);
    run<1, 3, 2, true>(d, bytes, rows, ksteps, 2);
    run<1, 3, 2, true>(d, bytes, rows, ksteps, 4);
    run<1, 3, 1, true>(d, bytes, rows, ksteps, 4);
    run<1, 3, 1, true>(d, bytes, rows, ksteps, 8);
    return 0;
}
