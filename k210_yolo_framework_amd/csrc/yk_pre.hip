// yk_pre.hip — GPU pre-processing immediately before the path:
// Helper._process_img's letterbox (tools/utils.py:378-399): scale = min(in_wh / img_wh),
// translation = ((in_wh - img_wh*scale)/2).astype(int), then
// skimage.transform.warp(img, AffineTransform(scale, translation).inverse, output_shape=in_hw, order=1,
//                        mode='constant', cval=0, preserve_range=True).astype('uint8')
// restated with skimage's own arithmetic (float64, one rounding per operation; this file is built with -ffp-contract=off):
//   M = inv(AffineTransform.params) = [[1/s, 0, -(tx*(1/s))], [0, 1/s, -(ty*(1/s))], [0, 0, 1]]   (numpy.linalg.inv of that matrix)
//   c = M00*x + M02, r = M11*y + M12                                   (_warps_cy._transform_metric)
//   minr/minc = floor, maxr/maxc = ceil, dr = r - minr, dc = c - minc   (interpolation.pxd bilinear_interpolation)
//   top = (1-dc)*I[minr,minc] + dc*I[minr,maxc]; bottom likewise on maxr; out = (1-dr)*top + dr*bottom; pixels outside -> cval 0
//   .astype('uint8') truncation.
// Pinned: tests/golden/letterbox_golden.npz holds outputs of the real scikit-image for seven source sizes; the kernel, the host
// mirror (helper.letterbox_bilinear) and the oracle reproduce them bit for bit.
// The per-image max normalisation that follows (utils.py:405) is fused into the stem conv (yk_run_u8).
#include "yk_common.h"

__global__ void __launch_bounds__(256) letterbox_u8_kernel(const uint8_t *__restrict__ src, int batch, int sh, int sw,
                                                           uint8_t *__restrict__ dst, int dh, int dw, double scale, int tx, int ty) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)batch * dh * dw;
    if (idx >= total) return;
    const int x = (int)(idx % dw), y = (int)((idx / dw) % dh), b = (int)(idx / ((size_t)dw * dh));
    const double inv = 1.0 / scale;
    const double c = inv * (double)x + (-((double)tx * inv)), r = inv * (double)y + (-((double)ty * inv));
    const double minc_f = floor(c), minr_f = floor(r);
    const int minc = (int)minc_f, minr = (int)minr_f, maxc = (int)ceil(c), maxr = (int)ceil(r);
    const double dc = c - minc_f, dr = r - minr_f;
    uint8_t *o = dst + idx * 3;
    const uint8_t *im = src + (size_t)b * sh * sw * 3;
    auto px = [&](int yy, int xx, int ch) -> double {
        return (yy >= 0 && yy < sh && xx >= 0 && xx < sw) ? (double)im[((size_t)yy * sw + xx) * 3 + ch] : 0.0;
    };
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const double top = (1.0 - dc) * px(minr, minc, ch) + dc * px(minr, maxc, ch);
        const double bottom = (1.0 - dc) * px(maxr, minc, ch) + dc * px(maxr, maxc, ch);
        o[ch] = (uint8_t)((1.0 - dr) * top + dr * bottom);    // astype('uint8'): truncation
    }
}

extern "C" int yk_letterbox_u8(const uint8_t *d_src, int batch, int src_h, int src_w, uint8_t *d_dst, int dst_h, int dst_w,
                               void *stream) {
    if (!d_src || !d_dst || batch <= 0 || src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0) {
        yk_set_error("yk_letterbox_u8: bad argument");
        return YK_ERR_ARG;
    }
    if (yk_current_device() < 0) {
        yk_set_error("yk_letterbox_u8: no HIP device");
        return YK_ERR_NO_DEVICE;
    }
    const double sx = (double)dst_w / (double)src_w, sy = (double)dst_h / (double)src_h;
    const double scale = sx < sy ? sx : sy;                                    // utils.py:381-382
    const int tx = (int)(((double)dst_w - (double)src_w * scale) / 2.0);       // .astype(int): truncation, utils.py:385
    const int ty = (int)(((double)dst_h - (double)src_h * scale) / 2.0);
    const size_t total = (size_t)batch * dst_h * dst_w;
    hipLaunchKernelGGL(letterbox_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_src, batch,
                       src_h, src_w, d_dst, dst_h, dst_w, scale, tx, ty);
    YK_HIP(hipGetLastError());
    return YK_OK;
}

// ---- `img / np.max(img)` (tools/utils.py:405) for a batch of u8 frames -> fp32, for the TRAINING input pipeline (the inference
// path fuses it into the stem conv).  numpy divides in float64 and the pipeline then casts to float32 (utils.py:436 py_function
// output type): one correctly rounded quotient per element.
__global__ void __launch_bounds__(1024) u8_image_max_kernel(const uint8_t *__restrict__ f, size_t per_image, unsigned *__restrict__ mx) {
    __shared__ unsigned part[16];
    const uint8_t *p = f + (size_t)blockIdx.x * per_image;
    unsigned m = 0;
    size_t done = 0;
    if ((((uintptr_t)p) & 15) == 0) {                              // 16 bytes per load (byte loads: 210 dependent-latency loads per thread, 28 us)
        const size_t nv = per_image / 16;
        const uint4 *v = reinterpret_cast<const uint4 *>(p);
        unsigned a = 0;                                           // byte-wise max of the four lanes of a dword, folded at the end
        for (size_t i = threadIdx.x; i < nv; i += 1024) {
            const uint4 q = v[i];
            const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // per-byte max of a and w[k]: compare byte lanes separately (no carries between them)
                const unsigned lo_a = a & 0x00ff00ffu, lo_w = w[k] & 0x00ff00ffu, hi_a = (a >> 8) & 0x00ff00ffu, hi_w = (w[k] >> 8) & 0x00ff00ffu;
                const unsigned lo = max(lo_a & 0xffffu, lo_w & 0xffffu) | (max(lo_a >> 16, lo_w >> 16) << 16);
                const unsigned hi = max(hi_a & 0xffffu, hi_w & 0xffffu) | (max(hi_a >> 16, hi_w >> 16) << 16);
                a = lo | (hi << 8);
            }
        }
        m = max(max(a & 255u, (a >> 8) & 255u), max((a >> 16) & 255u, a >> 24));
        done = nv * 16;
    }
    for (size_t i = done + threadIdx.x; i < per_image; i += 1024) m = max(m, (unsigned)p[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) m = max(m, part[i]);
        mx[blockIdx.x] = m;
    }
}
__global__ void __launch_bounds__(256) u8_normalise_kernel(const uint8_t *__restrict__ f, size_t per_image, const unsigned *__restrict__ mx,
                                                           float *__restrict__ out, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const unsigned m = mx[i / per_image];
    out[i] = (float)((double)f[i] / (double)m);                   // max 0 -> nan/inf exactly like numpy's 0/0 (an all-black image)
}
extern "C" int yk_normalise_u8(const uint8_t *d_frames, int batch, size_t per_image, float *d_out, void *stream) {
    if (!d_frames || !d_out || batch <= 0 || per_image == 0) {
        yk_set_error("yk_normalise_u8: bad argument");
        return YK_ERR_ARG;
    }
    const int dev = yk_current_device();
    if (dev < 0) {
        yk_set_error("yk_normalise_u8: no HIP device");
        return YK_ERR_NO_DEVICE;
    }
    unsigned *mx = (unsigned *)yk_scratch(dev, stream, 16, sizeof(unsigned) * (size_t)batch);
    if (!mx) return YK_ERR_NOMEM;
    hipLaunchKernelGGL(u8_image_max_kernel, dim3((unsigned)batch), dim3(1024), 0, (hipStream_t)stream, d_frames, per_image, mx);
    const size_t total = (size_t)batch * per_image;
    hipLaunchKernelGGL(u8_normalise_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_frames, per_image, mx,
                       d_out, total);
    YK_HIP(hipGetLastError());
    return YK_OK;
}
