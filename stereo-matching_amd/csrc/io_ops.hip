// Steps immediately before and after the hot path: gray conversion of the input images
// (stereo_io.py:47-75) and the accuracy metrics of a disparity map (stereo.py:532-577).
#include "common.hpp"

#pragma clang fp contract(off)

namespace shp {

// gray = float32(0.299)*R + float32(0.587)*G + float32(0.114)*B, evaluated left to right in float32 with a
// rounding after every operation (the census transform compares with strict '>', so the last bit matters);
// optional division by float32(scale) and clip to [0, 1].
template <typename T>
__global__ void to_gray_kernel(const T *__restrict__ img, size_t npix, int channels, int normalize, float scale,
                               float *__restrict__ out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        float g;
        if (channels == 1) {
            g = (float)img[i];
        } else {
            const T *p = img + i * channels;
            float r = (float)p[0], gg = (float)p[1], b = (float)p[2];
            float t0 = 0.299f * r;
            float t1 = 0.587f * gg;
            float t2 = 0.114f * b;
            g = t0 + t1;
            g = g + t2;
        }
        if (normalize) {
            g = g / scale;
            g = (g != g) ? g : fminf(fmaxf(g, 0.0f), 1.0f);  // np.clip keeps NaN
        }
        out[i] = g;
    }
}

static inline int ew_grid(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

int launch_to_gray(const void *d_img, int dtype, size_t npix, int channels, int normalize, float scale, float *d_out,
                   hipStream_t s) {
    if (npix == 0) return STEREO_OK;
    dim3 g(ew_grid(npix)), b(256);
    switch (dtype) {
        case STEREO_DT_U8:
            hipLaunchKernelGGL(to_gray_kernel<uint8_t>, g, b, 0, s, (const uint8_t *)d_img, npix, channels, normalize, scale, d_out);
            break;
        case STEREO_DT_U16:
            hipLaunchKernelGGL(to_gray_kernel<uint16_t>, g, b, 0, s, (const uint16_t *)d_img, npix, channels, normalize, scale, d_out);
            break;
        case STEREO_DT_F32:
            hipLaunchKernelGGL(to_gray_kernel<float>, g, b, 0, s, (const float *)d_img, npix, channels, normalize, scale, d_out);
            break;
        default: set_error("to_gray takes uint8, uint16 or float32 pixels"); return STEREO_EINVAL;
    }
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}

// nanmin / nanmax of a float32 buffer (stereo_io.py:38-39), as order-preserving integers.
__global__ void nan_minmax_kernel(const float *__restrict__ v, size_t n, uint32_t *__restrict__ mm) {
    uint32_t lo = 0xffffffffu, hi = 0u;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float x = v[i];
        if (x == x) {
            uint32_t k = f32_to_ordered(x);
            lo = k < lo ? k : lo;
            hi = k > hi ? k : hi;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t lo2 = __shfl_down(lo, off, 64), hi2 = __shfl_down(hi, off, 64);
        lo = lo2 < lo ? lo2 : lo;
        hi = hi2 > hi ? hi2 : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        if (lo < mm[0]) atomicMin(&mm[0], lo);
        if (hi > mm[1]) atomicMax(&mm[1], hi);
    }
}

int launch_nan_minmax(const float *d_v, size_t n, uint32_t *d_mm, hipStream_t s) {
    uint32_t init[2] = {0xffffffffu, 0u};
    SHP_HIP(hipMemcpyAsync(d_mm, init, 8, hipMemcpyHostToDevice, s));
    if (n) {
        hipLaunchKernelGGL(nan_minmax_kernel, dim3(ew_grid(n)), dim3(256), 0, s, d_v, n, d_mm);
        SHP_LAUNCH_CHECK();
    }
    return STEREO_OK;
}

// PBM / RMS accumulators over pixels with finite ground truth and a set mask:
// acc[0] = valid pixels, acc[1] = pixels with |d - gt| > threshold, acc_sq = sum of squared differences.
__global__ void metrics_kernel(const int32_t *__restrict__ disp, const float *__restrict__ gt,
                               const uint8_t *__restrict__ mask, size_t n, float threshold,
                               unsigned long long *__restrict__ acc, double *__restrict__ acc_sq) {
    unsigned long long valid = 0, bad = 0;
    double sq = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float g = gt[i];
        if (mask[i] && fabsf(g) <= 3.402823466e38f) {
            float diff = fabsf((float)disp[i] - g);
            valid += 1;
            bad += (diff > threshold) ? 1 : 0;
            float d2 = diff * diff;  // float32 square as the reference
            sq += (double)d2;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        valid += __shfl_down(valid, off, 64);
        bad += __shfl_down(bad, off, 64);
        sq += __shfl_down(sq, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&acc[0], valid);
        atomicAdd(&acc[1], bad);
        atomicAdd(acc_sq, sq);
    }
}

int launch_metrics(const int32_t *d_disp, const float *d_gt, const uint8_t *d_mask, size_t n, float threshold,
                   unsigned long long *d_acc, double *d_acc_sq, hipStream_t s) {
    SHP_HIP(hipMemsetAsync(d_acc, 0, 16, s));
    SHP_HIP(hipMemsetAsync(d_acc_sq, 0, 8, s));
    if (n) {
        hipLaunchKernelGGL(metrics_kernel, dim3(ew_grid(n)), dim3(256), 0, s, d_disp, d_gt, d_mask, n, threshold, d_acc,
                           d_acc_sq);
        SHP_LAUNCH_CHECK();
    }
    return STEREO_OK;
}

}  // namespace shp
