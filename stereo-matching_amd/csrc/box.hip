// Summed-area tables and clipped-window box means in float64 (guided_filter.py:21-179), plus the
// layer-at-a-time guided filter built from them.  This is the general path: any radius, any
// window index arrays.  The fused kernel in guided.hip is the fast path for radius <= 4.
#include "common.hpp"

#pragma clang fp contract(off)

namespace shp {

// ---- integral image -----------------------------------------------------------------------------
// Column pass: one lane per column, rows in sequence (the same association order as the
// reference's column cumulative sum), coalesced along x.
template <typename T>
__global__ __launch_bounds__(64) void integral_cols_kernel(const T *__restrict__ img, int H, int W,
                                                           double *__restrict__ tmp) {
    int x = blockIdx.x * 64 + threadIdx.x;
    if (x >= W) return;
    double s = 0.0;
    for (int y = 0; y < H; ++y) {
        s = s + (double)img[(size_t)y * W + x];
        tmp[(size_t)y * W + x] = s;
    }
}

// Row pass: the running sum along a row must add the columns strictly left to right to give the
// reference's bits for any float64 input (guided_filter.py:42-46), so one lane walks one row.  A
// wavefront owns 64 rows; 64x64 tiles go through LDS so that global loads and stores stay
// coalesced along x while each lane scans its own row of the tile.  Writes the zero-bordered
// (H+1, W+1) table.
__global__ __launch_bounds__(64) void integral_rows_kernel(const double *__restrict__ tmp, int H, int W,
                                                           double *__restrict__ integral) {
    __shared__ double tile[64][65];
    const int lane = threadIdx.x;
    const int r0 = blockIdx.x * 64;
    const int rows = min(64, H - r0);
    const size_t IW = (size_t)W + 1;
    if (blockIdx.x == 0)
        for (int x = lane; x <= W; x += 64) integral[x] = 0.0;
    if (lane < rows) integral[(size_t)(r0 + lane + 1) * IW] = 0.0;
    double carry = 0.0;
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int cols = min(64, W - x0);
        if (lane < cols)
            for (int i = 0; i < rows; ++i) tile[i][lane] = tmp[(size_t)(r0 + i) * W + x0 + lane];
        __syncthreads();
        if (lane < rows)
            for (int j = 0; j < cols; ++j) {
                carry = carry + tile[lane][j];
                tile[lane][j] = carry;
            }
        __syncthreads();
        if (lane < cols)
            for (int i = 0; i < rows; ++i) integral[(size_t)(r0 + i + 1) * IW + x0 + lane + 1] = tile[i][lane];
        __syncthreads();
    }
}

// I[y1+1,x1+1] - I[y0,x1+1] - I[y1+1,x0] + I[y0,x0], left to right (guided_filter.py:132-137),
// divided by the window area (guided_filter.py:179).
__global__ __launch_bounds__(256) void box_from_integral_kernel(const double *__restrict__ integral, int H, int W,
                                                                const int *__restrict__ y0a, const int *__restrict__ y1a,
                                                                const int *__restrict__ x0a, const int *__restrict__ x1a,
                                                                const double *__restrict__ area, int radius,
                                                                double *__restrict__ out) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t IW = (size_t)W + 1;
    int y0, y1, x0, x1;
    if (y0a) { y0 = y0a[y]; y1 = y1a[y]; x0 = x0a[x]; x1 = x1a[x]; }
    else {
        y0 = min(max(y - radius, 0), H - 1); y1 = min(max(y + radius, 0), H - 1);
        x0 = min(max(x - radius, 0), W - 1); x1 = min(max(x + radius, 0), W - 1);
    }
    double s = integral[(size_t)(y1 + 1) * IW + (x1 + 1)] - integral[(size_t)y0 * IW + (x1 + 1)]
               - integral[(size_t)(y1 + 1) * IW + x0] + integral[(size_t)y0 * IW + x0];
    double ar = area ? area[(size_t)y * W + x] : (double)((long long)(y1 - y0 + 1) * (long long)(x1 - x0 + 1));
    out[(size_t)y * W + x] = s / ar;
}

template <typename T>
static int launch_integral_t(const T *img, int H, int W, double *tmp, double *integral, hipStream_t s) {
    hipLaunchKernelGGL(integral_cols_kernel<T>, dim3(ceil_div(W, 64)), dim3(64), 0, s, img, H, W, tmp);
    SHP_LAUNCH_CHECK();
    hipLaunchKernelGGL(integral_rows_kernel, dim3(ceil_div(H, 64)), dim3(64), 0, s, tmp, H, W, integral);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}

int launch_integral(const void *d_img, int dtype, int H, int W, double *d_tmp, double *d_integral, hipStream_t s) {
    if (dtype == STEREO_DT_F64) return launch_integral_t(static_cast<const double *>(d_img), H, W, d_tmp, d_integral, s);
    if (dtype == STEREO_DT_F32) return launch_integral_t(static_cast<const float *>(d_img), H, W, d_tmp, d_integral, s);
    set_error("integral image takes float32 or float64");
    return STEREO_EINVAL;
}

int launch_box_from_integral(const double *d_integral, int H, int W, const int *y0, const int *y1, const int *x0,
                             const int *x1, const double *area, int radius, double *d_out, hipStream_t s) {
    dim3 grid(ceil_div(W, 64), ceil_div(H, 4));
    hipLaunchKernelGGL(box_from_integral_kernel, grid, dim3(256), 0, s, d_integral, H, W, y0, y1, x0, x1, area, radius,
                       d_out);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}

// box mean of a device plane; d_tmp: H*W doubles, d_integral: (H+1)*(W+1) doubles
int launch_box_mean(const void *d_img, int dtype, int H, int W, int radius, double *d_tmp, double *d_integral,
                    double *d_out, hipStream_t s) {
    SHP_TRY(launch_integral(d_img, dtype, H, W, d_tmp, d_integral, s));
    return launch_box_from_integral(d_integral, H, W, nullptr, nullptr, nullptr, nullptr, nullptr, radius, d_out, s);
}

// ---- element-wise pieces of the guided filter (guided_filter.py:202-216, :238-253) ------------

template <typename GT>
__global__ void gf_square_kernel(const GT *__restrict__ g, size_t n, double *__restrict__ g64, double *__restrict__ gg) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double v = (double)g[i];
        g64[i] = v;
        gg[i] = v * v;
    }
}
__global__ void gf_var_kernel(const double *__restrict__ mean_g, const double *__restrict__ mean_gg, size_t n,
                              double *__restrict__ var) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double v = mean_gg[i] - mean_g[i] * mean_g[i];
        var[i] = v > 0.0 ? v : (v != v ? v : 0.0);
    }
}
__global__ void gf_products_kernel(const double *__restrict__ g64, const float *__restrict__ src, size_t n,
                                   double *__restrict__ src64, double *__restrict__ gs) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double sv = (double)src[i];
        src64[i] = sv;
        gs[i] = g64[i] * sv;
    }
}
__global__ void gf_ab_kernel(const double *__restrict__ mean_src, const double *__restrict__ mean_gs,
                             const double *__restrict__ mean_g, const double *__restrict__ var, double eps32, size_t n,
                             double *__restrict__ a, double *__restrict__ b) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double cov = mean_gs[i] - mean_g[i] * mean_src[i];
        double den = var[i] + eps32;
        if (den <= 0.0) den = eps32;
        double av = cov / den;
        a[i] = av;
        b[i] = mean_src[i] - av * mean_g[i];
    }
}
__global__ void gf_out_kernel(const double *__restrict__ mean_a, const double *__restrict__ mean_b,
                              const double *__restrict__ g64, size_t n, float *__restrict__ out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (float)(mean_a[i] * g64[i] + mean_b[i]);
}

static inline int ew_grid(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// Workspace of the layered guided filter: 9 planes of H*W doubles + one (H+1)*(W+1) table.
size_t layered_guided_ws_bytes(int H, int W) {
    size_t n = (size_t)H * W;
    return align_up(n * 8, 256) * 9 + align_up(((size_t)H + 1) * ((size_t)W + 1) * 8, 256);
}

struct LayeredGuided {
    double *g64, *mean_g, *var, *t0, *t1, *t2, *t3, *tmp, *mean_gg, *integral;
};
static LayeredGuided carve_layered(void *ws, int H, int W) {
    size_t pl = align_up((size_t)H * W * 8, 256);
    unsigned char *b = static_cast<unsigned char *>(ws);
    LayeredGuided L;
    L.g64 = (double *)(b + 0 * pl); L.mean_g = (double *)(b + 1 * pl); L.var = (double *)(b + 2 * pl);
    L.t0 = (double *)(b + 3 * pl); L.t1 = (double *)(b + 4 * pl); L.t2 = (double *)(b + 5 * pl);
    L.t3 = (double *)(b + 6 * pl); L.tmp = (double *)(b + 7 * pl); L.mean_gg = (double *)(b + 8 * pl);
    L.integral = (double *)(b + 9 * pl);
    return L;
}

// prepare_guided_filter: fills g64, mean_g, mean_gg, var inside the workspace.
int layered_guided_prepare(const void *d_guide, int guide_dtype, int H, int W, int radius, void *ws, hipStream_t s) {
    LayeredGuided L = carve_layered(ws, H, W);
    size_t n = (size_t)H * W;
    if (guide_dtype == STEREO_DT_F64)
        hipLaunchKernelGGL(gf_square_kernel<double>, dim3(ew_grid(n)), dim3(256), 0, s, (const double *)d_guide, n, L.g64, L.t0);
    else if (guide_dtype == STEREO_DT_F32)
        hipLaunchKernelGGL(gf_square_kernel<float>, dim3(ew_grid(n)), dim3(256), 0, s, (const float *)d_guide, n, L.g64, L.t0);
    else { set_error("guide dtype must be float32 or float64"); return STEREO_EINVAL; }
    SHP_LAUNCH_CHECK();
    SHP_TRY(launch_box_mean(L.g64, STEREO_DT_F64, H, W, radius, L.tmp, L.integral, L.mean_g, s));
    SHP_TRY(launch_box_mean(L.t0, STEREO_DT_F64, H, W, radius, L.tmp, L.integral, L.mean_gg, s));
    hipLaunchKernelGGL(gf_var_kernel, dim3(ew_grid(n)), dim3(256), 0, s, L.mean_g, L.mean_gg, n, L.var);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}

void layered_guided_planes(void *ws, int H, int W, double **g64, double **mean_g, double **mean_gg, double **var) {
    LayeredGuided L = carve_layered(ws, H, W);
    *g64 = L.g64; *mean_g = L.mean_g; *mean_gg = L.mean_gg; *var = L.var;
}

// guided_filter_with_precompute on one float32 layer.
int layered_guided_apply(const float *d_src, int H, int W, int radius, double eps, void *ws, float *d_out,
                         hipStream_t s) {
    LayeredGuided L = carve_layered(ws, H, W);
    size_t n = (size_t)H * W;
    const double eps32 = (double)(float)eps;
    int g = ew_grid(n);
    hipLaunchKernelGGL(gf_products_kernel, dim3(g), dim3(256), 0, s, L.g64, d_src, n, L.t0, L.t1);
    SHP_LAUNCH_CHECK();
    SHP_TRY(launch_box_mean(L.t0, STEREO_DT_F64, H, W, radius, L.tmp, L.integral, L.t2, s));  // mean_src
    SHP_TRY(launch_box_mean(L.t1, STEREO_DT_F64, H, W, radius, L.tmp, L.integral, L.t3, s));  // mean_gs
    hipLaunchKernelGGL(gf_ab_kernel, dim3(g), dim3(256), 0, s, L.t2, L.t3, L.mean_g, L.var, eps32, n, L.t0, L.t1);
    SHP_LAUNCH_CHECK();
    SHP_TRY(launch_box_mean(L.t0, STEREO_DT_F64, H, W, radius, L.tmp, L.integral, L.t2, s));  // mean_a
    SHP_TRY(launch_box_mean(L.t1, STEREO_DT_F64, H, W, radius, L.tmp, L.integral, L.t3, s));  // mean_b
    hipLaunchKernelGGL(gf_out_kernel, dim3(g), dim3(256), 0, s, L.t2, L.t3, L.g64, n, d_out);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}

}  // namespace shp
