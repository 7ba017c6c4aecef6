// Summed-area tables and clipped-window box means in float64 (guided_filter.py:21-179), plus the
// layer-at-a-time guided filter built from them.  This is the general path: any radius, any
// window index arrays.  The fused kernel in guided.hip is the fast path for radius <= 4.
#include "common.hpp"

#pragma clang fp contract(off)

namespace shp {

// ---- integral image -----------------------------------------------------------------------------
// guided_filter.py:21-47: cumulative sum down the columns, then along the rows, into a zero-bordered
// (H+1, W+1) table.  Both passes are parallel scans:
//  * columns: the rows are cut into chunks of SCAN_ROWS; every (chunk, column) sums its rows (coalesced along x),
//    and a second sweep adds up the totals of the chunks above (its carry) and writes the running sums from there;
//  * rows: one wavefront per row walks the row in 64-column segments, each segment an inclusive wave scan in
//    registers (row_shr / row_bcast lane shifts, no LDS) plus the carry of the segments before it.
// The sums associate differently from the reference's sequential loops (chunk carries, scan trees), so a table of
// arbitrary float64 values agrees to rounding (relative 1e-12 in the tests) and a table of integer-valued inputs,
// where every partial sum is exact, bit for bit.  Non-finite inputs give the reference's pattern: a prefix that
// holds a NaN, or infinities of both signs, is NaN in any association.
constexpr int SCAN_ROWS = 64;

template <typename T>
__global__ __launch_bounds__(256) void integral_col_partial_kernel(const T *__restrict__ img, int H, int W,
                                                                   double *__restrict__ part) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const int y0 = blockIdx.y * SCAN_ROWS, y1 = min(H, y0 + SCAN_ROWS);
    double s = 0.0;
#pragma unroll 8
    for (int y = y0; y < y1; ++y) s = s + (double)img[(size_t)y * W + x];
    part[(size_t)blockIdx.y * W + x] = s;
}

template <typename T>
__global__ __launch_bounds__(256) void integral_col_scan_kernel(const T *__restrict__ img, int H, int W,
                                                                const double *__restrict__ part,
                                                                double *__restrict__ tmp) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const int y0 = blockIdx.y * SCAN_ROWS, y1 = min(H, y0 + SCAN_ROWS);
    double s = 0.0;   // carry: the totals of the chunks above, added top to bottom
    for (int c = 0; c < (int)blockIdx.y; ++c) s = s + part[(size_t)c * W + x];
#pragma unroll 8
    for (int y = y0; y < y1; ++y) {
        s = s + (double)img[(size_t)y * W + x];
        tmp[(size_t)y * W + x] = s;
    }
}

// lane shift of a float64 through two 32-bit DPP moves; lanes without a source (and rows outside ROW_MASK) read 0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_shift_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
    return __hiloint2double(hi, lo);
}
// inclusive prefix sum over the 64 lanes of a wavefront
__device__ __forceinline__ double wave_inclusive_scan(double v) {
    v = v + dpp_shift_f64<0x111, 0xf>(v);   // row_shr:1
    v = v + dpp_shift_f64<0x112, 0xf>(v);   // row_shr:2
    v = v + dpp_shift_f64<0x114, 0xf>(v);   // row_shr:4
    v = v + dpp_shift_f64<0x118, 0xf>(v);   // row_shr:8   -> scans of the four 16-lane rows
    v = v + dpp_shift_f64<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = v + dpp_shift_f64<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    return v;
}

// Row pass: tmp (column sums) -> integral, one wavefront per row, four segments of 64 columns in flight.
__global__ __launch_bounds__(256) void integral_row_scan_kernel(const double *__restrict__ tmp, int H, int W,
                                                                double *__restrict__ integral) {
    const int lane = threadIdx.x & 63;
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t IW = (size_t)W + 1;
    if (blockIdx.x == 0)
        for (int x = threadIdx.x; x <= W; x += 256) integral[x] = 0.0;   // the zero row above the image
    if (y >= H) return;
    const double *src = tmp + (size_t)y * W;
    double *dst = integral + (size_t)(y + 1) * IW;
    if (lane == 0) dst[0] = 0.0;
    double carry = 0.0;
    for (int x0 = 0; x0 < W; x0 += 256) {
        double v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = x0 + k * 64 + lane;
            v[k] = x < W ? src[x] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = wave_inclusive_scan(v[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = x0 + k * 64 + lane;
            const double out = carry + v[k];
            if (x < W) dst[x + 1] = out;
            carry = __shfl(out, 63);
        }
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

// d_tmp: H * W doubles; d_integral: (H + 1) * (W + 1) doubles, whose first n_chunks * W entries double as the
// scratch of the column pass (it is complete before the row pass writes the table)
template <typename T>
static int launch_integral_t(const T *img, int H, int W, double *tmp, double *integral, hipStream_t s) {
    const int n_chunks = ceil_div(H, SCAN_ROWS);
    double *part = integral;
    dim3 grid(ceil_div(W, 256), n_chunks);
    hipLaunchKernelGGL(integral_col_partial_kernel<T>, grid, dim3(256), 0, s, img, H, W, part);
    SHP_LAUNCH_CHECK();
    hipLaunchKernelGGL(integral_col_scan_kernel<T>, grid, dim3(256), 0, s, img, H, W, part, tmp);
    SHP_LAUNCH_CHECK();
    hipLaunchKernelGGL(integral_row_scan_kernel, dim3(ceil_div(H, 4)), dim3(256), 0, s, tmp, H, W, integral);
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
