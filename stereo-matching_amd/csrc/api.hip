// C ABI of libstereo_hip.so (see include/stereo_hip.h).
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "common.hpp"
#include <algorithm>
#include <climits>
#include <vector>

namespace shp {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? STEREO_ENOMEM : STEREO_EHIP;
}

// Optional timing of the dominant kernel: events are created by stereo_profile_begin (never in
// the launch path) and recorded on the launch stream around every fused-kernel launch.
struct ProfileState {
    std::vector<hipEvent_t> start, stop;
    int used = 0;
    bool enabled = false;
};
static ProfileState g_prof;
// debugging/testing: 0 = best kernel, 1 = general fused kernel, 2 = fast kernel with one column per lane
static int g_force_kernel = 0;

// ---- cache of device buffers ---------------------------------------------------------------------
namespace {
struct PoolEntry { void *p; size_t capacity; int device; };
std::mutex g_pool_mutex;
std::vector<PoolEntry> g_pool;
size_t g_pool_bytes = 0;
constexpr size_t POOL_MAX_BYTES = (size_t)8 << 30;   // cached (idle) bytes kept per process
}  // namespace

void *pool_alloc(size_t bytes, size_t *capacity) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        int best = -1;
        for (int i = 0; i < (int)g_pool.size(); ++i) {
            const PoolEntry &e = g_pool[i];
            // a cached buffer serves a request it is not much too large for
            if (e.device == dev && e.capacity >= bytes && e.capacity <= bytes + bytes / 4 + (1u << 20) &&
                (best < 0 || e.capacity < g_pool[best].capacity))
                best = i;
        }
        if (best >= 0) {
            PoolEntry e = g_pool[best];
            g_pool.erase(g_pool.begin() + best);
            g_pool_bytes -= e.capacity;
            *capacity = e.capacity;
            return e.p;
        }
    }
    void *p = nullptr;
    hipError_t err = hipMalloc(&p, bytes);
    if (err != hipSuccess) {
        // memory held by the cache may be what is missing
        (void)hipGetLastError();
        pool_release_all();
        err = hipMalloc(&p, bytes);
    }
    if (err != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(err));
        (void)hipGetLastError();
        return nullptr;
    }
    *capacity = bytes;
    return p;
}

void pool_free(void *p, size_t capacity) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        if (g_pool_bytes + capacity <= POOL_MAX_BYTES && g_pool.size() < 64) {
            g_pool.push_back(PoolEntry{p, capacity, dev});
            g_pool_bytes += capacity;
            return;
        }
    }
    (void)hipFree(p);
}

void pool_release_all() {
    std::vector<PoolEntry> drop;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        drop.swap(g_pool);
        g_pool_bytes = 0;
    }
    for (const PoolEntry &e : drop) (void)hipFree(e.p);
}


static int check_params(const StereoParams *p) {
    if (!p) { set_error("params is NULL"); return STEREO_EINVAL; }
    switch (p->filter_kind) {
        case STEREO_FILTER_GUIDED:
            if (p->guided_radius <= 0) { set_error("guided radius must be a positive integer"); return STEREO_EINVAL; }
            if (!(p->guided_eps > 0)) { set_error("guided eps must be positive"); return STEREO_EINVAL; }
            break;
        case STEREO_FILTER_MEDIAN:
        case STEREO_FILTER_MEDIAN_U8Q:
            if (p->median_radius <= 0) { set_error("median radius must be a positive integer"); return STEREO_EINVAL; }
            break;
        case STEREO_FILTER_GAUSSIAN:
            if (!(p->gaussian_sigma > 0)) { set_error("gaussian sigma must be positive"); return STEREO_EINVAL; }
            break;
        case STEREO_FILTER_BILATERAL:
            if (!(p->bilateral_sigma > 0)) { set_error("bilateral sigma must be positive"); return STEREO_EINVAL; }
            break;
        default: set_error("unknown filter kind %d", p->filter_kind); return STEREO_EINVAL;
    }
    return STEREO_OK;
}

static int check_image_dims(int H, int W) {
    if (H <= 0 || W <= 0) { set_error("image must be non-empty, got %d x %d", H, W); return STEREO_EINVAL; }
    if (H > 65535) { set_error("image height %d exceeds 65535", H); return STEREO_EUNSUPPORTED; }
    // the kernels address census words with 32-bit byte offsets (8 bytes per pixel at census radius > 4)
    if ((size_t)H * (size_t)W > ((size_t)1 << 27)) { set_error("image of %d x %d pixels exceeds 2^27 pixels", H, W); return STEREO_EUNSUPPORTED; }
    return STEREO_OK;
}

static size_t dtype_size(int dt) { return dt == STEREO_DT_F64 ? 8 : (dt == STEREO_DT_U8 ? 1 : 4); }

// layout of the fused workspace
struct FusedWs {
    size_t words_bytes, plane_f64;
    size_t off_lw, off_rw, off_mean, off_invden, off_gmax, off_mm, total;
};
// (8e-3: at 5e-3 two of 440 random frames were off by 1.1e-5 and 2.8e-5 of a cost unit where an overshoot next to a
// run of `large` costs brings the filtered cost close to zero; from 6e-3 up the same frames are within 1e-6)
constexpr double FAST_KERNEL_MIN_EPS = 8e-3;
// ... and the precise variant of the two-column kernel (float64 sums of the guide's low-order part, float64 second
// stage) from here on
constexpr double PRECISE_KERNEL_MIN_EPS = 0.0;    // (every eps: the variant keeps nothing in float32 whose error grows as eps falls)
constexpr int FUSED_MAX_DISPARITIES = 65536;  // per launch; sizes the per-layer min/max scratch of the median path
static FusedWs fused_ws_layout(int H, int W, int wct_radius) {
    FusedWs w;
    size_t n = (size_t)H * W;
    w.words_bytes = align_up(n * (wct_radius > 4 ? 8 : 4), 256);
    w.plane_f64 = align_up(n * 8, 256);
    w.off_lw = 0;
    w.off_rw = w.off_lw + w.words_bytes;
    w.off_mean = w.off_rw + w.words_bytes;
    w.off_invden = w.off_mean + w.plane_f64;
    w.off_gmax = w.off_invden + w.plane_f64;
    w.off_mm = w.off_gmax + 256;
    w.total = w.off_mm + (size_t)FUSED_MAX_DISPARITIES * 8;
    return w;
}

// Fused compute_disparity on device float32 images; keys receive the packed result.
static int fused_disparity_dev(const float *d_left, const float *d_right, int H, int W, int dmax,
                               const StereoParams *p, int d_begin, int d_end, void *ws, size_t ws_bytes,
                               uint64_t *d_keys, float *d_q_out, hipStream_t s,
                               const BandProgress *progress = nullptr) {
    SHP_TRY(check_params(p));
    SHP_TRY(check_image_dims(H, W));
    if (dmax <= 0) { set_error("dmax must be a positive integer"); return STEREO_EINVAL; }
    if (d_begin < 0 || d_end > dmax || d_begin > d_end) {
        set_error("disparity range [%d, %d) outside [0, %d)", d_begin, d_end, dmax);
        return STEREO_EINVAL;
    }
    CensusCfg cfg;
    SHP_TRY(make_census_cfg(p->wct_radius, p->base_weight, &cfg));
    FusedWs L = fused_ws_layout(H, W, p->wct_radius);
    if (ws_bytes < L.total) { set_error("workspace too small: %zu < %zu", ws_bytes, L.total); return STEREO_EINVAL; }
    unsigned char *base = static_cast<unsigned char *>(ws);
    void *lw = base + L.off_lw, *rw = base + L.off_rw;
    double *mean = reinterpret_cast<double *>(base + L.off_mean);
    double *invden = reinterpret_cast<double *>(base + L.off_invden);
    const bool merged_prepass = p->filter_kind == STEREO_FILTER_GUIDED && census_stats_merged(p->wct_radius, p->guided_radius);
    if (!merged_prepass) SHP_TRY(launch_census_pack_pair(d_left, d_right, STEREO_DT_F32, H, W, p->wct_radius, lw, rw, s));
    if (p->filter_kind == STEREO_FILTER_GUIDED) {
        if (!fused_guided_supported(p->guided_radius)) {
            set_error("fused guided path supports radius 1..8, got %d", p->guided_radius);
            return STEREO_EUNSUPPORTED;
        }
        float *gmax = reinterpret_cast<float *>(base + L.off_gmax);
        // (census words of both images and the guide statistics: one launch where the radii allow it)
        if (merged_prepass)
            SHP_TRY(launch_census_stats(d_left, d_right, H, W, p->wct_radius, p->guided_radius, p->guided_eps, lw, rw, mean,
                                        invden, gmax, s));
        else
            SHP_TRY(launch_guide_stats(d_left, STEREO_DT_F32, H, W, p->guided_radius, p->guided_eps, mean, invden,
                                       gmax, s));
        if (d_begin == d_end) return launch_fill_keys(d_keys, (size_t)H * W, s);
        // The fast kernels evaluate the second box stage in float32: its terms grow like 1 / sqrt(eps), and
        // below this eps their rounding can exceed 1e-5 of a cost unit where the filtered cost is near zero.
        // Smaller eps takes the general kernel (float64 throughout, bit-identical to the CPU restatement).
        const bool eps_ok = p->guided_eps >= FAST_KERNEL_MIN_EPS || g_force_kernel == 2;
        const bool pair_cfg = fused_pair_supported(cfg, p->guided_radius) && (g_force_kernel == 0 || g_force_kernel == 3);
        const bool pair = pair_cfg && g_force_kernel == 0 && eps_ok;
        // the same kernel with a float64 second stage: eps from PRECISE_KERNEL_MIN_EPS up to the fast kernels' limit
        const bool pair_precise = pair_cfg && p->guided_radius <= 4 && (g_force_kernel == 3 || p->guided_eps >= PRECISE_KERNEL_MIN_EPS);
        const bool fast = !pair_cfg && fused_fast_supported(cfg, p->guided_radius) && g_force_kernel != 1 && eps_ok;
        const int slices = (pair || pair_precise) ? fused_pair_slices(H, W, p->guided_radius, cfg.word64 != 0, d_end - d_begin)
                           : fast ? fused_fast_slices(H, W, p->guided_radius, d_end - d_begin, cfg.word64 != 0) : 1;
        if (slices > 1) SHP_TRY(launch_fill_keys(d_keys, (size_t)H * W, s));
        const bool prof = g_prof.enabled && g_prof.used < (int)g_prof.start.size();
        if (prof) SHP_HIP(hipEventRecord(g_prof.start[g_prof.used], s));
        // eps is judged against the guide's range m = max(1, max|guide|^2) on the device (the range is only known
        // there).  Every kernel that could serve the frame is enqueued with its own test -- the fast variants run iff
        // eps >= FAST_KERNEL_MIN_EPS * m, the precise variant of the two-column kernel (same launch) iff
        // PRECISE_KERNEL_MIN_EPS * m <= eps < FAST_KERNEL_MIN_EPS * m, the general kernel below that -- so exactly
        // one does the frame (the other returns at once) and the call stays asynchronous.  A forced kernel (test
        // hook) runs unconditionally.
        const float gate_eps = (float)p->guided_eps;
        const float gate_min = g_force_kernel == 0 ? (float)FAST_KERNEL_MIN_EPS : 0.f;
        const float gate_low = g_force_kernel == 0 ? (float)PRECISE_KERNEL_MIN_EPS : 0.f;
        int rc = STEREO_OK;
        bool gated = false;   // some kernel was enqueued with a test: the general kernel follows with the remaining range
        if (pair || pair_precise) {
            // one launch holds the fast and the precise variant of the two-column kernel; the test on the device
            // picks one (or neither, when eps lies below the precise variant's range)
            rc = launch_fused_guided_pair(lw, rw, cfg, d_left, mean, invden, gmax, H, W, dmax, p->guided_radius, d_begin,
                                          d_end, slices, d_keys, d_q_out, s, progress, gate_eps, gate_min, pair_precise,
                                          gate_low);
            gated = true;
        }
        if (rc == STEREO_OK && fast) {
            rc = launch_fused_guided_fast(lw, rw, cfg, d_left, mean, invden, gmax, H, W, dmax, p->guided_radius, d_begin,
                                          d_end, slices, d_keys, d_q_out, s, gate_eps, gate_min);
            gated = true;
        }
        // (a forced fast / precise kernel has done the frame; so has the two-column kernel when its precise
        //  variant serves every eps the fast one declines)
        const bool pair_serves_all = (pair || pair_precise) && pair_precise && PRECISE_KERNEL_MIN_EPS <= 0.0;
        if (rc == STEREO_OK && !(gated && gate_min == 0.f) && !pair_serves_all)
            rc = launch_fused_guided(lw, rw, &cfg, nullptr, d_left, STEREO_DT_F32, mean, invden, H, W, dmax,
                                     p->guided_radius, d_begin, d_end, d_keys, d_q_out, 0, s, gated ? gmax : nullptr,
                                     gate_eps, !gated ? 0.f : pair_precise ? gate_low : gate_min);
        if (prof) {
            SHP_HIP(hipEventRecord(g_prof.stop[g_prof.used], s));
            ++g_prof.used;
        }
        return rc;
    }
    int radius = p->median_radius;
    double sigma = 0.0;
    if (p->filter_kind == STEREO_FILTER_GAUSSIAN || p->filter_kind == STEREO_FILTER_BILATERAL) {
        sigma = p->filter_kind == STEREO_FILTER_GAUSSIAN ? p->gaussian_sigma : p->bilateral_sigma;
        radius = radius_from_sigma(sigma);
    }
    if (!fused_filter_supported(p->filter_kind, radius)) {
        set_error("filter kind %d with radius %d has no fused kernel (use the layer-at-a-time path)", p->filter_kind, radius);
        return STEREO_EUNSUPPORTED;
    }
    if (d_end - d_begin > FUSED_MAX_DISPARITIES) { set_error("more than %d disparities per launch", FUSED_MAX_DISPARITIES); return STEREO_EUNSUPPORTED; }
    const bool prof = g_prof.enabled && g_prof.used < (int)g_prof.start.size();
    if (prof) SHP_HIP(hipEventRecord(g_prof.start[g_prof.used], s));
    int rc = launch_fused_filter(lw, rw, cfg, H, W, p->filter_kind, radius, sigma, d_begin, d_end,
                                 reinterpret_cast<uint32_t *>(base + L.off_mm), d_keys, s, g_force_kernel == 1);
    if (prof) {
        SHP_HIP(hipEventRecord(g_prof.stop[g_prof.used], s));
        ++g_prof.used;
    }
    return rc;
}

// Tile height of the kernel fused_disparity_dev dispatches when that kernel reports row-band
// progress while it runs; 0 when it does not (the flags are then set after the kernel).
static int progress_tile_rows(const StereoParams *p, const CensusCfg &cfg) {
    if (p->filter_kind != STEREO_FILTER_GUIDED) return 0;
    const bool eps_ok = p->guided_eps >= PRECISE_KERNEL_MIN_EPS;   // (either variant of the two-column kernel)
    if (!(fused_pair_supported(cfg, p->guided_radius) && g_force_kernel == 0 && eps_ok)) return 0;
    return pair_tile_height(p->guided_radius, cfg.word64 != 0);
}

// whether compute_disparity with these parameters runs volume-free
static bool fused_path_applies(const StereoParams *p) {
    if (p->filter_kind == STEREO_FILTER_GUIDED) return fused_guided_supported(p->guided_radius);
    if (p->filter_kind == STEREO_FILTER_GAUSSIAN) return fused_filter_supported(p->filter_kind, radius_from_sigma(p->gaussian_sigma));
    if (p->filter_kind == STEREO_FILTER_BILATERAL) return fused_filter_supported(p->filter_kind, radius_from_sigma(p->bilateral_sigma));
    if (p->filter_kind == STEREO_FILTER_MEDIAN) {   // the exact median is fused for exactly summable census weights only
        CensusCfg cfg;
        if (make_census_cfg(p->wct_radius, p->base_weight, &cfg) != STEREO_OK || !cfg.exact) return false;
    }
    return fused_filter_supported(p->filter_kind, p->median_radius);
}

// sets *flag when some element is NaN or infinite
__global__ __launch_bounds__(256) void nonfinite_flag_kernel(const float *__restrict__ v, size_t n, uint32_t *flag) {
    bool bad = false;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        bad = bad || !(fabsf(v[i]) <= 3.402823466e38f);
    if (bad) *flag = 1u;
}

// stereo.py:145-170 one layer at a time on a (D', H, W) device volume whose layer k is
// disparity d_begin + k.  Used for every filter / parameter combination the fused kernels do
// not cover.  Everything is enqueued on `s`; the only host sync is the final flag read.
static int layered_aggregate_dev(const float *d_layers, const void *d_guide, int guide_dtype, int H, int W,
                                 const StereoParams *p, int d_begin, int d_end, uint64_t *d_keys, hipStream_t s) {
    const size_t n = (size_t)H * W;
    SHP_TRY(launch_fill_keys(d_keys, n, s));
    if (d_begin >= d_end) return STEREO_OK;
    DevBuf out, ws, kern, mm, q, sc;
    SHP_TRY(out.alloc(n * 4));
    int radius = 0;
    switch (p->filter_kind) {
        case STEREO_FILTER_GUIDED:
            SHP_TRY(ws.alloc(layered_guided_ws_bytes(H, W)));
            SHP_TRY(layered_guided_prepare(d_guide, guide_dtype, H, W, p->guided_radius, ws.p, s));
            break;
        case STEREO_FILTER_MEDIAN: break;
        case STEREO_FILTER_MEDIAN_U8Q:
            SHP_TRY(mm.alloc(16)); SHP_TRY(q.alloc(n)); SHP_TRY(sc.alloc(16));
            SHP_HIP(hipMemsetAsync(mm.p, 0, 16, s));
            break;
        case STEREO_FILTER_GAUSSIAN:
        case STEREO_FILTER_BILATERAL: {
            double sigma = p->filter_kind == STEREO_FILTER_GAUSSIAN ? p->gaussian_sigma : p->bilateral_sigma;
            radius = radius_from_sigma(sigma);
            if (radius > 32) { set_error("filter radius %d > 32 is not supported on the GPU path", radius); return STEREO_EUNSUPPORTED; }
            std::vector<float> k((size_t)(2 * radius + 1) * (2 * radius + 1));
            gaussian_kernel_host(radius, sigma, k.data());
            SHP_TRY(kern.alloc(k.size() * 4));
            SHP_HIP(hipMemcpyAsync(kern.p, k.data(), k.size() * 4, hipMemcpyHostToDevice, s));
            SHP_HIP(hipStreamSynchronize(s));  // k goes out of scope below
            break;
        }
        default: set_error("unknown filter kind"); return STEREO_EINVAL;
    }
    for (int d = d_begin; d < d_end; ++d) {
        const float *layer = d_layers + (size_t)(d - d_begin) * n;
        switch (p->filter_kind) {
            case STEREO_FILTER_GUIDED:
                SHP_TRY(layered_guided_apply(layer, H, W, p->guided_radius, p->guided_eps, ws.p, out.as<float>(), s));
                break;
            case STEREO_FILTER_MEDIAN: SHP_TRY(launch_median_f32(layer, H, W, p->median_radius, out.as<float>(), s)); break;
            case STEREO_FILTER_MEDIAN_U8Q:
                SHP_TRY(launch_median_u8q(layer, H, W, p->median_radius, mm.as<uint32_t>(), q.as<uint8_t>(), sc.as<float>(),
                                          out.as<float>(), s));
                break;
            case STEREO_FILTER_GAUSSIAN: SHP_TRY(launch_gaussian(layer, H, W, radius, kern.as<float>(), out.as<float>(), s)); break;
            case STEREO_FILTER_BILATERAL:
                SHP_TRY(launch_bilateral(layer, H, W, radius, kern.as<float>(), p->bilateral_sigma, out.as<float>(), s));
                break;
        }
        SHP_TRY(launch_wta_update(out.as<float>(), n, d, d_keys, s));
    }
    if (p->filter_kind == STEREO_FILTER_MEDIAN_U8Q) {
        uint32_t h[4] = {0, 0, 0, 0};
        SHP_HIP(hipMemcpyAsync(h, mm.p, 16, hipMemcpyDeviceToHost, s));
        SHP_HIP(hipStreamSynchronize(s));
        if (h[2]) { set_error("the opencv median needs finite values"); return STEREO_ENONFINITE; }
    }
    SHP_HIP(hipStreamSynchronize(s));
    return STEREO_OK;
}

static int finish_keys_to_host(uint64_t *d_keys, size_t n, int32_t *out_disp, float *out_cost) {
    DevBuf disp, cost;
    SHP_TRY(disp.alloc(n * 4)); SHP_TRY(cost.alloc(n * 4));
    SHP_TRY(launch_unpack_keys(d_keys, n, disp.as<int32_t>(), cost.as<float>(), nullptr));
    SHP_HIP(hipMemcpy(out_disp, disp.p, n * 4, hipMemcpyDeviceToHost));
    SHP_HIP(hipMemcpy(out_cost, cost.p, n * 4, hipMemcpyDeviceToHost));
    return STEREO_OK;
}

// The reference forms its box sums from integral images (guided_filter.py:33-46,132-137): a NaN or
// infinity in the guide turns every cumulative sum below and to the right of it into NaN, hence every
// window whose lower right corner lies there, through both box stages: all outputs at
// y >= yb - 2r, x >= xb - 2r.  The fused kernels sum windows directly and keep NaNs local, so the host
// entry points extend the damage to the reference's region (rare path: only when the guide is not
// finite).  fill(y, x0) is called for every row with the first poisoned column.
template <typename GT, typename Fill>
static void poison_like_reference(const GT *guide, int H, int W, int radius, Fill fill) {
    std::vector<int> first(H, INT32_MAX);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const double v = (double)guide[(size_t)y * W + x];
            if (!(std::fabs(v) <= 1.7976931348623157e308)) { first[y] = x; break; }
        }
    int run = INT32_MAX;
    std::vector<int> pm(H);
    for (int y = 0; y < H; ++y) { run = std::min(run, first[y]); pm[y] = run; }
    for (int y = 0; y < H; ++y) {
        const int xb = pm[std::min(y + 2 * radius, H - 1)];
        if (xb != INT32_MAX) fill(y, std::max(xb - 2 * radius, 0));
    }
}
static void poison_disparity(const void *guide, int guide_dtype, int H, int W, int radius, int32_t *disp, float *cost) {
    auto fill = [&](int y, int x0) {
        for (int x = x0; x < W; ++x) { disp[(size_t)y * W + x] = 0; cost[(size_t)y * W + x] = INFINITY; }
    };
    if (guide_dtype == STEREO_DT_F64) poison_like_reference(static_cast<const double *>(guide), H, W, radius, fill);
    else if (guide_dtype == STEREO_DT_F32) poison_like_reference(static_cast<const float *>(guide), H, W, radius, fill);
}
static int guide_is_finite(const void *d_gmax, bool *finite) {
    uint32_t w[2] = {0u, 0u};
    SHP_HIP(hipMemcpy(w, d_gmax, 8, hipMemcpyDeviceToHost));
    *finite = w[1] == 0u;
    return STEREO_OK;
}

}  // namespace shp

using namespace shp;

// ---- scratch of the entry point that allocates nothing --------------------------------------------------------
// Every GPU the library's code object is loaded on holds its own copy of this array, at its own address, until
// the process ends.  In host code the array's NAME is only a handle for hipGetSymbolAddress: the address a kernel
// may use is looked up once per device (noalloc_scratch_for_current_device) and kept in g_noalloc_ptr.
constexpr size_t NOALLOC_SCRATCH_BYTES = (size_t)STEREO_NOALLOC_MAX_PIXELS * 32 + 256 + (size_t)65536 * 8 + 1024;
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];
static void *g_noalloc_ptr[64] = {nullptr};
static std::mutex g_noalloc_mutex;

// Device address of `scratch` on the current device (looked up at the first use on that device; stereo_hip_init
// does it ahead of time).  No allocation: the array is part of the loaded code object.
static int noalloc_scratch_for_current_device(void **out) {
    int dev = 0;
    SHP_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { set_error("device index %d beyond the 64 this library keeps state for", dev); return STEREO_EUNSUPPORTED; }
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (!g_noalloc_ptr[dev]) {
        void *ptr = nullptr;
        SHP_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) { set_error("hipGetSymbolAddress returned a null address for the static scratch"); return STEREO_EHIP; }
        g_noalloc_ptr[dev] = ptr;
    }
    *out = g_noalloc_ptr[dev];
    return STEREO_OK;
}

extern "C" {

const char *stereo_hip_last_error(void) { return g_err; }
#ifndef STEREO_BUILD_ID
#define STEREO_BUILD_ID "unknown"
#endif
const char *stereo_hip_version(void) { return "stereo_hip 0.2 (gfx950) build " STEREO_BUILD_ID; }

int stereo_hip_device_count(int *count) {
    if (!count) { set_error("null pointer"); return STEREO_EINVAL; }
    if (!count) return STEREO_EINVAL;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return STEREO_OK;
}

int stereo_hip_init(int device) {
    int n = 0;
    SHP_HIP(hipGetDeviceCount(&n));
    if (n <= 0) { set_error("no HIP device is visible"); return STEREO_EHIP; }
    if (device < 0 || device >= n) { set_error("device %d out of range (0..%d)", device, n - 1); return STEREO_EINVAL; }
    SHP_HIP(hipSetDevice(device));
    void *unused = nullptr;
    return noalloc_scratch_for_current_device(&unused);   // so that the first no-allocation call has nothing to look up
}

int stereo_generate_offsets(int radius, int32_t *out) {
    if (!out) { set_error("null pointer"); return STEREO_EINVAL; }
    static const int dirs[8][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}, {1, 1}, {1, -1}, {-1, 1}, {-1, -1}};
    if (radius <= 0 || !out) { set_error("radius must be a positive integer"); return STEREO_EINVAL; }
    int n = 0;
    for (int k = 0; k < 8; ++k)
        for (int r = 1; r <= radius; ++r, ++n) {
            out[3 * n + 0] = dirs[k][0] * r;
            out[3 * n + 1] = dirs[k][1] * r;
            out[3 * n + 2] = r;
        }
    return STEREO_OK;
}

int stereo_compute_weights(const int32_t *offsets, int n, double base_weight, float *out) {
    if (!offsets || !out) { set_error("null pointer"); return STEREO_EINVAL; }
    if (n < 0 || (n > 0 && (!offsets || !out))) { set_error("bad offsets"); return STEREO_EINVAL; }
    for (int i = 0; i < n; ++i) out[i] = (float)(base_weight / std::ldexp(1.0, offsets[3 * i + 2] - 1));
    return STEREO_OK;
}

int stereo_large_value(int radius, double base_weight, float *out) {
    if (!out) { set_error("null pointer"); return STEREO_EINVAL; }
    CensusCfg cfg;
    SHP_TRY(make_census_cfg(radius, base_weight, &cfg));
    *out = cfg.large;
    return STEREO_OK;
}

int stereo_wct_cost_volume_host(const void *left, const void *right, int dtype, int H, int W, int dmax, int radius,
                                double base_weight, float *out_dsi) {
    if (!left || !right || !out_dsi) { set_error("null pointer"); return STEREO_EINVAL; }
    if (dmax <= 0) { set_error("dmax must be a positive integer"); return STEREO_EINVAL; }
    SHP_TRY(check_image_dims(H, W));
    if (dtype != STEREO_DT_F32 && dtype != STEREO_DT_F64 && dtype != STEREO_DT_U8) {
        set_error("unsupported image dtype %d", dtype);
        return STEREO_EINVAL;
    }
    CensusCfg cfg;
    SHP_TRY(make_census_cfg(radius, base_weight, &cfg));
    size_t n = (size_t)H * W, img_bytes = n * dtype_size(dtype), word_bytes = n * (cfg.word64 ? 8 : 4);
    DevBuf dl, dr, lw, rw, dsi;
    SHP_TRY(dl.alloc(img_bytes)); SHP_TRY(dr.alloc(img_bytes));
    SHP_TRY(lw.alloc(word_bytes)); SHP_TRY(rw.alloc(word_bytes));
    SHP_TRY(dsi.alloc(n * (size_t)dmax * sizeof(float)));
    SHP_HIP(hipMemcpy(dl.p, left, img_bytes, hipMemcpyHostToDevice));
    SHP_HIP(hipMemcpy(dr.p, right, img_bytes, hipMemcpyHostToDevice));
    SHP_TRY(launch_census_pack_pair(dl.p, dr.p, dtype, H, W, radius, lw.p, rw.p, nullptr));
    SHP_TRY(launch_dsi(lw.p, rw.p, H, W, dmax, cfg, dsi.as<float>(), nullptr));
    SHP_HIP(hipMemcpy(out_dsi, dsi.p, n * (size_t)dmax * sizeof(float), hipMemcpyDeviceToHost));
    return STEREO_OK;
}

int stereo_fused_workspace_bytes(int H, int W, const StereoParams *p, size_t *bytes) {
    if (!p || !bytes) { set_error("NULL argument"); return STEREO_EINVAL; }
    SHP_TRY(check_image_dims(H, W));
    *bytes = fused_ws_layout(H, W, p->wct_radius).total;
    return STEREO_OK;
}

int stereo_compute_disparity_dev(const float *d_left, const float *d_right, int H, int W, int dmax,
                                 const StereoParams *p, int d_begin, int d_end, void *ws, size_t ws_bytes,
                                 uint64_t *d_keys, void *stream) {
    if (!d_left || !d_right || !p || !ws || !d_keys) { set_error("null pointer"); return STEREO_EINVAL; }
    return fused_disparity_dev(d_left, d_right, H, W, dmax, p, d_begin, d_end, ws, ws_bytes, d_keys, nullptr,
                               static_cast<hipStream_t>(stream));
}

// Event pair of the profile around one kernel (or kernel chain) of a device entry point.
struct ProfScope {
    hipStream_t s;
    bool on;
    explicit ProfScope(hipStream_t stream) : s(stream), on(g_prof.enabled && g_prof.used < (int)g_prof.start.size()) {}
    int begin() { return on ? (hipEventRecord(g_prof.start[g_prof.used], s) == hipSuccess ? STEREO_OK : STEREO_EHIP) : STEREO_OK; }
    int end() {
        if (!on) return STEREO_OK;
        if (hipEventRecord(g_prof.stop[g_prof.used], s) != hipSuccess) return STEREO_EHIP;
        ++g_prof.used;
        return STEREO_OK;
    }
};

int stereo_wct_cost_volume_dev(const float *d_left, const float *d_right, int H, int W, int dmax, int radius,
                               double base_weight, void *ws, size_t ws_bytes, float *d_dsi, void *stream) {
    if (!d_left || !d_right || !ws || !d_dsi) { set_error("null pointer"); return STEREO_EINVAL; }
    if (dmax <= 0) { set_error("dmax must be a positive integer"); return STEREO_EINVAL; }
    SHP_TRY(check_image_dims(H, W));
    CensusCfg cfg;
    SHP_TRY(make_census_cfg(radius, base_weight, &cfg));
    FusedWs L = fused_ws_layout(H, W, radius);
    if (ws_bytes < L.total) { set_error("workspace too small: %zu < %zu", ws_bytes, L.total); return STEREO_EINVAL; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned char *base = static_cast<unsigned char *>(ws);
    void *lw = base + L.off_lw, *rw = base + L.off_rw;
    SHP_TRY(launch_census_pack_pair(d_left, d_right, STEREO_DT_F32, H, W, radius, lw, rw, s));
    ProfScope prof(s);
    SHP_TRY(prof.begin());
    SHP_TRY(launch_dsi(lw, rw, H, W, dmax, cfg, d_dsi, s));
    return prof.end();
}

// Cost codes of a volume: a header of PAIR_CODE_STATS_WORDS device words (256 bytes reserved), then one [H][W] layer
// of 16-bit codes per disparity of the encoded range.
constexpr size_t CODES_HEADER_BYTES = 256;
constexpr int CODES_MAX_DISPARITIES = 639;   // per encoding: a group of 128 pixels x all disparities fits the LDS
static bool aggregate_from_codes_applies(const StereoParams *p, int n_disp) {
    return p->filter_kind == STEREO_FILTER_GUIDED && p->guided_radius >= 1 && p->guided_radius <= 4 &&
           n_disp <= CODES_MAX_DISPARITIES;
}

int stereo_cost_codes_bytes(int H, int W, int n_disp, size_t *bytes) {
    if (!bytes) { set_error("null pointer"); return STEREO_EINVAL; }
    SHP_TRY(check_image_dims(H, W));
    if (n_disp < 0) { set_error("negative disparity count"); return STEREO_EINVAL; }
    *bytes = CODES_HEADER_BYTES + cost_codes_bytes(H, W, n_disp);
    return STEREO_OK;
}

int stereo_encode_cost_volume_dev(const float *d_dsi, int H, int W, int dmax, int d_begin, int d_end, void *d_codes,
                                  size_t codes_bytes, void *stream) {
    if (!d_dsi || !d_codes) { set_error("null pointer"); return STEREO_EINVAL; }
    SHP_TRY(check_image_dims(H, W));
    if (dmax <= 0) { set_error("dsi must have at least one disparity layer"); return STEREO_EINVAL; }
    if (d_begin < 0 || d_end > dmax || d_begin > d_end) { set_error("bad disparity range"); return STEREO_EINVAL; }
    const size_t need = CODES_HEADER_BYTES + cost_codes_bytes(H, W, d_end - d_begin);
    if (codes_bytes < need) { set_error("cost-code buffer too small: %zu < %zu", codes_bytes, need); return STEREO_EINVAL; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned char *base = static_cast<unsigned char *>(d_codes);
    ProfScope prof(s);
    SHP_TRY(prof.begin());
    SHP_TRY(launch_encode_cost_codes(d_dsi, H, W, dmax, d_begin, d_end,
                                     reinterpret_cast<uint16_t *>(base + CODES_HEADER_BYTES),
                                     reinterpret_cast<uint32_t *>(base), s));
    return prof.end();
}

// guide statistics + the kernels of aggregate_and_wta with the guided filter; d_codes (optional) holds the codes of
// layers [d_begin, d_end) of d_dsi
static int aggregate_guided_dev(const float *d_dsi, const void *d_codes, const float *d_guide, int H, int W, int dmax,
                                const StereoParams *p, int d_begin, int d_end, void *ws, size_t ws_bytes,
                                uint64_t *d_keys, hipStream_t s) {
    FusedWs L = fused_ws_layout(H, W, p->wct_radius);
    if (ws_bytes < L.total) { set_error("workspace too small: %zu < %zu", ws_bytes, L.total); return STEREO_EINVAL; }
    unsigned char *base = static_cast<unsigned char *>(ws);
    double *mean = reinterpret_cast<double *>(base + L.off_mean);
    double *invden = reinterpret_cast<double *>(base + L.off_invden);
    float *gmax = reinterpret_cast<float *>(base + L.off_gmax);
    SHP_TRY(launch_guide_stats(d_guide, STEREO_DT_F32, H, W, p->guided_radius, p->guided_eps, mean, invden, gmax, s));
    if (d_begin == d_end) return launch_fill_keys(d_keys, (size_t)H * W, s);
    ProfScope prof(s);
    SHP_TRY(prof.begin());
    const uint32_t *code_stats = nullptr;
    if (d_codes && aggregate_from_codes_applies(p, d_end - d_begin) && g_force_kernel != 1) {
        // The two-column kernel on the cost codes (its fast or its precise variant by eps against the guide's
        // range, chosen on the device), then the general kernel with the opposite test on the encoder's words:
        // it runs only for a volume whose costs are not whole numbers of a common unit, or too large to pack.
        const unsigned char *cb = static_cast<const unsigned char *>(d_codes);
        code_stats = reinterpret_cast<const uint32_t *>(cb);
        CensusCfg cfg;
        memset(&cfg, 0, sizeof(cfg));
        cfg.exact = 1; cfg.unit = 1.f; cfg.large = 1.f;
        const int slices = fused_pair_slices(H, W, p->guided_radius, false, d_end - d_begin);
        if (slices > 1) SHP_TRY(launch_fill_keys(d_keys, (size_t)H * W, s));
        const bool forced_precise = g_force_kernel == 3;
        SHP_TRY(launch_fused_guided_pair(nullptr, nullptr, cfg, d_guide, mean, invden, gmax, H, W, dmax, p->guided_radius,
                                         d_begin, d_end, slices, d_keys, nullptr, s, nullptr, (float)p->guided_eps,
                                         forced_precise ? 0.f : (float)FAST_KERNEL_MIN_EPS, true, 0.f,
                                         reinterpret_cast<const uint16_t *>(cb + CODES_HEADER_BYTES), code_stats, d_begin));
    }
    SHP_TRY(launch_fused_guided(nullptr, nullptr, nullptr, d_dsi, d_guide, STEREO_DT_F32, mean, invden, H, W, dmax,
                                p->guided_radius, d_begin, d_end, d_keys, nullptr, 0, s, nullptr, 0.f, 0.f, code_stats));
    return prof.end();
}

static int check_aggregate_args(const float *d_dsi, const float *d_guide, int H, int W, int dmax, const StereoParams *p,
                                int d_begin, int d_end, void *ws, uint64_t *d_keys) {
    if (!d_dsi || !d_guide || !p || !ws || !d_keys) { set_error("null pointer"); return STEREO_EINVAL; }
    SHP_TRY(check_params(p));
    SHP_TRY(check_image_dims(H, W));
    if (dmax <= 0) { set_error("dsi must have at least one disparity layer"); return STEREO_EINVAL; }
    if (d_begin < 0 || d_end > dmax || d_begin > d_end) { set_error("bad disparity range"); return STEREO_EINVAL; }
    if (p->filter_kind != STEREO_FILTER_GUIDED || !fused_guided_supported(p->guided_radius)) {
        set_error("the device entry point aggregates with the guided filter, radius 1..8");
        return STEREO_EUNSUPPORTED;
    }
    return STEREO_OK;
}

int stereo_aggregate_workspace_bytes(int H, int W, int dmax, const StereoParams *p, size_t *bytes) {
    if (!p || !bytes) { set_error("NULL argument"); return STEREO_EINVAL; }
    SHP_TRY(check_image_dims(H, W));
    if (dmax <= 0) { set_error("dsi must have at least one disparity layer"); return STEREO_EINVAL; }
    size_t total = align_up(fused_ws_layout(H, W, p->wct_radius).total, 256);
    if (aggregate_from_codes_applies(p, 0)) total += CODES_HEADER_BYTES + cost_codes_bytes(H, W, dmax);
    *bytes = total;
    return STEREO_OK;
}

int stereo_aggregate_wta_dev(const float *d_dsi, const float *d_guide, int H, int W, int dmax, const StereoParams *p,
                             int d_begin, int d_end, void *ws, size_t ws_bytes, uint64_t *d_keys, void *stream) {
    SHP_TRY(check_aggregate_args(d_dsi, d_guide, H, W, dmax, p, d_begin, d_end, ws, d_keys));
    hipStream_t s = static_cast<hipStream_t>(stream);
    // a workspace of stereo_aggregate_workspace_bytes holds the cost codes behind the fused workspace; a smaller
    // one (stereo_fused_workspace_bytes) makes the general kernel read the volume in place
    const size_t fused_total = align_up(fused_ws_layout(H, W, p->wct_radius).total, 256);
    const size_t codes_total = CODES_HEADER_BYTES + cost_codes_bytes(H, W, d_end - d_begin);
    void *codes = nullptr;
    if (aggregate_from_codes_applies(p, d_end - d_begin) && d_begin < d_end && ws_bytes >= fused_total + codes_total) {
        codes = static_cast<unsigned char *>(ws) + fused_total;
        SHP_TRY(stereo_encode_cost_volume_dev(d_dsi, H, W, dmax, d_begin, d_end, codes, codes_total, stream));
    }
    return aggregate_guided_dev(d_dsi, codes, d_guide, H, W, dmax, p, d_begin, d_end, ws, ws_bytes, d_keys, s);
}

int stereo_aggregate_wta_codes_dev(const void *d_codes, const float *d_dsi, const float *d_guide, int H, int W, int dmax,
                                   const StereoParams *p, int d_begin, int d_end, void *ws, size_t ws_bytes,
                                   uint64_t *d_keys, void *stream) {
    if (!d_codes) { set_error("null pointer"); return STEREO_EINVAL; }
    SHP_TRY(check_aggregate_args(d_dsi, d_guide, H, W, dmax, p, d_begin, d_end, ws, d_keys));
    return aggregate_guided_dev(d_dsi, d_codes, d_guide, H, W, dmax, p, d_begin, d_end, ws, ws_bytes, d_keys,
                                static_cast<hipStream_t>(stream));
}

int stereo_compute_disparity_noalloc_dev(const float *d_left, const float *d_right, int H, int W, int dmax,
                                         const StereoParams *p, int d_begin, int d_end, uint64_t *d_keys,
                                         void *stream) {
    if (!d_left || !d_right || !p || !d_keys) { set_error("null pointer"); return STEREO_EINVAL; }
    SHP_TRY(check_params(p));
    SHP_TRY(check_image_dims(H, W));
    if (dmax <= 0) { set_error("dmax must be a positive integer"); return STEREO_EINVAL; }
    if (d_begin < 0 || d_end > dmax || d_begin > d_end) {
        set_error("disparity range [%d, %d) outside [0, %d)", d_begin, d_end, dmax);
        return STEREO_EINVAL;
    }
    if ((size_t)H * (size_t)W > (size_t)STEREO_NOALLOC_MAX_PIXELS) {
        set_error("frame of %d x %d = %zu pixels exceeds the %d pixels the static scratch of the no-allocation entry "
                  "point holds: use stereo_compute_disparity_dev with a workspace", H, W, (size_t)H * W,
                  STEREO_NOALLOC_MAX_PIXELS);
        return STEREO_EINVAL;
    }
    {
        CensusCfg cfg;   // (census radius / weights are judged before the first device call)
        SHP_TRY(make_census_cfg(p->wct_radius, p->base_weight, &cfg));
    }
    const size_t need = fused_ws_layout(H, W, p->wct_radius).total;
    if (need > NOALLOC_SCRATCH_BYTES) { set_error("static scratch too small: %zu > %zu", need, NOALLOC_SCRATCH_BYTES); return STEREO_EINVAL; }
    void *ptr = nullptr;
    SHP_TRY(noalloc_scratch_for_current_device(&ptr));
    // the same kernels as stereo_compute_disparity_dev, with the device address of this GPU's array as workspace;
    // every word of it that a kernel reads is written (or cleared) by an earlier kernel of the same call
    return fused_disparity_dev(d_left, d_right, H, W, dmax, p, d_begin, d_end, ptr, NOALLOC_SCRATCH_BYTES, d_keys, nullptr,
                               static_cast<hipStream_t>(stream));
}

int stereo_progress_plan(int H, int W, const StereoParams *p, int n_bands, int32_t *row_end, int *bands_out) {
    if (!p) { set_error("null pointer"); return STEREO_EINVAL; }
    SHP_TRY(check_params(p));
    SHP_TRY(check_image_dims(H, W));
    if (!row_end || !bands_out || n_bands < 1 || n_bands > PROGRESS_MAX_BANDS) {
        set_error("1..%d bands and non-null outputs", PROGRESS_MAX_BANDS);
        return STEREO_EINVAL;
    }
    CensusCfg cfg;
    SHP_TRY(make_census_cfg(p->wct_radius, p->base_weight, &cfg));
    const int th = progress_tile_rows(p, cfg);
    int n = 0;
    if (th > 0) {
        const int tile_rows = ceil_div(H, th), per_band = ceil_div(tile_rows, n_bands);
        for (int first = 0; first < tile_rows; first += per_band) row_end[n++] = std::min(H, (first + per_band) * th);
    } else {
        n_bands = std::min(n_bands, H);
        for (int b = 0; b < n_bands; ++b) row_end[n++] = (int)((int64_t)(b + 1) * H / n_bands);
    }
    *bands_out = n;
    return STEREO_OK;
}

int stereo_compute_disparity_progress_dev(const float *d_left, const float *d_right, int H, int W, int dmax,
                                          const StereoParams *p, int d_begin, int d_end, void *ws, size_t ws_bytes,
                                          uint64_t *d_keys, int n_bands, uint32_t *d_progress, uint32_t epoch,
                                          void *stream) {
    if (!d_left || !d_right || !p || !ws || !d_keys || !d_progress) { set_error("null pointer"); return STEREO_EINVAL; }
    if (n_bands < 1 || n_bands > PROGRESS_MAX_BANDS) { set_error("1..%d bands", PROGRESS_MAX_BANDS); return STEREO_EINVAL; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    SHP_TRY(check_params(p));
    CensusCfg cfg;
    SHP_TRY(make_census_cfg(p->wct_radius, p->base_weight, &cfg));
    BandProgress pg;
    pg.words = d_progress;
    pg.epoch = epoch;
    const int th = progress_tile_rows(p, cfg);
    pg.band_tile_rows = th > 0 ? ceil_div(ceil_div(H, th), n_bands) : 0;
    // the bands the kernel will count must be the ones the plan announced
    if (th > 0 && ceil_div(ceil_div(H, th), pg.band_tile_rows) != n_bands) {
        set_error("n_bands = %d does not come from stereo_progress_plan for this frame", n_bands);
        return STEREO_EINVAL;
    }
    SHP_HIP(hipMemsetAsync(d_progress, 0, sizeof(uint32_t) * PROGRESS_MAX_BANDS, s));
    int rc = fused_disparity_dev(d_left, d_right, H, W, dmax, p, d_begin, d_end, ws, ws_bytes, d_keys, nullptr, s, &pg);
    // whatever ran (or failed to launch), nobody may be left waiting on this epoch
    int rc2 = launch_progress_finish(d_progress, n_bands, epoch, s);
    return rc != STEREO_OK ? rc : rc2;
}

int stereo_streams_overlap(void *stream_a, void *stream_b, int *overlap) {
    if (!overlap) { set_error("null pointer"); return STEREO_EINVAL; }
    bool o = false;
    SHP_TRY(streams_overlap(static_cast<hipStream_t>(stream_a), static_cast<hipStream_t>(stream_b), &o));
    *overlap = o ? 1 : 0;
    return STEREO_OK;
}

int stereo_wait_progress_dev(const uint32_t *d_progress, int band, uint32_t epoch, void *stream) {
    if (!d_progress) { set_error("null pointer"); return STEREO_EINVAL; }
    return launch_progress_wait(d_progress, band, epoch, static_cast<hipStream_t>(stream));
}

int stereo_hip_release_cache(void) {
    pool_release_all();
    return STEREO_OK;
}

int stereo_debug_force_general_kernel(int on) {
    g_force_kernel = on;
    return STEREO_OK;
}

int stereo_profile_begin(int max_launches) {
    if (max_launches < 0) { set_error("max_launches must be >= 0"); return STEREO_EINVAL; }
    while ((int)g_prof.start.size() < max_launches) {
        hipEvent_t a, b;
        SHP_HIP(hipEventCreate(&a));
        SHP_HIP(hipEventCreate(&b));
        g_prof.start.push_back(a);
        g_prof.stop.push_back(b);
    }
    g_prof.used = 0;
    g_prof.enabled = max_launches > 0;
    return STEREO_OK;
}

int stereo_profile_end(double *total_ms, int *launches) {
    if (!total_ms || !launches) { set_error("null pointer"); return STEREO_EINVAL; }
    double total = 0.0;
    for (int i = 0; i < g_prof.used; ++i) {
        SHP_HIP(hipEventSynchronize(g_prof.stop[i]));
        float ms = 0.f;
        SHP_HIP(hipEventElapsedTime(&ms, g_prof.start[i], g_prof.stop[i]));
        total += (double)ms;
    }
    if (total_ms) *total_ms = total;
    if (launches) *launches = g_prof.used;
    g_prof.enabled = false;
    g_prof.used = 0;
    return STEREO_OK;
}

int stereo_merge_keys_dev(uint64_t *d_keys, const uint64_t *d_other, size_t n, void *stream) {
    if (n && (!d_keys || !d_other)) { set_error("null pointer"); return STEREO_EINVAL; }
    return launch_merge_keys(d_keys, d_other, n, static_cast<hipStream_t>(stream));
}
int stereo_merge_key_slices_dev(uint64_t *d_slices, size_t chunk, int n_slices, void *stream) {
    if (!d_slices || n_slices < 1) { set_error("null pointer or no slices"); return STEREO_EINVAL; }
    return launch_merge_key_slices(d_slices, chunk, n_slices, static_cast<hipStream_t>(stream));
}
int stereo_keys_to_i64_dev(const uint64_t *d_keys, int64_t *d_out, size_t n, void *stream) {
    if (n && (!d_keys || !d_out)) { set_error("null pointer"); return STEREO_EINVAL; }
    return launch_keys_to_i64(d_keys, d_out, n, static_cast<hipStream_t>(stream));
}
int stereo_keys_from_i64_dev(const int64_t *d_in, uint64_t *d_keys, size_t n, void *stream) {
    if (n && (!d_in || !d_keys)) { set_error("null pointer"); return STEREO_EINVAL; }
    return launch_keys_from_i64(d_in, d_keys, n, static_cast<hipStream_t>(stream));
}
int stereo_unpack_keys_dev(const uint64_t *d_keys, size_t n, int32_t *d_disp, float *d_cost, void *stream) {
    if (n && (!d_keys || !d_disp || !d_cost)) { set_error("null pointer"); return STEREO_EINVAL; }
    return launch_unpack_keys(d_keys, n, d_disp, d_cost, static_cast<hipStream_t>(stream));
}

// Streams of the host entry point that overlaps its result copy with the search: one per device for the
// kernels, one for waits, unpacking and copies; created at first use (one caller at a time per device).
struct HostOverlap {
    hipStream_t compute = nullptr, copy = nullptr;
    hipEvent_t words_cleared = nullptr;   // orders the copy stream's first wait behind the clearing of the progress words
    uint32_t epoch = 0;                   // one per call: a recycled words buffer never holds a flag of the current call
};
static HostOverlap *host_overlap_for_current_device() {
    static HostOverlap table[64];
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    HostOverlap &h = table[dev];
    if (!h.compute) {
        hipStream_t a = nullptr, b = nullptr;
        hipEvent_t e = nullptr;
        if (hipStreamCreateWithFlags(&a, hipStreamNonBlocking) != hipSuccess) return nullptr;
        // The copy stream must not share a hardware queue with the compute stream (its band waits would start
        // after the search): of up to eight new streams the first one whose work runs beside the compute stream's
        // is kept (streams_overlap); if none does, the last one serves -- correct, merely without the overlap.
        // (rejected candidates stay alive until the choice is made: a destroyed stream's queue slot could be handed
        //  to the next candidate)
        std::vector<hipStream_t> rejected;
        for (int attempt = 0; attempt < 8; ++attempt) {
            hipStream_t c = nullptr;
            if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) break;
            bool overlap = false;
            const bool probed = streams_overlap(a, c, &overlap) == STEREO_OK;
            if (b) rejected.push_back(b);
            b = c;
            if (!probed || overlap) break;
        }
        for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
        if (!b) { (void)hipStreamDestroy(a); return nullptr; }
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            (void)hipStreamDestroy(a); (void)hipStreamDestroy(b);
            return nullptr;
        }
        h.compute = a; h.copy = b; h.words_cleared = e;
    }
    return &h;
}

int stereo_compute_disparity_host(const void *left, const void *right, int dtype, int H, int W, int dmax,
                                  const StereoParams *p, int d_begin, int d_end, int32_t *out_disp,
                                  float *out_cost) {
    if (!left || !right || !p || !out_disp || !out_cost) { set_error("null pointer"); return STEREO_EINVAL; }
    SHP_TRY(check_params(p));
    SHP_TRY(check_image_dims(H, W));
    if (dmax <= 0) { set_error("dmax must be a positive integer"); return STEREO_EINVAL; }
    if (d_begin < 0 || d_end > dmax || d_begin > d_end) { set_error("bad disparity range"); return STEREO_EINVAL; }
    if (dtype != STEREO_DT_F32 && dtype != STEREO_DT_F64 && dtype != STEREO_DT_U8) {
        set_error("unsupported image dtype %d", dtype);
        return STEREO_EINVAL;
    }
    CensusCfg cfg;
    SHP_TRY(make_census_cfg(p->wct_radius, p->base_weight, &cfg));
    const size_t n = (size_t)H * W;
    DevBuf keys;
    SHP_TRY(keys.alloc(n * 8));
    const bool fused = fused_path_applies(p) && dtype != STEREO_DT_F64 && d_end - d_begin <= FUSED_MAX_DISPARITIES;
    if (fused) {
        std::vector<float> lf, rf;
        const float *hl = static_cast<const float *>(left), *hr = static_cast<const float *>(right);
        if (dtype == STEREO_DT_U8) {  // uint8 -> float32 is exact and order preserving
            lf.resize(n); rf.resize(n);
            const uint8_t *l8 = static_cast<const uint8_t *>(left), *r8 = static_cast<const uint8_t *>(right);
            for (size_t i = 0; i < n; ++i) { lf[i] = (float)l8[i]; rf[i] = (float)r8[i]; }
            hl = lf.data(); hr = rf.data();
        }
        FusedWs L = fused_ws_layout(H, W, p->wct_radius);
        DevBuf dl, dr, ws;
        SHP_TRY(dl.alloc(n * 4)); SHP_TRY(dr.alloc(n * 4)); SHP_TRY(ws.alloc(L.total));
        SHP_HIP(hipMemcpy(dl.p, hl, n * 4, hipMemcpyHostToDevice));
        SHP_HIP(hipMemcpy(dr.p, hr, n * 4, hipMemcpyHostToDevice));
        // Large guided frames: the kernel reports finished row bands, and a second stream unpacks each band and
        // copies it to the caller while the kernel works on the bands below (the device-to-host copy of the two
        // result planes takes as long as the search itself).
        HostOverlap *ov = (n >= ((size_t)1 << 20) && d_begin < d_end && progress_tile_rows(p, cfg) > 0)
                              ? host_overlap_for_current_device() : nullptr;
        if (ov) {
            int32_t row_end[PROGRESS_MAX_BANDS];
            int n_bands = 0;
            SHP_TRY(stereo_progress_plan(H, W, p, 8, row_end, &n_bands));
            DevBuf words, disp, cost;
            SHP_TRY(words.alloc(sizeof(uint32_t) * 3 * PROGRESS_MAX_BANDS));
            SHP_TRY(disp.alloc(n * 4)); SHP_TRY(cost.alloc(n * 4));
            // The words come from the recycled pool: they are cleared before anything polls them (the copy
            // stream's first wait is ordered behind the clearing by an event), and the epoch differs from every
            // earlier call's, so a flag left in the buffer can never pass for this call's.
            const uint32_t epoch = ++ov->epoch ? ov->epoch : ++ov->epoch;
            SHP_HIP(hipMemsetAsync(words.p, 0, sizeof(uint32_t) * 3 * PROGRESS_MAX_BANDS, ov->compute));
            SHP_HIP(hipEventRecord(ov->words_cleared, ov->compute));
            SHP_HIP(hipStreamWaitEvent(ov->copy, ov->words_cleared, 0));
            int rc = stereo_compute_disparity_progress_dev(dl.as<float>(), dr.as<float>(), H, W, dmax, p, d_begin, d_end,
                                                           ws.p, L.total, keys.as<uint64_t>(), n_bands,
                                                           words.as<uint32_t>(), epoch, ov->compute);
            for (int b = 0, y0 = 0; rc == STEREO_OK && b < n_bands; y0 = row_end[b], ++b) {
                const size_t first = (size_t)y0 * W, count = (size_t)(row_end[b] - y0) * W;
                rc = launch_progress_wait(words.as<uint32_t>(), b, epoch, ov->copy);
                if (rc == STEREO_OK)
                    rc = launch_unpack_keys(keys.as<uint64_t>() + first, count, disp.as<int32_t>() + first,
                                            cost.as<float>() + first, ov->copy);
                if (rc == STEREO_OK && (hipMemcpyAsync(out_disp + first, disp.as<int32_t>() + first, count * 4,
                                                       hipMemcpyDeviceToHost, ov->copy) != hipSuccess ||
                                        hipMemcpyAsync(out_cost + first, cost.as<float>() + first, count * 4,
                                                       hipMemcpyDeviceToHost, ov->copy) != hipSuccess)) {
                    set_error("device-to-host copy of a row band failed");
                    rc = STEREO_EHIP;
                }
            }
            // both streams drain before the buffers go back to the pool, whatever happened above
            const hipError_t e1 = hipStreamSynchronize(ov->copy), e2 = hipStreamSynchronize(ov->compute);
            SHP_TRY(rc);
            SHP_HIP(e1); SHP_HIP(e2);
            // a band wait that gave up shipped rows that were not final: that is an error, not a result
            uint32_t timed_out[PROGRESS_MAX_BANDS];
            SHP_HIP(hipMemcpy(timed_out, words.as<uint32_t>() + 2 * PROGRESS_MAX_BANDS, sizeof(timed_out),
                              hipMemcpyDeviceToHost));
            for (int b = 0; b < n_bands; ++b)
                if (timed_out[b] != 0) {
                    set_error("the wait for row band %d of %d timed out: the rows copied out were not final", b, n_bands);
                    return STEREO_EHIP;
                }
        } else {
            SHP_TRY(fused_disparity_dev(dl.as<float>(), dr.as<float>(), H, W, dmax, p, d_begin, d_end, ws.p, L.total,
                                        keys.as<uint64_t>(), nullptr, nullptr));
            SHP_TRY(finish_keys_to_host(keys.as<uint64_t>(), n, out_disp, out_cost));
        }
        if (p->filter_kind == STEREO_FILTER_GUIDED && d_begin < d_end) {
            bool finite = true;
            SHP_TRY(guide_is_finite(static_cast<unsigned char *>(ws.p) + L.off_gmax, &finite));
            if (!finite) poison_disparity(hl, STEREO_DT_F32, H, W, p->guided_radius, out_disp, out_cost);
        }
        return STEREO_OK;
    }
    // general path: cost layers on the device, then one filter launch chain per layer
    const size_t img_bytes = n * dtype_size(dtype), word_bytes = n * (cfg.word64 ? 8 : 4);
    DevBuf dl, dr, lw, rw, layers, guide;
    SHP_TRY(dl.alloc(img_bytes)); SHP_TRY(dr.alloc(img_bytes));
    SHP_TRY(lw.alloc(word_bytes)); SHP_TRY(rw.alloc(word_bytes));
    SHP_TRY(layers.alloc(n * (size_t)(d_end - d_begin) * 4));
    SHP_HIP(hipMemcpy(dl.p, left, img_bytes, hipMemcpyHostToDevice));
    SHP_HIP(hipMemcpy(dr.p, right, img_bytes, hipMemcpyHostToDevice));
    SHP_TRY(launch_census_pack_pair(dl.p, dr.p, dtype, H, W, p->wct_radius, lw.p, rw.p, nullptr));
    SHP_TRY(launch_dsi_layers(lw.p, rw.p, H, W, d_begin, d_end, cfg, layers.as<float>(), nullptr));
    const void *g = dl.p;
    int gdt = dtype;
    if (dtype == STEREO_DT_U8) {  // the guide is the left image as float64 (guided_filter.py:202)
        std::vector<float> lf(n);
        const uint8_t *l8 = static_cast<const uint8_t *>(left);
        for (size_t i = 0; i < n; ++i) lf[i] = (float)l8[i];
        SHP_TRY(guide.alloc(n * 4));
        SHP_HIP(hipMemcpy(guide.p, lf.data(), n * 4, hipMemcpyHostToDevice));
        g = guide.p;
        gdt = STEREO_DT_F32;
    }
    SHP_TRY(layered_aggregate_dev(layers.as<float>(), g, gdt, H, W, p, d_begin, d_end, keys.as<uint64_t>(), nullptr));
    return finish_keys_to_host(keys.as<uint64_t>(), n, out_disp, out_cost);
}

int stereo_aggregate_wta_host(const float *dsi, const void *guide, int guide_dtype, int H, int W, int dmax,
                              const StereoParams *p, int d_begin, int d_end, int32_t *out_disp, float *out_cost) {
    if (!dsi || !guide || !p || !out_disp || !out_cost) { set_error("null pointer"); return STEREO_EINVAL; }
    SHP_TRY(check_params(p));
    SHP_TRY(check_image_dims(H, W));
    if (dmax <= 0) { set_error("dsi must have at least one disparity layer"); return STEREO_EINVAL; }
    if (d_begin < 0 || d_end > dmax || d_begin > d_end) { set_error("bad disparity range"); return STEREO_EINVAL; }
    if (guide_dtype != STEREO_DT_F32 && guide_dtype != STEREO_DT_F64) {
        set_error("guide dtype must be float32 or float64");
        return STEREO_EINVAL;
    }
    size_t n = (size_t)H * W;
    DevBuf ddsi, dg, mean, invden, keys, layers, gmax;
    bool check_guide = false;
    SHP_TRY(ddsi.alloc(n * (size_t)dmax * 4));
    SHP_TRY(dg.alloc(n * dtype_size(guide_dtype)));
    SHP_TRY(keys.alloc(n * 8));
    SHP_HIP(hipMemcpy(ddsi.p, dsi, n * (size_t)dmax * 4, hipMemcpyHostToDevice));
    SHP_HIP(hipMemcpy(dg.p, guide, n * dtype_size(guide_dtype), hipMemcpyHostToDevice));
    bool fused_guided = p->filter_kind == STEREO_FILTER_GUIDED && fused_guided_supported(p->guided_radius);
    if (fused_guided) {
        // The reference's integral images spread a NaN or infinite cost over the whole quadrant below and to the
        // right of it, layer by layer; the fused kernel's window sums would keep it local.  Such a volume takes
        // the layer-at-a-time path, whose box sums are integral images too.
        DevBuf flag;
        SHP_TRY(flag.alloc(4));
        SHP_HIP(hipMemsetAsync(flag.p, 0, 4, nullptr));
        const size_t nv = n * (size_t)dmax;
        hipLaunchKernelGGL(nonfinite_flag_kernel, dim3((unsigned)std::min<size_t>((nv + 255) / 256, 8192)), dim3(256), 0,
                           nullptr, ddsi.as<float>(), nv, flag.as<uint32_t>());
        SHP_LAUNCH_CHECK();
        uint32_t h = 0;
        SHP_HIP(hipMemcpy(&h, flag.p, 4, hipMemcpyDeviceToHost));
        if (h) fused_guided = false;
    }
    DevBuf ws, codes;
    size_t gmax_off = 0;
    if (fused_guided && guide_dtype == STEREO_DT_F32) {
        // float32 guide: the device path of the two-call API (cost codes + the two-column kernel where they apply)
        const FusedWs L = fused_ws_layout(H, W, p->wct_radius);
        gmax_off = L.off_gmax;
        SHP_TRY(ws.alloc(L.total));
        void *codes_p = nullptr;
        if (aggregate_from_codes_applies(p, d_end - d_begin) && d_begin < d_end) {
            const size_t cbytes = CODES_HEADER_BYTES + cost_codes_bytes(H, W, d_end - d_begin);
            SHP_TRY(codes.alloc(cbytes));
            codes_p = codes.p;
            SHP_TRY(stereo_encode_cost_volume_dev(ddsi.as<float>(), H, W, dmax, d_begin, d_end, codes_p, cbytes, nullptr));
        }
        SHP_TRY(aggregate_guided_dev(ddsi.as<float>(), codes_p, dg.as<float>(), H, W, dmax, p, d_begin, d_end, ws.p, L.total,
                                     keys.as<uint64_t>(), nullptr));
        check_guide = d_begin < d_end;
    } else if (fused_guided) {
        SHP_TRY(mean.alloc(n * 8)); SHP_TRY(invden.alloc(n * 8)); SHP_TRY(gmax.alloc(8));
        SHP_TRY(launch_guide_stats(dg.p, guide_dtype, H, W, p->guided_radius, p->guided_eps, mean.as<double>(),
                                   invden.as<double>(), gmax.p, nullptr));
        check_guide = d_begin < d_end;
        if (d_begin == d_end) SHP_TRY(launch_fill_keys(keys.as<uint64_t>(), n, nullptr));
        else
            SHP_TRY(launch_fused_guided(nullptr, nullptr, nullptr, ddsi.as<float>(), dg.p, guide_dtype,
                                        mean.as<double>(), invden.as<double>(), H, W, dmax, p->guided_radius,
                                        d_begin, d_end, keys.as<uint64_t>(), nullptr, 0, nullptr));
    } else {
        SHP_TRY(layers.alloc(n * (size_t)dmax * 4));
        SHP_TRY(launch_dsi_to_layers(ddsi.as<float>(), n, dmax, layers.as<float>(), nullptr));
        ddsi.release();
        SHP_TRY(layered_aggregate_dev(layers.as<float>() + (size_t)d_begin * n, dg.p, guide_dtype, H, W, p, d_begin,
                                      d_end, keys.as<uint64_t>(), nullptr));
    }
    SHP_TRY(finish_keys_to_host(keys.as<uint64_t>(), n, out_disp, out_cost));
    if (check_guide) {
        bool finite = true;
        SHP_TRY(guide_is_finite(ws.p ? static_cast<const unsigned char *>(ws.p) + gmax_off : gmax.p, &finite));
        if (!finite) poison_disparity(guide, guide_dtype, H, W, p->guided_radius, out_disp, out_cost);
    }
    return STEREO_OK;
}

int stereo_guided_filter_host(const double *guide, const float *src, int H, int W, int radius, double eps,
                              float *out) {
    if (!guide || !src || !out) { set_error("null pointer"); return STEREO_EINVAL; }
    SHP_TRY(check_image_dims(H, W));
    if (radius <= 0) { set_error("radius must be a positive integer"); return STEREO_EINVAL; }
    if (!(eps > 0)) { set_error("eps must be positive"); return STEREO_EINVAL; }
    size_t n = (size_t)H * W;
    DevBuf dg, ds, q, gmax;
    SHP_TRY(dg.alloc(n * 8)); SHP_TRY(ds.alloc(n * 4)); SHP_TRY(q.alloc(n * 4));
    SHP_HIP(hipMemcpy(dg.p, guide, n * 8, hipMemcpyHostToDevice));
    SHP_HIP(hipMemcpy(ds.p, src, n * 4, hipMemcpyHostToDevice));
    if (fused_guided_supported(radius)) {
        DevBuf mean, invden, keys;
        SHP_TRY(mean.alloc(n * 8)); SHP_TRY(invden.alloc(n * 8)); SHP_TRY(keys.alloc(n * 8)); SHP_TRY(gmax.alloc(8));
        SHP_TRY(launch_guide_stats(dg.p, STEREO_DT_F64, H, W, radius, eps, mean.as<double>(), invden.as<double>(),
                                   gmax.p, nullptr));
        SHP_TRY(launch_fused_guided(nullptr, nullptr, nullptr, ds.as<float>(), dg.p, STEREO_DT_F64, mean.as<double>(),
                                    invden.as<double>(), H, W, 1, radius, 0, 1, keys.as<uint64_t>(), q.as<float>(), 0,
                                    nullptr));
    } else {
        DevBuf ws;
        SHP_TRY(ws.alloc(layered_guided_ws_bytes(H, W)));
        SHP_TRY(layered_guided_prepare(dg.p, STEREO_DT_F64, H, W, radius, ws.p, nullptr));
        SHP_TRY(layered_guided_apply(ds.as<float>(), H, W, radius, eps, ws.p, q.as<float>(), nullptr));
    }
    SHP_HIP(hipMemcpy(out, q.p, n * 4, hipMemcpyDeviceToHost));
    // A NaN or infinite pixel of the guide or of the source reaches, through the reference's integral images,
    // every output below and to the right of it (from 2 * radius rows / columns before it).
    auto fill = [&](int y, int x0) {
        for (int x = x0; x < W; ++x) out[(size_t)y * W + x] = NAN;
    };
    bool guide_finite = true;
    if (gmax.p) SHP_TRY(guide_is_finite(gmax.p, &guide_finite));
    else
        for (size_t i = 0; i < n && guide_finite; ++i) guide_finite = std::fabs(guide[i]) <= 1.7976931348623157e308;
    if (!guide_finite) poison_like_reference(guide, H, W, radius, fill);
    bool src_finite = true;
    for (size_t i = 0; i < n && src_finite; ++i) src_finite = std::fabs(src[i]) <= 3.402823466e38f;
    if (!src_finite) poison_like_reference(src, H, W, radius, fill);
    return STEREO_OK;
}

// ---- guided filter with the guide's statistics kept on the device ---------------------------------------------
// prepare_guided_filter / guided_filter_with_precompute (guided_filter.py:182-253): the reference computes the
// guide's statistics once per scene and then filters layer after layer (stereo.py:143-150).  The plan is that
// split on the device: guide, box mean and 1 / (var + eps) stay resident, a call moves one layer in and one out.
struct GuidedPlan {
    int H = 0, W = 0, radius = 0, device = 0;
    double eps = 0.0;
    bool fused = false, guide_finite = true;
    DevBuf guide, mean, invden, gmax, src, out, keys, ws, flag;
    std::vector<int> poison_first;   // per row: first column the reference's integral images turn into NaN (INT32_MAX: none)
};

int stereo_guided_plan_create(const double *guide, int H, int W, int radius, double eps, void **plan_out) {
    if (!guide || !plan_out) { set_error("null pointer"); return STEREO_EINVAL; }
    *plan_out = nullptr;
    SHP_TRY(check_image_dims(H, W));
    if (radius <= 0) { set_error("radius must be a positive integer"); return STEREO_EINVAL; }
    if (!(eps > 0)) { set_error("eps must be positive"); return STEREO_EINVAL; }
    std::unique_ptr<GuidedPlan> p(new (std::nothrow) GuidedPlan);
    if (!p) { set_error("out of host memory"); return STEREO_ENOMEM; }
    p->H = H; p->W = W; p->radius = radius; p->eps = eps;
    SHP_HIP(hipGetDevice(&p->device));
    const size_t n = (size_t)H * W;
    SHP_TRY(p->guide.alloc(n * 8)); SHP_TRY(p->src.alloc(n * 4)); SHP_TRY(p->out.alloc(n * 4)); SHP_TRY(p->flag.alloc(4));
    SHP_HIP(hipMemcpy(p->guide.p, guide, n * 8, hipMemcpyHostToDevice));
    p->fused = fused_guided_supported(radius);
    if (p->fused) {
        SHP_TRY(p->mean.alloc(n * 8)); SHP_TRY(p->invden.alloc(n * 8)); SHP_TRY(p->keys.alloc(n * 8)); SHP_TRY(p->gmax.alloc(8));
        SHP_TRY(launch_guide_stats(p->guide.p, STEREO_DT_F64, H, W, radius, eps, p->mean.as<double>(), p->invden.as<double>(),
                                   p->gmax.p, nullptr));
        SHP_TRY(guide_is_finite(p->gmax.p, &p->guide_finite));
    } else {
        SHP_TRY(p->ws.alloc(layered_guided_ws_bytes(H, W)));
        SHP_TRY(layered_guided_prepare(p->guide.p, STEREO_DT_F64, H, W, radius, p->ws.p, nullptr));
        SHP_HIP(hipDeviceSynchronize());
        for (size_t i = 0; i < n && p->guide_finite; ++i) p->guide_finite = std::fabs(guide[i]) <= 1.7976931348623157e308;
    }
    if (!p->guide_finite) {
        p->poison_first.assign(H, INT32_MAX);
        poison_like_reference(guide, H, W, radius, [&](int y, int x0) { p->poison_first[y] = x0; });
    }
    *plan_out = p.release();
    return STEREO_OK;
}

int stereo_guided_plan_apply(void *plan, const float *src, float *out) {
    if (!plan || !src || !out) { set_error("null pointer"); return STEREO_EINVAL; }
    GuidedPlan *p = static_cast<GuidedPlan *>(plan);
    int dev = 0;
    SHP_HIP(hipGetDevice(&dev));
    if (dev != p->device) { set_error("the plan was made on device %d, the current device is %d", p->device, dev); return STEREO_EINVAL; }
    const int H = p->H, W = p->W;
    const size_t n = (size_t)H * W;
    SHP_HIP(hipMemcpyAsync(p->src.p, src, n * 4, hipMemcpyHostToDevice, nullptr));
    SHP_HIP(hipMemsetAsync(p->flag.p, 0, 4, nullptr));
    hipLaunchKernelGGL(nonfinite_flag_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, nullptr,
                       p->src.as<float>(), n, p->flag.as<uint32_t>());
    SHP_LAUNCH_CHECK();
    if (p->fused)
        SHP_TRY(launch_fused_guided(nullptr, nullptr, nullptr, p->src.as<float>(), p->guide.p, STEREO_DT_F64, p->mean.as<double>(),
                                    p->invden.as<double>(), H, W, 1, p->radius, 0, 1, p->keys.as<uint64_t>(), p->out.as<float>(),
                                    0, nullptr));
    else
        SHP_TRY(layered_guided_apply(p->src.as<float>(), H, W, p->radius, p->eps, p->ws.p, p->out.as<float>(), nullptr));
    SHP_HIP(hipMemcpy(out, p->out.p, n * 4, hipMemcpyDeviceToHost));
    uint32_t src_bad = 0;
    SHP_HIP(hipMemcpy(&src_bad, p->flag.p, 4, hipMemcpyDeviceToHost));
    // (non-finite pixels of the guide or of the source reach, through the reference's integral images, every output
    //  below and to the right of them: see poison_like_reference)
    auto fill = [&](int y, int x0) {
        for (int x = x0; x < W; ++x) out[(size_t)y * W + x] = NAN;
    };
    if (!p->guide_finite)
        for (int y = 0; y < H; ++y)
            if (p->poison_first[y] != INT32_MAX) fill(y, p->poison_first[y]);
    if (src_bad) poison_like_reference(src, H, W, p->radius, fill);
    return STEREO_OK;
}

int stereo_guided_plan_destroy(void *plan) {
    delete static_cast<GuidedPlan *>(plan);   // (the device buffers go back to the library's cache)
    return STEREO_OK;
}

int stereo_guided_prepare_host(const double *guide, int H, int W, int radius, double *mean_guide, double *mean_gg,
                               double *var_g) {
    if (!guide || !mean_guide || !mean_gg || !var_g) { set_error("null pointer"); return STEREO_EINVAL; }
    SHP_TRY(check_image_dims(H, W));
    if (radius <= 0) { set_error("radius must be a positive integer"); return STEREO_EINVAL; }
    size_t n = (size_t)H * W;
    DevBuf dg, ws;
    SHP_TRY(dg.alloc(n * 8)); SHP_TRY(ws.alloc(layered_guided_ws_bytes(H, W)));
    SHP_HIP(hipMemcpy(dg.p, guide, n * 8, hipMemcpyHostToDevice));
    SHP_TRY(layered_guided_prepare(dg.p, STEREO_DT_F64, H, W, radius, ws.p, nullptr));
    double *g64, *mg, *mgg, *var;
    layered_guided_planes(ws.p, H, W, &g64, &mg, &mgg, &var);
    SHP_HIP(hipMemcpy(mean_guide, mg, n * 8, hipMemcpyDeviceToHost));
    SHP_HIP(hipMemcpy(mean_gg, mgg, n * 8, hipMemcpyDeviceToHost));
    SHP_HIP(hipMemcpy(var_g, var, n * 8, hipMemcpyDeviceToHost));
    return STEREO_OK;
}

int stereo_integral_image_host(const double *image, int H, int W, double *out) {
    if (!image || !out) { set_error("null pointer"); return STEREO_EINVAL; }
    SHP_TRY(check_image_dims(H, W));
    size_t n = (size_t)H * W, ni = ((size_t)H + 1) * ((size_t)W + 1);
    DevBuf di, tmp, integ;
    SHP_TRY(di.alloc(n * 8)); SHP_TRY(tmp.alloc(n * 8)); SHP_TRY(integ.alloc(ni * 8));
    SHP_HIP(hipMemcpy(di.p, image, n * 8, hipMemcpyHostToDevice));
    SHP_TRY(launch_integral(di.p, STEREO_DT_F64, H, W, tmp.as<double>(), integ.as<double>(), nullptr));
    SHP_HIP(hipMemcpy(out, integ.p, ni * 8, hipMemcpyDeviceToHost));
    return STEREO_OK;
}

int stereo_box_filter_mean_host(const double *image, int H, int W, int radius, double *out) {
    if (!image || !out) { set_error("null pointer"); return STEREO_EINVAL; }
    SHP_TRY(check_image_dims(H, W));
    if (radius < 0) { set_error("radius must be a non-negative integer"); return STEREO_EINVAL; }
    size_t n = (size_t)H * W, ni = ((size_t)H + 1) * ((size_t)W + 1);
    DevBuf di, tmp, integ, dout;
    SHP_TRY(di.alloc(n * 8)); SHP_TRY(tmp.alloc(n * 8)); SHP_TRY(integ.alloc(ni * 8)); SHP_TRY(dout.alloc(n * 8));
    SHP_HIP(hipMemcpy(di.p, image, n * 8, hipMemcpyHostToDevice));
    SHP_TRY(launch_box_mean(di.p, STEREO_DT_F64, H, W, radius, tmp.as<double>(), integ.as<double>(), dout.as<double>(),
                            nullptr));
    SHP_HIP(hipMemcpy(out, dout.p, n * 8, hipMemcpyDeviceToHost));
    return STEREO_OK;
}

int stereo_box_filter_mean_indexed_host(const double *image, int H, int W, const int32_t *y0, const int32_t *y1,
                                        const int32_t *x0, const int32_t *x1, const double *area, double *out) {
    if (!image || !out) { set_error("null pointer"); return STEREO_EINVAL; }
    SHP_TRY(check_image_dims(H, W));
    for (int y = 0; y < H; ++y)
        if (y0[y] < 0 || y1[y] >= H || y0[y] > H) { set_error("row window out of range"); return STEREO_EINVAL; }
    for (int x = 0; x < W; ++x)
        if (x0[x] < 0 || x1[x] >= W || x0[x] > W) { set_error("column window out of range"); return STEREO_EINVAL; }
    size_t n = (size_t)H * W, ni = ((size_t)H + 1) * ((size_t)W + 1);
    DevBuf di, tmp, integ, dout, dy0, dy1, dx0, dx1, dar;
    SHP_TRY(di.alloc(n * 8)); SHP_TRY(tmp.alloc(n * 8)); SHP_TRY(integ.alloc(ni * 8)); SHP_TRY(dout.alloc(n * 8));
    SHP_TRY(dy0.alloc(H * 4)); SHP_TRY(dy1.alloc(H * 4)); SHP_TRY(dx0.alloc(W * 4)); SHP_TRY(dx1.alloc(W * 4));
    SHP_TRY(dar.alloc(n * 8));
    SHP_HIP(hipMemcpy(di.p, image, n * 8, hipMemcpyHostToDevice));
    SHP_HIP(hipMemcpy(dy0.p, y0, H * 4, hipMemcpyHostToDevice));
    SHP_HIP(hipMemcpy(dy1.p, y1, H * 4, hipMemcpyHostToDevice));
    SHP_HIP(hipMemcpy(dx0.p, x0, W * 4, hipMemcpyHostToDevice));
    SHP_HIP(hipMemcpy(dx1.p, x1, W * 4, hipMemcpyHostToDevice));
    SHP_HIP(hipMemcpy(dar.p, area, n * 8, hipMemcpyHostToDevice));
    SHP_TRY(launch_integral(di.p, STEREO_DT_F64, H, W, tmp.as<double>(), integ.as<double>(), nullptr));
    SHP_TRY(launch_box_from_integral(integ.as<double>(), H, W, dy0.as<int>(), dy1.as<int>(), dx0.as<int>(),
                                     dx1.as<int>(), dar.as<double>(), 0, dout.as<double>(), nullptr));
    SHP_HIP(hipMemcpy(out, dout.p, n * 8, hipMemcpyDeviceToHost));
    return STEREO_OK;
}

int stereo_median_filter_host(const float *image, int H, int W, int radius, int mode, float *out) {
    if (!image || !out) { set_error("null pointer"); return STEREO_EINVAL; }
    SHP_TRY(check_image_dims(H, W));
    if (radius <= 0) { set_error("radius must be a positive integer"); return STEREO_EINVAL; }
    size_t n = (size_t)H * W;
    DevBuf di, dout;
    SHP_TRY(di.alloc(n * 4)); SHP_TRY(dout.alloc(n * 4));
    SHP_HIP(hipMemcpy(di.p, image, n * 4, hipMemcpyHostToDevice));
    if (mode == STEREO_FILTER_MEDIAN) {
        SHP_TRY(launch_median_f32(di.as<float>(), H, W, radius, dout.as<float>(), nullptr));
    } else if (mode == STEREO_FILTER_MEDIAN_U8Q) {
        DevBuf mm, q, sc;
        SHP_TRY(mm.alloc(16)); SHP_TRY(q.alloc(n)); SHP_TRY(sc.alloc(16));
        SHP_HIP(hipMemset(mm.p, 0, 16));
        SHP_TRY(launch_median_u8q(di.as<float>(), H, W, radius, mm.as<uint32_t>(), q.as<uint8_t>(), sc.as<float>(),
                                  dout.as<float>(), nullptr));
        uint32_t h[4];
        SHP_HIP(hipMemcpy(h, mm.p, 16, hipMemcpyDeviceToHost));
        if (h[2]) { set_error("the opencv median needs finite values"); return STEREO_ENONFINITE; }
    } else { set_error("unknown median mode %d", mode); return STEREO_EINVAL; }
    SHP_HIP(hipMemcpy(out, dout.p, n * 4, hipMemcpyDeviceToHost));
    return STEREO_OK;
}

int stereo_median_filter_u8_host(const uint8_t *image, int H, int W, int radius, float *out) {
    if (!image || !out) { set_error("null pointer"); return STEREO_EINVAL; }
    SHP_TRY(check_image_dims(H, W));
    if (radius <= 0) { set_error("radius must be a positive integer"); return STEREO_EINVAL; }
    size_t n = (size_t)H * W;
    DevBuf di, dout;
    SHP_TRY(di.alloc(n)); SHP_TRY(dout.alloc(n * 4));
    SHP_HIP(hipMemcpy(di.p, image, n, hipMemcpyHostToDevice));
    SHP_TRY(launch_median_u8(di.as<uint8_t>(), H, W, radius, 1.0f, 0.0f, 0, dout.as<float>(), nullptr));
    SHP_HIP(hipMemcpy(out, dout.p, n * 4, hipMemcpyDeviceToHost));
    return STEREO_OK;
}

static int gauss_like_host(const float *image, int H, int W, double sigma, float *out, bool bilateral) {
    SHP_TRY(check_image_dims(H, W));
    if (!(sigma > 0)) { set_error("sigma must be positive"); return STEREO_EINVAL; }
    int r = radius_from_sigma(sigma);
    if (r > 32) { set_error("filter radius %d > 32 is not supported on the GPU path", r); return STEREO_EUNSUPPORTED; }
    size_t n = (size_t)H * W;
    std::vector<float> k((size_t)(2 * r + 1) * (2 * r + 1));
    gaussian_kernel_host(r, sigma, k.data());
    DevBuf di, dout, dk;
    SHP_TRY(di.alloc(n * 4)); SHP_TRY(dout.alloc(n * 4)); SHP_TRY(dk.alloc(k.size() * 4));
    SHP_HIP(hipMemcpy(di.p, image, n * 4, hipMemcpyHostToDevice));
    SHP_HIP(hipMemcpy(dk.p, k.data(), k.size() * 4, hipMemcpyHostToDevice));
    if (bilateral) SHP_TRY(launch_bilateral(di.as<float>(), H, W, r, dk.as<float>(), sigma, dout.as<float>(), nullptr));
    else SHP_TRY(launch_gaussian(di.as<float>(), H, W, r, dk.as<float>(), dout.as<float>(), nullptr));
    SHP_HIP(hipMemcpy(out, dout.p, n * 4, hipMemcpyDeviceToHost));
    return STEREO_OK;
}
int stereo_gaussian_filter_host(const float *image, int H, int W, double sigma, float *out) {
    if (!image || !out) { set_error("null pointer"); return STEREO_EINVAL; }
    return gauss_like_host(image, H, W, sigma, out, false);
}
int stereo_bilateral_filter_host(const float *image, int H, int W, double sigma, float *out) {
    if (!image || !out) { set_error("null pointer"); return STEREO_EINVAL; }
    return gauss_like_host(image, H, W, sigma, out, true);
}

int stereo_to_gray_host(const void *image, int dtype, int H, int W, int channels, int normalize, double scale,
                        float *out_gray) {
    if (!image || !out_gray) { set_error("null pointer"); return STEREO_EINVAL; }
    if (H < 0 || W < 0) { set_error("negative image size"); return STEREO_EINVAL; }
    if (channels != 1 && channels < 3) { set_error("unsupported image shape"); return STEREO_EINVAL; }
    if (normalize && !(scale > 0)) { set_error("max_value must be positive"); return STEREO_EINVAL; }
    size_t esz = dtype == STEREO_DT_U8 ? 1 : dtype == STEREO_DT_U16 ? 2 : dtype == STEREO_DT_F32 ? 4 : 0;
    if (!esz) { set_error("to_gray takes uint8, uint16 or float32 pixels"); return STEREO_EINVAL; }
    size_t npix = (size_t)H * W;
    if (npix == 0) return STEREO_OK;
    DevBuf di, dout;
    SHP_TRY(di.alloc(npix * channels * esz)); SHP_TRY(dout.alloc(npix * 4));
    SHP_HIP(hipMemcpy(di.p, image, npix * channels * esz, hipMemcpyHostToDevice));
    SHP_TRY(launch_to_gray(di.p, dtype, npix, channels, normalize, (float)scale, dout.as<float>(), nullptr));
    SHP_HIP(hipMemcpy(out_gray, dout.p, npix * 4, hipMemcpyDeviceToHost));
    return STEREO_OK;
}

int stereo_nan_minmax_host(const float *values, size_t n, float *out_min, float *out_max) {
    if ((n && !values) || !out_min || !out_max) { set_error("null pointer"); return STEREO_EINVAL; }
    DevBuf dv, mm;
    SHP_TRY(dv.alloc(n * 4)); SHP_TRY(mm.alloc(8));
    if (n) SHP_HIP(hipMemcpy(dv.p, values, n * 4, hipMemcpyHostToDevice));
    SHP_TRY(launch_nan_minmax(dv.as<float>(), n, mm.as<uint32_t>(), nullptr));
    uint32_t h[2];
    SHP_HIP(hipMemcpy(h, mm.p, 8, hipMemcpyDeviceToHost));
    *out_min = h[0] == 0xffffffffu && h[1] == 0u ? INFINITY : ordered_to_f32(h[0]);
    *out_max = h[0] == 0xffffffffu && h[1] == 0u ? -INFINITY : ordered_to_f32(h[1]);
    return STEREO_OK;
}

int stereo_pbm_rms_host(const int32_t *disparity, const float *ground_truth, const uint8_t *mask, int H, int W,
                        double bad_threshold, double *out) {
    if (!disparity || !ground_truth || !mask || !out) { set_error("null pointer"); return STEREO_EINVAL; }
    SHP_TRY(check_image_dims(H, W));
    if (!(bad_threshold > 0)) { set_error("bad_threshold must be positive"); return STEREO_EINVAL; }
    size_t n = (size_t)H * W;
    DevBuf dd, dg, dm, acc, sq;
    SHP_TRY(dd.alloc(n * 4)); SHP_TRY(dg.alloc(n * 4)); SHP_TRY(dm.alloc(n)); SHP_TRY(acc.alloc(16)); SHP_TRY(sq.alloc(8));
    SHP_HIP(hipMemcpy(dd.p, disparity, n * 4, hipMemcpyHostToDevice));
    SHP_HIP(hipMemcpy(dg.p, ground_truth, n * 4, hipMemcpyHostToDevice));
    SHP_HIP(hipMemcpy(dm.p, mask, n, hipMemcpyHostToDevice));
    SHP_TRY(launch_metrics(dd.as<int32_t>(), dg.as<float>(), dm.as<uint8_t>(), n, (float)bad_threshold,
                           acc.as<unsigned long long>(), sq.as<double>(), nullptr));
    unsigned long long h[2];
    double hs = 0.0;
    SHP_HIP(hipMemcpy(h, acc.p, 16, hipMemcpyDeviceToHost));
    SHP_HIP(hipMemcpy(&hs, sq.p, 8, hipMemcpyDeviceToHost));
    if (h[0] == 0) { set_error("ground truth has no valid pixel"); return STEREO_EINVAL; }
    out[0] = (double)h[1] / (double)h[0] * 100.0;
    out[1] = std::sqrt(hs / (double)h[0]);
    out[2] = (double)h[0];
    return STEREO_OK;
}

}  // extern "C"
