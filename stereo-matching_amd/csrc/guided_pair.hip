// Two-columns-per-lane variant of the fast fused census-cost + guided-filter + WTA kernel.
//
// Same arithmetic as guided_fast.hip (exact 32-bit stage-1 sums with the guide split
// G = k * 2^-s + g_lo, float64 a / b and vertical sums, self-referenced float32 stage 2), but a
// wavefront covers 128 image columns: lane l owns the adjacent columns 2l ("slot 0") and
// 2l + 1 ("slot 1").  That
//  * shrinks the horizontal halo from 4r of 64 columns to 4r of 128, and
//  * shares the lane chains of the box sums between the two slots.  A window of 2r + 1
//    consecutive columns ending in slot 0 of lane l holds r + 1 slot-0 values (lanes l-r..l) and
//    r slot-1 values (lanes l-r..l-1); ending in slot 1 it holds r + 1 slot-1 values and r
//    slot-0 values (lanes l-r+1..l).  One chain per slot of r shifted adds yields both the
//    r-term and the (r+1)-term partial sums, and one more add per slot combines them:
//    2r + 2 instructions for two columns instead of 4r.
// Workgroups are 8 wavefronts (the per-lane state is twice as large), each marching down the
// tile rows for one disparity at a time.
#include "guided_pair_kernel.hpp"

namespace shp {

template <int RG, bool WORD64>
static int launch_pair_t(const FastArgs &a, hipStream_t s) {
    return a.q_out ? launch_pair_q<RG, WORD64, true>(a, s) : launch_pair_q<RG, WORD64, false>(a, s);
}

int pair_tile_height(int guided_radius, bool word64) {
    switch (guided_radius) {
        case 1: return word64 ? PairTile<1, true>::TH : PairTile<1, false>::TH;
        case 2: return word64 ? PairTile<2, true>::TH : PairTile<2, false>::TH;
        case 3: return word64 ? PairTile<3, true>::TH : PairTile<3, false>::TH;
        case 4: return word64 ? PairTile<4, true>::TH : PairTile<4, false>::TH;
        case 5: return word64 ? PairTile<5, true>::TH : PairTile<5, false>::TH;
        default: return 0;
    }
}

bool fused_pair_supported(const CensusCfg &cfg, int guided_radius) {
    // (radius 1 included: 2.50 against 2.76 ms for the one-column kernel on a 2964 x 2000 frame with 270 disparities)
    return cfg.exact && guided_radius >= 1 && guided_radius <= 5 && cfg.unit > 0.f;
}

int fused_pair_slices(int H, int W, int guided_radius, bool word64, int n_disp) {
    const int th = pair_tile_height(guided_radius, word64);
    int tiles = ceil_div(W, 128 - 4 * guided_radius) * ceil_div(H, th);
    int s = 1;
    const char *e = getenv("STEREO_FAST_SLICES");
    if (e) return atoi(e) > 0 ? atoi(e) : 1;
    while (tiles * s < 1024 && s * FP_NW * 4 <= n_disp) s *= 2;
    return s;
}

// ---- tail split -----------------------------------------------------------------------------------------
// A compute unit holds one workgroup (the tile image fills its LDS), so a launch runs in rounds of as many
// workgroups as there are units, and the last round is usually partly empty: 1742 tiles on 256 units are 6.8
// rounds that take the time of 7.  The planner below replays the launch on a model of the chip (greedy
// assignment of workgroups, in index order, to the unit that frees first; a workgroup costs one unit of
// time per round of NW passes plus a fixed share) for a few ways of cutting the last tiles into
// disparity slices, and keeps the best one if it saves at least 1.5 %.  Split tiles merge their slices with
// atomicMin on the key array, which the launcher then fills with the initial key first.
struct SplitPlan { int first, slices; };

static SplitPlan plan_tail_split(int W, int H, int th, int rg, int census_radius, int d_begin, int d_end,
                                 int band_rows_max, int n_units, bool skips_left) {
    const int twv = 128 - 4 * rg, halo = 2 * rg;
    const int nx = ceil_div(W, twv), ny = ceil_div(H, th), T = nx * ny;
    SplitPlan none = {T, 1};
    if (n_units <= 0 || T < n_units) return none;
    // fixed share of a workgroup (staging, key write-out, the ragged last round of its wavefronts) in passes, and the
    // slower code variant of border tiles: fitted to measured launches (Motorcycle-sized frame, 13 ways of splitting)
    constexpr double STAGING = 1.0, BORDER = 1.12;
    auto cost = [&](int index, int slice, int n_slices) {
        const TilePos tp = pair_tile_of_index(index, nx, ny, band_rows_max);
        const int tx0 = tp.x * twv - pair_x_shift(W, twv), ty0 = tp.y * th;
        // passes at d >= lim are skipped (census costs: every column left of the first valid match)
        const int lim = skips_left ? std::min(d_end, tx0 - halo + 127 - census_radius + 1) : d_end;
        const int span = lim - d_begin - slice;
        const int passes = span > 0 ? ceil_div(span, n_slices) : 0;
        const bool interior = (ty0 - halo >= census_radius) && (ty0 + th + halo <= H - census_radius);
        return ceil_div(passes, FP_NW) * (interior ? 1.0 : BORDER) + STAGING;
    };
    // split tiles must all lie in the last band (the progress count of the other bands stays plain)
    const int last_band_first = ((ny - 1) / band_rows_max) * band_rows_max * nx;
    const int earliest = std::max(std::max(0, T - 2 * n_units), last_band_first);
    std::vector<double> units((size_t)n_units, 0.0);
    auto push_job = [](std::vector<double> &h, double c) {   // min-heap on the time a unit frees
        std::pop_heap(h.begin(), h.end(), std::greater<double>());
        h.back() += c;
        std::push_heap(h.begin(), h.end(), std::greater<double>());
    };
    std::make_heap(units.begin(), units.end(), std::greater<double>());
    for (int i = 0; i < earliest; ++i) push_job(units, cost(i, 0, 1));
    auto finish = [&](int first, int slices) {
        std::vector<double> h = units;
        for (int i = earliest; i < first; ++i) push_job(h, cost(i, 0, 1));
        for (int i = first; i < T; ++i)
            for (int z = 0; z < slices; ++z) push_job(h, cost(i, z, slices));
        return *std::max_element(h.begin(), h.end());
    };
    const double plain = finish(T, 1);
    SplitPlan best = none;
    double best_t = plain * (1.0 - 0.015);
    // candidates: whole multiples of a quarter of the units (a split that starts on a round boundary is the best one
    // when the whole tiles before it take equally long)
    const int step = std::max(1, n_units / 4);
    for (int first = (T - 1) / step * step; first >= earliest && first > 0; first -= step)
        for (int slices : {2, 3, 4, 6, 8}) {
            if ((d_end - d_begin) / slices < FP_NW) continue;   // a slice should fill the wavefronts at least once
            const double t = finish(first, slices);
            if (t < best_t) { best_t = t; best = {first, slices}; }
        }
    return best;
}

static int device_compute_units() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = -1;
        cached[dev] = n;
    }
    return cached[dev] > 0 ? cached[dev] : 0;
}

int launch_fused_guided_pair(const void *d_lw, const void *d_rw, const CensusCfg &cfg, const float *d_guide,
                             const double *d_mean, const double *d_invden, const float *d_gmax, int H, int W,
                             int dmax, int guided_radius, int d_begin, int d_end, int n_slices, uint64_t *d_keys,
                             float *d_q_out, hipStream_t s, const BandProgress *progress, float gate_eps,
                             float gate_min, bool precise, float gate_low, const uint16_t *d_codes,
                             const uint32_t *d_code_stats, int code_d0) {
    if (H <= 0 || W <= 0 || d_begin >= d_end) return STEREO_OK;
    if (d_end > (1 << 24)) { set_error("two-column fused kernel: disparities up to 2^24"); return STEREO_EUNSUPPORTED; }
    FastArgs a;
    memset(&a, 0, sizeof(a));
    a.lw = d_lw; a.rw = d_rw; a.guide = d_guide; a.mean = d_mean; a.invden = d_invden; a.gmax = d_gmax;
    a.codes = d_codes; a.code_stats = d_code_stats; a.code_d0 = code_d0; a.code_dn = d_end - d_begin;
    const bool from_codes = d_codes != nullptr;
    if (from_codes && (!d_code_stats || guided_radius > 4 || d_q_out || cfg.word64)) {
        set_error("two-column kernel on cost codes: guided radius 1..4, no layer output");
        return STEREO_EUNSUPPORTED;
    }
    a.keys = d_keys; a.q_out = d_q_out;
    a.H = H; a.W = W; a.D = dmax; a.d_begin = d_begin; a.d_end = d_end;
    a.n_slices = n_slices < 1 ? 1 : n_slices;
    a.radius = cfg.radius; a.unit = cfg.unit;
    {
        int e;
        a.unit_pow2 = std::frexp(cfg.unit, &e) == 0.5f ? 1 : 0;
        const int rg = guided_radius, win = 2 * rg + 1;
        a.lean_scale = rg <= 3 ? (1.0f / (float)(win * win * win * win)) * cfg.unit : 1.0f / (float)(win * win);
    }
    a.large_units = (uint32_t)std::lround((double)cfg.large / (double)cfg.unit);
    a.atomic_out = a.n_slices > 1;
    a.gate_eps = gate_eps; a.gate_min = gate_min; a.gate_low = gate_low; a.precise = precise ? 1 : 0;
    if (progress && progress->band_tile_rows > 0) {
        a.progress = progress->words;
        a.band_tile_rows = progress->band_tile_rows;
        a.epoch = progress->epoch;
    }
    const bool w64 = cfg.word64 != 0;
    a.split_first = 0; a.split_slices = 1;
    const char *no_split = getenv("STEREO_NO_TAIL_SPLIT");
    if (a.n_slices == 1 && !d_q_out && !(no_split && atoi(no_split))) {
        // the plan depends on the frame, the disparity range and the bands only: remember the last one
        struct Key { int W, H, rg, w64, r, d0, d1, band_rows, units, codes; };
        static std::mutex mu;
        static Key last_key = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        static SplitPlan last_plan = {0, 1};
        const int th = pair_tile_height(guided_radius, w64);
        const int ny = th > 0 ? ceil_div(H, th) : 0;
        const Key key = {W, H, guided_radius, w64 ? 1 : 0, cfg.radius, d_begin, d_end,
                         a.band_tile_rows > 0 ? a.band_tile_rows : ny, device_compute_units(), from_codes ? 1 : 0};
        if (th > 0 && guided_radius >= 1 && guided_radius <= 5) {
            std::lock_guard<std::mutex> lock(mu);
            if (memcmp(&key, &last_key, sizeof(Key)) != 0) {
                last_plan = plan_tail_split(W, H, th, guided_radius, cfg.radius, d_begin, d_end, key.band_rows, key.units,
                                            !from_codes);
                last_key = key;
            }
            if (last_plan.slices > 1) { a.split_first = last_plan.first; a.split_slices = last_plan.slices; }
        }
    }
    if (a.split_slices > 1) {
        // only the split tiles merge their slices with atomicMin and need the start value; every other tile stores
        const int th = pair_tile_height(guided_radius, w64), twv = 128 - 4 * guided_radius;
        const int nx = ceil_div(W, twv), ny = ceil_div(H, th);
        hipLaunchKernelGGL(pair_fill_split_keys_kernel, dim3(nx * ny - a.split_first), dim3(256), 0, s, d_keys, H, W, th,
                           twv, nx, ny, a.band_tile_rows > 0 ? a.band_tile_rows : ny, a.split_first);
        SHP_LAUNCH_CHECK();
    }
    if (from_codes) return launch_pair_codes(a, guided_radius, s);
    switch (guided_radius) {
        case 1: return w64 ? launch_pair_t<1, true>(a, s) : launch_pair_t<1, false>(a, s);
        case 2: return w64 ? launch_pair_t<2, true>(a, s) : launch_pair_t<2, false>(a, s);
        case 3: return w64 ? launch_pair_t<3, true>(a, s) : launch_pair_t<3, false>(a, s);
        case 4: return w64 ? launch_pair_t<4, true>(a, s) : launch_pair_t<4, false>(a, s);
        case 5: return w64 ? launch_pair_t<5, true>(a, s) : launch_pair_t<5, false>(a, s);
        default: set_error("fast fused guided kernel supports radius 1..5"); return STEREO_EUNSUPPORTED;
    }
}

}  // namespace shp
