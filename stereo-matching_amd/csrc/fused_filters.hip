// Volume-free compute_disparity for the median (uint8-quantised, "opencv"), Gaussian and
// bilateral aggregation filters: census cost -> filter -> running WTA in one kernel
// (census.py:187-202, filters.py:139-224,279-310, stereo.py:145-170).
//
// A 256-thread workgroup owns a 16 x 64 pixel tile (20 x 64 for the uint8 median) for the whole disparity range of the launch.
// Every thread keeps the census word, the border-mapped coordinates and the validity of "its"
// pixels of the tile + halo in registers; per disparity it writes their costs into one of two LDS
// cost tiles (float32, or the quantised byte for the median), the workgroup synchronises once, and
// every thread filters four (five) vertically adjacent output pixels from that tile and updates their
// packed (cost, disparity) keys in registers.  Nothing but the final keys is written to HBM.
//
// The median needs the minimum and maximum of each unfiltered cost layer (filters.py:141-151)
// before any pixel of the layer can be quantised: cost_minmax_kernel computes them for every
// disparity of the launch from the census words in a first pass.
#include <type_traits>
#include <vector>

#include "census_dev.hpp"

// The reference rescales a median with separate float32 multiply and add (filters.py:177-180), the
// quantisation is a subtract followed by a multiply, and the bilateral filter divides two float32
// sums: a fused multiply-add anywhere in these chains changes the last bit of some outputs.  Fused
// operations in this file are written out (fmaf) where they are wanted.
#pragma clang fp contract(off)

namespace shp {

constexpr int FF_TW = 64, FF_THREADS = 256;
// Outputs per thread (vertically adjacent) and rows per tile.  Five rows per thread amortise the tile's halo and the
// shared window rows better than four where the registers allow it (the uint8 median: 26.5 ms on a 7680 x 4320 frame
// with 256 disparities against 27.3 with four and 28.7 with six); the float filters keep four.
#ifndef FF_MEDIAN_RPT
#define FF_MEDIAN_RPT 5
#endif
#ifndef FF_GAUSS_RPT
#define FF_GAUSS_RPT 6
#endif
#ifndef FF_MEDX_RPT
#define FF_MEDX_RPT 4
#endif
template <int FILTER, int RAD> struct FFTile {
    // (uint8 median above radius 8: four rows, so that the cost tiles and the column counts stay within 64 KB of LDS)
    static constexpr int RPT = (FILTER == STEREO_FILTER_MEDIAN_U8Q) ? (RAD > 8 ? 4 : FF_MEDIAN_RPT) : (FILTER == STEREO_FILTER_GAUSSIAN) ? FF_GAUSS_RPT : (FILTER == STEREO_FILTER_MEDIAN) ? FF_MEDX_RPT : 4;
    static constexpr int TH = 4 * RPT;
};
constexpr int FF_MAX_RADIUS = 8;          // bilateral (2-D weight tables in the kernel arguments), exact median
constexpr int FF_MAX_MEDIAN_RADIUS = 12;  // uint8 median (bit-sliced counting; no weight tables)
constexpr int FF_MAX_GAUSS_RADIUS = 12;   // Gaussian (separable: a 1-D weight table)
constexpr int FF_MAX_BIL_RADIUS = 12;     // bilateral above FF_MAX_RADIUS: the 2-D table of log2 weights is formed in LDS from a 1-D one
#ifndef FF_BIL_UNROLL_MAX
#define FF_BIL_UNROLL_MAX 2
#endif

struct FFArgs {
    const void *lw, *rw;
    CensusCfg cfg;
    int H, W, d_begin, d_end;
    float kern[(2 * FF_MAX_RADIUS + 1) * (2 * FF_MAX_RADIUS + 1)];  // normalised spatial weights, row-major
    float kern1[2 * FF_MAX_GAUSS_RADIUS + 1];  // Gaussian: normalised 1-D weights
    // bilateral: log2 of the spatial weights, rows padded to an even length so that a pair of horizontally adjacent
    // weights is one aligned 8-byte scalar load
    alignas(8) float kernp[2 * FF_MAX_RADIUS + 1][2 * FF_MAX_RADIUS + 2];
    float vscale, inv_vscale;   // bilateral: sqrt(log2(e) / (2 sigma^2)) and its reciprocal
    // bilateral, radius above FF_MAX_RADIUS: log2 of the spatial weight of (dy, dx) = kern_lg1[dy] + kern_lg1[dx]
    float kern_lg1[2 * FF_MAX_BIL_RADIUS + 1];
    uint32_t *mm;         // median: [d - d_begin][2] order-preserving (min, max) of the cost layer
    uint64_t *keys;
    int plain_median;     // test hook: take the plain bisection instead of the bit-sliced count
    uint32_t code_large;  // exact median: code of an invalid pixel = largest valid cost in units + 1
};

__device__ __forceinline__ int ff_reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * (n - 1) - i;
    }
    return i;
}

// cost of pixel (row offset ro, column gx) at disparity d; lvalid: left census centre valid
template <typename WordT>
__device__ __forceinline__ float ff_cost(const WordT *__restrict__ rw, const CensusCfg &cfg, WordT lword, bool lvalid,
                                         int ro, int gx, int d) {
    const int xr = gx - d;
    const bool ok = lvalid && xr >= cfg.radius;
    const WordT rword = rw[ro + (ok ? xr : gx)];
    const float c = pair_cost<WordT>(lword, rword, cfg);
    return ok ? c : cfg.large;
}

// Minimum / maximum over the 64 lanes of a wavefront (every lane active), the same value in every lane: four
// data-parallel-primitive steps inside each row of 16 lanes, then the four rows through scalar registers.  (The
// butterfly of lane permutations this replaces goes through the LDS crossbar, about a hundred cycles a step, six
// dependent steps per reduction: with two wavefronts per SIMD that latency was most of the range pass.)
template <int CTRL>
__device__ __forceinline__ uint32_t ff_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, ff_dpp<0xB1>(v));    // quad_perm [1, 0, 3, 2]
    v = min(v, ff_dpp<0x4E>(v));    // quad_perm [2, 3, 0, 1]
    v = min(v, ff_dpp<0x141>(v));   // row_half_mirror
    v = min(v, ff_dpp<0x140>(v));   // row_mirror
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(a, b), min(c, d));
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { return ~wave_min_u32(~v); }

// ---- pass 1 (median only): min / max of every cost layer ---------------------------------------
// One wavefront owns MM_ROWS rows x 512 columns: 8 x MM_ROWS pixels per lane whose left words stay in
// registers over the disparity loop, so that the wavefront-wide reduction and the two global atomics
// are paid once per 2048 pixels and disparity.  This version serves weights that are not exactly summable
// (costs are float32 sums formed in the reference's order); cost_minmax_exact_kernel below serves the others.
constexpr int MM_ROWS = 4;
constexpr int MM_CHUNK = 256;  // disparities whose per-wavefront results are parked in LDS between merges
template <typename WordT>
__global__ __launch_bounds__(256) void cost_minmax_kernel(const WordT *__restrict__ lw, const WordT *__restrict__ rw,
                                                         CensusCfg cfg, int H, int W, int d_begin, int d_end,
                                                         uint32_t *__restrict__ mm) {
    constexpr int PPT = 8;  // pixels per lane and row: columns x0 + lane + 64 j
    const int lane = threadIdx.x & 63;
    const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * MM_ROWS;
    const int x0 = blockIdx.x * (64 * PPT);
    if (y0 >= H) return;
    const int R = cfg.radius;
    WordT lword[MM_ROWS][PPT];
    uint32_t valid_bits = 0u, inside_bits = 0u;  // bit r * PPT + j
#pragma unroll
    for (int r = 0; r < MM_ROWS; ++r) {
        const int y = min(y0 + r, H - 1);
        const bool yin = y0 + r < H, yvalid = y >= R && y < H - R;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int x = x0 + lane + 64 * j;
            const bool in = yin && x < W;
            const int xc = min(x, W - 1);
            lword[r][j] = lw[y * W + xc];
            if (in) inside_bits |= 1u << (r * PPT + j);
            if (in && yvalid && xc >= R && xc < W - R) valid_bits |= 1u << (r * PPT + j);
        }
    }
    // (min, max) of this wavefront's pixels for one disparity, reduced over the lanes
    auto layer_range = [&](int d, uint32_t &lo_out, uint32_t &hi_out) {
        uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
        for (int r = 0; r < MM_ROWS; ++r) {
            const int ro = min(y0 + r, H - 1) * W;
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                const int x = min(x0 + lane + 64 * j, W - 1);
                const bool in = (inside_bits >> (r * PPT + j)) & 1u;
                const bool ok = ((valid_bits >> (r * PPT + j)) & 1u) && x - d >= R;
                const uint32_t k = f32_to_ordered(ff_cost<WordT>(rw, cfg, lword[r][j], ok || ((valid_bits >> (r * PPT + j)) & 1u), ro, x, d));
                if (in) { lo = min(lo, k); hi = max(hi, k); }
            }
        }
        lo_out = wave_min_u32(lo); hi_out = wave_max_u32(hi);
    };
    // The per-disparity results of a wavefront are parked in LDS and merged into the global table once per
    // MM_CHUNK disparities by all lanes at once: the disparity loop then holds no global read-modify-write whose
    // latency every iteration would have to wait for, and two disparities are evaluated per iteration so that
    // twice as many loads are in flight.
    __shared__ uint32_t parked[4][2][MM_CHUNK];
    uint32_t (*mine)[MM_CHUNK] = parked[threadIdx.x >> 6];
    for (int dc = d_begin; dc < d_end; dc += MM_CHUNK) {
        const int n = min(MM_CHUNK, d_end - dc);
        for (int i = 0; i < n; i += 2) {
            uint32_t lo0, hi0, lo1 = 0xffffffffu, hi1 = 0u;
            layer_range(dc + i, lo0, hi0);
            if (i + 1 < n) layer_range(dc + i + 1, lo1, hi1);
            if (lane == 0) {
                mine[0][i] = lo0; mine[1][i] = hi0;
                if (i + 1 < n) { mine[0][i + 1] = lo1; mine[1][i + 1] = hi1; }
            }
        }
        // (the wavefront's own LDS writes are visible to its lanes after the wait the compiler places here)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int i = lane; i < n; i += 64) {
            uint32_t *m = mm + 2 * (size_t)(dc + i - d_begin);
            const uint32_t lo = mine[0][i], hi = mine[1][i];
            if (lo < m[0]) atomicMin(&m[0], lo);
            if (hi > m[1]) atomicMax(&m[1], hi);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
}

// The same pass for exactly summable weights (the cost is an integer number of units), written without
// branches or per-pixel selects: validity travels as all-ones / all-zero masks, the column test x - d >= radius is
// formed once per column and disparity, and `large` enters through one flag per lane (a pixel inside the image
// costs `large` iff it is invalid for every disparity -- known before the loop -- or its column fails the column test).
// The right words come from LDS: a workgroup (8 rows x 512 columns) stages, for MM_CHUNK disparities at a time, the
// words of its rows from MM_CHUNK - 1 columns left of the tile to its last column, and every disparity of the chunk
// reads them from there.  Read from global memory instead, the words were fetched from HBM again every few
// disparities (21.7 GB per launch on a 7680 x 4320 frame with 256 disparities: the wavefronts of a compute unit
// touch 128 KB per disparity, four times its vector cache, and those of an XCD all of its L2).
// HI_IS_LARGE: `large` is positive.  The first pixel of the frame is never a valid census centre, so every layer
// holds `large`, which then exceeds every valid cost: the maximum needs no arithmetic.
constexpr int MMX_ROWS = 2;                   // rows per wavefront
constexpr int MMX_COLS = 512;                 // columns per workgroup
constexpr int MMX_LW = MMX_COLS + MM_CHUNK;   // right words per staged row
template <typename WordT, bool HI_IS_LARGE>
__global__ __launch_bounds__(256) void cost_minmax_exact_kernel(const WordT *__restrict__ lw, const WordT *__restrict__ rw,
                                                               CensusCfg cfg, int H, int W, int d_begin, int d_end,
                                                               uint32_t *__restrict__ mm) {
    constexpr int PPT = MMX_COLS / 64;  // pixels per lane and row: columns x0 + lane + 64 j
    __shared__ WordT rtile[4 * MMX_ROWS][MMX_LW];
    __shared__ uint32_t parked[4][2][MM_CHUNK];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int yb = (int)blockIdx.y * 4 * MMX_ROWS;   // first row of the workgroup (inside the image)
    const int y0 = yb + wave * MMX_ROWS;              // first row of the wavefront (may lie below the image)
    const int x0 = blockIdx.x * MMX_COLS;
    const int x_last = min(x0 + MMX_COLS - 1, W - 1);
    const int R = cfg.radius;
    WordT lword[MMX_ROWS][PPT];
    uint32_t vmask[MMX_ROWS][PPT];     // all ones: left census centre valid and the pixel inside the image
    uint32_t col_inside[PPT];          // all ones: some row of the column lies inside the image
    int xcol[PPT];
    uint32_t const_large = 0u;         // all ones: some pixel of this lane is inside the image and never valid
#pragma unroll
    for (int j = 0; j < PPT; ++j) { col_inside[j] = 0u; xcol[j] = min(x0 + lane + 64 * j, W - 1); }
#pragma unroll
    for (int r = 0; r < MMX_ROWS; ++r) {
        const int y = min(y0 + r, H - 1);
        const bool yin = y0 + r < H, yvalid = y >= R && y < H - R;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int x = x0 + lane + 64 * j;
            const bool in = yin && x < W;
            const bool valid = in && yvalid && xcol[j] >= R && xcol[j] < W - R;
            lword[r][j] = lw[(size_t)y * W + xcol[j]];
            vmask[r][j] = valid ? 0xffffffffu : 0u;
            if (in) col_inside[j] = 0xffffffffu;
            if (in && !valid) const_large = 0xffffffffu;
        }
    }
    // the masks live in vector registers from here on (left to itself the compiler turns them back into lane
    // masks in scalar registers, runs out of those, and spills them through v_writelane / v_readlane)
#pragma unroll
    for (int r = 0; r < MMX_ROWS; ++r)
#pragma unroll
        for (int j = 0; j < PPT; ++j) asm volatile("" : "+v"(vmask[r][j]));
#pragma unroll
    for (int j = 0; j < PPT; ++j) asm volatile("" : "+v"(col_inside[j]));
    asm volatile("" : "+v"(const_large));
    const uint32_t large_key = f32_to_ordered(cfg.large);
    uint32_t boff[PPT];   // byte offset of the column's own word within a staged row
    const char *myrows = reinterpret_cast<const char *>(&rtile[wave * MMX_ROWS][0]);
    // One layer in two steps, so that the staged words of the next layer can be requested before the arithmetic of
    // this one (with two wavefronts per SIMD the LDS latency is not hidden otherwise): read_layer fetches the right
    // words of disparity dc + dl (chunk at dc), layer_range reduces them.
    auto read_layer = [&](int dl, WordT (&rword)[MMX_ROWS][PPT]) {
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            // (the staged row starts at column 0 or at the leftmost column the chunk reads: a column that fails the
            //  column test reads some staged word)
            const uint32_t o = __builtin_elementwise_sub_sat(boff[j], (uint32_t)dl * (uint32_t)sizeof(WordT));
#pragma unroll
            for (int r = 0; r < MMX_ROWS; ++r)
                rword[r][j] = *reinterpret_cast<const WordT *>(myrows + r * (MMX_LW * (int)sizeof(WordT)) + o);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto layer_range = [&](int d, const WordT (&rword)[MMX_ROWS][PPT], uint32_t &lo_out, uint32_t &hi_out) {
        uint32_t lo = 0xffffffffu, hi = 0u, large_seen = const_large;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const uint32_t okx = (uint32_t)((R - 1 - (xcol[j] - d)) >> 31);   // all ones iff x - d >= R
            large_seen |= col_inside[j] & ~okx;
#pragma unroll
            for (int r = 0; r < MMX_ROWS; ++r) {
                const uint32_t u = field_sum((WordT)(lword[r][j] ^ rword[r][j]));
                const uint32_t m = vmask[r][j] & okx;
                lo = min(lo, u | ~m);
                if (!HI_IS_LARGE) hi = max(hi, u & m);
            }
        }
        // units -> order-preserving float keys; a lane without a valid pixel keeps the neutral elements
        const bool none = lo == 0xffffffffu;
        lo = none ? lo : f32_to_ordered((float)lo * cfg.unit);
        hi = (HI_IS_LARGE || (none && hi == 0u)) ? 0u : f32_to_ordered((float)hi * cfg.unit);
        if (large_seen) { lo = min(lo, large_key); hi = max(hi, large_key); }
        lo_out = wave_min_u32(lo); hi_out = wave_max_u32(hi);
        __builtin_amdgcn_sched_barrier(0);
    };
    uint32_t (*mine)[MM_CHUNK] = parked[wave];
    for (int dc = d_begin; dc < d_end; dc += MM_CHUNK) {
        const int n = min(MM_CHUNK, d_end - dc);
        // stage the columns the chunk reads, x - d for the tile's x and the chunk's d clamped at the first column:
        // xb .. xe of the workgroup's rows (rows below the image repeat the last one)
        const int xb = max(x0 - dc - (n - 1), 0);
        const int xe = max(x_last - dc, xb);
        const int width = xe - xb + 1;       // <= MMX_COLS + n - 1 < MMX_LW
        __syncthreads();                     // (the previous chunk's readers are done)
        for (int i = threadIdx.x; i < 4 * MMX_ROWS * MMX_LW; i += 256) {
            const int row = i / MMX_LW, c = i - row * MMX_LW;
            if (c < width) rtile[row][c] = rw[(size_t)min(yb + row, H - 1) * W + xb + c];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PPT; ++j) boff[j] = (uint32_t)max(xcol[j] - dc - xb, 0) * (uint32_t)sizeof(WordT);
        WordT rw_a[MMX_ROWS][PPT], rw_b[MMX_ROWS][PPT];
        read_layer(0, rw_a);
        for (int i = 0; i < n; i += 2) {
            uint32_t lo0, hi0, lo1 = 0xffffffffu, hi1 = 0u;
            read_layer(min(i + 1, n - 1), rw_b);
            layer_range(dc + i, rw_a, lo0, hi0);
            read_layer(min(i + 2, n - 1), rw_a);
            if (i + 1 < n) layer_range(dc + i + 1, rw_b, lo1, hi1);
            if (lane == 0) {
                mine[0][i] = lo0; mine[1][i] = hi0;
                if (i + 1 < n) { mine[0][i + 1] = lo1; mine[1][i + 1] = hi1; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int i = lane; i < n; i += 64) {
            uint32_t *m = mm + 2 * (size_t)(dc + i - d_begin);
            const uint32_t lo = mine[0][i], hi = mine[1][i];
            if (lo < m[0]) atomicMin(&m[0], lo);
            if (hi > m[1]) atomicMax(&m[1], hi);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
}

__global__ void minmax_fill_kernel(uint32_t *mm, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { mm[2 * i] = 0xffffffffu; mm[2 * i + 1] = 0u; }
}

// Medians of a thread's vertically adjacent K x K windows of bytes (stored one per float slot, as integer
// bits) by bisection on the value: the median is the smallest m with count(window <= m) > K*K/2.
// NPW = 32 / FB window elements share a 32-bit word in FB-bit fields holding (byte - tlo); with a
// guard bit on top of each field, (m | guards) - word leaves the guard bit set exactly where the
// element is <= m, so one subtract, one and, one popcount compare NPW elements with m.
template <int K, int FB, int LW, int FF_ROWS_PER_THREAD>
__device__ __forceinline__ void median_bisect(const float *win, int tlo, int thi, int *med) {
    constexpr int NR = FF_ROWS_PER_THREAD + K - 1;
    constexpr int NPW = 32 / FB;
    constexpr int FULL = K / NPW;                // full words per window row
    constexpr int REM = K - NPW * FULL;          // elements of a row left over
    constexpr int NREST = (REM * K + NPW - 1) / NPW;
    constexpr int PAD = NREST * NPW - REM * K;   // empty fields (zero: always counted)
    uint32_t guard = 0u, ones = 0u;
#pragma unroll
    for (int f = 0; f < NPW; ++f) { guard |= 1u << (FB * f + FB - 1); ones |= 1u << (FB * f); }
    uint32_t words[NR][FULL > 0 ? FULL : 1];
    uint32_t rest[NR][REM > 0 ? REM : 1];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
#pragma unroll
        for (int w = 0; w < FULL; ++w) {
            uint32_t acc = 0u;
#pragma unroll
            for (int f = 0; f < NPW; ++f) acc |= (__float_as_uint(win[rr * LW + NPW * w + f]) - (uint32_t)tlo) << (FB * f);
            words[rr][w] = acc;
        }
#pragma unroll
        for (int f = 0; f < REM; ++f) rest[rr][f] = __float_as_uint(win[rr * LW + NPW * FULL + f]) - (uint32_t)tlo;
    }
    int steps = 0;
    while ((1 << steps) <= thi - tlo) ++steps;
#pragma unroll
    for (int j = 0; j < FF_ROWS_PER_THREAD; ++j) {
        uint32_t rw[NREST > 0 ? NREST : 1];
#pragma unroll
        for (int w = 0; w < NREST; ++w) {
            uint32_t acc = 0u;
#pragma unroll
            for (int f = 0; f < NPW; ++f) {
                const int e = NPW * w + f;
                if (e < REM * K) acc |= rest[j + e / (REM > 0 ? REM : 1)][e % (REM > 0 ? REM : 1)] << (FB * f);
            }
            rw[w] = acc;
        }
        constexpr int need = (K * K) / 2 + 1 + PAD;
        int lo = 0, hi = thi - tlo;
        for (int step = 0; step < steps; ++step) {
            const int mid = (lo + hi) >> 1;
            const uint32_t mg = (uint32_t)mid * ones + guard;
            int cnt = 0;
#pragma unroll
            for (int dy = 0; dy < K; ++dy)
#pragma unroll
                for (int w = 0; w < FULL; ++w) cnt += __popc((mg - words[j + dy][w]) & guard);
#pragma unroll
            for (int w = 0; w < NREST; ++w) cnt += __popc((mg - rw[w]) & guard);
            if (cnt >= need) hi = mid; else lo = mid + 1;
        }
        med[j] = lo + tlo;
    }
}

// ---- bit-sliced counting median (K x K windows whose bytes span fewer than 32 levels) --------------
// A byte b of the tile becomes the 32-bit mask of the levels t >= v, v = b - tlo (255, the quantised
// `large`, takes level 31, which no other byte reaches).  Bit t of a mask says "this element is <= level
// t", so adding masks bit position by bit position counts, for all 32 levels at once, the elements not
// above each level.  The additions are carry-save adders on whole words (two three-input logic
// instructions each): K masks of a column -> a count per level held in bs_bits(K) words (7 -> 3); K such
// columns -> the count of the window (49 -> 6 bits); the word of the predicate "count >= K * K / 2 + 1" has its
// lowest set bit at the median level.  Column counts are shared through LDS by the K windows that contain the column.
// (written as the three-input logic instruction with its truth table: from the C expressions the compiler forms
//  it for some of the adders only and spends two or three instructions on the others)
__device__ __forceinline__ uint32_t bs_xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint32_t bs_maj3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }
__device__ __forceinline__ uint32_t bs_or3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xFE); }
__device__ __forceinline__ uint32_t bs_and3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x80); }
// number of bits that hold n
constexpr int bs_bits(int n) { int b = 0; while (n) { ++b; n >>= 1; } return b; }
// N one-bit words of equal weight -> the bit planes of their sum, out[0 .. bs_bits(N)): a chain of full adders leaves
// one word of this weight and N / 2 carries of the next, which are summed the same way.
template <int N>
__device__ __forceinline__ void bs_sum(const uint32_t *w, uint32_t *out) {
    if constexpr (N == 1) {
        out[0] = w[0];
    } else {
        constexpr int F = (N - 1) / 2, C = N / 2;
        uint32_t carry[C];
        uint32_t acc = w[0];
#pragma unroll
        for (int i = 0; i < F; ++i) {
            carry[i] = bs_maj3(acc, w[2 * i + 1], w[2 * i + 2]);
            acc = bs_xor3(acc, w[2 * i + 1], w[2 * i + 2]);
        }
        if (N % 2 == 0) { carry[F] = acc & w[N - 1]; acc ^= w[N - 1]; }
        out[0] = acc;
        bs_sum<C>(carry, out + 1);
    }
}
// Bit planes s[W .. NB2) of S = sum over p of 2^p U_p, where U_p has the bit planes u[p][0 .. NB1): weight by weight,
// the planes of that weight and the NC carries of the weight below go through a chain of full adders.  (S fits NB2
// bits, so the carries out of the last weight are zero and dropped.)
template <int NB1, int NB2, int W, int NC>
__device__ __forceinline__ void bs_weighted_sum(const uint32_t (&u)[NB1][NB1], const uint32_t *cin, uint32_t *s) {
    if constexpr (W < NB2) {
        constexpr int P0 = W - (NB1 - 1) > 0 ? W - (NB1 - 1) : 0, P1 = W < NB1 - 1 ? W : NB1 - 1;
        constexpr int NIN = P1 >= P0 ? P1 - P0 + 1 : 0, N = NIN + NC;
        if constexpr (N == 0) {
            s[W] = 0u;
            bs_weighted_sum<NB1, NB2, W + 1, 0>(u, cin, s);
        } else {
            uint32_t w[N];
#pragma unroll
            for (int i = 0; i < NIN; ++i) w[i] = u[P0 + i][W - P0 - i];
#pragma unroll
            for (int i = 0; i < NC; ++i) w[NIN + i] = cin[i];
            constexpr int F = (N - 1) / 2, C = N / 2;
            uint32_t carry[C > 0 ? C : 1];
            uint32_t acc = w[0];
#pragma unroll
            for (int i = 0; i < F; ++i) {
                carry[i] = bs_maj3(acc, w[2 * i + 1], w[2 * i + 2]);
                acc = bs_xor3(acc, w[2 * i + 1], w[2 * i + 2]);
            }
            if (N > 1 && N % 2 == 0) { carry[F] = acc & w[N - 1]; acc ^= w[N - 1]; }
            s[W] = acc;
            bs_weighted_sum<NB1, NB2, W + 1, C>(u, carry, s);
        }
    }
}
// the word of the predicate S >= T for the bit planes s[0 .. NB) of S (T < 2^NB)
template <int NB, int T>
__device__ __forceinline__ uint32_t bs_at_least(const uint32_t *s) {
    uint32_t ge = 0u, eq = 0xffffffffu;
#pragma unroll
    for (int b = NB - 1; b >= 0; --b) {
        if ((T >> b) & 1) eq &= s[b];
        else { ge |= eq & s[b]; eq &= ~s[b]; }
    }
    return ge | eq;
}

// ---- pass 2: cost -> filter -> WTA ------------------------------------------------------------------
// EXACT: the census weights are exactly summable (the cost is the integer field sum times the unit).  A template
// parameter rather than a test of cfg.exact per pixel: with the branch inside the cost stage every load of a right
// word was waited for before the next one was issued.
template <int FILTER, bool WORD64, int RAD, bool EXACT>
__global__ __launch_bounds__(FF_THREADS) void fused_filter_kernel(FFArgs a) {
    using WordT = typename std::conditional<WORD64, uint64_t, uint32_t>::type;
    constexpr int FF_ROWS_PER_THREAD = FFTile<FILTER, RAD>::RPT, FF_TH = FFTile<FILTER, RAD>::TH;
    constexpr int K = 2 * RAD + 1;
    constexpr int LW = FF_TW + 2 * RAD, LH = FF_TH + 2 * RAD, NP = LW * LH;
    constexpr int PT = (NP + FF_THREADS - 1) / FF_THREADS;   // halo pixels per thread
    constexpr int NR = FF_ROWS_PER_THREAD + K - 1;           // tile rows one thread's windows span
    constexpr bool MEDIAN = FILTER == STEREO_FILTER_MEDIAN_U8Q;
    // Exact median (scipy / vectorized / naive methods) of exactly summable census costs: the tile holds the cost of a
    // pixel in integer units, or `code_large` (one above every valid cost) where the reference writes `large`.  The
    // map from codes to float32 costs is strictly increasing, so the median code is the code of the median cost.
    constexpr bool MEDX = FILTER == STEREO_FILTER_MEDIAN;
    __shared__ float cbuf[2][NP];
    // median: min / max byte of the cost tiles of disparities d, d + 1, d + 2 (slot = index mod 3); a slot is
    // reset two iterations before it is used, after the barrier that ends its last readers
    __shared__ uint32_t trange_buf[3][3];   // [slot][min byte, max byte other than 255, some byte is 255]
    // bit-sliced median: 3-bit column counts, [plane][tile row][halo column]
    constexpr bool BITSLICED = MEDIAN;
    constexpr int NB1 = bs_bits(K), NB2 = bs_bits(K * K);   // bits of a column count, of a window count
    constexpr int BS_CW = LW + 2;                            // row pitch of the column-count planes
    __shared__ uint32_t colcnt[BITSLICED ? NB1 * FF_TH * BS_CW : 1];
    static_assert(sizeof(float) * 2 * NP + sizeof(uint32_t) * (BITSLICED ? NB1 * FF_TH * BS_CW : 1) + 64 <= 64 * 1024,
                  "the cost tiles and the column counts must fit the 64 KB of static LDS");

    const WordT *lw = static_cast<const WordT *>(a.lw);
    const WordT *rw = static_cast<const WordT *>(a.rw);
    const int H = a.H, W = a.W, R = a.cfg.radius;
    const int x0 = blockIdx.x * FF_TW, y0 = blockIdx.y * FF_TH;
    const int tid = threadIdx.x;
    // bilateral windows too large for a table in the kernel arguments: log2 of the spatial weights, rows padded to an
    // even length, summed from the 1-D table (the first barrier of the disparity loop orders these writes)
    constexpr bool BIL_LDS_TABLE = FILTER == STEREO_FILTER_BILATERAL && RAD > FF_MAX_RADIUS;
    __shared__ __attribute__((aligned(8))) float lgtab[BIL_LDS_TABLE ? K * (K + 1) : 2];
    if (BIL_LDS_TABLE)
        for (int i = tid; i < K * (K + 1); i += FF_THREADS) {
            const int dy = i / (K + 1), dx = i - dy * (K + 1);
            lgtab[i] = dx < K ? a.kern_lg1[dy] + a.kern_lg1[dx] : 0.f;
        }

    // this thread's pixels of the tile + halo, with the filter's border rule applied
    WordT lword[PT];
    // byte offset of the pixel's own right word, and the largest disparity at which the pair is valid (-1: the
    // left pixel is no valid census centre): the pair (x, x - d) is valid iff d <= x - R.
    uint32_t boff[PT];
    int dlim[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int idx = min(tid + j * FF_THREADS, NP - 1);
        const int ly = idx / LW, lx = idx - ly * LW;
        int gy = y0 - RAD + ly, gx = x0 - RAD + lx;
        if (MEDIAN || MEDX) { gy = min(max(gy, 0), H - 1); gx = min(max(gx, 0), W - 1); }   // replicate (cv2.medianBlur, scipy mode="nearest")
        else { gy = ff_reflect101(gy, H); gx = ff_reflect101(gx, W); }               // reflect-101
        lword[j] = lw[gy * W + gx];
        boff[j] = (uint32_t)(gy * W + gx) * (uint32_t)sizeof(WordT);
        dlim[j] = (gy >= R && gy < H - R && gx >= R && gx < W - R) ? gx - R : -1;
    }

    const int lx = tid & 63, q = tid >> 6;
    uint64_t best[FF_ROWS_PER_THREAD];
#pragma unroll
    for (int j = 0; j < FF_ROWS_PER_THREAD; ++j) best[j] = SHP_KEY_INIT;

    if (MEDIAN || MEDX) {
        if (tid < 3) { trange_buf[tid][0] = MEDX ? 0xffffffffu : 255u; trange_buf[tid][1] = 0u; trange_buf[tid][2] = 0u; }
        __syncthreads();
    }
    // median: the quantisation constants of 64 layers at a time, one layer per lane (two float64 divisions
    // each), handed out per layer by a lane read
    float lane_qlo = 0.f, lane_qscale = 0.f, lane_qinv = 0.f;
    uint32_t lane_flags = 0u;   // bit 0: flat layer, bit 1: the layer's maximum is the census `large` value
    for (int d = a.d_begin; d < a.d_end; ++d) {
        float *cb = cbuf[(d - a.d_begin) & 1];
        // quantisation of this layer (filters.py:145-151,177-180)
        float qlo = 0.f, qscale = 0.f, qinv = 0.f;
        bool flat = false, census_levels = false;
        if (MEDIAN) {
            const int dl = (d - a.d_begin) & 63;
            if (dl == 0) {
                const int dd = min(d + (tid & 63), a.d_end - 1);
                const uint32_t *m = a.mm + 2 * (size_t)(dd - a.d_begin);
                lane_qlo = ordered_to_f32(m[0]);
                const float fhi = ordered_to_f32(m[1]);
                const double dlo = (double)lane_qlo, dhi = (double)fhi;
                const bool fl = dhi == dlo;
                lane_qscale = fl ? 0.f : (float)(255.0 / (dhi - dlo));
                lane_qinv = fl ? 0.f : (float)((dhi - dlo) / 255.0);
                lane_flags = (fl ? 1u : 0u) | ((fhi == a.cfg.large && lane_qlo >= 0.f && a.cfg.large > 0.f) ? 2u : 0u);
            }
            qlo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lane_qlo), dl));
            qscale = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lane_qscale), dl));
            qinv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lane_qinv), dl));
            const uint32_t fl = (uint32_t)__builtin_amdgcn_readlane((int)lane_flags, dl);
            flat = (fl & 1u) != 0u;
            // A census cost that is not `large` is at most a tenth of it (census.py:240), so in a layer whose
            // maximum is `large` and whose minimum is not negative every byte is 0 .. 26 or 255: the bit-sliced
            // median needs no range of the tile's bytes, and the bookkeeping for it is skipped.
            census_levels = BITSLICED && (fl & 2u) != 0u && !flat && !a.plain_median;
        }
        uint32_t *trange = trange_buf[(d - a.d_begin) % 3];
        uint32_t mylo = MEDX ? 2047u : 255u, myhi = 0u, my255 = 0u;
        // The right words of all of this thread's pixels are requested before any of them is used, and the loop
        // that uses them holds no branch that depends on the layer: the loads of one iteration are in flight together.
        WordT rword[PT];
        bool okv[PT];
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            okv[j] = d <= dlim[j];
            // (an invalid pair reads some word of the image: the subtraction saturates at the first one)
            const uint32_t o = __builtin_elementwise_sub_sat(boff[j], (uint32_t)d * (uint32_t)sizeof(WordT));
            rword[j] = *reinterpret_cast<const WordT *>(reinterpret_cast<const char *>(rw) + o);
        }
        // the last slot holds the few pixels beyond a multiple of the workgroup size: wavefronts whose lanes all
        // lie beyond the tile skip its arithmetic
        const bool last_slot_empty = (tid & ~63) + (PT - 1) * FF_THREADS >= NP;
        uint32_t bytes[PT];
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int idx = tid + j * FF_THREADS;
            bytes[j] = 0u;
            if (j == PT - 1 && last_slot_empty) continue;
            const WordT x = lword[j] ^ rword[j];
            if (MEDX) {
                const uint32_t e = okv[j] ? field_sum(x) : a.code_large;
                mylo = min(mylo, e);   // (a thread's surplus slot repeats a tile pixel)
                myhi = max(myhi, e);
                if (idx < NP) cb[idx] = __uint_as_float(e);
                continue;
            }
            float c = EXACT ? (float)field_sum(x) * a.cfg.unit : sequential_cost<WordT>(x, a.cfg);
            c = okv[j] ? c : a.cfg.large;
            if (MEDIAN) {
                // filters.py:148-151 clips the scaled cost to [0, 255] before rounding.  Here every cost of the tile
                // is a cost of the layer, so lo <= c <= hi with the layer's own lo and hi: c - lo >= 0, and the
                // product with the float32 scale exceeds 255 by rounding only (it rounds to 255 either way).  A
                // flat layer has scale 0 and finite costs: the product is 0.  Neither the clip nor a select is needed.
                const uint32_t b = (uint32_t)rintf((c - qlo) * qscale);
                bytes[j] = b;
                c = __uint_as_float(b);
            }
            if (FILTER == STEREO_FILTER_BILATERAL) c *= a.vscale;
            if (idx < NP) cb[idx] = c;
        }
        if (MEDIAN && !census_levels) {
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                if (j == PT - 1 && last_slot_empty) continue;
                const uint32_t b = bytes[j];
                mylo = min(mylo, b);   // (a thread's surplus slot repeats a tile pixel)
                if (b == 255u) my255 = 1u; else myhi = max(myhi, b);
            }
        }
        if (MEDIAN && !census_levels) {
            // one LDS atomic triple per wavefront
            mylo = wave_min_u32(mylo);
            myhi = wave_max_u32(myhi);
            my255 = wave_max_u32(my255);
            if ((tid & 63) == 0) {
                atomicMin(&trange[0], mylo);
                atomicMax(&trange[1], myhi);
                if (my255) atomicOr(&trange[2], 1u);
            }
        }
        if (MEDX) {
            // range of the tile's codes
            mylo = wave_min_u32(mylo);
            myhi = wave_max_u32(myhi);
            if ((tid & 63) == 0) {
                atomicMin(&trange[0], mylo);
                atomicMax(&trange[1], myhi);
            }
        }
        __syncthreads();
        if ((MEDIAN || MEDX) && tid == 0) {
            uint32_t *later = trange_buf[(d - a.d_begin + 2) % 3];
            later[0] = MEDX ? 0xffffffffu : 255u; later[1] = 0u; later[2] = 0u;
        }

        float outv[FF_ROWS_PER_THREAD];
        const float *win = cb + (q * FF_ROWS_PER_THREAD) * LW + lx;
        if (FILTER == STEREO_FILTER_GAUSSIAN) {
            // The reference's weight table exp(-(x^2 + y^2) / 2 sigma^2) / sum (filters.py:24-39) is the outer
            // product of a normalised 1-D Gaussian with itself up to float32 rounding: rows first (shared
            // by the four outputs), then columns -- 2 (2r + 1) multiply-adds per output instead of (2r + 1)^2.
            float hrow[NR];
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                float h = 0.f;
#pragma unroll
                for (int dx = 0; dx < K; ++dx) h = fmaf(win[rr * LW + dx], a.kern1[dx], h);
                hrow[rr] = h;
            }
#pragma unroll
            for (int j = 0; j < FF_ROWS_PER_THREAD; ++j) {
                float o = 0.f;
#pragma unroll
                for (int dy = 0; dy < K; ++dy) o = fmaf(hrow[j + dy], a.kern1[dy], o);
                outv[j] = o;
            }
        } else if (MEDX) {
            const int tlo = (int)trange[0], thi = max((int)trange[1], tlo);
            int med[FF_ROWS_PER_THREAD];
            if (thi - tlo < 32) median_bisect<K, 6, LW, FF_ROWS_PER_THREAD>(win, tlo, thi, med);
            else if (thi - tlo < 128) median_bisect<K, 8, LW, FF_ROWS_PER_THREAD>(win, tlo, thi, med);
            else median_bisect<K, 12, LW, FF_ROWS_PER_THREAD>(win, tlo, thi, med);
#pragma unroll
            for (int j = 0; j < FF_ROWS_PER_THREAD; ++j)
                outv[j] = (uint32_t)med[j] == a.code_large ? a.cfg.large : (float)med[j] * a.cfg.unit;
        } else if (FILTER == STEREO_FILTER_BILATERAL) {
            // w = spatial * exp(-(v - c)^2 / (2 sigma^2))  (filters.py:299-304) evaluated as one hardware
            // exp2 per window element: exp2(log2(spatial) - (v' - c')^2) on costs scaled by
            // sqrt(log2(e) / (2 sigma^2)) when they were written to the tile (sum w v' / sum w is scaled back at
            // the end): a subtract, a fused multiply-add and the exponential per element.  The weights agree
            // with the reference's to a few float32 ulps, the filtered cost to ~1e-6.
            // (Scaling after the exact difference instead -- one multiply more per element, 63.9 against 58.1 ms on
            //  a 7680 x 4320 frame with 256 disparities -- was measured to change nothing that can be told from
            //  the CPU restatement: 99.881 % identical disparities on a strip of that frame against 99.894 % this
            //  way, every other pixel a near-tie in either case.  Which layer wins a near-tie is decided by the last
            //  bit of exp(): the CPU restatement with libm's expf agrees with the reference's own NumPy result on
            //  98.4 % and 99.1 % of the pixels of the two golden fixtures.)
            float c[FF_ROWS_PER_THREAD], wsum[FF_ROWS_PER_THREAD], vsum[FF_ROWS_PER_THREAD];
#pragma unroll
            for (int j = 0; j < FF_ROWS_PER_THREAD; ++j) {
                c[j] = win[(j + RAD) * LW + RAD];
                wsum[j] = 0.f; vsum[j] = 0.f;
            }
            if (RAD <= FF_BIL_UNROLL_MAX) {
                // a window row is read once for the (up to four) outputs it belongs to
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) {
                    float v[K];
#pragma unroll
                    for (int dx = 0; dx < K; ++dx) v[dx] = win[rr * LW + dx];
#pragma unroll
                    for (int j = 0; j < FF_ROWS_PER_THREAD; ++j) {
                        const int dy = rr - j;
                        if (dy >= 0 && dy < K) {
#pragma unroll
                            for (int dx = 0; dx < K; ++dx) {
                                const float diff = v[dx] - c[j];
                                const float w = __builtin_amdgcn_exp2f(fmaf(-diff, diff, a.kern[dy * K + dx]));
                                wsum[j] += w;
                                vsum[j] = fmaf(v[dx], w, vsum[j]);
                            }
                        }
                    }
                }
            } else {
                // large windows: the rows are a run-time loop (fully unrolled, four windows of up to 289 elements
                // would not fit the instruction cache)
                // Two horizontally adjacent elements of a window travel through packed instructions (difference,
                // exponent, the two sums); an output's sums are kept as (even elements, odd elements) pairs.
                typedef float fpair __attribute__((ext_vector_type(2)));
                fpair ws2[FF_ROWS_PER_THREAD], vs2[FF_ROWS_PER_THREAD];
#pragma unroll
                for (int j = 0; j < FF_ROWS_PER_THREAD; ++j) { ws2[j] = fpair{0.f, 0.f}; vs2[j] = fpair{0.f, 0.f}; }
#pragma unroll 1
                for (int rr = 0; rr < NR; ++rr) {
                    float v[K];
#pragma unroll
                    for (int dx = 0; dx < K; ++dx) v[dx] = win[rr * LW + dx];
#pragma unroll
                    for (int j = 0; j < FF_ROWS_PER_THREAD; ++j) {
                        const int dy = rr - j;
                        if (dy >= 0 && dy < K) {
                            const float *kr;
                            if constexpr (BIL_LDS_TABLE) kr = &lgtab[dy * (K + 1)]; else kr = a.kernp[dy];
                            const fpair cc = fpair{c[j], c[j]};
#pragma unroll
                            for (int dx = 0; dx + 1 < K; dx += 2) {
                                const fpair vv = fpair{v[dx], v[dx + 1]};
                                const fpair diff = vv - cc;
                                const fpair arg = __builtin_elementwise_fma(-diff, diff, fpair{kr[dx], kr[dx + 1]});
                                const fpair w = fpair{__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
                                ws2[j] += w;
                                vs2[j] = __builtin_elementwise_fma(vv, w, vs2[j]);
                            }
                            {   // K is odd: the last element of the row
                                const float diff = v[K - 1] - c[j];
                                const float w = __builtin_amdgcn_exp2f(fmaf(-diff, diff, kr[K - 1]));
                                ws2[j].x += w;
                                vs2[j].x = fmaf(v[K - 1], w, vs2[j].x);
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < FF_ROWS_PER_THREAD; ++j) { wsum[j] = ws2[j].x + ws2[j].y; vsum[j] = vs2[j].x + vs2[j].y; }
            }
#pragma unroll
            for (int j = 0; j < FF_ROWS_PER_THREAD; ++j) outv[j] = ((wsum[j] <= 0.f) ? c[j] : vsum[j] / wsum[j]) * a.inv_vscale;
        } else {
            // the bytes of this tile (halo included) lie in [tlo, thi]: so do all its medians.  Cost
            // layers mostly hold a narrow band of levels (plus 255 where a cost is `large`).
            const int tlo = census_levels ? 0 : (int)trange[0];
            const int thi2 = census_levels ? 26 : max((int)trange[1], tlo);
            const int has255 = census_levels ? 1 : (int)trange[2];
            const int thi = has255 ? 255 : thi2;
            int med[FF_ROWS_PER_THREAD];
            if (BITSLICED && thi2 - tlo <= 30 && !a.plain_median) {
                // phase 1: counts of the K-row column segments under this thread's outputs, for halo column lx; the
                // 2 * RAD columns beyond 63 are taken by the first lanes of one wavefront
                auto column_counts = [&](int c, int qq, auto census_tag) {
                    uint32_t m[NR];
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) {
                        const uint32_t b = __float_as_uint(cb[(qq * FF_ROWS_PER_THREAD + rr) * LW + c]);
                        // (census layers: bytes are 0 .. 26 or 255 and the tile's lowest level is 0)
                        if (decltype(census_tag)::value) m[rr] = 0xffffffffu << min(b, 31u);
                        else m[rr] = 0xffffffffu << (b == 255u ? 31u : b - (uint32_t)tlo);
                    }
#pragma unroll
                    for (int j = 0; j < FF_ROWS_PER_THREAD; ++j) {
                        uint32_t bits[NB1];
                        bs_sum<K>(m + j, bits);
                        const int o = (qq * FF_ROWS_PER_THREAD + j) * BS_CW + c;
#pragma unroll
                        for (int pl = 0; pl < NB1; ++pl) colcnt[pl * FF_TH * BS_CW + o] = bits[pl];
                    }
                };
                // (the wavefront that takes the extra columns changes from tile to tile: the first wavefronts of
                //  the workgroups that share a compute unit would otherwise all load the same SIMD)
                const int xt = (tid - 64 * (int)((blockIdx.x + blockIdx.y) & 3u)) & (FF_THREADS - 1);
                const bool extra = xt < 4 * (LW - FF_TW);   // (8 * RAD jobs: one wavefront up to radius 8, two above)
                static_assert(4 * (LW - FF_TW) <= 128, "the extra columns are dealt to at most two wavefronts");
                if (census_levels) {
                    column_counts(lx, q, std::true_type{});
                    if (extra) column_counts(FF_TW + xt % (LW - FF_TW), xt / (LW - FF_TW), std::true_type{});
                } else {
                    column_counts(lx, q, std::false_type{});
                    if (extra) column_counts(FF_TW + xt % (LW - FF_TW), xt / (LW - FF_TW), std::false_type{});
                }
                __syncthreads();
                // phase 2: add the K columns of each window, find the first level with count >= K * K / 2 + 1
#pragma unroll
                for (int j = 0; j < FF_ROWS_PER_THREAD; ++j) {
                    const uint32_t *row = colcnt + (q * FF_ROWS_PER_THREAD + j) * BS_CW + lx;
                    uint32_t u[NB1][NB1];  // u[p]: sum of plane p over the K columns
#pragma unroll
                    for (int pl = 0; pl < NB1; ++pl) {
                        uint32_t m[K];
#pragma unroll
                        for (int dx = 0; dx < K; ++dx) m[dx] = row[pl * FF_TH * BS_CW + dx];
                        bs_sum<K>(m, u[pl]);
                    }
                    uint32_t ge;
                    if constexpr (K == 7) {
                        // S = U0 + 2 U1 + 4 U2, bit planes s0 .. s5 (S <= 49)
                        const uint32_t s0 = u[0][0];
                        const uint32_t s1 = u[0][1] ^ u[1][0], c2 = u[0][1] & u[1][0];
                        const uint32_t x4 = bs_xor3(u[0][2], u[1][1], u[2][0]), y4 = bs_maj3(u[0][2], u[1][1], u[2][0]);
                        const uint32_t s2 = x4 ^ c2, c4 = x4 & c2;
                        const uint32_t x8 = bs_xor3(u[1][2], u[2][1], y4), y8 = bs_maj3(u[1][2], u[2][1], y4);
                        const uint32_t s3 = x8 ^ c4, c8 = x8 & c4;
                        const uint32_t s4 = bs_xor3(u[2][2], y8, c8), s5 = bs_maj3(u[2][2], y8, c8);
                        // S >= 25 = 16 + 8 + 1
                        ge = s5 | bs_and3(s4, s3, bs_or3(s2, s1, s0));
                    } else {
                        uint32_t sb[NB2];
                        bs_weighted_sum<NB1, NB2, 0, 0>(u, nullptr, sb);
                        ge = bs_at_least<NB2, (K * K) / 2 + 1>(sb);
                    }
                    const int t = __builtin_ctz(ge);   // bit 31 is always set: all K * K elements are <= level 31
                    med[j] = t == 31 ? (has255 ? 255 : tlo + 31) : tlo + t;
                }
            } else {
                // Bytes other than 255 more than 30 levels apart.  Census costs never get here (a cost that is
                // not `large` is at most a tenth of it, so it quantises to 26 or less above the minimum); the
                // plain bisection below, which re-reads the window from LDS at every step, keeps the kernel
                // correct for any input without the registers of the packed variant.
#pragma unroll 1
                for (int j = 0; j < FF_ROWS_PER_THREAD; ++j) {
                    int lo = tlo, hi = thi;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        int cnt = 0;
#pragma unroll 1
                        for (int dy = 0; dy < K; ++dy)
#pragma unroll
                            for (int dx = 0; dx < K; ++dx) cnt += (int)__float_as_uint(win[(j + dy) * LW + dx]) <= mid;
                        if (cnt > (K * K) / 2) hi = mid; else lo = mid + 1;
                    }
                    med[j] = lo;
                }
            }
#pragma unroll
            for (int j = 0; j < FF_ROWS_PER_THREAD; ++j)
                outv[j] = flat ? qlo : (float)med[j] * qinv + qlo;   // two roundings (contraction is off in this file)
        }
#pragma unroll
        for (int j = 0; j < FF_ROWS_PER_THREAD; ++j) {
            const float v = outv[j];
            if (v == v) {
                const uint64_t key = make_key(v, (uint32_t)d);
                best[j] = key < best[j] ? key : best[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < FF_ROWS_PER_THREAD; ++j) {
        const int gy = y0 + q * FF_ROWS_PER_THREAD + j, gx = x0 + lx;
        if (gy < H && gx < W) a.keys[(size_t)gy * W + gx] = best[j];
    }
}

bool fused_filter_supported(int filter_kind, int radius) {
    // (the exact median in addition needs exactly summable census weights: launch_fused_filter refuses the others)
    return (filter_kind == STEREO_FILTER_MEDIAN_U8Q || filter_kind == STEREO_FILTER_GAUSSIAN ||
            filter_kind == STEREO_FILTER_BILATERAL || filter_kind == STEREO_FILTER_MEDIAN) &&
           radius >= 1 && radius <= (filter_kind == STEREO_FILTER_MEDIAN_U8Q ? FF_MAX_MEDIAN_RADIUS :
                                     filter_kind == STEREO_FILTER_GAUSSIAN ? FF_MAX_GAUSS_RADIUS :
                                     filter_kind == STEREO_FILTER_BILATERAL ? FF_MAX_BIL_RADIUS : FF_MAX_RADIUS);
}

size_t fused_filter_minmax_bytes(int n_disp) { return (size_t)n_disp * 8; }

template <int FILTER, bool WORD64, int RAD>
static int launch_ff(const FFArgs &a, hipStream_t s) {
    dim3 grid(ceil_div(a.W, FF_TW), ceil_div(a.H, FFTile<FILTER, RAD>::TH));
    // (the exact median works on integer codes: it has one form only)
    if (a.cfg.exact && a.cfg.unit > 0.f && FILTER != STEREO_FILTER_MEDIAN)
        hipLaunchKernelGGL((fused_filter_kernel<FILTER, WORD64, RAD, true>), grid, dim3(FF_THREADS), 0, s, a);
    else
        hipLaunchKernelGGL((fused_filter_kernel<FILTER, WORD64, RAD, false>), grid, dim3(FF_THREADS), 0, s, a);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}
template <int FILTER, bool WORD64>
static int launch_ff_r(const FFArgs &a, int radius, hipStream_t s) {
    switch (radius) {
        case 1: return launch_ff<FILTER, WORD64, 1>(a, s);
        case 2: return launch_ff<FILTER, WORD64, 2>(a, s);
        case 3: return launch_ff<FILTER, WORD64, 3>(a, s);
        case 4: return launch_ff<FILTER, WORD64, 4>(a, s);
        case 5: return launch_ff<FILTER, WORD64, 5>(a, s);
        case 6: return launch_ff<FILTER, WORD64, 6>(a, s);
        case 7: return launch_ff<FILTER, WORD64, 7>(a, s);
        case 8: return launch_ff<FILTER, WORD64, 8>(a, s);
        default: break;
    }
    if constexpr (FILTER == STEREO_FILTER_MEDIAN_U8Q || FILTER == STEREO_FILTER_GAUSSIAN || FILTER == STEREO_FILTER_BILATERAL) {
        switch (radius) {
            case 9: return launch_ff<FILTER, WORD64, 9>(a, s);
            case 10: return launch_ff<FILTER, WORD64, 10>(a, s);
            case 11: return launch_ff<FILTER, WORD64, 11>(a, s);
            case 12: return launch_ff<FILTER, WORD64, 12>(a, s);
            default: break;
        }
    }
    set_error("fused filter kernels support radius 1..%d (exact median: 1..%d)", FF_MAX_MEDIAN_RADIUS, FF_MAX_RADIUS);
    return STEREO_EUNSUPPORTED;
}
template <int FILTER>
static int launch_ff_w(const FFArgs &a, int radius, hipStream_t s) {
    return a.cfg.word64 ? launch_ff_r<FILTER, true>(a, radius, s) : launch_ff_r<FILTER, false>(a, radius, s);
}

// d_mm: scratch of fused_filter_minmax_bytes(d_end - d_begin) bytes (median only)
int launch_fused_filter(const void *d_lw, const void *d_rw, const CensusCfg &cfg, int H, int W, int filter_kind,
                        int radius, double sigma, int d_begin, int d_end, uint32_t *d_mm, uint64_t *d_keys,
                        hipStream_t s, bool plain_median) {
    if (H <= 0 || W <= 0) return STEREO_OK;
    if (!fused_filter_supported(filter_kind, radius)) { set_error("unsupported fused filter"); return STEREO_EUNSUPPORTED; }
    // (the kernels address the census words by 32-bit byte offsets)
    if ((size_t)H * W * 8 > 0xffffffffu) { set_error("image too large for the fused filter kernels"); return STEREO_EUNSUPPORTED; }
    if (d_begin >= d_end) return launch_fill_keys(d_keys, (size_t)H * W, s);
    FFArgs a;
    memset(&a, 0, sizeof(a));
    a.lw = d_lw; a.rw = d_rw; a.cfg = cfg; a.H = H; a.W = W; a.d_begin = d_begin; a.d_end = d_end;
    a.mm = d_mm; a.keys = d_keys; a.plain_median = plain_median ? 1 : 0;
    if (filter_kind == STEREO_FILTER_MEDIAN) {
        if (!(cfg.exact && cfg.unit > 0.f)) { set_error("the fused exact median needs exactly summable census weights"); return STEREO_EUNSUPPORTED; }
        // largest valid cost in units: all 8 directions x all rings set = 8 * (2^field_bits - 1)
        a.code_large = (cfg.word64 ? 8u * 255u : 8u * 15u) + 1u;
        return launch_ff_w<STEREO_FILTER_MEDIAN>(a, radius, s);
    }
    if (filter_kind != STEREO_FILTER_MEDIAN_U8Q) {
        if (filter_kind == STEREO_FILTER_GAUSSIAN) {
            // float32 exponentials as the reference forms them, normalised in float64
            const float denom32 = (float)(2.0 * sigma * sigma);
            double e[2 * FF_MAX_GAUSS_RADIUS + 1], sum = 0.0;
            for (int t = -radius; t <= radius; ++t) {
                const float ft = (float)t;
                e[t + radius] = (double)std::exp(-(ft * ft) / denom32);
                sum += e[t + radius];
            }
            for (int t = 0; t <= 2 * radius; ++t) a.kern1[t] = (float)(e[t] / sum);
        }
        if (filter_kind == STEREO_FILTER_BILATERAL && radius > FF_MAX_RADIUS) {
            // the reference's normalised float32 table (filters.py:24-39) is exp(-dy^2/2s^2) exp(-dx^2/2s^2) / sum up to
            // the rounding of its float32 exponentials: log2 k(dy, dx) = lg1[dy] + lg1[dx] with
            // lg1[t] = log2 k(t, centre) - log2 k(centre, centre) / 2, to about 1e-7 of the weight
            const int size = 2 * radius + 1;
            std::vector<float> k2(size * size);
            gaussian_kernel_host(radius, sigma, k2.data());
            const double half_centre = 0.5 * std::log2((double)k2[radius * size + radius]);
            for (int t = 0; t < size; ++t) a.kern_lg1[t] = (float)(std::log2((double)k2[t * size + radius]) - half_centre);
        } else if (filter_kind == STEREO_FILTER_BILATERAL) {
            gaussian_kernel_host(radius, sigma, a.kern);
            const int kk = (2 * radius + 1) * (2 * radius + 1);
            for (int i = 0; i < kk; ++i) a.kern[i] = (float)std::log2((double)a.kern[i]);
            for (int y = 0; y <= 2 * radius; ++y)
                for (int x = 0; x <= 2 * radius; ++x) a.kernp[y][x] = a.kern[y * (2 * radius + 1) + x];
        }
        if (filter_kind == STEREO_FILTER_BILATERAL) {
            const double denom = (double)(float)(2.0 * (sigma * sigma));   // the reference's float32 denominator
            a.vscale = (float)std::sqrt(1.4426950408889634 / denom);
            a.inv_vscale = (float)(1.0 / (double)a.vscale);
        }
    }
    switch (filter_kind) {
        case STEREO_FILTER_MEDIAN_U8Q: {
            const int nd = d_end - d_begin;
            hipLaunchKernelGGL(minmax_fill_kernel, dim3(ceil_div(nd, 256)), dim3(256), 0, s, d_mm, nd);
            SHP_LAUNCH_CHECK();
            const bool exact = cfg.exact && cfg.unit > 0.f;
            dim3 grid(ceil_div(W, 64 * 8), ceil_div(H, 4 * (exact ? MMX_ROWS : MM_ROWS)));
            const bool hil = cfg.large > 0.f;
            if (cfg.word64) {
                auto kern = exact ? (hil ? cost_minmax_exact_kernel<uint64_t, true> : cost_minmax_exact_kernel<uint64_t, false>)
                                  : cost_minmax_kernel<uint64_t>;
                hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, static_cast<const uint64_t *>(d_lw),
                                   static_cast<const uint64_t *>(d_rw), cfg, H, W, d_begin, d_end, d_mm);
            } else {
                auto kern = exact ? (hil ? cost_minmax_exact_kernel<uint32_t, true> : cost_minmax_exact_kernel<uint32_t, false>)
                                  : cost_minmax_kernel<uint32_t>;
                hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, static_cast<const uint32_t *>(d_lw),
                                   static_cast<const uint32_t *>(d_rw), cfg, H, W, d_begin, d_end, d_mm);
            }
            SHP_LAUNCH_CHECK();
            return launch_ff_w<STEREO_FILTER_MEDIAN_U8Q>(a, radius, s);
        }
        case STEREO_FILTER_GAUSSIAN: return launch_ff_w<STEREO_FILTER_GAUSSIAN>(a, radius, s);
        default: return launch_ff_w<STEREO_FILTER_BILATERAL>(a, radius, s);
    }
}

}  // namespace shp
