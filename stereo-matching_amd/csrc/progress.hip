// Row-band progress words: how a caller learns, while the fused kernel is still running, that
// the first rows of the frame are final and can be shipped to the other GPUs.
//
// words[0 .. 15]   finished workgroups per band (the kernel counts, the launcher clears)
// words[16 .. 31]  band flags: the launch's epoch once the band's keys are final
// words[32 .. 47]  time-out marks: non-zero once a wait for the band gave up (never cleared by the library:
//                  the caller clears them with the rest when it allocates the words, and must treat a
//                  non-zero mark as a failed frame)
//
// The flags are set by the fused kernel where it supports it and, in every case, by
// progress_finish_kernel enqueued right after it, so a waiter never depends on which kernel ran.
#include "common.hpp"

namespace shp {

__global__ __launch_bounds__(64) void progress_finish_kernel(uint32_t *words, int n_bands, uint32_t epoch) {
    const int b = threadIdx.x;
    if (b < n_bands)
        __hip_atomic_store(&words[PROGRESS_MAX_BANDS + b], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One lane polls the band's flag and sleeps between polls.  Epochs compare modulo 2^32.  The
// loop is bounded (about ten seconds of the 100 MHz wall clock): a producer that never ran must
// not leave a wavefront spinning forever; the time-out is recorded in words[32 + band].
__global__ __launch_bounds__(64) void progress_wait_kernel(uint32_t *words, int band, uint32_t epoch, uint64_t max_ticks) {
    if (threadIdx.x != 0) return;
    const uint64_t t0 = wall_clock64();
    for (;;) {
        const uint32_t seen = __hip_atomic_load(&words[PROGRESS_MAX_BANDS + band], __ATOMIC_ACQUIRE,
                                                __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int32_t)(seen - epoch) >= 0) break;
        if (wall_clock64() - t0 > max_ticks) {
            __hip_atomic_store(&words[2 * PROGRESS_MAX_BANDS + band], epoch ? epoch : 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        __builtin_amdgcn_s_sleep(64);
    }
}

int launch_progress_finish(uint32_t *words, int n_bands, uint32_t epoch, hipStream_t s) {
    if (n_bands < 1 || n_bands > PROGRESS_MAX_BANDS) { set_error("1..%d bands", PROGRESS_MAX_BANDS); return STEREO_EINVAL; }
    hipLaunchKernelGGL(progress_finish_kernel, dim3(1), dim3(64), 0, s, words, n_bands, epoch);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}

int launch_progress_wait(const uint32_t *words, int band, uint32_t epoch, hipStream_t s) {
    if (band < 0 || band >= PROGRESS_MAX_BANDS) { set_error("band %d outside 0..%d", band, PROGRESS_MAX_BANDS - 1); return STEREO_EINVAL; }
    hipLaunchKernelGGL(progress_wait_kernel, dim3(1), dim3(64), 0, s, const_cast<uint32_t *>(words), band, epoch,
                       (uint64_t)1000000000ull);
    SHP_LAUNCH_CHECK();
    return STEREO_OK;
}

// Whether work enqueued on stream `b` can run while a kernel enqueued EARLIER on stream `a` is still running.  The
// runtime maps streams onto a few hardware queues (four by default) and two streams that share one are served in
// the order of their submissions: a band wait on such a stream would start only after the search it is meant to
// accompany.  The mapping cannot be asked for, so it is tried out: a kernel on `a` waits (at most two milliseconds)
// for a flag that a kernel on `b`, enqueued after it, sets.  Synchronises both streams; allocates (set-up time only).
int streams_overlap(hipStream_t a, hipStream_t b, bool *overlap) {
    DevBuf words;
    SHP_TRY(words.alloc(sizeof(uint32_t) * 3 * PROGRESS_MAX_BANDS));
    SHP_HIP(hipMemsetAsync(words.p, 0, sizeof(uint32_t) * 3 * PROGRESS_MAX_BANDS, a));
    SHP_HIP(hipStreamSynchronize(a));
    SHP_HIP(hipStreamSynchronize(b));
    hipLaunchKernelGGL(progress_wait_kernel, dim3(1), dim3(64), 0, a, words.as<uint32_t>(), 0, 1u, (uint64_t)200000ull);
    SHP_LAUNCH_CHECK();
    hipLaunchKernelGGL(progress_finish_kernel, dim3(1), dim3(64), 0, b, words.as<uint32_t>(), 1, 1u);
    SHP_LAUNCH_CHECK();
    SHP_HIP(hipStreamSynchronize(a));
    SHP_HIP(hipStreamSynchronize(b));
    uint32_t timed_out = 0;
    SHP_HIP(hipMemcpy(&timed_out, words.as<uint32_t>() + 2 * PROGRESS_MAX_BANDS, 4, hipMemcpyDeviceToHost));
    *overlap = timed_out == 0;
    return STEREO_OK;
}

}  // namespace shp
