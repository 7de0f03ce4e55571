/*
 * include/xwb_minstd.h -- the reference's thread-local RNG (simulator_util.cpp:38-73), for replaying seeded reference runs.
 *
 * The reference draws from one std::default_random_engine per simulator thread -- libstdc++: minstd_rand0,
 * x <- 16807 x mod (2^31 - 1), outputs in [1, 2147483646] -- through
 *   util::get_rand_ind(n)       = std::uniform_int_distribution<int>(0, n - 1)(engine)
 *   util::get_rand_range_val(u) = std::uniform_real_distribution<float>(0, u)(engine)
 * and, when FLAGS_simulator_seed != 0, seeds the n-th thread's engine with
 *   int(std::hash<std::string>()(std::to_string(FLAGS_simulator_seed + n)))          (n = 1, 2, ... in creation order).
 * The two distributions are restated here from libstdc++'s algorithms (bits/uniform_int_dist.h: the down-scaling branch
 * `scaling = range / n; past = n * scaling; reject r >= past; r / scaling`; bits/random.tcc generate_canonical<float, 24>:
 * one engine call, float(x - 1) / float(2147483646.0L), clamped below 1) as inline functions for host and device; the
 * reference's own known answers (tests/test_simulator_seed.cpp:22-50) are checked through libxwb's exports of them
 * (xwb_minstd_*, include/xwb.h).  With rng_mode = XWB_RNG_MINSTD every env of a batch owns one engine state (global env id
 * g <-> the reference's thread number thread_base + g + 1) and SimpleRace's random reset (simple_race_simulator.cpp:86-89,
 * 196-199, 237-243, 267-284) and the teacher's task draw (teaching_task.cpp:204-213) consume it exactly as the reference
 * does.  Map generation stays on xwb-rng-v1: the reference generates maps with CPython's unseeded `random`.
 */
#ifndef XWB_MINSTD_H
#define XWB_MINSTD_H

#include <stdint.h>

#if defined(__HIPCC__)
#define XWB_MINSTD_FN __host__ __device__ static __forceinline__
#else
#define XWB_MINSTD_FN static inline
#endif

XWB_MINSTD_FN uint32_t xwb_minstd_seed_value(int64_t seed_as_int) {
    /* linear_congruential_engine::seed(result_type s): the int converts to unsigned long first; c == 0 so 0 maps to 1 */
    const uint64_t x = (uint64_t)seed_as_int % 2147483647ull;
    return (uint32_t)(x == 0 ? 1 : x);
}

XWB_MINSTD_FN uint32_t xwb_minstd_next(uint32_t *x) {
    *x = (uint32_t)((16807ull * (uint64_t)*x) % 2147483647ull);
    return *x;
}

/* util::get_rand_ind(size), size >= 1 */
XWB_MINSTD_FN int32_t xwb_minstd_rand_ind_state(uint32_t *x, int32_t size) {
    const uint64_t urng_range = 2147483645ull;             /* max() - min() */
    const uint64_t n = (uint64_t)size;
    if (urng_range + 1 == n) return (int32_t)(xwb_minstd_next(x) - 1u);
    const uint64_t scaling = urng_range / n, past = n * scaling;
    uint64_t r;
    do { r = (uint64_t)xwb_minstd_next(x) - 1ull; } while (r >= past);
    return (int32_t)(r / scaling);
}

/* util::get_rand_range_val(upper) */
XWB_MINSTD_FN float xwb_minstd_rand_range_state(uint32_t *x, float upper) {
    float c = (float)(xwb_minstd_next(x) - 1u) / 2147483648.0f;     /* float(2147483646.0L) rounds to 2^31 */
    if (c >= 1.0f) c = 0.99999994f;                                  /* nextafter(1.0f, 0.0f) */
    return c * (upper - 0.0f) + 0.0f;
}

#endif /* XWB_MINSTD_H */
