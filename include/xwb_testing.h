/* xwb_testing.h -- test and measurement hooks of libxwb.so.  NOT part of the drop-in boundary (include/xwb.h): exported under
 * their own version node (XWB_TESTING, csrc/libxwb.map) for tests/, tools/ and bench.py; a reference-side binding never needs them. */
#ifndef XWB_TESTING_H
#define XWB_TESTING_H
#include "xwb.h"
#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* enqueue on `stream` a wait for an epoch nobody publishes, with a watchdog of budget_us microseconds -- the batch is poisoned
 * once it expires (tests/test_gpu_queue_sync.py) */
int xwb_debug_stall_handoff(xwb_sim *sim, void *stream, int64_t budget_us);

/* average duration in microseconds of the named kernel ("step", "render" = the whole-batch render -- with the step's blocks
 * inside it on XWB_PATH_LAZY_FUSED --, "reset" = the map generator, "list" = the render of the envs a reset started) over the
 * launches recorded since xwb_profile_begin (hipEvents on the stream each launch runs on): bench.py's roofline.achieved */
int xwb_profile_begin(xwb_sim *sim);
int xwb_profile_end(xwb_sim *sim, void *stream, const char *kernel, double *avg_us, int64_t *launches);
int xwb_profile_stop(xwb_sim *sim);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
