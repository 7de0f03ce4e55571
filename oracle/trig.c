/* oracle/trig.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * cos / sin of the two reference call sites whose results the HIP kernels must match bit for bit
 * (SimpleRace, simple_race_simulator.cpp:227-243,386-430; XItem::get_item_image -> cv::getRotationMatrix2D,
 * xitem.cpp:47-60).  The reference calls the C library there, and so does the oracle BY DEFAULT: the host's libm,
 * i.e. a checker that shares no arithmetic source with the product (the HIP kernels use include/xwb_trig.h, because a
 * GPU's math library and glibc do not round the last double bit alike).  Every GPU parity test therefore compares the
 * kernels with libm results; they are expected to agree exactly because both call sites narrow (SimpleRace: to float;
 * the goal warp: to 1/1024-pixel fixed point), and tests/test_trig.py measures that on the CPU.
 * orc_set_trig_libm(0) switches these call sites to include/xwb_trig.h -- the product's definition -- which is how
 * tests/test_trig.py and the `trig` parameter of the GPU tests measure what the substitution changes (nothing, on
 * everything they sample). */
#include "oracle.h"
#include <math.h>

#include "../include/xwb_trig.h"

static int g_trig_libm = 1;

void orc_set_trig_libm(int on) { g_trig_libm = on ? 1 : 0; }
int  orc_get_trig_libm(void) { return g_trig_libm; }

double orc_trig_cos(double x) { return g_trig_libm ? cos(x) : xwb_cos(x); }
double orc_trig_sin(double x) { return g_trig_libm ? sin(x) : xwb_sin(x); }

/* bare entry points for the accuracy test */
void orc_xwb_sincos(double x, double *s, double *c) { xwb_sincos(x, s, c); }
