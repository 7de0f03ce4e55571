/* oracle/trig.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * cos / sin of the two reference call sites whose results the HIP kernels must match bit for bit
 * (SimpleRace, simple_race_simulator.cpp:227-243,386-430; XItem::get_item_image -> cv::getRotationMatrix2D,
 * xitem.cpp:47-60).  The reference calls the C library there.  Default here: include/xwb_trig.h, the one
 * deterministic definition the product uses as well -- a GPU's libm and glibc do not round the last double bit
 * alike, so "libm" is not a definition two machines can share.  orc_set_trig_libm(1) switches these call sites
 * back to the host's libm, which is how tests/test_trig.py measures what the substitution changes (reward /
 * observation bits of SimpleRace rollouts, pixels of warped goal icons): nothing, on everything it samples. */
#include "oracle.h"
#include <math.h>

#include "../include/xwb_trig.h"

static int g_trig_libm = 0;

void orc_set_trig_libm(int on) { g_trig_libm = on ? 1 : 0; }
int  orc_get_trig_libm(void) { return g_trig_libm; }

double orc_trig_cos(double x) { return g_trig_libm ? cos(x) : xwb_cos(x); }
double orc_trig_sin(double x) { return g_trig_libm ? sin(x) : xwb_sin(x); }

/* bare entry points for the accuracy test */
void orc_xwb_sincos(double x, double *s, double *c) { xwb_sincos(x, s, c); }
