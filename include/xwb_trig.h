/*
 * include/xwb_trig.h -- one deterministic double-precision sin / cos for libxwb.so and for its checker.
 *
 * Why: the reference calls the C library's double cos / sin in two places on the path
 *   - SimpleRace: BaseCar::move, RaceEngine::get_reward / get_screen (games/simple_race/simple_race_simulator.cpp:
 *     227-235, 386-430) -- results are narrowed to float at once;
 *   - XWorld2D egocentric: XItem::get_item_image -> cv::getRotationMatrix2D (games/xworld/xworld/xitem.cpp:47-60) --
 *     results feed a double matrix that is inverted and rounded to fixed point;
 * and a GPU's math library (ROCm ocml) and the host's (glibc) do not round the last double bit alike.  Bit-exact
 * parity between the HIP kernels and the CPU restatement therefore needs ONE definition of sin / cos that gives
 * the same bits on every target.  This is it: IEEE-754 double +, -, *, floor and comparisons only, in a fixed order,
 * no FMA (both sides build with -ffp-contract=off), no table, no library call.
 *
 * Accuracy: < 1 ulp (the classic 13th / 14th degree minimax kernels on [-pi/4, pi/4] after a two-step Cody-Waite
 * reduction with 33-bit pieces of pi/2, good to ~118 bits for |x| < 2^20 * pi/2).  glibc's results (< 1 ulp too)
 * differ from these in the last bit for a small fraction of arguments; narrowed to float the two agree except once in
 * ~2^29 calls.  tests/test_trig.py measures both rates against libm on the arguments the path can produce.
 *
 * The polynomial coefficients are the published minimax coefficients of the FreeBSD / fdlibm kernels
 * (k_sin.c, k_cos.c; Copyright (C) 1993 by Sun Microsystems, Inc.  Permission to use, copy, modify, and distribute
 * this software is freely granted, provided that this notice is preserved).
 */
#ifndef XWB_TRIG_H
#define XWB_TRIG_H

#if defined(__HIPCC__)
#define XWB_TRIG_FN __host__ __device__ static __forceinline__
#else
#include <math.h>
#define XWB_TRIG_FN static inline
#endif

/* sin(x + y) for |x| <= pi/4, y the tail of x */
XWB_TRIG_FN double xwb_ksin(double x, double y) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x;
    const double w = z * z;
    const double r = S2 + z * (S3 + z * S4) + z * w * (S5 + z * S6);
    const double v = z * x;
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

/* cos(x + y) for |x| <= pi/4, y the tail of x */
XWB_TRIG_FN double xwb_kcos(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    double w = z * z;
    const double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
    const double hz = 0.5 * z;
    w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}

/* *s = sin(x), *c = cos(x); |x| < ~1.6e6 (the path's arguments are below 7) */
XWB_TRIG_FN void xwb_sincos(double x, double *s, double *c) {
    const double INV_PIO2 = 6.36619772367581382433e-01;      /* 2 / pi */
    const double PIO2_1 = 1.57079632673412561417e+00;        /* first 33 bits of pi / 2 */
    const double PIO2_2 = 6.07710050630396597660e-11;        /* next 33 bits */
    const double PIO2_2T = 2.02226624879595063154e-21;       /* pi / 2 - (PIO2_1 + PIO2_2) */
    const double fn = floor(x * INV_PIO2 + 0.5);
    /* fn * PIO2_1 and fn * PIO2_2 are exact (33 + 20 bits); the tail keeps what the second subtraction rounds off */
    const double t = x - fn * PIO2_1;
    double w = fn * PIO2_2;
    const double r = t - w;
    w = fn * PIO2_2T - ((t - r) - w);
    const double y0 = r - w;
    const double y1 = (r - y0) - w;
    const double ks = xwb_ksin(y0, y1), kc = xwb_kcos(y0, y1);
    const int q = (int)((long long)fn & 3);
    *s = q == 0 ? ks : (q == 1 ? kc : (q == 2 ? -ks : -kc));
    *c = q == 0 ? kc : (q == 1 ? -ks : (q == 2 ? -kc : ks));
}

XWB_TRIG_FN double xwb_sin(double x) { double s, c; xwb_sincos(x, &s, &c); return s; }
XWB_TRIG_FN double xwb_cos(double x) { double s, c; xwb_sincos(x, &s, &c); return c; }

#endif /* XWB_TRIG_H */
