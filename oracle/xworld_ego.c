/* oracle/xworld_ego.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The egocentric view of XWorld2D (FLAGS_visible_radius > 0), restated on the CPU:
 *   XItem::get_item_image      games/xworld/xworld/xitem.cpp:33-63   per-item rotation / scale / offset warp
 *   XMap::to_image             games/xworld/xworld/xmap.cpp:125-206  canvas, black padding, crop, shadows, view rotation
 *   XMap::image_masking        games/xworld/xworld/xmap.cpp:273-362  field-of-view rectangle + wall shadows
 *   XWorldSimulator::get_screen_rgb / down_sample_image  xworld_simulator.cpp:287-307,508-545  the two resizes
 * and the OpenCV 3.2.0 functions they call (third party, pinned by cmake/opencv.cmake:5-6; absent from this image,
 * restated from the library's published algorithm -- PIXEL PARITY UNPINNED, as for the full-observation render):
 *   cv::getRotationMatrix2D    imgwarp.cpp: alpha = cos(a)*s, beta = sin(a)*s, [a b (1-a)cx - b cy; -b a b cx + (1-a)cy]
 *   cv::warpAffine, INTER_LINEAR, BORDER_CONSTANT on CV_8UC3: the matrix is inverted in double; source coordinates
 *     are fixed point with AB_BITS = 10 (adelta/bdelta = cvRound(M*x*1024), round_delta = 16), reduced to
 *     INTER_BITS = 5 fractional bits; cv::remap then blends the 2x2 neighbourhood with the 15-bit BilinearTab_i
 *     weights ((32-fx)(32-fy)*32, ...; the all-integer entry saturates to 32767 and is compensated by +1 on the
 *     diagonal weight, which never changes an 8-bit result) and rounds with (v + (1 << 14)) >> 15; neighbours
 *     outside the source take the border value.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "xworld_internal.h"

static int cv_round(double v) { return (int)lrint(v); }            /* cvRound: round half to even */

void orc_cv_get_rotation_matrix_2d(double cx, double cy, double angle_deg, double scale, double M[6]) {
    double angle = angle_deg * 3.1415926535897932384626433832795 / 180;   /* CV_PI */
    double alpha = orc_trig_cos(angle) * scale, beta = orc_trig_sin(angle) * scale;   /* trig.c */
    M[0] = alpha; M[1] = beta;  M[2] = (1 - alpha) * cx - beta * cy;
    M[3] = -beta; M[4] = alpha; M[5] = beta * cx + (1 - alpha) * cy;
}

void orc_cv_warp_affine_8uc3(const uint8_t *src, int sh, int sw, uint8_t *dst, int dh, int dw, const double Min[6],
                             const uint8_t border[3]) {
    double M[6];
    memcpy(M, Min, sizeof M);
    /* cv::warpAffine without WARP_INVERSE_MAP: invertAffineTransform inlined */
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D;
    M[3] *= -D; M[4] = A22;
    double b1 = -M[0] * M[2] - M[1] * M[5];
    double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    const int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, INTER_BITS = 5, INTER_TAB_SIZE = 1 << INTER_BITS;
    const int round_delta = AB_SCALE / INTER_TAB_SIZE / 2;
    uint8_t *tmp = NULL;
    if (src == dst) {                                              /* dst.data == src.data -> src = src.clone() */
        tmp = (uint8_t *)malloc((size_t)sh * sw * 3);
        memcpy(tmp, src, (size_t)sh * sw * 3);
        src = tmp;
    }
    for (int y = 0; y < dh; ++y) {
        int X0 = cv_round((M[1] * y + M[2]) * AB_SCALE) + round_delta;
        int Y0 = cv_round((M[4] * y + M[5]) * AB_SCALE) + round_delta;
        for (int x = 0; x < dw; ++x) {
            int adelta = cv_round(M[0] * x * AB_SCALE), bdelta = cv_round(M[3] * x * AB_SCALE);
            int X = (X0 + adelta) >> (AB_BITS - INTER_BITS);
            int Y = (Y0 + bdelta) >> (AB_BITS - INTER_BITS);
            int sx = X >> INTER_BITS, sy = Y >> INTER_BITS;          /* saturate_cast<short>: in range here */
            if (sx > 32767) sx = 32767;
            if (sx < -32768) sx = -32768;
            if (sy > 32767) sy = 32767;
            if (sy < -32768) sy = -32768;
            int fx = X & (INTER_TAB_SIZE - 1), fy = Y & (INTER_TAB_SIZE - 1);
            int wt[4] = {(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32};
            if (wt[0] == 32768) { wt[0] = 32767; wt[3] = 1; }       /* initInterTab2D's saturation fix-up */
            uint8_t *o = dst + ((size_t)y * dw + x) * 3;
            if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {  /* all four neighbours outside */
                o[0] = border[0]; o[1] = border[1]; o[2] = border[2];
                continue;
            }
            for (int c = 0; c < 3; ++c) {
                int v[4];
                for (int k = 0; k < 4; ++k) {
                    int px = sx + (k & 1), py = sy + (k >> 1);
                    v[k] = (px >= 0 && px < sw && py >= 0 && py < sh) ? src[((size_t)py * sw + px) * 3 + c] : border[c];
                }
                int acc = v[0] * wt[0] + v[1] * wt[1] + v[2] * wt[2] + v[3] * wt[3];
                o[c] = (uint8_t)((acc + (1 << 14)) >> 15);
            }
        }
    }
    free(tmp);
}

/* XItem::get_item_facing_dir, xitem.cpp:65-78: 0 right, 1 down, 2 left, 3 up */
int orc_facing_dir(double yaw) {
    const double eps = 1e-4, PI = 3.14159265358979323846;          /* M_PI */
    if (fabs(yaw) < eps) return 0;
    if (fabs(yaw - PI / 2) < eps) return 1;
    if (fabs(yaw - PI) < eps) return 2;
    return 3;
}

static int is_block_at(const orc_xworld *w, int x, int y) {
    return x >= 0 && x < w->width && y >= 0 && y < w->height && w->cube_n[y][x] > 0 &&
           w->ents[w->cube[y][x][0]].type == 1;
}

/* XMap::image_masking, xmap.cpp:273-362.  shadow: r*r flags, row-major [y][x] inside the ROI */
void orc_xw_image_masking(const orc_xworld *w, int ax0, int ay0, double yaw, int r, int *x_st_out, int *y_st_out,
                          uint8_t *shadow) {
    if (r % 2 != 1) abort();                                       /* CHECK_EQ(visible_radius_unit % 2, 1) */
    int xa = ax0 + r, ya = ay0 + r;
    int major_inc_x = 0, major_inc_y = 0, minor_inc_x = 0, minor_inc_y = 0, scan_x = 0, scan_y = 0;
    int dir = orc_facing_dir(yaw);
    if (dir == 0) { xa += r / 2; major_inc_y = 1; minor_inc_x = 1; }
    else if (dir == 3) { ya -= r / 2; major_inc_x = 1; minor_inc_y = -1; scan_y = r - 1; }
    else if (dir == 2) { xa -= r / 2; major_inc_y = 1; minor_inc_x = -1; scan_x = r - 1; }
    else { ya += r / 2; major_inc_x = 1; minor_inc_y = 1; }
    int x_st = xa - r / 2, y_st = ya - r / 2;
    uint8_t ray_starts[64];
    if (r > 64) abort();
    memset(ray_starts, 1, sizeof ray_starts);
    for (int o = -1; o <= 1; o += 2) {
        int block = 0, ray_x = ax0, ray_y = ay0;
        for (int k = 1; k <= r / 2; ++k) {
            ray_x += o * major_inc_x;
            ray_y += o * major_inc_y;
            if (block) ray_starts[r / 2 + o * k] = 0;
            if (is_block_at(w, ray_x, ray_y)) block = 1;
        }
    }
    memset(shadow, 0, (size_t)r * r);
    for (int k = 0; k < r; ++k) {
        int block = !ray_starts[k];
        int cur_x = scan_x, cur_y = scan_y;
        for (int j = 0; j < r; ++j) {
            if (block) shadow[cur_y * r + cur_x] = 1;
            int g_x = x_st - r + cur_x, g_y = y_st - r + cur_y;
            if (is_block_at(w, g_x, g_y)) block = 1;
            cur_x = (cur_x + minor_inc_x + r) % r;
            cur_y = (cur_y + minor_inc_y + r) % r;
        }
        scan_x += major_inc_x;
        scan_y += major_inc_y;
    }
    *x_st_out = x_st; *y_st_out = y_st;
}

/* XItem::get_item_image, xitem.cpp:33-63 */
void orc_xw_item_image(const orc_xworld *w, int ent, uint8_t *out /* 64*64*3 */) {
    const uint8_t *icon = w->icons64 + (size_t)w->ents[ent].icon * ITEM_SIZE * ITEM_SIZE * 3;
    const double PI = 3.14159265358979323846;
    double scale = w->e_scale[ent], offset = w->e_offset[ent];
    double M[6];
    orc_cv_get_rotation_matrix_2d(ITEM_SIZE / 2.0, ITEM_SIZE / 2.0, 90 - w->e_yaw[ent] * 180 / PI, scale, M);
    M[2] += (offset + scale / 2 - 0.5) * ITEM_SIZE;
    M[5] += (offset + scale / 2 - 0.5) * ITEM_SIZE;
    const uint8_t white[3] = {255, 255, 255};
    orc_cv_warp_affine_8uc3(icon, ITEM_SIZE, ITEM_SIZE, out, ITEM_SIZE, ITEM_SIZE, M, white);
}

/* XMap::to_image with visible_radius_unit = r > 0 and flag_illustration = false: the r*64 x r*64 view */
void orc_xw_ego_view(const orc_xworld *w, int r, uint8_t *view /* (r*64)^2 * 3, BGR interleaved */) {
    const int G = ITEM_SIZE, S = r * G;
    const orc_entity *a = &w->ents[w->agent_idx];
    double yaw = w->e_yaw[w->agent_idx];
    uint8_t *shadow = (uint8_t *)malloc((size_t)r * r);
    int x_st, y_st;
    orc_xw_image_masking(w, a->x, a->y, yaw, r, &x_st, &y_st, shadow);
    if (w->cfg.no_wall_shadow) memset(shadow, 0, (size_t)r * r);      /* xmap.cpp:170: if (FLAGS_wall_shadow) ... */
    /* world canvas (white) with the item images, padded by r cells of black, cropped to the ROI = cells
     * (x_st - r + i, y_st - r + j) of the unpadded map */
    uint8_t item[ITEM_SIZE * ITEM_SIZE * 3];
    for (int j = 0; j < r; ++j)
        for (int i = 0; i < r; ++i) {
            int gx = x_st - r + i, gy = y_st - r + j;
            int inside = gx >= 0 && gx < w->width && gy >= 0 && gy < w->height;
            int fill = inside ? 255 : 0;
            int have_item = 0;
            if (inside && !(shadow[j * r + i])) {
                /* items copied in stack order: the last one stays visible */
                if (w->cube_n[gy][gx] > 0) { orc_xw_item_image(w, w->cube[gy][gx][w->cube_n[gy][gx] - 1], item); have_item = 1; }
            }
            if (shadow[j * r + i]) fill = 0;                        /* FLAGS_wall_shadow: black.copyTo(grid) */
            for (int py = 0; py < G; ++py)
                for (int px = 0; px < G; ++px)
                    for (int c = 0; c < 3; ++c)
                        view[((size_t)(j * G + py) * S + (size_t)(i * G + px)) * 3 + c] =
                            have_item ? item[(py * G + px) * 3 + c] : (uint8_t)fill;
        }
    free(shadow);
    /* rotate the view according to the agent's yaw (xmap.cpp:196-200); default border = black */
    const double PI = 3.14159265358979323846;
    double M[6];
    orc_cv_get_rotation_matrix_2d(S / 2.0, S / 2.0, 90 + yaw * 180 / PI, 1.0, M);
    const uint8_t black[3] = {0, 0, 0};
    orc_cv_warp_affine_8uc3(view, S, S, view, S, S, M, black);
}
