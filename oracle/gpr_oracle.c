/*
 * gpr_oracle.c -- CPU ORACLE (test infrastructure, NOT product code). See gpr_oracle.h for scope and parity status.
 *
 * Written to follow the reference's NumPy code operation by operation (same comparisons, same inclusive /
 * strict inequalities, same evaluation order for floating point), NOT to be fast. Build with
 * -ffp-contract=off so that no a*b+c is fused except where fma() is written out.
 *
 * BE = gymnasium_planar_robotics/envs/basic_envs.py, PL = envs/planning/benchmark_planning_env.py,
 * GEO = utils/geometry_2D_utils.py, ROT = utils/rotations_utils.py (all under /root/reference).
 */
#include "gpr_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define MAXT 64 /* max tiles per axis handled by the oracle */

static void track(double *margin, double lhs, double rhs) {
    if (margin) {
        double d = fabs(lhs - rhs);
        if (d < *margin) *margin = d;
    }
}

/* ======================================================================================================
 * ctor-time tables
 * ==================================================================================================== */

/* numpy.linspace(start, stop, num, endpoint=True) (numpy/_core/function_base.py, published algorithm):
 *   step = (stop-start)/(num-1); y = arange(num)*step + start; y[-1] = stop.
 * Called as in get_tile_xy_pos / get_1D_tile_pos (BE:1300-1302) with tile_wl = tile_size*2. */
void gpro_tile_centres(int32_t n, double tile_half_size, double *out) {
    double tile_wl = tile_half_size * 2;
    double start = tile_wl / 2;
    double stop = (n - 1) * tile_wl + (tile_wl / 2);
    int32_t div = n - 1;
    double delta = stop - start;
    for (int32_t i = 0; i < n; ++i) {
        double y = (double)i;
        if (div > 0) {
            double step = delta / div;
            if (step == 0.0) {
                y = y / div;
                y = y * delta;
            } else {
                y = y * step;
            }
        } else {
            y = y * delta;
        }
        out[i] = y + start;
    }
    if (n > 1) out[n - 1] = stop;
}

static int lay_at(const gpro_layout *lay, int ix, int iy) {
    /* layout_tiles_wc (BE:221) is the layout padded by one zero row / column at the END; Python's negative index -1
     * therefore also lands on the zero padding. Out-of-range == 0. */
    if (ix < 0 || iy < 0 || ix >= lay->num_tiles_x || iy >= lay->num_tiles_y) return 0;
    return lay->layout[ix * lay->num_tiles_y + iy];
}

/* get_tile_indices_mask (BE:1312-1339). */
int32_t gpro_tile_indices_mask(const gpro_layout *lay, int32_t which, int32_t cap, int32_t *out_ix, int32_t *out_iy) {
    static const int8_t m3[9] = {1, 1, 1, 1, 1, 1, 1, 1, 1};
    static const int8_t mbl[4] = {1, 1, 0, 1}; /* BE:209 */
    static const int8_t mbr[4] = {1, 1, 1, 0}; /* BE:212 */
    static const int8_t mtl[4] = {0, 1, 1, 1}; /* BE:215 */
    static const int8_t mtr[4] = {1, 0, 1, 1}; /* BE:218 */
    const int8_t *mask = which == 0 ? m3 : which == 1 ? mbl : which == 2 ? mbr : which == 3 ? mtl : mtr;
    int ms = which == 0 ? 3 : 2;
    /* offsets (BE:1322-1323): odd size -> (s/2, s/2-1); even size -> (s/2-1, s/2-1) */
    int off0 = (ms % 2 == 1) ? ms / 2 : ms / 2 - 1;
    int off1 = ms / 2 - 1;
    int32_t cnt = 0;
    for (int ix = off0; ix < lay->num_tiles_x - off1 - 1; ++ix) {
        for (int iy = off0; iy < lay->num_tiles_y - off1 - 1; ++iy) {
            int ok = 1;
            /* slice [ix-off0 : ix+off1+2] has ms entries when in range (it always is for these loop bounds) */
            for (int a = 0; a < ms && ok; ++a)
                for (int b = 0; b < ms && ok; ++b)
                    if (mask[a * ms + b] != lay->layout[(ix - off0 + a) * lay->num_tiles_y + (iy - off0 + b)]) ok = 0;
            if (ok) {
                if (cnt < cap) {
                    out_ix[cnt] = ix;
                    out_iy[cnt] = iy;
                }
                ++cnt;
            }
        }
    }
    return cnt;
}

/* PL:242-245 (goal space high), PL:262-267 (min/max xy), PL:270-274 (min_goal_dist). */
void gpro_planning_constants(const gpro_layout *lay, const double *c_size, double *min_xy, double *max_xy,
                             double *min_goal_dist, double *goal_high) {
    double xc[MAXT], yc[MAXT];
    gpro_tile_centres(lay->num_tiles_x, lay->tile_hx, xc);
    gpro_tile_centres(lay->num_tiles_y, lay->tile_hy, yc);
    double mx = xc[0], my = yc[0];
    for (int i = 1; i < lay->num_tiles_x; ++i) if (xc[i] > mx) mx = xc[i];
    for (int i = 1; i < lay->num_tiles_y; ++i) if (yc[i] > my) my = yc[i];
    double hi[2] = {mx + (lay->tile_hx / 2), my + (lay->tile_hy / 2)}; /* tile_size is already a half size: quirk kept */
    for (int d = 0; d < 2; ++d) {
        double cs = lay->c_shape == 0 ? c_size[0] : c_size[d];
        double safety_margin = cs + lay->c_offset_wall + lay->c_offset;
        min_xy[d] = 0.0 + safety_margin;
        max_xy[d] = hi[d] - safety_margin;
        goal_high[d] = hi[d];
    }
    if (lay->c_shape == 0) {
        *min_goal_dist = 2 * (c_size[0] + lay->c_offset);
    } else {
        double a = c_size[0] + lay->c_offset, b = c_size[1] + lay->c_offset;
        *min_goal_dist = 2 * sqrt(a * a + b * b);
    }
}

/* ======================================================================================================
 * geometry_2D_utils
 * ==================================================================================================== */

/* det of [[ax,bx,cx],[ay,by,cy],[1,1,1]] (GEO:47-60 build exactly this matrix; numpy.linalg.det is LAPACK LU, whose
 * result equals this cofactor expansion up to rounding; the reference only uses sign(d1*d2) and |d1*d2| < 1e-7). */
static double det3(double ax, double ay, double bx, double by, double cx, double cy) {
    return ax * (by - cy) - bx * (ay - cy) + cx * (ay - by);
}

static int seg_intersect(const double *p1, const double *p2, const double *q1, const double *q2) {
    /* GEO:30-35 */
    int eq = 0;
    eq |= (fabs(p1[0] - q1[0]) < 1e-7) + (fabs(p1[1] - q1[1]) < 1e-7) == 2;
    eq |= (fabs(p1[0] - q2[0]) < 1e-7) + (fabs(p1[1] - q2[1]) < 1e-7) == 2;
    eq |= (fabs(p2[0] - q1[0]) < 1e-7) + (fabs(p2[1] - q1[1]) < 1e-7) == 2;
    eq |= (fabs(p2[0] - q2[0]) < 1e-7) + (fabs(p2[1] - q2[1]) < 1e-7) == 2;
    /* GEO:37-45 */
    int minmax = 0;
    for (int d = 0; d < 2; ++d) {
        double min_p = fmin(p1[d], p2[d]), max_p = fmax(p1[d], p2[d]);
        double min_q = fmin(q1[d], q2[d]), max_q = fmax(q1[d], q2[d]);
        int mask_pq = max_p < min_q, mask_qp = max_q < min_p;
        minmax += mask_pq * (1 - (fabs(max_p - min_q) < 1e-7)) + mask_qp * (1 - (fabs(max_q - min_p) < 1e-7));
    }
    /* GEO:47-64 */
    double d_ppq1 = det3(p1[0], p1[1], p2[0], p2[1], q1[0], q1[1]);
    double d_ppq2 = det3(p1[0], p1[1], p2[0], p2[1], q2[0], q2[1]);
    double d_qqp1 = det3(q1[0], q1[1], q2[0], q2[1], p1[0], p1[1]);
    double d_qqp2 = det3(q1[0], q1[1], q2[0], q2[1], p2[0], p2[1]);
    double pp = d_ppq1 * d_ppq2, qq = d_qqp1 * d_qqp2;
    /* np.sign(x) <= 0  <=>  x <= 0 (NaN excluded) */
    int orient = ((pp <= 0) || (fabs(pp) < 1e-7)) && ((qq <= 0) || (fabs(qq) < 1e-7));
    /* GEO:66-68: order matters */
    int res = 0;
    if (orient) res = 1;
    if (minmax >= 1) res = 0;
    if (eq) res = 1;
    return res;
}

void gpro_check_line_segments_intersect(int64_t n, const double *p1, const double *p2, const double *q1, const double *q2,
                                        uint8_t *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = (uint8_t)seg_intersect(p1 + 2 * i, p2 + 2 * i, q1 + 2 * i, q2 + 2 * i);
}

/* GEO:72-104 with ROT:414-461 (unit_vector: float32!) and ROT:248-274 (quat2mat). vx/vy: 4 vertices in the order
 * (-,-), (-,+), (+,+), (+,-) (GEO:91-97). */
static void rect_vertices(const double *qpos, double sx, double sy, double *vx, double *vy) {
    /* unit_vector(quats, axis=1): data = np.array(data, dtype=np.float32); length = sqrt(sum(data*data)); data /= length */
    float qf[4];
    for (int i = 0; i < 4; ++i) qf[i] = (float)qpos[3 + i];
    float len = ((qf[0] * qf[0] + qf[1] * qf[1]) + qf[2] * qf[2]) + qf[3] * qf[3];
    len = sqrtf(len);
    for (int i = 0; i < 4; ++i) qf[i] = qf[i] / len;
    /* quat2mat in float64 */
    double w = qf[0], x = qf[1], y = qf[2], z = qf[3];
    double Nq = ((w * w + x * x) + y * y) + z * z;
    double s = 2.0 / Nq;
    double X = x * s, Y = y * s, Z = z * s;
    double wZ = w * Z;
    double xX = x * X, xY = x * Y;
    double yY = y * Y, zZ = z * Z;
    double R00 = 1.0 - (yY + zZ), R01 = xY - wZ;
    double R10 = xY + wZ, R11 = 1.0 - (xX + zZ);
    if (!(Nq > DBL_EPSILON)) { /* ROT:274, _FLOAT_EPS = finfo(float64).eps */
        R00 = 1.0; R01 = 0.0; R10 = 0.0; R11 = 1.0;
    }
    const double lx[4] = {-sx, -sx, sx, sx};
    const double ly[4] = {-sy, sy, sy, -sy};
    for (int j = 0; j < 4; ++j) {
        /* rot_mats @ vertices_l (GEO:102). vertices_l is a swapaxes view, so for more than one rectangle NumPy's matmul
         * takes its own (non-BLAS) loop: out = 0; out += R[i,k]*v[k] for k = 0,1,2, unfused. (For a batch of exactly ONE
         * rectangle the view is contiguous and the product goes to BLAS, whose FMA chain can differ in the last bit: the
         * reference's vertices depend on batch size at the 1-ulp level. The batched form is the canonical one here;
         * verified bit-for-bit against the 6000-rectangle fixture.) The z term is R[i,2]*0 = 0. */
        double rx = R00 * lx[j] + R01 * ly[j];
        double ry = R10 * lx[j] + R11 * ly[j];
        vx[j] = rx + qpos[0];
        vy[j] = ry + qpos[1];
    }
}

void gpro_get_2D_rect_vertices(int64_t n, const double *qpos, const double *size, double *out) {
    for (int64_t i = 0; i < n; ++i)
        rect_vertices(qpos + 7 * i, size[2 * i], size[2 * i + 1], out + 8 * i, out + 8 * i + 4);
}

/* GEO:107-138: 4 x 4 edge pairs, edge k = vertex k -> vertex (k+1)%4 (np.roll(shift=-1)). */
static int rects_intersect(const double *qpos1, const double *qpos2, double s1x, double s1y, double s2x, double s2y) {
    double ax[4], ay[4], bx[4], by[4];
    rect_vertices(qpos1, s1x, s1y, ax, ay);
    rect_vertices(qpos2, s2x, s2y, bx, by);
    int cnt = 0;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) {
            double p1[2] = {ax[i], ay[i]}, p2[2] = {ax[(i + 1) & 3], ay[(i + 1) & 3]};
            double q1[2] = {bx[j], by[j]}, q2[2] = {bx[(j + 1) & 3], by[(j + 1) & 3]};
            cnt += seg_intersect(p1, p2, q1, q2);
        }
    }
    return cnt >= 1;
}

void gpro_check_rectangles_intersect(int64_t n, const double *qpos_r1, const double *qpos_r2, const double *size_r1,
                                     const double *size_r2, uint8_t *out) {
    for (int64_t i = 0; i < n; ++i)
        out[i] = (uint8_t)rects_intersect(qpos_r1 + 7 * i, qpos_r2 + 7 * i, size_r1[2 * i], size_r1[2 * i + 1], size_r2[2 * i],
                                          size_r2[2 * i + 1]);
}

/* ======================================================================================================
 * qpos_is_valid (BE:459-788)
 * ==================================================================================================== */

static int is_3x3_centre(const gpro_layout *lay, int ix, int iy) {
    /* membership in idx_x/y_tiles_3x3 (BE:206-207 via BE:1312-1339) */
    if (ix < 1 || iy < 1 || ix > lay->num_tiles_x - 2 || iy > lay->num_tiles_y - 2) return 0;
    for (int a = -1; a <= 1; ++a)
        for (int b = -1; b <= 1; ++b)
            if (lay->layout[(ix + a) * lay->num_tiles_y + (iy + b)] != 1) return 0;
    return 1;
}

/* does the 2x2 block with top-left (bx,by) equal the mask (m00 m01; m10 m11)?  (rows = x index, cols = y index) */
static int block_is(const gpro_layout *lay, int bx, int by, int m00, int m01, int m10, int m11) {
    if (bx < 0 || by < 0 || bx > lay->num_tiles_x - 2 || by > lay->num_tiles_y - 2) return 0;
    const int8_t *L = lay->layout;
    int ty = lay->num_tiles_y;
    return L[bx * ty + by] == m00 && L[bx * ty + by + 1] == m01 && L[(bx + 1) * ty + by] == m10 && L[(bx + 1) * ty + by + 1] == m11;
}

/* one qpos; cs0/cs1 already include offset_wall (+ offset). Returns validity; *above = number of tiles it is over. */
static int qpos_valid_one(const gpro_layout *lay, const double *xc, const double *yc, const double *qpos, double cs0, double cs1,
                          int *above_cnt, double *margin) {
    const int circle = lay->c_shape == 0;
    const int Tx = lay->num_tiles_x, Ty = lay->num_tiles_y;
    const double hx = lay->tile_hx, hy = lay->tile_hy;
    double vx[4], vy[4];
    if (!circle) rect_vertices(qpos, cs0, cs1, vx, vy); /* BE:496-497 */
    int any = 0, valid3 = 0, all_k = 1;
    for (int ix = 0; ix < Tx; ++ix) {
        for (int iy = 0; iy < Ty; ++iy) {
            double cx = xc[ix], cy = yc[iy];
            /* BE:508-513, inclusive, independent of whether the tile exists */
            int above = (cx - hx <= qpos[0]) && (qpos[0] <= cx + hx) && (cy - hy <= qpos[1]) && (qpos[1] <= cy + hy);
            /* (not tracked in `margin`: which of two adjacent tiles a centre is "above" does not change the verdict for a
             * shape of positive size -- the strict edge compares below decide, and they are tracked) */
            if (!above) continue;
            any += 1;
            if (is_3x3_centre(lay, ix, iy)) valid3 = 1; /* BE:530-538 */
            double min_x = cx - hx, max_x = cx + hx, min_y = cy - hy, max_y = cy + hy; /* BE:522-526 */
            int L = lay_at(lay, ix, iy);
            int rep = circle ? 1 : 4;
            int sum_valid = 0;
            int lmin_x = ix > 0, lmin_y = iy > 0, smax_x = ix < Tx - 1, smax_y = iy < Ty - 1; /* BE:575-578 */
            for (int r = 0; r < rep; ++r) {
                int mnx, mxx, mny, mxy;
                if (circle) { /* BE:545-558 */
                    double xm = qpos[0] - cs0, xp = qpos[0] + cs0, ym = qpos[1] - cs0, yp = qpos[1] + cs0;
                    mnx = L * (min_x < xm); mxx = L * (xp < max_x); mny = L * (min_y < ym); mxy = L * (yp < max_y);
                    track(margin, min_x, xm); track(margin, xp, max_x); track(margin, min_y, ym); track(margin, yp, max_y);
                } else { /* BE:560-572 */
                    mnx = L * (min_x < vx[r]); mxx = L * (vx[r] < max_x); mny = L * (min_y < vy[r]); mxy = L * (vy[r] < max_y);
                    track(margin, min_x, vx[r]); track(margin, vx[r], max_x); track(margin, min_y, vy[r]); track(margin, vy[r], max_y);
                }
                int mv = mnx * mxx * mny * mxy; /* BE:580 */
                /* BE:582-607 */
                int upd_mnx = (1 - mnx) * (lmin_x * lay_at(lay, ix, iy) * lay_at(lay, ix - 1, iy));
                mv += upd_mnx * mny * mxy;
                int upd_mnx_mny = (1 - mny) * (lmin_x * lmin_y * lay_at(lay, ix, iy) * lay_at(lay, ix, iy - 1) * lay_at(lay, ix - 1, iy - 1));
                mv += upd_mnx * upd_mnx_mny;
                int upd_mnx_mxy = (1 - mxy) * (lmin_x * smax_y * lay_at(lay, ix, iy) * lay_at(lay, ix, iy + 1) * lay_at(lay, ix - 1, iy + 1));
                mv += upd_mnx * upd_mnx_mxy;
                /* BE:609-634 */
                int upd_mxx = (1 - mxx) * (smax_x * lay_at(lay, ix, iy) * lay_at(lay, ix + 1, iy));
                mv += upd_mxx * mny * mxy;
                int upd_mxx_mny = (1 - mny) * (smax_x * lmin_y * lay_at(lay, ix, iy) * lay_at(lay, ix, iy - 1) * lay_at(lay, ix + 1, iy - 1));
                mv += upd_mxx * upd_mxx_mny;
                int upd_mxx_mxy = (1 - mxy) * (smax_x * smax_y * lay_at(lay, ix, iy) * lay_at(lay, ix, iy + 1) * lay_at(lay, ix + 1, iy + 1));
                mv += upd_mxx * upd_mxx_mxy;
                /* BE:636-648 */
                int upd_mny = (1 - mny) * (lmin_y * lay_at(lay, ix, iy) * lay_at(lay, ix, iy - 1));
                mv += upd_mny * mnx * mxx;
                int upd_mxy = (1 - mxy) * (smax_y * lay_at(lay, ix, iy) * lay_at(lay, ix, iy + 1));
                mv += upd_mxy * mnx * mxx;
                sum_valid += mv;
            }
            int valid_k;
            if (circle) {
                valid_k = (sum_valid == 1); /* BE:653 + BE:785 (== 1) */
            } else {
                valid_k = (sum_valid == 4); /* BE:655 */
                /* BE:657-783: L-shaped 2x2 neighbourhoods, rectangle test against the missing (diagonal) tile */
                struct { int ok_range; int bx, by, m00, m01, m10, m11, mx, my; } c4[4] = {
                    /* bottom left  BE:658-687: entry == (bl_x, bl_y+1); missing (bl_x+1, bl_y) */
                    {smax_x && lmin_y, ix, iy - 1, 1, 1, 0, 1, ix + 1, iy - 1},
                    /* bottom right BE:690-719: entry == (br_x, br_y); missing (br_x+1, br_y+1) */
                    {smax_x && smax_y, ix, iy, 1, 1, 1, 0, ix + 1, iy + 1},
                    /* top left     BE:722-751: entry == (tl_x+1, tl_y+1); missing (tl_x, tl_y) */
                    {lmin_x && lmin_y, ix - 1, iy - 1, 0, 1, 1, 1, ix - 1, iy - 1},
                    /* top right    BE:754-783: entry == (tr_x+1, tr_y); missing (tr_x, tr_y+1) */
                    {lmin_x && smax_y, ix - 1, iy, 1, 0, 1, 1, ix - 1, iy + 1},
                };
                for (int c = 0; c < 4; ++c) {
                    if (!(valid_k && c4[c].ok_range)) continue;
                    if (!block_is(lay, c4[c].bx, c4[c].by, c4[c].m00, c4[c].m01, c4[c].m10, c4[c].m11)) continue;
                    double qm[7] = {xc[c4[c].mx], yc[c4[c].my], 0, 1, 0, 0, 0};
                    int inter = rects_intersect(qpos, qm, cs0, cs1, hx, hy);
                    valid_k = (1 - inter) * valid_k;
                }
            }
            if (!valid_k) all_k = 0;
        }
    }
    *above_cnt = any;
    /* BE:538 (3x3 positions stay valid) and BE:785-786 */
    return valid3 || (any > 0 && all_k);
}

int64_t gpro_qpos_is_valid(const gpro_layout *lay, int64_t n, const double *qpos, const double *c_size, int32_t add_safety_offset,
                           int32_t *out, double *margin) {
    double xc[MAXT], yc[MAXT];
    gpro_tile_centres(lay->num_tiles_x, lay->tile_hx, xc);
    gpro_tile_centres(lay->num_tiles_y, lay->tile_hy, yc);
    int64_t n_off = 0;
    for (int64_t i = 0; i < n; ++i) {
        /* BE:487: c_size + offset_wall + int(add)*offset, left to right */
        double cs0 = c_size[2 * i] + lay->c_offset_wall + (double)(add_safety_offset != 0) * lay->c_offset;
        double cs1 = c_size[2 * i + 1] + lay->c_offset_wall + (double)(add_safety_offset != 0) * lay->c_offset;
        int above = 0;
        double m = INFINITY;
        out[i] = qpos_valid_one(lay, xc, yc, qpos + 7 * i, cs0, cs1, &above, margin ? &m : NULL);
        if (margin) margin[i] = m;
        if (above == 0) ++n_off;
    }
    return n_off;
}

/* ======================================================================================================
 * check_mover_collision (BE:355-424)
 * ==================================================================================================== */
int32_t gpro_check_mover_collision(const gpro_layout *lay, int32_t m, const double *qpos, const double *c_size,
                                   int32_t add_safety_offset, double *margin) {
    double cs[GPRO_MAX_MOVERS * 2];
    for (int i = 0; i < m; ++i) { /* BE:390 */
        cs[2 * i] = c_size[2 * i] + lay->c_offset * (double)(add_safety_offset != 0);
        cs[2 * i + 1] = c_size[2 * i + 1] + lay->c_offset * (double)(add_safety_offset != 0);
    }
    int coll = 0;
    if (lay->c_shape == 0) {
        /* BE:409: norm has shape (K,), (c_i + c_j) has shape (K,1): NumPy broadcasts the comparison to (K,K), i.e.
         * collision = any_{a,b} norm[b] <= csum[a]. Identical to the per-pair test when all radii are equal. */
        for (int ia = 0; ia < m - 1; ++ia) for (int ja = ia + 1; ja < m; ++ja) {
            double csum = cs[2 * ia] + cs[2 * ja];
            for (int ib = 0; ib < m - 1; ++ib) for (int jb = ib + 1; jb < m; ++jb) {
                double dx = qpos[7 * ib] - qpos[7 * jb], dy = qpos[7 * ib + 1] - qpos[7 * jb + 1];
                double norm = sqrt(dx * dx + dy * dy);
                track(margin, norm, csum);
                if (norm <= csum) coll = 1;
            }
        }
    } else {
        for (int i = 0; i < m - 1; ++i) for (int j = i + 1; j < m; ++j) {
            double dx = qpos[7 * i] - qpos[7 * j], dy = qpos[7 * i + 1] - qpos[7 * j + 1];
            double dist = sqrt(dx * dx + dy * dy);
            double ms = fmax(fmax(cs[2 * i], cs[2 * i + 1]), fmax(cs[2 * j], cs[2 * j + 1])); /* BE:412 */
            double lim = 2 * (fabs(ms) + fabs(ms));                                            /* BE:413-414, ord=1 */
            track(margin, dist, lim);
            if (dist <= lim) {
                if (rects_intersect(qpos + 7 * i, qpos + 7 * j, cs[2 * i], cs[2 * i + 1], cs[2 * j], cs[2 * j + 1])) coll = 1;
            }
        }
    }
    return coll;
}

/* ======================================================================================================
 * dynamics helpers
 * ==================================================================================================== */

/* PL:610-645, one row. Returns 1 if the clamp branch was taken. */
static int clamp_one(const double *cur, double max_value, const double *deriv, double dt, double *vals, double *ders, double *margin) {
    double t0 = dt * deriv[0] + cur[0], t1 = dt * deriv[1] + cur[1]; /* PL:631 */
    double norm = sqrt(t0 * t0 + t1 * t1);                           /* PL:633 */
    (void)margin; /* not a tracked decision: the clamp is continuous across norm == max_value */
    if (norm >= max_value) { /* PL:634 inclusive */
        vals[0] = max_value * (t0 / norm); /* PL:640-642 */
        vals[1] = max_value * (t1 / norm);
        ders[0] = (vals[0] - cur[0]) / dt; /* PL:643 */
        ders[1] = (vals[1] - cur[1]) / dt;
        return 1;
    }
    vals[0] = t0; vals[1] = t1; ders[0] = deriv[0]; ders[1] = deriv[1];
    return 0;
}

void gpro_ensure_max_dyn_val(int64_t n, const double *cur, const double *max_value, const double *deriv, double dt, double *out_vals,
                             double *out_derivs) {
    for (int64_t i = 0; i < n; ++i) clamp_one(cur + 2 * i, max_value[i], deriv + 2 * i, dt, out_vals + 2 * i, out_derivs + 2 * i, NULL);
}

static void reward_one(int32_t M, const double *ag, const double *dg, int mc, int wc, double thr, double rs, double *reward,
                       uint8_t *terminated, uint8_t *is_success, double *margin) {
    int n_reached = 0;
    for (int i = 0; i < M; ++i) { /* PL:647-664, PL:524 */
        double dx = ag[2 * i] - dg[2 * i], dy = ag[2 * i + 1] - dg[2 * i + 1];
        double dist = sqrt(dx * dx + dy * dy);
        track(margin, dist, thr);
        n_reached += dist <= thr;
    }
    int coll = mc || wc;
    double r = -rs * (double)coll;                       /* PL:529 */
    r += -1.0 * (double)(M - n_reached) * (double)(!coll); /* PL:530 */
    if (n_reached == M && !coll) r = rs;                  /* PL:531 */
    *reward = r;
    *terminated = (r == rs) || (r == -rs);                /* PL:478 */
    *is_success = (n_reached == M) && !mc && !wc;         /* PL:597 */
}

void gpro_compute_reward(int64_t batch, int32_t num_movers, const double *achieved, const double *desired, const uint8_t *mover_collision,
                         const uint8_t *wall_collision, double threshold_pos, double reward_success, double *reward, uint8_t *terminated,
                         uint8_t *is_success) {
    for (int64_t b = 0; b < batch; ++b)
        reward_one(num_movers, achieved + 2 * num_movers * b, desired + 2 * num_movers * b, mover_collision[b], wall_collision[b],
                   threshold_pos, reward_success, reward + b, terminated + b, is_success + b, NULL);
}

/* ======================================================================================================
 * Engine RNG spec "gpr-rng-v2" (engine-defined; the reference draws from NumPy PCG64 streams, which a GPU engine
 * can neither reproduce nor needs to: sigma = 1e-5 is 10x below the 1e-4 parity tolerance). Counter-based, 32-bit
 * ops only, order-independent, so the HIP kernels (one LANE per mover) and this oracle (one LOOP per env) produce the
 * same integers:
 *   key  h(seed, tick, kind, env, mover) = lowbias32 absorb chain
 *   draw r(h, k) = lowbias32(h + (k+1)*0x9E3779B9)                 (murmur-style finaliser over a Weyl sequence)
 *   normal pair q = Box-Muller on the upper 13 (radius) / lower 16 (angle) bits of draw q; uniform k = upper 24 bits of draw k
 * ==================================================================================================== */
static uint32_t mix32(uint32_t x) { /* "lowbias32" integer hash (C. Wellons) */
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

uint32_t gpro_rng_key(uint64_t seed, uint64_t tick, uint32_t kind, uint64_t env_id, uint32_t mover) {
    uint32_t h = 0x9E3779B9U;
    h = mix32(h ^ (uint32_t)seed);
    h = mix32(h ^ (uint32_t)(seed >> 32));
    h = mix32(h ^ (uint32_t)tick);
    h = mix32(h ^ (uint32_t)(tick >> 32));
    h = mix32(h ^ kind);
    h = mix32(h ^ (uint32_t)env_id);
    h = mix32(h ^ (uint32_t)(env_id >> 32));
    h = mix32(h ^ mover);
    return h;
}

uint32_t gpro_rng_draw(uint32_t key, uint32_t k) { return mix32(key + (k + 1u) * 0x9E3779B9U); }

double gpro_rng_uniform(uint32_t key, uint32_t k) { return (double)(gpro_rng_draw(key, k) >> 8) * (1.0 / 16777216.0); } /* [0,1) */

/* gpr-rng-v2: ONE 32-bit draw per pair; the Box-Muller radius from its upper 13 bits (u1 = (i + 1) / 8192: radius <= sqrt(26 ln 2) =
 * 4.245 sigma), the angle 2 pi k / 65536 from its lower 16 bits, looked up in a table over the FIRST OCTANT (j = 0 .. 8192) and unfolded
 * by symmetry -- the engine's double builds read the same two tables (192 KB, filled by the same libm calls on the same host), which is
 * what makes the noise bit-identical; its float build computes the same quantities with the GPU's log2 / sqrt / sin / cos. */
static double g_rad_tab[8192], g_trig_tab[8193][2];
static pthread_once_t g_tab_once = PTHREAD_ONCE_INIT;
static void fill_normal_tables(void) {
    for (uint32_t i = 0; i < 8192u; ++i) {
        const double u1 = (double)(i + 1u) * (1.0 / 8192.0); /* (0,1] */
        g_rad_tab[i] = sqrt(-2.0 * log(u1));
    }
    for (uint32_t j = 0; j <= 8192u; ++j) {
        const double u2 = (double)j * (1.0 / 65536.0);
        const double ang = 6.283185307179586476925286766559 * u2;
        g_trig_tab[j][0] = cos(ang);
        g_trig_tab[j][1] = sin(ang);
    }
}
void gpro_rng_normal_pair(uint32_t key, uint32_t q, double *n0, double *n1) {
    pthread_once(&g_tab_once, fill_normal_tables);
    const uint32_t r = gpro_rng_draw(key, q);
    const double rad = g_rad_tab[r >> 19];
    const uint32_t k = r & 0xFFFFu, oct = k >> 13, j = k & 8191u;
    const double *t = g_trig_tab[(oct & 1u) ? 8192u - j : j]; /* the octant's own angle, or its mirror image */
    const int swap = (int)(((oct + 1u) >> 1) & 1u);            /* octants 1, 2, 5, 6: cos and sin change places */
    double c = swap ? t[1] : t[0], s = swap ? t[0] : t[1];
    if (((oct + 2u) >> 2) & 1u) c = -c;                        /* octants 2 .. 5: cos < 0 */
    if (oct >> 2) s = -s;                                      /* octants 4 .. 7: sin < 0 */
    *n0 = rad * c;
    *n1 = rad * s;
}

enum { KIND_STEP_NOISE = 1, KIND_RESET_START = 2, KIND_RESET_GOAL = 3, KIND_RESET_NOISE = 4, KIND_GETTER_NOISE = 5 };

/* get_mover_qpos / qvel / qacc and the _arr forms (BE:801-855, 1244-1290) of every env, in the batched engine's terms (include/gpr.h,
 * gpr_get_mover_state): BE:821-824 reports z as the body's z minus the mover's half height = the gap above the tile, constant here
 * (z_report); BE:828 / 841 / 855 add N(0, std_noise[which] * int(add_noise)) to EVERY component. */
void gpro_get_mover_state(const gpro_env_cfg *cfg, int64_t num_envs, uint64_t seed, uint64_t tick, int64_t env_id_offset, int32_t which,
                          int32_t n_sel, const int32_t *mover_idx, double z_report, int32_t add_noise, const double *pos, const double *vel,
                          const double *acc, const double *mover_rot, double *out) {
    const int M = cfg->num_movers, nc = which == 0 ? 7 : 6;
    const double sigma = cfg->std_noise[which];
    for (int64_t e = 0; e < num_envs; ++e)
        for (int j = 0; j < n_sel; ++j) {
            const int m = mover_idx[j];
            const double *src = (which == 0 ? pos : which == 1 ? vel : acc) + (e * M + m) * 2;
            double v[8] = {src[0], src[1], 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (which == 0) {
                v[2] = z_report; v[3] = 1.0;
                if (mover_rot) { v[3] = cos(0.5 * mover_rot[(e * M + m) * 2]); v[6] = sin(0.5 * mover_rot[(e * M + m) * 2]); }
            } else if (which == 1 && mover_rot) {
                v[5] = mover_rot[(e * M + m) * 2 + 1];
            }
            if (add_noise) {
                const uint32_t key = gpro_rng_key(seed, tick, KIND_GETTER_NOISE, (uint64_t)(env_id_offset + e), (uint32_t)m);
                for (uint32_t k = 0; k < 4u; ++k) {
                    double n0, n1;
                    gpro_rng_normal_pair(key, k, &n0, &n1);
                    v[2 * k] = v[2 * k] + sigma * n0;
                    v[2 * k + 1] = v[2 * k + 1] + sigma * n1;
                }
            }
            for (int k = 0; k < nc; ++k) out[(e * n_sel + j) * nc + k] = cfg->storage_f32 ? (double)(float)v[k] : v[k];
        }
}

/* ======================================================================================================
 * Vectorised BenchmarkPlanningEnv (BE:1770-1950 template + PL callbacks), one env at a time.
 * ==================================================================================================== */
typedef struct {
    const gpro_env_cfg *cfg;
    double xc[MAXT], yc[MAXT];
    int noise;
} env_ctx;

static double q32(const gpro_env_cfg *cfg, double x) { return cfg->storage_f32 ? (double)(float)x : x; }

/* noisy copy of M xy pairs: normal pair number q of each mover's stream (x <- n0, y <- n1) */
static void noisy(const env_ctx *c, const uint32_t *keys, uint32_t q, int M, const double *xy, double sigma, double *out) {
    for (int i = 0; i < M; ++i) {
        double n0 = 0.0, n1 = 0.0;
        if (c->noise) gpro_rng_normal_pair(keys[i], q, &n0, &n1);
        out[2 * i] = xy[2 * i] + sigma * n0;
        out[2 * i + 1] = xy[2 * i + 1] + sigma * n1;
    }
}

/* A noisy qpos read of every mover (BE:828: qpos + N(0, sigma) on all seven components). x,y take normal pair q; for the
 * box shape the quaternion (1,0,0,0) takes pairs q+1 (w,x) and q+2 (y,z) -- the circle shape never looks at it, and
 * nothing looks at z. The engine's draw numbering: reads_per_check() pairs per collision check. */
static int reads_per_check(const env_ctx *c) { return c->cfg->lay.c_shape == 0 ? 1 : 3; }

static void noisy_qpos(const env_ctx *c, const uint32_t *keys, uint32_t q, int M, const double *xy, double sigma, int with_noise,
                       double *qpos) {
    for (int i = 0; i < M; ++i) {
        double *o = qpos + 7 * i;
        o[0] = xy[2 * i]; o[1] = xy[2 * i + 1]; o[2] = 0.0; o[3] = 1.0; o[4] = o[5] = o[6] = 0.0;
        if (with_noise && c->noise) {
            double n0, n1;
            gpro_rng_normal_pair(keys[i], q, &n0, &n1);
            o[0] = xy[2 * i] + sigma * n0; o[1] = xy[2 * i + 1] + sigma * n1;
            if (c->cfg->lay.c_shape != 0) {
                gpro_rng_normal_pair(keys[i], q + 1u, &n0, &n1);
                o[3] = 1.0 + sigma * n0; o[4] = 0.0 + sigma * n1;
                gpro_rng_normal_pair(keys[i], q + 2u, &n0, &n1);
                o[5] = 0.0 + sigma * n0; o[6] = 0.0 + sigma * n1;
            }
        } else {
            o[0] = xy[2 * i] + sigma * 0.0; o[1] = xy[2 * i + 1] + sigma * 0.0;
        }
    }
}

/* check_wall_collision(...).any() (BE:426-457) */
static int wall_any(const env_ctx *c, const double *qpos, int add_safety, double *margin) {
    const gpro_env_cfg *cfg = c->cfg;
    int M = cfg->num_movers, wall = 0;
    for (int i = 0; i < M; ++i) {
        double cs0 = cfg->c_size[2 * i] + cfg->lay.c_offset_wall + (double)(add_safety != 0) * cfg->lay.c_offset;
        double cs1 = cfg->c_size[2 * i + 1] + cfg->lay.c_offset_wall + (double)(add_safety != 0) * cfg->lay.c_offset;
        int above = 0;
        int valid = qpos_valid_one(&cfg->lay, c->xc, c->yc, qpos + 7 * i, cs0, cs1, &above, margin);
        if (!valid) wall = 1; /* a mover above no tile is a wall collision in the engine (reference: AssertionError, BE:514) */
    }
    return wall;
}

static int pair_any(const env_ctx *c, const double *qpos, int add_safety, double *margin) {
    return gpro_check_mover_collision(&c->cfg->lay, c->cfg->num_movers, qpos, c->cfg->c_size, add_safety, margin);
}

/* _get_obs (PL:536-573) + _get_info (PL:575-602) + reward/terminated (PL:459-534) */
static void observe(const env_ctx *c, const uint32_t *keys, uint32_t q0, const double *pos, const double *vel, const double *acc, const double *goal,
                    int mc, int wc, int64_t e, gpro_step_out *out, double *margin) {
    const gpro_env_cfg *cfg = c->cfg;
    int M = cfg->num_movers;
    int od = cfg->learn_jerk ? 4 * M : 2 * M;
    double npos[GPRO_MAX_MOVERS * 2], nvel[GPRO_MAX_MOVERS * 2];
    noisy(c, keys, q0, M, pos, cfg->std_noise[0], npos);     /* PL:554 */
    noisy(c, keys, q0 + 1, M, vel, cfg->std_noise[1], nvel); /* PL:555 */
    for (int k = 0; k < 2 * M; ++k) {
        out->observation[od * e + k] = q32(cfg, nvel[k]);
        if (cfg->learn_jerk) out->observation[od * e + 2 * M + k] = q32(cfg, acc[k]); /* PL:558-560: no noise, after all velocities */
        out->achieved_goal[2 * M * e + k] = q32(cfg, npos[k]);
        out->desired_goal[2 * M * e + k] = q32(cfg, goal[k]);
    }
    double r; uint8_t term, succ;
    reward_one(M, &out->achieved_goal[2 * M * e], &out->desired_goal[2 * M * e], mc, wc, cfg->threshold_pos, cfg->reward_success, &r,
               &term, &succ, margin);
    out->reward[e] = r;
    out->terminated[e] = term;
    out->is_success[e] = succ;
    out->mover_collision[e] = (uint8_t)mc;
    out->wall_collision[e] = (uint8_t)wc;
}

/* _reset_callback (PL:355-418) + reset() (BE:1770-1833) for one env; leaves the reset observation in out. */
/* A scripted candidate stream (gpro_planning_reset_candidates): what np_random.uniform hands out, in order -- the start loop consumes
 * sets from it until it accepts, the goal loop goes on from there (the two loops share ONE generator, PL:377 / PL:402). NULL: the
 * engine's own counter-based draws. */
typedef struct { const double *cand; int64_t stride; int32_t n; int32_t *n_used; } cand_stream;

static void reset_one_from(const env_ctx *c, uint64_t seed, uint64_t tick, int64_t env_id, int64_t e, double *pos, double *vel, double *acc,
                           double *goal, int32_t *elapsed, gpro_step_out *out, double *margin, const cand_stream *cs);

static void reset_one(const env_ctx *c, uint64_t seed, uint64_t tick, int64_t env_id, int64_t e, double *pos, double *vel, double *acc,
                      double *goal, int32_t *elapsed, gpro_step_out *out, double *margin) {
    reset_one_from(c, seed, tick, env_id, e, pos, vel, acc, goal, elapsed, out, margin, NULL);
}

static void reset_one_from(const env_ctx *c, uint64_t seed, uint64_t tick, int64_t env_id, int64_t e, double *pos, double *vel, double *acc,
                           double *goal, int32_t *elapsed, gpro_step_out *out, double *margin, const cand_stream *cs) {
    const gpro_env_cfg *cfg = c->cfg;
    int M = cfg->num_movers;
    uint32_t ks[GPRO_MAX_MOVERS], kg[GPRO_MAX_MOVERS], kn[GPRO_MAX_MOVERS];
    for (int i = 0; i < M; ++i) {
        ks[i] = gpro_rng_key(seed, tick, KIND_RESET_START, (uint64_t)env_id, (uint32_t)i);
        kg[i] = gpro_rng_key(seed, tick, KIND_RESET_GOAL, (uint64_t)env_id, (uint32_t)i);
        kn[i] = gpro_rng_key(seed, tick, KIND_RESET_NOISE, (uint64_t)env_id, (uint32_t)i);
    }
    double xy[GPRO_MAX_MOVERS * 2];
    int used = 0; /* sets taken from the scripted stream so far */
    const int trials0 = cs ? (cs->n < cfg->max_reset_trials ? cs->n : cfg->max_reset_trials) : cfg->max_reset_trials;
    for (int trial = 0; trial < trials0; ++trial) { /* PL:369-385 */
        for (int k = 0; k < 2 * M; ++k) {
            int d = k & 1;
            if (cs) { xy[k] = q32(cfg, cs->cand[(int64_t)trial * cs->stride + k]); continue; }
            double u = gpro_rng_uniform(ks[k >> 1], 2u * (uint32_t)trial + (uint32_t)d);
            xy[k] = q32(cfg, cfg->min_xy[d] + (cfg->max_xy[d] - cfg->min_xy[d]) * u); /* PL:377 */
        }
        used = trial + 1;
        double qp[GPRO_MAX_MOVERS * 7];
        noisy_qpos(c, ks, 0, M, xy, 0.0, 0, qp);
        int wall = wall_any(c, qp, 1, margin);  /* PL:379 add_safety_offset=True */
        int coll = pair_any(c, qp, 1, margin);  /* PL:381-383 */
        if (!coll && !wall) break;
    }
    memcpy(pos, xy, sizeof(double) * 2 * M);
    const int used0 = used;
    const int trials1 = cs ? (cs->n - used0 < cfg->max_reset_trials ? cs->n - used0 : cfg->max_reset_trials) : cfg->max_reset_trials;
    for (int trial = 0; trial < trials1; ++trial) { /* PL:395-413 */
        for (int k = 0; k < 2 * M; ++k) {
            int d = k & 1;
            if (cs) { xy[k] = q32(cfg, cs->cand[(int64_t)(used0 + trial) * cs->stride + k]); continue; }
            double u = gpro_rng_uniform(kg[k >> 1], 2u * (uint32_t)trial + (uint32_t)d);
            xy[k] = q32(cfg, cfg->min_xy[d] + (cfg->max_xy[d] - cfg->min_xy[d]) * u); /* PL:402 */
        }
        used = used0 + trial + 1;
        double qp[GPRO_MAX_MOVERS * 7];
        noisy_qpos(c, kg, 0, M, xy, 0.0, 0, qp);
        int wall = wall_any(c, qp, 1, margin); /* PL:403-404 */
        int ok = !wall;
        if (ok) {
            for (int i = 0; i < M && ok; ++i) for (int j = i + 1; j < M; ++j) { /* PL:407-413 */
                double dx = xy[2 * i] - xy[2 * j], dy = xy[2 * i + 1] - xy[2 * j + 1];
                double dist = sqrt(dx * dx + dy * dy);
                track(margin, dist, cfg->min_goal_dist);
                if (dist < cfg->min_goal_dist) { ok = 0; break; }
            }
        }
        if (ok) break;
    }
    memcpy(goal, xy, sizeof(double) * 2 * M);
    if (cs && cs->n_used) *cs->n_used = used;
    for (int k = 0; k < 2 * M; ++k) { vel[k] = 0.0; acc[k] = 0.0; } /* reload_model + mj_forward: v = 0, act = 0, qacc = 0 */
    *elapsed = 0;
    double qp[GPRO_MAX_MOVERS * 7];
    const uint32_t R = (uint32_t)reads_per_check(c);
    noisy_qpos(c, kn, 0, M, pos, cfg->std_noise[0], 1, qp);
    int wc = wall_any(c, qp, 1, margin); /* BE:1799-1801 add_safety_offset=True */
    noisy_qpos(c, kn, R, M, pos, cfg->std_noise[0], 1, qp);
    int mc = pair_any(c, qp, 0, margin); /* BE:1803-1805 add_safety_offset=False */
    observe(c, kn, 2u * R, pos, vel, acc, goal, mc, wc, e, out, margin);
    out->truncated[e] = 0;
}

static void ctx_init(env_ctx *c, const gpro_env_cfg *cfg) {
    c->cfg = cfg;
    gpro_tile_centres(cfg->lay.num_tiles_x, cfg->lay.tile_hx, c->xc);
    gpro_tile_centres(cfg->lay.num_tiles_y, cfg->lay.tile_hy, c->yc);
    c->noise = cfg->std_noise[0] != 0.0 || cfg->std_noise[1] != 0.0 || cfg->std_noise[2] != 0.0;
}

int32_t gpro_planning_reset(const gpro_env_cfg *cfg, int64_t num_envs, uint64_t seed, uint64_t tick, int64_t env_id_offset,
                            const uint8_t *mask, double *pos, double *vel, double *acc, double *goal, int32_t *elapsed,
                            gpro_step_out *out) {
    if (cfg->num_movers < 1 || cfg->num_movers > GPRO_MAX_MOVERS) return -1;
    env_ctx c;
    ctx_init(&c, cfg);
    int M = cfg->num_movers;
    for (int64_t e = 0; e < num_envs; ++e) {
        if (mask && !mask[e]) continue;
        double m = INFINITY;
        reset_one(&c, seed, tick, env_id_offset + e, e, pos + 2 * M * e, vel + 2 * M * e, acc + 2 * M * e, goal + 2 * M * e, elapsed + e,
                  out, out->margin ? &m : NULL);
        out->reward[e] = 0.0;
        out->terminated[e] = 0;
        if (out->margin) out->margin[e] = m;
    }
    return 0;
}

/* reset() with np_random.uniform scripted: cand [n_cand][E][2M] is the stream of candidate sets of every env, n_used [E] (nullable)
 * receives how many sets the env's two loops consumed. An exhausted stream ends a loop like max_reset_trials does (the last set stands). */
int32_t gpro_planning_reset_candidates(const gpro_env_cfg *cfg, int64_t num_envs, uint64_t seed, uint64_t tick, int64_t env_id_offset,
                                       const uint8_t *mask, int32_t n_cand, const double *cand, int32_t *n_used, double *pos, double *vel,
                                       double *acc, double *goal, int32_t *elapsed, gpro_step_out *out) {
    if (cfg->num_movers < 1 || cfg->num_movers > GPRO_MAX_MOVERS || n_cand < 2 || !cand) return -1;
    env_ctx c;
    ctx_init(&c, cfg);
    int M = cfg->num_movers;
    for (int64_t e = 0; e < num_envs; ++e) {
        if (mask && !mask[e]) continue;
        double m = INFINITY;
        cand_stream cs = {cand + 2 * M * e, 2 * (int64_t)M * num_envs, n_cand, n_used ? n_used + e : NULL};
        reset_one_from(&c, seed, tick, env_id_offset + e, e, pos + 2 * M * e, vel + 2 * M * e, acc + 2 * M * e, goal + 2 * M * e, elapsed + e,
                       out, out->margin ? &m : NULL, &cs);
        out->reward[e] = 0.0;
        out->terminated[e] = 0;
        if (out->margin) out->margin[e] = m;
    }
    return 0;
}

static void step_range(const env_ctx *c, int64_t e0, int64_t e1, uint64_t seed, uint64_t tick, int64_t env_id_offset, const double *action,
                       double *pos_a, double *vel_a, double *acc_a, double *goal_a, int32_t *elapsed_a, int32_t autoreset, gpro_step_out *out) {
    const gpro_env_cfg *cfg = c->cfg;
    const int M = cfg->num_movers;
    const int od = cfg->learn_jerk ? 4 * M : 2 * M;
    const double dt = cfg->dt;
    for (int64_t e = e0; e < e1; ++e) {
        double *pos = pos_a + 2 * M * e, *vel = vel_a + 2 * M * e, *acc = acc_a + 2 * M * e, *goal = goal_a + 2 * M * e;
        double m = INFINITY;
        double *margin = out->margin ? &m : NULL;
        if (autoreset == 2 && elapsed_a[e] < 0) { /* next-step autoreset: the action is ignored (Gymnasium AutoresetMode.NEXT_STEP) */
            reset_one(c, seed, tick, env_id_offset + e, e, pos, vel, acc, goal, elapsed_a + e, out, margin);
            out->reward[e] = 0.0;
            out->terminated[e] = 0;
            if (out->margin) out->margin[e] = m;
            continue;
        }
        uint32_t keys[GPRO_MAX_MOVERS];
        for (int i = 0; i < M; ++i) keys[i] = gpro_rng_key(seed, tick, KIND_STEP_NOISE, (uint64_t)(env_id_offset + e), (uint32_t)i);
        /* BE:1869-1873: clip to the action space (+-a_max, or +-j_max when learn_jerk; PL:257-259) */
        double lim = cfg->learn_jerk ? cfg->j_max : cfg->a_max;
        double a[GPRO_MAX_MOVERS * 2];
        for (int k = 0; k < 2 * M; ++k) a[k] = fmin(fmax(action[2 * M * e + k], -lim), lim);
        int wc = 0, mc = 0;
        const uint32_t R = (uint32_t)reads_per_check(c), S = 1u + 2u * R; /* normal pairs drawn per mover per cycle */
        for (int cyc = 0; cyc < cfg->num_cycles; ++cyc) { /* BE:1879 */
            /* _mujoco_step_callback (PL:420-450) */
            double nv[GPRO_MAX_MOVERS * 2], u[GPRO_MAX_MOVERS * 2];
            noisy(c, keys, S * (uint32_t)cyc, M, vel, cfg->std_noise[1], nv); /* PL:430 */
            for (int i = 0; i < M; ++i) {
                double vals[2], ders[2];
                if (cfg->learn_jerk) { /* PL:432-440 */
                    double acc_tmp[2], j1[2];
                    clamp_one(acc + 2 * i, cfg->a_max, a + 2 * i, dt, acc_tmp, j1, margin);
                    clamp_one(nv + 2 * i, cfg->v_max, acc_tmp, dt, vals, ders, margin);
                    if (acc_tmp[0] != ders[0] || acc_tmp[1] != ders[1]) {
                        u[2 * i] = (ders[0] - acc[2 * i]) / dt;
                        u[2 * i + 1] = (ders[1] - acc[2 * i + 1]) / dt;
                    } else {
                        u[2 * i] = j1[0]; u[2 * i + 1] = j1[1];
                    }
                } else { /* PL:442-443 */
                    clamp_one(nv + 2 * i, cfg->v_max, a + 2 * i, dt, vals, ders, margin);
                    u[2 * i] = ders[0]; u[2 * i + 1] = ders[1];
                }
            }
            /* mj_step (BE:1882) for the MJCF of BE:878-883,1132 + PL:305-319: qacc = ctrl (or the integrated activation,
             * actearly), semi-implicit Euler: v += dt*qacc; p += dt*v_new. Equivalence asserted by the reference's own
             * tests (tests/test_benchmark_planning_env.py:86-121,199-231). */
            for (int k = 0; k < 2 * M; ++k) {
                double qacc;
                if (cfg->learn_jerk) { acc[k] = acc[k] + dt * u[k]; qacc = acc[k]; }
                else { qacc = u[k]; acc[k] = qacc; }
                vel[k] = vel[k] + dt * qacc;
                pos[k] = pos[k] + dt * vel[k];
            }
            double qp[GPRO_MAX_MOVERS * 7];
            noisy_qpos(c, keys, S * (uint32_t)cyc + 1u, M, pos, cfg->std_noise[0], 1, qp);
            wc = wall_any(c, qp, 0, margin); /* BE:1888-1894 */
            noisy_qpos(c, keys, S * (uint32_t)cyc + 1u + R, M, pos, cfg->std_noise[0], 1, qp);
            mc = pair_any(c, qp, 0, margin); /* BE:1895-1901 */
            if (mc || wc) break;              /* BE:1904-1905 */
        }
        for (int k = 0; k < 2 * M; ++k) { pos[k] = q32(cfg, pos[k]); vel[k] = q32(cfg, vel[k]); acc[k] = q32(cfg, acc[k]); }
        observe(c, keys, S * (uint32_t)cfg->num_cycles, pos, vel, acc, goal, mc, wc, e, out, margin);
        elapsed_a[e] += 1;
        int trunc = cfg->max_episode_steps > 0 && elapsed_a[e] >= cfg->max_episode_steps; /* gymnasium TimeLimit */
        out->truncated[e] = (uint8_t)trunc;
        int done = out->terminated[e] || trunc;
        if (done && autoreset == 1) {
            if (out->final_observation) memcpy(out->final_observation + od * e, out->observation + od * e, sizeof(double) * od);
            if (out->final_achieved_goal) memcpy(out->final_achieved_goal + 2 * M * e, out->achieved_goal + 2 * M * e, sizeof(double) * 2 * M);
            if (out->final_desired_goal) memcpy(out->final_desired_goal + 2 * M * e, out->desired_goal + 2 * M * e, sizeof(double) * 2 * M);
            double r = out->reward[e];
            uint8_t t = out->terminated[e], s = out->is_success[e], pm = out->mover_collision[e], pw = out->wall_collision[e];
            reset_one(c, seed, tick, env_id_offset + e, e, pos, vel, acc, goal, elapsed_a + e, out, margin);
            out->reward[e] = r; out->terminated[e] = t; out->truncated[e] = (uint8_t)trunc;
            out->is_success[e] = s; out->mover_collision[e] = pm; out->wall_collision[e] = pw;
        } else if (done && autoreset == 2) {
            elapsed_a[e] = -1;
        }
        if (out->margin) out->margin[e] = m;
    }
}

int32_t gpro_planning_step(const gpro_env_cfg *cfg, int64_t num_envs, uint64_t seed, uint64_t tick, int64_t env_id_offset,
                           const double *action, double *pos, double *vel, double *acc, double *goal, int32_t *elapsed, int32_t autoreset,
                           gpro_step_out *out) {
    if (cfg->num_movers < 1 || cfg->num_movers > GPRO_MAX_MOVERS) return -1;
    env_ctx c;
    ctx_init(&c, cfg);
    step_range(&c, 0, num_envs, seed, tick, env_id_offset, action, pos, vel, acc, goal, elapsed, autoreset, out);
    return 0;
}

typedef struct {
    const env_ctx *c; int64_t e0, e1; uint64_t seed, tick; int64_t off; const double *action;
    double *pos, *vel, *acc, *goal; int32_t *elapsed; int32_t autoreset; gpro_step_out *out;
} mt_arg;

static void *mt_worker(void *p) {
    mt_arg *a = (mt_arg *)p;
    step_range(a->c, a->e0, a->e1, a->seed, a->tick, a->off, a->action, a->pos, a->vel, a->acc, a->goal, a->elapsed, a->autoreset, a->out);
    return NULL;
}

int32_t gpro_planning_step_mt(const gpro_env_cfg *cfg, int64_t num_envs, uint64_t seed, uint64_t tick, int64_t env_id_offset,
                              const double *action, double *pos, double *vel, double *acc, double *goal, int32_t *elapsed, int32_t autoreset,
                              gpro_step_out *out, int32_t num_threads) {
    if (cfg->num_movers < 1 || cfg->num_movers > GPRO_MAX_MOVERS) return -1;
    if (num_threads < 1) num_threads = 1;
    if (num_threads > 256) num_threads = 256;
    env_ctx c;
    ctx_init(&c, cfg);
    pthread_t th[256];
    mt_arg args[256];
    int64_t per = (num_envs + num_threads - 1) / num_threads;
    int n = 0;
    for (int t = 0; t < num_threads; ++t) {
        int64_t e0 = t * per, e1 = e0 + per > num_envs ? num_envs : e0 + per;
        if (e0 >= e1) break;
        args[n] = (mt_arg){&c, e0, e1, seed, tick, env_id_offset, action, pos, vel, acc, goal, elapsed, autoreset, out};
        pthread_create(&th[n], NULL, mt_worker, &args[n]);
        ++n;
    }
    for (int t = 0; t < n; ++t) pthread_join(th[t], NULL);
    return 0;
}

/* ======================================================================================================
 * BenchmarkPushingEnv (PU = envs/manipulation/benchmark_pushing_env.py). See gpr_oracle.h for what is pinned and what is
 * the engine's own contact model "gpr-push-v3".
 * ==================================================================================================== */
typedef struct { double x, y, th, vx, vy, w; } body2;


/* ------------------------------------------------------------------------------------------------------------------
 * Engine contact model "gpr-push-v3" (the engine's own -- the reference hands contact to MuJoCo and no
 * reference test exercises it: PARITY UNPINNED). One control cycle of nm (1 or 2) movers and the object; u[i] = commanded
 * acceleration of mover i (world x, y). Semi-implicit Euler: velocities first (external forces, then contact impulses), then
 * positions with the new velocities. Contacts: mover i <-> object for every mover (the movers do not touch each other: coming within
 * their collision circles ends the episode, as in the planning env).
 *
 * Per (mover, object) pair, in the MOVER's frame:
 *  - broad phase on the circumscribed circles; separating-axis test over the four face normals in the order mover x, mover y,
 *    object x, object y, the first smallest overlap wins: the contact normal n (mover -> object) and the REFERENCE box (the one
 *    that owns the axis); the other box is the INCIDENT one;
 *  - manifold: the incident box's face that faces the reference box most squarely gives the incident segment (two vertices);
 *    it is cut to the lateral extent of the reference face (along the segment's own line); an end whose depth below the reference
 *    face is >= 0 is a contact point: at most TWO points, with their depths. (v2 took the first <= 4 of the eight "vertex inside
 *    the other box" candidates.)
 *  - rows: one normal row per point, ONE friction row at the midpoint of the points, bounded by mu (lambda_0 + lambda_1). (v2: a
 *    friction row per point.)
 *  - solve: block Gauss-Seidel on the TOTAL impulses of the cycle. A visit of a pair solves its normal block EXACTLY -- the 2 x 2
 *    linear complementarity problem A l + b >= 0, l >= 0, l (A l + b) = 0 by enumerating the active sets (both, first, second,
 *    none) -- with the friction impulse held, then sets the friction impulse for the new normals (clamped). Two visits per pair
 *    and cycle, in the order pair 0, pair 1, pair 0, pair 1. (v2: four sweeps of scalar sequential impulses.) With ONE mover the
 *    two visits run on the impulses alone (the row velocities are affine in them: b + A lambda), the bodies take the result once.
 *  - penetration recovery: velocity bias depth / contact_timeconst per normal row, as in v2.
 *  - contact margin PUSH3_MARGIN (0.1 mm): boxes closer than the margin are "touching", and an end of the incident segment up to the
 *    margin ABOVE the reference face stays in the manifold as a speculative row -- its bias is depth / dt (negative: the gap may close
 *    within the cycle, not more). Without it a face-to-face push lost and regained its second point with every micro-radian of relative
 *    yaw (depths of 1e-7 m), and each loss turned the object by 0.1 rad/s.
 * fma(a, b, c) marks the fused operations: the device code fuses exactly these (the double builds follow this text operation for
 * operation; sin / cos come from the device's libm there, hence 1e-9 instead of bit equality in the tests).
 * ------------------------------------------------------------------------------------------------------------------ */
#define PUSH3_MARGIN 1e-4
typedef struct {
    int np;                       /* 0, 1, 2 contact points */
    double nx, ny;                /* contact normal, mover -> object, world frame */
    double rmn[2], ron[2];        /* lever arms x n of the two normal rows (mover's, object's) */
    double rmt, rot_;             /* lever arms x t of the friction row, t = (-ny, nx) */
    double bias[2];
    double a00, a01, a11, att, ant0, ant1; /* A = J M^-1 J^T of the three rows */
    double i00, i11, idet, itt;   /* 1 / a00, 1 / a11, 1 / (a00 a11 - a01^2), 1 / att: formed once per cycle, both visits multiply */
    double l0, l1, jt;            /* total impulses of the cycle so far */
} push3_pair;

/* 2 x 2 LCP  A l + b >= 0, l >= 0, complementary; A symmetric positive definite: exactly one active set is consistent */
static void push3_lcp2(const push3_pair *q, double b0, double b1, double *l0, double *l1) {
    const int two = q->np == 2;
    const double s0 = fmax(-b0 * q->i00, 0.0);        /* only row 0 active */
    const double x0 = fma(q->a01, b1, -(q->a11 * b0)) * q->idet, x1 = fma(q->a01, b0, -(q->a00 * b1)) * q->idet; /* both rows active */
    const double s1 = fmax(-b1 * q->i11, 0.0);        /* only row 1 active */
    const int both = two && x0 >= 0.0 && x1 >= 0.0;
    const int first = !both && (!two || fma(q->a01, s0, b1) >= 0.0);
    *l0 = both ? x0 : first ? s0 : 0.0;
    *l1 = both ? x1 : first ? 0.0 : s1;
}

/* contact geometry of one pair; dx, dy = object centre - mover centre (world). Returns q->np. */
static int push3_geometry(const gpro_push_params *pp, double dt, const body2 *m, const body2 *o, double dx, double dy, double inv_mm, double inv_mo,
                          double inv_Im, double inv_Io, push3_pair *q) {
    const double ax = pp->mover_half[0], ay = pp->mover_half[1], h = pp->object_half;
    q->np = 0; q->nx = 0.0; q->ny = 0.0; q->l0 = q->l1 = q->jt = 0.0;
    const double reach = sqrt(ax * ax + ay * ay) + h * sqrt(2.0) + PUSH3_MARGIN, reach2 = reach * reach;
    if (fma(dx, dx, dy * dy) > reach2) return 0; /* broad phase: the circumscribed circles (+ margin) are apart */
    const double cm = cos(m->th), sm = sin(m->th), co = cos(o->th), so = sin(o->th);
    const double c = fma(co, cm, so * sm), s = fma(so, cm, -(co * sm));     /* the object's axes in the mover's frame: (c, s), (-s, c) */
    const double lx = fma(dx, cm, dy * sm), ly = fma(dy, cm, -(dx * sm));   /* the object's centre in the mover's frame */
    const double ac = fabs(c), as = fabs(s);
    const double ro01 = h * (ac + as);
    const double d2 = fma(lx, c, ly * s), d3 = fma(ly, c, -(lx * s));
    const double ov0 = (ax + ro01) - fabs(lx), ov1 = (ay + ro01) - fabs(ly);
    const double ov2 = (fma(ax, ac, ay * as) + h) - fabs(d2), ov3 = (fma(ax, as, ay * ac) + h) - fabs(d3);
    if (fmin(fmin(ov0, ov1), fmin(ov2, ov3)) < -PUSH3_MARGIN) return 0;
    int k = 0; double best = ov0;
    if (ov1 < best) { best = ov1; k = 1; }
    if (ov2 < best) { best = ov2; k = 2; }
    if (ov3 < best) { best = ov3; k = 3; }
    /* everything below in the mover's frame. Reference box: centre (rcx, rcy), outward normal (rnx, rny) towards the incident box, face
     * tangent (-rny, rnx), half extents en (along the normal) and et (along the face). Incident box: centre (icx, icy), axes (ux, uy) and
     * (vx, vy) with half extents hu, hv. nlx, nly: the contact normal mover -> object. */
    double nlx, nly, rcx, rcy, rnx, rny, en, et, icx, icy, ux, uy, vx, vy, hu, hv;
    if (k < 2) { /* the mover's face is the reference face */
        nlx = k == 0 ? copysign(1.0, lx) : 0.0; nly = k == 0 ? 0.0 : copysign(1.0, ly);
        rcx = 0.0; rcy = 0.0; rnx = nlx; rny = nly; en = k == 0 ? ax : ay; et = k == 0 ? ay : ax;
        icx = lx; icy = ly; ux = c; uy = s; vx = -s; vy = c; hu = h; hv = h;
    } else {     /* the object's face is the reference face */
        const double sg = copysign(1.0, k == 2 ? d2 : d3);
        nlx = sg * (k == 2 ? c : -s); nly = sg * (k == 2 ? s : c);
        rcx = lx; rcy = ly; rnx = -nlx; rny = -nly; en = h; et = h;
        icx = 0.0; icy = 0.0; ux = 1.0; uy = 0.0; vx = 0.0; vy = 1.0; hu = ax; hv = ay;
    }
    /* incident face: the one whose outward normal is most opposed to the reference normal */
    const double pu = fma(rnx, ux, rny * uy), pv = fma(rnx, vx, rny * vy);
    double fx, fy, ex, ey; /* face centre offset from the incident centre, half edge vector */
    if (fabs(pu) >= fabs(pv)) { const double sgn = -copysign(1.0, pu); fx = sgn * hu * ux; fy = sgn * hu * uy; ex = hv * vx; ey = hv * vy; }
    else                      { const double sgn = -copysign(1.0, pv); fx = sgn * hv * vx; fy = sgn * hv * vy; ex = hu * ux; ey = hu * uy; }
    const double bx = (icx + fx) - rcx, by = (icy + fy) - rcy;               /* face centre relative to the reference centre */
    const double pax = bx - ex, pay = by - ey, pbx = bx + ex, pby = by + ey; /* the incident segment's ends */
    const double tx = -rny, ty = rnx;
    const double sa = fma(pax, tx, pay * ty), sb = fma(pbx, tx, pby * ty);   /* lateral coordinates on the reference face */
    const double da = en - fma(pax, rnx, pay * rny), db = en - fma(pbx, rnx, pby * rny); /* depths below the reference face */
    if (fmin(sa, sb) > et || fmax(sa, sb) < -et) return 0;                   /* the segment passes the face by */
    const double inv = 1.0 / (sb - sa);
    const double ta = (fmin(fmax(sa, -et), et) - sa) * inv, tb = (fmin(fmax(sb, -et), et) - sa) * inv; /* cut to [-et, et] along the segment */
    double px[2], py[2], dep[2];
    px[0] = fma(ta, pbx - pax, pax) + rcx; py[0] = fma(ta, pby - pay, pay) + rcy; dep[0] = fma(ta, db - da, da);
    px[1] = fma(tb, pbx - pax, pax) + rcx; py[1] = fma(tb, pby - pay, pay) + rcy; dep[1] = fma(tb, db - da, da);
    /* points with depth >= -margin, the deeper one first */
    const int k0 = dep[1] > dep[0] ? 1 : 0, k1 = 1 - k0;
    if (dep[k0] < -PUSH3_MARGIN) return 0;
    int np = dep[k1] >= -PUSH3_MARGIN ? 2 : 1;
    const double p0x = px[k0], p0y = py[k0], p1x = np == 2 ? px[k1] : p0x, p1y = np == 2 ? py[k1] : p0y;
    const double d0 = dep[k0], d1 = np == 2 ? dep[k1] : 0.0;
    /* rows: r x n and r x t are the same in every frame; the mover's lever arm is the point itself, the object's the point - (lx, ly) */
    const double mxc = 0.5 * (p0x + p1x), myc = 0.5 * (p0y + p1y);           /* friction row: midpoint of the points */
    q->rmn[0] = fma(p0x, nly, -(p0y * nlx)); q->ron[0] = fma(p0x - lx, nly, -((p0y - ly) * nlx));
    q->rmn[1] = fma(p1x, nly, -(p1y * nlx)); q->ron[1] = fma(p1x - lx, nly, -((p1y - ly) * nlx));
    q->rmt = fma(mxc, nlx, myc * nly); q->rot_ = fma(mxc - lx, nlx, (myc - ly) * nly);
    const double imm = inv_mm + inv_mo;
    const double wm0 = q->rmn[0] * inv_Im, wo0 = q->ron[0] * inv_Io, wm1 = q->rmn[1] * inv_Im, wo1 = q->ron[1] * inv_Io;
    const double wmt = q->rmt * inv_Im, wot = q->rot_ * inv_Io;
    q->a00 = fma(q->rmn[0], wm0, fma(q->ron[0], wo0, imm));
    q->a11 = fma(q->rmn[1], wm1, fma(q->ron[1], wo1, imm));
    q->a01 = fma(q->rmn[0], wm1, fma(q->ron[0], wo1, imm));
    q->att = fma(q->rmt, wmt, fma(q->rot_, wot, imm));
    q->ant0 = fma(q->rmn[0], wmt, q->ron[0] * wot);
    q->ant1 = fma(q->rmn[1], wmt, q->ron[1] * wot);
    /* two points that (nearly) coincide make the block singular: the deeper one stands for both */
    const double det = fma(q->a00, q->a11, -(q->a01 * q->a01));
    if (np == 2 && !(det > 1e-4 * (q->a00 * q->a11))) np = 1;
    q->i00 = 1.0 / q->a00; q->i11 = 1.0 / q->a11; q->idet = 1.0 / det; q->itt = 1.0 / q->att;
    q->bias[0] = d0 >= 0.0 ? d0 / pp->contact_timeconst : d0 / dt; /* recovery; speculative: the gap may close within the cycle */
    q->bias[1] = d1 >= 0.0 ? d1 / pp->contact_timeconst : d1 / dt;
    q->nx = fma(nlx, cm, -(nly * sm)); q->ny = fma(nlx, sm, nly * cm);
    q->np = np;
    return np;
}

/* one visit of a pair: the normal block exactly with the friction impulse held, then the friction impulse; (vn0, vn1, vt) = the row
 * velocities WITHOUT this pair's own impulses */
static void push3_visit(const gpro_push_params *pp, push3_pair *q, double vn0, double vn1, double vt) {
    const double b0 = fma(q->ant0, q->jt, vn0 - q->bias[0]), b1 = fma(q->ant1, q->jt, vn1 - q->bias[1]);
    push3_lcp2(q, b0, b1, &q->l0, &q->l1);
    const double lim = pp->friction_contact * (q->l0 + q->l1);
    const double r = fma(q->ant1, q->l1, fma(q->ant0, q->l0, vt));
    q->jt = fmin(fmax(-r * q->itt, -lim), lim);
}

static void push_cycle(const gpro_env_cfg *cfg, const gpro_push_params *pp, int nm, body2 *mv, body2 *o, const double (*u)[2]) {
    const double dt = cfg->dt;
    const double ax = pp->mover_half[0], ay = pp->mover_half[1], h = pp->object_half;
    const double Im = pp->mover_mass * (ax * ax + ay * ay) / 3.0, Io = pp->object_mass * (2.0 * h * h) / 3.0;
    const double inv_mm = 1.0 / pp->mover_mass, inv_mo = 1.0 / pp->object_mass, inv_Im = 1.0 / Im, inv_Io = 1.0 / Io;
    /* 1. external forces. mover: actuator force m*u (PU:305-319); yaw held by the impedance law tau = K(0 - th) - D w with
     * D = 2 sqrt(K m) (IMP:42-47, PU:261-267), damping treated implicitly. */
    const double Kr = pp->rot_stiffness, Dr = 2.0 * sqrt(Kr * pp->mover_mass);
    for (int i = 0; i < nm; ++i) {
        body2 *m = mv + i;
        m->vx = fma(dt, u[i][0], m->vx);
        m->vy = fma(dt, u[i][1], m->vy);
        m->w = fma(-(dt * Kr * inv_Im), m->th, m->w) / (1.0 + dt * Dr * inv_Im);
    }
    /* object: free-joint damping (PU:335, implicit like MuJoCo's Euler integrator), then Coulomb friction with the tile,
     * never reversing the motion */
    {
        const double sdl = 1.0 / (1.0 + dt * pp->object_damping * inv_mo);
        const double vx = o->vx * sdl, vy = o->vy * sdl, w = o->w / (1.0 + dt * pp->object_damping * inv_Io);
        const double sp = sqrt(fma(vx, vx, vy * vy)), dv = pp->friction_tile * pp->gravity * dt;
        const double kf = sp > dv ? (sp - dv) / sp : 0.0;
        o->vx = vx * kf; o->vy = vy * kf;
        const double dw = pp->friction_torsional * pp->object_mass * pp->gravity * inv_Io * dt, aw = fabs(w);
        o->w = aw <= dw ? 0.0 : copysign(aw - dw, w);
    }
    /* 2. contact geometry per pair */
    push3_pair pr[2];
    int any = 0;
    for (int i = 0; i < nm; ++i) any |= push3_geometry(pp, dt, mv + i, o, o->x - mv[i].x, o->y - mv[i].y, inv_mm, inv_mo, inv_Im, inv_Io, pr + i) > 0;
    /* 3. block Gauss-Seidel on the cycle's total impulses: two visits per pair, pair 0, pair 1, pair 0, pair 1 */
    if (any && nm == 1) { /* one pair: both visits on the impulses alone, the bodies take the result once */
        push3_pair *q = pr;
        body2 *m = mv;
        const double dvx = o->vx - m->vx, dvy = o->vy - m->vy;
        const double dvn = fma(dvx, q->nx, dvy * q->ny), dvt = fma(dvy, q->nx, -(dvx * q->ny));
        const double vn0 = fma(o->w, q->ron[0], fma(-m->w, q->rmn[0], dvn)), vn1 = fma(o->w, q->ron[1], fma(-m->w, q->rmn[1], dvn));
        const double vt = fma(o->w, q->rot_, fma(-m->w, q->rmt, dvt));
        push3_visit(pp, q, vn0, vn1, vt);
        push3_visit(pp, q, vn0, vn1, vt);
        const double jn = q->l0 + q->l1;
        const double px = fma(jn, q->nx, -(q->jt * q->ny)), py = fma(jn, q->ny, q->jt * q->nx); /* impulse on the object; t = (-ny, nx) */
        m->vx = fma(-px, inv_mm, m->vx); m->vy = fma(-py, inv_mm, m->vy);
        o->vx = fma(px, inv_mo, o->vx); o->vy = fma(py, inv_mo, o->vy);
        m->w -= fma(q->l1, q->rmn[1], fma(q->l0, q->rmn[0], q->jt * q->rmt)) * inv_Im;
        o->w += fma(q->l1, q->ron[1], fma(q->l0, q->ron[0], q->jt * q->rot_)) * inv_Io;
    } else if (any) {
        for (int it = 0; it < 2; ++it)
            for (int i = 0; i < nm; ++i) {
                push3_pair *q = pr + i;
                body2 *m = mv + i;
                if (q->np == 0) continue;
                const double dvx = o->vx - m->vx, dvy = o->vy - m->vy;
                const double dvn = fma(dvx, q->nx, dvy * q->ny), dvt = fma(dvy, q->nx, -(dvx * q->ny));
                /* the rows' velocities now, then without this pair's own impulses */
                double vn0 = fma(o->w, q->ron[0], fma(-m->w, q->rmn[0], dvn)), vn1 = fma(o->w, q->ron[1], fma(-m->w, q->rmn[1], dvn));
                double vt = fma(o->w, q->rot_, fma(-m->w, q->rmt, dvt));
                vn0 -= fma(q->ant0, q->jt, fma(q->a01, q->l1, q->a00 * q->l0));
                vn1 -= fma(q->ant1, q->jt, fma(q->a11, q->l1, q->a01 * q->l0));
                vt -= fma(q->att, q->jt, fma(q->ant1, q->l1, q->ant0 * q->l0));
                const double l0o = q->l0, l1o = q->l1, jto = q->jt;
                push3_visit(pp, q, vn0, vn1, vt);
                const double d0 = q->l0 - l0o, d1 = q->l1 - l1o, djt = q->jt - jto, djn = d0 + d1;
                const double px = fma(djn, q->nx, -(djt * q->ny)), py = fma(djn, q->ny, djt * q->nx);
                m->vx = fma(-px, inv_mm, m->vx); m->vy = fma(-py, inv_mm, m->vy);
                o->vx = fma(px, inv_mo, o->vx); o->vy = fma(py, inv_mo, o->vy);
                m->w -= fma(d1, q->rmn[1], fma(d0, q->rmn[0], djt * q->rmt)) * inv_Im;
                o->w += fma(d1, q->ron[1], fma(d0, q->ron[0], djt * q->rot_)) * inv_Io;
            }
    }
    /* 4. positions with the new velocities */
    for (int i = 0; i < nm; ++i) { body2 *m = mv + i; m->x = fma(dt, m->vx, m->x); m->y = fma(dt, m->vy, m->y); m->th = fma(dt, m->w, m->th); }
    o->x = fma(dt, o->vx, o->x); o->y = fma(dt, o->vy, o->y); o->th = fma(dt, o->w, o->th);
}

/* _get_obs (PU:529-576), _get_info (PU:578-607), compute_reward / compute_terminated (PU:457-527); nm movers: the observation
 * keeps the reference's np.concatenate((pos, vel[, acc]), axis=0).flatten() = all positions, all velocities[, all accelerations];
 * a mover-mover collision (2 movers) counts like a wall collision */
static void push_observe(const env_ctx *c, const gpro_push_params *pp, const uint32_t *keys, uint32_t q0, const double *pos, const double *vel,
                         const double *acc, const double *object, const double *goal, int wc, int mc, int64_t e, gpro_step_out *out,
                         double *margin) {
    const gpro_env_cfg *cfg = c->cfg;
    const int nm = cfg->num_movers;
    const int od = (cfg->learn_jerk ? 6 : 4) * nm;
    double *o = out->observation + od * e;
    for (int i = 0; i < nm; ++i) {
        double n0 = 0, n1 = 0, n2 = 0, n3 = 0;
        if (c->noise) { gpro_rng_normal_pair(keys[i], q0, &n0, &n1); gpro_rng_normal_pair(keys[i], q0 + 1u, &n2, &n3); }
        o[2 * i] = q32(cfg, pos[2 * i] + cfg->std_noise[0] * n0); o[2 * i + 1] = q32(cfg, pos[2 * i + 1] + cfg->std_noise[0] * n1); /* PU:551 */
        o[2 * nm + 2 * i] = q32(cfg, vel[2 * i] + cfg->std_noise[1] * n2); o[2 * nm + 2 * i + 1] = q32(cfg, vel[2 * i + 1] + cfg->std_noise[1] * n3); /* PU:552 */
        if (cfg->learn_jerk) { o[4 * nm + 2 * i] = q32(cfg, acc[2 * i]); o[4 * nm + 2 * i + 1] = q32(cfg, acc[2 * i + 1]); }      /* PU:553-555 */
    }
    double n4 = 0, n5 = 0;
    if (pp->object_noise != 0.0) gpro_rng_normal_pair(keys[0], q0 + 2u, &n4, &n5);
    double *ag = out->achieved_goal + 2 * e, *dg = out->desired_goal + 2 * e;
    ag[0] = q32(cfg, object[0] + pp->object_noise * n4); ag[1] = q32(cfg, object[1] + pp->object_noise * n5); /* PU:564-565 */
    dg[0] = q32(cfg, goal[0]); dg[1] = q32(cfg, goal[1]);
    double dx = ag[0] - dg[0], dy = ag[1] - dg[1], dist = sqrt(dx * dx + dy * dy);
    track(margin, dist, cfg->threshold_pos);
    int reached = dist <= cfg->threshold_pos;
    const int coll = wc || mc;
    double r = -50.0 * (double)(coll != 0);     /* PU:521, reward_wall_collision = -50 (PU:219) */
    r += -1.0 * (double)(!coll);                /* PU:522 */
    if (reached && !coll) r = 0.0;              /* PU:523 */
    out->reward[e] = r;
    out->terminated[e] = (r == -50.0);          /* PU:475 */
    out->is_success[e] = reached && !coll;      /* PU:602 */
    out->mover_collision[e] = (uint8_t)mc;
    out->wall_collision[e] = (uint8_t)wc;
}

/* PU:373-417. opt_start / opt_goal: the reset options 'mover_start_xy_pos' [2 nm] / 'object_goal_xy_pos' [2] of this env (NULL:
 * sampled). 2 movers: their starts are re-drawn together until the two collision circles (with the safety offset) are apart. */
static void push_reset_one(const env_ctx *c, const gpro_push_params *pp, uint64_t seed, uint64_t tick, int64_t env_id, int64_t e, double *pos,
                           double *vel, double *acc, double *rot, double *object, double *goal, int32_t *elapsed, const double *opt_start,
                           const double *opt_goal, gpro_step_out *out, double *margin, const cand_stream *cs) {
    const gpro_env_cfg *cfg = c->cfg;
    const int nm = cfg->num_movers;
    uint32_t ks[2], kr[2], kn[2];
    for (int i = 0; i < nm; ++i) {
        ks[i] = gpro_rng_key(seed, tick, KIND_RESET_START, (uint64_t)env_id, (uint32_t)i);
        kr[i] = gpro_rng_key(seed, tick, KIND_RESET_START, (uint64_t)env_id, (uint32_t)(GPRO_MAX_MOVERS + i));
        kn[i] = gpro_rng_key(seed, tick, KIND_RESET_NOISE, (uint64_t)env_id, (uint32_t)i);
    }
    const uint32_t kg = gpro_rng_key(seed, tick, KIND_RESET_GOAL, (uint64_t)env_id, 0u);
    if (opt_start) {
        for (int k = 0; k < 2 * nm; ++k) pos[k] = q32(cfg, opt_start[k]); /* PU:389-391 */
    } else {
        for (int trial = 0; trial < cfg->max_reset_trials; ++trial) {
            for (int i = 0; i < nm; ++i)
                for (int d = 0; d < 2; ++d) { /* PU:389-391; re-draws (2 movers only) come from a stream of their own */
                    const double u = trial == 0 ? gpro_rng_uniform(ks[i], (uint32_t)d) : gpro_rng_uniform(kr[i], 2u * (uint32_t)(trial - 1) + (uint32_t)d);
                    pos[2 * i + d] = q32(cfg, cfg->min_xy[d] + (cfg->max_xy[d] - cfg->min_xy[d]) * u);
                }
            if (nm == 1) break;
            double qp[2 * 7];
            noisy_qpos(c, ks, 0, nm, pos, 0.0, 0, qp);
            if (!pair_any(c, qp, 1, margin)) break;
        }
    }
    const int otrials = cs ? (cs->n < cfg->max_reset_trials ? cs->n : cfg->max_reset_trials) : cfg->max_reset_trials;
    for (int trial = 0; trial < otrials; ++trial) { /* PU:396-407 */
        for (int d = 0; d < 2; ++d)
            object[d] = cs ? q32(cfg, cs->cand[(int64_t)trial * cs->stride + d])
                           : q32(cfg, pp->obj_min_xy[d] + (pp->obj_max_xy[d] - pp->obj_min_xy[d]) * gpro_rng_uniform(ks[0], 2u + 2u * (uint32_t)trial + (uint32_t)d));
        if (cs && cs->n_used) *cs->n_used = trial + 1;
        int ok = 1;
        for (int i = 0; i < nm; ++i) {
            double dx = object[0] - pos[2 * i], dy = object[1] - pos[2 * i + 1], dist = sqrt(dx * dx + dy * dy);
            track(margin, dist, pp->min_mo_dist);
            if (!(dist > pp->min_mo_dist)) ok = 0; /* PU:407, strict, every mover */
        }
        if (ok) break;
    }
    for (int d = 0; d < 2; ++d) /* PU:412-414 */
        goal[d] = opt_goal ? q32(cfg, opt_goal[d]) : q32(cfg, pp->obj_min_xy[d] + (pp->obj_max_xy[d] - pp->obj_min_xy[d]) * gpro_rng_uniform(kg, (uint32_t)d));
    for (int k = 0; k < 2 * nm; ++k) { vel[k] = 0.0; acc[k] = 0.0; rot[k] = 0.0; }
    object[2] = object[3] = object[4] = object[5] = 0.0;
    *elapsed = 0;
    double qp[2 * 7];
    noisy_qpos(c, kn, 0, nm, pos, cfg->std_noise[0], 1, qp);
    int wc = wall_any(c, qp, 1, margin); /* BE:1799-1801 */
    int mc = 0;
    if (nm > 1) { noisy_qpos(c, kn, 1, nm, pos, cfg->std_noise[0], 1, qp); mc = pair_any(c, qp, 0, margin); } /* BE:1803-1805 */
    push_observe(c, pp, kn, 2u, pos, vel, acc, object, goal, wc, mc, e, out, margin);
    out->truncated[e] = 0;
}

int32_t gpro_pushing_reset(const gpro_env_cfg *cfg, const gpro_push_params *pp, int64_t num_envs, uint64_t seed, uint64_t tick,
                           int64_t env_id_offset, const uint8_t *mask, const double *opt_mover_start, const double *opt_goal, double *pos,
                           double *vel, double *acc, double *mover_rot, double *object, double *goal, int32_t *elapsed, gpro_step_out *out) {
    const int nm = cfg->num_movers;
    if (nm < 1 || nm > 2) return -1;
    env_ctx c;
    ctx_init(&c, cfg);
    for (int64_t e = 0; e < num_envs; ++e) {
        if (mask && !mask[e]) continue;
        double m = INFINITY;
        push_reset_one(&c, pp, seed, tick, env_id_offset + e, e, pos + 2 * nm * e, vel + 2 * nm * e, acc + 2 * nm * e, mover_rot + 2 * nm * e,
                       object + 6 * e, goal + 2 * e, elapsed + e, opt_mover_start ? opt_mover_start + 2 * nm * e : NULL,
                       opt_goal ? opt_goal + 2 * e : NULL, out, out->margin ? &m : NULL, NULL);
        out->reward[e] = 0.0; out->terminated[e] = 0;
        if (out->margin) out->margin[e] = m;
    }
    return 0;
}

/* gpro_pushing_reset with the object's start candidates scripted (PU:403-405): cand [n_cand][E][2], n_used [E] nullable */
int32_t gpro_pushing_reset_candidates(const gpro_env_cfg *cfg, const gpro_push_params *pp, int64_t num_envs, uint64_t seed, uint64_t tick,
                                      int64_t env_id_offset, const uint8_t *mask, const double *opt_mover_start, const double *opt_goal,
                                      int32_t n_cand, const double *cand, int32_t *n_used, double *pos, double *vel, double *acc,
                                      double *mover_rot, double *object, double *goal, int32_t *elapsed, gpro_step_out *out) {
    const int nm = cfg->num_movers;
    if (nm < 1 || nm > 2 || n_cand < 1 || !cand) return -1;
    env_ctx c;
    ctx_init(&c, cfg);
    for (int64_t e = 0; e < num_envs; ++e) {
        if (mask && !mask[e]) continue;
        double m = INFINITY;
        cand_stream cs = {cand + 2 * e, 2 * num_envs, n_cand, n_used ? n_used + e : NULL};
        push_reset_one(&c, pp, seed, tick, env_id_offset + e, e, pos + 2 * nm * e, vel + 2 * nm * e, acc + 2 * nm * e, mover_rot + 2 * nm * e,
                       object + 6 * e, goal + 2 * e, elapsed + e, opt_mover_start ? opt_mover_start + 2 * nm * e : NULL,
                       opt_goal ? opt_goal + 2 * e : NULL, out, out->margin ? &m : NULL, &cs);
        out->reward[e] = 0.0; out->terminated[e] = 0;
        if (out->margin) out->margin[e] = m;
    }
    return 0;
}

int32_t gpro_pushing_step(const gpro_env_cfg *cfg, const gpro_push_params *pp, int64_t num_envs, uint64_t seed, uint64_t tick,
                          int64_t env_id_offset, const double *action, double *pos_a, double *vel_a, double *acc_a, double *rot_a,
                          double *obj_a, double *goal_a, int32_t *elapsed_a, int32_t autoreset, gpro_step_out *out) {
    const int nm = cfg->num_movers;
    if (nm < 1 || nm > 2) return -1;
    env_ctx c;
    ctx_init(&c, cfg);
    const double dt = cfg->dt;
    const int od = (cfg->learn_jerk ? 6 : 4) * nm;
    for (int64_t e = 0; e < num_envs; ++e) {
        double *pos = pos_a + 2 * nm * e, *vel = vel_a + 2 * nm * e, *acc = acc_a + 2 * nm * e, *rot = rot_a + 2 * nm * e, *obj = obj_a + 6 * e,
               *goal = goal_a + 2 * e;
        double mg = INFINITY;
        double *margin = out->margin ? &mg : NULL;
        if (autoreset == 2 && elapsed_a[e] < 0) {
            push_reset_one(&c, pp, seed, tick, env_id_offset + e, e, pos, vel, acc, rot, obj, goal, elapsed_a + e, NULL, NULL, out, margin, NULL);
            out->reward[e] = 0.0; out->terminated[e] = 0;
            if (out->margin) out->margin[e] = mg;
            continue;
        }
        uint32_t keys[2];
        for (int i = 0; i < nm; ++i) keys[i] = gpro_rng_key(seed, tick, KIND_STEP_NOISE, (uint64_t)(env_id_offset + e), (uint32_t)i);
        const double lim = cfg->learn_jerk ? cfg->j_max : cfg->a_max;
        double a[4];
        for (int k = 0; k < 2 * nm; ++k) a[k] = fmin(fmax(action[2 * nm * e + k], -lim), lim);
        body2 m[2], o = {obj[0], obj[1], obj[2], obj[3], obj[4], obj[5]};
        for (int i = 0; i < nm; ++i) { body2 b = {pos[2 * i], pos[2 * i + 1], rot[2 * i], vel[2 * i], vel[2 * i + 1], rot[2 * i + 1]}; m[i] = b; }
        int wc = 0, mc = 0;
        for (int cyc = 0; cyc < cfg->num_cycles; ++cyc) {
            /* _mujoco_step_callback (PU:419-455), per mover */
            double u[2][2];
            for (int i = 0; i < nm; ++i) {
                double n0 = 0, n1 = 0;
                if (c.noise) gpro_rng_normal_pair(keys[i], 3u * (uint32_t)cyc, &n0, &n1);
                double nv[2] = {m[i].vx + cfg->std_noise[1] * n0, m[i].vy + cfg->std_noise[1] * n1}, uu[2], vals[2], ders[2];
                double *ac = acc + 2 * i;
                if (cfg->learn_jerk) {
                    double acc_tmp[2], j1[2];
                    clamp_one(ac, cfg->a_max, a + 2 * i, dt, acc_tmp, j1, margin);
                    clamp_one(nv, cfg->v_max, acc_tmp, dt, vals, ders, margin);
                    if (acc_tmp[0] != ders[0] || acc_tmp[1] != ders[1]) { uu[0] = (ders[0] - ac[0]) / dt; uu[1] = (ders[1] - ac[1]) / dt; }
                    else { uu[0] = j1[0]; uu[1] = j1[1]; }
                    ac[0] = ac[0] + dt * uu[0]; ac[1] = ac[1] + dt * uu[1];
                } else {
                    clamp_one(nv, cfg->v_max, a + 2 * i, dt, vals, ders, margin);
                    ac[0] = ders[0]; ac[1] = ders[1];
                }
                u[i][0] = ac[0]; u[i][1] = ac[1];
            }
            push_cycle(cfg, pp, nm, m, &o, (const double (*)[2])u); /* mj_step (BE:1882): engine model gpr-push-v3 */
            double xy[4], qp[2 * 7];
            for (int i = 0; i < nm; ++i) { xy[2 * i] = m[i].x; xy[2 * i + 1] = m[i].y; }
            noisy_qpos(&c, keys, 3u * (uint32_t)cyc + 1u, nm, xy, cfg->std_noise[0], 1, qp);
            wc = wall_any(&c, qp, 0, margin); /* BE:1888-1894 */
            if (nm > 1) { noisy_qpos(&c, keys, 3u * (uint32_t)cyc + 2u, nm, xy, cfg->std_noise[0], 1, qp); mc = pair_any(&c, qp, 0, margin); } /* BE:1896-1901 */
            if (wc || mc) break;
        }
        for (int i = 0; i < nm; ++i) {
            pos[2 * i] = q32(cfg, m[i].x); pos[2 * i + 1] = q32(cfg, m[i].y); vel[2 * i] = q32(cfg, m[i].vx); vel[2 * i + 1] = q32(cfg, m[i].vy);
            acc[2 * i] = q32(cfg, acc[2 * i]); acc[2 * i + 1] = q32(cfg, acc[2 * i + 1]); rot[2 * i] = q32(cfg, m[i].th); rot[2 * i + 1] = q32(cfg, m[i].w);
        }
        obj[0] = q32(cfg, o.x); obj[1] = q32(cfg, o.y); obj[2] = q32(cfg, o.th); obj[3] = q32(cfg, o.vx); obj[4] = q32(cfg, o.vy); obj[5] = q32(cfg, o.w);
        push_observe(&c, pp, keys, 3u * (uint32_t)cfg->num_cycles, pos, vel, acc, obj, goal, wc, mc, e, out, margin);
        elapsed_a[e] += 1;
        int trunc = cfg->max_episode_steps > 0 && elapsed_a[e] >= cfg->max_episode_steps;
        out->truncated[e] = (uint8_t)trunc;
        int done = out->terminated[e] || trunc;
        if (done && autoreset == 1) {
            if (out->final_observation) memcpy(out->final_observation + od * e, out->observation + od * e, sizeof(double) * od);
            if (out->final_achieved_goal) memcpy(out->final_achieved_goal + 2 * e, out->achieved_goal + 2 * e, sizeof(double) * 2);
            if (out->final_desired_goal) memcpy(out->final_desired_goal + 2 * e, out->desired_goal + 2 * e, sizeof(double) * 2);
            double r = out->reward[e];
            uint8_t t = out->terminated[e], s = out->is_success[e], pw = out->wall_collision[e], pm = out->mover_collision[e];
            push_reset_one(&c, pp, seed, tick, env_id_offset + e, e, pos, vel, acc, rot, obj, goal, elapsed_a + e, NULL, NULL, out, margin, NULL);
            out->reward[e] = r; out->terminated[e] = t; out->truncated[e] = (uint8_t)trunc; out->is_success[e] = s; out->wall_collision[e] = pw;
            out->mover_collision[e] = pm;
        } else if (done && autoreset == 2) {
            elapsed_a[e] = -1;
        }
        if (out->margin) out->margin[e] = mg;
    }
    return 0;
}
