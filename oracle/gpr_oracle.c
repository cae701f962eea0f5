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
 * Engine RNG spec "gpr-rng-v1" (engine-defined; the reference draws from NumPy PCG64 streams, which a GPU engine
 * can neither reproduce nor needs to: sigma = 1e-5 is 10x below the 1e-4 parity tolerance). Counter-based, 32-bit
 * ops only, order-independent, so the HIP kernels (one LANE per mover) and this oracle (one LOOP per env) produce the
 * same integers:
 *   key  h(seed, tick, kind, env, mover) = lowbias32 absorb chain
 *   draw r(h, k) = lowbias32(h + (k+1)*0x9E3779B9)                 (murmur-style finaliser over a Weyl sequence)
 *   normal pair q = Box-Muller on the two 16-bit halves of draw q; uniform k = upper 24 bits of draw k
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

void gpro_rng_normal_pair(uint32_t key, uint32_t q, double *n0, double *n1) {
    /* ONE 32-bit draw per pair: radius from the upper 16 bits, angle from the lower 16 (max radius 4.7 sigma) */
    uint32_t r = gpro_rng_draw(key, q);
    double u1 = (double)((r >> 16) + 1u) * (1.0 / 65536.0); /* (0,1] */
    double u2 = (double)(r & 0xFFFFu) * (1.0 / 65536.0);    /* [0,1) */
    double rad = sqrt(-2.0 * log(u1));
    double ang = 6.283185307179586476925286766559 * u2;
    *n0 = rad * cos(ang);
    *n1 = rad * sin(ang);
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
 * the engine's own contact model "gpr-push-v1".
 * ==================================================================================================== */
typedef struct { double x, y, th, vx, vy, w; } body2;

static double cross2(double ax, double ay, double bx, double by) { return ax * by - ay * bx; }

/* One control cycle of nm (1 or 2) movers and the object, model "gpr-push-v2". u[i] = commanded acceleration of mover i (world
 * x,y). Semi-implicit Euler: velocities first (external forces, then contact impulses), then positions with the new velocities.
 * Contacts: mover i <-> object for every mover (the movers do not touch each other: coming within their collision circles
 * ends the episode, as in the planning env). Per pair: broad phase on the circumradii, separating-axis test over the 4 face
 * normals, at most GPRO_PUSH_SLOTS contact points (the first vertices, in the fixed order object vertices 0..3 then mover
 * vertices 0..3, that lie inside the other box), effective masses inverted once; then 4 sweeps of sequential impulses over
 * mover 0's points, mover 1's points. */
#define GPRO_PUSH_SLOTS 4
typedef struct {
    int np;
    double nx, ny;
    double rmx[GPRO_PUSH_SLOTS], rmy[GPRO_PUSH_SLOTS], rox[GPRO_PUSH_SLOTS], roy[GPRO_PUSH_SLOTS];
    double rmn[GPRO_PUSH_SLOTS], ron[GPRO_PUSH_SLOTS], rmt[GPRO_PUSH_SLOTS], rot_[GPRO_PUSH_SLOTS];
    double inv_kn[GPRO_PUSH_SLOTS], inv_kt[GPRO_PUSH_SLOTS], bias[GPRO_PUSH_SLOTS], jn[GPRO_PUSH_SLOTS], jt[GPRO_PUSH_SLOTS];
} push_pair;

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
        m->vx = m->vx + dt * u[i][0];
        m->vy = m->vy + dt * u[i][1];
        m->w = (m->w + dt * (Kr * (0.0 - m->th)) * inv_Im) / (1.0 + dt * Dr * inv_Im);
    }
    /* object: free-joint damping (PU:335, implicit like MuJoCo's Euler integrator), then Coulomb friction with the tile,
     * never reversing the motion */
    double s = 1.0 / (1.0 + dt * pp->object_damping * inv_mo);
    o->vx = o->vx * s; o->vy = o->vy * s;
    o->w = o->w / (1.0 + dt * pp->object_damping * inv_Io);
    {
        double sp = sqrt(o->vx * o->vx + o->vy * o->vy), dv = pp->friction_tile * pp->gravity * dt;
        if (sp <= dv) { o->vx = 0.0; o->vy = 0.0; } else { double k = (sp - dv) / sp; o->vx = o->vx * k; o->vy = o->vy * k; }
        double dw = pp->friction_torsional * pp->object_mass * pp->gravity * inv_Io * dt, aw = fabs(o->w);
        if (aw <= dw) o->w = 0.0; else o->w = o->w * ((aw - dw) / aw);
    }
    /* 2. contact geometry per pair */
    push_pair pr[2];
    const double co = cos(o->th), so = sin(o->th);
    const double reach = sqrt(ax * ax + ay * ay) + h * sqrt(2.0), reach2 = reach * reach;
    const double sx[4] = {-1, -1, 1, 1}, sy[4] = {-1, 1, 1, -1};
    for (int i = 0; i < nm; ++i) {
        const body2 *m = mv + i;
        push_pair *q = pr + i;
        q->np = 0; q->nx = 0.0; q->ny = 0.0;
        const double dx = o->x - m->x, dy = o->y - m->y;
        if (dx * dx + dy * dy > reach2) continue; /* broad phase: the circumscribed circles are apart */
        const double cm = cos(m->th), sm = sin(m->th);
        const double axes[4][2] = {{cm, sm}, {-sm, cm}, {co, so}, {-so, co}};
        double best = INFINITY, nx = 0.0, ny = 0.0;
        int touching = 1;
        for (int k = 0; k < 4; ++k) {
            const double ex = axes[k][0], ey = axes[k][1];
            const double rm = ax * fabs(ex * cm + ey * sm) + ay * fabs(-ex * sm + ey * cm);
            const double ro = h * fabs(ex * co + ey * so) + h * fabs(-ex * so + ey * co);
            const double d = dx * ex + dy * ey;
            const double overlap = rm + ro - fabs(d);
            if (overlap < 0.0) { touching = 0; break; }
            if (overlap < best) { best = overlap; nx = d < 0.0 ? -ex : ex; ny = d < 0.0 ? -ey : ey; }
        }
        if (!touching) continue;
        /* contact points: object vertices inside the mover, then mover vertices inside the object; depth along n. If the boxes
         * cross without a vertex inside (rare), one point midway between the centres with the SAT depth. */
        double cx[GPRO_PUSH_SLOTS], cy[GPRO_PUSH_SLOTS], dep[GPRO_PUSH_SLOTS];
        int np = 0;
        const double rm_n = ax * fabs(nx * cm + ny * sm) + ay * fabs(-nx * sm + ny * cm);
        const double ro_n = h * fabs(nx * co + ny * so) + h * fabs(-nx * so + ny * co);
        for (int k = 0; k < 4; ++k) { /* object vertex in the mover frame */
            const double vx = o->x + co * (sx[k] * h) - so * (sy[k] * h), vy = o->y + so * (sx[k] * h) + co * (sy[k] * h);
            const double lx = (vx - m->x) * cm + (vy - m->y) * sm, ly = -(vx - m->x) * sm + (vy - m->y) * cm;
            if (fabs(lx) <= ax && fabs(ly) <= ay && np < GPRO_PUSH_SLOTS) {
                cx[np] = vx; cy[np] = vy; dep[np] = rm_n - ((vx - m->x) * nx + (vy - m->y) * ny); ++np;
            }
        }
        for (int k = 0; k < 4; ++k) { /* mover vertex in the object frame */
            const double vx = m->x + cm * (sx[k] * ax) - sm * (sy[k] * ay), vy = m->y + sm * (sx[k] * ax) + cm * (sy[k] * ay);
            const double lx = (vx - o->x) * co + (vy - o->y) * so, ly = -(vx - o->x) * so + (vy - o->y) * co;
            if (fabs(lx) <= h && fabs(ly) <= h && np < GPRO_PUSH_SLOTS) {
                cx[np] = vx; cy[np] = vy; dep[np] = ro_n + ((vx - o->x) * nx + (vy - o->y) * ny); ++np;
            }
        }
        if (np == 0) { cx[0] = m->x + 0.5 * dx; cy[0] = m->y + 0.5 * dy; dep[0] = best; np = 1; }
        q->np = np; q->nx = nx; q->ny = ny;
        const double tx = -ny, ty = nx;
        for (int k = 0; k < np; ++k) {
            q->rmx[k] = cx[k] - m->x; q->rmy[k] = cy[k] - m->y; q->rox[k] = cx[k] - o->x; q->roy[k] = cy[k] - o->y;
            q->rmn[k] = cross2(q->rmx[k], q->rmy[k], nx, ny); q->ron[k] = cross2(q->rox[k], q->roy[k], nx, ny);
            q->rmt[k] = cross2(q->rmx[k], q->rmy[k], tx, ty); q->rot_[k] = cross2(q->rox[k], q->roy[k], tx, ty);
            q->inv_kn[k] = 1.0 / (inv_mm + inv_mo + q->rmn[k] * q->rmn[k] * inv_Im + q->ron[k] * q->ron[k] * inv_Io);
            q->inv_kt[k] = 1.0 / (inv_mm + inv_mo + q->rmt[k] * q->rmt[k] * inv_Im + q->rot_[k] * q->rot_[k] * inv_Io);
            q->bias[k] = (dep[k] > 0.0 ? dep[k] : 0.0) / pp->contact_timeconst; /* penetration recovery velocity */
            q->jn[k] = 0.0; q->jt[k] = 0.0;
        }
    }
    /* 3. sequential impulses: 4 sweeps over mover 0's points, mover 1's points */
    const double mu = pp->friction_contact;
    for (int it = 0; it < 4; ++it) {
        for (int i = 0; i < nm; ++i) {
            body2 *m = mv + i;
            push_pair *q = pr + i;
            const double nx = q->nx, ny = q->ny, tx = -ny, ty = nx;
            for (int k = 0; k < q->np; ++k) {
                const double rmx = q->rmx[k], rmy = q->rmy[k], rox = q->rox[k], roy = q->roy[k];
                double rvx = (o->vx - o->w * roy) - (m->vx - m->w * rmy), rvy = (o->vy + o->w * rox) - (m->vy + m->w * rmx);
                double dj = -((rvx * nx + rvy * ny) - q->bias[k]) * q->inv_kn[k];
                double jn_new = q->jn[k] + dj; if (jn_new < 0.0) jn_new = 0.0;
                dj = jn_new - q->jn[k]; q->jn[k] = jn_new;
                m->vx -= dj * nx * inv_mm; m->vy -= dj * ny * inv_mm; m->w -= dj * q->rmn[k] * inv_Im;
                o->vx += dj * nx * inv_mo; o->vy += dj * ny * inv_mo; o->w += dj * q->ron[k] * inv_Io;
                /* Coulomb friction at the point */
                rvx = (o->vx - o->w * roy) - (m->vx - m->w * rmy); rvy = (o->vy + o->w * rox) - (m->vy + m->w * rmx);
                double djt = -(rvx * tx + rvy * ty) * q->inv_kt[k];
                double lim = mu * q->jn[k], jt_new = q->jt[k] + djt;
                if (jt_new > lim) jt_new = lim;
                if (jt_new < -lim) jt_new = -lim;
                djt = jt_new - q->jt[k]; q->jt[k] = jt_new;
                m->vx -= djt * tx * inv_mm; m->vy -= djt * ty * inv_mm; m->w -= djt * q->rmt[k] * inv_Im;
                o->vx += djt * tx * inv_mo; o->vy += djt * ty * inv_mo; o->w += djt * q->rot_[k] * inv_Io;
            }
        }
    }
    /* 4. positions with the new velocities */
    for (int i = 0; i < nm; ++i) { body2 *m = mv + i; m->x += dt * m->vx; m->y += dt * m->vy; m->th += dt * m->w; }
    o->x += dt * o->vx; o->y += dt * o->vy; o->th += dt * o->w;
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
            push_cycle(cfg, pp, nm, m, &o, (const double (*)[2])u); /* mj_step (BE:1882): engine model gpr-push-v2 */
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
