// gpr_push_kernels.hip -- fused BenchmarkPushingEnv step / reset kernels: NM = 1 (the reference's env, PU:196) or 2 movers
// (BASELINE.json's config 3; engine-defined) + 1 pushed object.
//
// LANES PER ENV. With one lane per env a batch of 4 096 envs is 64 wavefronts on 1 024 SIMDs, and every wavefront walks through the
// whole contact path in practically every cycle (27 % of the envs are within reach of the object at any time, 4 % touch: of 64 envs
// some env always is). So a launch gives every env 2^lane_shift adjacent lanes -- as many as keep the grid within ONE wavefront per
// SIMD (launch_push3: 16 lanes per env at 4 096 envs, 4 at 16 384, 1 from 65 536 on). All lanes of an env's group run the env
// redundantly (same loads, same arithmetic, same votes: nothing to exchange, results bit-identical to one lane per env); the group's
// first lane stores. A wavefront then holds 4-16 envs instead of 64 and its wave-level decisions "is any env within reach / touching"
// fail more often. Measured (scratch/pushlanes.py, MI355X, us per step): 4 096 envs x 2 movers 147 -> 128, x 1 mover 63.8 -> 61.6; f64
// 393 -> 348 / 207 -> 197. What it does NOT buy is a shorter kernel than its slowest env: an env that pushes the object through all 40
// cycles runs the full contact path 40 times whatever shares its wavefront, and that dependent chain (~600 instructions per cycle at
// ~6 clocks each for a lone wavefront) is the kernel's duration at every batch size up to one wavefront per SIMD. More wavefronts per
// SIMD do not hide it, they share the SIMD: 2 per SIMD +30 %, 4 per SIMD +85 % (same script).
// The two-mover float kernel does better than redundancy with its lanes: k_push<..., PAIR = true> gives each (mover, object) pair a lane
// of its own (push_cycle_pair below), taken while the grid stays within two wavefronts per SIMD.
//
// Pinned by the reference: control + clamp (PU:419-448), wall check (BE:426-788), obs / info / reward / terminated (PU:457-607),
// reset sampling (PU:373-417). The contact physics is the engine's own planar model "gpr-push-v3" (parity unpinned: the
// reference hands it to MuJoCo) -- see include/gpr.h and DESIGN.md section 9: ONE templated implementation (push_cycle3) for every
// precision, oracle/gpr_oracle.c:push_cycle operation for operation (1e-9 on every state in the double builds' tests).
#include <cstdlib>

#include "gpr.h"
#include "gpr_device.h"
#include "gpr_host.h"

namespace gpr {

template <typename RT> struct Body { RT x, y, th, vx, vy, w; };

// ---------------------------------------------------------------------------------------------------------------------
// Engine contact model "gpr-push-v3" -- oracle/gpr_oracle.c (push3_geometry, push3_lcp2, push3_visit, push_cycle) operation for
// operation, ONE templated implementation for every precision: the fused operations are exactly the oracle's fma()s, divisions and
// roots go through Math<RT> (correctly rounded in double; reciprocal / rsqrt instructions in float), sin / cos through the device's.
// Per (mover, object) pair, in the MOVER's frame: broad phase, separating axes (mover x, mover y, object x, object y; the first smallest
// overlap wins) -> contact normal + reference / incident box; the incident face's segment cut to the reference face's lateral extent
// -> at most TWO contact points with depths; two normal rows + ONE friction row at the points' midpoint; block Gauss-Seidel on the
// cycle's TOTAL impulses: a visit solves the pair's 2 x 2 normal block exactly (enumerated active sets) with the friction impulse held,
// then clamps the friction impulse to mu (l0 + l1); two visits per pair and cycle. (v2: up to four "vertex inside the other box" points,
// a friction row per point, four sweeps of scalar sequential impulses: ~600 instructions per contact cycle; v3: ~350.)
// Everything lives in registers; the contact block of a pair is entered only by wavefronts in which some env's circumscribed circles
// overlap, the solve only by wavefronts with an env in contact.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float csign_(float mag, float sgn) { return __builtin_copysignf(mag, sgn); }
__device__ __forceinline__ double csign_(double mag, double sgn) { return __builtin_copysign(mag, sgn); }
__device__ __forceinline__ float min_(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double min_(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float max_(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double max_(double a, double b) { return __builtin_fmax(a, b); }

constexpr double PUSH3_MARGIN = 1e-4;  // contact margin [m]: closer boxes touch; an incident end up to the margin above the reference face is a speculative row
template <typename RT> struct P3Pair {
    int np;                        // 0, 1, 2 contact points of this env's pair
    RT nx, ny;                     // contact normal, mover -> object, world frame
    RT rmn0, rmn1, ron0, ron1;     // lever arms x n of the two normal rows (mover's, object's)
    RT rmt, rot_;                  // lever arms x t of the friction row, t = (-ny, nx)
    RT bias0, bias1;
    RT a00, a01, a11, att, ant0, ant1;  // A = J M^-1 J^T of the three rows
    RT i00, i11, idet, itt;        // 1 / a00, 1 / a11, 1 / (a00 a11 - a01^2), 1 / att
    RT l0, l1, jt;                 // total impulses of the cycle so far
};

// 1. external forces: a mover's actuation and yaw impedance (PU:305-319, IMP:42-47, PU:261-267; damping implicit); the object's joint
// damping (PU:335, implicit) and Coulomb + torsional friction against the tile, never reversing the motion
template <typename RT> __device__ __forceinline__ void p3_forces_mover(const PushCst<RT> &Q, RT dt, Body<RT> &m, R2<RT> u) {
    m.vx = fma_(dt, u.x, m.vx);
    m.vy = fma_(dt, u.y, m.vy);
    m.w = Math<RT>::div_(fma_(-(dt * Q.Kr * Q.inv_Im), m.th, m.w), (RT)1 + dt * Q.Dr * Q.inv_Im);
}
template <typename RT> __device__ __forceinline__ void p3_forces_object(const PushCst<RT> &Q, Body<RT> &o) {
    const RT vx = o.vx * Q.damp_lin, vy = o.vy * Q.damp_lin, w = Math<RT>::div_(o.w, Q.damp_rot_den);
    const RT sp2 = fma_(vx, vx, vy * vy);
    RT kf;
    if constexpr (Math<RT>::literal) {
        const RT sp = Math<RT>::sqrt_(sp2);
        kf = sp > Q.fric_dv ? Math<RT>::div_(sp - Q.fric_dv, sp) : (RT)0;
    } else {
        kf = sp2 > Q.fric_dv * Q.fric_dv ? fma_(-Q.fric_dv, Math<RT>::rsqrt_(sp2), (RT)1) : (RT)0;  // (sp - dv) / sp
    }
    o.vx = vx * kf; o.vy = vy * kf;
    const RT aw = absr(w);
    o.w = aw <= Q.fric_dw ? (RT)0 : csign_(aw - Q.fric_dw, w);
}

// 2. contact geometry of the pair (mover m, object o); dx, dy = object centre - mover centre (world). Returns whether some env of the
// wavefront is within reach (wave-uniform); q.np == 0 for the lanes without a contact point (their rows hold finite junk that nothing uses).
template <typename RT>
__device__ __forceinline__ bool p3_geometry(const PushCst<RT> &Q, RT dt, const Body<RT> &m, const Body<RT> &o, RT dx, RT dy, P3Pair<RT> &q) {
    q.np = 0; q.nx = (RT)0; q.ny = (RT)0; q.l0 = (RT)0; q.l1 = (RT)0; q.jt = (RT)0;
    // (a pair whose block is left early keeps finite rows: with two movers the other pair's contact sends every lane through the visits,
    //  where this pair's impulses are selected to zero and multiplied with its rows)
    q.rmn0 = q.rmn1 = q.ron0 = q.ron1 = q.rmt = q.rot_ = q.bias0 = q.bias1 = (RT)0;
    q.a00 = q.a11 = q.att = (RT)1; q.a01 = q.ant0 = q.ant1 = (RT)0; q.i00 = q.i11 = q.idet = q.itt = (RT)1;
    const bool in_range = !(fma_(dx, dx, dy * dy) > Q.reach2);  // broad phase: the circumscribed circles overlap
    if (__ballot(in_range) == 0ull) return false;
    RT sm, cm, so, co;
    Math<RT>::sincos_(m.th, sm, cm);
    Math<RT>::sincos_(o.th, so, co);
    const RT c = fma_(co, cm, so * sm), s = fma_(so, cm, -(co * sm));    // the object's axes in the mover's frame: (c, s), (-s, c)
    const RT lx = fma_(dx, cm, dy * sm), ly = fma_(dy, cm, -(dx * sm));  // the object's centre in the mover's frame
    const RT ac = absr(c), as = absr(s);
    const RT ro01 = Q.h * (ac + as);
    const RT d2 = fma_(lx, c, ly * s), d3 = fma_(ly, c, -(lx * s));
    const RT ov0 = (Q.ax + ro01) - absr(lx), ov1 = (Q.ay + ro01) - absr(ly);
    const RT ov2 = (fma_(Q.ax, ac, Q.ay * as) + Q.h) - absr(d2), ov3 = (fma_(Q.ax, as, Q.ay * ac) + Q.h) - absr(d3);
    // (ONE wave-level decision per pair and cycle, the broad phase above: a decision that hangs on a vector compare stalls its wavefront for
    //  40-60 clocks, and the wavefront that sets the kernel's duration -- an env pushing through all 40 cycles -- never takes the early exits)
    bool touch = in_range & !(min_(min_(ov0, ov1), min_(ov2, ov3)) < (RT)-PUSH3_MARGIN);
    int k = 0;
    RT best = ov0;
    { const bool t = ov1 < best; best = t ? ov1 : best; k = t ? 1 : k; }
    { const bool t = ov2 < best; best = t ? ov2 : best; k = t ? 2 : k; }
    { const bool t = ov3 < best; best = t ? ov3 : best; k = t ? 3 : k; }
    // THE MANIFOLD IN A CANONICAL FRAME. The oracle states it for a general reference box (centre, normal, tangent: oracle/gpr_oracle.c
    // push3_geometry); here the same points are found in the REFERENCE box's own frame, turned so that the reference normal is +x: the
    // reference box A is axis-aligned with half extents (en, et), the incident box B sits at (pn > 0, pt) under the rotation (cc, ss) with
    // half extents (bu, bv). Getting there takes selects only -- which box is A, a swap of the coordinates when the axis is a "y" axis, a
    // mirror image when B lies on the negative side -- and the two reflections flip the sign of every cross product (`odd`); what was a
    // dot product with a general normal and tangent is now a coordinate (~70 fewer instructions per contact cycle than the general form).
    const bool mref = k < 2, kx = (k == 0) | (k == 2);  // the mover's face is the reference face; the axis is an "x" axis of its box
    // A = reference box: the mover (frame: the mover's) or the object (frame: the object's; the mover sits at -(d2, d3) there, turned by -yaw)
    const RT qx_ = mref ? lx : -d2, qy_ = mref ? ly : -d3;
    const RT srot = mref ? s : -s;
    const RT eAx = mref ? Q.ax : Q.h, eAy = mref ? Q.ay : Q.h, eBu = mref ? Q.h : Q.ax, eBv = mref ? Q.h : Q.ay;
    // a "y" axis: exchange the coordinates (B's axes exchange their roles: rotation (c, -s), extents exchanged)
    const RT pn_s = kx ? qx_ : qy_, pt = kx ? qy_ : qx_;
    const RT en = kx ? eAx : eAy, et = kx ? eAy : eAx;
    const RT bu = kx ? eBu : eBv, bv = kx ? eBv : eBu;
    // B on the negative side: mirror image (rotation (c, -s) again)
    const bool neg = pn_s < (RT)0;
    const RT pn = absr(pn_s);
    const RT ss = (kx != neg) ? srot : -srot, cc = c;
    const bool odd = (!kx) != neg;  // parity of the reflections
    // incident face: the one whose outward normal is most opposed to +x; B's axes: u = (cc, ss), v = (-ss, cc)
    const bool fu = absr(cc) >= absr(ss);
    const RT sgn = fu ? -csign_((RT)1, cc) : csign_((RT)1, ss);
    const RT hf = fu ? sgn * bu : sgn * bv, he = fu ? bv : bu;            // signed half extent along the face normal, half extent along the edge
    const RT fx = fu ? hf * cc : -(hf * ss), fy = fu ? hf * ss : hf * cc; // face centre offset from B's centre
    const RT ex = fu ? -(he * ss) : he * cc, ey = fu ? he * cc : he * ss; // half edge vector
    const RT bx = pn + fx, by = pt + fy;                                  // face centre relative to A's centre
    const RT pax = bx - ex, sa = by - ey, pbx = bx + ex, sb = by + ey;    // the incident segment's ends: (x, lateral coordinate)
    const RT da = en - pax, db = en - pbx;                                // depths below the reference face
    touch = touch & !((min_(sa, sb) > et) | (max_(sa, sb) < -et));        // the segment passes the face by
    const RT inv = Math<RT>::div_((RT)1, sb - sa);
    const RT ta = (min_(max_(sa, -et), et) - sa) * inv, tb = (min_(max_(sb, -et), et) - sa) * inv;  // cut to [-et, et] along the segment
    const RT ex2 = pbx - pax, ey2 = sb - sa, dd = db - da;
    const RT qx0 = fma_(ta, ex2, pax), qy0 = fma_(ta, ey2, sa), dp0 = fma_(ta, dd, da);
    const RT qx1 = fma_(tb, ex2, pax), qy1 = fma_(tb, ey2, sa), dp1 = fma_(tb, dd, da);
    // points with depth >= -margin, the deeper one first
    const bool sw = dp1 > dp0;
    const RT e0 = sw ? dp1 : dp0, e1 = sw ? dp0 : dp1;
    touch = touch & !(e0 < (RT)-PUSH3_MARGIN);
    bool two = e1 >= (RT)-PUSH3_MARGIN;
    const RT p0x = sw ? qx1 : qx0, p0y = sw ? qy1 : qy0;
    const RT p1x = two ? (sw ? qx0 : qx1) : p0x, p1y = two ? (sw ? qy0 : qy1) : p0y;
    const RT d0 = e0, d1 = two ? e1 : (RT)0;
    // rows. In the canonical frame the reference normal is +x: r x (+x) = -r.y, r . (+x) = r.x, for the lever arms r = point (from A's
    // centre) and point - (pn, pt) (from B's). The contact normal n points from the mover to the object: +x if the mover is A, -x if it is
    // B; an odd number of reflections flips the sign of the cross products once more.
    const RT mxc = (RT)0.5 * (p0x + p1x);                                  // friction row: midpoint of the points (only its x matters)
    const bool flip = mref != odd;                                         // r x n = -y (mover is A, even) ... = +y
    const RT y0a = flip ? -p0y : p0y, y1a = flip ? -p1y : p1y;             // cross products of A's lever arms with n
    const RT y0b = flip ? pt - p0y : p0y - pt, y1b = flip ? pt - p1y : p1y - pt;  // ... of B's
    q.rmn0 = mref ? y0a : y0b; q.ron0 = mref ? y0b : y0a;
    q.rmn1 = mref ? y1a : y1b; q.ron1 = mref ? y1b : y1a;
    q.rmt = mref ? mxc : pn - mxc; q.rot_ = mref ? mxc - pn : -mxc;
    const RT imm = Q.inv_mm + Q.inv_mo;
    const RT wm0 = q.rmn0 * Q.inv_Im, wo0 = q.ron0 * Q.inv_Io, wm1 = q.rmn1 * Q.inv_Im, wo1 = q.ron1 * Q.inv_Io;
    const RT wmt = q.rmt * Q.inv_Im, wot = q.rot_ * Q.inv_Io;
    q.a00 = fma_(q.rmn0, wm0, fma_(q.ron0, wo0, imm));
    q.a11 = fma_(q.rmn1, wm1, fma_(q.ron1, wo1, imm));
    q.a01 = fma_(q.rmn0, wm1, fma_(q.ron0, wo1, imm));
    q.att = fma_(q.rmt, wmt, fma_(q.rot_, wot, imm));
    q.ant0 = fma_(q.rmn0, wmt, q.ron0 * wot);
    q.ant1 = fma_(q.rmn1, wmt, q.ron1 * wot);
    // two points that (nearly) coincide make the block singular: the deeper one stands for both
    const RT det = fma_(q.a00, q.a11, -(q.a01 * q.a01));
    two = two & (det > (RT)1e-4 * (q.a00 * q.a11));
    q.i00 = Math<RT>::div_((RT)1, q.a00); q.i11 = Math<RT>::div_((RT)1, q.a11); q.idet = Math<RT>::div_((RT)1, det); q.itt = Math<RT>::div_((RT)1, q.att);
    q.bias0 = Math<RT>::div_(d0, d0 >= (RT)0 ? Q.tc : dt);  // recovery; speculative: the gap may close within the cycle, not more
    q.bias1 = Math<RT>::div_(d1, d1 >= (RT)0 ? Q.tc : dt);
    // the normal in the world: the reference box's axis, signed towards the object
    const RT Cw = mref ? cm : co, Sw = mref ? sm : so;
    const RT sN = (mref != neg) ? (RT)1 : (RT)-1;                           // (+1 for A = mover and B on the positive side; A = object: reversed)
    q.nx = sN * (kx ? Cw : -Sw); q.ny = sN * (kx ? Sw : Cw);
    q.np = touch ? (two ? 2 : 1) : 0;
    return true;
}

// the row velocities of a pair from the bodies: relative velocity of the object against the mover along n at the two points, along t
// at the friction row
template <typename RT>
__device__ __forceinline__ void p3_row_velocities(const P3Pair<RT> &q, const Body<RT> &m, const Body<RT> &o, RT &vn0, RT &vn1, RT &vt) {
    const RT dvx = o.vx - m.vx, dvy = o.vy - m.vy;
    const RT dvn = fma_(dvx, q.nx, dvy * q.ny), dvt = fma_(dvy, q.nx, -(dvx * q.ny));
    vn0 = fma_(o.w, q.ron0, fma_(-m.w, q.rmn0, dvn)); vn1 = fma_(o.w, q.ron1, fma_(-m.w, q.rmn1, dvn));
    vt = fma_(o.w, q.rot_, fma_(-m.w, q.rmt, dvt));
}

// 3. one visit of a pair: the 2 x 2 normal block exactly (A l + b >= 0, l >= 0, complementary: A is symmetric positive definite, exactly
// one active set is consistent) with the friction impulse held, then the friction impulse for the new normals. (vn0, vn1, vt) = the row
// velocities WITHOUT this pair's own impulses.
template <typename RT> __device__ __forceinline__ void p3_visit(const PushCst<RT> &Q, P3Pair<RT> &q, RT vn0, RT vn1, RT vt) {
    const RT b0 = fma_(q.ant0, q.jt, vn0 - q.bias0), b1 = fma_(q.ant1, q.jt, vn1 - q.bias1);
    const bool two = q.np == 2;
    const RT s0 = max_(-b0 * q.i00, (RT)0);  // only row 0 active
    const RT x0 = fma_(q.a01, b1, -(q.a11 * b0)) * q.idet, x1 = fma_(q.a01, b0, -(q.a00 * b1)) * q.idet;  // both rows active
    const RT s1 = max_(-b1 * q.i11, (RT)0);  // only row 1 active
    const bool both = two & (x0 >= (RT)0) & (x1 >= (RT)0);
    const bool first = !both & (!two | (fma_(q.a01, s0, b1) >= (RT)0));
    q.l0 = both ? x0 : first ? s0 : (RT)0;
    q.l1 = both ? x1 : first ? (RT)0 : s1;
    const RT lim = Q.mu * (q.l0 + q.l1);
    const RT r = fma_(q.ant1, q.l1, fma_(q.ant0, q.l0, vt));
    q.jt = min_(max_(-r * q.itt, -lim), lim);
}

// the bodies take the impulses (d0, d1, djt) of a pair's rows; a lane whose pair has no contact takes nothing
template <typename RT>
__device__ __forceinline__ void p3_apply(const PushCst<RT> &Q, const P3Pair<RT> &q, Body<RT> &m, Body<RT> &o, RT d0, RT d1, RT djt, bool mine) {
    d0 = mine ? d0 : (RT)0; d1 = mine ? d1 : (RT)0; djt = mine ? djt : (RT)0;
    const RT djn = d0 + d1;
    const RT px = fma_(djn, q.nx, -(djt * q.ny)), py = fma_(djn, q.ny, djt * q.nx);  // impulse on the object; t = (-ny, nx)
    m.vx = fma_(-px, Q.inv_mm, m.vx); m.vy = fma_(-py, Q.inv_mm, m.vy);
    o.vx = fma_(px, Q.inv_mo, o.vx); o.vy = fma_(py, Q.inv_mo, o.vy);
    m.w -= fma_(d1, q.rmn1, fma_(d0, q.rmn0, djt * q.rmt)) * Q.inv_Im;
    o.w += fma_(d1, q.ron1, fma_(d0, q.ron0, djt * q.rot_)) * Q.inv_Io;
}

// one pair alone (one mover; or pair lanes when the partner's pair does not touch): both visits on the impulses -- the row velocities are
// affine in them -- and the bodies take the result once
template <typename RT> __device__ __forceinline__ void p3_solve_alone(const PushCst<RT> &Q, P3Pair<RT> &q, Body<RT> &m, Body<RT> &o) {
    RT vn0, vn1, vt;
    p3_row_velocities(q, m, o, vn0, vn1, vt);
    p3_visit(Q, q, vn0, vn1, vt);
    p3_visit(Q, q, vn0, vn1, vt);
    p3_apply(Q, q, m, o, q.l0, q.l1, q.jt, q.np > 0);
}

// one visit of a pair among several: the rows' velocities now, minus this pair's own impulses so far; the bodies take the change
template <typename RT> __device__ __forceinline__ void p3_visit_bodies(const PushCst<RT> &Q, P3Pair<RT> &q, Body<RT> &m, Body<RT> &o, bool mine) {
    RT vn0, vn1, vt;
    p3_row_velocities(q, m, o, vn0, vn1, vt);
    vn0 -= fma_(q.ant0, q.jt, fma_(q.a01, q.l1, q.a00 * q.l0));
    vn1 -= fma_(q.ant1, q.jt, fma_(q.a11, q.l1, q.a01 * q.l0));
    vt -= fma_(q.att, q.jt, fma_(q.ant1, q.l1, q.ant0 * q.l0));
    const RT l0o = q.l0, l1o = q.l1, jto = q.jt;
    p3_visit(Q, q, vn0, vn1, vt);
    p3_apply(Q, q, m, o, q.l0 - l0o, q.l1 - l1o, q.jt - jto, mine & (q.np > 0));
    if (!(mine & (q.np > 0))) { q.l0 = l0o; q.l1 = l1o; q.jt = jto; }
}

// one control cycle of the NM movers and the object. rel0 (float builds): object - mover at the start of the step, formed in double --
// the bodies' x, y are then DISPLACEMENTS since the start of the step (a manifold changes discretely wherever a rounding error decides;
// summing 1e-3-sized increments onto 0.4 m positions in float put 3e-8 m into every cycle); the double builds pass zeros and absolute
// positions.
template <typename RT, int NM>
__device__ __forceinline__ void push_cycle3(const PushCst<RT> &Q, RT dt, Body<RT> (&mv)[NM], Body<RT> &o, const R2<RT> (&u)[NM], const R2<RT> (&rel0)[NM]) {
#pragma unroll
    for (int i = 0; i < NM; ++i) p3_forces_mover(Q, dt, mv[i], u[i]);
    p3_forces_object(Q, o);
    P3Pair<RT> pr[NM];
    bool any = false;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        RT dx = o.x - mv[i].x, dy = o.y - mv[i].y;
        if constexpr (!Math<RT>::literal) { dx = rel0[i].x + dx; dy = rel0[i].y + dy; }
        any |= p3_geometry(Q, dt, mv[i], o, dx, dy, pr[i]);
    }
    if (any) {  // (wave-uniform) block Gauss-Seidel on the cycle's total impulses: two visits per pair, pair 0, pair 1, pair 0, pair 1
        if constexpr (NM == 1) {
            p3_solve_alone(Q, pr[0], mv[0], o);
        } else {
            for (int it = 0; it < 2; ++it) {
#pragma unroll
                for (int i = 0; i < NM; ++i) p3_visit_bodies(Q, pr[i], mv[i], o, true);
            }
        }
    }
    // positions with the new velocities
#pragma unroll
    for (int i = 0; i < NM; ++i) { Body<RT> &m = mv[i]; m.x = fma_(dt, m.vx, m.x); m.y = fma_(dt, m.vy, m.y); m.th = fma_(dt, m.w, m.th); }
    o.x = fma_(dt, o.vx, o.x); o.y = fma_(dt, o.vy, o.y); o.th = fma_(dt, o.w, o.th);
}
template <typename RT, int NM>
__device__ __forceinline__ void push_cycle(const PushCst<RT> &Q, RT dt, Body<RT> (&mv)[NM], Body<RT> &o, const R2<RT> (&u)[NM]) {
    R2<RT> none[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) none[i] = R2<RT>{(RT)0, (RT)0};
    push_cycle3<RT, NM>(Q, dt, mv, o, u, none);
}
template <int NM>
__device__ __forceinline__ void push_cycle_fast(const PushCst<float> &Q, float dt, Body<float> (&mv)[NM], Body<float> &o, const R2<float> (&u)[NM],
                                                const R2<float> (&rel0)[NM]) {
    push_cycle3<float, NM>(Q, dt, mv, o, u, rel0);
}

// the value the other lane of this lane's couple holds (lanes 2k, 2k + 1): quad_perm [1,0,3,2]
__device__ __forceinline__ uint32_t partner_u32(uint32_t v) { return rot_u32<2, 1>(v); }
template <typename RT> __device__ __forceinline__ RT partner(RT v) { return rot<2, 1>(v); }

// TWO MOVERS, ONE LANE PER PAIR (k_push<..., PAIR = true>, float). The pairs' geometry blocks are where a two-mover cycle spends its time,
// and in one lane they run one after the other. Here an env's lanes come in couples: the even lane owns mover 0 and the pair (mover 0,
// object), the odd lane mover 1 and its pair; both carry the object. Arithmetic and order are those of push_cycle3<float, 2>: the visits
// pair 0, pair 1, pair 0, pair 1 on the bodies, the object's velocity handed from the lane that just visited to its partner (three
// quad_perm moves per visit). An env in which only ONE pair touches -- the usual push -- would run empty visits for the other: when no env
// of the wavefront has both pairs in contact the touching pair's lane solves alone on the impulses (the same two visits: with the other
// pair idle nothing changes the rows' free velocities in between) and hands the object over once.
__device__ __forceinline__ void push_cycle_pair(const PushCst<float> &Q, float dt, Body<float> &m, Body<float> &o, R2<float> u, R2<float> rel0, bool second) {
    p3_forces_mover(Q, dt, m, u);
    p3_forces_object(Q, o);
    P3Pair<float> q;
    if (p3_geometry(Q, dt, m, o, rel0.x + (o.x - m.x), rel0.y + (o.y - m.y), q)) {  // (wave-uniform: some lane of either pair touches)
        const bool other = partner_u32((uint32_t)q.np) != 0u;                   // the partner's pair touches
        auto from_partner = [&](bool take) {
            const float vx = partner(o.vx), vy = partner(o.vy), w = partner(o.w);
            o.vx = take ? vx : o.vx; o.vy = take ? vy : o.vy; o.w = take ? w : o.w;
        };
        if (__ballot((q.np > 0) & other) == 0ull) {
            p3_solve_alone(Q, q, m, o);
            from_partner((q.np == 0) & other);
        } else {
            for (int it = 0; it < 2; ++it) {
                p3_visit_bodies(Q, q, m, o, !second);
                from_partner(second);
                p3_visit_bodies(Q, q, m, o, second);
                from_partner(!second);
            }
        }
    }
    m.x = fma_(dt, m.vx, m.x); m.y = fma_(dt, m.vy, m.y); m.th = fma_(dt, m.w, m.th);
    o.x = fma_(dt, o.vx, o.x); o.y = fma_(dt, o.vy, o.y); o.th = fma_(dt, o.w, o.th);
}

template <typename RT, int NM> struct PushObs {
    R2<RT> mpos[NM], mvel[NM], ag;
    RT reward;
    bool terminated, is_success;
};

// _get_obs (PU:529-576), _get_info (PU:578-607), compute_reward / compute_terminated (PU:457-527)
template <typename ST, typename RT, int NM, bool NOISE>
__device__ __forceinline__ PushObs<RT, NM> push_observe(const KParams &P, const PushCst<RT> &Q, const uint32_t (&key)[NM], uint32_t q0,
                                                        const R2<RT> (&p)[NM], const R2<RT> (&v)[NM], RT ox, RT oy, R2<RT> g, bool wc, bool mc) {
    const Cst<RT> &C = cst<RT>(P);
    PushObs<RT, NM> o;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        o.mpos[i] = noisy<RT, NOISE>(p[i], C.sig_p, key[i], q0);
        o.mvel[i] = noisy<RT, NOISE>(v[i], C.sig_v, key[i], q0 + 1u);
    }
    o.ag = R2<RT>{ox, oy};
    if (Q.obj_noise != (RT)0) { const R2<RT> n = rng_normal_pair<RT>(key[0], q0 + 2u); o.ag.x = ox + Q.obj_noise * n.x; o.ag.y = oy + Q.obj_noise * n.y; }
    const RT dx = stored<ST, RT>(o.ag.x) - g.x, dy = stored<ST, RT>(o.ag.y) - g.y;
    const bool reached = Math<RT>::sqrt_(dx * dx + dy * dy) <= C.thr;  // PU:520
    const bool coll = wc | mc;               // (2 movers: coming within each other's collision circle counts like a wall)
    RT r = (RT)-50.0 * (RT)coll;             // PU:521
    r += (RT)-1.0 * (RT)(!coll);             // PU:522
    if (reached & !coll) r = (RT)0;          // PU:523
    o.reward = r;
    o.terminated = r == (RT)-50.0;           // PU:475
    o.is_success = reached & !coll;          // PU:602
    return o;
}

// A noisy read behind a wave-uniform "can it matter at all?" test must stay a BRANCH: left alone the compiler turns the test into selects and
// every cycle pays the Box-Muller draw (hash, log, sqrt, sin, cos: ~28 instructions, twice per cycle) that one cycle in hundreds needs.
// An empty volatile asm cannot be executed speculatively, so the block it sits in keeps its branch.
#define GPR_KEEP_BRANCH() asm volatile("" ::: "memory")

// the value lane L of the wavefront holds (L wave-uniform)
template <typename RT> __device__ __forceinline__ RT lane_bcast(RT v, int L);
template <> __device__ __forceinline__ float lane_bcast<float>(float v, int L) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), L));
}
template <> __device__ __forceinline__ double lane_bcast<double>(double v, int L) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, L), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), L);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// in-lane mover <-> mover check of the 2-mover env (BE:355-424 for one pair of circles)
template <typename RT> __device__ __forceinline__ bool pair_hit2(const Cst<RT> &C, R2<RT> a, R2<RT> b, RT *d_out = nullptr) {
    const RT dx = a.x - b.x, dy = a.y - b.y;
    if constexpr (Math<RT>::literal) {
        const RT d = Math<RT>::sqrt_(dx * dx + dy * dy);
        if (d_out) *d_out = d;
        return d <= C.max_csum_step;  // BE:409, inclusive
    } else {
        const RT d2 = dx * dx + dy * dy;
        if (d_out) *d_out = d2;
        return d2 <= C.max_csum_step * C.max_csum_step;
    }
}

// PU:373-417 for the envs whose `need` is set (wave-cooperative: every lane must call it). opt_start / opt_goal: the reset options
// 'mover_start_xy_pos' [E][2 NM] / 'object_goal_xy_pos' [E][2] (NULL: sampled).
template <typename ST, typename RT, int NM, bool NOISE, bool FULL>
__device__ __forceinline__ PushObs<RT, NM> push_reset_one(const KParams &P, const PushCst<RT> &Q, const PushPtrs &X, const Tables<RT> &T, int64_t e,
                                                          int64_t env_global, RT c, bool need, bool lead, R2<RT> (&p)[NM], R2<RT> &g, RT &ox, RT &oy,
                                                          bool &wc, bool &mc) {
    const Cst<RT> &C = cst<RT>(P);
    uint32_t ks[NM], kn[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        ks[i] = rng_key(P, KIND_RESET_START, env_global, (uint32_t)i);
        kn[i] = rng_key(P, KIND_RESET_NOISE, env_global, (uint32_t)i);
    }
    const uint32_t kg = rng_key(P, KIND_RESET_GOAL, env_global, 0u);
    if (X.opt_start != nullptr) {
#pragma unroll
        for (int i = 0; i < NM; ++i) p[i] = load2<ST, RT>(X.opt_start, 2 * NM * e + 2 * i);  // PU:389-391
    } else {
        uint32_t kr[NM] = {};
        if constexpr (NM > 1) {
#pragma unroll
            for (int i = 0; i < NM; ++i) kr[i] = rng_key(P, KIND_RESET_START, env_global, (uint32_t)(GPR_MAX_MOVERS + i));
        }
        bool pending = need;
        for (int trial = 0; trial < P.max_reset_trials; ++trial) {
            if (__ballot(pending) == 0ull) break;
            R2<RT> cn[NM];
#pragma unroll
            for (int i = 0; i < NM; ++i) {  // PU:389-391; re-draws (2 movers only) come from a stream of their own
                RT u0 = rng_uniform<RT>(ks[i], 0u), u1 = rng_uniform<RT>(ks[i], 1u);
                if constexpr (NM > 1) {
                    if (trial > 0) { u0 = rng_uniform<RT>(kr[i], 2u * (uint32_t)(trial - 1)); u1 = rng_uniform<RT>(kr[i], 2u * (uint32_t)(trial - 1) + 1u); }
                }
                cn[i].x = stored<ST, RT>(C.min_x + (C.max_x - C.min_x) * u0);
                cn[i].y = stored<ST, RT>(C.min_y + (C.max_y - C.min_y) * u1);
            }
            bool bad = false;
            if constexpr (NM > 1) {
                const RT dx = cn[0].x - cn[1].x, dy = cn[0].y - cn[1].y;
                bad = dist_le<RT>(Math<RT>::literal ? Math<RT>::sqrt_(dx * dx + dy * dy) : dx * dx + dy * dy, C.max_csum_safe);
            }
            if (pending) {
#pragma unroll
                for (int i = 0; i < NM; ++i) p[i] = cn[i];
                pending = bad;
            }
            if constexpr (NM == 1) break;
        }
        if constexpr (NM > 1) {  // trials exhausted: the last (overlapping) pair of starts stands -- counted (GPR_CNT_RESET_EXHAUSTED)
            if (pending & need & lead) diag_add(P, 0, 1u);
        }
    }
    // PU:396-407: object start by rejection, strict distance rule to every mover. Where a mover starts near the centre of the 3x3
    // tiles almost no object position qualifies (the reference then loops for thousands of trials, at the very centre forever), so
    // the search is wave-cooperative: every lane tries its own first 8 trials; the lanes still looking are then served one
    // at a time, all 64 lanes testing 64 consecutive trials of that env, the first valid one in trial order wins -- the same
    // candidate the sequential loop accepts (the draws are indexed by trial). After max_reset_trials the last candidate stands.
    // (X.cand: the caller's candidate stream [n_cand][E][2] instead of the engine's draws, gpr_push_reset_candidates; wave-uniform)
    const int max_trials = X.cand != nullptr ? (X.n_cand < P.max_reset_trials ? X.n_cand : P.max_reset_trials) : P.max_reset_trials;
    auto candidate = [&](uint32_t key, int64_t env, int trial, RT &cx, RT &cy) {
        if (X.cand != nullptr) {
            const R2<RT> cn = load2<ST, RT>(X.cand, ((int64_t)(trial < max_trials ? trial : max_trials - 1) * P.E + env) * 2);
            cx = cn.x; cy = cn.y;
            return;
        }
        cx = stored<ST, RT>(Q.omin_x + (Q.omax_x - Q.omin_x) * rng_uniform<RT>(key, 2u + 2u * (uint32_t)trial));
        cy = stored<ST, RT>(Q.omin_y + (Q.omax_y - Q.omin_y) * rng_uniform<RT>(key, 3u + 2u * (uint32_t)trial));
    };
    auto far_enough = [&](RT cx, RT cy, const R2<RT> (&mp)[NM]) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NM; ++i) { const RT dx = cx - mp[i].x, dy = cy - mp[i].y; ok &= Math<RT>::sqrt_(dx * dx + dy * dy) > Q.min_mo_dist; }
        return ok;
    };
    constexpr int NSELF = 8;
    bool pending = need;
    int n_tried = 0;
    // NO position at all keeps the distance when even the farthest corner of the object's range is within min_mo_dist of a mover (0.75 %
    // of the mover starts on the default layout: the reference loops for ever there): every one of the max_trials candidates then fails
    // and the last one stands -- taken directly (and counted, GPR_CNT_RESET_EXHAUSTED) instead of after 10 000 trials, 156 rounds of the
    // whole wavefront (at 65 536 envs some wavefront paid that in every step: 66 vs 47 us). 1e-6 of slack for the candidates' rounding.
    if (X.cand == nullptr) {
        bool none = false;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const RT fx = max_(absr(Q.omin_x - p[i].x), absr(Q.omax_x - p[i].x)), fy = max_(absr(Q.omin_y - p[i].y), absr(Q.omax_y - p[i].y));
            none |= Math<RT>::sqrt_(fx * fx + fy * fy) <= Q.min_mo_dist * (RT)(1.0 - 1e-6);
        }
        none &= pending;
        if (__ballot(none) != 0ull) {  // (wave-uniform; rare)
            RT cx, cy;
            candidate(ks[0], e, max_trials - 1, cx, cy);
            if (none) { ox = cx; oy = cy; pending = false; n_tried = max_trials; if (lead) diag_add(P, 0, 1u); }
        }
    }
    for (int trial = 0; trial < NSELF && trial < max_trials; ++trial) {
        if (__ballot(pending) == 0ull) break;
        RT cx, cy;
        candidate(ks[0], e, trial, cx, cy);
        if (pending) { ox = cx; oy = cy; pending = !far_enough(cx, cy, p); n_tried = trial + 1; }
    }
    const int gsh = X.lane_shift < 6 ? X.lane_shift : 6;  // lanes of a group inside this wavefront
    for (unsigned long long pm = __ballot(pending & lead); pm != 0ull; pm &= pm - 1ull) {  // (one search per env: its group's first lane)
        const int L = __builtin_ctzll(pm);
        const uint32_t ksL = (uint32_t)__builtin_amdgcn_readlane((int)ks[0], L);
        const int64_t eL = ((int64_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)e >> 32), L) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)e, L);
        R2<RT> mp[NM];
#pragma unroll
        for (int i = 0; i < NM; ++i) { mp[i].x = lane_bcast<RT>(p[i].x, L); mp[i].y = lane_bcast<RT>(p[i].y, L); }
        int found = max_trials - 1;
        bool exhausted = true;
        for (int base = NSELF; base < max_trials; base += 64) {
            const int trial = base + (int)__lane_id();
            RT cx, cy;
            candidate(ksL, eL, trial, cx, cy);
            const unsigned long long okm = __ballot((trial < max_trials) & far_enough(cx, cy, mp));
            if (okm != 0ull) { found = base + __builtin_ctzll(okm); exhausted = false; break; }
        }
        // PU:396-402: the reference goes on looping and warns every 100 trials; here the last candidate stands although it is too close to
        // a mover -- counted (GPR_CNT_RESET_EXHAUSTED; not for a caller-supplied candidate stream, whose last set stands by contract)
        if (exhausted & (X.cand == nullptr) & ((int)__lane_id() == L)) diag_add(P, 0, 1u);
        RT cx, cy;
        candidate(ksL, eL, found, cx, cy);
        if (((int)__lane_id() >> gsh) == (L >> gsh)) { ox = cx; oy = cy; n_tried = found + 1; }  // the env's whole group takes the result
    }
    if ((X.cand != nullptr) & (X.cand_used != nullptr) & need & lead) X.cand_used[e] = n_tried;
    if (X.opt_goal != nullptr) {
        g = load2<ST, RT>(X.opt_goal, 2 * e);  // PU:412-414
    } else {
        g.x = stored<ST, RT>(Q.omin_x + (Q.omax_x - Q.omin_x) * rng_uniform<RT>(kg, 0u));
        g.y = stored<ST, RT>(Q.omin_y + (Q.omax_y - Q.omin_y) * rng_uniform<RT>(kg, 1u));
    }
    const CircleSize<RT> cw_safe = circle_size<RT>(P, (c + C.c_offset_wall) + C.c_offset);
    wc = false;
    R2<RT> pp[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        const R2<RT> pw = noisy<RT, NOISE>(p[i], C.sig_p, kn[i], 0u);
        wc |= !pos_valid_circle<RT, FULL>(T, P, pw.x, pw.y, cw_safe);  // BE:1799-1801
        pp[i] = noisy<RT, NOISE>(p[i], C.sig_p, kn[i], 1u);
    }
    mc = false;
    if constexpr (NM > 1) mc = pair_hit2<RT>(C, pp[0], pp[1]);  // BE:1803-1805
    R2<RT> zero[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) zero[i] = R2<RT>{(RT)0, (RT)0};
    return push_observe<ST, RT, NM, NOISE>(P, Q, kn, 2u, p, zero, ox, oy, g, wc, mc);
}

template <typename ST, typename RT, int NM> __device__ __forceinline__ void push_store(const KParams &P, int64_t e, bool jerk, const PushObs<RT, NM> &o,
                                                                                           const R2<RT> (&acc)[NM], R2<RT> g, void *obs, void *ag, void *dg) {
    const int od = (jerk ? 6 : 4) * NM;  // all positions, all velocities[, all accelerations] (PU:540-545)
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        store2<ST, RT>(obs, e * od + 2 * i, o.mpos[i]);
        store2<ST, RT>(obs, e * od + 2 * NM + 2 * i, o.mvel[i]);
        if (jerk) store2<ST, RT>(obs, e * od + 4 * NM + 2 * i, acc[i]);
    }
    store2<ST, RT>(ag, e * 2, o.ag);
    store2<ST, RT>(dg, e * 2, g);
}

template <typename ST, typename RT, int NM, bool JERK, bool NOISE, bool FULL, bool RESET_ONLY, bool PAIR = false>
__global__ void __launch_bounds__(BLOCK) k_push(const KParams P, const PushCst<RT> Q, const PushPtrs X) {
    static_assert(!PAIR || (NM == 2 && !Math<RT>::literal && !JERK && !RESET_ONLY), "pair lanes: the two-mover float step kernel");
    extern __shared__ __align__(16) unsigned char lds[];
    const Cst<RT> &C = cst<RT>(P);
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int lsh = X.lane_shift;                                   // 2^lsh lanes per env: the group runs the env redundantly
    const bool lead = (t & ((1ll << lsh) - 1ll)) == 0;              // ... and its first lane stores
    const bool live = lead & ((t >> lsh) < P.E);
    const int64_t e = (t >> lsh) < P.E ? (t >> lsh) : P.E - 1;      // (groups past the last env shadow it; never stored)
    const int64_t env_global = P.env_id_offset + e;
    if constexpr (!FULL) stage_tables(lds, P.tables, P.table_bytes);
    const Tables<RT> T = make_tables<RT>(lds, P);
    const RT c = make_tables<RT>(P.tables, P).csz[0];
    R2<RT> p[NM], v[NM], acc[NM], rot[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        p[i] = load2<ST, RT>(P.pos, 2 * NM * e + 2 * i); v[i] = load2<ST, RT>(P.vel, 2 * NM * e + 2 * i);
        acc[i] = load2<ST, RT>(P.acc, 2 * NM * e + 2 * i); rot[i] = load2<ST, RT>(X.mover_rot, 2 * NM * e + 2 * i);
    }
    R2<RT> g = load2<ST, RT>(P.goal, 2 * e);
    const ST *ob = static_cast<const ST *>(X.object) + 6 * e;
    Body<RT> o{(RT)ob[0], (RT)ob[1], (RT)ob[2], (RT)ob[3], (RT)ob[4], (RT)ob[5]};
    int el = P.elapsed[e];
    uint32_t cnt_old = 0u;
    const int64_t cnt_row = t >> 6;
    if ((P.counters != nullptr) & (__lane_id() < 6u)) cnt_old = P.counters[cnt_row * 6 + __lane_id()];

    bool wc = false, mc = false, trunc = false, need, reset_first = false, done = false;
    PushObs<RT, NM> out;
    if constexpr (RESET_ONLY) {
        need = P.mask ? (P.mask[e] != 0) : true;
    } else {
        R2<RT> a[NM];
        const RT lim = JERK ? C.j_max : C.a_max;
        uint32_t key[NM];
        bool outside = false;  // BE:1871-1873: the reference warns before it clips (counted when the caller asked for it, see below)
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            a[i] = load2<ST, RT>(P.action, 2 * NM * e + 2 * i);
            outside |= (absr(a[i].x) > lim) | (absr(a[i].y) > lim);
            a[i].x = fmin(fmax(a[i].x, -lim), lim);
            a[i].y = fmin(fmax(a[i].y, -lim), lim);
            key[i] = rng_key(P, KIND_STEP_NOISE, env_global, (uint32_t)i);
        }
        if (P.count_clipped) {  // (wave-uniform, opt-in)
            const unsigned long long om = __ballot(outside & live);
            if ((om != 0ull) & (__lane_id() == 0u)) diag_add(P, 1, (uint32_t)__popcll(om));
        }
        reset_first = (P.autoreset == 2) & (el < 0);
        const CircleSize<RT> cw_step = circle_size<RT>(P, (c + C.c_offset_wall) + (RT)0 * C.c_offset);
        Body<RT> m[NM];
#pragma unroll
        for (int i = 0; i < NM; ++i) m[i] = Body<RT>{p[i].x, p[i].y, rot[i].x, v[i].x, v[i].y, rot[i].y};
        // throughput (float) mode: the bodies' x, y are DISPLACEMENTS since the start of the step (see push_cycle_fast); base_* = the
        // positions the step starts from, rel0 = object - mover there (formed in double). The double builds keep absolute positions.
        R2<RT> base_m[NM], base_o{(RT)0, (RT)0};
        R2<float> rel0[NM];
#pragma unroll
        for (int i = 0; i < NM; ++i) { base_m[i] = R2<RT>{(RT)0, (RT)0}; rel0[i] = R2<float>{0.f, 0.f}; }
        if constexpr (!Math<RT>::literal) {
            base_o = R2<RT>{o.x, o.y};
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                base_m[i] = R2<RT>{m[i].x, m[i].y};
                rel0[i] = R2<float>{(float)((double)o.x - (double)m[i].x), (float)((double)o.y - (double)m[i].y)};
                m[i].x = (RT)0; m[i].y = (RT)0;
            }
            o.x = (RT)0; o.y = (RT)0;
        }
        auto mover_xy = [&](int i) { return R2<RT>{base_m[i].x + m[i].x, base_m[i].y + m[i].y}; };  // absolute position of mover i
        bool frozen = reset_first;
        const bool eager = P.eager_noise != 0;
        const RT band_pair = (RT)2 * C.band_p;
        const RT pair_thr = Math<RT>::literal ? C.max_csum_step : C.max_csum_step * C.max_csum_step;
        const RT pair_band = Math<RT>::literal ? band_pair : (RT)2 * C.max_csum_step * band_pair + band_pair * band_pair;
        if constexpr (PAIR) {
            // one lane per (mover, object) pair -- see push_cycle_pair: this lane's mover, its action, its noise stream; the object in both
            const bool second = (threadIdx.x & 1u) != 0u;
            auto pick = [&](RT a0, RT a1) { return second ? a1 : a0; };
            Body<RT> mm{pick(m[0].x, m[1].x), pick(m[0].y, m[1].y), pick(m[0].th, m[1].th), pick(m[0].vx, m[1].vx), pick(m[0].vy, m[1].vy), pick(m[0].w, m[1].w)};
            const R2<RT> am{pick(a[0].x, a[1].x), pick(a[0].y, a[1].y)}, basem{pick(base_m[0].x, base_m[1].x), pick(base_m[0].y, base_m[1].y)};
            const R2<float> rel0m{second ? rel0[1].x : rel0[0].x, second ? rel0[1].y : rel0[0].y};
            R2<RT> accm{pick(acc[0].x, acc[1].x), pick(acc[0].y, acc[1].y)};
            const uint32_t keym = second ? key[1] : key[0];
            // ONE wave-level decision per cycle for everything that may need a noisy read (a decision that hangs on a vector compare stalls its
            // wavefront 40-60 clocks): "can the NEXT cycle's speed clamp fire / is the wall check / is the mover check within the noise band?"
            // are voted on together at the cycle's end; the clamp's answer is a wave-uniform flag by the time the next cycle starts.
            // (Taken with this cycle's `frozen`: an env that stops in this cycle may be looked at once more than needed -- a reading that is
            // not needed changes nothing, PL:636-637.)
            auto clamp_near = [&]() -> bool {
                const RT t0 = mm.vx + C.dt * am.x, t1 = mm.vy + C.dt * am.y, vl = C.v_max - C.band_v;
                return !(t0 * t0 + t1 * t1 < vl * vl) | (vl <= (RT)0);
            };
            bool look = eager | (__ballot(clamp_near() & !frozen) != 0ull);
            for (int cyc = 0; cyc < P.num_cycles; ++cyc) {
                // PU:419-448. While |v + dt a| stays below v_max - band_v in every lane, no velocity reading can make the clamp fire
                // (PL:634): the command is the action itself, and neither the reading nor the clamp is evaluated
                R2<RT> vn{mm.vx, mm.vy};
                R2<RT> qaccm = am;
                if (look) {
                    GPR_KEEP_BRANCH();
                    if constexpr (NOISE) vn = noisy<RT, true>(vn, C.sig_v, keym, 3u * (uint32_t)cyc);
                    R2<RT> vnext;
                    qaccm = clamp_dyn<RT>(vn, C.v_max, am, C.dt, C.inv_dt, vnext);
                }
                if (!frozen) {  // (both lanes of an env are frozen together: the exchanges inside always find their partner)
                    push_cycle_pair(Q, C.dt, mm, o, qaccm, rel0m, second);
                    accm = qaccm;
                }
                const R2<RT> pm{basem.x + mm.x, basem.y + mm.y};
                bool wi, unc = false, pnear = false;
                if constexpr (NOISE) wi = !pos_valid_circle<RT, FULL, true>(T, P, pm.x, pm.y, cw_step, C.band_p, &unc);
                else wi = !pos_valid_circle<RT, FULL>(T, P, pm.x, pm.y, cw_step);
                RT d;
                bool hit = pair_hit2<RT>(C, pm, R2<RT>{partner(pm.x), partner(pm.y)}, &d);  // (symmetric in its two points, bit for bit)
                if constexpr (NOISE) pnear = absr(d - pair_thr) <= pair_band;
                const bool cn = clamp_near();
                look = eager;
                if (eager | (__ballot((unc | pnear | cn) & !frozen) != 0ull)) {
                    GPR_KEEP_BRANCH();
                    if constexpr (NOISE) {
                        if (eager | (__ballot(unc & !frozen) != 0ull)) {
                            const R2<RT> pw = noisy<RT, true>(pm, C.sig_p, keym, 3u * (uint32_t)cyc + 1u);
                            wi = !pos_valid_circle<RT, FULL>(T, P, pw.x, pw.y, cw_step);
                        }
                        if (eager | (__ballot(pnear & !frozen) != 0ull)) {
                            const R2<RT> qm = noisy<RT, true>(pm, C.sig_p, keym, 3u * (uint32_t)cyc + 2u);
                            hit = pair_hit2<RT>(C, qm, R2<RT>{partner(qm.x), partner(qm.y)});
                        }
                    }
                    look = eager | (__ballot(cn & !frozen) != 0ull);
                }
                const bool w = wi | (partner_u32(wi ? 1u : 0u) != 0u);
                if (!frozen) { wc = w; mc = hit; frozen = w | hit; }  // BE:1904-1905
            }
            // every lane gets both movers back
            const Body<RT> mo{partner(mm.x), partner(mm.y), partner(mm.th), partner(mm.vx), partner(mm.vy), partner(mm.w)};
            const R2<RT> acco{partner(accm.x), partner(accm.y)};
#define GPR_BACK(f) m[0].f = second ? mo.f : mm.f; m[NM - 1].f = second ? mm.f : mo.f
            GPR_BACK(x); GPR_BACK(y); GPR_BACK(th); GPR_BACK(vx); GPR_BACK(vy); GPR_BACK(w);
#undef GPR_BACK
            acc[0] = R2<RT>{second ? acco.x : accm.x, second ? acco.y : accm.y}; acc[NM - 1] = R2<RT>{second ? accm.x : acco.x, second ? accm.y : acco.y};
        } else {
        if constexpr (NM == 1) {
        // (one mover: its two votes per cycle -- clamp, wall band -- stay where they are; the merged form below measured +5 % here,
        //  -6 % with two movers)
        for (int cyc = 0; cyc < P.num_cycles; ++cyc) {
            // _mujoco_step_callback (PU:419-455): the same clamp chain as the planning env, on a noisy velocity read, per mover
            // lazy noise as in the planning kernels: a reading is drawn only if some lane's comparison could depend on it
            R2<RT> qacc[NM];
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                R2<RT> vn{m[i].vx, m[i].vy};
                // (acceleration actions) while |v + dt a| stays below v_max - band_v in every lane, no velocity reading can make the clamp
                // fire (PL:634): the command is the action itself, and neither the reading nor the clamp is evaluated
                bool nd = true;
                if constexpr (!JERK) {
                    const RT t0 = m[i].vx + C.dt * a[i].x, t1 = m[i].vy + C.dt * a[i].y, vl = C.v_max - C.band_v;
                    nd = !(t0 * t0 + t1 * t1 < vl * vl) | (vl <= (RT)0);
                }
                const bool look = eager | (__ballot(nd & !frozen) != 0ull);
                if constexpr (NOISE) {
                    if (look) { GPR_KEEP_BRANCH(); vn = noisy<RT, true>(vn, C.sig_v, key[i], 3u * (uint32_t)cyc); }
                }
                R2<RT> vnext;
                if constexpr (JERK) {
                    R2<RT> acc_tmp;
                    const R2<RT> j1 = clamp_dyn<RT>(acc[i], C.a_max, a[i], C.dt, C.inv_dt, acc_tmp);
                    const R2<RT> acc_p = clamp_dyn<RT>(vn, C.v_max, acc_tmp, C.dt, C.inv_dt, vnext);
                    const bool diff = (acc_tmp.x != acc_p.x) | (acc_tmp.y != acc_p.y);
                    const R2<RT> uu{diff ? div_rcp<RT>(acc_p.x - acc[i].x, C.dt, C.inv_dt) : j1.x, diff ? div_rcp<RT>(acc_p.y - acc[i].y, C.dt, C.inv_dt) : j1.y};
                    qacc[i].x = acc[i].x + C.dt * uu.x; qacc[i].y = acc[i].y + C.dt * uu.y;
                } else {
                    qacc[i] = a[i];
                    if (look) { GPR_KEEP_BRANCH(); qacc[i] = clamp_dyn<RT>(vn, C.v_max, a[i], C.dt, C.inv_dt, vnext); }
                }
            }
            // mj_step (BE:1882): engine model gpr-push-v3. An env that has collided stands still (BE:1904-1905): its lanes sit the
            // cycle out (the votes inside push_cycle then only count the envs that still move, as they should)
            if (!frozen) {
                if constexpr (Math<RT>::literal) push_cycle<RT, NM>(Q, C.dt, m, o, qacc);
                else push_cycle_fast<NM>(Q, C.dt, m, o, qacc, rel0);
#pragma unroll
                for (int i = 0; i < NM; ++i) acc[i] = qacc[i];
            }
            bool w = false, hit = false;  // BE:1888-1901
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                bool wi;
                const R2<RT> pm = mover_xy(i);
                if constexpr (NOISE) {
                    bool unc;
                    wi = !pos_valid_circle<RT, FULL, true>(T, P, pm.x, pm.y, cw_step, C.band_p, &unc);
                    if (eager | (__ballot(unc & !frozen) != 0ull)) {
                        GPR_KEEP_BRANCH();
                        const R2<RT> pw = noisy<RT, true>(pm, C.sig_p, key[i], 3u * (uint32_t)cyc + 1u);
                        wi = !pos_valid_circle<RT, FULL>(T, P, pw.x, pw.y, cw_step);
                    }
                } else {
                    wi = !pos_valid_circle<RT, FULL>(T, P, pm.x, pm.y, cw_step);
                }
                w |= wi;
            }
            if constexpr (NM > 1) {
                RT d;
                const R2<RT> pm0 = mover_xy(0), pm1 = mover_xy(NM - 1);
                hit = pair_hit2<RT>(C, pm0, pm1, &d);
                if constexpr (NOISE) {
                    if (eager | (__ballot((absr(d - pair_thr) <= pair_band) & !frozen) != 0ull)) {
                        GPR_KEEP_BRANCH();
                        const R2<RT> q0 = noisy<RT, true>(pm0, C.sig_p, key[0], 3u * (uint32_t)cyc + 2u);
                        const R2<RT> q1 = noisy<RT, true>(pm1, C.sig_p, key[NM - 1], 3u * (uint32_t)cyc + 2u);
                        hit = pair_hit2<RT>(C, q0, q1);
                    }
                }
            }
            if (!frozen) { wc = w; mc = hit; frozen = w | hit; }  // BE:1904-1905
        }
        } else {
        // one wave-level decision per cycle for everything that may need a noisy read (see the pair-lane loop above)
        auto clamp_near = [&](int i) -> bool {
            if constexpr (JERK) return true;
            const RT t0 = m[i].vx + C.dt * a[i].x, t1 = m[i].vy + C.dt * a[i].y, vl = C.v_max - C.band_v;
            return !(t0 * t0 + t1 * t1 < vl * vl) | (vl <= (RT)0);
        };
        bool look[NM];
#pragma unroll
        for (int i = 0; i < NM; ++i) look[i] = JERK | eager | (__ballot(clamp_near(i) & !frozen) != 0ull);
        for (int cyc = 0; cyc < P.num_cycles; ++cyc) {
            // _mujoco_step_callback (PU:419-455): the same clamp chain as the planning env, on a noisy velocity read, per mover
            // lazy noise as in the planning kernels: a reading is drawn only if some lane's comparison could depend on it
            // (acceleration actions) while |v + dt a| stays below v_max - band_v in every lane, no velocity reading can make the clamp
            // fire (PL:634): the command is the action itself, and neither the reading nor the clamp is evaluated
            R2<RT> qacc[NM];
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                R2<RT> vn{m[i].vx, m[i].vy};
                if constexpr (NOISE) {
                    if (look[i]) { GPR_KEEP_BRANCH(); vn = noisy<RT, true>(vn, C.sig_v, key[i], 3u * (uint32_t)cyc); }
                }
                R2<RT> vnext;
                if constexpr (JERK) {
                    R2<RT> acc_tmp;
                    const R2<RT> j1 = clamp_dyn<RT>(acc[i], C.a_max, a[i], C.dt, C.inv_dt, acc_tmp);
                    const R2<RT> acc_p = clamp_dyn<RT>(vn, C.v_max, acc_tmp, C.dt, C.inv_dt, vnext);
                    const bool diff = (acc_tmp.x != acc_p.x) | (acc_tmp.y != acc_p.y);
                    const R2<RT> uu{diff ? div_rcp<RT>(acc_p.x - acc[i].x, C.dt, C.inv_dt) : j1.x, diff ? div_rcp<RT>(acc_p.y - acc[i].y, C.dt, C.inv_dt) : j1.y};
                    qacc[i].x = acc[i].x + C.dt * uu.x; qacc[i].y = acc[i].y + C.dt * uu.y;
                } else {
                    qacc[i] = a[i];
                    if (look[i]) { GPR_KEEP_BRANCH(); qacc[i] = clamp_dyn<RT>(vn, C.v_max, a[i], C.dt, C.inv_dt, vnext); }
                }
            }
            // mj_step (BE:1882): engine model gpr-push-v3. An env that has collided stands still (BE:1904-1905): its lanes sit the
            // cycle out (the votes inside push_cycle then only count the envs that still move, as they should)
            if (!frozen) {
                if constexpr (Math<RT>::literal) push_cycle<RT, NM>(Q, C.dt, m, o, qacc);
                else push_cycle_fast<NM>(Q, C.dt, m, o, qacc, rel0);
#pragma unroll
                for (int i = 0; i < NM; ++i) acc[i] = qacc[i];
            }
            bool hit = false;  // BE:1888-1901
            bool wi[NM], unc[NM], cn[NM];
            bool any_near = false, pnear = false;
            R2<RT> pm[NM];
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                pm[i] = mover_xy(i);
                unc[i] = false;
                if constexpr (NOISE) wi[i] = !pos_valid_circle<RT, FULL, true>(T, P, pm[i].x, pm[i].y, cw_step, C.band_p, &unc[i]);
                else wi[i] = !pos_valid_circle<RT, FULL>(T, P, pm[i].x, pm[i].y, cw_step);
                cn[i] = JERK ? false : clamp_near(i);
                any_near |= unc[i] | cn[i];
            }
            if constexpr (NM > 1) {
                RT d;
                hit = pair_hit2<RT>(C, pm[0], pm[NM - 1], &d);
                if constexpr (NOISE) pnear = absr(d - pair_thr) <= pair_band;
                any_near |= pnear;
            }
#pragma unroll
            for (int i = 0; i < NM; ++i) look[i] = JERK | eager;
            if (eager | (__ballot(any_near & !frozen) != 0ull)) {
                GPR_KEEP_BRANCH();
                if constexpr (NOISE) {
#pragma unroll
                    for (int i = 0; i < NM; ++i) {
                        if (eager | (__ballot(unc[i] & !frozen) != 0ull)) {
                            const R2<RT> pw = noisy<RT, true>(pm[i], C.sig_p, key[i], 3u * (uint32_t)cyc + 1u);
                            wi[i] = !pos_valid_circle<RT, FULL>(T, P, pw.x, pw.y, cw_step);
                        }
                    }
                    if constexpr (NM > 1) {
                        if (eager | (__ballot(pnear & !frozen) != 0ull)) {
                            const R2<RT> q0 = noisy<RT, true>(pm[0], C.sig_p, key[0], 3u * (uint32_t)cyc + 2u);
                            const R2<RT> q1 = noisy<RT, true>(pm[NM - 1], C.sig_p, key[NM - 1], 3u * (uint32_t)cyc + 2u);
                            hit = pair_hit2<RT>(C, q0, q1);
                        }
                    }
                }
                if constexpr (!JERK) {
#pragma unroll
                    for (int i = 0; i < NM; ++i) look[i] = eager | (__ballot(cn[i] & !frozen) != 0ull);
                }
            }
            bool w = false;
#pragma unroll
            for (int i = 0; i < NM; ++i) w |= wi[i];
            if (!frozen) { wc = w; mc = hit; frozen = w | hit; }  // BE:1904-1905
        }
        }
        }
        o.x = base_o.x + o.x; o.y = base_o.y + o.y;  // (float mode: back to absolute positions)
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const R2<RT> pm = mover_xy(i);
            p[i] = R2<RT>{stored<ST, RT>(pm.x), stored<ST, RT>(pm.y)}; v[i] = R2<RT>{stored<ST, RT>(m[i].vx), stored<ST, RT>(m[i].vy)};
            rot[i] = R2<RT>{stored<ST, RT>(m[i].th), stored<ST, RT>(m[i].w)};
            acc[i] = R2<RT>{stored<ST, RT>(acc[i].x), stored<ST, RT>(acc[i].y)};
        }
        o = Body<RT>{stored<ST, RT>(o.x), stored<ST, RT>(o.y), stored<ST, RT>(o.th), stored<ST, RT>(o.vx), stored<ST, RT>(o.vy), stored<ST, RT>(o.w)};
        out = push_observe<ST, RT, NM, NOISE>(P, Q, key, 3u * (uint32_t)P.num_cycles, p, v, o.x, o.y, g, wc, mc);
        if (!reset_first) el = el + 1;
        trunc = (P.max_episode_steps > 0) & (el >= P.max_episode_steps);
        done = (out.terminated | trunc) & !reset_first;
        need = reset_first | ((P.autoreset == 1) & done);
        if (live & (P.autoreset == 1) & done & (P.final_obs != nullptr)) push_store<ST, RT, NM>(P, e, JERK, out, acc, g, P.final_obs, P.final_ag, P.final_dg);
        if ((P.autoreset == 2) & done) el = -1;
        if (P.counters != nullptr) {  // episode counters (include/gpr.h: GPR_CNT_*), one bit per env
            const unsigned long long head = __ballot(live & !reset_first);
            const uint32_t c0 = (uint32_t)__popcll(__ballot(done) & head), c1 = (uint32_t)__popcll(__ballot(out.terminated) & head);
            const uint32_t c2 = (uint32_t)__popcll(__ballot(trunc) & head), c3 = (uint32_t)__popcll(__ballot(out.is_success) & head);
            const uint32_t c4 = (uint32_t)__popcll(__ballot(wc) & head), c5 = (uint32_t)__popcll(__ballot(mc) & head);
            const uint32_t L = __lane_id();
            const uint32_t add = L == 0u ? c0 : L == 1u ? c1 : L == 2u ? c2 : L == 3u ? c3 : L == 4u ? c4 : c5;
            if (L < 6u) P.counters[cnt_row * 6 + L] = cnt_old + add;
        }
    }
    if (__ballot(need) != 0ull) {  // pos_valid_circle votes: every lane takes part
        R2<RT> rp[NM], rg = g;
#pragma unroll
        for (int i = 0; i < NM; ++i) rp[i] = p[i];
        RT rox = o.x, roy = o.y;
        bool rwc, rmc;
        const PushObs<RT, NM> ro = push_reset_one<ST, RT, NM, NOISE, FULL>(P, Q, X, T, e, env_global, c, need, lead, rp, rg, rox, roy, rwc, rmc);
        if (need) {
            g = rg;
#pragma unroll
            for (int i = 0; i < NM; ++i) { p[i] = rp[i]; v[i] = R2<RT>{(RT)0, (RT)0}; acc[i] = v[i]; rot[i] = v[i]; }
            o = Body<RT>{rox, roy, (RT)0, (RT)0, (RT)0, (RT)0};
            el = 0;
            if (RESET_ONLY | reset_first) { out = ro; out.reward = (RT)0; out.terminated = false; trunc = false; wc = rwc; mc = rmc; }
            else {
                out.ag = ro.ag;
#pragma unroll
                for (int i = 0; i < NM; ++i) { out.mpos[i] = ro.mpos[i]; out.mvel[i] = ro.mvel[i]; }
            }
        }
    }
    if (live & (RESET_ONLY ? need : true)) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            store2<ST, RT>(P.pos, 2 * NM * e + 2 * i, p[i]); store2<ST, RT>(P.vel, 2 * NM * e + 2 * i, v[i]);
            store2<ST, RT>(P.acc, 2 * NM * e + 2 * i, acc[i]); store2<ST, RT>(X.mover_rot, 2 * NM * e + 2 * i, rot[i]);
        }
        store2<ST, RT>(P.goal, 2 * e, g);
        ST *ow = static_cast<ST *>(X.object) + 6 * e;
        ow[0] = (ST)o.x; ow[1] = (ST)o.y; ow[2] = (ST)o.th; ow[3] = (ST)o.vx; ow[4] = (ST)o.vy; ow[5] = (ST)o.w;
        push_store<ST, RT, NM>(P, e, JERK, out, acc, g, P.obs, P.ag, P.dg);
        P.elapsed[e] = el;
        static_cast<ST *>(P.reward)[e] = (ST)out.reward;
        P.terminated[e] = (uint8_t)out.terminated; P.truncated[e] = (uint8_t)trunc; P.is_success[e] = (uint8_t)out.is_success;
        P.mover_collision[e] = (uint8_t)mc; P.wall_collision[e] = (uint8_t)wc;
    }
}

static int push_simds() {
    static int simds = 0;  // (the devices of a node are alike)
    if (simds == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        simds = 1024;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) simds = 4 * prop.multiProcessorCount;
    }
    return simds;
}
// log2 of the lanes every env gets: the most (up to 64) that keep the grid within one wavefront per SIMD (see the top of this file)
static int push_lane_shift(int64_t E) {
    const int simds = push_simds();
    if (const char *f = std::getenv("GPR_PUSH_LANES")) {  // experiment knob: lanes per env (a power of two up to 64)
        int sh = 0;
        while ((2 << sh) <= std::atoi(f) && sh < 6) ++sh;
        return sh;
    }
    int sh = 0;
    while (sh < 6 && ((E << (sh + 1)) + 63) / 64 <= simds) ++sh;
    return sh;
}
template <typename ST, typename RT, int NM, bool JERK, bool NOISE, bool FULL>
static hipError_t launch_push3(bool is_reset, const KParams &P, const PushCst<RT> &Q, const PushPtrs &X0, hipStream_t s) {
    const unsigned lds = FULL ? 0u : P.table_bytes;
    PushPtrs X = X0;
    // (a reset is short: one lane per env)
    X.lane_shift = is_reset ? 0 : push_lane_shift(P.E);
    if constexpr (NM == 2 && !Math<RT>::literal && !JERK) {
        // two movers in float: a lane per (mover, object) pair (push_cycle_pair) while that keeps the grid within two wavefronts per SIMD --
        // beyond, the SIMDs are busy anyway and the couple's redundant object arithmetic only costs. GPR_PUSH_PAIR=0 / 1: never / always.
        bool pair = ((2 * P.E + 63) / 64) <= 2 * (int64_t)push_simds();
        if (const char *f = std::getenv("GPR_PUSH_PAIR")) pair = std::atoi(f) != 0;
        if (pair && !is_reset) {
            if (X.lane_shift < 1) X.lane_shift = 1;
            const unsigned grid = (unsigned)(((P.E << X.lane_shift) + BLOCK - 1) / BLOCK);
            hipLaunchKernelGGL((k_push<ST, RT, NM, JERK, NOISE, FULL, false, true>), dim3(grid), dim3(BLOCK), lds, s, P, Q, X);
            return hipGetLastError();
        }
    }
    const unsigned grid = (unsigned)(((P.E << X.lane_shift) + BLOCK - 1) / BLOCK);
    if (is_reset) hipLaunchKernelGGL((k_push<ST, RT, NM, JERK, NOISE, FULL, true>), dim3(grid), dim3(BLOCK), lds, s, P, Q, X);
    else hipLaunchKernelGGL((k_push<ST, RT, NM, JERK, NOISE, FULL, false>), dim3(grid), dim3(BLOCK), lds, s, P, Q, X);
    return hipGetLastError();
}
template <typename ST, typename RT, bool JERK, bool NOISE, bool FULL>
static hipError_t launch_push2(bool is_reset, const KParams &P, const PushCst<RT> &Q, const PushPtrs &X, hipStream_t s) {
    return P.M == 2 ? launch_push3<ST, RT, 2, JERK, NOISE, FULL>(is_reset, P, Q, X, s) : launch_push3<ST, RT, 1, JERK, NOISE, FULL>(is_reset, P, Q, X, s);
}
template <typename ST, typename RT> hipError_t launch_push(bool is_reset, const KParams &P, const PushCst<RT> &Q, const PushPtrs &X, hipStream_t s) {
    const bool jerk = P.learn_jerk != 0, noise = P.noise != 0;
    bool full = false;
    if constexpr (!Math<RT>::literal) full = P.layout_full != 0;
    if constexpr (Math<RT>::literal) {
        if (jerk) return noise ? launch_push2<ST, RT, true, true, false>(is_reset, P, Q, X, s) : launch_push2<ST, RT, true, false, false>(is_reset, P, Q, X, s);
        return noise ? launch_push2<ST, RT, false, true, false>(is_reset, P, Q, X, s) : launch_push2<ST, RT, false, false, false>(is_reset, P, Q, X, s);
    } else {
        if (jerk) return hipErrorInvalidValue;  // float arithmetic never runs jerk mode (see gpr_env_kernels.hip)
        if (full) return noise ? launch_push2<ST, RT, false, true, true>(is_reset, P, Q, X, s) : launch_push2<ST, RT, false, false, true>(is_reset, P, Q, X, s);
        return noise ? launch_push2<ST, RT, false, true, false>(is_reset, P, Q, X, s) : launch_push2<ST, RT, false, false, false>(is_reset, P, Q, X, s);
    }
}
hipError_t set_normal_table_push(const double *tab) { return hipMemcpyToSymbol(HIP_SYMBOL(g_normal_tab), &tab, sizeof(tab)); }
template hipError_t launch_push<float, float>(bool, const KParams &, const PushCst<float> &, const PushPtrs &, hipStream_t);
template hipError_t launch_push<float, double>(bool, const KParams &, const PushCst<double> &, const PushPtrs &, hipStream_t);
template hipError_t launch_push<double, double>(bool, const KParams &, const PushCst<double> &, const PushPtrs &, hipStream_t);

}  // namespace gpr
