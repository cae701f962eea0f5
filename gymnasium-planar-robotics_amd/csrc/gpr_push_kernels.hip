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
// of its own (push_cycle_pair below; 4 096 envs: 131 -> 105.5 us), taken while the grid stays within two wavefronts per SIMD.
//
// Pinned by the reference: control + clamp (PU:419-448), wall check (BE:426-788), obs / info / reward / terminated (PU:457-607),
// reset sampling (PU:373-417). The contact physics is the engine's own planar model "gpr-push-v2" (parity unpinned: the
// reference hands it to MuJoCo) -- see include/gpr.h and DESIGN.md section 9. The double builds run push_cycle, which is
// oracle/gpr_oracle.c:push_cycle operation for operation (1e-9 on every state in the tests); the float build runs
// push_cycle_fast, the same model written for the fewest instructions (equal up to float rounding).
//
// Everything a cycle touches lives in registers: the contact points of a pair are at most PUSH_SLOTS entries of fixed register
// arrays (written through unrolled selects, never indexed dynamically -- the first version kept them in scratch memory and
// spent its time waiting for it), their effective masses are inverted once per cycle, and the contact block of a pair is
// entered only by wavefronts in which some env's circumscribed circles overlap.
#include <cstdlib>

#include "gpr.h"
#include "gpr_device.h"
#include "gpr_host.h"

namespace gpr {

constexpr int PUSH_SLOTS = 4;

template <typename RT> struct Body { RT x, y, th, vx, vy, w; };
template <typename RT> struct PushPair {
    int np;
    RT nx, ny;
    RT rmx[PUSH_SLOTS], rmy[PUSH_SLOTS], rox[PUSH_SLOTS], roy[PUSH_SLOTS], rmn[PUSH_SLOTS], ron[PUSH_SLOTS], rmt[PUSH_SLOTS], rot_[PUSH_SLOTS];
    RT inv_kn[PUSH_SLOTS], inv_kt[PUSH_SLOTS], bias[PUSH_SLOTS], jn[PUSH_SLOTS], jt[PUSH_SLOTS];
};

template <typename RT> __device__ __forceinline__ RT cross2(RT ax, RT ay, RT bx, RT by) { return ax * by - ay * bx; }

// one control cycle of the NM movers and the object: oracle/gpr_oracle.c push_cycle operation for operation (the double builds;
// instantiated for float too, where push_cycle_fast below is what runs)
template <typename RT, int NM>
__device__ __forceinline__ void push_cycle(const PushCst<RT> &Q, RT dt, Body<RT> (&mv)[NM], Body<RT> &o, const R2<RT> (&u)[NM]) {
    // 1. external forces
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        Body<RT> &m = mv[i];
        m.vx = m.vx + dt * u[i].x;
        m.vy = m.vy + dt * u[i].y;
        m.w = Math<RT>::div_(m.w + dt * (Q.Kr * ((RT)0 - m.th)) * Q.inv_Im, (RT)1 + dt * Q.Dr * Q.inv_Im);
    }
    o.vx = o.vx * Q.damp_lin; o.vy = o.vy * Q.damp_lin;
    o.w = Math<RT>::div_(o.w, Q.damp_rot_den);
    {
        const RT sp = Math<RT>::sqrt_(o.vx * o.vx + o.vy * o.vy);
        if (sp <= Q.fric_dv) { o.vx = (RT)0; o.vy = (RT)0; } else { const RT k = Math<RT>::div_(sp - Q.fric_dv, sp); o.vx = o.vx * k; o.vy = o.vy * k; }
        const RT aw = absr(o.w);
        if (aw <= Q.fric_dw) o.w = (RT)0; else o.w = o.w * Math<RT>::div_(aw - Q.fric_dw, aw);
    }
    // 2. contact geometry per pair (mover i, object)
    PushPair<RT> pr[NM];
    bool any_contact = false;
    const RT sx[4] = {-1, -1, 1, 1}, sy[4] = {-1, 1, 1, -1};
    RT so = (RT)0, co = (RT)1;
    bool have_o = false;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        const Body<RT> &m = mv[i];
        PushPair<RT> &q = pr[i];
        q.np = 0; q.nx = (RT)0; q.ny = (RT)0;
        const RT dx = o.x - m.x, dy = o.y - m.y;
        const bool in_range = !(dx * dx + dy * dy > Q.reach2);  // broad phase: the circumscribed circles overlap
        if (__ballot(in_range) == 0ull) continue;
        if (!have_o) { Math<RT>::sincos_(o.th, so, co); have_o = true; }
        RT sm, cm;
        Math<RT>::sincos_(m.th, sm, cm);
        const RT ex4[4] = {cm, -sm, co, -so}, ey4[4] = {sm, cm, so, co};
        RT best = Math<RT>::inf(), nx = (RT)0, ny = (RT)0;
        bool touching = in_range;
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // separating-axis test over the 4 face normals
            const RT ex = ex4[k], ey = ey4[k];
            const RT rm = Q.ax * absr(ex * cm + ey * sm) + Q.ay * absr(-ex * sm + ey * cm);
            const RT ro = Q.h * absr(ex * co + ey * so) + Q.h * absr(-ex * so + ey * co);
            const RT d = dx * ex + dy * ey;
            const RT overlap = rm + ro - absr(d);
            if (touching) {
                if (overlap < (RT)0) touching = false;
                else if (overlap < best) { best = overlap; nx = d < (RT)0 ? -ex : ex; ny = d < (RT)0 ? -ey : ey; }
            }
        }
        if (__ballot(touching) == 0ull) continue;
        any_contact = true;
        // contact points: object vertices inside the mover, then mover vertices inside the object (at most PUSH_SLOTS, in this
        // order); depth along n. Slot `np` is addressed through unrolled selects: the arrays stay in registers.
        RT cx[PUSH_SLOTS] = {0, 0, 0, 0}, cy[PUSH_SLOTS] = {0, 0, 0, 0}, dep[PUSH_SLOTS] = {0, 0, 0, 0};
        int np = 0;
        auto add_point = [&](bool inside, RT vx, RT vy, RT d) {
#pragma unroll
            for (int s = 0; s < PUSH_SLOTS; ++s) {
                const bool w = inside & (np == s);
                cx[s] = w ? vx : cx[s]; cy[s] = w ? vy : cy[s]; dep[s] = w ? d : dep[s];
            }
            np += (int)(inside & (np < PUSH_SLOTS));
        };
        const RT rm_n = Q.ax * absr(nx * cm + ny * sm) + Q.ay * absr(-nx * sm + ny * cm);
        const RT ro_n = Q.h * absr(nx * co + ny * so) + Q.h * absr(-nx * so + ny * co);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const RT vx = o.x + co * (sx[k] * Q.h) - so * (sy[k] * Q.h), vy = o.y + so * (sx[k] * Q.h) + co * (sy[k] * Q.h);
            const RT lx = (vx - m.x) * cm + (vy - m.y) * sm, ly = -(vx - m.x) * sm + (vy - m.y) * cm;
            add_point(touching & (absr(lx) <= Q.ax) & (absr(ly) <= Q.ay), vx, vy, rm_n - ((vx - m.x) * nx + (vy - m.y) * ny));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const RT vx = m.x + cm * (sx[k] * Q.ax) - sm * (sy[k] * Q.ay), vy = m.y + sm * (sx[k] * Q.ax) + cm * (sy[k] * Q.ay);
            const RT lx = (vx - o.x) * co + (vy - o.y) * so, ly = -(vx - o.x) * so + (vy - o.y) * co;
            add_point(touching & (absr(lx) <= Q.h) & (absr(ly) <= Q.h), vx, vy, ro_n + ((vx - o.x) * nx + (vy - o.y) * ny));
        }
        if (touching & (np == 0)) { cx[0] = m.x + (RT)0.5 * dx; cy[0] = m.y + (RT)0.5 * dy; dep[0] = best; np = 1; }
        q.np = np; q.nx = nx; q.ny = ny;
        const RT tx = -ny, ty = nx;
#pragma unroll
        for (int k = 0; k < PUSH_SLOTS; ++k) {
            if (__ballot(k < np) == 0ull) break;
            q.rmx[k] = cx[k] - m.x; q.rmy[k] = cy[k] - m.y; q.rox[k] = cx[k] - o.x; q.roy[k] = cy[k] - o.y;
            q.rmn[k] = cross2(q.rmx[k], q.rmy[k], nx, ny); q.ron[k] = cross2(q.rox[k], q.roy[k], nx, ny);
            q.rmt[k] = cross2(q.rmx[k], q.rmy[k], tx, ty); q.rot_[k] = cross2(q.rox[k], q.roy[k], tx, ty);
            q.inv_kn[k] = Math<RT>::div_((RT)1, Q.inv_mm + Q.inv_mo + q.rmn[k] * q.rmn[k] * Q.inv_Im + q.ron[k] * q.ron[k] * Q.inv_Io);
            q.inv_kt[k] = Math<RT>::div_((RT)1, Q.inv_mm + Q.inv_mo + q.rmt[k] * q.rmt[k] * Q.inv_Im + q.rot_[k] * q.rot_[k] * Q.inv_Io);
            q.bias[k] = Math<RT>::div_(dep[k] > (RT)0 ? dep[k] : (RT)0, Q.tc);
            q.jn[k] = (RT)0; q.jt[k] = (RT)0;
        }
    }
    // 3. sequential impulses: 4 sweeps over mover 0's points, mover 1's points
    if (any_contact) {  // (wave-uniform)
        for (int it = 0; it < 4; ++it) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                Body<RT> &m = mv[i];
                PushPair<RT> &q = pr[i];
                const RT nx = q.nx, ny = q.ny, tx = -ny, ty = nx;
#pragma unroll
                for (int k = 0; k < PUSH_SLOTS; ++k) {
                    if (__ballot(k < q.np) == 0ull) break;
                    if (k < q.np) {
                        const RT rmx = q.rmx[k], rmy = q.rmy[k], rox = q.rox[k], roy = q.roy[k];
                        RT rvx = (o.vx - o.w * roy) - (m.vx - m.w * rmy), rvy = (o.vy + o.w * rox) - (m.vy + m.w * rmx);
                        RT dj = -((rvx * nx + rvy * ny) - q.bias[k]) * q.inv_kn[k];
                        RT jn_new = q.jn[k] + dj; if (jn_new < (RT)0) jn_new = (RT)0;
                        dj = jn_new - q.jn[k]; q.jn[k] = jn_new;
                        m.vx -= dj * nx * Q.inv_mm; m.vy -= dj * ny * Q.inv_mm; m.w -= dj * q.rmn[k] * Q.inv_Im;
                        o.vx += dj * nx * Q.inv_mo; o.vy += dj * ny * Q.inv_mo; o.w += dj * q.ron[k] * Q.inv_Io;
                        rvx = (o.vx - o.w * roy) - (m.vx - m.w * rmy); rvy = (o.vy + o.w * rox) - (m.vy + m.w * rmx);
                        RT djt = -(rvx * tx + rvy * ty) * q.inv_kt[k];
                        const RT lim = Q.mu * q.jn[k];
                        RT jt_new = q.jt[k] + djt;
                        if (jt_new > lim) jt_new = lim;
                        if (jt_new < -lim) jt_new = -lim;
                        djt = jt_new - q.jt[k]; q.jt[k] = jt_new;
                        m.vx -= djt * tx * Q.inv_mm; m.vy -= djt * ty * Q.inv_mm; m.w -= djt * q.rmt[k] * Q.inv_Im;
                        o.vx += djt * tx * Q.inv_mo; o.vy += djt * ty * Q.inv_mo; o.w += djt * q.rot_[k] * Q.inv_Io;
                    }
                }
            }
        }
    }
    // 4. positions with the new velocities
#pragma unroll
    for (int i = 0; i < NM; ++i) { Body<RT> &m = mv[i]; m.x += dt * m.vx; m.y += dt * m.vy; m.th += dt * m.w; }
    o.x += dt * o.vx; o.y += dt * o.vy; o.th += dt * o.w;
}

// THROUGHPUT FORM of push_cycle (float arithmetic only): the same model, the same order of the sequential impulses, written for
// the fewest vector instructions -- one lane per env and one wavefront per SIMD at 65 536 envs means the kernel lasts as long as
// the instruction chain of a wavefront in which some env is in contact, ~1 000 instructions per cycle in the literal form above:
//   * the geometry of a pair lives in the MOVER's frame: one relative rotation, two of the four SAT axes are coordinate axes,
//     the object's vertices are four sums and the mover's are constants;
//   * the eight "vertex inside the other box" tests only set bits of a mask; the (at most PUSH_SLOTS) contact points are then
//     rebuilt from their bit index -- no register arrays written through selects;
//   * the relative velocity at a contact point along n / t is n.(vo - vm) + wo (ro x n) - wm (rm x n): the lever arms
//     themselves are never stored, the products with the inverse masses / inertias are formed once per cycle.
// Equal to push_cycle up to float rounding (tests/test_gpu_pushing.py holds the float build to the oracle through it).
// POSITIONS ARE DISPLACEMENTS. A body's x, y here are its displacement since the start of the env-step (at most 8 cm: float ulp 7e-9 m) and
// rel0[i] is object - mover i at the step's start, formed once in double: the geometry only ever needs object - mover. Summing the
// 1e-3-sized per-cycle increments onto ~0.4 m positions in float instead put 3e-8 m of rounding into every cycle, and a contact
// manifold changes discretely (a vertex enters or leaves a face, the axis of least overlap switches) wherever such an error decides:
// the float build then pushes the 10 g object one cycle earlier or later than the fp64 oracle -- the "events" of tests/test_gpu_pushing.py.
// the contact points of one (mover, object) pair in the throughput form
struct FastPair {
    int np, npw;  // contact points of this env's pair; the most of any env of the wavefront (scalar: bounds the slot loops)
    float nx, ny;
    float rmn[PUSH_SLOTS], ron[PUSH_SLOTS], rmt[PUSH_SLOTS], rot_[PUSH_SLOTS];  // lever arms x n, x t
    float wmn[PUSH_SLOTS], won[PUSH_SLOTS], wmt[PUSH_SLOTS], wot[PUSH_SLOTS];   // ... times the inverse inertias
    float inv_kn[PUSH_SLOTS], inv_kt[PUSH_SLOTS], bias[PUSH_SLOTS], jn[PUSH_SLOTS], jt[PUSH_SLOTS];
};

// 1. external forces: a mover's actuation and yaw impedance; the object's damping and friction against the tile
__device__ __forceinline__ void fast_forces_mover(const PushCst<float> &Q, float dt, Body<float> &m, R2<float> u) {
    const float w_gain = dt * Q.Kr * Q.inv_Im, w_den = __builtin_amdgcn_rcpf(1.f + dt * Q.Dr * Q.inv_Im);
    m.vx = __builtin_fmaf(dt, u.x, m.vx);
    m.vy = __builtin_fmaf(dt, u.y, m.vy);
    m.w = __builtin_fmaf(-w_gain, m.th, m.w) * w_den;
}
__device__ __forceinline__ void fast_forces_object(const PushCst<float> &Q, Body<float> &o) {
    const float vx = o.vx * Q.damp_lin, vy = o.vy * Q.damp_lin, w = o.w * __builtin_amdgcn_rcpf(Q.damp_rot_den);
    const float sp2 = __builtin_fmaf(vx, vx, vy * vy);
    const float k = sp2 > Q.fric_dv * Q.fric_dv ? __builtin_fmaf(-Q.fric_dv, __builtin_amdgcn_rsqf(sp2), 1.f) : 0.f;  // (sp - dv) / sp
    o.vx = vx * k; o.vy = vy * k;
    const float aw = __builtin_fabsf(w);
    o.w = aw <= Q.fric_dw ? 0.f : __builtin_copysignf(aw - Q.fric_dw, w);
}

// 2. contact geometry of the pair (mover m, object), in the mover's frame. Returns whether some env of the wavefront touches (wave-uniform);
// q.npw stays 0 otherwise and the slots hold nothing. so / co: sine and cosine of the object's yaw, formed on first use (have_o).
__device__ __forceinline__ bool fast_pair_geometry(const PushCst<float> &Q, const Body<float> &m, const Body<float> &o, R2<float> rel0, float &so, float &co,
                                                   bool &have_o, float imm, float inv_tc, FastPair &q) {
    q.np = 0; q.npw = 0; q.nx = 0.f; q.ny = 0.f;
    const float dx = rel0.x + (o.x - m.x), dy = rel0.y + (o.y - m.y);
    const bool in_range = !(__builtin_fmaf(dx, dx, dy * dy) > Q.reach2);  // broad phase: the circumscribed circles overlap
    if (__ballot(in_range) == 0ull) return false;
    if (!have_o) { Math<float>::sincos_(o.th, so, co); have_o = true; }
    float sm, cm;
    Math<float>::sincos_(m.th, sm, cm);
    const float c = __builtin_fmaf(co, cm, so * sm), s = __builtin_fmaf(so, cm, -(co * sm));  // the object's axes in the mover's frame: (c, s), (-s, c)
    const float lx = __builtin_fmaf(dx, cm, dy * sm), ly = __builtin_fmaf(dy, cm, -(dx * sm)); // the object's centre in the mover's frame
    const float ac = __builtin_fabsf(c), as = __builtin_fabsf(s);
    // separating axes in the order of push_cycle: mover x, mover y, object x, object y; the first smallest overlap wins
    const float ro01 = Q.h * (ac + as);
    const float d2 = __builtin_fmaf(lx, c, ly * s), d3 = __builtin_fmaf(ly, c, -(lx * s));
    const float ov0 = (Q.ax + ro01) - __builtin_fabsf(lx), ov1 = (Q.ay + ro01) - __builtin_fabsf(ly);
    const float ov2 = (__builtin_fmaf(Q.ax, ac, Q.ay * as) + Q.h) - __builtin_fabsf(d2);
    const float ov3 = (__builtin_fmaf(Q.ax, as, Q.ay * ac) + Q.h) - __builtin_fabsf(d3);
    const bool touching = in_range & !(__builtin_fminf(__builtin_fminf(ov0, ov1), __builtin_fminf(ov2, ov3)) < 0.f);
    if (__ballot(touching) == 0ull) return false;
    float best = ov0, nlx = __builtin_copysignf(1.f, lx), nly = 0.f;  // the normal in the mover's frame, towards the object
    { const bool t = ov1 < best; best = t ? ov1 : best; nlx = t ? 0.f : nlx; nly = t ? __builtin_copysignf(1.f, ly) : nly; }
    { const bool t = ov2 < best; best = t ? ov2 : best; nlx = t ? __builtin_copysignf(c, c * d2) : nlx; nly = t ? __builtin_copysignf(s, s * d2) : nly; }
    { const bool t = ov3 < best; best = t ? ov3 : best; nlx = t ? __builtin_copysignf(s, -(s * d3)) : nlx; nly = t ? __builtin_copysignf(c, c * d3) : nly; }
    const float nx = __builtin_fmaf(nlx, cm, -(nly * sm)), ny = __builtin_fmaf(nlx, sm, nly * cm);
    const float rm_n = __builtin_fmaf(Q.ax, __builtin_fabsf(nlx), Q.ay * __builtin_fabsf(nly));
    const float ro_n = Q.h * (__builtin_fabsf(__builtin_fmaf(nlx, c, nly * s)) + __builtin_fabsf(__builtin_fmaf(nly, c, -(nlx * s))));
    // candidates: bit k = object vertex k inside the mover, bit 4 + k = mover vertex k inside the object (vertex order of push_cycle)
    const float hc = Q.h * c, hs = Q.h * s, e1 = hc - hs, e2 = hc + hs;
    auto in_m = [&](float vx, float vy) { return (__builtin_fabsf(vx) <= Q.ax) & (__builtin_fabsf(vy) <= Q.ay); };
    const float qxm = -Q.ax - lx, qxp = Q.ax - lx, qym = -Q.ay - ly, qyp = Q.ay - ly;  // mover vertices relative to the object's centre
    const float qyms = qym * s, qyps = qyp * s, qxms = qxm * s, qxps = qxp * s;
    auto in_o = [&](float qx, float qy, float qys, float qxs) {
        return (__builtin_fabsf(__builtin_fmaf(qx, c, qys)) <= Q.h) & (__builtin_fabsf(__builtin_fmaf(qy, c, -qxs)) <= Q.h);
    };
    unsigned mask = (in_m(lx - e1, ly - e2) ? 1u : 0u) | (in_m(lx - e2, ly + e1) ? 2u : 0u) | (in_m(lx + e1, ly + e2) ? 4u : 0u) |
                    (in_m(lx + e2, ly - e1) ? 8u : 0u) | (in_o(qxm, qym, qyms, qxms) ? 16u : 0u) | (in_o(qxm, qyp, qyps, qxms) ? 32u : 0u) |
                    (in_o(qxp, qyp, qyps, qxps) ? 64u : 0u) | (in_o(qxp, qym, qyms, qxps) ? 128u : 0u);
    mask = touching ? mask : 0u;
    const bool lone = touching & (mask == 0u);  // boxes overlap but no vertex is inside the other: one point between the centres
    const int np_raw = __popc(mask);
    const int np = lone ? 1 : (np_raw < PUSH_SLOTS ? np_raw : PUSH_SLOTS);
    q.np = np; q.nx = nx; q.ny = ny;
    // (one wave-level decision per pair and cycle instead of one per slot and sweep: the loops below are scalar-bounded)
    const int npw = __ballot(np > 1) == 0ull ? 1 : __ballot(np > 2) == 0ull ? 2 : __ballot(np > 3) == 0ull ? 3 : 4;
    q.npw = npw;
#pragma unroll
    for (int k = 0; k < PUSH_SLOTS; ++k) {
        if (k >= npw) break;
        const int idx = mask != 0u ? __builtin_ctz(mask) : 0;
        mask &= mask - 1u;
        const bool of_mover = idx >= 4;
        const float sxf = (idx & 2) ? 1.f : -1.f, syf = ((idx + 1) & 2) ? 1.f : -1.f;
        // the point in the mover's frame and its depth along n
        float px = of_mover ? sxf * Q.ax : __builtin_fmaf(-syf, hs, __builtin_fmaf(sxf, hc, lx));
        float py = of_mover ? syf * Q.ay : __builtin_fmaf(syf, hc, __builtin_fmaf(sxf, hs, ly));
        const float pn = of_mover ? __builtin_fmaf(px - lx, nlx, (py - ly) * nly) : -__builtin_fmaf(px, nlx, py * nly);
        float dep = (of_mover ? ro_n : rm_n) + pn;
        float rmx = __builtin_fmaf(px, cm, -(py * sm)), rmy = __builtin_fmaf(px, sm, py * cm);
        if (k == 0) { rmx = lone ? 0.5f * dx : rmx; rmy = lone ? 0.5f * dy : rmy; dep = lone ? best : dep; }
        const float rox = rmx - dx, roy = rmy - dy;
        const float rmn = __builtin_fmaf(rmx, ny, -(rmy * nx)), ron = __builtin_fmaf(rox, ny, -(roy * nx));
        const float rmt = __builtin_fmaf(rmx, nx, rmy * ny), rot = __builtin_fmaf(rox, nx, roy * ny);  // r x t with t = (-ny, nx)
        q.rmn[k] = rmn; q.ron[k] = ron; q.rmt[k] = rmt; q.rot_[k] = rot;
        q.wmn[k] = rmn * Q.inv_Im; q.won[k] = ron * Q.inv_Io; q.wmt[k] = rmt * Q.inv_Im; q.wot[k] = rot * Q.inv_Io;
        q.inv_kn[k] = __builtin_amdgcn_rcpf(__builtin_fmaf(rmn, q.wmn[k], __builtin_fmaf(ron, q.won[k], imm)));
        q.inv_kt[k] = __builtin_amdgcn_rcpf(__builtin_fmaf(rmt, q.wmt[k], __builtin_fmaf(rot, q.wot[k], imm)));
        q.bias[k] = __builtin_fmaxf(dep, 0.f) * inv_tc;
        q.jn[k] = 0.f; q.jt[k] = 0.f;
    }
    return true;
}

// 3a. sequential impulses of ONE pair with at most two points in any env of the wavefront (the usual face-to-face push): the same
// Gauss-Seidel sweeps on the IMPULSES. With b = J V (the relative velocities at the points before any impulse) and A = J M^-1 J^T (formed
// once per cycle: n.n = t.t = 1, n.t = 0, the lever-arm products are there) the relative velocity along a row is b + A lambda, an update is
// three fma's on the other rows' impulses instead of re-deriving it from six body velocities and pushing the impulse back into them
// (14 instead of 35 instructions per point and sweep); the bodies take the summed impulses at the end. A lane without contact
// (q.np == 0) changes nothing.
__device__ __forceinline__ void fast_impulse_solve(const PushCst<float> &Q, float imm, Body<float> &m, Body<float> &o, const FastPair &q) {
    const float nx = q.nx, ny = q.ny;
    const bool two = q.np > 1;
    const float dvx = o.vx - m.vx, dvy = o.vy - m.vy, dvn = __builtin_fmaf(dvx, nx, dvy * ny), dvt = __builtin_fmaf(dvy, nx, -(dvx * ny));
    const float bn0 = __builtin_fmaf(o.w, q.ron[0], __builtin_fmaf(-m.w, q.rmn[0], dvn)), bt0 = __builtin_fmaf(o.w, q.rot_[0], __builtin_fmaf(-m.w, q.rmt[0], dvt));
    const float a_n0t0 = __builtin_fmaf(q.rmn[0], q.wmt[0], q.ron[0] * q.wot[0]);
    const bool both = q.npw > 1;  // (wave-uniform; slot 1 holds nothing otherwise)
    float bn1 = 0.f, bt1 = 0.f, a_n1t1 = 0.f, a_n0n1 = 0.f, a_t0t1 = 0.f, a_n0t1 = 0.f, a_t0n1 = 0.f, ikn1 = 0.f, ikt1 = 0.f, bias1 = 0.f;
    if (both) {
        bn1 = __builtin_fmaf(o.w, q.ron[1], __builtin_fmaf(-m.w, q.rmn[1], dvn)); bt1 = __builtin_fmaf(o.w, q.rot_[1], __builtin_fmaf(-m.w, q.rmt[1], dvt));
        a_n1t1 = __builtin_fmaf(q.rmn[1], q.wmt[1], q.ron[1] * q.wot[1]);
        a_n0n1 = __builtin_fmaf(q.rmn[0], q.wmn[1], __builtin_fmaf(q.ron[0], q.won[1], imm));
        a_t0t1 = __builtin_fmaf(q.rmt[0], q.wmt[1], __builtin_fmaf(q.rot_[0], q.wot[1], imm));
        a_n0t1 = __builtin_fmaf(q.rmn[0], q.wmt[1], q.ron[0] * q.wot[1]); a_t0n1 = __builtin_fmaf(q.rmt[0], q.wmn[1], q.rot_[0] * q.won[1]);
        ikn1 = two ? q.inv_kn[1] : 0.f; ikt1 = two ? q.inv_kt[1] : 0.f;  // an env with one point: its second row stays at zero
        bias1 = q.bias[1];
    }
    float jn0 = 0.f, jt0 = 0.f, jn1 = 0.f, jt1 = 0.f;
    for (int it = 0; it < 4; ++it) {
        float r = bn0 + a_n0t0 * jt0;
        if (both) r = __builtin_fmaf(a_n0t1, jt1, __builtin_fmaf(a_n0n1, jn1, r));
        jn0 = __builtin_fmaxf((q.bias[0] - r) * q.inv_kn[0], 0.f);
        r = __builtin_fmaf(a_n0t0, jn0, bt0);
        if (both) r = __builtin_fmaf(a_t0t1, jt1, __builtin_fmaf(a_t0n1, jn1, r));
        float lim = Q.mu * jn0;
        jt0 = __builtin_fminf(__builtin_fmaxf(-r * q.inv_kt[0], -lim), lim);
        if (both) {
            r = __builtin_fmaf(a_n1t1, jt1, __builtin_fmaf(a_t0n1, jt0, __builtin_fmaf(a_n0n1, jn0, bn1)));
            jn1 = __builtin_fmaxf((bias1 - r) * ikn1, 0.f);
            r = __builtin_fmaf(a_n1t1, jn1, __builtin_fmaf(a_t0t1, jt0, __builtin_fmaf(a_n0t1, jn0, bt1)));
            lim = Q.mu * jn1;
            jt1 = __builtin_fminf(__builtin_fmaxf(-r * ikt1, -lim), lim);
        }
    }
    if (q.np == 0) { jn0 = 0.f; jt0 = 0.f; }  // (an env without contact in a wavefront with one)
    const float jn = jn0 + jn1, jt = jt0 + jt1;
    const float px = __builtin_fmaf(jn, nx, -(jt * ny)), py = __builtin_fmaf(jn, ny, jt * nx);  // summed impulse on the object; t = (-ny, nx)
    m.vx = __builtin_fmaf(-px, Q.inv_mm, m.vx); m.vy = __builtin_fmaf(-py, Q.inv_mm, m.vy);
    o.vx = __builtin_fmaf(px, Q.inv_mo, o.vx); o.vy = __builtin_fmaf(py, Q.inv_mo, o.vy);
    float dwm = __builtin_fmaf(jn0, q.wmn[0], jt0 * q.wmt[0]), dwo = __builtin_fmaf(jn0, q.won[0], jt0 * q.wot[0]);
    if (both) {  // (slot 1 holds nothing in a wavefront whose envs all have one point)
        dwm = __builtin_fmaf(jn1, q.wmn[1], __builtin_fmaf(jt1, q.wmt[1], dwm));
        dwo = __builtin_fmaf(jn1, q.won[1], __builtin_fmaf(jt1, q.wot[1], dwo));
    }
    m.w -= dwm; o.w += dwo;
}

// 3b. one visit of a pair's points in a sweep, on the body velocities (any number of points, any number of pairs in contact);
// `mine`: the lane takes part (pair lanes: only the lanes of the pair whose turn it is)
__device__ __forceinline__ void fast_velocity_rows(const PushCst<float> &Q, Body<float> &m, Body<float> &o, FastPair &q, bool mine = true) {
    const float nx = q.nx, ny = q.ny;
    const float nmx = nx * Q.inv_mm, nmy = ny * Q.inv_mm, nox = nx * Q.inv_mo, noy = ny * Q.inv_mo;
#pragma unroll
    for (int k = 0; k < PUSH_SLOTS; ++k) {
        if (k >= q.npw) break;
        if ((k < q.np) & mine) {
            float dvx = o.vx - m.vx, dvy = o.vy - m.vy;
            const float vn = __builtin_fmaf(dvx, nx, __builtin_fmaf(dvy, ny, __builtin_fmaf(o.w, q.ron[k], -(m.w * q.rmn[k]))));
            const float jn_new = __builtin_fmaxf(__builtin_fmaf(q.bias[k] - vn, q.inv_kn[k], q.jn[k]), 0.f);
            const float dj = jn_new - q.jn[k];
            q.jn[k] = jn_new;
            m.vx = __builtin_fmaf(-dj, nmx, m.vx); m.vy = __builtin_fmaf(-dj, nmy, m.vy); m.w = __builtin_fmaf(-dj, q.wmn[k], m.w);
            o.vx = __builtin_fmaf(dj, nox, o.vx); o.vy = __builtin_fmaf(dj, noy, o.vy); o.w = __builtin_fmaf(dj, q.won[k], o.w);
            dvx = o.vx - m.vx; dvy = o.vy - m.vy;
            const float vt = __builtin_fmaf(dvy, nx, __builtin_fmaf(-dvx, ny, __builtin_fmaf(o.w, q.rot_[k], -(m.w * q.rmt[k]))));
            const float lim = Q.mu * jn_new;
            const float jt_new = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(-vt, q.inv_kt[k], q.jt[k]), -lim), lim);
            const float djt = jt_new - q.jt[k];
            q.jt[k] = jt_new;
            // t = (-ny, nx)
            m.vx = __builtin_fmaf(djt, nmy, m.vx); m.vy = __builtin_fmaf(-djt, nmx, m.vy); m.w = __builtin_fmaf(-djt, q.wmt[k], m.w);
            o.vx = __builtin_fmaf(-djt, noy, o.vx); o.vy = __builtin_fmaf(djt, nox, o.vy); o.w = __builtin_fmaf(djt, q.wot[k], o.w);
        }
    }
}

template <int NM>
__device__ __forceinline__ void push_cycle_fast(const PushCst<float> &Q, float dt, Body<float> (&mv)[NM], Body<float> &o, const R2<float> (&u)[NM],
                                                const R2<float> (&rel0)[NM]) {
#pragma unroll
    for (int i = 0; i < NM; ++i) fast_forces_mover(Q, dt, mv[i], u[i]);
    fast_forces_object(Q, o);
    FastPair pr[NM];
    bool any_contact = false;
    float so = 0.f, co = 1.f;
    bool have_o = false;
    const float imm = Q.inv_mm + Q.inv_mo, inv_tc = __builtin_amdgcn_rcpf(Q.tc);
#pragma unroll
    for (int i = 0; i < NM; ++i) any_contact |= fast_pair_geometry(Q, mv[i], o, rel0[i], so, co, have_o, imm, inv_tc, pr[i]);
    // 3. sequential impulses: 4 sweeps over mover 0's points, mover 1's points
    bool in_impulse_space = false;
    if constexpr (NM == 1) {
        in_impulse_space = any_contact & (pr[0].npw <= 2);
        if (in_impulse_space) fast_impulse_solve(Q, imm, mv[0], o, pr[0]);
    } else {
        // Two movers: the sweeps visit mover 0's points, then mover 1's -- and an env in which only ONE of the two pairs touches has an
        // empty half in every sweep: the one-pair solve above is the same computation. It applies when no env of the wavefront has both
        // pairs in contact (with 4 envs per wavefront at 4 096 envs: practically always). Which pair is per lane: operands selected, the
        // mover's velocity written back to the selected mover. (A pair whose block was skipped, or whose second slot was never filled,
        // holds nothing: zeros are selected there.)
        const bool ran0 = pr[0].npw > 0, ran1 = pr[NM - 1].npw > 0;  // (wave-uniform)
        const int npw = pr[0].npw > pr[NM - 1].npw ? pr[0].npw : pr[NM - 1].npw;
        in_impulse_space = any_contact & (npw <= 2) & (__ballot((pr[0].np > 0) & (pr[NM - 1].np > 0)) == 0ull);
        if (in_impulse_space) {
            if (ran0 & !ran1) fast_impulse_solve(Q, imm, mv[0], o, pr[0]);
            else if (ran1 & !ran0) fast_impulse_solve(Q, imm, mv[NM - 1], o, pr[NM - 1]);
            else {
                const FastPair &a = pr[0], &b = pr[NM - 1];
                const bool sel = b.np > 0;  // this env's touching pair is mover 1's (neither touches: pair 0, np = 0, nothing happens)
                const bool a1 = a.npw > 1, b1 = b.npw > 1;
                FastPair q;
                q.np = sel ? b.np : a.np; q.npw = npw; q.nx = sel ? b.nx : a.nx; q.ny = sel ? b.ny : a.ny;
#define GPR_SEL0(f) q.f[0] = sel ? b.f[0] : a.f[0]
#define GPR_SEL1(f) q.f[1] = sel ? (b1 ? b.f[1] : 0.f) : (a1 ? a.f[1] : 0.f)
                GPR_SEL0(rmn); GPR_SEL0(ron); GPR_SEL0(rmt); GPR_SEL0(rot_); GPR_SEL0(wmn); GPR_SEL0(won); GPR_SEL0(wmt); GPR_SEL0(wot);
                GPR_SEL0(inv_kn); GPR_SEL0(inv_kt); GPR_SEL0(bias);
                GPR_SEL1(rmn); GPR_SEL1(ron); GPR_SEL1(rmt); GPR_SEL1(rot_); GPR_SEL1(wmn); GPR_SEL1(won); GPR_SEL1(wmt); GPR_SEL1(wot);
                GPR_SEL1(inv_kn); GPR_SEL1(inv_kt); GPR_SEL1(bias);
#undef GPR_SEL0
#undef GPR_SEL1
                Body<float> m = mv[0];
                m.vx = sel ? mv[NM - 1].vx : mv[0].vx; m.vy = sel ? mv[NM - 1].vy : mv[0].vy; m.w = sel ? mv[NM - 1].w : mv[0].w;
                fast_impulse_solve(Q, imm, m, o, q);
                mv[0].vx = sel ? mv[0].vx : m.vx; mv[0].vy = sel ? mv[0].vy : m.vy; mv[0].w = sel ? mv[0].w : m.w;
                mv[NM - 1].vx = sel ? m.vx : mv[NM - 1].vx; mv[NM - 1].vy = sel ? m.vy : mv[NM - 1].vy; mv[NM - 1].w = sel ? m.w : mv[NM - 1].w;
            }
        }
    }
    if (in_impulse_space) {
    } else if (any_contact) {  // (wave-uniform)
        for (int it = 0; it < 4; ++it) {
#pragma unroll
            for (int i = 0; i < NM; ++i) fast_velocity_rows(Q, mv[i], o, pr[i]);
        }
    }
    // 4. positions with the new velocities
#pragma unroll
    for (int i = 0; i < NM; ++i) { Body<float> &m = mv[i]; m.x = __builtin_fmaf(dt, m.vx, m.x); m.y = __builtin_fmaf(dt, m.vy, m.y); m.th = __builtin_fmaf(dt, m.w, m.th); }
    o.x = __builtin_fmaf(dt, o.vx, o.x); o.y = __builtin_fmaf(dt, o.vy, o.y); o.th = __builtin_fmaf(dt, o.w, o.th);
}

// the value the other lane of this lane's couple holds (lanes 2k, 2k + 1): quad_perm [1,0,3,2]
__device__ __forceinline__ uint32_t partner_u32(uint32_t v) { return rot_u32<2, 1>(v); }
template <typename RT> __device__ __forceinline__ RT partner(RT v) { return rot<2, 1>(v); }

// TWO MOVERS, ONE LANE PER PAIR (k_push<..., PAIR = true>). The two pairs' geometry blocks (~250 instructions each) are where a two-mover
// cycle spends its time, and in one lane they run one after the other. Here an env's lanes come in couples: the even lane owns mover 0
// and the pair (mover 0, object), the odd lane mover 1 and its pair; both carry the object. A cycle is then as long as the ONE-mover
// cycle plus the exchange: the object's new velocity goes from the lane whose pair touched to its partner (three quad_perm moves).
// Arithmetic and its order are those of push_cycle_fast<2>: with one pair in contact the one-pair impulse-space solve in that pair's lane;
// otherwise (some env of the wavefront has both pairs touching, or more than two points) the velocity sweeps, mover 0's points, then
// mover 1's, the object's velocity handed over after every visit.
__device__ __forceinline__ void push_cycle_pair(const PushCst<float> &Q, float dt, Body<float> &m, Body<float> &o, R2<float> u, R2<float> rel0, bool second) {
    fast_forces_mover(Q, dt, m, u);
    fast_forces_object(Q, o);
    FastPair q;
    float so = 0.f, co = 1.f;
    bool have_o = false;
    const float imm = Q.inv_mm + Q.inv_mo, inv_tc = __builtin_amdgcn_rcpf(Q.tc);
    if (fast_pair_geometry(Q, m, o, rel0, so, co, have_o, imm, inv_tc, q)) {  // (wave-uniform: some lane of either pair touches)
        const bool other = partner_u32((uint32_t)q.np) != 0u;                // the partner's pair touches
        auto from_partner = [&](bool take) {
            const float vx = partner(o.vx), vy = partner(o.vy), w = partner(o.w);
            o.vx = take ? vx : o.vx; o.vy = take ? vy : o.vy; o.w = take ? w : o.w;
        };
        if ((q.npw <= 2) & (__ballot((q.np > 0) & other) == 0ull)) {
            fast_impulse_solve(Q, imm, m, o, q);
            from_partner((q.np == 0) & other);
        } else {
            for (int it = 0; it < 4; ++it) {
                fast_velocity_rows(Q, m, o, q, !second);
                from_partner(second);
                fast_velocity_rows(Q, m, o, q, second);
                from_partner(!second);
            }
        }
    }
    m.x = __builtin_fmaf(dt, m.vx, m.x); m.y = __builtin_fmaf(dt, m.vy, m.y); m.th = __builtin_fmaf(dt, m.w, m.th);
    o.x = __builtin_fmaf(dt, o.vx, o.x); o.y = __builtin_fmaf(dt, o.vy, o.y); o.th = __builtin_fmaf(dt, o.w, o.th);
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
            for (int cyc = 0; cyc < P.num_cycles; ++cyc) {
                R2<RT> vn{mm.vx, mm.vy};
                if constexpr (NOISE) {
                    const RT t0 = mm.vx + C.dt * am.x, t1 = mm.vy + C.dt * am.y, vl = C.v_max - C.band_v;
                    const bool nd = !(t0 * t0 + t1 * t1 < vl * vl) | (vl <= (RT)0);
                    if (eager | (__ballot(nd & !frozen) != 0ull)) vn = noisy<RT, true>(vn, C.sig_v, keym, 3u * (uint32_t)cyc);
                }
                R2<RT> vnext;
                const R2<RT> qaccm = clamp_dyn<RT>(vn, C.v_max, am, C.dt, C.inv_dt, vnext);
                if (!frozen) {  // (both lanes of an env are frozen together: the exchanges inside always find their partner)
                    push_cycle_pair(Q, C.dt, mm, o, qaccm, rel0m, second);
                    accm = qaccm;
                }
                const R2<RT> pm{basem.x + mm.x, basem.y + mm.y};
                bool wi;
                if constexpr (NOISE) {
                    bool unc;
                    wi = !pos_valid_circle<RT, FULL, true>(T, P, pm.x, pm.y, cw_step, C.band_p, &unc);
                    if (eager | (__ballot(unc & !frozen) != 0ull)) {
                        const R2<RT> pw = noisy<RT, true>(pm, C.sig_p, keym, 3u * (uint32_t)cyc + 1u);
                        wi = !pos_valid_circle<RT, FULL>(T, P, pw.x, pw.y, cw_step);
                    }
                } else {
                    wi = !pos_valid_circle<RT, FULL>(T, P, pm.x, pm.y, cw_step);
                }
                const bool w = wi | (partner_u32(wi ? 1u : 0u) != 0u);
                RT d;
                bool hit = pair_hit2<RT>(C, pm, R2<RT>{partner(pm.x), partner(pm.y)}, &d);  // (symmetric in its two points, bit for bit)
                if constexpr (NOISE) {
                    if (eager | (__ballot((absr(d - pair_thr) <= pair_band) & !frozen) != 0ull)) {
                        const R2<RT> qm = noisy<RT, true>(pm, C.sig_p, keym, 3u * (uint32_t)cyc + 2u);
                        hit = pair_hit2<RT>(C, qm, R2<RT>{partner(qm.x), partner(qm.y)});
                    }
                }
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
        for (int cyc = 0; cyc < P.num_cycles; ++cyc) {
            // _mujoco_step_callback (PU:419-455): the same clamp chain as the planning env, on a noisy velocity read, per mover
            // lazy noise as in the planning kernels: a reading is drawn only if some lane's comparison could depend on it
            R2<RT> qacc[NM];
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                R2<RT> vn{m[i].vx, m[i].vy};
                if constexpr (NOISE) {
                    bool nd = true;
                    if constexpr (!JERK) {
                        const RT t0 = m[i].vx + C.dt * a[i].x, t1 = m[i].vy + C.dt * a[i].y, vl = C.v_max - C.band_v;
                        nd = !(t0 * t0 + t1 * t1 < vl * vl) | (vl <= (RT)0);
                    }
                    if (eager | (__ballot(nd & !frozen) != 0ull)) vn = noisy<RT, true>(vn, C.sig_v, key[i], 3u * (uint32_t)cyc);
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
                    qacc[i] = clamp_dyn<RT>(vn, C.v_max, a[i], C.dt, C.inv_dt, vnext);
                }
            }
            // mj_step (BE:1882): engine model gpr-push-v2. An env that has collided stands still (BE:1904-1905): its lanes sit the
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
                        const R2<RT> q0 = noisy<RT, true>(pm0, C.sig_p, key[0], 3u * (uint32_t)cyc + 2u);
                        const R2<RT> q1 = noisy<RT, true>(pm1, C.sig_p, key[NM - 1], 3u * (uint32_t)cyc + 2u);
                        hit = pair_hit2<RT>(C, q0, q1);
                    }
                }
            }
            if (!frozen) { wc = w; mc = hit; frozen = w | hit; }  // BE:1904-1905
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
