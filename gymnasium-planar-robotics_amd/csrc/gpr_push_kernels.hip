// gpr_push_kernels.hip -- fused BenchmarkPushingEnv step / reset kernels (one lane per env: 1 mover + 1 object).
//
// Pinned by the reference: control + clamp (PU:419-448), wall check (BE:426-788), obs / info / reward / terminated (PU:457-607),
// reset sampling (PU:373-417). The contact physics is the engine's own planar model "gpr-push-v1" (parity unpinned: the
// reference hands it to MuJoCo) -- see include/gpr.h and DESIGN.md.
#include "gpr_device.h"
#include "gpr_host.h"

namespace gpr {

template <typename RT> struct Body { RT x, y, th, vx, vy, w; };

template <typename RT> __device__ __forceinline__ RT cross2(RT ax, RT ay, RT bx, RT by) { return ax * by - ay * bx; }

// one control cycle of the two planar bodies (mirrors oracle/gpr_oracle.c push_cycle operation for operation; the float build
// takes its quotients through v_rcp_f32 and its sines / cosines from v_sin_f32 / v_cos_f32)
template <typename RT> __device__ __forceinline__ void push_cycle(const PushCst<RT> &Q, RT dt, Body<RT> &m, Body<RT> &o, RT ux, RT uy) {
    // 1. external forces
    m.vx = m.vx + dt * ux;
    m.vy = m.vy + dt * uy;
    m.w = Math<RT>::div_(m.w + dt * (Q.Kr * ((RT)0 - m.th)) * Q.inv_Im, (RT)1 + dt * Q.Dr * Q.inv_Im);
    o.vx = o.vx * Q.damp_lin; o.vy = o.vy * Q.damp_lin;
    o.w = Math<RT>::div_(o.w, Q.damp_rot_den);
    {
        const RT sp = Math<RT>::sqrt_(o.vx * o.vx + o.vy * o.vy);
        if (sp <= Q.fric_dv) { o.vx = (RT)0; o.vy = (RT)0; } else { const RT k = Math<RT>::div_(sp - Q.fric_dv, sp); o.vx = o.vx * k; o.vy = o.vy * k; }
        const RT aw = absr(o.w);
        if (aw <= Q.fric_dw) o.w = (RT)0; else o.w = o.w * Math<RT>::div_(aw - Q.fric_dw, aw);
    }
    // 2. contact: separating-axis test over the 4 face normals
    RT sm, cm, so, co;
    Math<RT>::sincos_(m.th, sm, cm);
    Math<RT>::sincos_(o.th, so, co);
    const RT ex4[4] = {cm, -sm, co, -so}, ey4[4] = {sm, cm, so, co};
    const RT dx = o.x - m.x, dy = o.y - m.y;
    RT best = Math<RT>::inf(), nx = (RT)0, ny = (RT)0;
    bool touching = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
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
    if (touching) {
        RT cx[9], cy[9], dep[9], jn[9], jt[9];
        int np = 0;
        const RT sx[4] = {-1, -1, 1, 1}, sy[4] = {-1, 1, 1, -1};
        const RT rm_n = Q.ax * absr(nx * cm + ny * sm) + Q.ay * absr(-nx * sm + ny * cm);
        const RT ro_n = Q.h * absr(nx * co + ny * so) + Q.h * absr(-nx * so + ny * co);
        for (int k = 0; k < 4; ++k) {
            const RT vx = o.x + co * (sx[k] * Q.h) - so * (sy[k] * Q.h), vy = o.y + so * (sx[k] * Q.h) + co * (sy[k] * Q.h);
            const RT lx = (vx - m.x) * cm + (vy - m.y) * sm, ly = -(vx - m.x) * sm + (vy - m.y) * cm;
            if (absr(lx) <= Q.ax && absr(ly) <= Q.ay) {
                cx[np] = vx; cy[np] = vy; dep[np] = rm_n - ((vx - m.x) * nx + (vy - m.y) * ny); jn[np] = (RT)0; jt[np] = (RT)0; ++np;
            }
        }
        for (int k = 0; k < 4; ++k) {
            const RT vx = m.x + cm * (sx[k] * Q.ax) - sm * (sy[k] * Q.ay), vy = m.y + sm * (sx[k] * Q.ax) + cm * (sy[k] * Q.ay);
            const RT lx = (vx - o.x) * co + (vy - o.y) * so, ly = -(vx - o.x) * so + (vy - o.y) * co;
            if (absr(lx) <= Q.h && absr(ly) <= Q.h) {
                cx[np] = vx; cy[np] = vy; dep[np] = ro_n + ((vx - o.x) * nx + (vy - o.y) * ny); jn[np] = (RT)0; jt[np] = (RT)0; ++np;
            }
        }
        if (np == 0) { cx[0] = m.x + (RT)0.5 * dx; cy[0] = m.y + (RT)0.5 * dy; dep[0] = best; jn[0] = (RT)0; jt[0] = (RT)0; np = 1; }
        const RT tx = -ny, ty = nx;
        for (int it = 0; it < 4; ++it) {
            for (int k = 0; k < np; ++k) {
                const RT rmx = cx[k] - m.x, rmy = cy[k] - m.y, rox = cx[k] - o.x, roy = cy[k] - o.y;
                RT rvx = (o.vx - o.w * roy) - (m.vx - m.w * rmy), rvy = (o.vy + o.w * rox) - (m.vy + m.w * rmx);
                const RT rmn = cross2(rmx, rmy, nx, ny), ron = cross2(rox, roy, nx, ny);
                const RT kn = Q.inv_mm + Q.inv_mo + rmn * rmn * Q.inv_Im + ron * ron * Q.inv_Io;
                const RT bias = Math<RT>::div_(dep[k] > (RT)0 ? dep[k] : (RT)0, Q.tc);
                RT dj = Math<RT>::div_(-((rvx * nx + rvy * ny) - bias), kn);
                RT jn_new = jn[k] + dj; if (jn_new < (RT)0) jn_new = (RT)0;
                dj = jn_new - jn[k]; jn[k] = jn_new;
                m.vx -= dj * nx * Q.inv_mm; m.vy -= dj * ny * Q.inv_mm; m.w -= dj * rmn * Q.inv_Im;
                o.vx += dj * nx * Q.inv_mo; o.vy += dj * ny * Q.inv_mo; o.w += dj * ron * Q.inv_Io;
                rvx = (o.vx - o.w * roy) - (m.vx - m.w * rmy); rvy = (o.vy + o.w * rox) - (m.vy + m.w * rmx);
                const RT rmt = cross2(rmx, rmy, tx, ty), rot_ = cross2(rox, roy, tx, ty);
                const RT kt = Q.inv_mm + Q.inv_mo + rmt * rmt * Q.inv_Im + rot_ * rot_ * Q.inv_Io;
                RT djt = Math<RT>::div_(-(rvx * tx + rvy * ty), kt);
                const RT lim = Q.mu * jn[k];
                RT jt_new = jt[k] + djt;
                if (jt_new > lim) jt_new = lim;
                if (jt_new < -lim) jt_new = -lim;
                djt = jt_new - jt[k]; jt[k] = jt_new;
                m.vx -= djt * tx * Q.inv_mm; m.vy -= djt * ty * Q.inv_mm; m.w -= djt * rmt * Q.inv_Im;
                o.vx += djt * tx * Q.inv_mo; o.vy += djt * ty * Q.inv_mo; o.w += djt * rot_ * Q.inv_Io;
            }
        }
    }
    // 3. positions with the new velocities
    m.x += dt * m.vx; m.y += dt * m.vy; m.th += dt * m.w;
    o.x += dt * o.vx; o.y += dt * o.vy; o.th += dt * o.w;
}

template <typename RT> struct PushObs {
    R2<RT> mpos, mvel, ag;
    RT reward;
    bool terminated, is_success;
};

// _get_obs (PU:529-576), _get_info (PU:578-607), compute_reward / compute_terminated (PU:457-527)
template <typename ST, typename RT, bool NOISE>
__device__ __forceinline__ PushObs<RT> push_observe(const KParams &P, const PushCst<RT> &Q, uint32_t key, uint32_t q0, R2<RT> p, R2<RT> v, RT ox, RT oy,
                                                    R2<RT> g, bool wc) {
    const Cst<RT> &C = cst<RT>(P);
    PushObs<RT> o;
    o.mpos = noisy<RT, NOISE>(p, C.sig_p, key, q0);
    o.mvel = noisy<RT, NOISE>(v, C.sig_v, key, q0 + 1u);
    o.ag = R2<RT>{ox, oy};
    if (Q.obj_noise != (RT)0) { const R2<RT> n = rng_normal_pair<RT>(key, q0 + 2u); o.ag.x = ox + Q.obj_noise * n.x; o.ag.y = oy + Q.obj_noise * n.y; }
    const RT dx = stored<ST, RT>(o.ag.x) - g.x, dy = stored<ST, RT>(o.ag.y) - g.y;
    const bool reached = Math<RT>::sqrt_(dx * dx + dy * dy) <= C.thr;  // PU:520
    RT r = (RT)-50.0 * (RT)wc;          // PU:521
    r += (RT)-1.0 * (RT)(!wc);          // PU:522
    if (reached & !wc) r = (RT)0;       // PU:523
    o.reward = r;
    o.terminated = r == (RT)-50.0;      // PU:475
    o.is_success = reached & !wc;       // PU:602
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

template <typename ST, typename RT, bool NOISE, bool FULL>
__device__ __forceinline__ PushObs<RT> push_reset_one(const KParams &P, const PushCst<RT> &Q, const Tables<RT> &T, int64_t env_global, RT c,
                                                      bool need, R2<RT> &p, R2<RT> &g, RT &ox, RT &oy, bool &wc) {
    const Cst<RT> &C = cst<RT>(P);
    const uint32_t ks = rng_key(P, KIND_RESET_START, env_global, 0u), kg = rng_key(P, KIND_RESET_GOAL, env_global, 0u);
    const uint32_t kn = rng_key(P, KIND_RESET_NOISE, env_global, 0u);
    p.x = stored<ST, RT>(C.min_x + (C.max_x - C.min_x) * rng_uniform<RT>(ks, 0u));  // PU:389-391
    p.y = stored<ST, RT>(C.min_y + (C.max_y - C.min_y) * rng_uniform<RT>(ks, 1u));
    // PU:396-407: object start by rejection, strict distance rule. Where the mover starts near the centre of the 3x3 tiles
    // almost no object position qualifies (the reference then loops for thousands of trials, at the very centre forever), so
    // the search is wave-cooperative: every lane tries its own first 8 trials; the lanes still looking are then served one
    // at a time, all 64 lanes testing 64 consecutive trials of that env, the first valid one in trial order wins -- the same
    // candidate the sequential loop accepts (the draws are indexed by trial). After max_reset_trials the last candidate stands.
    auto candidate = [&](uint32_t key, int trial, RT &cx, RT &cy) {
        cx = stored<ST, RT>(Q.omin_x + (Q.omax_x - Q.omin_x) * rng_uniform<RT>(key, 2u + 2u * (uint32_t)trial));
        cy = stored<ST, RT>(Q.omin_y + (Q.omax_y - Q.omin_y) * rng_uniform<RT>(key, 3u + 2u * (uint32_t)trial));
    };
    auto far_enough = [&](RT cx, RT cy, RT mx, RT my) { const RT dx = cx - mx, dy = cy - my; return Math<RT>::sqrt_(dx * dx + dy * dy) > Q.min_mo_dist; };
    constexpr int NSELF = 8;
    bool pending = need;
    for (int trial = 0; trial < NSELF && trial < P.max_reset_trials; ++trial) {
        if (__ballot(pending) == 0ull) break;
        RT cx, cy;
        candidate(ks, trial, cx, cy);
        if (pending) { ox = cx; oy = cy; pending = !far_enough(cx, cy, p.x, p.y); }
    }
    for (unsigned long long pm = __ballot(pending); pm != 0ull; pm &= pm - 1ull) {
        const int L = __builtin_ctzll(pm);
        const uint32_t ksL = (uint32_t)__builtin_amdgcn_readlane((int)ks, L);
        const RT mx = lane_bcast<RT>(p.x, L), my = lane_bcast<RT>(p.y, L);
        int found = P.max_reset_trials - 1;
        for (int base = NSELF; base < P.max_reset_trials; base += 64) {
            const int trial = base + (int)__lane_id();
            RT cx, cy;
            candidate(ksL, trial, cx, cy);
            const unsigned long long okm = __ballot((trial < P.max_reset_trials) & far_enough(cx, cy, mx, my));
            if (okm != 0ull) { found = base + __builtin_ctzll(okm); break; }
        }
        RT cx, cy;
        candidate(ksL, found, cx, cy);
        if ((int)__lane_id() == L) { ox = cx; oy = cy; }
    }
    g.x = stored<ST, RT>(Q.omin_x + (Q.omax_x - Q.omin_x) * rng_uniform<RT>(kg, 0u));  // PU:412-414
    g.y = stored<ST, RT>(Q.omin_y + (Q.omax_y - Q.omin_y) * rng_uniform<RT>(kg, 1u));
    const CircleSize<RT> cw_safe = circle_size<RT>(P, (c + C.c_offset_wall) + C.c_offset);
    const R2<RT> pw = noisy<RT, NOISE>(p, C.sig_p, kn, 0u);
    wc = !pos_valid_circle<RT, FULL>(T, P, pw.x, pw.y, cw_safe);  // BE:1799-1801
    return push_observe<ST, RT, NOISE>(P, Q, kn, 2u, p, R2<RT>{(RT)0, (RT)0}, ox, oy, g, wc);
}

template <typename ST, typename RT> __device__ __forceinline__ void push_store(const KParams &P, int64_t e, bool jerk, const PushObs<RT> &o,
                                                                                   R2<RT> acc, R2<RT> g, void *obs, void *ag, void *dg) {
    const int od = jerk ? 6 : 4;
    store2<ST, RT>(obs, e * od, o.mpos);
    store2<ST, RT>(obs, e * od + 2, o.mvel);
    if (jerk) store2<ST, RT>(obs, e * od + 4, acc);
    store2<ST, RT>(ag, e * 2, o.ag);
    store2<ST, RT>(dg, e * 2, g);
}

template <typename ST, typename RT, bool JERK, bool NOISE, bool FULL, bool RESET_ONLY>
__global__ void __launch_bounds__(BLOCK) k_push(const KParams P, const PushCst<RT> Q, const PushPtrs X) {
    extern __shared__ __align__(16) unsigned char lds[];
    const Cst<RT> &C = cst<RT>(P);
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool live = t < P.E;
    const int64_t e = live ? t : P.E - 1;
    const int64_t env_global = P.env_id_offset + e;
    if constexpr (!FULL) stage_tables(lds, P.tables, P.table_bytes);
    const Tables<RT> T = make_tables<RT>(lds, P);
    const RT c = make_tables<RT>(P.tables, P).csz[0];
    R2<RT> p = load2<ST, RT>(P.pos, 2 * e), v = load2<ST, RT>(P.vel, 2 * e), acc = load2<ST, RT>(P.acc, 2 * e), g = load2<ST, RT>(P.goal, 2 * e);
    R2<RT> rot = load2<ST, RT>(X.mover_rot, 2 * e);
    const ST *ob = static_cast<const ST *>(X.object) + 6 * e;
    Body<RT> o{(RT)ob[0], (RT)ob[1], (RT)ob[2], (RT)ob[3], (RT)ob[4], (RT)ob[5]};
    int el = P.elapsed[e];

    bool wc = false, trunc = false, need, reset_first = false;
    PushObs<RT> out;
    if constexpr (RESET_ONLY) {
        need = P.mask ? (P.mask[e] != 0) : true;
    } else {
        R2<RT> a = load2<ST, RT>(P.action, 2 * e);
        const RT lim = JERK ? C.j_max : C.a_max;
        a.x = fmin(fmax(a.x, -lim), lim);
        a.y = fmin(fmax(a.y, -lim), lim);
        reset_first = (P.autoreset == 2) & (el < 0);
        const uint32_t key = rng_key(P, KIND_STEP_NOISE, env_global, 0u);
        const CircleSize<RT> cw_step = circle_size<RT>(P, (c + C.c_offset_wall) + (RT)0 * C.c_offset);
        Body<RT> m{p.x, p.y, rot.x, v.x, v.y, rot.y};
        bool frozen = reset_first;
        const bool eager = P.eager_noise != 0;
        for (int cyc = 0; cyc < P.num_cycles; ++cyc) {
            // _mujoco_step_callback (PU:419-455): the same clamp chain as the planning env, on a noisy velocity read
            // lazy noise as in the planning kernels: a reading is drawn only if some lane's comparison could depend on it
            R2<RT> vn{m.vx, m.vy};
            if constexpr (NOISE) {
                bool need = true;
                if constexpr (!JERK) {
                    const RT t0 = m.vx + C.dt * a.x, t1 = m.vy + C.dt * a.y, lim = C.v_max - C.band_v;
                    need = !(t0 * t0 + t1 * t1 < lim * lim) | (lim <= (RT)0);
                }
                if (eager | (__ballot(need & !frozen) != 0ull)) vn = noisy<RT, true>(vn, C.sig_v, key, 3u * (uint32_t)cyc);
            }
            R2<RT> qacc, vnext;
            if constexpr (JERK) {
                R2<RT> acc_tmp;
                const R2<RT> j1 = clamp_dyn<RT>(acc, C.a_max, a, C.dt, C.inv_dt, acc_tmp);
                const R2<RT> acc_p = clamp_dyn<RT>(vn, C.v_max, acc_tmp, C.dt, C.inv_dt, vnext);
                const bool diff = (acc_tmp.x != acc_p.x) | (acc_tmp.y != acc_p.y);
                const R2<RT> u{diff ? (acc_p.x - acc.x) / C.dt : j1.x, diff ? (acc_p.y - acc.y) / C.dt : j1.y};
                qacc.x = acc.x + C.dt * u.x; qacc.y = acc.y + C.dt * u.y;
            } else {
                qacc = clamp_dyn<RT>(vn, C.v_max, a, C.dt, C.inv_dt, vnext);
            }
            Body<RT> m2 = m, o2 = o;
            push_cycle<RT>(Q, C.dt, m2, o2, qacc.x, qacc.y);  // mj_step (BE:1882): engine model gpr-push-v1
            if (!frozen) { m = m2; o = o2; acc = qacc; }
            bool w;  // BE:1888-1894
            if constexpr (NOISE) {
                bool unc;
                w = !pos_valid_circle<RT, FULL, true>(T, P, m.x, m.y, cw_step, C.band_p, &unc);
                if (eager | (__ballot(unc & !frozen) != 0ull)) {
                    const R2<RT> pw = noisy<RT, true>(R2<RT>{m.x, m.y}, C.sig_p, key, 3u * (uint32_t)cyc + 1u);
                    w = !pos_valid_circle<RT, FULL>(T, P, pw.x, pw.y, cw_step);
                }
            } else {
                w = !pos_valid_circle<RT, FULL>(T, P, m.x, m.y, cw_step);
            }
            if (!frozen) { wc = w; frozen = w; }                                    // BE:1904-1905
        }
        p = R2<RT>{stored<ST, RT>(m.x), stored<ST, RT>(m.y)}; v = R2<RT>{stored<ST, RT>(m.vx), stored<ST, RT>(m.vy)};
        rot = R2<RT>{stored<ST, RT>(m.th), stored<ST, RT>(m.w)};
        acc = R2<RT>{stored<ST, RT>(acc.x), stored<ST, RT>(acc.y)};
        o = Body<RT>{stored<ST, RT>(o.x), stored<ST, RT>(o.y), stored<ST, RT>(o.th), stored<ST, RT>(o.vx), stored<ST, RT>(o.vy), stored<ST, RT>(o.w)};
        out = push_observe<ST, RT, NOISE>(P, Q, key, 3u * (uint32_t)P.num_cycles, p, v, o.x, o.y, g, wc);
        if (!reset_first) el = el + 1;
        trunc = (P.max_episode_steps > 0) & (el >= P.max_episode_steps);
        const bool done = (out.terminated | trunc) & !reset_first;
        need = reset_first | ((P.autoreset == 1) & done);
        if (live & (P.autoreset == 1) & done & (P.final_obs != nullptr)) push_store<ST, RT>(P, e, JERK, out, acc, g, P.final_obs, P.final_ag, P.final_dg);
        if ((P.autoreset == 2) & done) el = -1;
    }
    if (__ballot(need) != 0ull) {  // pos_valid_circle votes: every lane takes part
        R2<RT> rp = p, rg = g;
        RT rox = o.x, roy = o.y;
        bool rwc;
        const PushObs<RT> ro = push_reset_one<ST, RT, NOISE, FULL>(P, Q, T, env_global, c, need, rp, rg, rox, roy, rwc);
        if (need) {
            p = rp; g = rg; v = R2<RT>{(RT)0, (RT)0}; acc = v; rot = v;
            o = Body<RT>{rox, roy, (RT)0, (RT)0, (RT)0, (RT)0};
            el = 0;
            if (RESET_ONLY | reset_first) { out = ro; out.reward = (RT)0; out.terminated = false; trunc = false; wc = rwc; }
            else { out.mpos = ro.mpos; out.mvel = ro.mvel; out.ag = ro.ag; }
        }
    }
    if (live & (RESET_ONLY ? need : true)) {
        store2<ST, RT>(P.pos, 2 * e, p); store2<ST, RT>(P.vel, 2 * e, v); store2<ST, RT>(P.acc, 2 * e, acc); store2<ST, RT>(P.goal, 2 * e, g);
        store2<ST, RT>(X.mover_rot, 2 * e, rot);
        ST *ow = static_cast<ST *>(X.object) + 6 * e;
        ow[0] = (ST)o.x; ow[1] = (ST)o.y; ow[2] = (ST)o.th; ow[3] = (ST)o.vx; ow[4] = (ST)o.vy; ow[5] = (ST)o.w;
        push_store<ST, RT>(P, e, JERK, out, acc, g, P.obs, P.ag, P.dg);
        P.elapsed[e] = el;
        static_cast<ST *>(P.reward)[e] = (ST)out.reward;
        P.terminated[e] = (uint8_t)out.terminated; P.truncated[e] = (uint8_t)trunc; P.is_success[e] = (uint8_t)out.is_success;
        P.mover_collision[e] = 0; P.wall_collision[e] = (uint8_t)wc;
    }
}

template <typename ST, typename RT, bool JERK, bool NOISE, bool FULL>
static hipError_t launch_push2(bool is_reset, const KParams &P, const PushCst<RT> &Q, const PushPtrs &X, hipStream_t s) {
    const unsigned grid = (unsigned)((P.E + BLOCK - 1) / BLOCK), lds = FULL ? 0u : P.table_bytes;
    if (is_reset) hipLaunchKernelGGL((k_push<ST, RT, JERK, NOISE, FULL, true>), dim3(grid), dim3(BLOCK), lds, s, P, Q, X);
    else hipLaunchKernelGGL((k_push<ST, RT, JERK, NOISE, FULL, false>), dim3(grid), dim3(BLOCK), lds, s, P, Q, X);
    return hipGetLastError();
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
template hipError_t launch_push<float, float>(bool, const KParams &, const PushCst<float> &, const PushPtrs &, hipStream_t);
template hipError_t launch_push<float, double>(bool, const KParams &, const PushCst<double> &, const PushPtrs &, hipStream_t);
template hipError_t launch_push<double, double>(bool, const KParams &, const PushCst<double> &, const PushPtrs &, hipStream_t);

}  // namespace gpr
