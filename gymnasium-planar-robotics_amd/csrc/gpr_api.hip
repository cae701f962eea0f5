// gpr_api.hip -- the C ABI of libgpr_hip.so (include/gpr.h): handle management, ctor-time tables, launch plumbing.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/gpr.h"
#include "gpr_host.h"
#include "gpr_launch_worker.h"

using namespace gpr;

namespace {

thread_local std::string g_create_error;

struct TableSet {
    void *dev = nullptr;
    uint32_t bytes = 0, off_ax = 0, off_ay = 0, off_lut = 0, off_nbr = 0, off_csz = 0, off_cx = 0, off_cy = 0, off_lutp = 0;
};

}  // namespace

struct gpr_env {
    gpr_config cfg{};
    std::vector<int8_t> layout;
    std::vector<double> xc, yc, csize;
    TableSet tab_f32, tab_f64;
    KParams base{};
    gpr_state state{};
    bool bound = false;
    uint64_t seed = 0, tick = 0;
    int lpe = 1;
    bool push_ready = false;
    size_t wave_timing_words = 0;
    int device = 0;  // the HIP device the handle was created on: every call must be made with it current
    // gpr_step_n: per-step parameter blocks. Two slots used in turn, each a pinned host array, a device array and an event recorded
    // behind the launch that reads the slot: a slot is rewritten only after that launch is complete, whatever stream it ran on
    struct StepSlot { KParams *host = nullptr, *dev = nullptr; hipEvent_t done = nullptr; int cap = 0; bool busy = false; };
    StepSlot step_slots[2];
    int step_slot_next = 0;
    uint32_t *counters = nullptr;  // [waves][GPR_NUM_COUNTERS] per-wavefront episode counters (gpr_enable_counters)
    size_t counter_rows = 0;
    bool counters_on = false;
    uint32_t *diag = nullptr;      // [4] device words: pending reset-exhausted / clipped counts, and what gpr_read_counters has moved out so far
    bool count_clipped = false;    // gpr_count_clipped_actions
    PushCst<float> push_f{};
    PushCst<double> push_d{};
    PushPtrs push_ptrs{};
    mutable std::string err;
};

namespace {

int fail(gpr_env *env, int code, const std::string &msg) {
    if (env) env->err = msg; else g_create_error = msg;
    return code;
}
int fail_hip(gpr_env *env, hipError_t e, const char *what) {
    return fail(env, GPR_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

// numpy.linspace(w/2, (n-1)*w + w/2, n) as get_tile_xy_pos uses it (BE:1292-1310); used when the caller passes no centres
std::vector<double> tile_centres(int n, double half) {
    std::vector<double> out((size_t)n);
    const double w = half * 2, start = w / 2, stop = (n - 1) * w + w / 2;
    const double step = n > 1 ? (stop - start) / (n - 1) : 0.0;
    for (int i = 0; i < n; ++i) out[(size_t)i] = i * step + start;
    if (n > 1) out[(size_t)n - 1] = stop;
    return out;
}

int tile_at(const gpr_env &e, int ix, int iy) {  // layout_tiles_wc (BE:221): anything outside the layout is "no tile"
    if (ix < 0 || iy < 0 || ix >= e.cfg.num_tiles_x || iy >= e.cfg.num_tiles_y) return 0;
    return e.layout[(size_t)ix * e.cfg.num_tiles_y + iy];
}

// mask_valid of BE:580-648 for one tile as a function of the four "safe" compares (bit0 min_x, bit1 max_x, bit2 min_y,
// bit3 max_y), stored as 16 x 2 bits. For a missing tile every compare is multiplied by layout == 0, so every entry is 0.
uint32_t tile_lut(const gpr_env &e, int ix, int iy) {
    const int L = tile_at(e, ix, iy);
    const int left = tile_at(e, ix - 1, iy), right = tile_at(e, ix + 1, iy), down = tile_at(e, ix, iy - 1), up = tile_at(e, ix, iy + 1);
    const int ld = tile_at(e, ix - 1, iy - 1), lu = tile_at(e, ix - 1, iy + 1), rd = tile_at(e, ix + 1, iy - 1), ru = tile_at(e, ix + 1, iy + 1);
    uint32_t lut = 0;
    for (int bits = 0; bits < 16; ++bits) {
        const int mnx = L * (bits & 1), mxx = L * ((bits >> 1) & 1), mny = L * ((bits >> 2) & 1), mxy = L * ((bits >> 3) & 1);
        int mv = mnx * mxx * mny * mxy;                  // BE:580
        const int u_mnx = (1 - mnx) * (L * left);        // BE:583-586 (index masks are implied: outside == no tile)
        mv += u_mnx * mny * mxy;                         // BE:587
        mv += u_mnx * ((1 - mny) * (L * down * ld));     // BE:589-597
        mv += u_mnx * ((1 - mxy) * (L * up * lu));       // BE:599-607
        const int u_mxx = (1 - mxx) * (L * right);       // BE:610-613
        mv += u_mxx * mny * mxy;                         // BE:614
        mv += u_mxx * ((1 - mny) * (L * down * rd));     // BE:616-624
        mv += u_mxx * ((1 - mxy) * (L * up * ru));       // BE:626-634
        mv += ((1 - mny) * (L * down)) * mnx * mxx;      // BE:637-641
        mv += ((1 - mxy) * (L * up)) * mnx * mxx;        // BE:644-648
        lut |= (uint32_t)(mv > 3 ? 3 : mv) << (2 * bits);
    }
    return lut;
}

template <typename RT> void put(std::vector<uint8_t> &blob, uint32_t off, size_t i, RT v) { std::memcpy(blob.data() + off + i * sizeof(RT), &v, sizeof(RT)); }

template <typename RT> int build_tables(gpr_env *e, TableSet &ts) {
    const int Tx = e->cfg.num_tiles_x, Ty = e->cfg.num_tiles_y;
    const double hx = e->cfg.tile_half_size[0], hy = e->cfg.tile_half_size[1];
    auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
    uint32_t off = 0;
    ts.off_ax = off; off = align16(off + (uint32_t)(Tx * 4 * sizeof(RT)));
    ts.off_ay = off; off = align16(off + (uint32_t)(Ty * 4 * sizeof(RT)));
    ts.off_lut = off; off = align16(off + (uint32_t)(Tx * Ty * 4));
    ts.off_nbr = off; off = align16(off + (uint32_t)(Tx * Ty * 2));
    ts.off_csz = off; off = align16(off + (uint32_t)(GPR_MAX_MOVERS * 2 * sizeof(RT)));
    ts.off_cx = off; off = align16(off + (uint32_t)(Tx * sizeof(RT)));
    ts.off_cy = off; off = align16(off + (uint32_t)(Ty * sizeof(RT)));
    ts.off_lutp = off; off = align16(off + (uint32_t)((Tx + 2) * (Ty + 2) * 4));
    ts.bytes = off;
    std::vector<uint8_t> blob(off, 0);
    const double inf = std::numeric_limits<double>::infinity();
    for (int i = 0; i < Tx; ++i) {  // edges exactly as BE:509-510 / BE:522-523 form them: centre -/+ half size, in f64
        put<RT>(blob, ts.off_ax, (size_t)4 * i + 0, (RT)(e->xc[(size_t)i] - hx));
        put<RT>(blob, ts.off_ax, (size_t)4 * i + 1, (RT)(e->xc[(size_t)i] + hx));
        put<RT>(blob, ts.off_ax, (size_t)4 * i + 2, (RT)(i > 0 ? e->xc[(size_t)i - 1] + hx : -inf));
        put<RT>(blob, ts.off_ax, (size_t)4 * i + 3, (RT)(i < Tx - 1 ? e->xc[(size_t)i + 1] - hx : inf));
        put<RT>(blob, ts.off_cx, (size_t)i, (RT)e->xc[(size_t)i]);
    }
    for (int i = 0; i < Ty; ++i) {
        put<RT>(blob, ts.off_ay, (size_t)4 * i + 0, (RT)(e->yc[(size_t)i] - hy));
        put<RT>(blob, ts.off_ay, (size_t)4 * i + 1, (RT)(e->yc[(size_t)i] + hy));
        put<RT>(blob, ts.off_ay, (size_t)4 * i + 2, (RT)(i > 0 ? e->yc[(size_t)i - 1] + hy : -inf));
        put<RT>(blob, ts.off_ay, (size_t)4 * i + 3, (RT)(i < Ty - 1 ? e->yc[(size_t)i + 1] - hy : inf));
        put<RT>(blob, ts.off_cy, (size_t)i, (RT)e->yc[(size_t)i]);
    }
    for (int ix = 0; ix < Tx; ++ix)
        for (int iy = 0; iy < Ty; ++iy) {
            put<uint32_t>(blob, ts.off_lut, (size_t)ix * Ty + iy, tile_lut(*e, ix, iy));
            put<uint32_t>(blob, ts.off_lutp, (size_t)(ix + 1) * (Ty + 2) + (iy + 1), tile_lut(*e, ix, iy));
            uint16_t n = 0;
            for (int dx = -1; dx <= 1; ++dx)
                for (int dy = -1; dy <= 1; ++dy)
                    if (tile_at(*e, ix + dx, iy + dy)) n |= (uint16_t)(1u << ((dx + 1) * 3 + (dy + 1)));
            put<uint16_t>(blob, ts.off_nbr, (size_t)ix * Ty + iy, n);
        }
    for (int i = 0; i < e->cfg.num_movers * 2; ++i) put<RT>(blob, ts.off_csz, (size_t)i, (RT)e->csize[(size_t)i]);
    hipError_t he = hipMalloc(&ts.dev, ts.bytes);
    if (he != hipSuccess) return fail_hip(e, he, "hipMalloc(tables)");
    he = hipMemcpy(ts.dev, blob.data(), ts.bytes, hipMemcpyHostToDevice);
    if (he != hipSuccess) return fail_hip(e, he, "hipMemcpy(tables)");
    return GPR_OK;
}

template <typename RT> void fill_cst(const gpr_env &e, Cst<RT> &c) {
    const gpr_config &g = e.cfg;
    c.dt = (RT)g.cycle_time; c.inv_dt = (RT)(1.0 / g.cycle_time);
    c.v_max = (RT)g.v_max; c.a_max = (RT)g.a_max; c.j_max = (RT)g.j_max; c.thr = (RT)g.threshold_pos;
    c.sig_p = (RT)g.std_noise[0]; c.sig_v = (RT)g.std_noise[1]; c.sig_a = (RT)g.std_noise[2];
    c.inv_wx = (RT)(1.0 / (2.0 * g.tile_half_size[0])); c.inv_wy = (RT)(1.0 / (2.0 * g.tile_half_size[1]));
    c.min_x = (RT)g.min_xy_pos[0]; c.min_y = (RT)g.min_xy_pos[1]; c.max_x = (RT)g.max_xy_pos[0]; c.max_y = (RT)g.max_xy_pos[1];
    c.min_goal_dist = (RT)g.min_goal_dist; c.reward_success = (RT)50.0;  // PL:220
    c.c_offset = (RT)g.c_offset; c.c_offset_wall = (RT)g.c_offset_wall;
    c.tile_hx = (RT)g.tile_half_size[0]; c.tile_hy = (RT)g.tile_half_size[1];
    // BE:390 + BE:409: largest (c_i + off) + (c_j + off) over mover pairs; -1 when there is no pair (never <= a distance)
    double ms = -1.0, mf = -1.0;
    for (int i = 0; i < g.num_movers - 1; ++i)
        for (int j = i + 1; j < g.num_movers; ++j) {
            double ci = e.csize[(size_t)2 * i], cj = e.csize[(size_t)2 * j];
            double s0 = (ci + g.c_offset * 0.0) + (cj + g.c_offset * 0.0), s1 = (ci + g.c_offset * 1.0) + (cj + g.c_offset * 1.0);
            ms = s0 > ms ? s0 : ms; mf = s1 > mf ? s1 : mf;
        }
    c.max_csum_step = (RT)ms; c.max_csum_safe = (RT)mf;
    // box shape: the edges of two rectangles cannot cross while their bounding boxes are apart
    double m0 = -1.0, m1 = -1.0;
    for (int i = 0; i < g.num_movers - 1; ++i)
        for (int j = i + 1; j < g.num_movers; ++j) {
            m0 = std::max(m0, e.csize[(size_t)2 * i] + e.csize[(size_t)2 * j]);
            m1 = std::max(m1, e.csize[(size_t)2 * i + 1] + e.csize[(size_t)2 * j + 1]);
        }
    c.max_b0sum = (RT)m0; c.max_b1sum = (RT)m1;
    // noise is bounded (gpr-rng-v2 draws the Box-Muller radius from 13 bits): |reading - truth| <= sigma * R per component.
    // The slack covers the rounding of the compared expressions (a few ulp of a coordinate of a few metres).
    const double slack = sizeof(RT) == 8 ? 1e-12 : 1e-5;
    c.band_p = (RT)(g.std_noise[0] * NORMAL_MAX_RADIUS * 1.001 + slack);
    c.band_v = (RT)(g.std_noise[1] * NORMAL_MAX_RADIUS * 1.001 + slack);
    // per-cycle displacement bound of one mover: after the clamp |v| <= v_max + band_v (PL:634-643), p moves by dt * v; slack
    // for the rounding of p = p0 + dp (a few ulp of a coordinate) and of dt * v
    c.step1 = (RT)(g.cycle_time * (g.v_max + (double)c.band_v) * 1.001 + (sizeof(RT) == 8 ? 1e-12 : 2e-6));
    c.gx0 = (RT)(e.xc.front() - g.tile_half_size[0]); c.gx1 = (RT)(e.xc.back() + g.tile_half_size[0]);
    c.gy0 = (RT)(e.yc.front() - g.tile_half_size[1]); c.gy1 = (RT)(e.yc.back() + g.tile_half_size[1]);
}

int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// gpr-rng-v2 in double: the radius table over the 13 radius bits of a draw and the (cos, sin) table over one octant of its 16 angle bits
// (gpr_device.h), filled with this host's libm exactly as oracle/gpr_oracle.c fills its own. One set per device, kept for the life of
// the process.
hipError_t ensure_normal_tables() {
    static std::vector<int> ready;
    int dev = 0;
    hipError_t he = hipGetDevice(&dev);
    if (he != hipSuccess) return he;
    for (int d : ready) if (d == dev) return hipSuccess;
    std::vector<double> h((size_t)8192 + 2 * 8193);
    for (uint32_t i = 0; i < 8192u; ++i) {
        const double u1 = (double)(i + 1u) * (1.0 / 8192.0);
        h[i] = std::sqrt(-2.0 * std::log(u1));
    }
    for (uint32_t j = 0; j <= 8192u; ++j) {
        const double u2 = (double)j * (1.0 / 65536.0);
        const double ang = 6.283185307179586476925286766559 * u2;
        h[(size_t)8192 + 2 * j] = std::cos(ang);
        h[(size_t)8192 + 2 * j + 1] = std::sin(ang);
    }
    double *d = nullptr;
    he = hipMalloc(reinterpret_cast<void **>(&d), h.size() * sizeof(double));
    if (he != hipSuccess) return he;
    he = hipMemcpy(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice);
    if (he == hipSuccess) he = set_normal_table_env<float, double>(d);
    if (he == hipSuccess) he = set_normal_table_env<double, double>(d);
    if (he == hipSuccess) he = set_normal_table_push(d);
    if (he == hipSuccess) he = set_normal_table_ops(d);
    if (he == hipSuccess) ready.push_back(dev);
    return he;
}

hipStream_t as_stream(void *s) { return static_cast<hipStream_t>(s); }

// Thin wavefronts: the most wavefronts a gpr_step launch is spread over by default. One wavefront per SIMD (1 024 on MI355X) for groups of
// up to 4 lanes per env, two for 8, four for 16 -- measured, us per step as one launch at budgets 0 / 1 024 / 2 048 / 4 096: 4 096 envs x
// 4 movers 10.9 / 9.3 / 9.6 / 11.2, 8 192 x 4: 11.1 / 10.1 / 10.3 / 12.1, 8 192 x 8: 15.9 / 15.9 / 14.8 / 16.4, 4 096 x 16: 52.9 / 53.0 /
// 42.6 / 40.0 (the 16-lane kernels share their reset searches over the block's wavefronts: more wavefronts, more searchers).
int32_t default_wave_budget(int lpe) {
    if (const char *tw = std::getenv("GPR_THIN_WAVES")) {
        const long v = std::atol(tw);
        return (int32_t)(v < 0 ? 0 : (v > 65536 ? 65536 : v));
    }
    return 1024 * (lpe >= 16 ? 4 : lpe >= 8 ? 2 : 1);
}

// The handle's tables, counters and kernel attributes live on ONE device; launching from a thread whose current device is another
// one would run the kernel there on pointers it cannot reach. Refuse instead of faulting.
int check_device(gpr_env *e, const char *who) {
    int dev = -1;
    const hipError_t he = hipGetDevice(&dev);
    if (he != hipSuccess) return fail_hip(e, he, who);
    if (dev != e->device)
        return fail(e, GPR_E_INVALID, std::string(who) + ": the handle was created on device " + std::to_string(e->device) + " but device " +
                                          std::to_string(dev) + " is current (hipSetDevice / torch.cuda.device before the call)");
    return GPR_OK;
}

KParams make_params(gpr_env *e, const gpr_step_out *out, bool is_f64_ops) {
    KParams P = e->base;
    const TableSet &ts = (is_f64_ops || e->cfg.precision != GPR_PRECISION_F32 || e->cfg.learn_jerk) ? e->tab_f64 : e->tab_f32;
    P.tables = ts.dev; P.table_bytes = ts.bytes;
    P.off_ax = ts.off_ax; P.off_ay = ts.off_ay; P.off_lut = ts.off_lut; P.off_nbr = ts.off_nbr; P.off_csz = ts.off_csz;
    P.off_cx = ts.off_cx; P.off_cy = ts.off_cy; P.off_lutp = ts.off_lutp;
    P.seed_lo = (uint32_t)e->seed; P.seed_hi = (uint32_t)(e->seed >> 32);
    P.tick_lo = (uint32_t)e->tick; P.tick_hi = (uint32_t)(e->tick >> 32);
    P.pos = e->state.pos_dev; P.vel = e->state.vel_dev; P.acc = e->state.acc_dev; P.goal = e->state.goal_dev;
    P.elapsed = e->state.elapsed_dev;
    P.counters = e->counters_on ? e->counters : nullptr;
    P.diag = e->diag; P.count_clipped = e->count_clipped ? 1 : 0;
    P.n_steps = 1;
    P.thin_shift = thin_shift_for(e->cfg.num_envs, e->lpe, e->base.thin_waves);  // (step kernels; launch_kernel clears it for resets)
    if (out) {
        P.obs = out->observation_dev; P.ag = out->achieved_goal_dev; P.dg = out->desired_goal_dev; P.reward = out->reward_dev;
        P.terminated = out->terminated_dev; P.truncated = out->truncated_dev; P.is_success = out->is_success_dev;
        P.mover_collision = out->mover_collision_dev; P.wall_collision = out->wall_collision_dev;
        P.final_obs = out->final_observation_dev; P.final_ag = out->final_achieved_goal_dev; P.final_dg = out->final_desired_goal_dev;
        P.agent_obs = out->agent_observation_dev; P.agent_reward = out->agent_reward_dev;
    }
    return P;
}

int check_out(gpr_env *e, const gpr_step_out *o) {
    if (!o) return fail(e, GPR_E_INVALID, "gpr_step_out is NULL");
    if (!o->observation_dev || !o->achieved_goal_dev || !o->desired_goal_dev || !o->reward_dev || !o->terminated_dev || !o->truncated_dev ||
        !o->is_success_dev || !o->mover_collision_dev || !o->wall_collision_dev)
        return fail(e, GPR_E_INVALID, "gpr_step_out: every non-final_* pointer must be set");
    const bool any_final = o->final_observation_dev || o->final_achieved_goal_dev || o->final_desired_goal_dev;
    const bool all_final = o->final_observation_dev && o->final_achieved_goal_dev && o->final_desired_goal_dev;
    if (any_final && !all_final) return fail(e, GPR_E_INVALID, "gpr_step_out: set all three final_* pointers or none");
    if ((o->agent_observation_dev != nullptr) != (o->agent_reward_dev != nullptr))
        return fail(e, GPR_E_INVALID, "gpr_step_out: set both agent_observation_dev and agent_reward_dev or neither");
    return GPR_OK;
}

int launch_env(gpr_env *e, bool is_reset, const KParams &P, hipStream_t s) {
    if (const int rc = check_device(e, is_reset ? "gpr_reset" : "gpr_step")) return rc;
    hipError_t he;
    switch (e->cfg.precision) {
        case GPR_PRECISION_F32:
            // jerk mode integrates in double even with float storage (see launch_flags in gpr_env_kernels.hip)
            he = e->cfg.learn_jerk ? launch_env_kernel<float, double>(is_reset, e->lpe, P, s)
                                   : launch_env_kernel<float, float>(is_reset, e->lpe, P, s);
            break;
        case GPR_PRECISION_F32_F64ACC: he = launch_env_kernel<float, double>(is_reset, e->lpe, P, s); break;
        default: he = launch_env_kernel<double, double>(is_reset, e->lpe, P, s); break;
    }
    if (he != hipSuccess) return fail_hip(e, he, is_reset ? "reset kernel launch" : "step kernel launch");
    return GPR_OK;
}

}  // namespace

namespace {
template <typename RT> void fill_push(const gpr_env &e, const gpr_push_params &pp, PushCst<RT> &q) {
    const double dt = e.cfg.cycle_time;
    const double ax = pp.mover_half_size[0], ay = pp.mover_half_size[1], h = pp.object_half_size;
    const double Im = pp.mover_mass * (ax * ax + ay * ay) / 3.0, Io = pp.object_mass * (2.0 * h * h) / 3.0;
    const double inv_mo = 1.0 / pp.object_mass, inv_Io = 1.0 / Io;
    q.inv_mm = (RT)(1.0 / pp.mover_mass); q.inv_mo = (RT)inv_mo; q.inv_Im = (RT)(1.0 / Im); q.inv_Io = (RT)inv_Io;
    q.ax = (RT)ax; q.ay = (RT)ay; q.h = (RT)h;
    q.Kr = (RT)pp.rot_stiffness; q.Dr = (RT)(2.0 * std::sqrt(pp.rot_stiffness * pp.mover_mass));
    q.damp_lin = (RT)(1.0 / (1.0 + dt * pp.object_damping * inv_mo)); q.damp_rot_den = (RT)(1.0 + dt * pp.object_damping * inv_Io);
    q.fric_dv = (RT)(pp.friction_tile * pp.gravity * dt);
    q.fric_dw = (RT)(pp.friction_torsional * pp.object_mass * pp.gravity * inv_Io * dt);
    q.mu = (RT)pp.friction_contact; q.tc = (RT)pp.contact_timeconst;
    q.obj_noise = (RT)pp.object_noise; q.min_mo_dist = (RT)pp.min_mo_dist;
    q.omin_x = (RT)pp.obj_min_xy[0]; q.omin_y = (RT)pp.obj_min_xy[1]; q.omax_x = (RT)pp.obj_max_xy[0]; q.omax_y = (RT)pp.obj_max_xy[1];
    const double reach = std::sqrt(ax * ax + ay * ay) + h * std::sqrt(2.0) + 1e-4;  // (+ the contact margin of gpr-push-v3, PUSH3_MARGIN)
    q.reach2 = (RT)(reach * reach);
}

int push_launch(gpr_env *env, bool is_reset, const KParams &P, hipStream_t s, const void *opt_start = nullptr, const void *opt_goal = nullptr,
                int32_t n_cand = 0, const void *cand = nullptr, int32_t *cand_used = nullptr) {
    if (const int rc = check_device(env, is_reset ? "gpr_push_reset" : "gpr_push_step")) return rc;
    if (P.agent_obs != nullptr) return fail(env, GPR_E_UNSUPPORTED, "the pushing env has no per-agent views (agent_observation_dev must be NULL)");
    hipError_t he;
    PushPtrs X = env->push_ptrs;
    X.opt_start = opt_start; X.opt_goal = opt_goal;
    X.cand = cand; X.cand_used = cand_used; X.n_cand = n_cand;
    switch (env->cfg.precision) {
        case GPR_PRECISION_F32:
            he = env->cfg.learn_jerk ? launch_push<float, double>(is_reset, P, env->push_d, X, s)
                                     : launch_push<float, float>(is_reset, P, env->push_f, X, s);
            break;
        case GPR_PRECISION_F32_F64ACC: he = launch_push<float, double>(is_reset, P, env->push_d, X, s); break;
        default: he = launch_push<double, double>(is_reset, P, env->push_d, X, s); break;
    }
    return he == hipSuccess ? GPR_OK : fail_hip(env, he, is_reset ? "push reset kernel launch" : "push step kernel launch");
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// LAUNCH WORKERS (gpr_step_many): gpr_launch_worker.h holds the worker itself (plain C++, also built under -fsanitize=thread against a
// stub launch by tests/model/launch_worker_stress.cpp). Workers are created on first use and never joined (a process that is exiting
// must not join threads that may be inside the HIP runtime); a worker that does not answer within GPR_LAUNCH_TIMEOUT_MS (default
// 30 000) is abandoned and replaced. GPR_LAUNCH_THREADS=0 keeps every launch on the calling thread.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
int run_launch_job(const LaunchJob &j) {
    thread_local int device = -1;  // (the current device is per thread)
    gpr_env *env = static_cast<gpr_env *>(j.env);
    if (env->device != device) {
        const hipError_t he = hipSetDevice(env->device);
        if (he != hipSuccess) return fail_hip(env, he, "gpr_step_many: hipSetDevice on a launch worker");
        device = env->device;
    }
    return gpr_step(env, j.action, static_cast<const gpr_step_out *>(j.out), j.stream);
}
constexpr int MAX_LAUNCH_WORKERS = 7;
constexpr int LAUNCH_SPIN_LIMIT = 20000;  // ~200 us of polling before a worker goes to sleep
LaunchWorker *g_launch_pool[MAX_LAUNCH_WORKERS] = {};
LaunchWorker *launch_worker(int i) {  // worker i, created on first use (called under gpr_step_many's pool lock)
    LaunchWorker *&w = g_launch_pool[i];
    if (w && w->lost()) w = nullptr;  // (the abandoned worker and its thread are leaked on purpose: it may still be inside the runtime)
    if (!w) {
        w = new LaunchWorker(run_launch_job, LAUNCH_SPIN_LIMIT);
        std::thread([p = w] { p->run(); }).detach();
    }
    return w;
}
bool launch_threads_enabled() {
    static const bool on = [] { const char *e = std::getenv("GPR_LAUNCH_THREADS"); return !(e && e[0] == '0'); }();
    return on;
}
long launch_timeout_ms() {
    static const long ms = [] { const char *e = std::getenv("GPR_LAUNCH_TIMEOUT_MS"); const long v = e ? std::atol(e) : 0; return v > 0 ? v : 30000L; }();
    return ms;
}
}  // namespace

// one wavefront that keeps a queue busy for `ticks` of the 100 MHz wall clock (gpr_streams_overlap)
__global__ void k_spin(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

extern "C" {

int gpr_streams_overlap(void *stream_a, void *stream_b, int32_t *overlap_out) {
    if (!overlap_out || stream_a == stream_b) return fail(nullptr, GPR_E_INVALID, "gpr_streams_overlap: two different streams and an output are needed");
    hipStream_t a = as_stream(stream_a), b = as_stream(stream_b);
    hipError_t he = hipStreamSynchronize(a);
    if (he == hipSuccess) he = hipStreamSynchronize(b);
    if (he != hipSuccess) return fail_hip(nullptr, he, "gpr_streams_overlap");
    constexpr long long TICKS = 15000;  // 150 us per kernel: together 150 us on two hardware queues, 300 us on one
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {  // (the first launch on a fresh stream also creates its queue)
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, TICKS);
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, TICKS);
        he = hipStreamSynchronize(a);
        if (he == hipSuccess) he = hipStreamSynchronize(b);
        if (he != hipSuccess) return fail_hip(nullptr, he, "gpr_streams_overlap");
        best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    *overlap_out = best < 235.0 ? 1 : 0;
    return GPR_OK;
}

const char *gpr_version(void) { return "gpr-hip 0.2 (gfx950; rng gpr-rng-v2; contact gpr-push-v3; one lane per mover)"; }

const char *gpr_last_error(const gpr_env *env) { return env ? env->err.c_str() : g_create_error.c_str(); }

int gpr_create(const gpr_config *cfg, gpr_env **out_env) {
    if (!cfg || !out_env) return fail(nullptr, GPR_E_INVALID, "gpr_create: NULL argument");
    *out_env = nullptr;
    const gpr_config &c = *cfg;
    if (c.num_envs < 1) return fail(nullptr, GPR_E_INVALID, "num_envs must be >= 1");
    if (c.num_movers < 1 || c.num_movers > GPR_MAX_MOVERS) return fail(nullptr, GPR_E_INVALID, "num_movers must be in [1, 16]");
    if (c.num_tiles_x < 1 || c.num_tiles_y < 1 || (int64_t)c.num_tiles_x * c.num_tiles_y > GPR_MAX_TILES)
        return fail(nullptr, GPR_E_INVALID, "tile layout must have between 1 and 4096 cells");
    if (!c.layout_tiles || !c.c_size) return fail(nullptr, GPR_E_INVALID, "layout_tiles and c_size are required");
    if (!(c.tile_half_size[0] > 0) || !(c.tile_half_size[1] > 0)) return fail(nullptr, GPR_E_INVALID, "tile_half_size must be > 0");
    if (c.c_shape != GPR_SHAPE_CIRCLE && c.c_shape != GPR_SHAPE_BOX) return fail(nullptr, GPR_E_INVALID, "c_shape must be circle or box");
    if (!(c.cycle_time > 0) || c.num_cycles < 0) return fail(nullptr, GPR_E_INVALID, "cycle_time must be > 0 and num_cycles >= 0");
    if (c.autoreset_mode < GPR_AUTORESET_DISABLED || c.autoreset_mode > GPR_AUTORESET_NEXT_STEP)
        return fail(nullptr, GPR_E_INVALID, "unknown autoreset_mode");
    if (c.precision < GPR_PRECISION_F32 || c.precision > GPR_PRECISION_F64) return fail(nullptr, GPR_E_INVALID, "unknown precision");
    if (c.max_reset_trials < 1) return fail(nullptr, GPR_E_INVALID, "max_reset_trials must be >= 1");
    for (int d = 0; d < 3; ++d)
        if (!(c.std_noise[d] >= 0)) return fail(nullptr, GPR_E_INVALID, "std_noise must be >= 0");
    int n_tiles = 0;
    for (int64_t i = 0; i < (int64_t)c.num_tiles_x * c.num_tiles_y; ++i) {
        if (c.layout_tiles[i] != 0 && c.layout_tiles[i] != 1) return fail(nullptr, GPR_E_INVALID, "layout_tiles must contain only 0 and 1");
        n_tiles += c.layout_tiles[i];
    }
    if (n_tiles == 0) return fail(nullptr, GPR_E_INVALID, "layout_tiles has no tile");

    {
        const hipError_t he = ensure_normal_tables();
        if (he != hipSuccess) return fail_hip(nullptr, he, "gpr_create: normal-draw tables");
    }
    gpr_env *e = new gpr_env();
    e->cfg = c;
    (void)hipGetDevice(&e->device);
    e->layout.assign(c.layout_tiles, c.layout_tiles + (size_t)c.num_tiles_x * c.num_tiles_y);
    e->xc = c.tile_x_centres ? std::vector<double>(c.tile_x_centres, c.tile_x_centres + c.num_tiles_x) : tile_centres(c.num_tiles_x, c.tile_half_size[0]);
    e->yc = c.tile_y_centres ? std::vector<double>(c.tile_y_centres, c.tile_y_centres + c.num_tiles_y) : tile_centres(c.num_tiles_y, c.tile_half_size[1]);
    e->csize.assign(c.c_size, c.c_size + (size_t)c.num_movers * 2);
    e->cfg.layout_tiles = e->layout.data();
    e->cfg.tile_x_centres = e->xc.data();
    e->cfg.tile_y_centres = e->yc.data();
    e->cfg.c_size = e->csize.data();
    e->lpe = next_pow2(c.num_movers);

    int rc = build_tables<float>(e, e->tab_f32);
    if (rc == GPR_OK) rc = build_tables<double>(e, e->tab_f64);
    if (rc != GPR_OK) {
        g_create_error = e->err;
        gpr_destroy(e);
        return rc;
    }
    if (e->tab_f64.bytes > 100u * 1024u) {  // (cannot happen with GPR_MAX_TILES = 4096: ~47 KB; the kernels stage the tables in LDS)
        g_create_error = "tile tables exceed the LDS budget of the step kernels";
        gpr_destroy(e);
        return GPR_E_UNSUPPORTED;
    }
    {
        hipError_t he = hipMalloc(reinterpret_cast<void **>(&e->diag), 4 * sizeof(uint32_t));
        if (he == hipSuccess) he = hipMemset(e->diag, 0, 4 * sizeof(uint32_t));
        if (he != hipSuccess) {
            g_create_error = std::string("gpr_create: diagnostics words: ") + hipGetErrorString(he);
            gpr_destroy(e);
            return GPR_E_HIP;
        }
    }
    KParams &P = e->base;
    P.E = c.num_envs; P.env_id_offset = c.env_id_offset;
    P.M = c.num_movers; P.Tx = c.num_tiles_x; P.Ty = c.num_tiles_y; P.c_shape = c.c_shape;
    P.num_cycles = c.num_cycles; P.max_episode_steps = c.max_episode_steps; P.autoreset = c.autoreset_mode;
    P.max_reset_trials = c.max_reset_trials; P.learn_jerk = c.learn_jerk != 0;
    P.noise = (c.std_noise[0] != 0.0) || (c.std_noise[1] != 0.0) || (c.std_noise[2] != 0.0);
    // "full layout" shortcut of the throughput-mode wall check: every cell holds a tile and every collision circle, with all
    // offsets added, is smaller than half a tile
    // (box shape: its circumradius; then every vertex stays within the neighbours of the centre's tile, a grid without holes has
    // no missing-corner case (BE:657-783), and BE:459-788 reduces to "all four vertices strictly inside the grid")
    bool full = n_tiles == c.num_tiles_x * c.num_tiles_y;
    for (int i = 0; i < c.num_movers; ++i) {
        const double o = c.c_offset_wall + c.c_offset;
        const double cs = c.c_shape == GPR_SHAPE_CIRCLE ? c.c_size[2 * i] + o : std::hypot(c.c_size[2 * i] + o, c.c_size[2 * i + 1] + o);
        full = full && cs < c.tile_half_size[0] && cs < c.tile_half_size[1];
    }
    P.layout_full = full;
    // validation switch: GPR_EAGER_NOISE=1 draws every noise sample of every cycle instead of only those that can matter
    const char *eager = std::getenv("GPR_EAGER_NOISE");
    P.eager_noise = eager && eager[0] == '1';
    // thin wavefronts (gpr_env_kernels.hip: step_body / launch_kernel): a gpr_step launch whose batch leaves SIMDs idle is spread over up
    // to this many wavefronts (default_wave_budget; gpr_set_wave_budget changes it, e.g. for handles that step side by side)
    P.thin_waves = default_wave_budget(e->lpe);
    P.thin_shift = 0;
    P.wave_timing = nullptr;
#ifdef GPR_WAVE_TIMING
    // debug build: per-wavefront clocks of the last step kernel, dumped to $GPR_WAVE_TIMING (a file name) by gpr_destroy
    if (std::getenv("GPR_WAVE_TIMING")) {
        e->wave_timing_words = (size_t)(std::max<int64_t>((c.num_envs * e->lpe + 63) / 64, (int64_t)P.thin_waves) + 8) * 16;
        if (hipMalloc(&P.wave_timing, e->wave_timing_words * 4) == hipSuccess) (void)hipMemset(P.wave_timing, 0, e->wave_timing_words * 4);
        else P.wave_timing = nullptr;
    }
#endif  // also runs both collision checks on every cycle (no skip ladder)
    fill_cst<float>(*e, P.cf);
    fill_cst<double>(*e, P.cd);
    *out_env = e;
    return GPR_OK;
}

void gpr_destroy(gpr_env *env) {
    if (!env) return;
#ifdef GPR_WAVE_TIMING
    if (env->base.wave_timing) {
        std::vector<unsigned int> h(env->wave_timing_words);
        (void)hipMemcpy(h.data(), env->base.wave_timing, h.size() * 4, hipMemcpyDeviceToHost);
        if (FILE *f = fopen(std::getenv("GPR_WAVE_TIMING"), "wb")) { fwrite(h.data(), 4, h.size(), f); fclose(f); }
        (void)hipFree(env->base.wave_timing);
    }
#endif
    if (env->counters) (void)hipFree(env->counters);
    if (env->diag) (void)hipFree(env->diag);
    for (auto &sl : env->step_slots) {
        if (sl.busy) (void)hipEventSynchronize(sl.done);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.dev) (void)hipFree(sl.dev);
    }
    if (env->tab_f32.dev) (void)hipFree(env->tab_f32.dev);
    if (env->tab_f64.dev) (void)hipFree(env->tab_f64.dev);
    delete env;
}

int gpr_bind_state(gpr_env *env, const gpr_state *state) {
    if (!env || !state) return fail(env, GPR_E_INVALID, "gpr_bind_state: NULL argument");
    if (!state->pos_dev || !state->vel_dev || !state->acc_dev || !state->goal_dev || !state->elapsed_dev)
        return fail(env, GPR_E_INVALID, "gpr_bind_state: every state pointer must be set");
    env->state = *state;
    env->bound = true;
    return GPR_OK;
}

int gpr_seed(gpr_env *env, uint64_t seed) {
    if (!env) return fail(nullptr, GPR_E_INVALID, "gpr_seed: NULL handle");
    env->seed = seed;
    env->tick = 0;
    return GPR_OK;
}

int gpr_get_rng_state(const gpr_env *env, uint64_t *seed_out, uint64_t *tick_out) {
    if (!env || !seed_out || !tick_out) return fail(nullptr, GPR_E_INVALID, "gpr_get_rng_state: NULL argument");
    *seed_out = env->seed;
    *tick_out = env->tick;
    return GPR_OK;
}

int gpr_set_rng_state(gpr_env *env, uint64_t seed, uint64_t tick) {
    if (!env) return fail(nullptr, GPR_E_INVALID, "gpr_set_rng_state: NULL handle");
    env->seed = seed;
    env->tick = tick;
    return GPR_OK;
}

int gpr_enable_counters(gpr_env *env, int32_t enable) {
    if (!env) return fail(nullptr, GPR_E_INVALID, "gpr_enable_counters: NULL handle");
    if (const int rc = check_device(env, "gpr_enable_counters")) return rc;
    if (enable) {
        // one row per wavefront of the step kernels' GRID -- whole blocks: the wavefronts of the last block that hold no env still
        // read and write their (zero) rows. Planning: lpe lanes per env; pushing: up to PUSH_LANES lanes per env (k_push).
        const int64_t lanes = env->push_ready ? std::max<int64_t>(env->lpe, PUSH_LANES) : env->lpe;
        // (a thin gpr_step launch has up to base.thin_waves wavefronts, rounded up to whole blocks)
        // (whole blocks of up to 512 threads: the widest block any step kernel is built with)
        const size_t rows = std::max((size_t)((env->cfg.num_envs * lanes + 511) / 512) * 8u, (size_t)env->base.thin_waves + 8u);
        if (env->counters && rows > env->counter_rows) {  // (gpr_push_configure came after the table was made: a larger one, counts restart)
            (void)hipDeviceSynchronize();
            (void)hipFree(env->counters);
            env->counters = nullptr;
        }
        if (!env->counters) {
            env->counter_rows = rows;
            const size_t bytes = env->counter_rows * GPR_NUM_COUNTERS * sizeof(uint32_t);
            hipError_t he = hipMalloc(reinterpret_cast<void **>(&env->counters), bytes);
            if (he != hipSuccess) { env->counters = nullptr; return fail_hip(env, he, "gpr_enable_counters: hipMalloc"); }
            he = hipMemset(env->counters, 0, bytes);
            if (he != hipSuccess) return fail_hip(env, he, "gpr_enable_counters: hipMemset");
        }
    }
    env->counters_on = enable != 0;
    return GPR_OK;
}

int gpr_set_wave_budget(gpr_env *env, int32_t max_wavefronts) {
    if (!env) return fail(nullptr, GPR_E_INVALID, "gpr_set_wave_budget: NULL handle");
    const int32_t w = max_wavefronts < 0 ? default_wave_budget(env->lpe) : (max_wavefronts > 65536 ? 65536 : max_wavefronts);
    if (env->counters && (size_t)w + 8u > env->counter_rows)  // (a launch's grid: the budget rounded up to whole blocks of up to 8 wavefronts)
        return fail(env, GPR_E_INVALID, "gpr_set_wave_budget: the counter table was made for fewer wavefronts (set the budget before gpr_enable_counters)");
#ifdef GPR_WAVE_TIMING
    if (env->base.wave_timing && (size_t)(w + 8) * 16 > env->wave_timing_words) return fail(env, GPR_E_INVALID, "gpr_set_wave_budget: beyond the timing table");
#endif
    env->base.thin_waves = w;
    return GPR_OK;
}

int gpr_read_counters(gpr_env *env, uint64_t *totals_dev, void *hip_stream) {
    if (!env || !totals_dev) return fail(env, GPR_E_INVALID, "gpr_read_counters: NULL argument");
    if (!env->counters) return fail(env, GPR_E_UNBOUND, "gpr_read_counters: call gpr_enable_counters first");
    if (const int rc = check_device(env, "gpr_read_counters")) return rc;
    hipError_t he = launch_counter_reduce(env->counters, (int64_t)env->counter_rows, totals_dev, env->diag, as_stream(hip_stream));
    return he == hipSuccess ? GPR_OK : fail_hip(env, he, "counter reduce kernel launch");
}

int gpr_count_clipped_actions(gpr_env *env, int32_t enable) {
    if (!env) return fail(nullptr, GPR_E_INVALID, "gpr_count_clipped_actions: NULL handle");
    env->count_clipped = enable != 0;
    return GPR_OK;
}

int gpr_read_diagnostics(gpr_env *env, uint64_t *totals_host, void *hip_stream) {
    if (!env || !totals_host) return fail(env, GPR_E_INVALID, "gpr_read_diagnostics: NULL argument");
    if (const int rc = check_device(env, "gpr_read_diagnostics")) return rc;
    uint32_t h[4] = {0, 0, 0, 0};
    // ordered behind the kernels of `hip_stream`; the host waits for them (this is a diagnostic read, not a hot-path call)
    hipError_t he = hipMemcpyAsync(h, env->diag, sizeof(h), hipMemcpyDeviceToHost, as_stream(hip_stream));
    if (he == hipSuccess) he = hipStreamSynchronize(as_stream(hip_stream));
    if (he != hipSuccess) return fail_hip(env, he, "gpr_read_diagnostics");
    totals_host[0] = (uint64_t)h[0] + h[2];
    totals_host[1] = (uint64_t)h[1] + h[3];
    return GPR_OK;
}

int gpr_get_mover_state(gpr_env *env, int32_t which, int32_t n_sel, const int32_t *mover_idx_dev, double z_report, int32_t add_noise, void *out_dev,
                        void *hip_stream) {
    if (!env) return fail(nullptr, GPR_E_INVALID, "gpr_get_mover_state: NULL handle");
    if (!env->bound) return fail(env, GPR_E_UNBOUND, "gpr_get_mover_state: call gpr_bind_state first");
    if (which < 0 || which > 2 || n_sel < 1 || !mover_idx_dev || !out_dev) return fail(env, GPR_E_INVALID, "gpr_get_mover_state: bad argument");
    if (const int rc = check_device(env, "gpr_get_mover_state")) return rc;
    KParams P = make_params(env, nullptr, true);
    const void *rot = env->push_ready ? env->push_ptrs.mover_rot : nullptr;
    hipError_t he;
    switch (env->cfg.precision) {
        case GPR_PRECISION_F32: he = launch_mover_state<float, float>(P, which, n_sel, mover_idx_dev, z_report, add_noise, rot, out_dev, as_stream(hip_stream)); break;
        case GPR_PRECISION_F32_F64ACC: he = launch_mover_state<float, double>(P, which, n_sel, mover_idx_dev, z_report, add_noise, rot, out_dev, as_stream(hip_stream)); break;
        default: he = launch_mover_state<double, double>(P, which, n_sel, mover_idx_dev, z_report, add_noise, rot, out_dev, as_stream(hip_stream)); break;
    }
    if (he != hipSuccess) return fail_hip(env, he, "mover state kernel launch");
    if (add_noise) env->tick += 1;  // (a noisy read consumes random numbers, like the reference's rng_noise.normal)
    return GPR_OK;
}

int gpr_reset(gpr_env *env, const uint8_t *mask_dev, const gpr_step_out *out, void *hip_stream) {
    if (!env) return fail(nullptr, GPR_E_INVALID, "gpr_reset: NULL handle");
    if (!env->bound) return fail(env, GPR_E_UNBOUND, "gpr_reset: call gpr_bind_state first");
    int rc = check_out(env, out);
    if (rc != GPR_OK) return rc;
    KParams P = make_params(env, out, false);
    P.mask = mask_dev;
    rc = launch_env(env, true, P, as_stream(hip_stream));
    if (rc == GPR_OK) env->tick += 1;
    return rc;
}

int gpr_reset_candidates(gpr_env *env, const uint8_t *mask_dev, int32_t n_cand, const void *cand_dev, int32_t *n_used_dev, const gpr_step_out *out,
                         void *hip_stream) {
    if (!env) return fail(nullptr, GPR_E_INVALID, "gpr_reset_candidates: NULL handle");
    if (!env->bound) return fail(env, GPR_E_UNBOUND, "gpr_reset_candidates: call gpr_bind_state first");
    if (n_cand < 2 || !cand_dev) return fail(env, GPR_E_INVALID, "gpr_reset_candidates: needs a stream of at least 2 candidate sets (one start set, one goal set)");
    int rc = check_out(env, out);
    if (rc != GPR_OK) return rc;
    KParams P = make_params(env, out, false);
    P.mask = mask_dev;
    P.cand = cand_dev; P.cand_used = n_used_dev; P.n_cand = n_cand;
    rc = launch_env(env, true, P, as_stream(hip_stream));
    if (rc == GPR_OK) env->tick += 1;
    return rc;
}

int gpr_step(gpr_env *env, const void *action_dev, const gpr_step_out *out, void *hip_stream) {
    if (!env) return fail(nullptr, GPR_E_INVALID, "gpr_step: NULL handle");
    if (!env->bound) return fail(env, GPR_E_UNBOUND, "gpr_step: call gpr_bind_state first");
    if (!action_dev) return fail(env, GPR_E_INVALID, "gpr_step: action is NULL");
    int rc = check_out(env, out);
    if (rc != GPR_OK) return rc;
    KParams P = make_params(env, out, false);
    P.action = action_dev;
    rc = launch_env(env, false, P, as_stream(hip_stream));
    if (rc == GPR_OK) env->tick += 1;
    return rc;
}

int gpr_step_many(int32_t n, gpr_env *const *envs, const void *const *actions_dev, const gpr_step_out *const *outs, void *const *hip_streams) {
    if (n < 1 || !envs || !actions_dev || !outs || !hip_streams) return fail(nullptr, GPR_E_INVALID, "gpr_step_many: bad argument");
    for (int32_t i = 0; i < n; ++i) {  // validate everything before anything is launched
        if (!envs[i]) return fail(nullptr, GPR_E_INVALID, "gpr_step_many: NULL handle");
        if (!envs[i]->bound) return fail(envs[i], GPR_E_UNBOUND, "gpr_step_many: call gpr_bind_state first");
        if (!actions_dev[i]) return fail(envs[i], GPR_E_INVALID, "gpr_step_many: action is NULL");
        const int rc = check_out(envs[i], outs[i]);
        if (rc != GPR_OK) return rc;
        for (int32_t j = 0; j < i; ++j)  // (two launches of one handle would run on two threads and race on its call counter)
            if (envs[j] == envs[i]) return fail(envs[i], GPR_E_INVALID, "gpr_step_many: the same handle appears twice");
    }
    if (n == 1 || !launch_threads_enabled()) {
        for (int32_t i = 0; i < n; ++i) {
            const int rc = gpr_step(envs[i], actions_dev[i], outs[i], hip_streams[i]);
            if (rc != GPR_OK) return rc;
        }
        return GPR_OK;
    }
    // launches 1 .. n-1 go to the launch workers (gpr_launch_worker.h), launch 0 is issued here; handles beyond the pool stay on this thread
    static std::mutex pool_lock;  // (one gpr_step_many at a time per process uses the workers)
    std::lock_guard<std::mutex> lk(pool_lock);
    for (int32_t i = 0; i < n; ++i) {
        const int rc = check_device(envs[i], "gpr_step_many");  // (the workers set their own current device: check the caller's here)
        if (rc != GPR_OK) return rc;
    }
    const int32_t n_workers = std::min<int32_t>(n - 1, MAX_LAUNCH_WORKERS);
    for (int32_t w = 0; w < n_workers; ++w)
        launch_worker(w)->post(LaunchJob{envs[1 + w], actions_dev[1 + w], outs[1 + w], hip_streams[1 + w], GPR_OK});
    int rc = gpr_step(envs[0], actions_dev[0], outs[0], hip_streams[0]);
    for (int32_t i = 1 + n_workers; i < n; ++i) {
        const int r = gpr_step(envs[i], actions_dev[i], outs[i], hip_streams[i]);
        if (rc == GPR_OK) rc = r;
    }
    for (int32_t w = 0; w < n_workers; ++w) {
        int r = launch_worker(w)->wait(launch_timeout_ms());
        if (r == LaunchWorker::WAIT_TIMED_OUT)  // (the worker is abandoned and replaced at the next call; its launch may or may not have been issued)
            r = fail(envs[1 + w], GPR_E_HIP, "gpr_step_many: launch worker " + std::to_string(w) + " did not return within " +
                                                 std::to_string(launch_timeout_ms()) + " ms (GPR_LAUNCH_TIMEOUT_MS); GPR_LAUNCH_THREADS=0 issues every launch on the calling thread");
        if (rc == GPR_OK) rc = r;
    }
    return rc;
}

int gpr_step_n(gpr_env *env, int32_t n_steps, const void *action_dev, const gpr_step_out *out, void *hip_stream) {
    if (!env) return fail(nullptr, GPR_E_INVALID, "gpr_step_n: NULL handle");
    if (!env->bound) return fail(env, GPR_E_UNBOUND, "gpr_step_n: call gpr_bind_state first");
    if (!action_dev || n_steps < 1) return fail(env, GPR_E_INVALID, "gpr_step_n: action is NULL or n_steps < 1");
    if (env->cfg.autoreset_mode == GPR_AUTORESET_DISABLED && n_steps > 1)
        return fail(env, GPR_E_INVALID, "gpr_step_n: several steps per launch need an autoreset mode (a finished env must go on)");
    int rc = check_out(env, out);
    if (rc != GPR_OK) return rc;
    rc = check_device(env, "gpr_step_n");  // before anything is allocated for the handle: the slots below live on the current device
    if (rc != GPR_OK) return rc;
    if (out->desired_goal_dev == env->state.goal_dev && n_steps > 1)
        return fail(env, GPR_E_INVALID, "gpr_step_n: desired_goal must not alias the goal state (every step keeps its own)");
    KParams P = make_params(env, out, false);
    P.action = action_dev;
    if (n_steps == 1) {
        rc = launch_env(env, false, P, as_stream(hip_stream));
        if (rc == GPR_OK) env->tick += 1;
        return rc;
    }
    const int64_t E = env->cfg.num_envs, M = env->cfg.num_movers;
    const int64_t sz = env->cfg.precision == GPR_PRECISION_F64 ? 8 : 4, obs_dim = 2 * M * (env->cfg.learn_jerk ? 2 : 1);
    const int64_t ss_action = E * 2 * M * sz, ss_obs = E * obs_dim * sz, ss_goal = E * 2 * M * sz, ss_reward = E * sz, ss_flag = E;
    // one parameter block per step: row i of every tensor, call counter tick + i
    gpr_env::StepSlot &sl = env->step_slots[env->step_slot_next];
    env->step_slot_next ^= 1;
    if (sl.busy) {  // the launch that last read this slot (two calls ago) must be through before the slot is rewritten
        const hipError_t he = hipEventSynchronize(sl.done);
        if (he != hipSuccess) return fail_hip(env, he, "gpr_step_n: hipEventSynchronize");
        sl.busy = false;
    }
    if (sl.cap < n_steps) {
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.dev) (void)hipFree(sl.dev);
        sl.host = sl.dev = nullptr; sl.cap = 0;
        hipError_t he = hipHostMalloc(reinterpret_cast<void **>(&sl.host), sizeof(KParams) * (size_t)n_steps, hipHostMallocDefault);
        if (he == hipSuccess) he = hipMalloc(reinterpret_cast<void **>(&sl.dev), sizeof(KParams) * (size_t)n_steps);
        if (he == hipSuccess && !sl.done) he = hipEventCreateWithFlags(&sl.done, hipEventDisableTiming);
        if (he != hipSuccess) return fail_hip(env, he, "gpr_step_n: parameter staging");
        sl.cap = n_steps;
    }
    for (int i = 0; i < n_steps; ++i) sl.host[i] = P;
    auto adv = [](auto *&ptr, int64_t bytes) { ptr = reinterpret_cast<std::remove_reference_t<decltype(ptr)>>(reinterpret_cast<uintptr_t>(ptr) + (uintptr_t)bytes); };
    for (int i = 1; i < n_steps; ++i) {
        KParams &Q = sl.host[i];
        adv(Q.action, i * ss_action);
        adv(Q.obs, i * ss_obs); adv(Q.ag, i * ss_goal); adv(Q.dg, i * ss_goal); adv(Q.reward, i * ss_reward);
        adv(Q.terminated, i * ss_flag); adv(Q.truncated, i * ss_flag); adv(Q.is_success, i * ss_flag);
        adv(Q.mover_collision, i * ss_flag); adv(Q.wall_collision, i * ss_flag);
        if (Q.final_obs) { adv(Q.final_obs, i * ss_obs); adv(Q.final_ag, i * ss_goal); adv(Q.final_dg, i * ss_goal); }
        if (Q.agent_obs) { adv(Q.agent_obs, i * E * M * (4 * M + 2) * sz); adv(Q.agent_reward, i * E * M * sz); }
        const uint64_t tick = env->tick + (uint64_t)i;
        Q.tick_lo = (uint32_t)tick; Q.tick_hi = (uint32_t)(tick >> 32);
    }
    // pinned source: a true asynchronous copy, ordered on the caller's stream in front of the launch that reads the device array
    hipError_t he = hipMemcpyAsync(sl.dev, sl.host, sizeof(KParams) * (size_t)n_steps, hipMemcpyHostToDevice, as_stream(hip_stream));
    if (he != hipSuccess) return fail_hip(env, he, "gpr_step_n: parameter upload");
    P.n_steps = n_steps;
    P.step_params = sl.dev;
    rc = launch_env(env, false, P, as_stream(hip_stream));
    if (rc == GPR_OK) env->tick += (uint64_t)n_steps;
    // the upload is enqueued whether or not the launch went out, and it reads sl.host: track it (or wait for it here)
    if (hipEventRecord(sl.done, as_stream(hip_stream)) == hipSuccess) sl.busy = true;
    else (void)hipStreamSynchronize(as_stream(hip_stream));
    return rc;
}

int gpr_compute_reward(gpr_env *env, int64_t batch, const void *achieved_dev, const void *desired_dev, const uint8_t *mover_collision_dev,
                       const uint8_t *wall_collision_dev, void *reward_dev, uint8_t *terminated_dev, void *hip_stream) {
    if (!env) return fail(nullptr, GPR_E_INVALID, "gpr_compute_reward: NULL handle");
    if (batch == 0) return GPR_OK;
    if (batch < 0 || !achieved_dev || !desired_dev || !mover_collision_dev || !wall_collision_dev || !reward_dev)
        return fail(env, GPR_E_INVALID, "gpr_compute_reward: bad argument");
    if (const int drc = check_device(env, "gpr_compute_reward")) return drc;
    KParams P = make_params(env, nullptr, true);
    hipError_t he = env->cfg.precision == GPR_PRECISION_F64
                        ? launch_reward<double>(P, batch, achieved_dev, desired_dev, mover_collision_dev, wall_collision_dev, reward_dev, terminated_dev, as_stream(hip_stream))
                        : launch_reward<float>(P, batch, achieved_dev, desired_dev, mover_collision_dev, wall_collision_dev, reward_dev, terminated_dev, as_stream(hip_stream));
    return he == hipSuccess ? GPR_OK : fail_hip(env, he, "reward kernel launch");
}

int gpr_agent_views(gpr_env *env, const gpr_step_out *out, void *agent_obs_dev, void *agent_reward_dev, void *hip_stream) {
    if (!env) return fail(nullptr, GPR_E_INVALID, "gpr_agent_views: NULL handle");
    int rc = check_out(env, out);
    if (rc != GPR_OK) return rc;
    if (!agent_obs_dev || !agent_reward_dev) return fail(env, GPR_E_INVALID, "gpr_agent_views: bad argument");
    if (env->cfg.autoreset_mode == GPR_AUTORESET_SAME_STEP && !out->final_achieved_goal_dev)
        return fail(env, GPR_E_INVALID, "gpr_agent_views: same-step autoreset needs final_achieved_goal / final_desired_goal bound in `out` "
                                        "(the per-agent reward of an env that ended is computed from its terminal state)");
    if (const int drc = check_device(env, "gpr_agent_views")) return drc;
    KParams P = make_params(env, out, true);
    const int obs_dim = env->cfg.num_movers * 2 * (env->cfg.learn_jerk ? 2 : 1);
    hipError_t he = env->cfg.precision == GPR_PRECISION_F64
                        ? launch_agent_views<double>(P, env->lpe, obs_dim, agent_obs_dev, agent_reward_dev, as_stream(hip_stream))
                        : launch_agent_views<float>(P, env->lpe, obs_dim, agent_obs_dev, agent_reward_dev, as_stream(hip_stream));
    return he == hipSuccess ? GPR_OK : fail_hip(env, he, "agent views kernel launch");
}

int gpr_push_configure(gpr_env *env, const gpr_push_params *pp, const gpr_push_state *st) {
    if (!env || !pp || !st) return fail(env, GPR_E_INVALID, "gpr_push_configure: NULL argument");
    if (env->cfg.num_movers < 1 || env->cfg.num_movers > 2 || env->cfg.c_shape != GPR_SHAPE_CIRCLE)
        return fail(env, GPR_E_INVALID, "gpr_push_configure: BenchmarkPushingEnv has one mover (PU:195-196; the engine also offers two) with a "
                                        "circular collision shape");
    if (!(pp->mover_mass > 0) || !(pp->object_mass > 0) || !(pp->object_half_size > 0) || !(pp->mover_half_size[0] > 0) ||
        !(pp->mover_half_size[1] > 0) || !(pp->contact_timeconst > 0) || !(pp->object_damping >= 0) || !(pp->gravity >= 0))
        return fail(env, GPR_E_INVALID, "gpr_push_configure: masses, sizes and the contact time constant must be positive");
    if (!st->mover_rot_dev || !st->object_dev) return fail(env, GPR_E_INVALID, "gpr_push_configure: state pointers must be set");
    fill_push<float>(*env, *pp, env->push_f);
    fill_push<double>(*env, *pp, env->push_d);
    env->push_ptrs = PushPtrs{st->mover_rot_dev, st->object_dev, nullptr, nullptr, nullptr, nullptr, 0, 0};
    env->push_ready = true;
    if (env->counters && env->counters_on) return gpr_enable_counters(env, 1);  // (the pushing grid is wider than the table was made for)
    return GPR_OK;
}

int gpr_push_reset_opts(gpr_env *env, const uint8_t *mask_dev, const void *mover_start_xy_dev, const void *object_goal_xy_dev,
                        const gpr_step_out *out, void *hip_stream) {
    if (!env) return fail(nullptr, GPR_E_INVALID, "gpr_push_reset: NULL handle");
    if (!env->bound || !env->push_ready) return fail(env, GPR_E_UNBOUND, "gpr_push_reset: call gpr_bind_state and gpr_push_configure first");
    int rc = check_out(env, out);
    if (rc != GPR_OK) return rc;
    KParams P = make_params(env, out, false);
    P.mask = mask_dev;
    rc = push_launch(env, true, P, as_stream(hip_stream), mover_start_xy_dev, object_goal_xy_dev);
    if (rc == GPR_OK) env->tick += 1;
    return rc;
}

int gpr_push_reset_candidates(gpr_env *env, const uint8_t *mask_dev, const void *mover_start_xy_dev, const void *object_goal_xy_dev, int32_t n_cand,
                              const void *object_cand_dev, int32_t *n_used_dev, const gpr_step_out *out, void *hip_stream) {
    if (!env) return fail(nullptr, GPR_E_INVALID, "gpr_push_reset_candidates: NULL handle");
    if (!env->bound || !env->push_ready) return fail(env, GPR_E_UNBOUND, "gpr_push_reset_candidates: call gpr_bind_state and gpr_push_configure first");
    if (n_cand < 1 || !object_cand_dev) return fail(env, GPR_E_INVALID, "gpr_push_reset_candidates: needs at least one object candidate");
    int rc = check_out(env, out);
    if (rc != GPR_OK) return rc;
    KParams P = make_params(env, out, false);
    P.mask = mask_dev;
    rc = push_launch(env, true, P, as_stream(hip_stream), mover_start_xy_dev, object_goal_xy_dev, n_cand, object_cand_dev, n_used_dev);
    if (rc == GPR_OK) env->tick += 1;
    return rc;
}

int gpr_push_reset(gpr_env *env, const uint8_t *mask_dev, const gpr_step_out *out, void *hip_stream) {
    return gpr_push_reset_opts(env, mask_dev, nullptr, nullptr, out, hip_stream);
}

int gpr_push_step(gpr_env *env, const void *action_dev, const gpr_step_out *out, void *hip_stream) {
    if (!env) return fail(nullptr, GPR_E_INVALID, "gpr_push_step: NULL handle");
    if (!env->bound || !env->push_ready) return fail(env, GPR_E_UNBOUND, "gpr_push_step: call gpr_bind_state and gpr_push_configure first");
    if (!action_dev) return fail(env, GPR_E_INVALID, "gpr_push_step: action is NULL");
    int rc = check_out(env, out);
    if (rc != GPR_OK) return rc;
    KParams P = make_params(env, out, false);
    P.action = action_dev;
    rc = push_launch(env, false, P, as_stream(hip_stream));
    if (rc == GPR_OK) env->tick += 1;
    return rc;
}

int gpr_qpos_is_valid(gpr_env *env, int64_t n, const double *qpos_dev, const double *c_size_dev, int32_t add_safety_offset, int32_t *out_dev,
                      void *hip_stream) {
    if (!env) return fail(nullptr, GPR_E_INVALID, "gpr_qpos_is_valid: NULL handle");
    if (n == 0) return GPR_OK;
    if (n < 0 || !qpos_dev || !c_size_dev || !out_dev) return fail(env, GPR_E_INVALID, "gpr_qpos_is_valid: bad argument");
    if (const int drc = check_device(env, "gpr_qpos_is_valid")) return drc;
    KParams P = make_params(env, nullptr, true);
    hipError_t he = launch_qpos_valid(P, n, qpos_dev, c_size_dev, add_safety_offset, out_dev, as_stream(hip_stream));
    return he == hipSuccess ? GPR_OK : fail_hip(env, he, "qpos_is_valid kernel launch");
}

int gpr_check_mover_collision(gpr_env *env, int64_t batch, const double *qpos_dev, const double *c_size_dev, int32_t add_safety_offset,
                              uint8_t *out_dev, void *hip_stream) {
    if (!env) return fail(nullptr, GPR_E_INVALID, "gpr_check_mover_collision: NULL handle");
    if (batch == 0) return GPR_OK;
    if (batch < 0 || !qpos_dev || !c_size_dev || !out_dev) return fail(env, GPR_E_INVALID, "gpr_check_mover_collision: bad argument");
    if (const int drc = check_device(env, "gpr_check_mover_collision")) return drc;
    KParams P = make_params(env, nullptr, true);
    hipError_t he = launch_mover_collision(P, batch, qpos_dev, c_size_dev, add_safety_offset, out_dev, as_stream(hip_stream));
    return he == hipSuccess ? GPR_OK : fail_hip(env, he, "mover collision kernel launch");
}

int gpr_check_line_segments_intersect(int64_t n, const double *p1_dev, const double *p2_dev, const double *q1_dev, const double *q2_dev,
                                      uint8_t *out_dev, void *hip_stream) {
    if (n == 0) return GPR_OK;
    if (n < 0 || !p1_dev || !p2_dev || !q1_dev || !q2_dev || !out_dev) return fail(nullptr, GPR_E_INVALID, "gpr_check_line_segments_intersect: bad argument");
    hipError_t he = launch_segments(n, p1_dev, p2_dev, q1_dev, q2_dev, out_dev, as_stream(hip_stream));
    return he == hipSuccess ? GPR_OK : fail_hip(nullptr, he, "segments kernel launch");
}

int gpr_get_2D_rect_vertices(int64_t n, const double *qpos_dev, const double *size_dev, double *out_dev, void *hip_stream) {
    if (n == 0) return GPR_OK;
    if (n < 0 || !qpos_dev || !size_dev || !out_dev) return fail(nullptr, GPR_E_INVALID, "gpr_get_2D_rect_vertices: bad argument");
    hipError_t he = launch_rect_vertices(n, qpos_dev, size_dev, out_dev, as_stream(hip_stream));
    return he == hipSuccess ? GPR_OK : fail_hip(nullptr, he, "rect vertices kernel launch");
}

int gpr_check_rectangles_intersect(int64_t n, const double *qpos_r1_dev, const double *qpos_r2_dev, const double *size_r1_dev,
                                   const double *size_r2_dev, uint8_t *out_dev, void *hip_stream) {
    if (n == 0) return GPR_OK;
    if (n < 0 || !qpos_r1_dev || !qpos_r2_dev || !size_r1_dev || !size_r2_dev || !out_dev)
        return fail(nullptr, GPR_E_INVALID, "gpr_check_rectangles_intersect: bad argument");
    hipError_t he = launch_rects(n, qpos_r1_dev, qpos_r2_dev, size_r1_dev, size_r2_dev, out_dev, as_stream(hip_stream));
    return he == hipSuccess ? GPR_OK : fail_hip(nullptr, he, "rectangles kernel launch");
}

int gpr_ensure_max_dyn_val(int64_t n, const double *cur_dev, const double *max_value_dev, const double *deriv_dev, double cycle_time,
                           double *out_vals_dev, double *out_derivs_dev, void *hip_stream) {
    if (n == 0) return GPR_OK;
    if (n < 0 || !cur_dev || !max_value_dev || !deriv_dev || !out_vals_dev || !out_derivs_dev || !(cycle_time > 0))
        return fail(nullptr, GPR_E_INVALID, "gpr_ensure_max_dyn_val: bad argument");
    hipError_t he = launch_clamp(n, cur_dev, max_value_dev, deriv_dev, cycle_time, out_vals_dev, out_derivs_dev, as_stream(hip_stream));
    return he == hipSuccess ? GPR_OK : fail_hip(nullptr, he, "clamp kernel launch");
}

}  // extern "C"
