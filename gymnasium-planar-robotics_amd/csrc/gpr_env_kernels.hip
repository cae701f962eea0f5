// gpr_env_kernels.hip -- fused BenchmarkPlanningEnv step / reset kernels for gfx950 (MI355X).
//
// One launch = one env-step of every env: clip -> num_cycles x [noisy velocity read, clamp, semi-implicit Euler, wall
// check, pair sweep, early-out freeze] -> observation / info / reward / terminated / truncated -> auto-reset. State lives
// in registers for the whole step; HBM is touched once per array on the way in and once on the way out.
//
// step():  BE:1835-1950 (template), PL:420-450 (control), PL:536-602 (obs/info), PL:459-534 (reward/terminated)
// reset(): BE:1770-1833, PL:355-418
#include <type_traits>
#include "gpr_device.h"
#include "gpr_host.h"

namespace gpr {

// a noisy qpos read for the box shape (BE:828 perturbs all seven components): x,y take normal pair q, the identity quaternion
// takes pairs q+1 (w,x) and q+2 (y,z); z is never looked at
template <typename RT, bool NOISE> __device__ __forceinline__ Pose<RT> noisy_pose(R2<RT> p, RT sigma, uint32_t key, uint32_t q) {
    Pose<RT> o{p.x, p.y, (RT)1, (RT)0, (RT)0, (RT)0};
    if constexpr (NOISE) {
        const R2<RT> n0 = rng_normal_pair<RT>(key, q), n1 = rng_normal_pair<RT>(key, q + 1u), n2 = rng_normal_pair<RT>(key, q + 2u);
        o.x = p.x + sigma * n0.x; o.y = p.y + sigma * n0.y;
        o.qw = (RT)1 + sigma * n1.x; o.qx = (RT)0 + sigma * n1.y;
        o.qy = (RT)0 + sigma * n2.x; o.qz = (RT)0 + sigma * n2.y;
    }
    return o;
}

template <typename RT> struct Obs {
    R2<RT> vel, ag;  // noisy velocity (observation) and noisy position (achieved_goal)
    RT reward;
    bool terminated, is_success;
};

// _get_obs (PL:536-573), _get_info (PL:575-602), compute_reward (PL:502-534), compute_terminated (PL:459-479)
template <typename ST, typename RT, int LPE, bool NOISE>
__device__ __forceinline__ Obs<RT> observe(const KParams &P, uint32_t key, uint32_t q0, R2<RT> p, R2<RT> v, R2<RT> g, bool mc, bool wc,
                                           bool live) {
    const Cst<RT> &C = cst<RT>(P);
    Obs<RT> o;
    o.ag = noisy<RT, NOISE>(p, C.sig_p, key, q0);       // PL:554
    o.vel = noisy<RT, NOISE>(v, C.sig_v, key, q0 + 1u);  // PL:555
    // the reward is a function of the returned (stored) achieved / desired goal
    RT dx = stored<ST, RT>(o.ag.x) - g.x, dy = stored<ST, RT>(o.ag.y) - g.y;
    RT d2 = dx * dx + dy * dy;
    bool reached = live & sqrt_within<RT, false>(d2, C.thr);  // PL:524, inclusive
    int n_reached = __popc(group_bits<LPE>(reached));
    bool coll = mc | wc;
    RT r = -C.reward_success * (RT)coll;                      // PL:529
    r += (RT)-1.0 * (RT)(P.M - n_reached) * (RT)(!coll);      // PL:530
    bool all_reached = n_reached == P.M;
    if (all_reached & !coll) r = C.reward_success;            // PL:531
    o.reward = r;
    o.terminated = (r == C.reward_success) | (r == -C.reward_success);  // PL:478
    o.is_success = all_reached & !mc & !wc;                               // PL:597
    return o;
}

// _reset_callback (PL:355-418) + the collision flags / observation of reset() (BE:1797-1831) for the env groups whose
// `need` is set. Wave-cooperative: every lane of the wavefront must call it (it votes).
//
// PARALLEL TRIALS. The reference samples all movers of an env at once and rejects the whole set on any conflict
// (PL:369-385: ~1.7 trials per reset with 4 movers, ~45 with 16). The draws are indexed by trial, so the candidate the
// sequential loop accepts is simply the first valid one in trial order -- and S lane groups that all work on the SAME env
// (S a power of two, wave-uniform; the S groups are adjacent and start at a multiple of S) find it S trials at a time:
// group k of the set evaluates trial base + k, a ballot shows which are valid, the first one wins and its candidate is
// fetched by ds_bpermute. S = 1 is the plain sequential loop (k_reset: one env per lane group). All S groups of a set
// return the same p, g, flags and observation; the caller lets one of them store.
// SCRIPT (k_reset only, S = 1): when P.cand is set the candidates come from the caller's stream [n_cand][E][2M] instead of the engine's
// draws -- set k of the stream is the k-th sample np_random.uniform hands out: the start loop consumes sets until it accepts one, the
// goal loop goes on from the next (the reference's two loops share one generator, PL:377 / PL:402). An exhausted stream ends a loop
// like max_reset_trials does (the last set stands).
// MODE (the shared searches of the 16-lane step kernels, step_body): 0 = the whole reset; 1 = this wavefront's stay with the shared search
// coop_pass of the env -- trials are claimed from *coop_next, the first valid one is posted to *coop_best, nothing else is touched;
// 2 = the search is over -- the start is candidate coop_ts, the goal candidate coop_tg -- and the rest of the reset (collision flags,
// first observation) is formed.
template <typename ST, typename RT, int LPE, bool NOISE, bool FULL, int SHAPE, bool SCRIPT = false, int MODE = 0>
__device__ __forceinline__ Obs<RT> group_reset(const KParams &P, const Tables<RT> &T, int64_t env_global, int mv, int mvc, bool live,
                                               bool need, RT c, RT c1, int S, R2<RT> &p, R2<RT> &g, bool &wc, bool &mc, int coop_pass = 0,
                                               int *coop_next = nullptr, int *coop_best = nullptr, int coop_ts = 0, int coop_tg = 0) {
    const Cst<RT> &C = cst<RT>(P);
    const int M = P.M;
    const CircleSize<RT> cw_safe = circle_size<RT>(P, (c + C.c_offset_wall) + C.c_offset);  // BE:487 with add_safety_offset=True
    const RT ws0 = (c + C.c_offset_wall) + C.c_offset, ws1 = (c1 + C.c_offset_wall) + C.c_offset;  // box: wall sizes, BE:487
    const RT ps0 = c + C.c_offset, ps1 = c1 + C.c_offset;                                        // box: pair sizes with offset, BE:390
    // box shape on a grid without holes (FULL): a candidate has the identity quaternion, its rectangle is axis-aligned, and
    // BE:459-788 is "the rectangle lies strictly inside the grid" -- the circle's table-free test with per-axis sizes
    const CircleSize<RT> cw_box_safe{ws0, (RT)0, (RT)0, (RT)0.5 * (C.gx1 - C.gx0) - ws0, (RT)0.5 * (C.gy1 - C.gy0) - ws1};
    constexpr uint32_t R = SHAPE == 0 ? 1u : 3u;  // normal pairs per noisy qpos read
    const uint32_t ks = rng_key(P, KIND_RESET_START, env_global, (uint32_t)mvc);
    const uint32_t kg = rng_key(P, KIND_RESET_GOAL, env_global, (uint32_t)mvc);
    const uint32_t kn = rng_key(P, KIND_RESET_NOISE, env_global, (uint32_t)mvc);
    const RT sx = C.max_x - C.min_x, sy = C.max_y - C.min_y;
    const int gi = (int)(__lane_id() / LPE);
    const int slot = gi & (S - 1), lead_lane = (gi & ~(S - 1)) * LPE;  // my trial offset in a round; first lane of my env's set
    const unsigned long long set_mask = S * LPE >= 64 ? ~0ull : ((1ull << (S * LPE)) - 1ull);

    const bool scripted = SCRIPT && P.cand != nullptr;  // (wave-uniform)
    int used = 0;                                        // scripted: sets of the stream this env has consumed so far
    auto candidate = [&](uint32_t key, int trial) {
        R2<RT> cn;
        if (SCRIPT && scripted) {
            const int k = used + trial < P.n_cand ? used + trial : P.n_cand - 1;
            cn = load2<ST, RT>(P.cand, ((int64_t)k * P.E + (env_global - P.env_id_offset)) * (2 * (int64_t)M) + 2 * mvc);
        } else {
            cn.x = stored<ST, RT>(C.min_x + sx * rng_uniform<RT>(key, 2u * (uint32_t)trial));  // PL:377 / PL:402
            cn.y = stored<ST, RT>(C.min_y + sy * rng_uniform<RT>(key, 2u * (uint32_t)trial + 1u));
        }
        if (mv >= M) cn.x = cn.y = far_away<RT>();
        return cn;
    };
    auto from_lane = [&](RT v, int lane) {  // the value `lane` holds (per-lane source)
        if constexpr (sizeof(RT) == 4) {
            return __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(lane << 2, (int)__float_as_uint(v)));
        } else {
            const uint64_t b = (uint64_t)__double_as_longlong(v);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(lane << 2, (int)(uint32_t)b);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(lane << 2, (int)(uint32_t)(b >> 32));
            return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
        }
    };
    // is this lane's member cn of its group's candidate set of `pass` in conflict with a wall or another member (wave-cooperative: every lane calls it)
    auto conflict = [&](int pass, R2<RT> cn) {
        bool w, hit;
        if constexpr (SHAPE == 0) {
            w = live & !pos_valid_circle<RT, FULL>(T, P, cn.x, cn.y, cw_safe);                    // PL:379 / PL:403-404
            hit = live & (pass == 0 ? pair_closer<RT, LPE, false>(cn.x, cn.y, C.max_csum_safe)
                                    : pair_closer<RT, LPE, true>(cn.x, cn.y, C.min_goal_dist));  // PL:381-383 / PL:407-413
        } else {
            const Pose<RT> q{cn.x, cn.y, (RT)1, (RT)0, (RT)0, (RT)0};
            if constexpr (FULL) w = live & !(wall_reach_full<RT>(C, cn.x, cn.y, cw_box_safe) < cw_box_safe.wx);  // axis-aligned candidate
            else w = live & !pos_valid_box<RT, false>(T, P, q, ws0, ws1, (RT)0, nullptr);
            if (pass == 0) {  // a candidate set is axis-aligned: bounding boxes decide, the 16 segment tests per pair only in a sliver
                bool sure = false, unsure = false;
                AabbSweep<RT, LPE, 1>::run(cn.x, cn.y, ps0, ps1, sure, unsure);
                if (__ballot(live & unsure) == 0ull) hit = live & sure;
                else hit = live & box_pair_hit<RT, LPE>(q, ps0, ps1, (RT)0, true, nullptr);
            }
            else hit = live & dist_lt<RT>(pair_min_dist<RT, LPE>(cn.x, cn.y), C.min_goal_dist);
        }
        return w | hit;
    };
    auto rejected = [&](int pass, R2<RT> cn) { return group_any<LPE>(conflict(pass, cn)); };  // the verdict on the lane group's candidate set
    if constexpr (MODE == 1) {
        // one wavefront's stay with a shared search (all S groups on the same env): rounds of S trials claimed from *coop_next until no
        // unclaimed trial lies below the best valid one posted in *coop_best; the next claim is made BEFORE the round in hand is evaluated
        // (its LDS round trip hides behind the round; a claim that turns out to lie above the best trial is simply dropped)
        const uint32_t key = coop_pass == 0 ? ks : kg;
        const int max_t = P.max_reset_trials;
#ifndef GPR_COOP_RPC
#define GPR_COOP_RPC 2
#endif
        constexpr int RPC = GPR_COOP_RPC;  // rounds per claim (same device, us per step at 4 096 / 32 768 envs x 16 movers -- 1: 58.1 / 119.8, 2: 52.9 / 114.9, 4: 53.6 / 115.4)
        int cl = 0, seen = 0;   // (lane 0) the claim made ahead and the best trial seen with it; judged when its turn comes
        auto claim = [&]() {
            if (__lane_id() == 0u) {
                cl = __hip_atomic_fetch_add(coop_next, RPC * S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                seen = __hip_atomic_load(coop_best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        claim();
        for (bool found = false; !found;) {
            const int b = __builtin_amdgcn_readfirstlane(cl), lim = __builtin_amdgcn_readfirstlane(seen);
            if (!(b < (lim < max_t ? lim : max_t))) break;  // nothing unclaimed below the best valid trial (or the bound): the search is closed
            claim();
            for (int r = 0; r < RPC && !found; ++r) {
                // one trial per lane group (S == 64 / LPE == 4 groups of 16 lanes): the first group without a lane in conflict, straight
                // from the ballot in scalar registers
                static_assert(LPE == 16, "the shared search is the 16-lane kernels'");
                const int t0 = b + r * S;
                const unsigned long long bm = __ballot(conflict(coop_pass, candidate(key, t0 + slot)));
                const uint32_t lo = (uint32_t)bm, hi = (uint32_t)(bm >> 32);
                const int gv = (lo & 0xFFFFu) == 0u ? 0 : (lo >> 16) == 0u ? 1 : (hi & 0xFFFFu) == 0u ? 2 : (hi >> 16) == 0u ? 3 : 4;
                if ((gv < 4) & (t0 + gv < max_t)) {  // (trials grow with the group: if the first valid one lies beyond the bound, all do)
                    if (__lane_id() == 0u) __hip_atomic_fetch_min(coop_best, t0 + gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    found = true;  // (whatever this wavefront has claimed beyond lies above the trial just posted)
                }
            }
        }
        return Obs<RT>{};
    }
    if constexpr (MODE == 2) {
        if (need) { p = candidate(ks, coop_ts); g = candidate(kg, coop_tg); }
    }
    for (int pass = 0; pass < (MODE == 2 ? 0 : 2); ++pass) {  // 0: start positions (PL:369-385), 1: goals (PL:395-413)
        const uint32_t key = pass == 0 ? ks : kg;
        bool pending = need;
        R2<RT> cand = pass == 0 ? p : g;
        int taken = 0;  // scripted: sets this pass consumed
        for (int base = 0; base < P.max_reset_trials; base += S) {
            if (__ballot(pending) == 0ull) break;
            const int trial = base + slot;
            const R2<RT> cn = candidate(key, trial);
            bool bad = rejected(pass, cn);
            if (SCRIPT && scripted) {  // (every lane went through the votes above; only the bookkeeping differs)
                if (pending) taken = trial + 1;
                if (used + trial >= P.n_cand - 1) bad = false;  // the stream's last set stands whatever it is
            }
            if (S == 1) {  // (wave-uniform) the sequential form: the candidate stands as soon as it is valid, or as the last one
                if (pending) { cand = cn; pending = bad; }
            } else {
                const unsigned long long okm = __ballot(pending & !bad & (trial < P.max_reset_trials));
                const unsigned long long seg = (okm >> lead_lane) & set_mask;   // the valid trials of my env's set, in trial order
                const int wlane = lead_lane + (__builtin_ctzll(seg | (1ull << 63)) & ~(LPE - 1)) + mv;  // (my mover in) the first valid group
                const RT wx = from_lane(cn.x, wlane & 63), wy = from_lane(cn.y, wlane & 63);
                if (pending & (seg != 0ull)) { cand.x = wx; cand.y = wy; pending = false; }
            }
        }
        if ((S != 1) & pending) cand = candidate(key, P.max_reset_trials - 1);  // trials exhausted: the last candidate stands
        // ... and it is INVALID: the reference keeps looping and warns every 100 trials (PL:371-375, 397-401); the engine counts the
        // search (GPR_CNT_RESET_EXHAUSTED), the host layer turns a non-zero count into that warning
        if (!(SCRIPT && scripted) && (pending & need & live & (mv == 0) & (slot == 0))) diag_add(P, 0, 1u);
        if (need) { if (pass == 0) p = cand; else g = cand; }
        used += taken;
    }
    if (SCRIPT && scripted) {
        if ((P.cand_used != nullptr) & live & need & (mv == 0)) P.cand_used[env_global - P.env_id_offset] = used < P.n_cand ? used : P.n_cand;  // (a loop that found the stream empty took nothing)
    }

    // BE:1799-1805: wall check with the safety offset, mover check without, both on noisy reads
    if constexpr (SHAPE == 0) {
        bool lazy_done = false;
        {
            // as in the step loop: a noisy read is within band_p of the truth, so wherever the noise-free verdicts have more room than
            // that they stand and the two draws are skipped (exact). In parity mode too: a double-precision draw is a hash and two
            // dependent table loads, and a served reset claim used to cost a wavefront 12 000 clocks
            if (P.eager_noise == 0) {
                bool w_unc;
                const bool w_ok = pos_valid_circle<RT, FULL, true>(T, P, p.x, p.y, cw_safe, C.band_p, &w_unc);
                const RT band_pair = (RT)2 * C.band_p, thr2 = C.max_csum_step * C.max_csum_step;
                RT md;  // the smallest squared pair distance
                if constexpr (Math<RT>::literal) md = pair_min_d2<RT, LPE>(p.x, p.y);
                else md = pair_min_dist<RT, LPE>(p.x, p.y);
                const bool p_unc = absr(md - thr2) <= (RT)2 * C.max_csum_step * band_pair + band_pair * band_pair + (RT)1e-12 * thr2;
                if (__ballot(live & need & (w_unc | p_unc)) == 0ull) {
                    wc = group_any<LPE>(live & !w_ok);
                    mc = group_any<LPE>(live & (md <= thr2));
                    lazy_done = true;
                }
            }
        }
        if (!lazy_done) {
            R2<RT> pw = noisy<RT, NOISE>(p, C.sig_p, kn, 0u);
            wc = group_any<LPE>(live & !pos_valid_circle<RT, FULL>(T, P, pw.x, pw.y, cw_safe));
            R2<RT> pp = noisy<RT, NOISE>(p, C.sig_p, kn, R);
            mc = group_any<LPE>(live & pair_closer<RT, LPE, false>(pp.x, pp.y, C.max_csum_step));
        }
    } else {
        bool lazy_done = false;
        if constexpr (FULL) {
            // grid without holes, throughput mode: the accepted start is axis-aligned and (almost always) further from the walls
            // and from the other movers than a noisy read can bridge: then the two noisy checks are decided without their reads
            // (six normal pairs, rotated vertices, the table walk, 16 segment pairs); exact, as in the step loop
            if (P.eager_noise == 0) {
                const RT bdiag = Math<RT>::sqrt_(ws0 * ws0 + ws1 * ws1);
                const RT band_b = C.band_p * ((RT)1 + (RT)3.5 * bdiag) + (RT)1e-6 * bdiag;
                const RT e = wall_reach_full<RT>(C, p.x, p.y, cw_box_safe);
                const bool w_unc = live & !(__builtin_fabsf(cw_box_safe.wx - e) > band_b);
                const bool near_b = live & BoxNear<RT, LPE, 1>::run(p.x, p.y, C.max_b0sum + (RT)2 * band_b + (RT)2e-6, C.max_b1sum + (RT)2 * band_b + (RT)2e-6);
                if (__ballot(w_unc | near_b) == 0ull) {
                    wc = group_any<LPE>(live & !(e < cw_box_safe.wx));
                    mc = false;
                    lazy_done = true;
                }
            }
        }
        if (!lazy_done) {
            const Pose<RT> qw_ = noisy_pose<RT, NOISE>(p, C.sig_p, kn, 0u);
            wc = group_any<LPE>(live & !pos_valid_box<RT, false>(T, P, qw_, ws0, ws1, (RT)0, nullptr));
            const Pose<RT> qp = noisy_pose<RT, NOISE>(p, C.sig_p, kn, R);
            mc = group_any<LPE>(live & box_pair_hit<RT, LPE>(qp, c + C.c_offset * (RT)0, c1 + C.c_offset * (RT)0, (RT)0, true, nullptr));
        }
    }
    return observe<ST, RT, LPE, NOISE>(P, kn, 2u * R, p, R2<RT>{(RT)0, (RT)0}, g, mc, wc, live);
}

template <typename ST, typename RT>
__device__ __forceinline__ void store_env_scalars(const KParams &P, int64_t env, int el, RT reward, bool term, bool trunc, bool succ, bool mc,
                                                  bool wc) {
    P.elapsed[env] = el;
    static_cast<ST *>(P.reward)[env] = (ST)reward;
    P.terminated[env] = (uint8_t)term;
    P.truncated[env] = (uint8_t)trunc;
    P.is_success[env] = (uint8_t)succ;
    P.mover_collision[env] = (uint8_t)mc;
    P.wall_collision[env] = (uint8_t)wc;
}

template <typename ST, typename RT, bool JERK>
__device__ __forceinline__ void store_obs(const KParams &P, void *obs, void *ag, void *dg, int64_t env, int mv, int64_t idx, R2<RT> ov,
                                          R2<RT> acc, R2<RT> a_g, R2<RT> g, bool write_dg) {
    if constexpr (JERK) {  // PL:560: all velocities, then all accelerations
        int64_t base = env * (4 * (int64_t)P.M) + 2 * mv;
        store2<ST, RT>(obs, base, ov);
        store2<ST, RT>(obs, base + 2 * P.M, acc);
    } else {
        store2<ST, RT>(obs, idx, ov);
    }
    store2<ST, RT>(ag, idx, a_g);
    if (write_dg) store2<ST, RT>(dg, idx, g);
}

// -----------------------------------------------------------------------------------------------------------------
// per-agent views in the step kernel's epilogue (gpr_step_out.agent_observation_dev; layout: include/gpr.h, gpr_agent_views)
// -----------------------------------------------------------------------------------------------------------------
// The lanes of an env group hold every mover's observed velocity and position: agent i's row is its own six values plus, for every
// other mover j in ascending order, (p_j - p_i, v_j - v_i) -- lane rotations inside the group. Differences are taken in the STORAGE
// type on the values as stored, exactly as the stand-alone k_agent_views (gpr_ops_kernels.hip) forms them from memory.
template <typename ST, int LPE, int D> struct ViewGather {
    static __device__ __forceinline__ void run(ST px, ST py, ST vx, ST vy, int mv, int M, ST *row) {
        if constexpr (D < LPE) {
            const ST ox = rot<LPE, D>(px), oy = rot<LPE, D>(py), ovx = rot<LPE, D>(vx), ovy = rot<LPE, D>(vy);
            const int j = (mv + D) & (LPE - 1);
            if (j < M && mv < M) {  // (a lane that writes nothing is called with mv = LPE)
                ST *dst = row + 6 + 4 * (j < mv ? j : j - 1);
                dst[0] = ox - px; dst[1] = oy - py; dst[2] = ovx - vx; dst[3] = ovy - vy;
            }
            ViewGather<ST, LPE, D + 1>::run(px, py, vx, vy, mv, M, row);
        }
    }
};
// the env reward (PL:519-531) split per agent from the achieved / desired goal the reward was computed from, as stored
template <typename ST> __device__ __forceinline__ ST agent_reward_of(const KParams &P, ST reward, ST ax, ST ay, ST gx, ST gy) {
    const double r = (double)reward, rs = P.cd.reward_success, dx = (double)ax - (double)gx, dy = (double)ay - (double)gy;
    const bool shared = (r == rs) | (r == -rs);
    return shared ? (ST)r : (ST)(sqrt(dx * dx + dy * dy) <= P.cd.thr ? 0.0 : -1.0);
}
// bytes of LDS every wavefront of a step kernel owns: the velocity-reading rows of the cycle loop (12 rows + the zero row of 64
// float2, or 6 + 1 rows of double2 in the parity loop) and, once the loop is through, the staging area of its agents' view rows
constexpr int WAVE_SCRATCH_BYTES = 7 * 64 * 16;  // 7 168 >= 13 * 64 * 8

// -----------------------------------------------------------------------------------------------------------------
// step
// -----------------------------------------------------------------------------------------------------------------
// One env-step of every env of the block. MULTI (gpr_step_n): the block calls it once per step with that step's parameter block
// (`first`: the first step of the launch), wavefronts reset their own envs and never wait for each other.
// What a wavefront of gpr_step_n keeps in registers from one step of the launch to the next: its envs' state as it was stored
// (a reset env's as its reset left it), the next step's actions (loaded a whole step ahead) and the per-lane constants --
// a step then starts computing at once instead of waiting for its own stores to come back through L2.
template <typename RT> struct StepCarry {
    R2<RT> p, v, g, acc, a_next;
    RT c, c1;
    int el;
    uint32_t cnt;
};
// the value lane `lane` holds (per-lane source)
__device__ __forceinline__ float lane_fetch(float v, int lane) { return __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(lane << 2, (int)__float_as_uint(v))); }
__device__ __forceinline__ double lane_fetch(double v, int lane) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(lane << 2, (int)(uint32_t)b), hi = (uint32_t)__builtin_amdgcn_ds_bpermute(lane << 2, (int)(uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
#ifndef GPR_THIN_KB
#define GPR_THIN_KB 10
#endif
#ifndef GPR_THIN_KB8
#define GPR_THIN_KB8 8
#endif
#ifndef GPR_THIN_KB16
#define GPR_THIN_KB16 8
#endif
template <typename ST, typename RT, int LPE, bool JERK, bool NOISE, bool FULL, int SHAPE, bool MULTI, bool LONGBLK = false>
__device__ __forceinline__ void step_body(const KParams &P, bool first, StepCarry<RT> *carry = nullptr, const void *next_action = nullptr) {
    extern __shared__ __align__(16) unsigned char lds[];
    const Cst<RT> &C = cst<RT>(P);
    const int M = P.M;
    // THIN WAVEFRONTS (P.thin_shift, wave-uniform; make_params decides from the handle's wavefront budget). What a wavefront's step costs beyond its quiet cycles are its
    // EVENTS -- flagged blocks, cycles on the exact path, clamp-form blocks, resets served -- and their number grows with the envs the
    // wavefront holds, while a batch that leaves most SIMDs idle pays for the slowest wavefront only. So a small batch is spread over more
    // wavefronts: only the first 64 >> thin_shift lanes of a wavefront hold movers (the others are dead lanes, as in a batch's last
    // wavefront), i.e. fewer envs and fewer events per wavefront. Results do not depend on it (random streams are keyed by global env id).
    const int per = 64 >> P.thin_shift;  // lanes of a wavefront that hold movers (a multiple of LPE)
    const int lane64 = (int)(threadIdx.x & 63);
    const int64_t wv = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;  // this wavefront, counted over the grid
    const int64_t t = wv * per + lane64;
    const int mv = lane64 & (LPE - 1);
    const int64_t env = lane64 < per ? t / LPE : P.E;
    const bool live_env = env < P.E;
    const bool live = live_env & (mv < M);
    const int64_t envc = live_env ? env : P.E - 1;  // dead lanes shadow a real env (no OOB, never stored)
    const int mvc = mv < M ? mv : M - 1;
    const int64_t idx = envc * (2 * (int64_t)M) + 2 * mvc;
    const int64_t env_global = P.env_id_offset + envc;

#ifdef GPR_WAVE_TIMING
    const long long wt0 = __builtin_amdgcn_s_memtime();
    int wt_clamp = 0, wt_near = 0, wt_served = 0, wt_rows = 0, wt_direct = 0, wt_due = 0, wt_exact = 0;
    long long wt_loop0 = 0, wt_loop1 = 0;
#define GPR_WT(...) __VA_ARGS__
#else
#define GPR_WT(...)
#endif
    // one raw slice of LDS per wavefront: the velocity-reading rows of the cycle loops, then the staging area of the agent views
    __shared__ __align__(16) unsigned char wave_scratch[BLOCK / 64][WAVE_SCRATCH_BYTES];
    unsigned char *const my_scratch = wave_scratch[threadIdx.x >> 6];
    const int64_t env_block_base = (int64_t)blockIdx.x * (BLOCK / 64) * per / LPE;  // first env of this block
    // RESET QUEUE of the block. A reset is a pure function of (seed, tick, env id): it does not need the env's final state, only
    // the knowledge THAT the env ends in this call -- and that is known early: a TimeLimit truncation (and a next-step-mode
    // reset) before the first cycle, a collision in the cycle it happens. An env is queued the moment it is known to end
    // (local env index * 2 + "reset only"), and every wavefront that is through with its own cycles serves the queue, so that
    // the resets of the block overlap with the cycles of its slowest wavefront (one that clamps in all 40 cycles takes twice
    // as long as a quiet one) instead of waiting for it. Entries: -1 = reserved but not written yet.
    // (gpr_step_n, MULTI: no queue -- every wavefront resets its own envs at the end of each of its steps, so that wavefronts never
    // wait for each other and a block needs no barrier between two steps)
    __shared__ int reset_count, reset_claimed, waves_through, reset_list[BLOCK];
    // the block-cooperative search of the 16-lane kernels (see "SERVING THE QUEUE" below), per queue entry: the next unclaimed trial and the
    // smallest valid trial found so far of the start / goal search, chunks in flight, 1 = someone is forming the reset from the result
#ifdef GPR_NO_COOP
    constexpr bool COOP = false;
#else
    constexpr bool COOP = !MULTI && LPE == 16;
#endif
    constexpr int QCAP = COOP ? BLOCK / LPE : 1;
    __shared__ int coop_next[2 * QCAP], coop_best[2 * QCAP], coop_pend[QCAP], coop_state[QCAP];
    if constexpr (!MULTI) {
        reset_list[threadIdx.x] = -1;
        if (threadIdx.x == 0) { reset_count = 0; reset_claimed = 0; waves_through = 0; }
        if constexpr (COOP) {
            if (threadIdx.x < 2 * QCAP) { coop_next[threadIdx.x] = 0; coop_best[threadIdx.x] = 0x7fffffff; }
            if (threadIdx.x < QCAP) { coop_pend[threadIdx.x] = 0; coop_state[threadIdx.x] = 0; }
        }
    }
    bool queued = false;
    auto enqueue = [&](bool reset_only) {
        queued = true;
        if constexpr (MULTI) return;
        const int slot = atomicAdd(&reset_count, 1);
        __hip_atomic_store(&reset_list[slot], (int)(env - env_block_base) * 2 + (int)reset_only, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);  // (only live lanes queue)
    };

    // every HBM read of the step is issued here, before the tables are staged: the latencies overlap
    // (gpr_step_n after its first step: nothing is read -- the wavefront carried it over in registers, see StepCarry)
    const bool carried = MULTI && !first && carry != nullptr;
    R2<RT> a, p, v, g, acc{(RT)0, (RT)0};
    int el;
    // episode counters (include/gpr.h: GPR_CNT_*): lanes 0..5 of a wavefront own the six words of the wavefront's row
    uint32_t cnt_old = 0u;
    const int64_t cnt_row = wv;
    RT c, c1;
    if (carried) {
        a = carry->a_next; p = carry->p; v = carry->v; g = carry->g; acc = carry->acc; el = carry->el; cnt_old = carry->cnt;
        c = carry->c; c1 = carry->c1;
    } else {
        a = load2<ST, RT>(P.action, idx);
        p = load2<ST, RT>(P.pos, idx);
        v = load2<ST, RT>(P.vel, idx);
        g = load2<ST, RT>(P.goal, idx);
        if constexpr (JERK) acc = load2<ST, RT>(P.acc, idx);
        el = P.elapsed[envc];
        if ((P.counters != nullptr) & (__lane_id() < 6u)) cnt_old = P.counters[cnt_row * 6 + __lane_id()];
        c = make_tables<RT>(P.tables, P).csz[2 * mvc];       // per-mover size straight from the (L2-resident) device table
        c1 = make_tables<RT>(P.tables, P).csz[2 * mvc + 1];  // box: second half size
    }
    if (MULTI && carry != nullptr) {
        if (next_action != nullptr) carry->a_next = load2<ST, RT>(next_action, idx);  // consumed a whole step from now
        carry->c = c; carry->c1 = c1;
    }
    // the lane's noise key (eight hash rounds, five of them on the scalar unit) depends on nothing that is loaded: formed HERE, while the
    // loads are in flight (the compiler had put it behind their arrival), and pinned so that it stays here
    // (float kernels of up to 8 lanes per env: -1 to -2 %; the 16-lane and the double kernels have no register to spare for it: +6 % / +1 %)
    uint32_t key = 0u;
    if constexpr (NOISE) {
        key = rng_key(P, KIND_STEP_NOISE, env_global, (uint32_t)mvc);
        if constexpr (!Math<RT>::literal && LPE <= 8) asm volatile("" : "+v"(key));
    }
    if (mv >= M) { p.x = p.y = far_away<RT>(); v.x = v.y = (RT)0; a.x = a.y = (RT)0; acc.x = acc.y = (RT)0; }  // group padding
    // the throughput-mode wall check of a full grid needs no table at all (pos_valid_circle): no LDS staging, no barrier
    if constexpr (!FULL || SHAPE == 1 || Math<RT>::literal) {  // (the box functions and the literal checks read the tables)
        if (!MULTI || first) stage_tables(lds, P.tables, P.table_bytes);
    } else if constexpr (!MULTI) {
        __syncthreads();  // (reset_count is initialised)
    }
    const Tables<RT> T = make_tables<RT>(lds, P);
    const CircleSize<RT> cw_step = circle_size<RT>(P, (c + C.c_offset_wall) + (RT)0 * C.c_offset);  // BE:487, add_safety_offset=False

    // BE:1871-1873: clip to the action space (PL:257-259)
    const RT lim = JERK ? C.j_max : C.a_max;
    if (P.count_clipped) {  // (wave-uniform, opt-in) the reference warns before it clips, BE:1871-1873: count the env-steps it would warn about
        const bool outside = live & (mv < M) & ((absr(a.x) > lim) | (absr(a.y) > lim));
        const unsigned long long om = __ballot(group_any<LPE>(outside) & (mv == 0));
        if ((om != 0ull) & (__lane_id() == 0u)) diag_add(P, 1, (uint32_t)__popcll(om));
    }
    a.x = fmin(fmax(a.x, -lim), lim);
    a.y = fmin(fmax(a.y, -lim), lim);

    const bool reset_first = (P.autoreset == 2) & (el < 0);  // next-step autoreset: this call only resets the env
    {
        const bool trunc_known = (P.autoreset == 1) & (P.max_episode_steps > 0) & (el + 1 >= P.max_episode_steps);
        if (live & (mv == 0) & (reset_first | trunc_known)) enqueue(reset_first);
    }
    // per-env decisions are kept as wave masks (one bit per lane, identical within an env group): scalar registers
    unsigned long long frozen_m = __ballot(reset_first), wc_m = 0ull, mc_m = 0ull;
    const unsigned long long live_m = __ballot(live);

    // Throughput (float) mode integrates DISPLACEMENTS: p = p0 + dp, v = v0 + dv with dp, dv summed from zero inside the
    // step (jerk mode never runs in float, see launch_flags). A constant velocity adds the same 1e-3-sized increment to a ~1 m position every cycle; summed
    // directly in float that rounds the same way 2000 times per episode (measured: 1.6e-4 m drift over 4200 cycles, above the
    // 1e-4 parity tolerance). Summed as a displacement, the large-magnitude rounding happens once per env-step.
    // Parity (double) mode keeps the reference's v += dt*a; p += dt*v as written.
    const R2<RT> p0 = p, v0 = v;
    R2<RT> dp{(RT)0, (RT)0}, dv{(RT)0, (RT)0};

    // LAZY NOISE. The reference perturbs three readings per mover per cycle (velocity for the clamp, position for the wall
    // check, position again for the mover check: PL:430, BE:1888-1901). A reading can differ from the truth by at most
    // band = sigma * NORMAL_MAX_RADIUS, so whenever the noise-free quantity is further than `band` from every threshold it is
    // compared with, the outcome is the same for every possible draw and the draw is skipped -- results are bit-identical to
    // drawing always (GPR_EAGER_NOISE=1 does that; tests compare the two). The branch is per wavefront: one uncertain lane
    // makes all 64 draw.
    const bool eager = P.eager_noise != 0;
    constexpr uint32_t R = SHAPE == 0 ? 1u : 3u, S = 1u + 2u * R;  // normal pairs per qpos read / per mover per cycle
    // box shape: wall sizes (BE:487, add_safety_offset=False), and how far a vertex can move under a noisy read: the centre by
    // band_p, plus a rotation of at most 2*sqrt(3)*band_p rad (noise on the identity quaternion) times the half diagonal
    const RT bw0 = (c + C.c_offset_wall) + (RT)0 * C.c_offset, bw1 = (c1 + C.c_offset_wall) + (RT)0 * C.c_offset;
    const RT bdiag = Math<RT>::sqrt_(bw0 * bw0 + bw1 * bw1);
    const RT band_b = C.band_p * ((RT)1 + (RT)3.5 * bdiag) + (RT)1e-6 * bdiag;
    const RT band_pair = (RT)2 * C.band_p;  // two movers, each off by up to band_p
    const R2<RT> dta{C.dt * a.x, C.dt * a.y};
    const RT vlim = C.v_max - C.band_v, vlim2 = vlim > (RT)0 ? vlim * vlim : (RT)-1;
    // the mover check is uncertain while |dist - thr| <= band_pair; on squared distances: |d2 - thr^2| <= 2 thr band + band^2
    const RT pair_thr = Math<RT>::literal ? C.max_csum_step : C.max_csum_step * C.max_csum_step;
    const RT pair_band = Math<RT>::literal ? band_pair : (RT)2 * C.max_csum_step * band_pair + band_pair * band_pair;
    const RT pair_near2 = (C.max_csum_step + band_pair) * (C.max_csum_step + band_pair) * (RT)1.000001;  // squared, with the noise band

    if constexpr (!Math<RT>::literal && (SHAPE == 0 || FULL)) {
    // ---------------------------------------------------------------------------------------------------------------
    // Throughput loop: float arithmetic, circle shape, acceleration actions. Same decisions as the generic loop below.
    //
    // Measured on MI355X (scratch/ubench): a SIMD retires a wave64 vector instruction every ~4.5 clocks in dependent code
    // (2.7 with independent neighbours, ~10 for rsq/log/sin), and a wave-level decision that hangs on a vector compare
    // (v_cmp -> VCC -> s_cbranch) costs its wavefront 40-60 clocks of latency. With the four wavefronts per SIMD this
    // workload provides, the loop is vector-issue bound as long as it takes few such decisions per cycle. So:
    //  * the common cycle is straight-line: v += dt a; p += dt v (4 packed instructions), the wall reach (4) and the smallest
    //    pair distance (9) of the new position, and ONE combined test "is any lane within the noise band of a wall or of
    //    another mover?" (2 compares, 1 branch). Only if it fires are the exact, noisy checks (and the freeze) evaluated;
    //  * the speed clamp (PL:634-643) is not even looked at while it provably cannot fire: the acceleration is constant
    //    during an env-step, so |v + dt a| grows by at most |dt a| per cycle, and a lane that is k |dt a| below v_max - band_v
    //    stays unclamped for k cycles. The wave-wide k (rungs 3, 10, 40) sets the next cycle at which speeds are looked at;
    //    a wavefront with a lane at the speed limit (14 % of them) runs the clamp every cycle, in line.
    // All shortcuts are exact: results are bit-identical to evaluating everything on every cycle, which GPR_EAGER_NOISE=1
    // does (the thresholds are then set so that every shortcut test fails) and the tests compare. Lanes that must not vote
    // (dead / padding lanes, frozen envs) are taken out through per-lane thresholds instead of scalar mask arithmetic.
    // ---------------------------------------------------------------------------------------------------------------
    // cycles per block of the block loop below: 8 where the per-cycle trace of a block fits the registers (4 lanes per env,
    // circles: 127 of the 128 registers that keep 4 wavefronts per SIMD; one decision per 8 cycles: -2 % at 65 536 envs, -6 % at
    // 4 096), else 4 (8 lanes per env: +10 % with blocks of 8; layouts with holes, whose flagged blocks are replayed: +8 %)
    // (LONGBLK = THIN launches, k_step_thin: few envs per wavefront, so a block is rarely flagged and fewer, longer blocks pay -- and at one
    // wavefront per SIMD the longer trace has the registers. Up to 4 lanes per env: 10 cycles (4 096 x 4 as one launch: blocks of 8 / 10 /
    // 20: 8.92 / 8.53 / 8.83 us); 8 lanes: 8 cycles, one pair sweep -- four ds_bpermute rotations -- per 8 instead of per 4 cycles (8 192 x 8:
    // 13.3 -> 12.6 us; blocks of 10 / 20: 12.8 / 13.9; at full occupancy the longer trace costs a wavefront per SIMD, 65 536 x 8: 23.4 ->
    // 24.6, so whole-wavefront launches keep blocks of 4); 16 lanes: 8 cycles (4 096 x 16: 30.0 -> 28.7 us; 10: 28.9))
    constexpr int KB = (SHAPE == 0 && FULL) ? ((LONGBLK && LPE <= 4) ? GPR_THIN_KB : (LONGBLK && LPE == 8) ? GPR_THIN_KB8 : (LONGBLK && LPE == 16) ? GPR_THIN_KB16 : LPE <= 4 ? 8 : 4) : 4;
    constexpr int KC1 = KB - 1, KC2 = 3 * KB - 1, KC3 = 39;  // rungs of the clamp ladder: 1 block, 3 blocks, 40 cycles
    const bool every = P.eager_noise != 0;
    const RT INF = Math<RT>::inf();
    // CHORD PRE-TEST OF A BLOCK (circles). Inside a block a mover's velocity changes by at most A dt per cycle -- A = sqrt(2) a_max (the
    // action is clipped per axis, BE:1871-1873; the speed clamp projects t = v_read + dt a onto the ball of radius v_max, PL:634-643, and
    // a projection onto a convex set moves a velocity that is inside it no further than the unclamped update would) plus what the noisy
    // velocity reads of two cycles add -- so the block's positions q_0 .. q_KB have second differences of at most A dt^2 and lie within
    // chord_dw = A dt^2 KB^2 / 8 (+ float slack) of the chord from q_0 to q_KB. Hence, for every cycle k of the block:
    //  * wall: the reach is convex and 1-Lipschitz, so reach(q_k) <= max(reach(q_0), reach(q_KB)) + chord_dw;
    //  * pair: the relative position r = q_i - q_j stays within 2 chord_dw of ITS chord, and on a chord of length L
    //    |r(t)|^2 = (1 - t) |r_0|^2 + t |r_KB|^2 - t (1 - t) L^2 >= min(|r_0|^2, |r_KB|^2) - L^2 / 4, with L^2 / 4 <= (s_i + s_j) / 2,
    //    s = the squared displacement of a mover over the block (|a - b|^2 <= 2 |a|^2 + 2 |b|^2).
    // So ONE reach and ONE pair sweep at the block's end (the start's are the previous block's) decide whether any cycle of the block can
    // come within the noise band of a wall or of another mover; only if so are the cycles' own reaches formed, from the states the block
    // kept anyway, and the flagged ones examined as before. The pre-test is conservative, what it lets through is untouched: results stay
    // bit-identical to evaluating every cycle (GPR_EAGER_NOISE=1; the tests compare). -DGPR_NO_CHORD builds the per-cycle form.
#ifdef GPR_NO_CHORD
    constexpr bool CHORD = false;
#else
    constexpr bool CHORD = SHAPE == 0;
#endif
    const RT chord_a = (RT)1.4143 * C.a_max + (RT)3 * C.band_v * C.inv_dt;
    const RT chord_dw = chord_a * C.dt * C.dt * (RT)(KB * KB / 8.0) + (RT)2e-6;
    // Box shape on a grid without holes: noise-free the rectangle is axis-aligned (identity quaternion), so its wall test is the
    // circle's with per-axis sizes, "close to a wall" allows for the vertex displacement band_b of a noisy read, and "close to
    // another mover" means centres within the summed circumradii (+ 2 band_b); the exact rectangle tests (BE:559-783,
    // GEO:107-138: 16 segment pairs per rectangle pair) run only in the cycles in which something is that close.
    const CircleSize<RT> cw_thr = SHAPE == 0 ? cw_step
        : CircleSize<RT>{bw0, (RT)0, (RT)0, (RT)0.5 * (C.gx1 - C.gx0) - bw0, (RT)0.5 * (C.gy1 - C.gy0) - bw1};
    const RT band_w = SHAPE == 0 ? C.band_p : band_b;
    RT wx_l = cw_thr.wx;                                   // wall: valid iff reach < wx_l (full grid)
    // LAYOUTS WITH HOLES (circles): the block loop's wall pre-test against a per-lane FREE BOX instead of the grid -- the largest
    // rectangle of existing tiles inside the 3x3 neighbourhood of the tile the mover starts the step above that leaves the start
    // position the most room (T.nbr). A circle smaller than half a tile whose bounding box lies inside such a
    // rectangle only ever sees existing tiles, and BE:459-788 then finds no missing neighbour on any side it reaches over: valid,
    // exactly as on a grid without holes. Positions outside the box prove nothing: their blocks are replayed through exact_cycle's
    // table-driven test (what every cycle of such a layout went through before). fb_x / fb_y: the box's centre relative to the
    // grid's, fb_k: half width minus half height.
    RT fb_x = (RT)0, fb_y = (RT)0, fb_k = cw_thr.wx - cw_thr.wy;
    if constexpr (!FULL) {
        const int ix = clampi(Math<RT>::floor_int(p0.x * C.inv_wx), 0, P.Tx - 1), iy = clampi(Math<RT>::floor_int(p0.y * C.inv_wy), 0, P.Ty - 1);
        const uint32_t n = T.nbr[ix * P.Ty + iy];
        const int xl = (n >> 1) & 1u ? -1 : 0, xr = (n >> 7) & 1u ? 1 : 0;
        const bool usable = ((n >> 4) & 1u) & (cw_step.cfx < (RT)0.5) & (cw_step.cfy < (RT)0.5);
        // every maximal rectangle of existing tiles around the centre tile has one of four x ranges and, for it, the rows that exist
        // over the whole range; the one in which the start position has the most room is the lane's free box
        RT best = -INF, hx_b = (RT)-1;
        auto candidate = [&](int cxl, int cxr) {
            const uint32_t row_dn = (cxl ? 1u : 0u) | 8u | (cxr ? 64u : 0u), row_up = row_dn << 2;  // bit (dx + 1) * 3 + (dy + 1)
            const int yl = (n & row_dn) == row_dn ? -1 : 0, yr = (n & row_up) == row_up ? 1 : 0;
            const RT x_lo = T.ax[4 * (ix + cxl)], x_hi = T.ax[4 * (ix + cxr) + 1], y_lo = T.ay[4 * (iy + yl)], y_hi = T.ay[4 * (iy + yr) + 1];
            const RT hx = (RT)0.5 * (x_hi - x_lo) - cw_step.c, hy = (RT)0.5 * (y_hi - y_lo) - cw_step.c;
            const RT mx = (RT)0.5 * (x_lo + x_hi), my = (RT)0.5 * (y_lo + y_hi);
            const RT room = __builtin_fminf(hx - __builtin_fabsf(p0.x - mx), hy - __builtin_fabsf(p0.y - my));
            if (room > best) {
                best = room; hx_b = hx; fb_k = hx - hy;
                fb_x = mx - (RT)0.5 * (C.gx0 + C.gx1); fb_y = my - (RT)0.5 * (C.gy0 + C.gy1);
            }
        };
        candidate(xl, xr); candidate(xl, 0); candidate(0, xr); candidate(0, 0);
        wx_l = usable ? hx_b : (RT)-1;
    }
    RT w0 = every ? -INF : wx_l - band_w;                   // certainly valid, whatever the noise, iff reach < w0
    const RT wband = every ? INF : band_w;
    const RT pb = C.max_csum_step + band_pair;
    // box shape: "close to another mover" = the bounding boxes of the two (noise-free: axis-aligned) rectangles, each inflated
    // by the vertex displacement of a noisy read, overlap
    RT bnx = every ? INF : C.max_b0sum + (RT)2 * band_b + (RT)2e-6, bny = every ? INF : C.max_b1sum + (RT)2 * band_b + (RT)2e-6;
    RT pthr_l = pair_thr;                                  // pair: hit iff d2 <= pthr_l
    RT p0r = every ? INF : pb * pb * (RT)1.00001;          // certainly no hit, whatever the noise, iff d2 > p0r
    const RT pband = every ? INF : pair_band;
    // chord pre-test: (sqrt(p0r) + 2 chord_dw)^2 <= p0r + chord_padd; a lane that is out (p0r = -1) stays below every value the test forms
    const RT chord_padd = (RT)4 * chord_dw * (pb * (RT)1.00001 + chord_dw) * (RT)1.0001;
    // ... which relies on |v| <= v_max + band_v at the block's start (true after every cycle the engine integrated; a caller may have
    // written anything into the state): a block of the clamp form that starts faster takes the per-cycle path
    const RT chord_vcap = ((RT)1 + (RT)2 * C.band_v / C.v_max) * ((RT)1 + (RT)2 * C.band_v / C.v_max) * (RT)1.00001;
    // Per-lane integration operands. A frozen env (BE:1904 break, or an env that only resets in this call) keeps its state:
    // its lanes integrate with a zero time step, a zero increment and an unreachable clamp, and compare against wall limits of
    // +inf and pair limits of -1 (never reached).
    R2<RT> dtl{C.dt * a.x, C.dt * a.y};
    RT dt_l = C.dt, vmax2_l = C.v_max * C.v_max;
    const RT vlim_c = NOISE ? vlim : C.v_max;             // below this speed the clamp cannot fire (vlim = v_max - band_v)
    RT vneed2_l = every ? (RT)-1 : (vlim_c > (RT)0 ? vlim_c * vlim_c : (RT)-1);
    // clamp rungs: the squared speed below which the lane stays unclamped for KCi more cycles
    const RT adt = Math<RT>::sqrt_(dtl.x * dtl.x + dtl.y * dtl.y) * (RT)1.00001 + (RT)1e-7;
    auto rung = [&](int k) { const RT m = vlim_c * (RT)0.99999 - (RT)k * adt; return m > (RT)0 ? m * m : (RT)-1; };
    RT c1r = rung(KC1), c2r = rung(KC2), c3r = rung(KC3);
#define GPR_LANE_OUT() do { wx_l = w0 = INF; pthr_l = p0r = bnx = bny = (RT)-1; c1r = c2r = c3r = INF; if (!every) vneed2_l = INF; } while (0)
    if (!live) GPR_LANE_OUT();
    if (lane_of(frozen_m)) { dtl.x = dtl.y = (RT)0; dt_l = (RT)0; vmax2_l = INF; GPR_LANE_OUT(); }
    // index of a noise draw of this cycle, formed in scalar registers where it is needed (kept out of the loop's induction
    // variables: the common cycle draws nothing)
    auto draw_index = [&](int cycle, uint32_t which) { uint32_t q = S * (uint32_t)cycle + which; asm volatile("" : "+s"(q)); return q; };
    int next_c = 0;  // wave-uniform: the first cycle at which the speed clamp has to be looked at again
    // velocity-reading rows of this wavefront's needy lanes (see the clamp path); row 0 is all zeros
    constexpr int VN_ROWS = 12;
    static_assert((VN_ROWS + 1) * 64 * sizeof(float2) <= WAVE_SCRATCH_BYTES, "reading rows exceed the wavefront's LDS slice");
    float2 *const vn_tab = reinterpret_cast<float2 *>(my_scratch);  // [VN_ROWS + 1][64]
    vn_tab[__lane_id()] = make_float2(0.f, 0.f);
    const float2 *vn_ptr = vn_tab;  // per lane: my row (row 0 until the lane needs readings)
    int vn_rows = 0;                                 // wave-uniform: rows in use
    bool vn_on = !every;                  // false: every lane draws per cycle (eager mode, > VN_ROWS needy lanes, > 64 cycles)
    unsigned long long vn_uncovered = ~0ull;  // lanes that may NOT simply read their row: no row of their own yet (or !vn_on: all)
    const R2<RT> q0{p0.x - (RT)0.5 * (C.gx0 + C.gx1), p0.y - (RT)0.5 * (C.gy0 + C.gy1)};
    // Shared arithmetic of the state update, written once so that every path below rounds identically (explicit fma's: the
    // results must not depend on which contraction the compiler picks in which context).
    auto speed2 = [&](R2<RT> t) { return __builtin_fmaf(t.x, t.x, t.y * t.y); };
    auto integrate = [&](R2<RT> dvn) {  // mj_step (BE:1882): semi-implicit Euler on the displacements
        dv = dvn;
        v.x = v0.x + dv.x; v.y = v0.y + dv.y;
        dp.x = __builtin_fmaf(dt_l, v.x, dp.x); dp.y = __builtin_fmaf(dt_l, v.y, dp.y);  // the position moves with the NEW velocity
    };
    // PL:634-643 on t = v_read + dt a (v_read = v + nv): clamped lanes get dv' = (v_max t/|t| - nv) - v0
    auto clamp_dv = [&](R2<RT> t, R2<RT> nv, RT n2, R2<RT> dvn) {
        const bool cl = n2 >= vmax2_l;  // PL:634, inclusive
        const RT sc = C.v_max * Math<RT>::rsqrt_(n2);
        dvn.x = cl ? __builtin_fmaf(t.x, sc, -nv.x) - v0.x : dvn.x;
        dvn.y = cl ? __builtin_fmaf(t.y, sc, -nv.y) - v0.y : dvn.y;
        return dvn;
    };
    // Velocity readings (PL:430) matter only to the lanes inside the clamp band -- one or two lanes of the wavefront, the same
    // ones for the rest of the env-step -- and a reading is a pure function of (lane key, cycle). So when a lane first needs
    // one, the 64 lanes draw, once, THAT lane's readings for cycles 0..63 (lane c draws cycle c) into a row of LDS; from then
    // on a cycle costs every lane one ds_read (its own row, or the all-zero row) instead of a Box-Muller draw of which 63
    // results are not looked at. Returns false when the rows are (or have to be) given up: then every lane draws per cycle.
    auto build_rows = [&](unsigned long long fresh) {
        if (!vn_on) return false;
        for (unsigned long long f = fresh; f != 0ull; f &= f - 1ull) {
            if ((vn_rows == VN_ROWS) | (P.num_cycles > 64)) { vn_on = false; vn_uncovered = ~0ull; return false; }
            const int L = __builtin_ctzll(f);
            const uint32_t kL = (uint32_t)__builtin_amdgcn_readlane((int)key, L);
            const R2<RT> n = rng_normal_pair<RT>(kL, S * (uint32_t)__lane_id());
            ++vn_rows;
            GPR_WT(++wt_rows;)
            vn_tab[vn_rows * 64 + __lane_id()] = make_float2(C.sig_v * n.x, C.sig_v * n.y);
            if ((int)__lane_id() == L) vn_ptr = vn_tab + vn_rows * 64;
            vn_uncovered &= ~(1ull << L);
        }
        return true;
    };

    // chord pre-test: the wall reach (block_reach) and the smallest squared pair distance of the CURRENT state -- the start of the next
    // block's chords; kept up to date by whatever advances the state (a block leaves its end's, exact_cycle its cycle's)
    RT e_cur = (RT)0, m_cur = INF;
    // ---- one control cycle, everything evaluated in its place: the path of the cycles in which something may happen ----
    auto exact_cycle = [&](int cyc) __attribute__((always_inline)) {
        // _mujoco_step_callback (PL:420-450) + mj_step (BE:1882), see the generic loop for the algebra
        R2<RT> dvn{dv.x + dtl.x, dv.y + dtl.y};
        if (cyc >= next_c) {
            GPR_WT(++wt_due;)
            R2<RT> t{v0.x + dvn.x, v0.y + dvn.y};  // the velocity this cycle ends with unless the clamp fires
            RT n2 = speed2(t);
            float2 vn_r = make_float2(0.f, 0.f);
            if constexpr (NOISE) vn_r = vn_ptr[cyc & 63];  // issued here so that its LDS latency hides behind the tests below
            const unsigned long long needm = __ballot(!(n2 < vneed2_l));
            if (needm == 0ull) {
                const int k = __ballot(!(n2 < c1r)) != 0ull ? 0 : __ballot(!(n2 < c2r)) != 0ull ? KC1 : __ballot(!(n2 < c3r)) != 0ull ? KC2 : KC3;
                next_c = cyc + 1 + k;
            } else {
                GPR_WT(++wt_clamp;)
                R2<RT> nv{(RT)0, (RT)0};
                if constexpr (NOISE) {
                    const unsigned long long fresh = needm & vn_uncovered;
                    if (fresh == 0ull) {  // the steady state: one scalar decision, the reading is already on its way
                        nv.x = vn_r.x; nv.y = vn_r.y;
                    } else if (build_rows(fresh)) {
                        const float2 r = vn_ptr[cyc];
                        nv.x = r.x; nv.y = r.y;
                    } else {
                        GPR_WT(++wt_direct;)
                        const R2<RT> n = rng_normal_pair<RT>(key, draw_index(cyc, 0u));
                        nv.x = C.sig_v * n.x; nv.y = C.sig_v * n.y;
                    }
                    t.x += nv.x; t.y += nv.y;
                    n2 = speed2(t);
                }
                dvn = clamp_dv(t, nv, n2, dvn);
            }
        }
        integrate(dvn);
        // the checks work on q = p - grid centre = q0 + dp (pair distances do not care, the wall reach wants it anyway);
        // p itself is only needed by the table-driven wall check of general layouts, by noisy reads, and after the loop
        const R2<RT> q{q0.x + dp.x, q0.y + dp.y};
        // BE:1888-1901, noise-free and with the noise bands added: can anything at all have happened?
        RT e = (RT)0, md = INF;
        unsigned long long sp_w, sp_p = 0ull;
        bool w_valid = true, w_unc = false, near_b = false;
        if constexpr (FULL) {
            e = wall_reach_centred<RT>(q.x, q.y, cw_thr);
            sp_w = __ballot(!(e < w0));
            if constexpr (CHORD) e_cur = e;
        } else {
            p.x = p0.x + dp.x; p.y = p0.y + dp.y;
            w_valid = pos_valid_circle<RT, FULL, true>(T, P, p.x, p.y, cw_step, C.band_p, &w_unc);
            sp_w = every ? ~0ull : (__ballot(!w_valid | w_unc) & live_m & ~frozen_m);
            if constexpr (CHORD) e_cur = __builtin_fmaxf(__builtin_fabsf(q.x - fb_x), __builtin_fabsf(q.y - fb_y) + fb_k);  // (block_reach)
        }
        if constexpr (LPE > 1) {
            if constexpr (SHAPE == 0) {
                md = pair_min_dist<RT, LPE>(q.x, q.y);
                sp_p = __ballot(md <= p0r);
                if constexpr (CHORD) m_cur = md;
            } else {
                near_b = BoxNear<RT, LPE, 1>::run(q.x, q.y, bnx, bny);
                sp_p = __ballot(near_b);
            }
        }
        if ((sp_w | sp_p) == 0ull) return;
        GPR_WT(++wt_near;)
        p.x = p0.x + dp.x; p.y = p0.y + dp.y;
        // ---- something is close: the exact checks, each on its own noisy position read ----
        unsigned long long wraw, hraw = 0ull;
        if constexpr (SHAPE == 1) {
            // the exact rectangle tests (BE:559-783, GEO:107-138) on noisy reads, for the lanes that can need them: the noise-free
            // (axis-aligned) verdicts stand wherever no read within the noise bound could change them
            const unsigned long long act = live_m & ~frozen_m;
            wraw = __ballot(!(e < wx_l));
            if constexpr (NOISE) {
                if (every | ((__ballot(!(__builtin_fabsf(wx_l - e) > wband)) & act) != 0ull)) {
                    const Pose<RT> qn = noisy_pose<RT, true>(p, C.sig_p, key, draw_index(cyc, 1u));
                    wraw = __ballot(!pos_valid_box<RT, false>(T, P, qn, bw0, bw1, (RT)0, nullptr));
                }
                if (every | ((__ballot(near_b) & act) != 0ull)) {
                    // three zones on the noise-free (axis-aligned) rectangles: overlapping by more than two reads can undo -> a hit
                    // whatever is read (a mover moves up to 2 mm per cycle, the band is 0.1 mm wide: the usual collision); within
                    // the band -> the noisy read and the rectangle tests decide; apart by more -> nothing
                    bool sure = false, unsure = false;
                    AabbSweep<RT, LPE, 1>::run(q.x, q.y, c, c1, sure, unsure, (RT)2 * band_b + (RT)2e-6);
                    if (!every & ((__ballot(unsure) & act) == 0ull)) {
                        hraw = __ballot(sure);
                    } else {
                        const Pose<RT> qn = noisy_pose<RT, true>(p, C.sig_p, key, draw_index(cyc, 1u + R));
                        hraw = __ballot(box_pair_hit<RT, LPE>(qn, c, c1, (RT)0, true, nullptr));
                    }
                }
            } else {
                if ((__ballot(near_b) & act) != 0ull) hraw = __ballot(box_pair_hit<RT, LPE>(Pose<RT>{p.x, p.y, (RT)1, (RT)0, (RT)0, (RT)0}, c, c1, (RT)0, true, nullptr));
            }
            wraw &= act; hraw &= act;
        } else {
        if constexpr (FULL) {
            wraw = __ballot(!(e < wx_l));
            if constexpr (NOISE) {
                if (__ballot(!(__builtin_fabsf(wx_l - e) > wband)) != 0ull) {
                    const R2<RT> pw = noisy<RT, true>(p, C.sig_p, key, draw_index(cyc, 1u));
                    wraw = __ballot(!(wall_reach_full<RT>(C, pw.x, pw.y, cw_step) < wx_l));
                }
            }
        } else {
            wraw = __ballot(!w_valid);
            if constexpr (NOISE) {
                if (every | ((__ballot(w_unc) & live_m & ~frozen_m) != 0ull)) {
                    const R2<RT> pw = noisy<RT, true>(p, C.sig_p, key, draw_index(cyc, 1u));
                    wraw = __ballot(!pos_valid_circle<RT, FULL>(T, P, pw.x, pw.y, cw_step));
                }
            }
            wraw &= live_m & ~frozen_m;
        }
        if constexpr (LPE > 1) {
            hraw = __ballot(md <= pthr_l);  // BE:409, inclusive
            if constexpr (NOISE) {
                if (__ballot(absr(md - pthr_l) <= pband) != 0ull) {
                    const R2<RT> pp = noisy<RT, true>(p, C.sig_p, key, draw_index(cyc, 1u + R));
                    hraw = __ballot(pair_min_dist<RT, LPE>(pp.x, pp.y) <= pthr_l);
                }
            }
        }
        }
        // BE:1904-1905: break at the first collision -- the env's state and flags stay those of the colliding cycle
        if ((wraw | hraw) != 0ull) {
            wc_m |= wraw;
            mc_m |= hraw;
            const unsigned long long nf = group_or<LPE>(wraw | hraw);
            frozen_m |= nf;
            if (lane_of(nf)) {
                dtl.x = dtl.y = (RT)0; dt_l = (RT)0; vmax2_l = INF; GPR_LANE_OUT();
                if ((P.autoreset == 1) & live & (mv == 0) & !queued) enqueue(false);  // a collision ends the env (PL:478): reset it
            }
        }
    };

    // BLOCKS OF KB CYCLES, CHECKED AFTERWARDS (grids without holes; circles, and boxes through their axis-aligned pre-tests). A SIMD of this machine retires a plain vector
    // instruction every ~2.5 clocks and a packed one every ~5 with the four wavefronts this workload gives it, and every
    // wave-level decision (v_cmp -> SGPR -> s_cbranch) stalls its wavefront for 40-60 clocks (scratch/ubench/simd.hip,
    // cycle.hip). So a block of KB cycles runs straight-line: integrate, wall reach and pair sweep of every cycle, each
    // cycle's state and reaches kept in registers, the closest approach to a wall / to another mover as a running max / min --
    // and ONE decision at the block's end: did anything come within the noise band of a threshold? If not (the common case) the
    // block is done. If so, the flagged cycles are looked at in order, each with exact_cycle's noisy tests on the state saved
    // for that cycle; an env that collides is rolled back to the state of its colliding cycle (BE:1904-1905) and taken out,
    // which also takes it out of the tests of the block's later cycles. Same operations in the same order as the
    // cycle-by-cycle path, so the results are bit-identical to it (GPR_EAGER_NOISE=1 runs only exact_cycle; tests compare).
    // The speed clamp: lanes that may reach the limit inside the block (the rung test of the cycle-by-cycle ladder) make the
    // wavefront run the block's clamp form, in which every lane evaluates PL:634-643 branch-free on its reading from LDS
    // (lanes without a row read zeros and cannot clamp).
    GPR_WT(wt_loop0 = __builtin_amdgcn_s_memtime();)
    static_assert(KC1 == KB - 1, "the first rung covers exactly one block");
    const bool fast_ok = !every;
    auto block_reach = [&](RT qx, RT qy) -> RT {  // the block's wall pre-test: grid without holes / free box (see fb_x)
        if constexpr (FULL) return wall_reach_centred<RT>(qx, qy, cw_thr);
        else return __builtin_fmaxf(__builtin_fabsf(qx - fb_x), __builtin_fabsf(qy - fb_y) + fb_k);
    };
    // the block's pair pre-test per cycle, ONE number per lane: circles: the smallest squared pair distance (near iff <= p0r); boxes:
    // the smallest over the partners of max(|dx| - bnx, |dy| - bny) (near iff <= 0: some partner's inflated bounding box overlaps)
    auto pair_reach = [&](RT qx, RT qy) -> RT {
        if constexpr (SHAPE == 0) return pair_min_dist<RT, LPE>(qx, qy);
        else return BoxNearMetric<RT, LPE, 1>::run(qx, qy, bnx, bny);
    };
    if constexpr (CHORD) {
        e_cur = block_reach(q0.x, q0.y);
        if constexpr (LPE > 1) m_cur = pair_reach(q0.x, q0.y);
    }
    for (int cyc = 0; cyc < P.num_cycles;) {  // BE:1879
        int n_exact = P.num_cycles - cyc;
        if (fast_ok & (cyc + KB <= P.num_cycles)) {
            unsigned long long needy = 0ull;
            bool rows_ok = true;
            if (cyc + KB - 1 >= next_c) {
                const RT n2 = speed2(R2<RT>{v0.x + (dv.x + dtl.x), v0.y + (dv.y + dtl.y)});
                needy = __ballot(!(n2 < c1r));
                if (needy == 0ull) {
                    next_c = cyc + 1 + (__ballot(!(n2 < c2r)) != 0ull ? KC1 : __ballot(!(n2 < c3r)) != 0ull ? KC2 : KC3);
                } else if constexpr (NOISE) {
                    const unsigned long long fresh = needy & vn_uncovered;
                    if (fresh != 0ull) rows_ok = build_rows(fresh);
                }
            }
            if (rows_ok) {
                R2<RT> dps[KB], dvs[KB];
                RT es[KB], ms[KB];  // (the per-cycle form's reaches; the chord form makes them when a block is flagged)
                RT eacc = -INF, macc = INF;
                const R2<RT> dp_b = dp, dv_b = dv;  // (box shape, layouts with holes: a flagged block is replayed cycle by cycle from here)
                unsigned long long over = 0ull;     // chord form: lanes of a clamp-form block that start it faster than the engine leaves a mover
                if (needy == 0ull) {
#pragma unroll
                    for (int j = 0; j < KB; ++j) {
                        integrate(R2<RT>{dv.x + dtl.x, dv.y + dtl.y});
                        dps[j] = dp; dvs[j] = dv;
                        if constexpr (!CHORD) {
                            const R2<RT> q{q0.x + dp.x, q0.y + dp.y};
                            es[j] = block_reach(q.x, q.y);
                            eacc = __builtin_fmaxf(eacc, es[j]);
                            ms[j] = INF;
                            if constexpr (LPE > 1) { ms[j] = pair_reach(q.x, q.y); macc = __builtin_fminf(macc, ms[j]); }
                        }
                    }
                } else {
                    GPR_WT(wt_clamp += KB;)
                    if constexpr (CHORD) over = __ballot(speed2(R2<RT>{v0.x + dv.x, v0.y + dv.y}) > vmax2_l * chord_vcap);
                    float2 rd[KB];
                    if constexpr (NOISE) {
#pragma unroll
                        for (int j = 0; j < KB; ++j) rd[j] = vn_ptr[(cyc + j) & 63];
                    }
#pragma unroll
                    for (int j = 0; j < KB; ++j) {
                        const R2<RT> dvn{dv.x + dtl.x, dv.y + dtl.y};
                        R2<RT> t{v0.x + dvn.x, v0.y + dvn.y}, nv{(RT)0, (RT)0};
                        if constexpr (NOISE) { nv.x = rd[j].x; nv.y = rd[j].y; t.x += nv.x; t.y += nv.y; }
                        integrate(clamp_dv(t, nv, speed2(t), dvn));
                        dps[j] = dp; dvs[j] = dv;
                        if constexpr (!CHORD) {
                            const R2<RT> q{q0.x + dp.x, q0.y + dp.y};
                            es[j] = block_reach(q.x, q.y);
                            eacc = __builtin_fmaxf(eacc, es[j]);
                            ms[j] = INF;
                            if constexpr (LPE > 1) { ms[j] = pair_reach(q.x, q.y); macc = __builtin_fminf(macc, ms[j]); }
                        }
                    }
                }
                // (the near limit of a lane that is out -- dead, padding, frozen -- is -1: never reached; bnx is its flag for boxes)
                bool flagged;
                unsigned long long near_w = ~0ull, near_p = ~0ull;  // chord form: lanes whose wall / pair pre-test failed (else: not known apart)
                if constexpr (CHORD) {
                    near_p = 0ull;
                    // the chord pre-test (see chord_dw): the block's end against its start
                    const R2<RT> qe{q0.x + dp.x, q0.y + dp.y};
                    const RT e_end = block_reach(qe.x, qe.y);
                    RT m_end = INF;
                    if constexpr (LPE > 1) {
                        m_end = pair_reach(qe.x, qe.y);
                        const RT ddx = dp.x - dp_b.x, ddy = dp.y - dp_b.y;
                        const RT s2 = __builtin_fmaf(ddx, ddx, ddy * ddy);
                        const RT tst = __builtin_fmaf((RT)-0.5, s2 + group_max<LPE>(s2), __builtin_fminf(m_cur, m_end));
                        near_p = __ballot(tst <= p0r + chord_padd);
                    }
                    near_w = __ballot(!(__builtin_fmaxf(e_cur, e_end) + chord_dw < w0));
                    if (over != 0ull) { near_w = ~0ull; near_p = ~0ull; }  // (no bound on such a block: everything is looked at)
                    flagged = (near_w | near_p) != 0ull;
                    e_cur = e_end; m_cur = m_end;
                } else {
                    flagged = (__ballot(!(eacc < w0)) | __ballot(macc <= (SHAPE == 0 ? p0r : ((RT)0 < bnx ? (RT)0 : (RT)-1)))) != 0ull;
                }
                if ((SHAPE == 1 || !FULL) && flagged) {
                    // box shape: the exact rectangle tests (rotated vertices, table walk, 16 segment pairs per rectangle pair) are
                    // long; instead of one copy of them per cycle of the block, the block is taken back and replayed through
                    // exact_cycle (ONE copy, in the loop below)
                    GPR_WT(++wt_near;)
                    dp = dp_b; dv = dv_b;
                    v.x = v0.x + dv.x; v.y = v0.y + dv.y;
                } else {
                if (flagged) {
                    GPR_WT(++wt_near;)
                    // The cycles of a flagged block, in order. HW / HP: can a wall / a pair be near at all in this block? The chord
                    // pre-test proves the other kind clear for EVERY cycle and lane (its reach stays below w0, its distances above p0r:
                    // no hit, nothing within the noise band), so that kind's metric is not even formed -- the usual flagged block has
                    // one collision of one kind in it. Same verdicts as looking at both (the per-cycle build, GPR_NO_CHORD, does).
                    auto examine = [&](auto hw_tag, auto hp_tag) __attribute__((always_inline)) {
                    constexpr bool HW = decltype(hw_tag)::value, HP = decltype(hp_tag)::value && LPE > 1;
#pragma unroll
                    for (int j = 0; j < KB; ++j) {
                        // (w0 / p0r / wx_l / pthr_l of an env that collided in an earlier cycle of this block are +inf / -1 by now)
                        const RT pnj = SHAPE == 0 ? p0r : ((RT)0 < bnx ? (RT)0 : (RT)-1);
                        RT ej = -INF, mj = INF;
                        if constexpr (CHORD) {  // the cycle's reaches from the state kept for it
                            const R2<RT> qj{q0.x + dps[j].x, q0.y + dps[j].y};
                            if constexpr (HW) ej = block_reach(qj.x, qj.y);
                            if constexpr (HP) mj = pair_reach(qj.x, qj.y);
                        } else {
                            ej = es[j]; mj = ms[j];
                        }
                        unsigned long long fl = 0ull;
                        if constexpr (HW) fl |= __ballot(!(ej < w0));
                        if constexpr (HP) fl |= __ballot(mj <= pnj);
                        if (fl == 0ull) continue;
                        GPR_WT(++wt_exact;)
                        const R2<RT> pj{p0.x + dps[j].x, p0.y + dps[j].y};
                        unsigned long long wraw = 0ull, hraw = 0ull;
                        if constexpr (HW) {
                            wraw = __ballot(!(ej < wx_l));
                            if constexpr (NOISE) {
                                if (__ballot(!(__builtin_fabsf(wx_l - ej) > wband)) != 0ull) {
                                    const R2<RT> pw = noisy<RT, true>(pj, C.sig_p, key, draw_index(cyc + j, 1u));
                                    wraw = __ballot(!(wall_reach_full<RT>(C, pw.x, pw.y, cw_step) < wx_l));
                                }
                            }
                        }
                        if constexpr (HP) {
                            hraw = __ballot(mj <= pthr_l);  // BE:409, inclusive
                            if constexpr (NOISE) {
                                if (__ballot(absr(mj - pthr_l) <= pband) != 0ull) {
                                    const R2<RT> pp = noisy<RT, true>(pj, C.sig_p, key, draw_index(cyc + j, 1u + R));
                                    hraw = __ballot(pair_min_dist<RT, LPE>(pp.x, pp.y) <= pthr_l);
                                }
                            }
                        }
                        if ((wraw | hraw) != 0ull) {  // BE:1904-1905: the env stops in this cycle
                            wc_m |= wraw;
                            mc_m |= hraw;
                            const unsigned long long nf = group_or<LPE>(wraw | hraw);
                            frozen_m |= nf;
                            if (lane_of(nf)) {
                                dp = dps[j]; dv = dvs[j];
                                dtl.x = dtl.y = (RT)0; dt_l = (RT)0; vmax2_l = INF; GPR_LANE_OUT();
                                if ((P.autoreset == 1) & live & (mv == 0) & !queued) enqueue(false);
                            }
                        }
                    }
                    };
                    // (groups of up to 4 lanes: 4 movers x 65 536 envs -1.5 %; wider groups lose -- 8 movers +8 %, 16 movers +3 % -- three copies of their long sweeps)
                    if constexpr (CHORD && LPE > 1 && LPE <= 4 && !MULTI) {  // (gpr_step_n: the copies cost it six spilled registers)
                        if (near_p == 0ull) examine(std::true_type{}, std::false_type{});
                        else if (near_w == 0ull) examine(std::false_type{}, std::true_type{});
                        else examine(std::true_type{}, std::true_type{});
                    } else {
                        examine(std::true_type{}, std::true_type{});
                    }
                    v.x = v0.x + dv.x; v.y = v0.y + dv.y;
                }
                cyc += KB;
                continue;
                }
            }
            n_exact = KB;
        }
        for (int j = 0; j < n_exact; ++j, ++cyc) exact_cycle(cyc);
    }
    GPR_WT(wt_loop1 = __builtin_amdgcn_s_memtime();)
    p.x = p0.x + dp.x; p.y = p0.y + dp.y;
#undef GPR_LANE_OUT
    } else {
    // velocity-reading rows (parity-mode loop), see the clamp path of the throughput loop; row 0 is all zeros
    constexpr int VG_ROWS = 6;
    typedef typename V2<RT>::type VgVec;
    static_assert((VG_ROWS + 1) * 64 * sizeof(VgVec) <= WAVE_SCRATCH_BYTES, "reading rows exceed the wavefront's LDS slice");
    VgVec *const vg_tab = reinterpret_cast<VgVec *>(my_scratch);  // [VG_ROWS + 1][64]
    if constexpr (Math<RT>::literal && NOISE) { VgVec z; z.x = (RT)0; z.y = (RT)0; vg_tab[__lane_id()] = z; }
    const VgVec *vg_ptr = vg_tab;
    int vg_rows = 0;
    bool vg_on = true;
    unsigned long long vg_uncovered = ~0ull;
    // one control cycle of the generic loop in two halves: advance = _mujoco_step_callback (PL:420-450) + mj_step (BE:1882) ...
    auto advance = [&](int cyc) __attribute__((always_inline)) {
        // _mujoco_step_callback (PL:420-450) + mj_step (BE:1882)
        if constexpr (Math<RT>::literal) {
            R2<RT> qacc, acc_tmp = a, j1 = a;
            if constexpr (JERK) j1 = clamp_dyn<RT>(acc, C.a_max, a, C.dt, C.inv_dt, acc_tmp);  // PL:434-436 (no noise on qacc, PL:433)
            R2<RT> vn = v;
            if constexpr (NOISE) {
                // the velocity reading only matters if the speed clamp can fire (PL:634): |v + dt*a| within band_v of v_max
                const RT t0 = v.x + C.dt * acc_tmp.x, t1 = v.y + C.dt * acc_tmp.y;
                const RT lim = C.v_max - C.band_v;
                const bool need = !(t0 * t0 + t1 * t1 < lim * lim) | (lim <= (RT)0);
                const unsigned long long needm = __ballot(need);
                if (eager) {
                    vn = noisy<RT, true>(v, C.sig_v, key, S * (uint32_t)cyc);  // PL:430
                } else if (needm != 0ull) {
                    // as in the throughput loop: the readings of a needy lane for cycles 0..63 are drawn once, by the 64 lanes,
                    // into a row of LDS (a double-precision Box-Muller draw is several hundred instructions)
                    const unsigned long long fresh = needm & vg_uncovered;
                    if (fresh != 0ull) {
                        if (vg_on) {
                            for (unsigned long long f = fresh; f != 0ull; f &= f - 1ull) {
                                if ((vg_rows == VG_ROWS) | (P.num_cycles > 64)) { vg_on = false; vg_uncovered = ~0ull; break; }
                                const int L = __builtin_ctzll(f);
                                const uint32_t kL = (uint32_t)__builtin_amdgcn_readlane((int)key, L);
                                const R2<RT> n = rng_normal_pair<RT>(kL, S * (uint32_t)__lane_id());
                                ++vg_rows;
                                GPR_WT(++wt_rows;)
                                VgVec w; w.x = C.sig_v * n.x; w.y = C.sig_v * n.y;
                                vg_tab[vg_rows * 64 + __lane_id()] = w;
                                if ((int)__lane_id() == L) vg_ptr = vg_tab + vg_rows * 64;
                                vg_uncovered &= ~(1ull << L);
                            }
                        }
                    }
                    if (vg_on) {
                        const VgVec r = vg_ptr[cyc];
                        vn.x = v.x + r.x; vn.y = v.y + r.y;  // (lanes without a row of their own read zeros: they cannot clamp)
                    } else {
                        vn = noisy<RT, true>(v, C.sig_v, key, S * (uint32_t)cyc);
                    }
                }
            }
            if constexpr (JERK) {  // PL:437-440
                R2<RT> vnext;
                R2<RT> acc_p = clamp_dyn<RT>(vn, C.v_max, acc_tmp, C.dt, C.inv_dt, vnext);
                bool diff = (acc_tmp.x != acc_p.x) | (acc_tmp.y != acc_p.y);
                R2<RT> u;
                u.x = diff ? div_rcp<RT>(acc_p.x - acc.x, C.dt, C.inv_dt) : j1.x;
                u.y = diff ? div_rcp<RT>(acc_p.y - acc.y, C.dt, C.inv_dt) : j1.y;
                // mj_step with dyntype=integrator, actearly (PL:307-310): act += dt*ctrl, qacc = act_new
                qacc.x = acc.x + C.dt * u.x;
                qacc.y = acc.y + C.dt * u.y;
            } else {  // PL:442-443; mj_step with a plain force actuator of gain = mass: qacc = ctrl
                // no lane within band_v of the speed limit: ensure_max_dyn_val returns the action unchanged (PL:636-637), and
                // its square root and divisions (long sequences in double) are not evaluated
                const RT t0 = v.x + C.dt * a.x, t1 = v.y + C.dt * a.y;
                if (eager | (__ballot(!(t0 * t0 + t1 * t1 < vlim2)) != 0ull)) {
                    R2<RT> vnext;
                    qacc = clamp_dyn<RT>(vn, C.v_max, a, C.dt, C.inv_dt, vnext);
                } else {
                    qacc = a;
                }
            }
            // MuJoCo's default (semi-implicit) Euler step on a free, undamped, gravity-compensated body (BE:878-883, BE:1132)
            if (!lane_of(frozen_m)) {
                acc = qacc;
                v.x = v.x + C.dt * qacc.x; v.y = v.y + C.dt * qacc.y;
                p.x = p.x + C.dt * v.x; p.y = p.y + C.dt * v.y;
            }
        } else {
            // throughput mode, acceleration actions only. With t = v_read + dt*a (v_read = v + noise):
            //   not clamped (|t| < v_max):  ctrl = a                      -> v' = v + dt*a                (PL:636-637)
            //   clamped:                    ctrl = (v_max t/|t| - v_read)/dt -> v' = v_max t/|t| - noise  (PL:640-643)
            // written on the displacement dv = v - v0 so that the small increments are summed among themselves.
            R2<RT> t{v.x + dta.x, v.y + dta.y}, nv{(RT)0, (RT)0};
            RT n2 = t.x * t.x + t.y * t.y;
            if constexpr (NOISE) {
                // the velocity reading only matters if the speed clamp can fire (PL:634): |v + dt*a| within band_v of v_max
                if (eager | (__ballot(!(n2 < vlim2)) != 0ull)) {
                    const R2<RT> n = rng_normal_pair<RT>(key, S * (uint32_t)cyc);  // PL:430
                    nv.x = C.sig_v * n.x; nv.y = C.sig_v * n.y;
                    t.x += nv.x; t.y += nv.y;
                    n2 = t.x * t.x + t.y * t.y;
                }
            }
            const bool cl = n2 >= C.v_max * C.v_max;  // PL:634, inclusive
            const RT sc = C.v_max * Math<RT>::rsqrt_(n2);
            if (!lane_of(frozen_m)) {
                dv.x = cl ? (t.x * sc - nv.x) - v0.x : dv.x + dta.x;
                dv.y = cl ? (t.y * sc - nv.y) - v0.y : dv.y + dta.y;
                v.x = v0.x + dv.x; v.y = v0.y + dv.y;
                dp.x += C.dt * v.x; dp.y += C.dt * v.y;  // semi-implicit Euler: the position moves with the NEW velocity
                p.x = p0.x + dp.x; p.y = p0.y + dp.y;
            }
        }
    };
    // ... and checks = BE:1888-1901 on the position pc the cycle ended with: raw per-lane verdicts as wave masks
    auto checks = [&](int cyc, R2<RT> pc, unsigned long long &wraw, unsigned long long &hraw) __attribute__((always_inline)) {
        // BE:1888-1901: wall check and mover check, each on its own noisy position read. Results are kept as wave masks
        // (one bit per lane, straight out of the vector compares) and restricted to live, unfrozen lanes with scalar ALU ops.
        if constexpr (SHAPE == 1) {
            const Pose<RT> q0{pc.x, pc.y, (RT)1, (RT)0, (RT)0, (RT)0};
            if constexpr (NOISE) {
                bool unc;
                wraw = __ballot(!pos_valid_box<RT, true>(T, P, q0, bw0, bw1, band_b, &unc));
                if (eager | ((__ballot(unc) & live_m) != 0ull)) {
                    const Pose<RT> qn = noisy_pose<RT, true>(pc, C.sig_p, key, S * (uint32_t)cyc + 1u);
                    wraw = __ballot(!pos_valid_box<RT, false>(T, P, qn, bw0, bw1, (RT)0, nullptr));
                }
                // no partner within the distance pre-filter (BE:411-414) even after two worst-case reads: no collision, no draw
                bool near;
                (void)box_pair_hit<RT, LPE>(q0, c, c1, (RT)2 * band_b, false, &near);
                hraw = 0ull;
                if (eager | ((__ballot(near) & live_m) != 0ull)) {
                    const Pose<RT> qn = noisy_pose<RT, true>(pc, C.sig_p, key, S * (uint32_t)cyc + 1u + R);
                    hraw = __ballot(box_pair_hit<RT, LPE>(qn, c, c1, (RT)0, true, nullptr));
                }
            } else {
                wraw = __ballot(!pos_valid_box<RT, false>(T, P, q0, bw0, bw1, (RT)0, nullptr));
                hraw = __ballot(box_pair_hit<RT, LPE>(q0, c, c1, (RT)0, true, nullptr));
            }
        } else if constexpr (NOISE) {
            bool unc;
            wraw = __ballot(!pos_valid_circle<RT, FULL, true>(T, P, pc.x, pc.y, cw_step, C.band_p, &unc));
            if (eager | ((__ballot(unc) & live_m) != 0ull)) {
                const R2<RT> pw = noisy<RT, true>(pc, C.sig_p, key, S * (uint32_t)cyc + 1u);
                wraw = __ballot(!pos_valid_circle<RT, FULL>(T, P, pw.x, pw.y, cw_step));
            }
            hraw = 0ull;
            const RT md2 = pair_min_d2<RT, LPE>(pc.x, pc.y);
            if (eager | ((__ballot(md2 <= pair_near2) & live_m) != 0ull)) {  // else: no pair can touch
                bool near_thr;
                if constexpr (Math<RT>::literal) {
                    hraw = __ballot(sqrt_within<RT, false>(md2, C.max_csum_step));  // BE:409, inclusive
                    // |dist - limit| <= band, asked of d2 (a superset: more draws at most, never another verdict)
                    const RT lo = C.max_csum_step - pair_band, hi = C.max_csum_step + pair_band;
                    near_thr = ((lo <= (RT)0) | (md2 >= lo * lo * (RT)(1.0 - 1e-12))) & (md2 <= hi * hi * (RT)(1.0 + 1e-12));
                } else {
                    const RT md = pair_min_dist<RT, LPE>(pc.x, pc.y);
                    hraw = __ballot(md <= pair_thr);
                    near_thr = absr(md - pair_thr) <= pair_band;
                }
                if (eager | ((__ballot(near_thr) & live_m) != 0ull)) {
                    const R2<RT> pp = noisy<RT, true>(pc, C.sig_p, key, S * (uint32_t)cyc + 1u + R);
                    hraw = __ballot(pair_closer<RT, LPE, false>(pp.x, pp.y, C.max_csum_step));
                }
            }
        } else {
            wraw = __ballot(!pos_valid_circle<RT, FULL>(T, P, pc.x, pc.y, cw_step));
            hraw = 0ull;
            if ((__ballot(pair_min_d2<RT, LPE>(pc.x, pc.y) <= pair_near2) & live_m) != 0ull)
                hraw = __ballot(pair_closer<RT, LPE, false>(pc.x, pc.y, C.max_csum_step));
        }
    };
    auto settle = [&](unsigned long long wraw, unsigned long long hraw) {
        // BE:1904-1905: break at the first collision -- the env's state and flags stay those of the colliding cycle.
        // wc_m / mc_m collect the raw per-lane hits of not-yet-frozen envs; only the freeze decision needs the per-env OR now.
        const unsigned long long wb = wraw & live_m & ~frozen_m, mb = hraw & live_m & ~frozen_m;
        wc_m |= wb;
        mc_m |= mb;
        const unsigned long long nf = group_or<LPE>(wb | mb);
        frozen_m |= nf;
        return nf;
    };
    int cyc = 0;
    GPR_WT(wt_loop0 = __builtin_amdgcn_s_memtime();)
    if constexpr (Math<RT>::literal && FULL && SHAPE == 0) {
        // PARITY MODE ON A GRID WITHOUT HOLES: the reference's arithmetic, but its checks only where they can matter. The state
        // advances in double exactly as above (v += dt a; p += dt v, the same operations in the same order). Whether a cycle
        // needs the literal wall / pair checks at all is decided in FLOAT on a float copy of the position: on a full grid with
        // circles smaller than half a tile, BE:459-788 is "the circle's bounding box lies strictly inside the grid" (DESIGN.md:
        // interior tile edges never invalidate), so a position further from the four outer limits and from every other mover
        // than noise band + float slack is valid for every possible read, and the literal tests -- table look-ups, tile-edge
        // uncertainty, double-precision Box-Muller draws -- are evaluated only in the flagged cycles, on the double state saved
        // for them. Blocks of KB cycles, one decision per block, as in the throughput loop; cycles in which a lane may reach
        // the speed limit go through advance() one by one (its clamp is a sqrt and four divisions in double) -- and so does
        // every cycle of jerk mode (two clamps per cycle, PL:432-440), which still skips the literal checks of unflagged cycles.
        constexpr int KB = 8;
        const float INFf = __builtin_huge_valf();
        const float cxf = (float)((RT)0.5 * (C.gx0 + C.gx1)), cyf = (float)((RT)0.5 * (C.gy0 + C.gy1));
        const float kwf = (float)(cw_step.wx - cw_step.wy);
        float w0f = eager ? -INFf : (float)(cw_step.wx - C.band_p) - 4e-6f;                  // surely valid iff reach < w0f
        const float pbf = (float)(C.max_csum_step + band_pair) + 8e-6f;
        float p0f = eager ? INFf : pbf * pbf * 1.00001f;                                     // surely no hit iff d2 > p0f
        // clamp rungs (double): the squared speed |v + dt a|^2 below which a lane stays outside the clamp band for k more cycles
        const R2<RT> dta_d{C.dt * a.x, C.dt * a.y};
        const RT adt = sqrt(dta_d.x * dta_d.x + dta_d.y * dta_d.y) * (RT)1.000001 + (RT)1e-12;
        const RT vl = (C.v_max - C.band_v) * (RT)0.999999;
        auto rung = [&](int k) { const RT m = vl - (RT)k * adt; return m > (RT)0 ? m * m : (RT)-1; };
        RT r1 = eager ? (RT)-1 : rung(KB - 1), r2 = eager ? (RT)-1 : rung(3 * KB - 1), r3 = eager ? (RT)-1 : rung(P.num_cycles - 1);
        auto lane_out = [&]() { w0f = INFf; p0f = -1.f; r1 = r2 = r3 = Math<RT>::inf(); };
        if (!live | lane_of(frozen_m)) lane_out();
        int next_c = 0;
        auto reaches = [&](R2<RT> pc, float &e, float &m) {
            const float qx = (float)pc.x - cxf, qy = (float)pc.y - cyf;
            e = __builtin_fmaxf(__builtin_fabsf(qx), __builtin_fabsf(qy) + kwf);
            m = INFf;
            if constexpr (LPE > 1) m = pair_min_dist<float, LPE>(qx, qy);
        };
        auto literal_cycle = [&](int c, R2<RT> pc, R2<RT> vc) {  // the literal checks of cycle c on its saved state
            unsigned long long wraw, hraw;
            GPR_WT(++wt_exact;)
            checks(c, pc, wraw, hraw);
            const unsigned long long nf = settle(wraw, hraw);
            if (nf != 0ull) {
                if (lane_of(nf)) { p = pc; v = vc; lane_out(); }
            }
        };
        // chord pre-test of a quiet block (see chord_dw in the throughput loop): no clamp and no velocity read inside such a block, so its
        // positions have second differences of exactly dt^2 a, |a| <= sqrt(2) a_max -- ONE reach and ONE pair sweep at the block's end
        // instead of one per cycle; e_curf / m_curf: those of the current state (the start of the next block's chords)
#ifdef GPR_NO_CHORD
        constexpr bool CHORDL = false;
#else
        constexpr bool CHORDL = !JERK;
#endif
        const float dwf = (float)((RT)1.4143 * C.a_max * C.dt * C.dt * (RT)(KB * KB / 8.0)) + 2e-6f;
        const float paddf = 4.f * dwf * (pbf * 1.00001f + dwf) * 1.0001f;
        float e_curf = 0.f, m_curf = INFf;
        if constexpr (CHORDL) reaches(p, e_curf, m_curf);
        while (cyc + KB <= P.num_cycles) {
            if (JERK | (cyc + KB - 1 >= next_c)) {
                const RT t0 = v.x + dta_d.x, t1 = v.y + dta_d.y, n2 = t0 * t0 + t1 * t1;
                if (JERK | (__ballot(!(n2 < r1)) != 0ull)) {
                    // some lane may clamp within this block: its cycles one by one (advance() decides about readings and clamp)
                    GPR_WT(wt_clamp += KB;)
                    for (int j = 0; j < KB; ++j, ++cyc) {
                        advance(cyc);
                        float e, m;
                        reaches(p, e, m);
                        e_curf = e; m_curf = m;
                        if ((__ballot(!(e < w0f)) | __ballot(m <= p0f)) != 0ull) literal_cycle(cyc, p, v);
                    }
                    continue;
                }
                next_c = cyc + 1 + (__ballot(!(n2 < r2)) != 0ull ? KB - 1 : __ballot(!(n2 < r3)) != 0ull ? 3 * KB - 1 : P.num_cycles - 1);
            }
            // a quiet block: qacc = a in every cycle (PL:636-637)
            const R2<RT> vb = v, pb = p;
            float eacc = -INFf, macc = INFf;
            const bool moving = !lane_of(frozen_m);
            bool flagged;
            if constexpr (CHORDL) {
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    if (moving) {
                        v.x = v.x + dta_d.x; v.y = v.y + dta_d.y;
                        p.x = p.x + C.dt * v.x; p.y = p.y + C.dt * v.y;
                    }
                }
                float e_end, m_end;
                reaches(p, e_end, m_end);
                unsigned long long near_p = 0ull;
                if constexpr (LPE > 1) {
                    const float ddx = (float)(p.x - pb.x), ddy = (float)(p.y - pb.y);
                    const float s2 = __builtin_fmaf(ddx, ddx, ddy * ddy);
                    const float tst = __builtin_fmaf(-0.5f, s2 + group_max<LPE>(s2), __builtin_fminf(m_curf, m_end));
                    near_p = __ballot(tst <= p0f + paddf);
                }
                flagged = (__ballot(!(__builtin_fmaxf(e_curf, e_end) + dwf < w0f)) | near_p) != 0ull;
                e_curf = e_end; m_curf = m_end;
            } else {
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    if (moving) {
                        v.x = v.x + dta_d.x; v.y = v.y + dta_d.y;
                        p.x = p.x + C.dt * v.x; p.y = p.y + C.dt * v.y;
                    }
                    float e, m;
                    reaches(p, e, m);
                    eacc = __builtin_fmaxf(eacc, e); macc = __builtin_fminf(macc, m);
                }
                flagged = (__ballot(!(eacc < w0f)) | __ballot(macc <= p0f)) != 0ull;
            }
            if (moving) acc = a;
            if (flagged) {
                // a flagged block (rare): its cycles again from the block's start, the same additions in the same order, each
                // with its own decision (keeping the four positions and their distances in registers instead would cost the
                // kernel its fourth wavefront per SIMD)
                R2<RT> vj = vb, pj = pb;
                for (int j = 0; j < KB; ++j) {
                    if (moving) {
                        vj.x = vj.x + dta_d.x; vj.y = vj.y + dta_d.y;
                        pj.x = pj.x + C.dt * vj.x; pj.y = pj.y + C.dt * vj.y;
                    }
                    float e, m;
                    reaches(pj, e, m);
                    if ((__ballot(!(e < w0f)) | __ballot(m <= p0f)) == 0ull) continue;
                    literal_cycle(cyc + j, pj, vj);
                }
            }
            cyc += KB;
        }
    }
    for (; cyc < P.num_cycles; ++cyc) {  // BE:1879
        advance(cyc);
        unsigned long long wraw, hraw;
        checks(cyc, p, wraw, hraw);
        (void)settle(wraw, hraw);
    }
    GPR_WT(wt_loop1 = __builtin_amdgcn_s_memtime();)
    }
    // EPILOGUE. Everything below addresses memory through indices that are derived AGAIN here from the thread index (made opaque to the
    // optimiser): the 64-bit element indices of the prologue's loads would otherwise stay in vector registers across the cycle loops --
    // the registers the loops were budgeted to the last one for (4 wavefronts per SIMD), i.e. spills to scratch memory, whose mere
    // presence costs a launch ~2 us (measured: 17.1 -> 19.0 us per step when the view rows below first pushed seven of them out).
    unsigned tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    {
    const int lane64 = (int)(tid_e & 63);
    const int64_t wv = ((int64_t)blockIdx.x * BLOCK + tid_e) >> 6;
    const int64_t t = wv * per + lane64;
    const int mv = lane64 & (LPE - 1);
    const int64_t env = lane64 < per ? t / LPE : P.E;
    const bool live_env = env < P.E;
    const bool live = live_env & (mv < M);
    const int64_t envc = live_env ? env : P.E - 1;
    const int mvc = mv < M ? mv : M - 1;
    const int64_t idx = envc * (2 * (int64_t)M) + 2 * mvc;
    const int64_t cnt_row = wv;
    wc_m = group_or<LPE>(wc_m);
    mc_m = group_or<LPE>(mc_m);
    const bool wc_step = lane_of(wc_m), mc_step = lane_of(mc_m);
    bool wc = wc_step, mc = mc_step;

    // state as it will be stored
    p.x = stored<ST, RT>(p.x); p.y = stored<ST, RT>(p.y);
    v.x = stored<ST, RT>(v.x); v.y = stored<ST, RT>(v.y);
    acc.x = stored<ST, RT>(acc.x); acc.y = stored<ST, RT>(acc.y);

    Obs<RT> o = observe<ST, RT, LPE, NOISE>(P, key, S * (uint32_t)P.num_cycles, p, v, g, mc, wc, live);
    int el_new = reset_first ? el : el + 1;
    const bool trunc = (P.max_episode_steps > 0) & (el_new >= P.max_episode_steps);  // gymnasium TimeLimit
    const bool done = (o.terminated | trunc) & !reset_first;
    const bool need = live_env & (reset_first | ((P.autoreset == 1) & done));
    if ((P.autoreset == 2) & done) el_new = -1;
    if (need) el_new = 0;
    if (P.counters != nullptr) {  // wave-uniform; popcounts of the per-env flags (one bit per env: its mover-0 lane)
        const unsigned long long head = __ballot(live & (mv == 0) & !reset_first);
        const uint32_t c0 = (uint32_t)__popcll(__ballot(done) & head), c1 = (uint32_t)__popcll(__ballot(o.terminated) & head);
        const uint32_t c2 = (uint32_t)__popcll(__ballot(trunc) & head), c3 = (uint32_t)__popcll(__ballot(o.is_success) & head);
        const uint32_t c4 = (uint32_t)__popcll(__ballot(wc) & head), c5 = (uint32_t)__popcll(__ballot(mc) & head);
        const uint32_t L = __lane_id();
        const uint32_t add = L == 0u ? c0 : L == 1u ? c1 : L == 2u ? c2 : L == 3u ? c3 : L == 4u ? c4 : c5;
        if (L < 6u) P.counters[cnt_row * 6 + L] = cnt_old + add;
        if (MULTI && carry != nullptr) carry->cnt = cnt_old + add;
    }
    if (MULTI && carry != nullptr) {  // the state the next step of this launch starts from (reset envs: overwritten below)
        carry->p = p; carry->v = v; carry->acc = acc; carry->g = g; carry->el = el_new;
    }

    // What the stepping wavefront stores: everything of the envs that go on, and of an env that ended in this call its
    // terminal observation (same-step mode: final_*) and the step's scalars. The reset itself -- new state, goal and
    // observation; for an env that only resets in this call (next-step mode) also the scalars -- is left to the reset pass.
    if (live) {
        if (!need) {
            store2<ST, RT>(P.pos, idx, p);
            store2<ST, RT>(P.vel, idx, v);
            if constexpr (JERK) store2<ST, RT>(P.acc, idx, acc);
            store_obs<ST, RT, JERK>(P, P.obs, P.ag, P.dg, envc, mv, idx, o.vel, acc, o.ag, g, P.dg != P.goal);
        } else if ((P.autoreset == 1) & (P.final_obs != nullptr)) {
            store_obs<ST, RT, JERK>(P, P.final_obs, P.final_ag, P.final_dg, envc, mv, idx, o.vel, acc, o.ag, g, true);
        }
        if ((mv == 0) & !reset_first) store_env_scalars<ST, RT>(P, envc, el_new, o.reward, o.terminated, trunc, o.is_success, mc, wc);
        if (need & (mv == 0) & !queued) enqueue(reset_first);  // (a success, or a collision found by the generic loop)
    }

    // PER-AGENT VIEWS (gpr_step_out.agent_observation_dev), same launch: the rows of this wavefront's agents ((env, mover) in lane order)
    // are consecutive in agent_obs, 4M + 2 values each (264 B with 16 movers). Written by their own lanes they would be 64 scattered
    // pieces per store instruction; so the wavefront assembles them in its LDS slice (the reading rows are dead by now) -- as many
    // whole envs per pass as fit -- and streams every env's M rows out as one span. An env that is reset in this call gets its rows
    // from whoever serves its reset (serve() below): its span is skipped here. The per-agent reward always comes from the state the
    // env reward was computed from, i.e. from here.
    if (P.agent_obs != nullptr) {  // (wave-uniform)
        const int W = 4 * M + 2;
        ST *const aobs = static_cast<ST *>(P.agent_obs);
        const ST vx = (ST)o.vel.x, vy = (ST)o.vel.y, px = (ST)o.ag.x, py = (ST)o.ag.y, gx = (ST)g.x, gy = (ST)g.y;
        if (live & !reset_first) static_cast<ST *>(P.agent_reward)[envc * M + mv] = agent_reward_of<ST>(P, (ST)o.reward, px, py, gx, gy);
        const int lane = (int)__lane_id(), ei = lane / LPE;
        const int64_t env_w0 = (t - lane) / LPE;  // first env of this wavefront
        int64_t n_env64 = P.E - env_w0;
        const int n_env = (int)(n_env64 < 0 ? 0 : (n_env64 > per / LPE ? per / LPE : n_env64));
        const unsigned long long need_m = __ballot(need);
        const int env_bytes = M * W * (int)sizeof(ST);
        int epp = WAVE_SCRATCH_BYTES / env_bytes;  // whole envs per pass
        epp = epp > 64 / LPE ? 64 / LPE : epp;
        if (epp >= 1) {
            ST *const slice = reinterpret_cast<ST *>(my_scratch);
            for (int e0 = 0; e0 < n_env; e0 += epp) {
                const bool mine = live & (ei >= e0) & (ei < e0 + epp);
                ST *row = slice + ((mine ? ei - e0 : 0) * M + mvc) * W;
                if (mine) { row[0] = vx; row[1] = vy; row[2] = px; row[3] = py; row[4] = gx; row[5] = gy; }
                ViewGather<ST, LPE, 1>::run(px, py, vx, vy, mine ? mv : LPE, M, row);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wavefront's LDS writes have landed
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int n8 = env_bytes / 8;  // (M W is even: an env's rows are a whole number of 8-byte pieces, 8-byte aligned)
                for (int k = 0; k < epp && e0 + k < n_env; ++k) {
                    if ((need_m >> ((e0 + k) * LPE)) & 1ull) continue;  // (wave-uniform) reset in this call: rows come from the reset
                    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(slice + k * M * W);
                    unsigned long long *dst = reinterpret_cast<unsigned long long *>(aobs + (env_w0 + e0 + k) * (int64_t)M * W);
                    for (int i = lane; i < n8; i += 64) __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the next pass overwrites the slice: LDS operations of a wavefront stay in order)
            }
        } else {  // (an env's rows do not fit the slice: f64 storage with 15 or 16 movers) every lane stores its own row
            const bool wr = live & !need;
            ST *row = aobs + (envc * M + mvc) * (int64_t)W;
            if (wr) { row[0] = vx; row[1] = vy; row[2] = px; row[3] = py; row[4] = gx; row[5] = gy; }
            ViewGather<ST, LPE, 1>::run(px, py, vx, vy, wr ? mv : LPE, M, row);
        }
    }

    // SERVING THE QUEUE: claim up to 64 / LPE queued envs at a time until nothing unclaimed is left; the lane groups a claim
    // leaves idle join the others as parallel trial slots (group_reset: S groups per env). Whatever is queued later is served
    // by a wavefront that finishes later -- at the latest by the one that queued it.
    // A wavefront that finds the queue empty stays (sleeping between looks) until every wavefront of the block is through with
    // its cycles: the envs the slowest wavefront queues while it is still cycling are then reset before it is done, and
    // nothing but its own last-minute entries is left for the end. Every wavefront gets here without waiting for anything, so
    // the wait ends.
    // envs per claim. Wide env groups reject most candidates (PL:381-383 on all movers at once: ~45 trials per reset with 16
    // movers), so their claims stay small enough for every env to get at least four parallel trial slots
    constexpr int EPW = LPE >= 8 ? (64 / LPE >= 4 ? 64 / LPE / 4 : 1) : 64 / LPE;
    constexpr int GROUPS = 64 / LPE;
    GPR_WT(const long long wt1 = __builtin_amdgcn_s_memtime();)
    // one claim: n envs, S lane groups each; lane group gi works for claim slot gi / S on env `renv`
    // (found: the search was run elsewhere -- the block-cooperative form -- and ended with start candidate ts, goal candidate tg)
    auto serve = [&](int n, int S, bool have, int64_t renv, bool rfirst, R2<RT> &rp, R2<RT> &rg, bool found = false, int ts = 0, int tg = 0)
                     __attribute__((always_inline)) {
        const int gi = (int)(threadIdx.x & 63) / LPE;
        const bool rlive = have & (mv < M);
        rp = R2<RT>{(RT)0, (RT)0}; rg = R2<RT>{(RT)0, (RT)0};
        bool rwc, rmc;
        Obs<RT> ro;
        if (COOP && found) ro = group_reset<ST, RT, LPE, NOISE, FULL, SHAPE, false, 2>(P, T, P.env_id_offset + renv, mv, mvc, rlive, have, c, c1, S, rp, rg, rwc, rmc, 0, nullptr, nullptr, ts, tg);
        else ro = group_reset<ST, RT, LPE, NOISE, FULL, SHAPE>(P, T, P.env_id_offset + renv, mv, mvc, rlive, have, c, c1, S, rp, rg, rwc, rmc);
        if (rlive & ((gi & (S - 1)) == 0)) {  // one group of the env's set stores
            const int64_t ridx = renv * (2 * (int64_t)M) + 2 * mvc;
            const R2<RT> zero{(RT)0, (RT)0};
            store2<ST, RT>(P.pos, ridx, rp);
            store2<ST, RT>(P.vel, ridx, zero);
            if constexpr (JERK) store2<ST, RT>(P.acc, ridx, zero);
            store2<ST, RT>(P.goal, ridx, rg);
            store_obs<ST, RT, JERK>(P, P.obs, P.ag, P.dg, renv, mv, ridx, ro.vel, zero, ro.ag, rg, P.dg != P.goal);
            // an env that only resets in this call: the reset IS the call's result (BE:1797-1831)
            if ((mv == 0) & rfirst) store_env_scalars<ST, RT>(P, renv, 0, (RT)0, false, false, ro.is_success, rmc, rwc);
        }
        if (P.agent_obs != nullptr) {  // (wave-uniform) the reset env's view rows: first observation of the new episode
            const bool wr = rlive & ((gi & (S - 1)) == 0);
            const int W = 4 * M + 2;
            const ST vx = (ST)ro.vel.x, vy = (ST)ro.vel.y, px = (ST)ro.ag.x, py = (ST)ro.ag.y, gx = (ST)rg.x, gy = (ST)rg.y;
            ST *row = static_cast<ST *>(P.agent_obs) + ((have ? renv : 0) * M + mvc) * (int64_t)W;
            if (wr) { row[0] = vx; row[1] = vy; row[2] = px; row[3] = py; row[4] = gx; row[5] = gy; }
            ViewGather<ST, LPE, 1>::run(px, py, vx, vy, wr ? mv : LPE, M, row);
            if (wr & rfirst) static_cast<ST *>(P.agent_reward)[renv * M + mv] = (ST)0;  // (a call that only resets carries reward 0, for every agent too)
        }
        (void)n;
    };
    if constexpr (MULTI) {
        // the wavefront's own envs that end in this step, EPW at a time, in env order
        const int gi = (int)(threadIdx.x & 63) / LPE;
        const int64_t env_w0 = (t - (int64_t)(threadIdx.x & 63)) / LPE;  // first env of this wavefront
        unsigned long long todo = __ballot(need & (mv == 0));           // one bit per needy env: lane gi * LPE
        const unsigned long long rf_m = __ballot(reset_first & (mv == 0));
        while (todo != 0ull) {
            const unsigned long long claim = todo;
            int n = __popcll(todo);
            n = n < EPW ? n : EPW;
            int S = 1;
            while (2 * S * n <= GROUPS) S *= 2;
            const int eslot = gi / S;
            int tgt = 0;  // the env group claim slot `eslot` stands for: the eslot-th needy group
            unsigned long long mm = todo;
            for (int r = 0; r < n; ++r) {
                const int g = __builtin_ctzll(mm) / LPE;
                if (eslot == r) tgt = g;
                mm &= mm - 1ull;
            }
            todo = mm;
            const bool have = eslot < n;
            const bool rfirst = ((rf_m >> (tgt * LPE)) & 1ull) != 0ull;
            GPR_WT(wt_served += n;)
            R2<RT> rp, rg;
            serve(n, S, have, env_w0 + tgt, rfirst, rp, rg);
            if (carry != nullptr) {
                // the new state goes to the env's own lanes: the r-th needy group of `claim` was served by the lane groups
                // r * S .. r * S + S - 1, every one of which holds the result (mover mv in lane mv of the group)
                const int bit = gi * LPE, my_r = __popcll(claim & ((1ull << bit) - 1ull));
                const bool mine = (((claim >> bit) & 1ull) != 0ull) & (my_r < n);
                const int src = (my_r < n ? my_r : 0) * S * LPE + mv;
                const R2<RT> hp{lane_fetch(rp.x, src), lane_fetch(rp.y, src)}, hg{lane_fetch(rg.x, src), lane_fetch(rg.y, src)};
                if (mine) { carry->p = hp; carry->g = hg; carry->v = R2<RT>{(RT)0, (RT)0}; carry->acc = R2<RT>{(RT)0, (RT)0}; }
            }
        }
    } else if constexpr (COOP) {
        // 16 LANES PER ENV: IDLE WAVEFRONTS JOIN THE SEARCHES STILL RUNNING. A candidate set of 16 movers is valid once in ~45 trials, a
        // wavefront tries four per round, and the unluckiest reset of a batch needs hundreds: at 4 096 envs (one wavefront per SIMD) the
        // kernel lasted as long as that one search, run by ONE wavefront while the other three of its block -- of its CU -- had nothing left
        // to do. The trial a search ends with is the first valid one in trial order, whoever evaluates it. So every search (an entry's
        // start positions, its goals: two independent searches) hands out its rounds of four trials from a counter in LDS: a wavefront
        // takes a search nobody has begun if there is one (the block's resets then run side by side, as before), else joins the oldest one
        // still open, and stays with it -- one atomic add per round -- until no unclaimed trial lies below the best valid one (atomic min)
        // posted so far. Whoever then sees both searches of an entry closed with nobody at work on them forms the reset from the two
        // trial numbers (group_reset MODE 2). Closed stays closed (the counter only grows, the best only shrinks), and a wavefront
        // counts as at work from before its first claim: one that claims after somebody saw "closed" gets a round at or above the best
        // trial and leaves, one that claimed before is seen at work or has posted already.
        auto ld = [](const int *a) { return __hip_atomic_load(a, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); };
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&waves_through, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int max_t = P.max_reset_trials;
        GPR_WT(long long ct_scan = 0, ct_sess = 0, ct_form = 0; int cn_poll = 0, cn_sess = 0, cn_help = 0;)
        for (;;) {
            int act = 0, sel = 0;  // act: 0 nothing to do right now, 1 join search `sel` (2 * entry + pass), 2 form the reset of entry `sel`, 3 through
            GPR_WT(const long long ct0 = __builtin_amdgcn_s_memtime();)
            {
                // one look at the whole queue: lane i reads entry i (the counters before the "at work" count, see above), ballots decide
                const int lane = (int)(threadIdx.x & 63), li = lane < QCAP ? lane : 0;
                const bool all_through = ld(&waves_through) == BLOCK / 64;  // (read before the count: nothing is queued after it was seen set)
                const int cnt = __hip_atomic_load(&reset_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int st = __hip_atomic_load(&coop_state[li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int b0 = __hip_atomic_load(&coop_best[2 * li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int b1 = __hip_atomic_load(&coop_best[2 * li + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int n0 = __hip_atomic_load(&coop_next[2 * li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int n1 = __hip_atomic_load(&coop_next[2 * li + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");  // the counters first, then who is at work
                const int pd = __hip_atomic_load(&coop_pend[li], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const bool mine = (lane < QCAP) & (lane < cnt) & (st == 0);
                const bool o0 = mine & (n0 < (b0 < max_t ? b0 : max_t)), o1 = mine & (n1 < (b1 < max_t ? b1 : max_t));
                const unsigned long long m_open = __ballot(mine), m_new0 = __ballot(o0 & (n0 == 0)), m_new1 = __ballot(o1 & (n1 == 0));
                const unsigned long long m_o0 = __ballot(o0), m_o1 = __ballot(o1), m_form = __ballot(mine & !o0 & !o1 & (pd == 0));
                if ((m_new0 | m_new1) != 0ull) {  // a search nobody has begun
                    const int i = __builtin_ctzll(m_new0 | m_new1);
                    act = 1; sel = 2 * i + (((m_new0 >> i) & 1ull) != 0ull ? 0 : 1);
                } else {
                    if (m_form != 0ull) {  // both searches of an entry closed, nobody at work on them: one wavefront forms the reset
                        const int i = __builtin_ctzll(m_form);
                        int won = 0;
                        if (lane == 0) {
                            int expect = 0;
                            won = __hip_atomic_compare_exchange_strong(&coop_state[i], &expect, 1, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1 : 0;
                        }
                        if (__builtin_amdgcn_readfirstlane(won) != 0) { act = 2; sel = i; }
                    }
#ifndef GPR_COOP_NO_HELP
                    if (act == 0 && (m_o0 | m_o1) != 0ull) {  // join the oldest search still open
                        const int i = __builtin_ctzll(m_o0 | m_o1);
                        act = 1; sel = 2 * i + (((m_o0 >> i) & 1ull) != 0ull ? 0 : 1);
                        GPR_WT(cn_help += 1;)
                    }
#endif
                }
                if (act == 1 && lane == 0) __hip_atomic_fetch_add(&coop_pend[sel >> 1], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);  // at work from here on
                if (act == 0 && m_open == 0ull && all_through) act = 3;
            }
            act = __builtin_amdgcn_readfirstlane(act);
            GPR_WT(const long long ct1 = __builtin_amdgcn_s_memtime(); ct_scan += ct1 - ct0;)
            if (act == 3) break;
            if (act == 0) { GPR_WT(cn_poll += 1;) __builtin_amdgcn_s_sleep(8); continue; }
            sel = __builtin_amdgcn_readfirstlane(sel);
            const int ei = act == 1 ? sel >> 1 : sel;
            int entry;
            do { entry = __hip_atomic_load(&reset_list[ei], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); } while (entry < 0);
            const int64_t renv = env_block_base + (entry >> 1);
            R2<RT> rp{(RT)0, (RT)0}, rg{(RT)0, (RT)0};
            if (act == 1) {
                bool rwc, rmc;
                group_reset<ST, RT, LPE, NOISE, FULL, SHAPE, false, 1>(P, T, P.env_id_offset + renv, mv, mvc, mv < M, true, c, c1, GROUPS, rp, rg, rwc, rmc, sel & 1, &coop_next[sel], &coop_best[sel]);
                if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&coop_pend[ei], -1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                GPR_WT(ct_sess += __builtin_amdgcn_s_memtime() - ct1; cn_sess += 1;)
            } else {
                int ts = ld(&coop_best[2 * ei]), tg = ld(&coop_best[2 * ei + 1]);  // (final: see above)
                if (((threadIdx.x & 63) == 0) & ((ts >= max_t) | (tg >= max_t))) diag_add(P, 0, (uint32_t)(ts >= max_t) + (uint32_t)(tg >= max_t));
                ts = ts < max_t ? ts : max_t - 1; tg = tg < max_t ? tg : max_t - 1;  // trials exhausted: the last candidate stands (counted, see group_reset)
                GPR_WT(wt_served += 1;)
                serve(1, GROUPS, true, renv, (entry & 1) != 0, rp, rg, true, ts, tg);
                GPR_WT(ct_form += __builtin_amdgcn_s_memtime() - ct1;)
            }
        }
        GPR_WT(wt_clamp = (int)ct_scan; wt_near = (int)ct_sess; wt_rows = (int)ct_form; wt_direct = cn_poll; wt_due = cn_sess; wt_exact = cn_help;)  // (timing builds: the cooperative search's own clocks in place of the loop's counts)
    } else {
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&waves_through, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (;;) {
        int base = 0, n = 0;
        if ((threadIdx.x & 63) == 0) {
            for (;;) {
                const int all_through = __hip_atomic_load(&waves_through, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == BLOCK / 64;
                const int c0 = __hip_atomic_load(&reset_claimed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int cnt = __hip_atomic_load(&reset_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                n = cnt - c0 < EPW ? cnt - c0 : EPW;
#ifdef GPR_SERVE_MIN
                if ((n > 0) & (n < GPR_SERVE_MIN) & !all_through) n = 0;  // (experiment: a claim waits until GPR_SERVE_MIN envs are queued or every wavefront is through)
#endif
                if (n <= 0) {
                    n = 0;
                    if (all_through) break;  // (read before the counters: nothing can be queued after it was seen set)
                    __builtin_amdgcn_s_sleep(4);
                    continue;
                }
                if (atomicCAS(&reset_claimed, c0, c0 + n) == c0) { base = c0; break; }
            }
        }
        base = __builtin_amdgcn_readfirstlane(base);
        n = __builtin_amdgcn_readfirstlane(n);
        if (n == 0) break;
        GPR_WT(wt_served += n;)
        // S lane groups per claimed env: the largest power of two with n * S <= 64 / LPE
        int S = 1;
        while (2 * S * n <= GROUPS) S *= 2;
        const int gi = (int)(threadIdx.x & 63) / LPE;
        const int eslot = gi / S;
        const bool have = eslot < n;
        int entry;
        do { entry = __hip_atomic_load(&reset_list[base + (have ? eslot : 0)], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); } while (entry < 0);
        R2<RT> rp, rg;
        serve(n, S, have, env_block_base + (entry >> 1), (entry & 1) != 0, rp, rg);
    }
    }
#ifdef GPR_WAVE_TIMING
    if (P.wave_timing && (threadIdx.x & 63) == 0) {  // clocks: cycles + own stores | queue service incl. waiting; counts
        const long long wt2 = __builtin_amdgcn_s_memtime();
        unsigned int *d = P.wave_timing + 16 * ((size_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6));
        d[0] = (unsigned)(wt1 - wt0); d[1] = (unsigned)(wt2 - wt1); d[2] = (unsigned)wt_clamp; d[3] = (unsigned)wt_near; d[4] = (unsigned)wt_served;
        d[5] = (unsigned)wt_rows; d[6] = (unsigned)wt_direct; d[7] = (unsigned)wt_due;
        d[8] = (unsigned)(wt_loop0 - wt0); d[9] = (unsigned)(wt_loop1 - wt_loop0); d[10] = (unsigned)(wt1 - wt_loop1); d[11] = (unsigned)wt_exact;
        d[12] = (unsigned)wt0; d[13] = (unsigned)wt2;  // low words of the absolute clock: start / end of the wavefront
        d[14] = __builtin_amdgcn_s_getreg((31 << 11) | 4); d[15] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_ID, XCC_ID
    }
#endif
    }  // (epilogue)
#undef GPR_WT
}

// Wavefronts per SIMD the circle / full-grid kernels of up to 4 lanes per env are held to: 4 = 128 registers. The double-precision
// acceleration-mode kernel needs 148: at 4 it spilled 20 of them to scratch memory (a launch with scratch costs ~2 us more, and the
// spill traffic sits around the cycle loop), at 3 it spills nothing. Same device, us per env-step of 65 536 envs (scratch/ab_sub.py):
// one launch per step 50.8 (4) vs 56.6 (3); two sub-batches on two streams -- bench.py's way -- 44.7 vs 41.6. make
// EXTRA=-DGPR_LITERAL_WAVES=4 builds the other one. (Jerk mode fits 128 registers without spills.)
#ifndef GPR_LITERAL_WAVES
#define GPR_LITERAL_WAVES 3
#endif
template <typename ST, typename RT, int LPE, bool JERK, bool NOISE, bool FULL, int SHAPE>
#ifndef GPR_BOX_WAVES
#define GPR_BOX_WAVES 1
#endif
__global__ void __launch_bounds__(BLOCK, (SHAPE == 0 && FULL && LPE <= 4) ? ((Math<RT>::literal && !JERK) ? GPR_LITERAL_WAVES : 4)
                                         : ((SHAPE == 1 && FULL && !Math<RT>::literal) ? GPR_BOX_WAVES : 1)) k_step(const KParams P) {
#ifndef GPR_NO_WARM_KERNARG
    warm_kernarg();
#endif
    step_body<ST, RT, LPE, JERK, NOISE, FULL, SHAPE, false>(P, true);
}
// k_step for THIN launches of 8-lane groups: blocks of 8 cycles (step_body: LONGBLK). A kernel of its own name, so that k_step's stays what
// the profiles of every round call it.
template <typename ST, typename RT, int LPE, bool JERK, bool NOISE, bool FULL, int SHAPE>
__global__ void __launch_bounds__(BLOCK) k_step_thin(const KParams P) {
#ifndef GPR_NO_WARM_KERNARG
    warm_kernarg();
#endif
    step_body<ST, RT, LPE, JERK, NOISE, FULL, SHAPE, false, true>(P, true);
}
// gpr_step_n: n env-steps in ONE launch (actions and outputs [n][E][.]). Every wavefront walks through its envs' steps on its own:
// no launch, no grid-wide drain and no simultaneous load burst of the whole batch between two steps, and no barrier either (a
// wavefront resets its own envs). Step i reads its parameter block -- output rows i, call counter tick + i -- from steps[i] in
// device memory through scalar loads, exactly as gpr_step's kernel reads the kernel argument segment (a per-step copy in
// registers cost 40 VGPRs and a wavefront per SIMD), so the n steps are bit for bit the n calls of gpr_step.
template <typename ST, typename RT, int LPE, bool JERK, bool NOISE, bool FULL, int SHAPE>
__global__ void __launch_bounds__(BLOCK, (!Math<RT>::literal && SHAPE == 0 && FULL && LPE <= 4) ? 4 : 1) k_step_n(const KParams *__restrict__ steps, int n_steps) {
    // a wavefront never reads back what it stored in an earlier step of the launch (StepCarry), and no other wavefront touches
    // its envs: the steps need no fence between them, the stores drain while the next step computes
    StepCarry<RT> carry;
    for (int st = 0; st < n_steps; ++st) {
        const void *next_action = st + 1 < n_steps ? steps[st + 1].action : nullptr;
        step_body<ST, RT, LPE, JERK, NOISE, FULL, SHAPE, true>(steps[st], st == 0, &carry, next_action);
    }
}
// -----------------------------------------------------------------------------------------------------------------
// reset (explicit): BE:1770-1833 for the envs selected by mask (all when mask == NULL)
// -----------------------------------------------------------------------------------------------------------------
template <typename ST, typename RT, int LPE, bool JERK, bool NOISE, bool FULL, int SHAPE>
__global__ void __launch_bounds__(BLOCK) k_reset(const KParams P) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int M = P.M;
    const int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int mv = (int)(t & (LPE - 1));
    const int64_t env = t / LPE;
    const bool live_env = env < P.E;
    const bool live = live_env & (mv < M);
    const int64_t envc = live_env ? env : P.E - 1;
    const int mvc = mv < M ? mv : M - 1;
    const int64_t idx = envc * (2 * (int64_t)M) + 2 * mvc;
    const int64_t env_global = P.env_id_offset + envc;
    const bool need = live_env & (P.mask ? (P.mask[envc] != 0) : true);
    const RT c = make_tables<RT>(P.tables, P).csz[2 * mvc], c1 = make_tables<RT>(P.tables, P).csz[2 * mvc + 1];
    R2<RT> p = load2<ST, RT>(P.pos, idx), g = load2<ST, RT>(P.goal, idx);
    if (mv >= M) p.x = p.y = far_away<RT>();
    if constexpr (!FULL || SHAPE == 1 || Math<RT>::literal) stage_tables(lds, P.tables, P.table_bytes);
    const Tables<RT> T = make_tables<RT>(lds, P);
    bool wc, mc;
    Obs<RT> o = group_reset<ST, RT, LPE, NOISE, FULL, SHAPE, true>(P, T, env_global, mv, mvc, live, need, c, c1, 1, p, g, wc, mc);
    if (live & need) {
        const R2<RT> zero{(RT)0, (RT)0};
        store2<ST, RT>(P.pos, idx, p);
        store2<ST, RT>(P.vel, idx, zero);
        store2<ST, RT>(P.acc, idx, zero);
        store2<ST, RT>(P.goal, idx, g);
        store_obs<ST, RT, JERK>(P, P.obs, P.ag, P.dg, envc, mv, idx, o.vel, zero, o.ag, g, P.dg != P.goal);
        if (mv == 0) store_env_scalars<ST, RT>(P, envc, 0, (RT)0, false, false, o.is_success, mc, wc);
    }
}

// -----------------------------------------------------------------------------------------------------------------
// launchers
// -----------------------------------------------------------------------------------------------------------------
// GPR_ENV_PART: the float / float kernels are built in TWO translation units (Makefile). Part 0: the circle shape in groups of up to 8
// lanes, with the compiler's SLP vectoriser on (the headline kernel loses 1.5 % without the packed instructions it forms). Part 1: the
// 'box' shape and the 16-lane groups, compiled -fno-slp-vectorize -- the packed instructions the vectoriser forms out of the rectangle
// tests and the reset searches need register moves to pair their operands and a wait state each on gfx950: the box kernel drops from 205
// to 160 registers (three wavefronts per SIMD instead of two; 83.7 -> 78.6 us as one launch, 52.7 -> 51.4 with four sub-batches), the
// 16-mover kernel gains 3 %. Undefined: one unit with everything (the parity precisions).
#ifndef GPR_ENV_PART
#define GPR_ENV_PART -1
#endif
template <typename ST, typename RT, int LPE, bool JERK, bool NOISE, bool FULL, int SHAPE>
static hipError_t launch_kernel(bool is_reset, const KParams &P, hipStream_t s) {
    constexpr bool WIDE = LPE == 16 || SHAPE != 0;
    if constexpr (GPR_ENV_PART == 0 && WIDE) {
        return launch_env_kernel_wide<ST, RT>(is_reset, LPE, P, s);  // (the kernel is not instantiated in this unit)
    } else if constexpr (GPR_ENV_PART == 1 && !WIDE) {
        return hipErrorInvalidValue;                                  // (nor this one in the other)
    } else {
    // thin wavefronts (step_body): P.thin_shift comes from make_params (gpr_api.hip: the handle's wavefront budget) -- in the kernel
    // argument for gpr_step, in every step's parameter block for gpr_step_n; resets keep whole wavefronts
    KParams Q = P;
    if (is_reset) Q.thin_shift = 0;
    const int64_t live_per_block = (int64_t)(BLOCK / 64) * (64 >> Q.thin_shift);
    const int64_t threads = P.E * LPE;
    const unsigned grid = (unsigned)((threads + live_per_block - 1) / live_per_block);
    const unsigned lds = (FULL && SHAPE == 0 && !Math<RT>::literal) ? 0u : P.table_bytes;
    // the tile tables of a large layout (64 x 64 cells in f64: 47 KB) plus the kernels' static LDS exceed the 64 KB a launch
    // gets by default: raise the kernel's limit once (gfx950 has 160 KB per CU; gpr_create bounds the table size)
    if (lds > 32u * 1024u) {
        static LdsFlag raised[3][GPR_MAX_DEVICES];  // per (kernel, device): the attribute belongs to the device's code object
        const int which = is_reset ? 0 : (P.n_steps > 1 ? 2 : 1);
        const void *fn = is_reset ? reinterpret_cast<const void *>(&k_reset<ST, RT, LPE, JERK, NOISE, FULL, SHAPE>)
                                  : (P.n_steps > 1 ? reinterpret_cast<const void *>(&k_step_n<ST, RT, LPE, JERK, NOISE, FULL, SHAPE>)
                                                   : reinterpret_cast<const void *>(&k_step<ST, RT, LPE, JERK, NOISE, FULL, SHAPE>));
        const hipError_t he = raise_dynamic_lds(fn, 100 * 1024, raised[which]);
        if (he != hipSuccess) return he;
    }
    if (is_reset) {
        hipLaunchKernelGGL((k_reset<ST, RT, LPE, JERK, NOISE, FULL, SHAPE>), dim3(grid), dim3(BLOCK), lds, s, P);
    } else if (P.n_steps > 1) {
        hipLaunchKernelGGL((k_step_n<ST, RT, LPE, JERK, NOISE, FULL, SHAPE>), dim3(grid), dim3(BLOCK), lds, s, P.step_params, P.n_steps);
    } else {
        if constexpr (SHAPE == 0 && FULL && !Math<RT>::literal) {
            if (Q.thin_shift > 0) {  // a thin launch of 8-lane groups: the variant with blocks of 8 cycles (step_body: LONGBLK)
                hipLaunchKernelGGL((k_step_thin<ST, RT, LPE, JERK, NOISE, FULL, SHAPE>), dim3(grid), dim3(BLOCK), lds, s, Q);
                return hipGetLastError();
            }
        }
        hipLaunchKernelGGL((k_step<ST, RT, LPE, JERK, NOISE, FULL, SHAPE>), dim3(grid), dim3(BLOCK), lds, s, Q);
    }
    return hipGetLastError();
    }
}
template <typename ST, typename RT, int LPE, bool JERK, bool NOISE> static hipError_t launch_one(bool is_reset, const KParams &P, hipStream_t s) {
    if (P.c_shape != 0) {  // 'box' collision shape; grids without holes have a fast wall test here too (see gpr_create)
        if constexpr (!Math<RT>::literal) {
            if (P.layout_full) return launch_kernel<ST, RT, LPE, JERK, NOISE, true, 1>(is_reset, P, s);
        }
        return launch_kernel<ST, RT, LPE, JERK, NOISE, false, 1>(is_reset, P, s);
    }
    // grids without holes: the throughput-mode wall check has a table-free form (pos_valid_circle), and the parity-mode step
    // loop evaluates its literal checks only near the outer walls (acceleration actions): separate kernels
    if (P.layout_full && (!Math<RT>::literal || !is_reset)) return launch_kernel<ST, RT, LPE, JERK, NOISE, true, 0>(is_reset, P, s);
    return launch_kernel<ST, RT, LPE, JERK, NOISE, false, 0>(is_reset, P, s);
}
template <typename ST, typename RT, int LPE> static hipError_t launch_flags(bool is_reset, const KParams &P, hipStream_t s) {
    const bool jerk = P.learn_jerk != 0, noise = P.noise != 0;
    if (jerk) {
        // Jerk mode is never run in float arithmetic: under the speed clamp the acceleration is a finite difference of
        // velocities over dt = 1e-3 (PL:643), which turns float's 2.4e-7 velocity grid into 1e-3-sized acceleration errors
        // that the activation integrator then carries (measured: 1.5e-2 m/s^2 after one env-step). gpr_step() routes
        // learn_jerk + GPR_PRECISION_F32 to float storage with double arithmetic.
        if constexpr (Math<RT>::literal) {
            return noise ? launch_one<ST, RT, LPE, true, true>(is_reset, P, s) : launch_one<ST, RT, LPE, true, false>(is_reset, P, s);
        } else {
            return hipErrorInvalidValue;
        }
    }
    return noise ? launch_one<ST, RT, LPE, false, true>(is_reset, P, s) : launch_one<ST, RT, LPE, false, false>(is_reset, P, s);
}
template <typename ST, typename RT> static hipError_t launch_by_lanes(bool is_reset, int lpe, const KParams &P, hipStream_t s) {
    switch (lpe) {
        case 1: return launch_flags<ST, RT, 1>(is_reset, P, s);
        case 2: return launch_flags<ST, RT, 2>(is_reset, P, s);
        case 4: return launch_flags<ST, RT, 4>(is_reset, P, s);
        case 8: return launch_flags<ST, RT, 8>(is_reset, P, s);
        case 16: return launch_flags<ST, RT, 16>(is_reset, P, s);
        default: return hipErrorInvalidValue;
    }
}

#ifndef GPR_ENV_TYPES
#error "define GPR_ENV_TYPES as ST,RT (one translation unit per precision)"
#endif
#if GPR_ENV_PART == 1
template <typename ST, typename RT> hipError_t launch_env_kernel_wide(bool is_reset, int lpe, const KParams &P, hipStream_t s) {
    return launch_by_lanes<ST, RT>(is_reset, lpe, P, s);
}
template hipError_t launch_env_kernel_wide<GPR_ENV_TYPES>(bool, int, const KParams &, hipStream_t);
#else
template <typename ST, typename RT> hipError_t launch_env_kernel(bool is_reset, int lpe, const KParams &P, hipStream_t s) {
    return launch_by_lanes<ST, RT>(is_reset, lpe, P, s);
}
// this translation unit's instance of the normal-draw tables (gpr_device.h)
template <typename ST, typename RT> hipError_t set_normal_table_env(const double *tab) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_normal_tab), &tab, sizeof(tab));
}
template hipError_t set_normal_table_env<GPR_ENV_TYPES>(const double *);
template hipError_t launch_env_kernel<GPR_ENV_TYPES>(bool, int, const KParams &, hipStream_t);
#endif

}  // namespace gpr
