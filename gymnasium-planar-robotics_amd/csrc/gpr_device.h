// gpr_device.h -- device-side building blocks of the gfx950 engine (hand-written HIP, CDNA4 only).
//
// Work decomposition: ONE LANE PER MOVER. An env with M movers occupies LPE = next_pow2(M) adjacent lanes of one
// 64-wide wavefront ("env group"), so 65 536 envs x 4 movers fill 4 096 wavefronts (4 per SIMD) instead of the 1 024 an
// env-per-lane mapping would give. State, actions and observations are row-major [E][2M] (x,y interleaved), so lane
// (env, mover) moves one naturally aligned 8-byte (float2) or 16-byte (double2) element per array and a wavefront
// touches one contiguous 512 B / 1 KiB span per access. Mover<->mover data (the O(M^2) pair sweep) moves through DPP
// lane permutes inside the group, per-env decisions through one wavefront ballot. No MFMA: there is no contraction here.
//
// Reference citations: BE = gymnasium_planar_robotics/envs/basic_envs.py, PL = envs/planning/benchmark_planning_env.py,
// GEO = utils/geometry_2D_utils.py, ROT = utils/rotations_utils.py of ubi-coro/gymnasium-planar-robotics.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gpr {

constexpr uint32_t KIND_STEP_NOISE = 1, KIND_RESET_START = 2, KIND_RESET_GOAL = 3, KIND_RESET_NOISE = 4, KIND_GETTER_NOISE = 5;
#ifndef GPR_BLOCK
#define GPR_BLOCK 256
#endif
constexpr int PUSH_LANES = 64;  // most lanes per env the pushing kernels use (gpr_push_kernels.hip: launch_push3)
constexpr int BLOCK = GPR_BLOCK;  // threads per block of the step / reset / push kernels (a build-time experiment knob)

// Scalar constants in the compute type. The host fills a float and a double copy so that a kernel reads its constants
// straight from the kernarg segment into SGPRs (gfx950 has no scalar f64->f32 convert).
template <typename RT> struct Cst {
    RT dt, inv_dt, v_max, a_max, j_max, thr, sig_p, sig_v, sig_a;
    RT inv_wx, inv_wy;
    RT min_x, min_y, max_x, max_y, min_goal_dist, reward_success;
    RT c_offset, c_offset_wall, tile_hx, tile_hy;
    RT max_csum_step, max_csum_safe;  // max over mover pairs of (c_i + c_j), without / with the safety offset (BE:390,409)
    RT max_b0sum, max_b1sum;          // box shape: max over mover pairs of the summed half sizes, per axis (no offset)
    RT gx0, gx1, gy0, gy1;            // outer edges of the tile grid: lo of the first tile, hi of the last tile, per axis
    RT band_p, band_v;                // how far a noisy position / velocity reading can be from the true one (+ rounding slack)
    RT step1;                         // upper bound of |p(cycle k+1) - p(cycle k)| of one mover: dt (v_max + band_v), with rounding slack
};

// Everything a kernel needs, passed by value (kernarg segment).
struct KParams {
    int64_t E, env_id_offset;
    int32_t M, Tx, Ty, c_shape;
    int32_t num_cycles, max_episode_steps, autoreset, max_reset_trials, learn_jerk, noise, layout_full, eager_noise;
    uint32_t seed_lo, seed_hi, tick_lo, tick_hi;
    Cst<float> cf;
    Cst<double> cd;
    // device tables (one buffer per compute type): axis tables, per-tile LUT, neighbour masks, per-mover sizes
    const void *tables;
    uint32_t table_bytes, off_ax, off_ay, off_lut, off_nbr, off_csz, off_cx, off_cy, off_lutp;
    // state
    void *pos, *vel, *acc, *goal;
    int32_t *elapsed;
    const void *action;
    const uint8_t *mask;
    // gpr_reset_candidates: a caller-supplied stream of candidate sets [n_cand][E][2M] (storage type) instead of the engine's own draws;
    // the start loop consumes sets until it accepts one, the goal loop goes on from there (PL:369-413). cand_used [E] nullable.
    const void *cand;
    int32_t *cand_used;
    int32_t n_cand;
    // outputs
    void *obs, *ag, *dg, *reward;
    uint8_t *terminated, *truncated, *is_success, *mover_collision, *wall_collision;
    void *final_obs, *final_ag, *final_dg;
    // per-agent views written by the step kernel itself (gpr_step_out.agent_observation_dev / agent_reward_dev), nullable
    void *agent_obs, *agent_reward;
    int32_t n_steps;            // gpr_step_n: env-steps per launch (1: gpr_step) ...
    const struct KParams *step_params;  // ... and (host side only) the device array of the n per-step parameter blocks
    uint32_t *counters;         // [wavefronts][6] episode counters (include/gpr.h: GPR_CNT_*), NULL = not counting
    // diagnostics the reference prints as warnings: [0] reset searches that ran out of max_reset_trials (PL:371-375, 397-401: the last,
    // INVALID candidate then stands), [1] env-steps whose action lay outside the action space and was clipped (BE:1871-1873; counted
    // only when count_clipped is set). Rare events: one device-scope atomic add per event / per wavefront with an event; never NULL.
    uint32_t *diag;
    int32_t count_clipped;
    unsigned int *wave_timing;  // debug builds (-DGPR_WAVE_TIMING): 8 words per wavefront of the step kernel, else NULL
    // thin wavefronts (step_body): the most wavefronts a gpr_step launch of this handle may be spread over (host side), and the launch's
    // log2 of "wavefront lanes / lanes that hold movers" (set by launch_kernel)
    int32_t thin_waves, thin_shift;
};
// BenchmarkPushingEnv additions (include/gpr.h: gpr_push_params), in the compute type
template <typename RT> struct PushCst {
    RT inv_mm, inv_mo, inv_Im, inv_Io;  // inverse masses / planar inertias of mover and object
    RT ax, ay, h;                       // half sizes
    RT Kr, Dr;                          // yaw impedance: stiffness, damping 2 sqrt(K m) (IMP:42-47)
    RT damp_lin, damp_rot_den;          // 1 / (1 + dt d / m) and (1 + dt d / I): implicit joint damping (PU:335)
    RT fric_dv, fric_dw;                // velocity lost per cycle to tile friction: mu g dt, torsional
    RT mu, tc;                          // contact friction, contact time constant (penetration recovery)
    RT obj_noise, min_mo_dist, omin_x, omin_y, omax_x, omax_y;
    RT reach2;                          // (circumradius of a mover + circumradius of the object)^2: the contact broad phase
};
struct PushPtrs {
    void *mover_rot, *object;
    const void *opt_start, *opt_goal;   // reset options 'mover_start_xy_pos' [E][2 NM] / 'object_goal_xy_pos' [E][2] (PU:389-391, 412-414), nullable
    const void *cand;                   // gpr_push_reset_candidates: the object's start candidates [n_cand][E][2] (PU:403-405), nullable
    int32_t *cand_used;                 // [E] nullable: candidates consumed
    int32_t n_cand;
    int32_t lane_shift;                 // log2(lanes per env) of this launch (see k_push: small batches spread over more wavefronts)
};
// A kernel reads its KParams (760 bytes = 12 lines of the scalar cache) piecemeal: the compiler loads a field where it is first needed and
// waits for it there -- nine dependent round trips before the first cycle of the step kernel, each one a miss (every launch has a fresh
// kernel-argument segment). warm_kernarg() asks for one word of every line at once and waits once: the loads that follow hit the scalar
// cache. (The words land in registers the compiler knows nothing about after the block, hence the wait inside it.)
__device__ __forceinline__ void warm_kernarg() {
    static_assert(sizeof(KParams) <= 12 * 64, "one load per 64-byte line of KParams");
    auto kp = __builtin_amdgcn_kernarg_segment_ptr();
    uint32_t w0, w1, w2, w3, w4, w5, w6, w7, w8, w9, w10, w11;
    asm volatile("s_load_dword %0, %12, 0x0\n\ts_load_dword %1, %12, 0x40\n\ts_load_dword %2, %12, 0x80\n\ts_load_dword %3, %12, 0xc0\n\t"
                 "s_load_dword %4, %12, 0x100\n\ts_load_dword %5, %12, 0x140\n\ts_load_dword %6, %12, 0x180\n\ts_load_dword %7, %12, 0x1c0\n\t"
                 "s_load_dword %8, %12, 0x200\n\ts_load_dword %9, %12, 0x240\n\ts_load_dword %10, %12, 0x280\n\ts_load_dword %11, %12, 0x2c0\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(w0), "=&s"(w1), "=&s"(w2), "=&s"(w3), "=&s"(w4), "=&s"(w5), "=&s"(w6), "=&s"(w7), "=&s"(w8), "=&s"(w9), "=&s"(w10), "=&s"(w11)
                 : "s"(kp) : "memory");
}
// thin wavefronts: log2 of "wavefront lanes / lanes that hold movers" for a launch of E envs in groups of lpe lanes under a budget of
// `budget` wavefronts -- the largest spread that keeps whole env groups in a wavefront and the launch within the budget (0: whole wavefronts)
inline int thin_shift_for(int64_t E, int lpe, int64_t budget) {
    const int64_t natural = (E * lpe + 63) / 64;
    int sh = 0;
    while ((64 >> (sh + 1)) >= lpe && (natural << (sh + 1)) <= budget) ++sh;
    return sh;
}
template <typename RT> __host__ __device__ __forceinline__ const Cst<RT> &cst(const KParams &P);
template <> __host__ __device__ __forceinline__ const Cst<float> &cst<float>(const KParams &P) { return P.cf; }
template <> __host__ __device__ __forceinline__ const Cst<double> &cst<double>(const KParams &P) { return P.cd; }

// ---------------------------------------------------------------------------------------------------------------------
// small vector + math traits
// ---------------------------------------------------------------------------------------------------------------------
template <typename T> struct V2;
template <> struct V2<float> { using type = float2; };
template <> struct V2<double> { using type = double2; };

template <typename RT> struct R2 { RT x, y; };

template <typename ST, typename RT> __device__ __forceinline__ R2<RT> load2(const void *base, int64_t idx) {
    typename V2<ST>::type v = *reinterpret_cast<const typename V2<ST>::type *>(static_cast<const ST *>(base) + idx);
    return R2<RT>{(RT)v.x, (RT)v.y};
}
// State and observation stores are WRITE-THROUGH (sc1): they leave the CU while the other wavefronts still compute, and the
// kernel ends without megabytes of dirty L2 lines to write back (a dependent kernel boundary costs its predecessor's dirty
// bytes / 6 TB/s on this machine: MI355X_MICROARCH.md, row "boundary" -- ~1.9 us for the 11 MB one step writes).
template <typename ST, typename RT> __device__ __forceinline__ void store2(void *base, int64_t idx, R2<RT> v) {
    typename V2<ST>::type o;
    o.x = (ST)v.x;
    o.y = (ST)v.y;
#ifdef GPR_PLAIN_STORES
    *reinterpret_cast<typename V2<ST>::type *>(static_cast<ST *>(base) + idx) = o;
#else
    if constexpr (sizeof(ST) == 4) {
        unsigned long long bits;
        __builtin_memcpy(&bits, &o, 8);
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(static_cast<ST *>(base) + idx), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        // ONE 16-byte store (two 8-byte write-through ones were counted -- and written -- as two partial sectors: the f64 kernel's
        // WRITE_SIZE was twice its algorithmic bytes, profiles/round3_pmc_summary.txt); idx is even: 16-byte aligned. A plain store:
        // the write-through form of this width exists only as inline assembly, whose register hazards the compiler cannot see.
        typedef double D2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<D2 *>(static_cast<ST *>(base) + idx) = D2{(double)o.x, (double)o.y};
    }
#endif
}
// value as it will read back from storage (float storage under double arithmetic rounds here)
template <typename ST, typename RT> __device__ __forceinline__ RT stored(RT v) { return (RT)(ST)v; }

template <typename RT> struct Math;
template <> struct Math<float> {
    static constexpr bool literal = false;  // throughput mode: squared-norm compares, rsqrt, reciprocal of dt
    static __device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
    static __device__ __forceinline__ float rsqrt_(float x) { return __builtin_amdgcn_rsqf(x); }
    static __device__ __forceinline__ int floor_int(float x) { return __float2int_rd(x); }
    static __device__ __forceinline__ float inf() { return __builtin_huge_valf(); }
    static __device__ __forceinline__ float div_(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }  // 1 ulp reciprocal
    static __device__ __forceinline__ void sincos_(float x, float &s, float &c) {  // v_sin / v_cos take revolutions
        const float r = x * 0.15915494309189535f;
        s = __builtin_amdgcn_sinf(r); c = __builtin_amdgcn_cosf(r);
    }
};
template <> struct Math<double> {
    static constexpr bool literal = true;  // parity mode: the reference's operations in the reference's order
    static __device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }
    static __device__ __forceinline__ double rsqrt_(double x) { return 1.0 / sqrt(x); }
    static __device__ __forceinline__ int floor_int(double x) { return __double2int_rd(x); }
    static __device__ __forceinline__ double inf() { return __builtin_huge_val(); }
    static __device__ __forceinline__ double div_(double a, double b) { return a / b; }
    static __device__ __forceinline__ void sincos_(double x, double &s, double &c) { sincos(x, &s, &c); }
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
template <typename RT> __device__ __forceinline__ RT absr(RT v) { return v < (RT)0 ? -v : v; }

// ---------------------------------------------------------------------------------------------------------------------
// RNG spec "gpr-rng-v2": counter based, 32-bit ops. key = hash(seed, tick, kind, env, mover); draw k = lowbias32(key +
// (k+1)*golden); normal pair q = Box-Muller on the upper 13 / lower 16 bits of draw q. Mirrors oracle/gpr_oracle.c (gpro_rng_*).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t rng_key(const KParams &P, uint32_t kind, int64_t env_global, uint32_t mover) {
    uint32_t h = 0x9E3779B9U;  // the first five rounds are wave-uniform (scalar ALU)
    h = mix32(h ^ P.seed_lo);
    h = mix32(h ^ P.seed_hi);
    h = mix32(h ^ P.tick_lo);
    h = mix32(h ^ P.tick_hi);
    h = mix32(h ^ kind);
    h = mix32(h ^ (uint32_t)env_global);
    h = mix32(h ^ (uint32_t)((uint64_t)env_global >> 32));
    h = mix32(h ^ mover);
    return h;
}
__device__ __forceinline__ uint32_t rng_draw(uint32_t key, uint32_t k) { return mix32(key + (k + 1u) * 0x9E3779B9U); }

template <typename RT> __device__ __forceinline__ RT rng_uniform(uint32_t key, uint32_t k) {  // [0,1), 24 bits
    return (RT)(rng_draw(key, k) >> 8) * (RT)(1.0 / 16777216.0);
}
// gpr-rng-v2 (normal pairs; keys, draws and uniforms are v1's): ONE 32-bit draw per pair, the Box-Muller radius from its upper 13 bits --
// u1 = (i + 1) / 8192, radius sqrt(-2 ln u1) <= sqrt(26 ln 2): each component of a normal pair is bounded --, the angle 2 pi k / 65536
// from its lower 16.
constexpr double NORMAL_MAX_RADIUS = 4.245263256915119;  // sqrt(26 ln 2)
template <typename RT> __device__ __forceinline__ R2<RT> rng_normal_pair(uint32_t key, uint32_t q);
template <> __device__ __forceinline__ R2<float> rng_normal_pair<float>(uint32_t key, uint32_t q) {
    const uint32_t r = rng_draw(key, q);
    const float a = (float)((r >> 19) + 1u);  // (0, 8192] -> u1 = a * 2^-13
    const float u2 = (float)(r & 0xFFFFu) * (1.0f / 65536.0f);
    // v_log_f32 is log2: -2 ln(u1) = -2 ln2 (log2(a) - 13). v_sin / v_cos take revolutions, so 2*pi*u2 needs no multiply.
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * (__builtin_amdgcn_logf(a) - 13.0f));
    return R2<float>{rad * __builtin_amdgcn_cosf(u2), rad * __builtin_amdgcn_sinf(u2)};
}
// Double precision: log / sin / cos in double are several hundred instructions per draw, and the arguments take few values. So the
// radius comes from a table of 8 192 doubles and (cos, sin) from a table over ONE OCTANT, (cos, sin)(2 pi j / 65536) for j = 0 .. 8192,
// unfolded by symmetry -- 192 KB, filled once per device by gpr_create with the host's libm: the very values the oracle's tables hold
// (oracle/gpr_oracle.c: gpro_rng_normal_pair does the same unfolding), so the noise is bit-identical to the oracle's. (v1's tables
// covered the whole circle and 16 bits of radius: 1.5 MB, fetched through every XCD's L2 by every launch -- 12 MB of fabric traffic
// per 16 384-env launch of the f64 kernel, more than the launch's own 9.8 MB.)
// One translation unit = one instance of this pointer (set through set_normal_table_* of that unit).
static __device__ const double *g_normal_tab = nullptr;  // [8192] radius, then [8193][2] (cos, sin) of the first octant
template <> __device__ __forceinline__ R2<double> rng_normal_pair<double>(uint32_t key, uint32_t q) {
    const uint32_t r = rng_draw(key, q);
    const double *tab = g_normal_tab;
    const double rad = tab[r >> 19];
    const uint32_t k = r & 0xFFFFu, oct = k >> 13, j = k & 8191u;
    const double2 t = reinterpret_cast<const double2 *>(tab + 8192)[(oct & 1u) ? 8192u - j : j];  // the octant's own angle, or its mirror image
    const bool swap = (((oct + 1u) >> 1) & 1u) != 0u;       // octants 1, 2, 5, 6: cos and sin change places
    double c = swap ? t.y : t.x, s = swap ? t.x : t.y;
    c = (((oct + 2u) >> 2) & 1u) != 0u ? -c : c;             // octants 2 .. 5: cos < 0
    s = (oct >> 2) != 0u ? -s : s;                           // octants 4 .. 7: sin < 0
    return R2<double>{rad * c, rad * s};
}

__device__ __forceinline__ void diag_add(const KParams &P, int which, uint32_t n) {
    __hip_atomic_fetch_add(P.diag + which, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// a noisy 2-vector reading: value + sigma * N(0,1) per component (BE:828,841), from normal pair q of the lane's stream
template <typename RT, bool NOISE> __device__ __forceinline__ R2<RT> noisy(R2<RT> v, RT sigma, uint32_t key, uint32_t q) {
    if constexpr (NOISE) {
        R2<RT> n = rng_normal_pair<RT>(key, q);
        return R2<RT>{v.x + sigma * n.x, v.y + sigma * n.y};  // BE:828,841: value + N(0, sigma)
    } else {
        return v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// lane permutes inside an env group of LPE lanes (LPE in {1,2,4,8,16}; a group never straddles a 16-lane DPP row)
// ---------------------------------------------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
// value of lane (group_base + (mover + D) % LPE)
template <int LPE, int D> __device__ __forceinline__ uint32_t rot_u32(uint32_t v) {
    static_assert(D >= 1 && D < LPE, "rotation distance");
    if constexpr (LPE == 2) {
        return dpp_u32<0xB1>(v);  // quad_perm [1,0,3,2]
    } else if constexpr (LPE == 4) {
        if constexpr (D == 1) return dpp_u32<0x39>(v);  // quad_perm [1,2,3,0]
        else if constexpr (D == 2) return dpp_u32<0x4E>(v);  // quad_perm [2,3,0,1]
        else return dpp_u32<0x93>(v);  // quad_perm [3,0,1,2]
    } else if constexpr (LPE == 16) {
        return dpp_u32<0x120 + (16 - D)>(v);  // row_ror:(16-D): lane i receives lane (i + D) % 16 of its row
    } else {  // LPE == 8: no 8-lane DPP rotate; LDS-crossbar permute (no LDS memory traffic)
        int lane = (int)__lane_id();
        int src = (lane & ~(LPE - 1)) | ((lane + D) & (LPE - 1));
        return (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)v);
    }
}
template <int LPE, int D> __device__ __forceinline__ float rot(float v) { return __uint_as_float(rot_u32<LPE, D>(__float_as_uint(v))); }
template <int LPE, int D> __device__ __forceinline__ double rot(double v) {
    uint64_t b = (uint64_t)__double_as_longlong(v);
    uint32_t lo = rot_u32<LPE, D>((uint32_t)b), hi = rot_u32<LPE, D>((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// the largest value any lane of this lane's env group holds (a cyclic all-reduce: rotations by 1, 2, 4, 8)
template <int LPE> __device__ __forceinline__ float group_max(float v) {
    if constexpr (LPE >= 2) v = __builtin_fmaxf(v, rot<LPE, 1>(v));
    if constexpr (LPE >= 4) v = __builtin_fmaxf(v, rot<LPE, 2>(v));
    if constexpr (LPE >= 8) v = __builtin_fmaxf(v, rot<LPE, 4>(v));
    if constexpr (LPE >= 16) v = __builtin_fmaxf(v, rot<LPE, 8>(v));
    return v;
}

// (rot(x) - x, rot(y) - y) with the lane permute folded into the subtract: one v_sub_f32_dpp per component instead of
// v_mov_b32_dpp + v_sub (the compiler pairs the two subtracts into a v_pk_add_f32, which cannot carry a DPP modifier, and
// then needs the moves). s_nop 1: a DPP read needs two wait states after the VALU write of its source.
#define GPR_ROT_DIFF2(CTRL)                                                                                              \
    asm("s_nop 1\n\tv_sub_f32_dpp %0, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                      \
        "v_sub_f32_dpp %1, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1"                                      \
        : "=&v"(dx), "=&v"(dy) : "v"(x), "v"(y))
template <int LPE, int D> __device__ __forceinline__ void rot_diff2(float x, float y, float &dx, float &dy) {
    if constexpr (LPE == 2) { GPR_ROT_DIFF2("quad_perm:[1,0,3,2]"); }
    else if constexpr (LPE == 4 && D == 1) { GPR_ROT_DIFF2("quad_perm:[1,2,3,0]"); }
    else if constexpr (LPE == 4 && D == 2) { GPR_ROT_DIFF2("quad_perm:[2,3,0,1]"); }
    else if constexpr (LPE == 16 && D == 1) { GPR_ROT_DIFF2("row_ror:15"); }
    else if constexpr (LPE == 16 && D == 2) { GPR_ROT_DIFF2("row_ror:14"); }
    else if constexpr (LPE == 16 && D == 3) { GPR_ROT_DIFF2("row_ror:13"); }
    else if constexpr (LPE == 16 && D == 4) { GPR_ROT_DIFF2("row_ror:12"); }
    else if constexpr (LPE == 16 && D == 5) { GPR_ROT_DIFF2("row_ror:11"); }
    else if constexpr (LPE == 16 && D == 6) { GPR_ROT_DIFF2("row_ror:10"); }
    else if constexpr (LPE == 16 && D == 7) { GPR_ROT_DIFF2("row_ror:9"); }
    else if constexpr (LPE == 16 && D == 8) { GPR_ROT_DIFF2("row_ror:8"); }
    else { dx = rot<LPE, D>(x) - x; dy = rot<LPE, D>(y) - y; }
}
#undef GPR_ROT_DIFF2
template <int LPE, int D> __device__ __forceinline__ void rot_diff2(double x, double y, double &dx, double &dy) {
    dx = rot<LPE, D>(x) - x; dy = rot<LPE, D>(y) - y;
}
// rotations D and D + 1 in ONE block: the sources are not written inside it, so the wait states are paid once for the four subtracts
#define GPR_ROT_DIFF4(C0, C1)                                                                                            \
    asm("s_nop 1\n\tv_sub_f32_dpp %0, %4, %4 " C0 " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                        \
        "v_sub_f32_dpp %1, %5, %5 " C0 " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                    \
        "v_sub_f32_dpp %2, %4, %4 " C1 " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                    \
        "v_sub_f32_dpp %3, %5, %5 " C1 " row_mask:0xf bank_mask:0xf bound_ctrl:1"                                        \
        : "=&v"(dx0), "=&v"(dy0), "=&v"(dx1), "=&v"(dy1) : "v"(x), "v"(y))
template <int LPE, int D> __device__ __forceinline__ void rot_diff4(float x, float y, float &dx0, float &dy0, float &dx1, float &dy1) {
    if constexpr (LPE == 4 && D == 1) { GPR_ROT_DIFF4("quad_perm:[1,2,3,0]", "quad_perm:[2,3,0,1]"); }
    else if constexpr (LPE == 16 && D == 1) { GPR_ROT_DIFF4("row_ror:15", "row_ror:14"); }
    else if constexpr (LPE == 16 && D == 3) { GPR_ROT_DIFF4("row_ror:13", "row_ror:12"); }
    else if constexpr (LPE == 16 && D == 5) { GPR_ROT_DIFF4("row_ror:11", "row_ror:10"); }
    else if constexpr (LPE == 16 && D == 7) { GPR_ROT_DIFF4("row_ror:9", "row_ror:8"); }
    else { rot_diff2<LPE, D>(x, y, dx0, dy0); rot_diff2<LPE, D + 1>(x, y, dx1, dy1); }
}
#undef GPR_ROT_DIFF4
template <int LPE, int D> __device__ __forceinline__ void rot_diff4(double x, double y, double &dx0, double &dy0, double &dx1, double &dy1) {
    rot_diff2<LPE, D>(x, y, dx0, dy0); rot_diff2<LPE, D + 1>(x, y, dx1, dy1);
}

// bits of the predicate for the LPE lanes of this lane's env group (bit m = mover m)
template <int LPE> __device__ __forceinline__ uint32_t group_bits(bool pred) {
    unsigned long long b = __ballot(pred);
    if constexpr (LPE == 1) {
        return (uint32_t)pred;
    } else {
        int base = (int)__lane_id() & ~(LPE - 1);
        return (uint32_t)(b >> base) & ((1u << LPE) - 1u);
    }
}
// Per-env decisions as WAVE MASKS: one 64-bit scalar value holds one bit per lane, so "any lane of my env group" and the
// frozen / collision bookkeeping of the 40-cycle loop are a handful of scalar-ALU instructions per wavefront instead of
// vector instructions per lane. group_or: every bit of a group is set iff any bit of the group was set.
template <int LPE> __device__ __forceinline__ unsigned long long group_or(unsigned long long b) {
    if constexpr (LPE == 1) {
        return b;
    } else {
        constexpr unsigned long long base = LPE == 2 ? 0x5555555555555555ull : LPE == 4 ? 0x1111111111111111ull
                                          : LPE == 8 ? 0x0101010101010101ull : 0x0001000100010001ull;
        unsigned long long g = b | (b >> 1);
        if constexpr (LPE >= 4) g |= g >> 2;
        if constexpr (LPE >= 8) g |= g >> 4;
        if constexpr (LPE >= 16) g |= g >> 8;
        g &= base;                                 // bit 0 of each group = OR of the group
        return g * ((1ull << LPE) - 1ull);         // replicate it over the group (no carries: groups are LPE bits apart)
    }
}
template <int LPE> __device__ __forceinline__ unsigned long long group_any_mask(bool pred) { return group_or<LPE>(__ballot(pred)); }
__device__ __forceinline__ bool lane_of(unsigned long long mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }
template <int LPE> __device__ __forceinline__ bool group_any(bool pred) { return lane_of(group_any_mask<LPE>(pred)); }

// ---------------------------------------------------------------------------------------------------------------------
// tile tables (built on the host by gpr_create, staged into LDS by every block)
//   axis table, per tile i: {lo, hi, prev_hi, next_lo} = {c_i - h, c_i + h, c_{i-1} + h (-inf), c_{i+1} - h (+inf)}
//     computed in f64 exactly as BE:509-512 / BE:522-526 do, then rounded to the compute type
//   lut, per tile: 16 x 2 bits: mask_valid (BE:580-648, capped at 3) for each combination of the four "safe" compares
//   nbr, per tile: 9 bits: which of the 3x3 neighbourhood tiles exist (bit = (dx+1)*3 + (dy+1))
// ---------------------------------------------------------------------------------------------------------------------
template <typename RT> struct Tables {
    const RT *ax, *ay;    // [T][4]
    const uint32_t *lut;  // [Tx*Ty]
    const uint32_t *lutp; // [(Tx+2)*(Ty+2)]: lut with a ring of zeros ("no tile") around it, for clamped lookups
    const uint16_t *nbr;  // [Tx*Ty]
    const RT *csz;        // [GPR_MAX_MOVERS][2]
    const RT *cx, *cy;    // tile centres [Tx], [Ty] (box shape only)
};
template <typename RT> __device__ __forceinline__ Tables<RT> make_tables(const void *base, const KParams &P) {
    const char *b = static_cast<const char *>(base);
    Tables<RT> T;
    T.ax = reinterpret_cast<const RT *>(b + P.off_ax);
    T.ay = reinterpret_cast<const RT *>(b + P.off_ay);
    T.lut = reinterpret_cast<const uint32_t *>(b + P.off_lut);
    T.nbr = reinterpret_cast<const uint16_t *>(b + P.off_nbr);
    T.csz = reinterpret_cast<const RT *>(b + P.off_csz);
    T.cx = reinterpret_cast<const RT *>(b + P.off_cx);
    T.cy = reinterpret_cast<const RT *>(b + P.off_cy);
    T.lutp = reinterpret_cast<const uint32_t *>(b + P.off_lutp);
    return T;
}
__device__ __forceinline__ void stage_tables(void *lds, const void *global, uint32_t bytes) {
    uint32_t *d = static_cast<uint32_t *>(lds);
    const uint32_t *s = static_cast<const uint32_t *>(global);
    for (uint32_t i = threadIdx.x; i < bytes / 4u; i += blockDim.x) d[i] = s[i];
    __syncthreads();
}

// one tile, circle shape (BE:545-558 + BE:580-653): four strict compares select 2 bits of the tile's LUT
template <typename RT>
__device__ __forceinline__ bool tile_valid_circle(const Tables<RT> &T, int Ty, int ix, int iy, RT xlo, RT xhi, RT ylo, RT yhi, RT x,
                                                  RT y, RT c) {
    uint32_t bits = (uint32_t)(xlo < x - c) | ((uint32_t)(x + c < xhi) << 1) | ((uint32_t)(ylo < y - c) << 2) |
                    ((uint32_t)(y + c < yhi) << 3);
    uint32_t val = (T.lut[ix * Ty + iy] >> (2u * bits)) & 3u;
    return val == 1u;
}

// collision size of one circle as the wall check wants it: metres, and (throughput mode) in tile widths
template <typename RT> struct CircleSize {
    RT c, cfx, cfy;  // radius in metres and in tile widths
    RT wx, wy;       // full-grid test: how far the centre may be from the grid centre, per axis
};
template <typename RT> __device__ __forceinline__ CircleSize<RT> circle_size(const KParams &P, RT c) {
    const Cst<RT> &C = cst<RT>(P);
    return CircleSize<RT>{c, c * C.inv_wx, c * C.inv_wy, (RT)0.5 * (C.gx1 - C.gx0) - c, (RT)0.5 * (C.gy1 - C.gy0) - c};
}

// Full grid (every cell holds a tile, circle smaller than half a tile): how far the circle's centre reaches from the grid
// centre, as ONE number comparable with cs.wx: max(|x - cx|, |y - cy| + (wx - wy)). The position is valid iff it is < cs.wx
// (both axis margins positive). Three vector instructions after the centre subtraction.
template <typename RT> __device__ __forceinline__ RT wall_reach_centred(RT qx, RT qy, const CircleSize<RT> &cs) {  // q = p - grid centre
    return __builtin_fmaxf(__builtin_fabsf(qx), __builtin_fabsf(qy) + (cs.wx - cs.wy));
}
template <typename RT> __device__ __forceinline__ RT wall_reach_full(const Cst<RT> &C, RT x, RT y, const CircleSize<RT> &cs) {
    return wall_reach_centred<RT>(x - (RT)0.5 * (C.gx0 + C.gx1), y - (RT)0.5 * (C.gy0 + C.gy1), cs);
}

// generic (any number of tiles under the centre, incl. none) form of BE:518-540 + BE:785, table edges compared literally
template <typename RT> __device__ __forceinline__ bool pos_valid_circle_slow(const Tables<RT> &T, int Tx, int Ty, int ix, int iy, RT x, RT y, RT c) {
    bool any = false, all = true, valid3 = false;
    for (int a = ix - 1; a <= ix + 1; ++a) {
        if (a < 0 || a >= Tx) continue;
        const RT *fa = T.ax + 4 * a;
        bool in_x = (fa[0] <= x) & (x <= fa[1]);
        for (int b = iy - 1; b <= iy + 1; ++b) {
            if (b < 0 || b >= Ty) continue;
            const RT *fb = T.ay + 4 * b;
            bool above = in_x & (fb[0] <= y) & (y <= fb[1]);
            if (above) {
                any = true;
                valid3 |= (T.nbr[a * Ty + b] == 0x1FFu);  // centre of a complete 3x3 block, BE:530-538
                all &= tile_valid_circle(T, Ty, a, b, fa[0], fa[1], fb[0], fb[1], x, y, c);
            }
        }
    }
    return valid3 | (any & all);
}

// qpos_is_valid for one circle (BE:459-788 with ignore_orientation). Must be called by all lanes of the wavefront
// convergently (it votes on whether the single-tile fast path holds for everyone).
// UNC: also report in `uncertain` whether a displacement of the centre by up to `band` in x and in y could change the
// verdict (or the path taken to it). Conservative: false guarantees that every such displacement yields the same verdict.
template <typename RT, bool FULL, bool UNC = false>
__device__ __forceinline__ bool pos_valid_circle(const Tables<RT> &T, const KParams &P, RT x, RT y, CircleSize<RT> cs, RT band = (RT)0,
                                                 bool *uncertain = nullptr) {
    const int Tx = P.Tx, Ty = P.Ty;
    if constexpr (Math<RT>::literal) {
        if constexpr (FULL) {
            // grid without holes: BE:459-788 is "the circle's bounding box lies strictly inside the grid" (see the throughput
            // branch); the table edges are the reference's to within rounding, so a margin beyond 1e-9 (+ band) decides, for
            // this position and for every read displaced by up to `band` -- only a wavefront with a lane closer than that to the
            // outer limit walks the tables
            const Cst<RT> &C = cst<RT>(P);
            const RT mx = cs.wx - absr(x - (RT)0.5 * (C.gx0 + C.gx1)), my = cs.wy - absr(y - (RT)0.5 * (C.gy0 + C.gy1));
            const RT m = mx < my ? mx : my, room = band + (RT)1e-9;
            const bool in = m > room, out = m < -room;
            if (__ballot(!(in | out)) == 0ull) {
                if constexpr (UNC) *uncertain = false;
                return in;
            }
        }
        // parity mode: every compare against the tile edges exactly as NumPy formed them
        const RT c = cs.c;
        int ix = clampi(Math<RT>::floor_int(x * cst<RT>(P).inv_wx), 0, Tx - 1);
        int iy = clampi(Math<RT>::floor_int(y * cst<RT>(P).inv_wy), 0, Ty - 1);
        const RT *ex = T.ax + 4 * ix, *ey = T.ay + 4 * iy;
        RT xlo = ex[0], xhi = ex[1], xph = ex[2], xnl = ex[3];
        RT ylo = ey[0], yhi = ey[1], yph = ey[2], ynl = ey[3];
        // exactly one tile contains the point, and it is the guessed one ("above a tile" is edge-inclusive, BE:508-513)
        bool single = (xlo <= x) & (x <= xhi) & (xph < x) & (x < xnl) & (ylo <= y) & (y <= yhi) & (yph < y) & (y < ynl);
        if constexpr (UNC) {
            *uncertain = !single | (x - xlo <= band) | (xhi - x <= band) | (y - ylo <= band) | (yhi - y <= band) |
                         (absr((x - c) - xlo) <= band) | (absr(xhi - (x + c)) <= band) | (absr((y - c) - ylo) <= band) |
                         (absr(yhi - (y + c)) <= band);
        }
        if (__ballot(!single) == 0ull) return tile_valid_circle(T, Ty, ix, iy, xlo, xhi, ylo, yhi, x, y, c);
        return pos_valid_circle_slow<RT>(T, Tx, Ty, ix, iy, x, y, c);  // rare: on a tile edge, outside the grid, NaN
    } else {
        const Cst<RT> &C = cst<RT>(P);
        if constexpr (FULL) {
            // Every cell of the grid holds a tile and the circle is smaller than half a tile: an edge between two tiles
            // never invalidates (BE:582-648 find the neighbour), so BE:459-788 reduces to "the circle's bounding box lies
            // strictly inside the grid": the smaller of the two axis margins is positive. No table.
            const RT e = wall_reach_full<RT>(C, x, y, cs);
            if constexpr (UNC) *uncertain = !(__builtin_fabsf(cs.wx - e) > band);
            return e < cs.wx;
        }
        // General layouts, in tile coordinates: fx = x / tile_width; its integer part names the tile, its fraction is the
        // position inside it, and the four "safe" compares of BE:545-558 become frac > c/w and frac < 1 - c/w. One LDS read
        // (the tile's LUT, from a zero-padded copy so that any position outside the grid reads "no tile"), issued early and
        // consumed late. This shortcut is exact only while the centre is at least EPS tile widths away from every tile edge
        // (EPS >> the float rounding of x / w, ~3e-6): then the tile named by the integer part is provably the only tile the
        // reference would find the centre "above". Otherwise (rare) the whole wavefront re-evaluates with the literal test.
        constexpr RT EPS = (RT)2e-5;
        const RT fx = x * C.inv_wx, fy = y * C.inv_wy;
        const RT flx = __builtin_floorf(fx), fly = __builtin_floorf(fy);
        const RT frx = fx - flx, fry = fy - fly;
        const int ix = (int)flx, iy = (int)fly;
        const int ixc = clampi(ix, -1, Tx), iyc = clampi(iy, -1, Ty);
        const uint32_t lut = T.lutp[__umul24((unsigned)(ixc + 1), (unsigned)(Ty + 2)) + (unsigned)(iyc + 1)];
        const uint32_t sh = (frx > cs.cfx ? 2u : 0u) | (frx < (RT)1 - cs.cfx ? 4u : 0u) | (fry > cs.cfy ? 8u : 0u) | (fry < (RT)1 - cs.cfy ? 16u : 0u);
        const RT edge = __builtin_fmaxf(__builtin_fabsf(frx - (RT)0.5), __builtin_fabsf(fry - (RT)0.5));  // 0.5 = on a tile edge
        const bool inside = edge < (RT)0.5 - EPS;
        if constexpr (UNC) {
            const RT bx = band * C.inv_wx, by = band * C.inv_wy, bt = __builtin_fmaxf(bx, by);
            *uncertain = !(edge < (RT)0.5 - EPS - bt) | (__builtin_fabsf(frx - cs.cfx) <= bx) | (__builtin_fabsf(frx - ((RT)1 - cs.cfx)) <= bx) |
                         (__builtin_fabsf(fry - cs.cfy) <= by) | (__builtin_fabsf(fry - ((RT)1 - cs.cfy)) <= by);
        }
        bool valid = ((lut >> sh) & 3u) == 1u;
        if (__ballot(!inside) != 0ull) valid = pos_valid_circle_slow<RT>(T, Tx, Ty, clampi(ix, 0, Tx - 1), clampi(iy, 0, Ty - 1), x, y, cs.c);
        return valid;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// geometry_2D_utils (GEO:9-138) + the rotation helpers it uses (ROT:248-274, ROT:414-461), for the 'box' collision shape
// ---------------------------------------------------------------------------------------------------------------------
// orientation determinant of [[ax,bx,cx],[ay,by,cy],[1,1,1]] (GEO:47-60 feed this matrix to numpy.linalg.det)
template <typename RT> __device__ __forceinline__ RT det3(RT ax, RT ay, RT bx, RT by, RT cx, RT cy) {
    return ax * (by - cy) - bx * (ay - cy) + cx * (ay - by);
}

// GEO:9-69 for one pair of segments p1p2, q1q2
template <typename RT>
__device__ __forceinline__ bool seg_intersect(RT p1x, RT p1y, RT p2x, RT p2y, RT q1x, RT q1y, RT q2x, RT q2y) {
    const RT eps = (RT)1e-7;
    // GEO:30-35: a shared end point always counts
    bool eq = ((absr(p1x - q1x) < eps) & (absr(p1y - q1y) < eps)) | ((absr(p1x - q2x) < eps) & (absr(p1y - q2y) < eps)) |
              ((absr(p2x - q1x) < eps) & (absr(p2y - q1y) < eps)) | ((absr(p2x - q2x) < eps) & (absr(p2y - q2y) < eps));
    // GEO:37-45: bounding boxes apart (by more than eps) in x or y
    RT minpx = fmin(p1x, p2x), maxpx = fmax(p1x, p2x), minqx = fmin(q1x, q2x), maxqx = fmax(q1x, q2x);
    RT minpy = fmin(p1y, p2y), maxpy = fmax(p1y, p2y), minqy = fmin(q1y, q2y), maxqy = fmax(q1y, q2y);
    int mm = (int)((maxpx < minqx) & !(absr(maxpx - minqx) < eps)) + (int)((maxqx < minpx) & !(absr(maxqx - minpx) < eps)) +
             (int)((maxpy < minqy) & !(absr(maxpy - minqy) < eps)) + (int)((maxqy < minpy) & !(absr(maxqy - minpy) < eps));
    // GEO:52-64: orientation test with 1e-7 tolerance on the determinant products
    RT pp = det3(p1x, p1y, p2x, p2y, q1x, q1y) * det3(p1x, p1y, p2x, p2y, q2x, q2y);
    RT qq = det3(q1x, q1y, q2x, q2y, p1x, p1y) * det3(q1x, q1y, q2x, q2y, p2x, p2y);
    bool orient = ((pp <= (RT)0) | (absr(pp) < eps)) & ((qq <= (RT)0) | (absr(qq) < eps));
    bool res = orient;      // GEO:66
    if (mm >= 1) res = false;  // GEO:67
    if (eq) res = true;        // GEO:68
    return res;
}

// GEO:72-104: the four (x,y) vertices of a yaw-rotated rectangle in the order (-,-), (-,+), (+,+), (+,-).
// q = (w,x,y,z). The quaternion is normalised in FLOAT32 (unit_vector casts, ROT:447), the matrix is built in RT.
template <typename RT>
__device__ __forceinline__ void rect_vertices(RT px, RT py, RT qw, RT qx, RT qy, RT qz, RT sx, RT sy, RT *vx, RT *vy) {
    float f0 = (float)qw, f1 = (float)qx, f2 = (float)qy, f3 = (float)qz;
    float len = ((f0 * f0 + f1 * f1) + f2 * f2) + f3 * f3;
    len = sqrtf(len);
    f0 = f0 / len; f1 = f1 / len; f2 = f2 / len; f3 = f3 / len;
    RT w = (RT)f0, x = (RT)f1, y = (RT)f2, z = (RT)f3;
    RT Nq = ((w * w + x * x) + y * y) + z * z;  // ROT:256
    RT s = (RT)2.0 / Nq;
    RT X = x * s, Y = y * s, Z = z * s;
    RT wZ = w * Z, xX = x * X, xY = x * Y, yY = y * Y, zZ = z * Z;
    RT R00 = (RT)1.0 - (yY + zZ), R01 = xY - wZ, R10 = xY + wZ, R11 = (RT)1.0 - (xX + zZ);
    const RT feps = sizeof(RT) == 8 ? (RT)2.220446049250313e-16 : (RT)1.1920929e-07f;
    if (!(Nq > feps)) { R00 = (RT)1; R01 = (RT)0; R10 = (RT)0; R11 = (RT)1; }  // ROT:274
    const RT lx[4] = {-sx, -sx, sx, sx}, ly[4] = {-sy, sy, sy, -sy};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // rot_mats @ vertices_l (GEO:102), k-ordered and unfused like NumPy's batched matmul loop; the z term is R*0
        vx[j] = (R00 * lx[j] + R01 * ly[j]) + px;
        vy[j] = (R10 * lx[j] + R11 * ly[j]) + py;
    }
}

// GEO:107-138: any of the 4 x 4 edge pairs intersects (containment without edge crossing is NOT an intersection)
template <typename RT> __device__ __forceinline__ bool rects_intersect_v(const RT *ax, const RT *ay, const RT *bx, const RT *by) {
    bool any = false;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            any |= seg_intersect<RT>(ax[i], ay[i], ax[(i + 1) & 3], ay[(i + 1) & 3], bx[j], by[j], bx[(j + 1) & 3], by[(j + 1) & 3]);
    return any;
}

// one tile, box shape (BE:560-572, 580-655, 657-783). v*: the mover's vertices; cx/cy: tile centre tables.
template <typename RT>
__device__ __forceinline__ bool tile_valid_box(const Tables<RT> &T, const RT *tcx, const RT *tcy, int Tx, int Ty, int ix, int iy, RT xlo,
                                               RT xhi, RT ylo, RT yhi, const RT *vx, const RT *vy, RT thx, RT thy,
                                               bool *corner_tested = nullptr) {
    const uint32_t lut = T.lut[ix * Ty + iy];
    uint32_t sum = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        uint32_t bits = (uint32_t)(xlo < vx[r]) | ((uint32_t)(vx[r] < xhi) << 1) | ((uint32_t)(ylo < vy[r]) << 2) |
                        ((uint32_t)(vy[r] < yhi) << 3);
        sum += (lut >> (2u * bits)) & 3u;
    }
    bool valid = sum == 4u;  // BE:655
    if (!valid) return false;
    // BE:657-783: the tile and both its side neighbours towards a diagonal exist but the diagonal tile does not
    const uint32_t n = T.nbr[ix * Ty + iy];
    auto bit = [n](int dx, int dy) { return (n >> ((dx + 1) * 3 + (dy + 1))) & 1u; };
    const int ddx[4] = {+1, +1, -1, -1}, ddy[4] = {-1, +1, -1, +1};  // bl, br, tl, tr in the reference's naming
    for (int k = 0; k < 4; ++k) {
        int dx = ddx[k], dy = ddy[k];
        if (bit(0, 0) & bit(dx, 0) & bit(0, dy) & (bit(dx, dy) ^ 1u)) {
            int mx = ix + dx, my = iy + dy;
            if (mx < 0 || my < 0 || mx >= Tx || my >= Ty) continue;  // reference: mask_idx_* (BE:659,691,723,755)
            if (corner_tested) *corner_tested = true;
            RT bx[4], by[4];
            rect_vertices<RT>(tcx[mx], tcy[my], (RT)1, (RT)0, (RT)0, (RT)0, thx, thy, bx, by);
            if (rects_intersect_v<RT>(vx, vy, bx, by)) valid = false;
        }
    }
    return valid;
}

// qpos_is_valid (BE:459-788), any shape, tables in any address space, no wave-level voting (one thread = one qpos)
template <typename RT>
__device__ __forceinline__ bool pos_valid_any(const Tables<RT> &T, const RT *tcx, const RT *tcy, const KParams &P, bool box, RT px, RT py,
                                              RT qw, RT qx, RT qy, RT qz, RT cs0, RT cs1) {
    const int Tx = P.Tx, Ty = P.Ty;
    const Cst<RT> &C = cst<RT>(P);
    RT vx[4], vy[4];
    if (box) rect_vertices<RT>(px, py, qw, qx, qy, qz, cs0, cs1, vx, vy);  // BE:496-497
    int gx = clampi(Math<RT>::floor_int(px * C.inv_wx), 0, Tx - 1);
    int gy = clampi(Math<RT>::floor_int(py * C.inv_wy), 0, Ty - 1);
    bool any = false, all = true, valid3 = false;
    for (int a = gx - 1; a <= gx + 1; ++a) {
        if (a < 0 || a >= Tx) continue;
        const RT *fa = T.ax + 4 * a;
        bool in_x = (fa[0] <= px) & (px <= fa[1]);
        for (int b = gy - 1; b <= gy + 1; ++b) {
            if (b < 0 || b >= Ty) continue;
            const RT *fb = T.ay + 4 * b;
            if (!(in_x & (fb[0] <= py) & (py <= fb[1]))) continue;  // BE:508-513
            any = true;
            valid3 |= (T.nbr[a * Ty + b] == 0x1FFu);
            bool vk = box ? tile_valid_box<RT>(T, tcx, tcy, Tx, Ty, a, b, fa[0], fa[1], fb[0], fb[1], vx, vy, C.tile_hx, C.tile_hy)
                          : tile_valid_circle<RT>(T, Ty, a, b, fa[0], fa[1], fb[0], fb[1], px, py, cs0);
            all &= vk;
        }
    }
    return valid3 | (any & all);
}

// ---------------------------------------------------------------------------------------------------------------------
// 'box' collision shape inside the fused env kernels
// ---------------------------------------------------------------------------------------------------------------------
template <typename RT> struct Pose {
    RT x, y, qw, qx, qy, qz;  // qpos without z
};

// qpos_is_valid (BE:459-788) for one rotated rectangle. UNC: also report whether moving the centre by up to `band` per axis
// and every vertex by up to `band` could change the verdict (conservative).
template <typename RT, bool UNC>
__device__ __forceinline__ bool pos_valid_box(const Tables<RT> &T, const KParams &P, const Pose<RT> &q, RT c0, RT c1, RT band, bool *uncertain) {
    const int Tx = P.Tx, Ty = P.Ty;
    const Cst<RT> &C = cst<RT>(P);
    RT vx[4], vy[4];
    rect_vertices<RT>(q.x, q.y, q.qw, q.qx, q.qy, q.qz, c0, c1, vx, vy);  // BE:496-497
    const int gx = clampi(Math<RT>::floor_int(q.x * C.inv_wx), 0, Tx - 1), gy = clampi(Math<RT>::floor_int(q.y * C.inv_wy), 0, Ty - 1);
    bool any = false, all = true, valid3 = false, corner = false;
    for (int a = gx - 1; a <= gx + 1; ++a) {
        if (a < 0 || a >= Tx) continue;
        const RT *fa = T.ax + 4 * a;
        const bool in_x = (fa[0] <= q.x) & (q.x <= fa[1]);
        for (int b = gy - 1; b <= gy + 1; ++b) {
            if (b < 0 || b >= Ty) continue;
            const RT *fb = T.ay + 4 * b;
            if (!(in_x & (fb[0] <= q.y) & (q.y <= fb[1]))) continue;  // BE:508-513
            any = true;
            valid3 |= (T.nbr[a * Ty + b] == 0x1FFu);  // BE:530-538
            all &= tile_valid_box<RT>(T, T.cx, T.cy, Tx, Ty, a, b, fa[0], fa[1], fb[0], fb[1], vx, vy, C.tile_hx, C.tile_hy, &corner);
        }
    }
    if constexpr (UNC) {
        // every decision above compares the centre or a vertex with an edge line of the centre's tile (or runs a corner test)
        const RT *fa = T.ax + 4 * gx, *fb = T.ay + 4 * gy;
        bool unc = corner | !((q.x - fa[0] > band) & (fa[1] - q.x > band) & (q.y - fb[0] > band) & (fb[1] - q.y > band));
        for (int r = 0; r < 4; ++r)
            unc |= (absr(vx[r] - fa[0]) <= band) | (absr(vx[r] - fa[1]) <= band) | (absr(vy[r] - fb[0]) <= band) | (absr(vy[r] - fb[1]) <= band);
        *uncertain = unc;
    }
    return valid3 | (any & all);
}

// check_mover_collision, box branch (BE:410-422), seen from one lane: partners at rotation distance 1..LPE/2.
//   near: some partner passes the distance pre-filter dist <= 4*max_size (+ extra, for the lazy-noise decision)
//   hit : the pre-filter (without extra) passes and the two rectangles intersect
template <typename RT, int LPE, int D> struct BoxSweep {
    static __device__ __forceinline__ void run(const RT *vx, const RT *vy, RT x, RT y, RT cmax, RT extra, bool with_rects, bool &hit, bool &near) {
        if constexpr (D <= LPE / 2 && D >= 1 && LPE > 1) {
            const RT dx = x - rot<LPE, D>(x), dy = y - rot<LPE, D>(y);
            const RT dist = Math<RT>::sqrt_(dx * dx + dy * dy);                 // BE:411
            const RT ms = fmax(cmax, rot<LPE, D>(cmax));                        // BE:412
            const RT lim = (RT)2 * (absr(ms) + absr(ms));                       // BE:413-414 (ord = 1 norm of (ms, ms))
            near |= dist <= lim + extra;
            const bool pre = dist <= lim;
            if (with_rects && (__ballot(pre) != 0ull)) {                        // rectangles only when some lane needs them
                // edges of two rectangles can only cross (GEO:107-138, 1e-7 tolerances) where their bounding boxes overlap: the
                // 16 segment-pair tests run only if some lane's pair gets that far
                const RT lox = fmin(fmin(vx[0], vx[1]), fmin(vx[2], vx[3])), hix = fmax(fmax(vx[0], vx[1]), fmax(vx[2], vx[3]));
                const RT loy = fmin(fmin(vy[0], vy[1]), fmin(vy[2], vy[3])), hiy = fmax(fmax(vy[0], vy[1]), fmax(vy[2], vy[3]));
                const RT slack = (RT)1e-6;
                const bool boxes = pre & (lox <= rot<LPE, D>(hix) + slack) & (rot<LPE, D>(lox) <= hix + slack) &
                                   (loy <= rot<LPE, D>(hiy) + slack) & (rot<LPE, D>(loy) <= hiy + slack);
                if (__ballot(boxes) != 0ull) {
                    RT ox[4], oy[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { ox[r] = rot<LPE, D>(vx[r]); oy[r] = rot<LPE, D>(vy[r]); }
                    hit |= boxes & rects_intersect_v<RT>(vx, vy, ox, oy);      // BE:417-422
                }
            }
            BoxSweep<RT, LPE, D + 1>::run(vx, vy, x, y, cmax, extra, with_rects, hit, near);
        }
    }
};
// box shape, noise-free (axis-aligned) rectangles: is some partner's bounding box, both inflated, within reach? |dx| <= sx and
// |dy| <= sy with sx, sy = summed half sizes + what two noisy reads can add. Conservative for any noisy read of both poses.
template <typename RT, int LPE, int D> struct BoxNear {
    static __device__ __forceinline__ bool run(RT x, RT y, RT sx, RT sy) {
        if constexpr (LPE == 1) {
            return false;
        } else {
            RT dx, dy;
            rot_diff2<LPE, D>(x, y, dx, dy);
            const bool n = (absr(dx) <= sx) & (absr(dy) <= sy);
            if constexpr (D < LPE / 2) return n | BoxNear<RT, LPE, D + 1>::run(x, y, sx, sy);
            else return n;
        }
    }
};
// the same as ONE number per lane: the smallest over the partners of max(|dx| - sx, |dy| - sy); <= 0 iff BoxNear holds
template <typename RT, int LPE, int D> struct BoxNearMetric {
    static __device__ __forceinline__ RT run(RT x, RT y, RT sx, RT sy) {
        if constexpr (LPE == 1) {
            return Math<RT>::inf();
        } else {
            RT dx, dy;
            rot_diff2<LPE, D>(x, y, dx, dy);
            const RT m = fmax(absr(dx) - sx, absr(dy) - sy);
            if constexpr (D < LPE / 2) return fmin(m, BoxNearMetric<RT, LPE, D + 1>::run(x, y, sx, sy));
            else return m;
        }
    }
};
// BE:392-422 for AXIS-ALIGNED rectangles (identity quaternions: every reset candidate), without vertices or segment tests: the edges
// of two such rectangles cross iff their bounding boxes -- themselves -- overlap and neither lies inside the other, and the
// distance pre-filter BE:411-414 (4 x the larger half size) always holds for overlapping ones. `sure`: some partner surely
// intersects; `unsure`: some partner is within 2e-6 of touching or of lying inside (GEO:107-138 compares with 1e-7 tolerances) --
// then the caller runs the general test. s0, s1: this lane's half sizes (with whatever offset the check uses). With eps = what two
// noisy reads can add to / take from the gap, `sure` and "neither" hold for every possible pair of reads (the step loop's lazy noise).
template <typename RT, int LPE, int D> struct AabbSweep {
    static __device__ __forceinline__ void run(RT x, RT y, RT s0, RT s1, bool &sure, bool &unsure, RT eps = (RT)2e-6) {
        if constexpr (D <= LPE / 2 && D >= 1 && LPE > 1) {
            const RT adx = absr(x - rot<LPE, D>(x)), ady = absr(y - rot<LPE, D>(y));
            const RT o0 = rot<LPE, D>(s0), o1 = rot<LPE, D>(s1);
            const RT gap = fmax(adx - (s0 + o0), ady - (s1 + o1));  // < 0: the rectangles overlap
            const bool inside = (adx + fmin(s0, o0) < fmax(s0, o0) + eps) & (ady + fmin(s1, o1) < fmax(s1, o1) + eps);
            sure |= (gap < -eps) & !inside;
            unsure |= (absr(gap) <= eps) | ((gap < -eps) & inside);
            AabbSweep<RT, LPE, D + 1>::run(x, y, s0, s1, sure, unsure, eps);
        }
    }
};
template <typename RT, int LPE>
__device__ __forceinline__ bool box_pair_hit(const Pose<RT> &q, RT c0, RT c1, RT extra, bool with_rects, bool *near_out) {
    RT vx[4] = {0, 0, 0, 0}, vy[4] = {0, 0, 0, 0};
    if (with_rects) rect_vertices<RT>(q.x, q.y, q.qw, q.qx, q.qy, q.qz, c0, c1, vx, vy);
    bool hit = false, near = false;
    BoxSweep<RT, LPE, 1>::run(vx, vy, q.x, q.y, fmax(c0, c1), extra, with_rects, hit, near);
    if (near_out) *near_out = near;
    return hit;
}

// ---------------------------------------------------------------------------------------------------------------------
// pair sweep (BE:392-409): every unordered pair {i,j} is seen by lane i at rotation distance d = (j-i) mod LPE <= LPE/2.
// Returns the smallest pair distance this lane saw (squared in throughput mode), +inf if none. Padding lanes of a group
// (mover index >= M) must hold far_away() coordinates.
// NumPy broadcasts BE:409 to any_{a,b}(norm[b] <= csum[a]) == (min norm <= max csum); identical to the per-pair test for
// equal radii, and reproduced as is for unequal ones.
// ---------------------------------------------------------------------------------------------------------------------
template <typename RT, int LPE, int D> struct PairSweep {
    // smallest of the pair distances at rotation D .. LPE/2
    static __device__ __forceinline__ RT run(RT x, RT y) {
        if constexpr (Math<RT>::literal) {
            const RT dx = x - rot<LPE, D>(x), dy = y - rot<LPE, D>(y);
            const RT m = Math<RT>::sqrt_(dx * dx + dy * dy);
            if constexpr (D < LPE / 2) {
                const RT rest = PairSweep<RT, LPE, D + 1>::run(x, y);
                return rest < m ? rest : m;
            } else {
                return m;
            }
        } else if constexpr (LPE == 2) {
            RT dx, dy;
            rot_diff2<LPE, D>(x, y, dx, dy);
            return __builtin_fmaf(dx, dx, dy * dy);
        } else {
            // two rotations at a time: the four differences come out of DPP subtracts, the two squared distances out of one
            // packed multiply + one packed fma
            static_assert(D % 2 == 1 && D + 1 <= LPE / 2, "rotations are taken in pairs");
            typedef RT V2 __attribute__((ext_vector_type(2)));
            RT dx0, dy0, dx1, dy1;
            rot_diff4<LPE, D>(x, y, dx0, dy0, dx1, dy1);
            if constexpr (LPE >= 8) {
                // wide groups: plain multiplies and fma's (gfx950 wants a wait state around every dependent packed instruction -- the
                // assembler pads them with s_nop 0, and a padded packed pair is no cheaper than two plain instructions), three-way minima
                const RT m0 = __builtin_fmaf(dx0, dx0, dy0 * dy0), m1 = __builtin_fmaf(dx1, dx1, dy1 * dy1);
                if constexpr (D + 1 < LPE / 2) return __builtin_fminf(__builtin_fminf(PairSweep<RT, LPE, D + 2>::run(x, y), m0), m1);  // (v_min3_f32)
                else return __builtin_fminf(m0, m1);
            } else {
                const V2 dx{dx0, dx1}, dy{dy0, dy1};
                const V2 m = __builtin_elementwise_fma(dx, dx, dy * dy);
                const RT mm = __builtin_fminf(m.x, m.y);
                if constexpr (D + 1 < LPE / 2) return __builtin_fminf(PairSweep<RT, LPE, D + 2>::run(x, y), mm);
                else return mm;
            }
        }
    }
};
// Lanes that pad an env group (mover index >= M) are parked at FAR_AWAY so that no live mover ever sees them as a partner.
template <typename RT> __device__ __forceinline__ RT far_away() { return (RT)1e18f; }  // (finite as a float too: the parity loop's float pre-tests)
template <typename RT, int LPE> __device__ __forceinline__ RT pair_min_dist(RT x, RT y) {
    if constexpr (LPE == 1) {
        return Math<RT>::inf();
    } else {
        return PairSweep<RT, LPE, 1>::run(x, y);
    }
}
// smallest SQUARED pair distance this lane saw, in any precision and without a square root: the prefilter of the parity-mode
// loops (a pair further apart than the limit plus the noise band cannot collide; only then is the literal sqrt form evaluated)
template <typename RT, int LPE, int D> struct PairSweepSq {
    static __device__ __forceinline__ RT run(RT x, RT y) {
        const RT dx = x - rot<LPE, D>(x), dy = y - rot<LPE, D>(y);
        const RT m = dx * dx + dy * dy;
        if constexpr (D < LPE / 2) {
            const RT rest = PairSweepSq<RT, LPE, D + 1>::run(x, y);
            return rest < m ? rest : m;
        } else {
            return m;
        }
    }
};
template <typename RT, int LPE> __device__ __forceinline__ RT pair_min_d2(RT x, RT y) {
    if constexpr (LPE == 1) return Math<RT>::inf();
    else return PairSweepSq<RT, LPE, 1>::run(x, y);
}
// compare a pair_min_dist result against a distance limit: <= (collision, BE:409) or < (goal spacing, PL:409)
template <typename RT> __device__ __forceinline__ bool dist_le(RT m, RT lim) { return Math<RT>::literal ? (m <= lim) : (m <= lim * lim); }
template <typename RT> __device__ __forceinline__ bool dist_lt(RT m, RT lim) { return Math<RT>::literal ? (m < lim) : (m < lim * lim); }
// Parity mode, "sqrt(d2) <= lim" (STRICT: "< lim") as the reference forms it, without the square root wherever it cannot
// matter: a correctly rounded square root is monotone and exact to half an ulp, so d2 outside lim^2 (1 -+ 1e-12) decides the
// compare; only a wavefront with a lane inside that sliver takes the root (a double sqrt is ~19 instructions, a division ~12:
// scratch/ubench/f64.hip). Wave-cooperative (it votes): call convergently.
template <typename RT, bool STRICT> __device__ __forceinline__ bool sqrt_within(RT d2, RT lim) {
    if constexpr (!Math<RT>::literal) {
        return STRICT ? d2 < lim * lim : d2 <= lim * lim;
    } else {
        const RT l2 = lim * lim;
        const bool in = d2 < l2 * (RT)(1.0 - 1e-12), out = d2 > l2 * (RT)(1.0 + 1e-12);
        if (__ballot(!(in | out)) == 0ull) return in;
        const RT d = Math<RT>::sqrt_(d2);
        return STRICT ? d < lim : d <= lim;
    }
}
// is any partner of this lane's env group closer than lim (BE:409 inclusive; STRICT: PL:409)? pair_min_dist + dist_le / dist_lt
// with the parity-mode roots taken only in the sliver above (the minimum of the roots is the root of the minimum).
template <typename RT, int LPE, bool STRICT> __device__ __forceinline__ bool pair_closer(RT x, RT y, RT lim) {
    if constexpr (!Math<RT>::literal) {
        const RT m = pair_min_dist<RT, LPE>(x, y);
        return STRICT ? dist_lt<RT>(m, lim) : dist_le<RT>(m, lim);
    } else {
        return sqrt_within<RT, STRICT>(pair_min_d2<RT, LPE>(x, y), lim);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// ensure_max_dyn_val (PL:610-645) for one mover: returns the derivative to apply; `next` receives the next value
// ---------------------------------------------------------------------------------------------------------------------
// a / b in parity mode, given rb = the correctly rounded 1 / b: product + one residual correction -- the sequence the compiler
// emits for an IEEE division minus its reciprocal refinement and range scaling (12 instructions -> 3; the operands here are O(1)),
// so quotients over a shared divisor (the clamp's norm; dt, whose reciprocal comes from the host) cost one reciprocal together.
template <typename RT> __device__ __forceinline__ RT div_rcp(RT a, RT b, RT rb) {
    if constexpr (Math<RT>::literal) { const RT q = a * rb; return fma(fma(-q, b, a), rb, q); }
    else return a * rb;
}
// (PLAIN: the stand-alone operator gpr_ensure_max_dyn_val takes arbitrary caller values -- it keeps the compiler's divisions)
template <typename RT, bool PLAIN = false> __device__ __forceinline__ R2<RT> clamp_dyn(R2<RT> cur, RT max_value, R2<RT> deriv, RT dt, RT inv_dt, R2<RT> &next) {
    RT t0 = dt * deriv.x + cur.x, t1 = dt * deriv.y + cur.y;  // PL:631
    RT n2 = t0 * t0 + t1 * t1;
    if constexpr (Math<RT>::literal) {
        RT norm = sqrt(n2);  // PL:633
        if (norm >= max_value) {  // PL:634, inclusive
            if constexpr (PLAIN) {
                next.x = max_value * (t0 / norm);
                next.y = max_value * (t1 / norm);
                return R2<RT>{(next.x - cur.x) / dt, (next.y - cur.y) / dt};
            }
            const RT rn = (RT)1 / norm;  // (four quotients over two divisors: div_rcp)
            next.x = max_value * div_rcp<RT>(t0, norm, rn);  // PL:640-642
            next.y = max_value * div_rcp<RT>(t1, norm, rn);
            return R2<RT>{div_rcp<RT>(next.x - cur.x, dt, inv_dt), div_rcp<RT>(next.y - cur.y, dt, inv_dt)};  // PL:643
        }
        next.x = t0; next.y = t1;
        return deriv;
    } else {
        bool cl = n2 >= max_value * max_value;
        RT s = max_value * Math<RT>::rsqrt_(n2);
        RT nx = cl ? t0 * s : t0, ny = cl ? t1 * s : t1;
        next.x = nx; next.y = ny;
        return R2<RT>{cl ? (nx - cur.x) * inv_dt : deriv.x, cl ? (ny - cur.y) * inv_dt : deriv.y};
    }
}

}  // namespace gpr
