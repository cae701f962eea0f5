// gpr_ops_kernels.hip -- the reference's stand-alone operators as f64 device kernels (one thread per item):
// geometry_2D_utils (GEO:9-138), ensure_max_dyn_val (PL:610-645), qpos_is_valid (BE:459-788),
// check_mover_collision (BE:355-424), batched compute_reward / compute_terminated (PL:459-534).
// Compiled with -ffp-contract=off: every a*b+c stays two roundings, like NumPy.
#include "gpr_device.h"
#include "gpr_host.h"

namespace gpr {

static inline unsigned grid_for(int64_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }

__global__ void __launch_bounds__(BLOCK) k_segments(int64_t n, const double *p1, const double *p2, const double *q1, const double *q2,
                                                    uint8_t *out) {
    int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    out[i] = (uint8_t)seg_intersect<double>(p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1], q1[2 * i], q1[2 * i + 1], q2[2 * i],
                                            q2[2 * i + 1]);
}

__global__ void __launch_bounds__(BLOCK) k_rect_vertices(int64_t n, const double *qpos, const double *size, double *out) {
    int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const double *q = qpos + 7 * i;
    double vx[4], vy[4];
    rect_vertices<double>(q[0], q[1], q[3], q[4], q[5], q[6], size[2 * i], size[2 * i + 1], vx, vy);
    for (int j = 0; j < 4; ++j) { out[8 * i + j] = vx[j]; out[8 * i + 4 + j] = vy[j]; }
}

__global__ void __launch_bounds__(BLOCK) k_rects(int64_t n, const double *r1, const double *r2, const double *s1, const double *s2,
                                                 uint8_t *out) {
    int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const double *a = r1 + 7 * i, *b = r2 + 7 * i;
    double ax[4], ay[4], bx[4], by[4];
    rect_vertices<double>(a[0], a[1], a[3], a[4], a[5], a[6], s1[2 * i], s1[2 * i + 1], ax, ay);
    rect_vertices<double>(b[0], b[1], b[3], b[4], b[5], b[6], s2[2 * i], s2[2 * i + 1], bx, by);
    out[i] = (uint8_t)rects_intersect_v<double>(ax, ay, bx, by);
}

__global__ void __launch_bounds__(BLOCK) k_clamp(int64_t n, const double *cur, const double *maxv, const double *der, double dt, double *vals,
                                                 double *ders) {
    int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    R2<double> nx;
    R2<double> d = clamp_dyn<double, true>(R2<double>{cur[2 * i], cur[2 * i + 1]}, maxv[i], R2<double>{der[2 * i], der[2 * i + 1]}, dt, 1.0 / dt, nx);
    vals[2 * i] = nx.x; vals[2 * i + 1] = nx.y;
    ders[2 * i] = d.x; ders[2 * i + 1] = d.y;
}

// qpos_is_valid: tables are read from global memory (L2-resident, a few KB)
__global__ void __launch_bounds__(BLOCK) k_qpos_valid(const KParams P, int64_t n, const double *qpos, const double *c_size, int add_safety,
                                                      int32_t *out) {
    int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const Tables<double> T = make_tables<double>(P.tables, P);
    const Cst<double> &C = P.cd;
    const double *q = qpos + 7 * i;
    // BE:487: c_size + offset_wall + int(add_safety_offset) * offset, left to right
    double cs0 = (c_size[2 * i] + C.c_offset_wall) + (double)(add_safety != 0) * C.c_offset;
    double cs1 = (c_size[2 * i + 1] + C.c_offset_wall) + (double)(add_safety != 0) * C.c_offset;
    out[i] = (int32_t)pos_valid_any<double>(T, T.cx, T.cy, P, P.c_shape != 0, q[0], q[1], q[3], q[4], q[5], q[6], cs0, cs1);
}

// check_mover_collision: one thread per set of M movers
__global__ void __launch_bounds__(BLOCK) k_mover_collision(const KParams P, int64_t batch, const double *qpos, const double *c_size,
                                                           int add_safety, uint8_t *out) {
    int64_t b = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (b >= batch) return;
    const int M = P.M;
    const double off = P.cd.c_offset * (double)(add_safety != 0);  // BE:390
    const double *Q = qpos + 7 * (int64_t)M * b;
    bool coll = false;
    if (P.c_shape == 0) {
        // BE:409 broadcasts (K,) <= (K,1) to (K,K): any_{a,b} norm[b] <= csum[a]  ==  min norm <= max csum
        double min_norm = __builtin_huge_val(), max_csum = -__builtin_huge_val();
        for (int i = 0; i < M - 1; ++i)
            for (int j = i + 1; j < M; ++j) {
                double dx = Q[7 * i] - Q[7 * j], dy = Q[7 * i + 1] - Q[7 * j + 1];
                double norm = sqrt(dx * dx + dy * dy);
                double csum = (c_size[2 * i] + off) + (c_size[2 * j] + off);
                min_norm = norm < min_norm ? norm : min_norm;
                max_csum = csum > max_csum ? csum : max_csum;
            }
        coll = min_norm <= max_csum;
    } else {
        for (int i = 0; i < M - 1; ++i)
            for (int j = i + 1; j < M; ++j) {
                double ci0 = c_size[2 * i] + off, ci1 = c_size[2 * i + 1] + off, cj0 = c_size[2 * j] + off, cj1 = c_size[2 * j + 1] + off;
                double dx = Q[7 * i] - Q[7 * j], dy = Q[7 * i + 1] - Q[7 * j + 1];
                double dist = sqrt(dx * dx + dy * dy);                     // BE:411
                double ms = fmax(fmax(ci0, ci1), fmax(cj0, cj1));           // BE:412
                if (dist <= 2 * (fabs(ms) + fabs(ms))) {                    // BE:413-414 (ord=1 norm of (ms, ms))
                    const double *a = Q + 7 * i, *c = Q + 7 * j;
                    double ax[4], ay[4], bx[4], by[4];
                    rect_vertices<double>(a[0], a[1], a[3], a[4], a[5], a[6], ci0, ci1, ax, ay);
                    rect_vertices<double>(c[0], c[1], c[3], c[4], c[5], c[6], cj0, cj1, bx, by);
                    coll |= rects_intersect_v<double>(ax, ay, bx, by);     // BE:417-422
                }
            }
    }
    out[b] = (uint8_t)coll;
}

// compute_reward / compute_terminated on a HER batch (PL:459-534), one thread per transition
template <typename ST> __global__ void __launch_bounds__(BLOCK) k_reward(const KParams P, int64_t batch, const ST *ag, const ST *dg,
                                                                         const uint8_t *mcs, const uint8_t *wcs, ST *reward,
                                                                         uint8_t *terminated) {
    int64_t b = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (b >= batch) return;
    const int M = P.M;
    const double thr = P.cd.thr, rs = P.cd.reward_success;
    int n_reached = 0;
    for (int i = 0; i < M; ++i) {
        double dx = (double)ag[2 * M * b + 2 * i] - (double)dg[2 * M * b + 2 * i];
        double dy = (double)ag[2 * M * b + 2 * i + 1] - (double)dg[2 * M * b + 2 * i + 1];
        n_reached += sqrt(dx * dx + dy * dy) <= thr;  // PL:522-525
    }
    bool coll = (mcs[b] != 0) | (wcs[b] != 0);
    double r = -rs * (double)coll;
    r += -1.0 * (double)(M - n_reached) * (double)(!coll);
    if (n_reached == M && !coll) r = rs;
    reward[b] = (ST)r;
    if (terminated) terminated[b] = (uint8_t)((r == rs) | (r == -rs));
}

// per-agent views (include/gpr.h: gpr_agent_views): one lane per (env, mover); the other movers' columns arrive by lane
// permutes inside the env group -- the "per-agent obs gather" is intra-wavefront, no memory round trip
template <typename ST, int LPE, int D> struct AgentGather {
    static __device__ __forceinline__ void run(ST px, ST py, ST vx, ST vy, int mv, int M, ST *row) {
        if constexpr (D < LPE) {
            const ST ox = rot<LPE, D>(px), oy = rot<LPE, D>(py), ovx = rot<LPE, D>(vx), ovy = rot<LPE, D>(vy);
            const int j = (mv + D) & (LPE - 1);
            if (j < M && mv < M) {
                ST *dst = row + 6 + 4 * (j < mv ? j : j - 1);  // ascending j, skipping the agent itself
                dst[0] = ox - px; dst[1] = oy - py; dst[2] = ovx - vx; dst[3] = ovy - vy;
            }
            AgentGather<ST, LPE, D + 1>::run(px, py, vx, vy, mv, M, row);
        }
    }
};
// The rows of a wavefront's agents ((env, mover) in lane order, dead lanes skipped) are consecutive in agent_obs, each 4M+2 values
// long (264 B for 16 movers): written by their own lanes they are 64 scattered 16-byte pieces per store instruction (measured: 38 %
// more bytes written than the rows hold, partial sectors). So every wavefront assembles its rows in its own slice of LDS and then
// streams the whole span out, consecutive lanes writing consecutive 16-byte (or 4-byte) pieces.
constexpr int AV_BLOCK = 128;  // 2 wavefronts x 64 rows x (4 * 16 + 2) values x 4 B = 33 KB of LDS in the widest case (f64: 66 KB, raised)
template <typename ST, int LPE>
__global__ void __launch_bounds__(AV_BLOCK) k_agent_views(const KParams P, int obs_dim, ST *agent_obs, ST *agent_reward) {
    extern __shared__ __align__(16) unsigned char av_lds[];
    const int M = P.M, W = 4 * M + 2;
    const ST *obs = static_cast<const ST *>(P.obs), *ag = static_cast<const ST *>(P.ag), *dg = static_cast<const ST *>(P.dg);
    const ST *reward = static_cast<const ST *>(P.reward);
    const int64_t t = (int64_t)blockIdx.x * AV_BLOCK + threadIdx.x;
    const int lane = (int)(threadIdx.x & 63);
    const int mv = (int)(t & (LPE - 1));
    const int64_t env = t / LPE;
    const bool live = (env < P.E) & (mv < M);
    const int64_t envc = env < P.E ? env : P.E - 1;
    const int mvc = mv < M ? mv : M - 1;
    const ST vx = obs[envc * obs_dim + 2 * mvc], vy = obs[envc * obs_dim + 2 * mvc + 1];
    const ST px = ag[envc * 2 * M + 2 * mvc], py = ag[envc * 2 * M + 2 * mvc + 1];
    const ST gx = dg[envc * 2 * M + 2 * mvc], gy = dg[envc * 2 * M + 2 * mvc + 1];
    // this wavefront's slice of LDS: row r = (env index within the wavefront) * M + mover
    ST *slice = reinterpret_cast<ST *>(av_lds) + (size_t)(threadIdx.x >> 6) * (64 / LPE) * M * W;
    const int64_t env_w0 = (t - lane) / LPE;                      // first env of the wavefront
    ST *row = slice + ((int)(env - env_w0) * M + mvc) * W;
    if (live) { row[0] = vx; row[1] = vy; row[2] = px; row[3] = py; row[4] = gx; row[5] = gy; }
    AgentGather<ST, LPE, 1>::run(px, py, vx, vy, live ? mv : LPE, M, row);  // dead lanes write nothing (mv = LPE fails mv < M)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wavefront's LDS writes have landed
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // stream the span out
    int64_t n_env = P.E - env_w0;
    n_env = n_env < 0 ? 0 : (n_env > 64 / LPE ? 64 / LPE : n_env);
    const int total = (int)n_env * M * W;                          // values in this wavefront's span
    ST *dst = agent_obs + env_w0 * (int64_t)M * W;
    constexpr int V = 16 / (int)sizeof(ST);                        // values per 16-byte piece
    if ((((env_w0 * (int64_t)M * W) % V) == 0) & (sizeof(ST) == 4 || sizeof(ST) == 8)) {
        typedef ST VecT __attribute__((ext_vector_type(V)));
        const int nv = total / V;
        for (int i = lane; i < nv; i += 64) reinterpret_cast<VecT *>(dst)[i] = reinterpret_cast<const VecT *>(slice)[i];
        for (int i = nv * V + lane; i < total; i += 64) dst[i] = slice[i];
    } else {
        for (int i = lane; i < total; i += 64) dst[i] = slice[i];
    }
    if (live) {
        // The env reward (PL:519-531) was computed from the state the step ended in. With same-step autoreset an env that ended
        // in this call already shows the new episode in obs / ag / dg: its terminal achieved / desired goal are in final_*.
        ST rpx = px, rpy = py, rgx = gx, rgy = gy;
        const bool ended = (P.autoreset == 1) & ((P.terminated[envc] | P.truncated[envc]) != 0);
        if (ended & (P.final_ag != nullptr)) {
            const ST *fa = static_cast<const ST *>(P.final_ag), *fd = static_cast<const ST *>(P.final_dg);
            rpx = fa[envc * 2 * M + 2 * mvc]; rpy = fa[envc * 2 * M + 2 * mvc + 1];
            rgx = fd[envc * 2 * M + 2 * mvc]; rgy = fd[envc * 2 * M + 2 * mvc + 1];
        }
        const double r = (double)reward[envc], rs = P.cd.reward_success, dx = (double)rpx - (double)rgx, dy = (double)rpy - (double)rgy;
        const bool shared = (r == rs) | (r == -rs);  // PL:529-531
        // next-step autoreset: a call that only reset the env (its step count is 0 afterwards) carries reward 0, for every agent too
        const bool reset_only = (P.autoreset == 2) & (P.elapsed[envc] == 0);
        agent_reward[envc * M + mvc] = reset_only ? (ST)0 : shared ? (ST)r : (ST)(sqrt(dx * dx + dy * dy) <= P.cd.thr ? 0.0 : -1.0);
    }
}
template <typename ST> hipError_t launch_agent_views(const KParams &P, int lpe, int obs_dim, void *agent_obs, void *agent_reward, hipStream_t s) {
    const unsigned grid = (unsigned)((P.E * lpe + AV_BLOCK - 1) / AV_BLOCK);
    const unsigned lds = (unsigned)((AV_BLOCK / 64) * (64 / lpe) * P.M * (4 * P.M + 2) * sizeof(ST));
    hipError_t err = hipSuccess;
    static LdsFlag raised[17][GPR_MAX_DEVICES];  // per (kernel = lanes per env, device); one array per ST (this function is a template)
    auto a = [&](auto K) {
        if (lds > 48u * 1024u) err = raise_dynamic_lds(reinterpret_cast<const void *>(K), 80 * 1024, raised[lpe & 16 ? 16 : lpe]);
        if (err == hipSuccess)
            hipLaunchKernelGGL(K, dim3(grid), dim3(AV_BLOCK), lds, s, P, obs_dim, static_cast<ST *>(agent_obs), static_cast<ST *>(agent_reward));
    };
    switch (lpe) {
        case 1: a(k_agent_views<ST, 1>); break;
        case 2: a(k_agent_views<ST, 2>); break;
        case 4: a(k_agent_views<ST, 4>); break;
        case 8: a(k_agent_views<ST, 8>); break;
        case 16: a(k_agent_views<ST, 16>); break;
        default: return hipErrorInvalidValue;
    }
    return err != hipSuccess ? err : hipGetLastError();
}
template hipError_t launch_agent_views<float>(const KParams &, int, int, void *, void *, hipStream_t);
template hipError_t launch_agent_views<double>(const KParams &, int, int, void *, void *, hipStream_t);

// episode counters: one block sums the per-wavefront rows (a few thousand) column by column, adds to the caller's totals,
// clears the rows. Runs on the stream of the step kernels, so it sees complete rows. The two diagnostic words (KParams::diag: reset
// searches that ran out of trials, clipped env-steps) are MOVED into totals[6], totals[7] the same way; diag[2], diag[3] keep what was
// moved so far, so that gpr_read_diagnostics can report totals since gpr_create whoever read in between.
__global__ void __launch_bounds__(BLOCK) k_counter_reduce(uint32_t *rows, int64_t n_rows, unsigned long long *totals, uint32_t *diag) {
    __shared__ unsigned long long part[BLOCK / 64][6];
    unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t r = threadIdx.x; r < n_rows; r += BLOCK) {
#pragma unroll
        for (int k = 0; k < 6; ++k) { acc[k] += rows[r * 6 + k]; rows[r * 6 + k] = 0u; }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        unsigned long long v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        unsigned long long v = 0;
        for (int w = 0; w < BLOCK / 64; ++w) v += part[w][threadIdx.x];
        totals[threadIdx.x] += v;
    } else if (threadIdx.x < 8) {
        const int k = threadIdx.x - 6;
        const uint32_t v = __hip_atomic_exchange(diag + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        diag[2 + k] += v;
        totals[threadIdx.x] += v;
    }
}
hipError_t launch_counter_reduce(uint32_t *rows, int64_t n_rows, uint64_t *totals, uint32_t *diag, hipStream_t s) {
    hipLaunchKernelGGL(k_counter_reduce, dim3(1), dim3(BLOCK), 0, s, rows, n_rows, reinterpret_cast<unsigned long long *>(totals), diag);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// get_mover_qpos / get_mover_qvel / get_mover_qacc and their _arr forms (BE:801-855, 1244-1290) for every env of the handle: one thread
// per (env, selected mover) row. which 0: qpos (x, y, z, qw, qx, qy, qz) -- z is the distance between the mover's bottom and the tile's
// top (BE:821-824: the body's z minus the mover's half height), which the planar engine keeps at `z_report`; the quaternion is the
// identity for the planning env and the yaw rotation (cos(yaw/2), 0, 0, sin(yaw/2)) when the handle carries the pushing env's
// mover_rot. which 1: qvel (vx, vy, 0, 0, 0, yaw rate); which 2: qacc (ax, ay, 0, 0, 0, 0). add_noise: Gaussian noise of
// std_noise[which] on EVERY component (BE:828, 841, 855), from the engine's counter RNG: key (seed, tick, KIND_GETTER_NOISE, env, mover),
// components (2k, 2k+1) take normal pair k.
// ---------------------------------------------------------------------------------------------------------------------
template <typename ST, typename RT>
__global__ void __launch_bounds__(BLOCK) k_mover_state(KParams P, int which, int n_sel, const int32_t *sel, RT z_report, int add_noise,
                                                       const ST *mover_rot, ST *out) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= P.E * n_sel) return;
    const int64_t e = i / n_sel;
    const int m = clampi(sel[i - e * n_sel], 0, P.M - 1);  // (an index outside the env's movers must not read outside the state tensors)
    const Cst<RT> &C = cst<RT>(P);
    const ST *src = static_cast<const ST *>(which == 0 ? P.pos : which == 1 ? P.vel : P.acc) + (e * P.M + m) * 2;
    RT v[8] = {(RT)src[0], (RT)src[1], (RT)0, (RT)0, (RT)0, (RT)0, (RT)0, (RT)0};
    const int nc = which == 0 ? 7 : 6;
    if (which == 0) {
        v[2] = z_report; v[3] = (RT)1;
        if (mover_rot != nullptr) {
            RT sn, cs;
            Math<RT>::sincos_((RT)0.5 * (RT)mover_rot[(e * P.M + m) * 2], sn, cs);
            v[3] = cs; v[6] = sn;
        }
    } else if (which == 1 && mover_rot != nullptr) {
        v[5] = (RT)mover_rot[(e * P.M + m) * 2 + 1];
    }
    if (add_noise) {
        const RT sigma = which == 0 ? C.sig_p : which == 1 ? C.sig_v : C.sig_a;
        const uint32_t key = rng_key(P, KIND_GETTER_NOISE, P.env_id_offset + e, (uint32_t)m);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const R2<RT> n = rng_normal_pair<RT>(key, (uint32_t)k);
            v[2 * k] = v[2 * k] + sigma * n.x;
            v[2 * k + 1] = v[2 * k + 1] + sigma * n.y;
        }
    }
    ST *o = out + i * nc;
    for (int k = 0; k < nc; ++k) o[k] = (ST)v[k];
}
template <typename ST, typename RT>
hipError_t launch_mover_state(const KParams &P, int which, int n_sel, const int32_t *sel, double z_report, int add_noise, const void *mover_rot,
                              void *out, hipStream_t s) {
    const int64_t n = P.E * n_sel;
    if (n > 0)
        hipLaunchKernelGGL((k_mover_state<ST, RT>), dim3(grid_for(n)), dim3(BLOCK), 0, s, P, which, n_sel, sel, (RT)z_report, add_noise,
                           static_cast<const ST *>(mover_rot), static_cast<ST *>(out));
    return hipGetLastError();
}
template hipError_t launch_mover_state<float, float>(const KParams &, int, int, const int32_t *, double, int, const void *, void *, hipStream_t);
template hipError_t launch_mover_state<float, double>(const KParams &, int, int, const int32_t *, double, int, const void *, void *, hipStream_t);
template hipError_t launch_mover_state<double, double>(const KParams &, int, int, const int32_t *, double, int, const void *, void *, hipStream_t);
hipError_t set_normal_table_ops(const double *tab) { return hipMemcpyToSymbol(HIP_SYMBOL(g_normal_tab), &tab, sizeof(tab)); }

hipError_t launch_segments(int64_t n, const double *p1, const double *p2, const double *q1, const double *q2, uint8_t *out, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_segments, dim3(grid_for(n)), dim3(BLOCK), 0, s, n, p1, p2, q1, q2, out);
    return hipGetLastError();
}
hipError_t launch_rect_vertices(int64_t n, const double *qpos, const double *size, double *out, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_rect_vertices, dim3(grid_for(n)), dim3(BLOCK), 0, s, n, qpos, size, out);
    return hipGetLastError();
}
hipError_t launch_rects(int64_t n, const double *r1, const double *r2, const double *s1, const double *s2, uint8_t *out, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_rects, dim3(grid_for(n)), dim3(BLOCK), 0, s, n, r1, r2, s1, s2, out);
    return hipGetLastError();
}
hipError_t launch_clamp(int64_t n, const double *cur, const double *maxv, const double *der, double dt, double *vals, double *ders,
                        hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_clamp, dim3(grid_for(n)), dim3(BLOCK), 0, s, n, cur, maxv, der, dt, vals, ders);
    return hipGetLastError();
}
hipError_t launch_qpos_valid(const KParams &P, int64_t n, const double *qpos, const double *c_size, int add_safety, int32_t *out,
                             hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_qpos_valid, dim3(grid_for(n)), dim3(BLOCK), 0, s, P, n, qpos, c_size, add_safety, out);
    return hipGetLastError();
}
hipError_t launch_mover_collision(const KParams &P, int64_t batch, const double *qpos, const double *c_size, int add_safety, uint8_t *out,
                                  hipStream_t s) {
    if (batch > 0) hipLaunchKernelGGL(k_mover_collision, dim3(grid_for(batch)), dim3(BLOCK), 0, s, P, batch, qpos, c_size, add_safety, out);
    return hipGetLastError();
}
template <typename ST> hipError_t launch_reward(const KParams &P, int64_t batch, const void *ag, const void *dg, const uint8_t *mc,
                                                const uint8_t *wc, void *reward, uint8_t *terminated, hipStream_t s) {
    if (batch > 0)
        hipLaunchKernelGGL((k_reward<ST>), dim3(grid_for(batch)), dim3(BLOCK), 0, s, P, batch, static_cast<const ST *>(ag),
                           static_cast<const ST *>(dg), mc, wc, static_cast<ST *>(reward), terminated);
    return hipGetLastError();
}
template hipError_t launch_reward<float>(const KParams &, int64_t, const void *, const void *, const uint8_t *, const uint8_t *, void *, uint8_t *,
                                         hipStream_t);
template hipError_t launch_reward<double>(const KParams &, int64_t, const void *, const void *, const uint8_t *, const uint8_t *, void *,
                                          uint8_t *, hipStream_t);

}  // namespace gpr
