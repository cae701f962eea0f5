/*
 * gpr.h -- C ABI of libgpr_hip.so, the MI355X (gfx950) engine behind the vectorised planar-robotics environments.
 *
 * The reference (ubi-coro/gymnasium-planar-robotics) has NO native / FFI boundary: its only interface is the Python
 * Gymnasium class API. This header is therefore the boundary a maintainer would bind if the per-env hot path of
 *   BasicPlanarRoboticsSingleAgentEnv.step / reset        (gymnasium_planar_robotics/envs/basic_envs.py:1770-1950)
 *   BenchmarkPlanningEnv callbacks                          (envs/planning/benchmark_planning_env.py:355-688)
 *   BasicPlanarRoboticsEnv collision checks                 (envs/basic_envs.py:355-788)
 *   utils/geometry_2D_utils.py                              (:9-138)
 * moved behind a shared library. Each entry point cites the reference interface it replaces. INTEGRATION.md shows the
 * ctypes stub for the reference side.
 *
 * Conventions
 *  - plain C types only; every *_dev pointer is a DEVICE pointer owned by the caller (e.g. torch.Tensor.data_ptr());
 *    the library never allocates or frees caller-visible memory. It keeps one small device table per handle.
 *  - every call is asynchronous on the hipStream_t passed as `void *hip_stream` (NULL = default stream).
 *  - return value: 0 = ok, negative = error (GPR_E_*); gpr_last_error() returns a message. No exceptions, no asserts on
 *    the device: a mover that is above no tile is reported as wall_collision = 1 (the reference asserts, BE:514).
 *  - a handle is not thread-safe; use one handle per GPU / rank.
 *  - "real" below = float for GPR_PRECISION_F32 and GPR_PRECISION_F32_F64ACC, double for GPR_PRECISION_F64.
 */
#ifndef GPR_H
#define GPR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPR_MAX_MOVERS 16
#define GPR_MAX_TILES 4096 /* num_tiles_x * num_tiles_y */

enum {
    GPR_OK = 0,
    GPR_E_INVALID = -1,   /* bad argument / configuration */
    GPR_E_HIP = -2,       /* a HIP runtime call failed */
    GPR_E_UNBOUND = -3,   /* gpr_bind_state() has not been called */
    GPR_E_UNSUPPORTED = -4
};

enum { GPR_SHAPE_CIRCLE = 0, GPR_SHAPE_BOX = 1 };                  /* collision_params['shape'], BE:260 */
enum { GPR_AUTORESET_DISABLED = 0, GPR_AUTORESET_SAME_STEP = 1, GPR_AUTORESET_NEXT_STEP = 2 };
enum {
    GPR_PRECISION_F32 = 0,        /* float storage, float arithmetic (throughput mode; B(M) = 72 M + 17 bytes / env-step) */
    GPR_PRECISION_F32_F64ACC = 1, /* float storage, double arithmetic inside a step */
    GPR_PRECISION_F64 = 2         /* double storage and arithmetic (the reference's dtype; exact-decision parity mode) */
};

/* Constructor arguments of BenchmarkPlanningEnv (PL:165-185) that reach the hot path, plus the engine's own. */
typedef struct {
    int64_t num_envs;               /* envs owned by THIS handle (local shard) */
    int64_t env_id_offset;          /* global id of local env 0: random streams are keyed by global id, so results do not
                                       depend on how envs are sharded over GPUs */
    int32_t num_movers;             /* 1..GPR_MAX_MOVERS */
    int32_t num_tiles_x, num_tiles_y;
    const int8_t *layout_tiles;     /* HOST, [num_tiles_x * num_tiles_y] row-major in {0,1} (PL:167, BE:195) */
    double tile_half_size[2];       /* tile_params['size'][:2], half sizes (BE:201) */
    const double *tile_x_centres;   /* HOST [num_tiles_x] = x_pos_tiles[:,0] of get_tile_xy_pos (BE:1292-1310); NULL: computed */
    const double *tile_y_centres;   /* HOST [num_tiles_y]; NULL: computed */
    int32_t c_shape;                /* GPR_SHAPE_* */
    const double *c_size;           /* HOST [num_movers * 2]: circle radius in [2i]; box half sizes in [2i],[2i+1] (BE:261) */
    double c_offset, c_offset_wall; /* BE:262-263 */
    double cycle_time;              /* MuJoCo option/timestep, BE:283,1132 */
    int32_t num_cycles;             /* PL:177 */
    double v_max, a_max, j_max;     /* PL:179-181 */
    int32_t learn_jerk;             /* PL:182 */
    double threshold_pos;           /* PL:183 */
    double std_noise[3];            /* position, velocity, acceleration (BE:184-192) */
    int32_t max_episode_steps;      /* TimeLimit of the registered ids (gymnasium_planar_robotics/__init__.py:28); <= 0: none */
    int32_t autoreset_mode;         /* GPR_AUTORESET_* */
    double min_xy_pos[2], max_xy_pos[2]; /* PL:262-267 */
    double min_goal_dist;           /* PL:270-274 */
    int32_t max_reset_trials;       /* bound on the rejection loops of PL:369 / PL:395 */
    int32_t precision;              /* GPR_PRECISION_* */
} gpr_config;

/* Environment state, DEVICE memory owned by the caller. Row-major [num_envs][2*num_movers] "real":
 * (x,y) of mover 0, (x,y) of mover 1, ... -- the same ordering as the reference's achieved_goal (PL:561-569). */
typedef struct {
    void *pos_dev;        /* qpos[:2] of every mover */
    void *vel_dev;        /* qvel[:2] */
    void *acc_dev;        /* qacc[:2] == actuator activation in jerk mode (PL:305-312, 433) */
    void *goal_dev;       /* self.goals (PL:415) */
    int32_t *elapsed_dev; /* [num_envs] steps since reset (TimeLimit._elapsed_steps); -1 = "reset me" in next-step mode */
} gpr_state;

/* What step()/reset() return (BE:1950, BE:1833), batched. DEVICE memory owned by the caller. */
typedef struct {
    void *observation_dev;   /* [E][2M] (velocities) or [E][4M] (velocities then accelerations) real, PL:560-567 */
    void *achieved_goal_dev; /* [E][2M] real, PL:569 */
    void *desired_goal_dev;  /* [E][2M] real, PL:571; may alias gpr_state.goal_dev (then it is not rewritten) */
    void *reward_dev;        /* [E] real, PL:502-534 */
    uint8_t *terminated_dev, *truncated_dev;                        /* [E], PL:459-479 / TimeLimit */
    uint8_t *is_success_dev, *mover_collision_dev, *wall_collision_dev; /* [E] info dict, PL:596-601 */
    void *final_observation_dev, *final_achieved_goal_dev, *final_desired_goal_dev; /* nullable; same-step autoreset only */
    /* nullable (both or none): per-agent views of a PettingZoo-parallel layer, written by gpr_step / gpr_step_n IN THE SAME LAUNCH
     * (the step kernel holds every mover's observation in the lanes of its env group): agent_observation_dev [E][M][4M+2] real,
     * agent_reward_dev [E][M] real -- layout and semantics as documented at gpr_agent_views below, which produces the same bytes from
     * the stored outputs in a launch of its own (used after gpr_reset, whose kernel does not write views). */
    void *agent_observation_dev, *agent_reward_dev;
} gpr_step_out;

typedef struct gpr_env gpr_env;

/* BenchmarkPlanningEnv.__init__ (PL:165-291): validates the configuration, builds the tile tables (BE:203-221). */
int gpr_create(const gpr_config *cfg, gpr_env **out_env);
void gpr_destroy(gpr_env *env);
/* message of the last failed call on this handle (env == NULL: of the last failed gpr_create on this thread) */
const char *gpr_last_error(const gpr_env *env);
int gpr_bind_state(gpr_env *env, const gpr_state *state);
/* reset(seed=...) part of BE:1789-1791: re-keys every random stream and restarts the call counter */
int gpr_seed(gpr_env *env, uint64_t seed);
/* Position of the handle's random streams (RNG spec gpr-rng-v2: every draw is a pure function of seed, call counter `tick`, kind,
 * global env id, mover, draw index). Together with the caller-owned gpr_state tensors this is ALL the state of a run: save the
 * pair next to the tensors, restore both into a fresh handle, and the run continues bit for bit (the reference checkpoints
 * nothing; its counterpart is pickling `np_random` / `rng_noise`, BE:182,1789-1791). */
int gpr_get_rng_state(const gpr_env *env, uint64_t *seed_out, uint64_t *tick_out);
int gpr_set_rng_state(gpr_env *env, uint64_t seed, uint64_t tick);
/* reset() (BE:1770-1833 + PL:355-418). mask_dev: nullable [E] bytes, non-zero = reset this env. Fills `out`
 * (reward 0, terminated/truncated 0) for the envs it resets. */
int gpr_reset(gpr_env *env, const uint8_t *mask_dev, const gpr_step_out *out, void *hip_stream);
/* reset() on a caller-supplied stream of candidate sets instead of the engine's own draws -- the reference's
 * `self.np_random.uniform(low=self.min_xy_pos, high=self.max_xy_pos, size=(num_movers, 2))` (PL:377, PL:402) with the generator in the
 * caller's hands (scripted / curriculum resets; and how tests/golden/reset_template.npz -- reset() and _reset_callback called as written
 * on a scripted generator -- is checked). cand_dev [n_cand][E][2M] real: set k is the k-th sample the generator hands out. Per env the
 * start loop (PL:369-385) consumes sets until one passes its checks, the goal loop (PL:395-413) goes on from the next one; a loop that
 * reaches the stream's end keeps the last set (as max_reset_trials does). n_used_dev [E] int32, nullable: sets consumed per env. */
int gpr_reset_candidates(gpr_env *env, const uint8_t *mask_dev, int32_t n_cand, const void *cand_dev, int32_t *n_used_dev,
                         const gpr_step_out *out, void *hip_stream);
/* step(action) (BE:1835-1950 + PL:420-450 + TimeLimit). action_dev: [E][2M] real, clipped to the action space. */
int gpr_step(gpr_env *env, const void *action_dev, const gpr_step_out *out, void *hip_stream);

/* gpr_step on n handles in ONE call: handle i steps with actions_dev[i] into outs[i] on hip_streams[i]. For an env batch that is
 * stepped as several sub-batches, each with its own handle (contiguous env range, env_id_offset) and its own HIP stream, so that the
 * slow wavefronts at the end of one sub-batch's kernel overlap with the bulk of another's ("sub-batches"): the
 * host pays the call overhead once per env-step instead of once per sub-batch. Everything is validated before the first launch. */
int gpr_step_many(int32_t n, gpr_env *const *envs, const void *const *actions_dev, const gpr_step_out *const *outs,
                  void *const *hip_streams);

/* Do kernels on these two HIP streams really run side by side? The runtime maps streams onto a few hardware queues (four by default),
 * and two streams that land on the same queue execute strictly one after the other -- sub-batches on such a pair gain nothing. Runs
 * three pairs of 150-us single-wavefront kernels and writes 1 (they overlapped) or 0; both streams are synchronised. */
int gpr_streams_overlap(void *hip_stream_a, void *hip_stream_b, int32_t *overlap_out);

/* Thin wavefronts: a gpr_step / gpr_step_n launch whose batch would leave SIMDs idle is spread over more wavefronts, each holding fewer envs (what a
 * wavefront's step costs beyond its 40 quiet cycles -- flagged blocks, clamp-form blocks, resets served -- grows with the envs it holds,
 * and a small batch lasts as long as its slowest wavefront). Results do not depend on it: every random stream is keyed by global env id.
 * max_wavefronts = the most wavefronts one launch of this handle may be spread over; 0 turns it off; negative restores the default
 * (1 024 x {1, 1, 1, 2, 4} for groups of {1, 2, 4, 8, 16} lanes per env on MI355X's 1 024 SIMDs; GPR_THIN_WAVES overrides it at
 * gpr_create). Handles that step side by side (gpr_step_many) share the machine: give each its part. Call between launches; with
 * counters enabled call it before gpr_enable_counters (the per-wavefront counter table is sized by it).
 * Reference counterpart: none (the batch axis itself is the engine's; BE:1835-1950 steps one env). */
int gpr_set_wave_budget(gpr_env *env, int32_t max_wavefronts);

/* n_steps env-steps in ONE launch, for open-loop rollouts (random / scripted / replayed actions; the benchmark's rollouts): what
 * n_steps calls of gpr_step do, bit for bit (step i draws from call counter tick + i), without a launch, a grid-wide drain and a
 * whole-batch load burst between two steps. action_dev is [n_steps][E][2M] real; EVERY tensor of `out` (final_* included, if set)
 * is [n_steps][E][...], contiguous: row i holds what the i-th gpr_step would have returned. Needs an autoreset mode. */
int gpr_step_n(gpr_env *env, int32_t n_steps, const void *action_dev, const gpr_step_out *out, void *hip_stream);

/* Episode counters (engine addition, SURVEY.md section 5: the reference logs nothing). Every step kernel adds, per wavefront,
 * the popcounts of its done masks to a small device table owned by the handle; gpr_read_counters sums that table into the
 * caller's uint64 totals (added to what is there), and clears the table. Counting starts with the first
 * gpr_enable_counters(env, 1); while disabled the step kernels do not touch the table. Totals 6 and 7 are the two diagnostics the
 * reference prints as warnings (they are counted whether or not the episode counters are enabled, see gpr_read_diagnostics). */
enum {
    GPR_CNT_EPISODES = 0,     /* env-steps with terminated | truncated */
    GPR_CNT_TERMINATED = 1,   /* PL:459-479 */
    GPR_CNT_TRUNCATED = 2,    /* TimeLimit */
    GPR_CNT_SUCCESS = 3,      /* info['is_success'] */
    GPR_CNT_WALL = 4,         /* info['wall_collision'] */
    GPR_CNT_PAIR = 5,         /* info['mover_collision'] */
    GPR_CNT_RESET_EXHAUSTED = 6, /* rejection searches of a reset (starts, goals, the pushing env's object) that used up max_reset_trials:
                                    the last candidate stands although it is INVALID. The reference never gives up and warns every 100
                                    trials (PL:371-375, 397-401, PU:398-402); the host layer warns when this count is non-zero */
    GPR_CNT_ACTIONS_CLIPPED = 7, /* env-steps whose action lay outside the action space and was clipped (the reference warns, BE:1871-1873);
                                    counted only after gpr_count_clipped_actions(env, 1) */
    GPR_NUM_COUNTERS = 8
};
int gpr_enable_counters(gpr_env *env, int32_t enable);
int gpr_read_counters(gpr_env *env, uint64_t *totals_dev /* [GPR_NUM_COUNTERS], device, caller-owned */, void *hip_stream);
/* opt-in: let the step kernels count the env-steps the reference would print "action is not within the action space" for (BE:1871-1873) */
int gpr_count_clipped_actions(gpr_env *env, int32_t enable);
/* The two diagnostics since gpr_create as HOST values: totals_host[0] = GPR_CNT_RESET_EXHAUSTED, [1] = GPR_CNT_ACTIONS_CLIPPED. Waits for
 * hip_stream (a diagnostic read, not a hot-path call); independent of gpr_enable_counters / gpr_read_counters. Pass the stream the
 * handle's steps (and gpr_read_counters calls) are issued on: the words are read behind whatever that stream holds, and a step or a
 * counter reduction running on ANOTHER stream at the same moment may be seen half-way (a count a step too low or too high, never lost). */
int gpr_read_diagnostics(gpr_env *env, uint64_t *totals_host /* [2] */, void *hip_stream);

/* get_mover_qpos / get_mover_qvel / get_mover_qacc and get_mover_qpos_arr / _qvel_arr / _qacc_arr (BE:801-855, 1244-1290) for every env
 * of the handle. which: 0 qpos -> out [E][n_sel][7] = (x, y, z, qw, qx, qy, qz); 1 qvel -> [E][n_sel][6] = (vx, vy, vz, wa, wb, wc);
 * 2 qacc -> [E][n_sel][6]. mover_idx_dev [n_sel] int32 in [0, num_movers): row j of an env is mover mover_idx[j] (the order of the
 * reference's mover_names argument; an index outside the range is clamped into it). z is what the reference reports, the distance between the mover's bottom and the tile's top (BE:821-824: body z minus the
 * half height) -- constant in this planar engine, passed by the caller (initial_mover_zpos); the quaternion is the identity (planning
 * env) or the yaw rotation (pushing env); z / roll / pitch rates and accelerations are 0. add_noise: Gaussian noise of std_noise[which]
 * on EVERY component, as BE:828 / 841 / 855 add it, from random stream kind 5 of gpr-rng-v2 (key (seed, call counter, 5, global env id,
 * mover), components 2k and 2k+1 = normal pair k); a noisy read advances the call counter. Real type = the handle's storage type. */
int gpr_get_mover_state(gpr_env *env, int32_t which, int32_t n_sel, const int32_t *mover_idx_dev, double z_report, int32_t add_noise,
                        void *out_dev, void *hip_stream);
/* compute_reward / compute_terminated on HER batches (PL:459-534): achieved/desired [batch][2M] real. */
int gpr_compute_reward(gpr_env *env, int64_t batch, const void *achieved_dev, const void *desired_dev,
                       const uint8_t *mover_collision_dev, const uint8_t *wall_collision_dev, void *reward_dev,
                       uint8_t *terminated_dev, void *hip_stream);
/* qpos_is_valid (BE:459-788) on n qpos [n][7] f64; c_size_dev [n][2] f64 before offsets; out [n] int32. */
int gpr_qpos_is_valid(gpr_env *env, int64_t n, const double *qpos_dev, const double *c_size_dev, int32_t add_safety_offset,
                      int32_t *out_dev, void *hip_stream);
/* check_mover_collision (BE:355-424) on `batch` independent sets of num_movers qpos: qpos [batch][M][7] f64,
 * c_size [M][2] f64 (device); out [batch] bytes. */
int gpr_check_mover_collision(gpr_env *env, int64_t batch, const double *qpos_dev, const double *c_size_dev,
                              int32_t add_safety_offset, uint8_t *out_dev, void *hip_stream);

/* Per-agent views for a PettingZoo-parallel layer. The reference's BasicPlanarRoboticsMultiAgentEnv (BE:1629-1693) only names
 * one agent per mover ("mover_i", BE:1692-1693) and defines no multi-agent observation or reward, so this layout is the
 * engine's own (no reference parity to claim):
 *   agent_obs_dev    [E][M][4M+2] real: own velocity (2), own position (2), own goal (2), then for every OTHER mover j in
 *                    ascending j: position of j relative to the agent (2), velocity of j relative to the agent (2)
 *   agent_reward_dev [E][M] real: the env reward of PL:519-531 split per agent -- -50 / +50 for every agent on collision /
 *                    success, otherwise -1 for an agent whose own goal is not reached (PL:524) and 0 for one that is
 * Inputs are the outputs of gpr_step / gpr_reset as bound in `out` (observation [E][obs_dim], achieved_goal, desired_goal [E][2M],
 * reward, terminated, truncated [E]). With same-step autoreset the observation of an env that ended in this call is already the
 * new episode's; its per-agent REWARD is then computed from final_achieved_goal / final_desired_goal (the terminal state the env
 * reward was computed from), which must be bound in `out` in that mode. */
int gpr_agent_views(gpr_env *env, const gpr_step_out *out, void *agent_obs_dev, void *agent_reward_dev, void *hip_stream);

/* ---- BenchmarkPushingEnv (gymnasium_planar_robotics/envs/manipulation/benchmark_pushing_env.py, "PU") -----------------
 * One mover pushes one box over 3x3 tiles. The handle is created by gpr_create with num_movers = 1, a 3x3 layout and the
 * 'circle' collision shape (PU:195-196); gpr_push_configure adds the object. num_movers = 2 is the engine's extension to
 * BASELINE.json's "2 movers + 1 pushed object": both movers push the object, the two movers ending the episode like a wall
 * collision when their collision circles meet (the planning env's rule, BE:355-424); every per-mover array doubles.
 * Pinned by the reference: mover control and clamping (PU:419-448), observation / reward / terminated / info (PU:457-607), reset
 * sampling (PU:373-417), wall check. NOT pinned ("parity unpinned"): contact dynamics -- the reference delegates them to MuJoCo,
 * which this pipeline cannot run and which no reference test exercises; the kernels integrate the engine's own planar
 * rigid-body model gpr-push-v3: 3-DoF boxes; per (mover, object) pair a separating-axis test, a clipped two-point
 * manifold with a 0.1 mm contact margin, two normal rows solved exactly (2 x 2 LCP) + one friction row, two block Gauss-Seidel visits
 * per cycle; tile friction, joint damping (PU:335), impedance hold of the movers' yaw (PU:261-267). */
typedef struct {
    double mover_mass;          /* mover_params['mass'] (BE:230) */
    double mover_half_size[2];  /* mover_params['size'][:2] (BE:250) */
    double object_half_size;    /* PU:173 */
    double object_mass;         /* PU:175 */
    double object_damping;      /* PU:335 */
    double friction_tile, friction_contact, friction_torsional; /* MuJoCo defaults: 1, 1, 0.005 */
    double gravity;             /* 9.81 */
    double rot_stiffness;       /* PU:266 */
    double contact_timeconst;   /* MuJoCo default solref[0] = 0.02 */
    double object_noise;        /* PU:178 */
    double min_mo_dist;         /* PU:279-288 */
    double obj_min_xy[2], obj_max_xy[2]; /* PU:257-258 */
} gpr_push_params;

/* extra state, DEVICE (NM = num_movers): mover_rot [E][2 NM] = (yaw, yaw rate) per mover; object [E][6] = (x, y, yaw, vx, vy, yaw
 * rate). gpr_state.goal_dev holds the object's goal [E][2]; pos / vel / acc are the movers' [E][2 NM]. */
typedef struct {
    void *mover_rot_dev;
    void *object_dev;
} gpr_push_state;

int gpr_push_configure(gpr_env *env, const gpr_push_params *params, const gpr_push_state *state);
/* reset() / step() of BenchmarkPushingEnv; out.observation_dev is [E][4 NM] (all mover positions, all velocities) or [E][6 NM] (+ all
 * accelerations, PU:540-545), achieved_goal / desired_goal are [E][2] (PU:564-568); action_dev is [E][2 NM]. */
int gpr_push_reset(gpr_env *env, const uint8_t *mask_dev, const gpr_step_out *out, void *hip_stream);
/* ... with the reset options of PU:389-391 / PU:412-414: mover_start_xy_dev [E][2 NM] real replaces the sampled mover start(s)
 * ('mover_start_xy_pos'), object_goal_xy_dev [E][2] real the sampled goal ('object_goal_xy_pos'); either may be NULL. The object's
 * start is still rejection-sampled against min_mo_dist from the given mover position(s), as PU:396-407 does. */
int gpr_push_reset_opts(gpr_env *env, const uint8_t *mask_dev, const void *mover_start_xy_dev, const void *object_goal_xy_dev,
                        const gpr_step_out *out, void *hip_stream);
/* ... and with the object's start candidates (PU:403-405: `np_random.uniform(object_min_xy_pos, object_max_xy_pos)` inside the rejection
 * loop) from the caller: object_cand_dev [n_cand][E][2] real, tried in order until one is farther than min_mo_dist from every mover
 * (PU:407), the last one standing otherwise; n_used_dev [E] int32 nullable. */
int gpr_push_reset_candidates(gpr_env *env, const uint8_t *mask_dev, const void *mover_start_xy_dev, const void *object_goal_xy_dev,
                              int32_t n_cand, const void *object_cand_dev, int32_t *n_used_dev, const gpr_step_out *out, void *hip_stream);
int gpr_push_step(gpr_env *env, const void *action_dev, const gpr_step_out *out, void *hip_stream);

/* ---- stateless operators (f64, device pointers) --------------------------------------------------------------- */
/* geometry_2D_utils.check_line_segments_intersect (GEO:9-69): p1,p2,q1,q2 [n][2]; out [n] bytes */
int gpr_check_line_segments_intersect(int64_t n, const double *p1_dev, const double *p2_dev, const double *q1_dev,
                                      const double *q2_dev, uint8_t *out_dev, void *hip_stream);
/* geometry_2D_utils.get_2D_rect_vertices (GEO:72-104): qpos [n][7], size [n][2]; out [n][2][4] */
int gpr_get_2D_rect_vertices(int64_t n, const double *qpos_dev, const double *size_dev, double *out_dev, void *hip_stream);
/* geometry_2D_utils.check_rectangles_intersect (GEO:107-138) */
int gpr_check_rectangles_intersect(int64_t n, const double *qpos_r1_dev, const double *qpos_r2_dev, const double *size_r1_dev,
                                   const double *size_r2_dev, uint8_t *out_dev, void *hip_stream);
/* BenchmarkPlanningEnv.ensure_max_dyn_val (PL:610-645), row-wise: cur/deriv [n][2], max_value [n] */
int gpr_ensure_max_dyn_val(int64_t n, const double *cur_dev, const double *max_value_dev, const double *deriv_dev,
                           double cycle_time, double *out_vals_dev, double *out_derivs_dev, void *hip_stream);

/* library / build info: "gfx950;..." */
const char *gpr_version(void);

#ifdef __cplusplus
}
#endif
#endif
