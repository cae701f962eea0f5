// gpr_host.h -- launcher declarations shared by the translation units of libgpr_hip.so
#pragma once
#include <atomic>

#include "gpr_device.h"

namespace gpr {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE attribute of a kernel: remember per (kernel, device) that it was
// raised, and only once the call succeeded. `flags` is the kernel's own static array (zero-initialised). The launch workers of
// gpr_step_many reach this from several threads for the same kernel: the flags are atomic, and setting the attribute twice is harmless.
constexpr int GPR_MAX_DEVICES = 64;
using LdsFlag = std::atomic<bool>;
inline hipError_t raise_dynamic_lds(const void *fn, int bytes, LdsFlag (&flags)[GPR_MAX_DEVICES]) {
    int dev = 0;
    hipError_t he = hipGetDevice(&dev);
    if (he != hipSuccess) return he;
    if (dev >= 0 && dev < GPR_MAX_DEVICES && flags[dev].load(std::memory_order_acquire)) return hipSuccess;
    he = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (he == hipSuccess && dev >= 0 && dev < GPR_MAX_DEVICES) flags[dev].store(true, std::memory_order_release);
    return he;
}

// fused env kernels, one instantiation per (storage, compute) precision, each in its own translation unit
template <typename ST, typename RT> hipError_t launch_env_kernel(bool is_reset, int lpe, const KParams &P, hipStream_t s);
// (the float kernels of the 'box' shape and of the 16-lane groups live in a translation unit of their own, see GPR_ENV_PART in gpr_env_kernels.hip)
template <typename ST, typename RT> hipError_t launch_env_kernel_wide(bool is_reset, int lpe, const KParams &P, hipStream_t s);

// P carries the step outputs (obs / ag / dg / reward / terminated / truncated / final_*) the views are built from
template <typename ST> hipError_t launch_agent_views(const KParams &P, int lpe, int obs_dim, void *agent_obs, void *agent_reward, hipStream_t s);
// sums the per-wavefront episode counters [rows][6] into totals[0..5] (uint64, added) and clears the rows; moves the two diagnostic
// words diag[0..1] into totals[6..7] (and into diag[2..3], the running sums gpr_read_diagnostics reports)
hipError_t launch_counter_reduce(uint32_t *rows, int64_t n_rows, uint64_t *totals, uint32_t *diag, hipStream_t s);
// get_mover_qpos / qvel / qacc (+ _arr) of every env: out [E][n_sel][7 or 6] (gpr_ops_kernels.hip)
template <typename ST, typename RT>
hipError_t launch_mover_state(const KParams &P, int which, int n_sel, const int32_t *sel, double z_report, int add_noise, const void *mover_rot,
                              void *out, hipStream_t s);
hipError_t set_normal_table_ops(const double *tab);

// the double-precision normal-draw tables (gpr_device.h: g_normal_tab), one pointer instance per translation unit that draws in double
template <typename ST, typename RT> hipError_t set_normal_table_env(const double *tab);
hipError_t set_normal_table_push(const double *tab);

// BenchmarkPushingEnv kernels (gpr_push_kernels.hip)
template <typename ST, typename RT> hipError_t launch_push(bool is_reset, const KParams &P, const PushCst<RT> &Q, const PushPtrs &X, hipStream_t s);

// stand-alone f64 operators (gpr_ops_kernels.hip)
hipError_t launch_segments(int64_t n, const double *p1, const double *p2, const double *q1, const double *q2, uint8_t *out, hipStream_t s);
hipError_t launch_rect_vertices(int64_t n, const double *qpos, const double *size, double *out, hipStream_t s);
hipError_t launch_rects(int64_t n, const double *r1, const double *r2, const double *s1, const double *s2, uint8_t *out, hipStream_t s);
hipError_t launch_clamp(int64_t n, const double *cur, const double *maxv, const double *der, double dt, double *vals, double *ders, hipStream_t s);
hipError_t launch_qpos_valid(const KParams &P, int64_t n, const double *qpos, const double *c_size, int add_safety, int32_t *out, hipStream_t s);
hipError_t launch_mover_collision(const KParams &P, int64_t batch, const double *qpos, const double *c_size, int add_safety, uint8_t *out,
                                  hipStream_t s);
template <typename ST> hipError_t launch_reward(const KParams &P, int64_t batch, const void *ag, const void *dg, const uint8_t *mc,
                                                const uint8_t *wc, void *reward, uint8_t *terminated, hipStream_t s);

}  // namespace gpr
