// gpr_launch_worker.h -- the launch workers of gpr_step_many (gpr_api.hip), in plain C++ so that the very same code is built a second
// time for the host under -fsanitize=thread against a stub launch (tests/model/launch_worker_stress.cpp).
//
// A kernel launch costs the host ~3.5 us (scratch/ubench/graphlaunch.hip: 3.46 us per launch issued from one thread, 19.6 us for four on
// four streams in a sustained loop) -- as long as a 16 384-env sub-batch takes the GPU. Launches on DIFFERENT streams from DIFFERENT
// host threads do run in parallel (4.3 us for the same four), so gpr_step_many hands launch i > 0 to worker thread i and issues launch 0
// itself. A worker polls its mailbox while steps keep coming (a step is due every ~13 us) and goes to sleep on a condition variable
// after `spin_limit` polls (~200 us) without work, so an idle process burns nothing.
//
// The handshake. The poster does   state := POSTED ; if (asleep) notify      and the worker, before it blocks,
//                                  asleep := true  ; if (state != POSTED) wait.
// That is the store-buffering (Dekker) shape: with release / acquire both loads may return the old values -- x86 really does this, a
// store sits in the store buffer while the following load is served -- the worker then sleeps on a posted job, the poster skips the
// notify and wait() never returns. Round 3 shipped exactly that. All four accesses are SEQUENTIALLY CONSISTENT
// now: in the single total order either the worker's load follows the poster's store (it sees the job and does not sleep) or the
// poster's load follows the worker's store (it sees `asleep`, takes the mutex -- which the worker holds from before `asleep := true`
// until cv.wait has parked it -- and its notify cannot be early). `SEQ_CST` is a template parameter only so that the stress test can
// build the broken form (release stores, acquire loads) too and show that it catches it.
//
// wait() is bounded: a worker that does not answer within `timeout_ms` (a launch stuck inside the runtime) makes wait() return -1
// with the worker marked lost; the pool then replaces it and the caller reports an error instead of spinning for ever.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace gpr {

struct LaunchJob {
    void *env = nullptr;
    const void *action = nullptr;
    const void *out = nullptr;
    void *stream = nullptr;
    int rc = 0;
};

using LaunchFn = int (*)(const LaunchJob &);  // runs one job on the calling (worker) thread, returns its C-ABI code

template <bool SEQ_CST = true> class LaunchWorkerT {
    static constexpr std::memory_order HS_STORE = SEQ_CST ? std::memory_order_seq_cst : std::memory_order_release;
    static constexpr std::memory_order HS_LOAD = SEQ_CST ? std::memory_order_seq_cst : std::memory_order_acquire;

public:
    static constexpr int IDLE = 0, POSTED = 1, DONE = 2;
    static constexpr int WAIT_TIMED_OUT = -1000;

    LaunchWorkerT(LaunchFn fn, int spin_limit) : fn_(fn), spin_limit_(spin_limit) {}

    // the worker thread's body (never returns unless `quit` was posted: only the tests do that)
    void run() {
        for (;;) {
            int spins = 0;
            while (state_.load(std::memory_order_acquire) != POSTED) {
                if (++spins < spin_limit_) {
                    cpu_relax();
                } else {
                    std::unique_lock<std::mutex> lk(m_);
                    asleep_.store(true, HS_STORE);
                    cv_.wait(lk, [&] { return state_.load(HS_LOAD) == POSTED; });
                    asleep_.store(false, std::memory_order_relaxed);
                    sleeps_.fetch_add(1, std::memory_order_relaxed);
                    spins = 0;
                }
            }
            if (quit_) {
                state_.store(DONE, std::memory_order_release);
                return;
            }
            job_.rc = fn_(job_);
            state_.store(DONE, std::memory_order_release);
        }
    }

    // called by ONE thread at a time (gpr_step_many's pool lock), only while the worker is IDLE
    void post(const LaunchJob &j) {
        job_ = j;
        state_.store(POSTED, HS_STORE);
        if (asleep_.load(HS_LOAD)) {
            std::lock_guard<std::mutex> lk(m_);
            cv_.notify_one();
        }
    }

    // the posted job's return code, or WAIT_TIMED_OUT (the worker is then lost: never post to it again)
    int wait(long timeout_ms) {
        int spins = 0;
        std::chrono::steady_clock::time_point deadline{};
        while (state_.load(std::memory_order_acquire) != DONE) {
            cpu_relax();
            if ((++spins & 0x3fff) == 0) {  // look at the clock every ~100 us
                const auto now = std::chrono::steady_clock::now();
                if (deadline == std::chrono::steady_clock::time_point{}) deadline = now + std::chrono::milliseconds(timeout_ms);
                else if (now > deadline) { lost_ = true; return WAIT_TIMED_OUT; }
            }
        }
        const int rc = job_.rc;
        state_.store(IDLE, std::memory_order_relaxed);
        return rc;
    }

    void post_quit() { quit_ = true; post(LaunchJob{}); }
    bool lost() const { return lost_; }
    long sleeps() const { return sleeps_.load(std::memory_order_relaxed); }

private:
    static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    LaunchFn fn_;
    int spin_limit_;
    std::atomic<int> state_{IDLE};
    std::atomic<bool> asleep_{false};
    std::atomic<long> sleeps_{0};
    LaunchJob job_;
    std::mutex m_;
    std::condition_variable cv_;
    bool quit_ = false, lost_ = false;
};

using LaunchWorker = LaunchWorkerT<>;

}  // namespace gpr
