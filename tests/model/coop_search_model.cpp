// A host model of the shared reset searches of the 16-lane step kernels (gymnasium-planar-robotics_amd/csrc/gpr_env_kernels.hip,
// "16 LANES PER ENV: IDLE WAVEFRONTS JOIN THE SEARCHES STILL RUNNING"): the same counters (next unclaimed trial, best valid trial, wavefronts
// at work, formed), the same order of reads and writes, one host thread per wavefront, validity of a trial a pure function of (entry,
// search, trial). Checks, over many randomly scheduled runs: every queued entry is formed exactly once, with the FIRST valid trial of each
// of its two searches (or "exhausted" when there is none below the bound), and every wavefront leaves. TEST INFRASTRUCTURE: it models the
// protocol, it shares no code with the kernel.
//   usage: coop_search_model <runs> <seed>      exit code 0 = all runs consistent
#include <atomic>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

namespace {
constexpr int WAVES = 4, QCAP = 16, GROUPS = 4, RPC = 2;

inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

struct Block {
    std::atomic<int> reset_count{0}, waves_through{0};
    std::atomic<int> reset_list[QCAP], next[2 * QCAP], best[2 * QCAP], pend[QCAP], state[QCAP];
    std::atomic<int> formed_s[QCAP], formed_g[QCAP], formed_n[QCAP];
    uint32_t salt = 0;
    int max_t = 10000, accept = 45;
    Block() {
        for (int i = 0; i < QCAP; ++i) {
            reset_list[i] = -1; pend[i] = 0; state[i] = 0; formed_s[i] = -1; formed_g[i] = -1; formed_n[i] = 0;
            next[2 * i] = 0; next[2 * i + 1] = 0; best[2 * i] = INT_MAX; best[2 * i + 1] = INT_MAX;
        }
    }
    bool valid(int entry, int q, int trial) const { return mix(salt ^ (uint32_t)(entry * 2 + q) * 0x9E3779B9U ^ (uint32_t)trial * 0x85EBCA6BU) % (uint32_t)accept == 0u; }
};

void maybe_yield(uint32_t &r) { r = mix(r + 1); if ((r & 3u) == 0u) std::this_thread::yield(); if ((r & 63u) == 1u) { volatile int spin = (int)(r >> 20); while (spin-- > 0) {} } }

// one wavefront: first its own "cycles" (it queues some entries at random moments), then the queue service of the kernel
void wave(Block &B, int w, int to_queue, uint32_t r, std::atomic<int> &gate) {
    gate.fetch_add(1);
    while (gate.load() < WAVES) {}  // all wavefronts start together
    for (int k = 0; k < to_queue; ++k) {
        maybe_yield(r);
        const int slot = B.reset_count.fetch_add(1);
        maybe_yield(r);
        B.reset_list[slot].store(w * 100 + k, std::memory_order_release);  // (the entry is written after the count moved on, as in the kernel)
    }
    B.waves_through.fetch_add(1, std::memory_order_release);
    for (;;) {
        // ---- one look at the queue (the kernel: lane i reads entry i; counters first, then who is at work) ----
        const bool all_through = B.waves_through.load(std::memory_order_acquire) == WAVES;
        const int cnt = B.reset_count.load();
        int st[QCAP], b0[QCAP], b1[QCAP], n0[QCAP], n1[QCAP], pd[QCAP];
        for (int i = 0; i < QCAP; ++i) { st[i] = B.state[i].load(); b0[i] = B.best[2 * i].load(); b1[i] = B.best[2 * i + 1].load(); n0[i] = B.next[2 * i].load(); n1[i] = B.next[2 * i + 1].load(); }
        std::atomic_thread_fence(std::memory_order_acquire);
        maybe_yield(r);
        for (int i = 0; i < QCAP; ++i) pd[i] = B.pend[i].load();
        int act = 0, sel = 0, form = -1, help = -1;
        bool open = false;
        for (int i = 0; i < QCAP && i < cnt; ++i) {
            if (st[i] != 0) continue;
            open = true;
            const bool o0 = n0[i] < (b0[i] < B.max_t ? b0[i] : B.max_t), o1 = n1[i] < (b1[i] < B.max_t ? b1[i] : B.max_t);
            if (act == 0 && o0 && n0[i] == 0) { act = 1; sel = 2 * i; }
            if (act == 0 && o1 && n1[i] == 0) { act = 1; sel = 2 * i + 1; }
            if (help < 0 && (o0 || o1)) help = 2 * i + (o0 ? 0 : 1);
#ifdef MODEL_BREAK_AT_WORK  // (mutation: form without looking at who is at work -- the checks below must catch this)
            if (form < 0 && !o0 && !o1) form = i;
#else
            if (form < 0 && !o0 && !o1 && pd[i] == 0) form = i;
#endif
        }
        if (act == 0 && form >= 0) { int expect = 0; if (B.state[form].compare_exchange_strong(expect, 1)) { act = 2; sel = form; } }
        if (act == 0 && help >= 0) { act = 1; sel = help; }
        if (act == 1) B.pend[sel >> 1].fetch_add(1);  // at work from here on
        if (act == 0 && !open && all_through) return;
        if (act == 0) { std::this_thread::yield(); continue; }
        const int ei = act == 1 ? sel >> 1 : sel;
        while (B.reset_list[ei].load(std::memory_order_acquire) < 0) std::this_thread::yield();
        if (act == 1) {
            // ---- this wavefront's stay with search `sel`: the next claim is made before the rounds in hand are evaluated ----
            int cl = B.next[sel].fetch_add(RPC * GROUPS, std::memory_order_relaxed), seen = B.best[sel].load(std::memory_order_relaxed);
            for (bool found = false; !found;) {
                const int b = cl, lim = seen;
                if (!(b < (lim < B.max_t ? lim : B.max_t))) break;
                cl = B.next[sel].fetch_add(RPC * GROUPS, std::memory_order_relaxed); seen = B.best[sel].load(std::memory_order_relaxed);
                for (int rr = 0; rr < RPC && !found; ++rr) {
                    maybe_yield(r);
                    for (int g = 0; g < GROUPS; ++g) {
                        const int t = b + rr * GROUPS + g;
                        if (t < B.max_t && B.valid(ei, sel & 1, t)) {
                            int cur = B.best[sel].load(std::memory_order_relaxed);
                            while (t < cur && !B.best[sel].compare_exchange_weak(cur, t, std::memory_order_relaxed)) {}
                            found = true;
                            break;
                        }
                    }
                }
            }
            B.pend[ei].fetch_sub(1);
        } else {
            maybe_yield(r);
            B.formed_s[ei] = B.best[2 * ei].load(std::memory_order_acquire);
            B.formed_g[ei] = B.best[2 * ei + 1].load(std::memory_order_acquire);
            B.formed_n[ei].fetch_add(1);
        }
    }
}
}  // namespace

int main(int argc, char **argv) {
    const int runs = argc > 1 ? std::atoi(argv[1]) : 2000;
    uint32_t seed = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 1u;
    for (int run = 0; run < runs; ++run) {
        Block B;
        seed = mix(seed + 0x9E3779B9U);
        B.salt = seed;
        B.accept = (run % 5 == 0) ? 400 : 45;              // some runs with long searches
        B.max_t = (run % 7 == 0) ? 6 : (run % 11 == 0 ? 37 : 10000);  // ... and some with a bound most searches run into
        int q[WAVES], total = 0;
        for (int w = 0; w < WAVES; ++w) { q[w] = (int)(mix(seed + w) % 5u); if (total + q[w] > QCAP) q[w] = QCAP - total; total += q[w]; }
        std::vector<std::thread> th;
        std::atomic<int> gate{0};
        for (int w = 0; w < WAVES; ++w) th.emplace_back(wave, std::ref(B), w, q[w], mix(seed ^ (uint32_t)w), std::ref(gate));
        for (auto &t : th) t.join();
        for (int i = 0; i < total; ++i) {
            int want[2];
            for (int s = 0; s < 2; ++s) { want[s] = INT_MAX; for (int t = 0; t < B.max_t; ++t) if (B.valid(i, s, t)) { want[s] = t; break; } }
            if (B.formed_n[i] != 1 || B.formed_s[i] != want[0] || B.formed_g[i] != want[1]) {
                std::printf("run %d entry %d: formed %d times with trials %d / %d, expected %d / %d\n", run, i, B.formed_n[i].load(), B.formed_s[i].load(), B.formed_g[i].load(), want[0], want[1]);
                return 1;
            }
        }
    }
    std::printf("ok: %d runs\n", runs);
    return 0;
}
