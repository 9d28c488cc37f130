// hostlib.h -- shared pieces of libdpmmhost.so: the passive thread pool and the counter-based RNG.
#pragma once
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <time.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace dpmmh {

// Passive thread pool with a bounded spin.  OpenMP's idle workers spin-wait after every parallel region; in a
// container with a CFS CPU quota (the GPU box: 256 visible CPUs, 16 CPUs of quota) that burns the quota and the
// whole process gets throttled for ~50 ms every 100 ms.  Workers here sleep on a condition variable between jobs;
// after finishing a job they first poll for a successor for at most `spin_us` microseconds (the host step of a sweep
// issues its parallel regions back to back: a futex wake-up costs 20-50 us each, the bounded poll a few hundred ns),
// and they sleep through the GPU phases.  Items are handed out dynamically through an atomic counter.
class Pool {
  public:
    static Pool &get() { static Pool p; return p; }
    void set_spin_us(int us) { spin_ns_.store(us < 0 ? 0 : (int64_t)us * 1000); }
    // keep the workers (existing and future) on the CPUs of one NUMA node; node < 0 or an unreadable topology: no change
    void set_numa_node(int node) {
        cpu_set_t set;
        if (!node_cpus(node, &set)) return;
        std::lock_guard<std::mutex> lk(call_mu_);
        affinity_ = set; have_affinity_ = true;
        for (auto &t : threads_) pthread_setaffinity_np(t.native_handle(), sizeof(set), &set);
    }
    // CPUs of a NUMA node that this process may use (/sys/devices/system/node/nodeN/cpulist intersected with the current mask)
    static bool node_cpus(int node, cpu_set_t *out) {
        if (node < 0) return false;
        char path[96];
        snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
        FILE *f = fopen(path, "r");
        if (!f) return false;
        char buf[4096] = {0};
        const bool ok = fgets(buf, sizeof(buf), f) != nullptr;
        fclose(f);
        if (!ok) return false;
        cpu_set_t allowed;
        CPU_ZERO(&allowed);
        if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
        CPU_ZERO(out);
        int n = 0;
        for (char *p = buf; *p && *p != '\n';) {
            char *e;
            long a = strtol(p, &e, 10), b = a;
            if (e == p) break;
            if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); }
            for (long c = a; c <= b && c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &allowed)) { CPU_SET(c, out); ++n; }
            p = (*e == ',') ? e + 1 : e;
        }
        return n > 0;
    }
    // run fn(item, slot) for item in [0, n) on up to `nthreads` threads (slot < nthreads identifies the thread)
    void run(int n, int nthreads, const std::function<void(int, int)> &fn) {
        if (n <= 0) return;
        if (nthreads > n) nthreads = n;
        if (nthreads <= 1) { for (int i = 0; i < n; ++i) fn(i, 0); return; }
        std::unique_lock<std::mutex> call_lock(call_mu_);   // one job at a time
        ensure(nthreads - 1);
        fn_ = &fn; n_ = n; next_.store(0); pending_.store(nthreads - 1);
        {
            std::lock_guard<std::mutex> lk(mu_);             // a worker between its predicate check and its sleep must not miss this
            state_.store(((state_.load() >> 16) + 1) << 16 | (uint64_t)(nthreads - 1), std::memory_order_release);
        }
        cv_.notify_all();
        work(0);
        if (pending_.load(std::memory_order_acquire) != 0) {
            const int64_t lim = spin_ns_.load();
            const int64_t t0 = lim > 0 ? now_ns() : 0;
            while (pending_.load(std::memory_order_acquire) != 0) {
                if (lim <= 0 || now_ns() - t0 > lim) {
                    std::unique_lock<std::mutex> lk(mu_);
                    done_cv_.wait(lk, [&] { return pending_.load(std::memory_order_acquire) == 0; });
                    break;
                }
                cpu_relax();
            }
        }
        fn_ = nullptr;
    }
    // Wake sleeping workers ahead of a job that is about to arrive (the master knows when the GPU will hand the statistics back):
    // they poll for at most `spin_us` and start at once instead of paying the futex wake-up (20-50 us for a dozen threads) inside the
    // timed path; if nothing arrives they go back to sleep.
    void prewake() {
        { std::lock_guard<std::mutex> lk(mu_); pre_.fetch_add(1, std::memory_order_release); }
        cv_.notify_all();
    }
  private:
    Pool() {}
    ~Pool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_.store(true); state_.store(((state_.load() >> 16) + 1) << 16); }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    static int64_t now_ns() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (int64_t)t.tv_sec * 1000000000 + t.tv_nsec; }
    static void cpu_relax() { __builtin_ia32_pause(); }
    void ensure(int nworkers) {
        while ((int)threads_.size() < nworkers) {
            const int id = (int)threads_.size();
            threads_.emplace_back([this, id] { loop(id); });
            if (have_affinity_) pthread_setaffinity_np(threads_.back().native_handle(), sizeof(affinity_), &affinity_);
        }
    }
    void work(int slot) {
        for (;;) {
            const int i = next_.fetch_add(1);
            if (i >= n_) break;
            (*fn_)(i, slot);
        }
    }
    void loop(int id) {
        uint64_t seen = 0;                                   // generation of the last job this worker looked at
        uint64_t seen_pre = 0;                               // last prewake() this worker answered
        for (;;) {
            uint64_t st = state_.load(std::memory_order_acquire);
            if ((st >> 16) == seen) {
                const int64_t lim = spin_ns_.load();
                if (lim > 0 && seen != 0) {
                    const int64_t t0 = now_ns();
                    while (((st = state_.load(std::memory_order_acquire)) >> 16) == seen && now_ns() - t0 < lim) cpu_relax();
                }
                if ((st >> 16) == seen) {
                    std::unique_lock<std::mutex> lk(mu_);
                    cv_.wait(lk, [&] { return (state_.load(std::memory_order_acquire) >> 16) != seen || pre_.load(std::memory_order_acquire) != seen_pre; });
                    st = state_.load(std::memory_order_acquire);
                    seen_pre = pre_.load(std::memory_order_acquire);
                    if ((st >> 16) == seen) continue;        // prewake: back to the bounded poll
                }
            }
            seen = st >> 16;                                 // generation and worker count come from ONE snapshot
            if (stop_.load()) return;
            if (id >= (int)(st & 0xFFFF)) continue;          // not needed for this job
            work(id + 1);
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(mu_);
                done_cv_.notify_one();
            }
        }
    }
    std::mutex mu_, call_mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> threads_;
    const std::function<void(int, int)> *fn_ = nullptr;
    std::atomic<int> next_{0}, pending_{0};
    int n_ = 0;
    std::atomic<uint64_t> state_{0};                         // (generation << 16) | workers wanted
    std::atomic<int64_t> spin_ns_{0};
    std::atomic<uint64_t> pre_{0};
    std::atomic<bool> stop_{false};
    cpu_set_t affinity_;
    bool have_affinity_ = false;
};

struct Philox {
    uint32_t key[2];
    uint32_t ctr[4];
    uint32_t out[4];
    int have = 0;
    double spare = 0.0;
    bool has_spare = false;
    Philox(uint64_t seed, uint32_t id, uint32_t epoch, uint32_t stream) {
        key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32);
        ctr[0] = 0; ctr[1] = id; ctr[2] = epoch; ctr[3] = stream;
    }
    void refill() {
        uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
        for (int r = 0; r < 10; ++r) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
            c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
        ctr[0] += 1;  // 2^32 blocks per (id, epoch, stream): ample
        have = 4;
    }
    uint32_t u32() { if (!have) refill(); return out[--have]; }
    double uniform() {  // (0,1), 53 bits
        const uint64_t a = u32(), b = u32();
        return ((double)(((a << 32) | b) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    }
    double normal() {
        if (has_spare) { has_spare = false; return spare; }
        const double u1 = uniform(), u2 = uniform();
        const double r = sqrt(-2.0 * log(u1));
        double s, c;
        sincos(6.283185307179586476925 * u2, &s, &c);
        spare = r * s; has_spare = true;
        return r * c;
    }
    double gamma(double a) {  // Marsaglia-Tsang, shape a > 0, scale 1
        if (a < 1.0) {
            const double u = uniform();
            return gamma(a + 1.0) * pow(u, 1.0 / a);
        }
        const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
        for (;;) {
            double x, v;
            do { x = normal(); v = 1.0 + c * x; } while (v <= 0.0);
            v = v * v * v;
            const double u = uniform();
            if (u < 1.0 - 0.0331 * x * x * x * x) return d * v;
            if (log(u) < 0.5 * x * x + d * (1.0 - v + log(v))) return d * v;
        }
    }
};


}  // namespace dpmmh
