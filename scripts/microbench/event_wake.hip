// How long after a kernel's last store does the host learn that it finished?  A kernel spins ~200 us and then writes a flag to pinned host
// memory; one thread polls the flag, the main thread waits in hipEventSynchronize / hipStreamSynchronize.  Prints the delay of each wait
// relative to the moment the flag became visible, and the cost of re-launching from the host (flag seen -> next kernel's first store).
//   hipcc --offload-arch=gfx950 -O2 -o event_wake.bin event_wake.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#include <algorithm>
static inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void spin_then_flag(volatile unsigned *flag, unsigned value, long long cycles) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    __threadfence_system();
    *flag = value;
}
__global__ void flag_only(volatile unsigned *flag, unsigned value) { *flag = value; }
int main() {
    unsigned *flag; hipHostMalloc((void **)&flag, 64, hipHostMallocDefault); *flag = 0;
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    std::vector<double> d_evt, d_str, d_relaunch;
    for (int it = 1; it <= 60; ++it) {
        const unsigned v = 2 * it;
        std::atomic<double> t_flag{0.0};
        hipLaunchKernelGGL(spin_then_flag, dim3(1), dim3(64), 0, st, flag, v, 20000LL);     // wall_clock64: 100 MHz -> 200 us
        hipEventRecord(ev, st);
        std::thread poller([&] { while (*(volatile unsigned *)flag != v) {} t_flag = now_us(); });
        if (it % 2) hipEventSynchronize(ev); else hipStreamSynchronize(st);
        const double t_wait = now_us();
        poller.join();
        (it % 2 ? d_evt : d_str).push_back(t_wait - t_flag.load());
        // relaunch latency: host sees the flag (polling) and launches a kernel that writes the next value
        hipLaunchKernelGGL(spin_then_flag, dim3(1), dim3(64), 0, st, flag, v + 1000000u, 20000LL);
        while (*(volatile unsigned *)flag != v + 1000000u) {}
        const double t0 = now_us();
        hipLaunchKernelGGL(flag_only, dim3(1), dim3(64), 0, st, flag, v + 1);
        while (*(volatile unsigned *)flag != v + 1) {}
        d_relaunch.push_back(now_us() - t0);
        hipStreamSynchronize(st);
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mn = [](std::vector<double> v) { return *std::min_element(v.begin(), v.end()); };
    printf("hipEventSynchronize returns  %.1f us (median; min %.1f) after the kernel's flag is visible to a polling thread\n", med(d_evt), mn(d_evt));
    printf("hipStreamSynchronize returns %.1f us (median; min %.1f) after the flag\n", med(d_str), mn(d_str));
    printf("flag seen -> launch -> that kernel's store visible: %.1f us (median; min %.1f)\n", med(d_relaunch), mn(d_relaunch));
    return 0;
}
