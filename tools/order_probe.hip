// order_probe.hip -- what does it cost to let two kernels share the chip on this platform?  (round 5, pipelined frame step)
//   1. hipExtAnyOrderLaunch: a kernel launched with it does not wait for the previous kernel of ITS OWN stream (AQL barrier bit
//      clear).  Honoured here?  Also when the launch is captured into a hipGraph?
//   2. cross-queue joins: hipEventRecord + hipStreamWaitEvent vs hipStreamWaitValue32 on signal memory written by a kernel:
//      latency from the writer's end to the waiter's first instruction.
//   3. what a hipEventRecord between two kernels of one stream costs the second one.
// Timestamps: s_memrealtime (100 MHz, one counter for the chip).   hipcc --offload-arch=gfx950 -O2 tools/order_probe.hip -o /tmp/order_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin_kernel(unsigned long long *ts, int slot, unsigned ticks) {      // ts[2*slot] = start, ts[2*slot+1] = end
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) ts[2 * slot] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0 && blockIdx.x == 0) ts[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
}
__global__ void spin_then_flag(unsigned long long *ts, int slot, unsigned ticks, unsigned *flag) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) ts[2 * slot] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) {
        ts[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
        __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static double us(unsigned long long a, unsigned long long b) { return ((double)b - (double)a) / 100.0; }

int main() {
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    unsigned long long *ts, h[64];
    CK(hipMalloc(&ts, sizeof(h)));
    const unsigned T = 5000;   // 50 us
    // warm
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ts, 0, 100u);
    CK(hipStreamSynchronize(s));

    // ---- 1a. plain: K0 then K1 in one stream ----
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ts, 0, T);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ts, 1, T);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost));
        printf("1a in-order      : K1 starts %.1f us after K0 starts (K0 runs %.1f), gap end->start %.1f us\n", us(h[0], h[2]), us(h[0], h[1]), us(h[1], h[2]));
    }
    // ---- 1b. any-order ----
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ts, 0, T);
        hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ts, 1, T);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ts, 2, 100u);      // in-order again: must wait for both
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost));
        printf("1b any-order     : K1 starts %.1f us after K0 starts (overlap %s); K2 (in-order) starts %.1f us after max(end K0, end K1)\n",
               us(h[0], h[2]), h[2] < h[1] ? "YES" : "no", us(std::max(h[1], h[3]), h[4]));
    }
    // ---- 1c. any-order captured into a graph ----
    {
        hipGraph_t g; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ts, 0, T);
        hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ts, 1, T);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ts, 2, 100u);
        hipError_t e = hipStreamEndCapture(s, &g);
        if (e != hipSuccess) printf("1c capture with any-order launch: %s\n", hipGetErrorString(e));
        else {
            CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipGraphLaunch(ex, s));
                CK(hipStreamSynchronize(s));
                CK(hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost));
                printf("1c any-order in a graph: K1 starts %.1f us after K0 starts (overlap %s); K2 starts %.1f us after the later end\n",
                       us(h[0], h[2]), h[2] < h[1] ? "YES" : "no", us(std::max(h[1], h[3]), h[4]));
            }
        }
    }
    // ---- 2a. cross-queue join with an event ----
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ts, 0, T);
        CK(hipEventRecord(ev, s));
        CK(hipStreamWaitEvent(s2, ev, 0));
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s2, ts, 1, 100u);
        CK(hipStreamSynchronize(s2)); CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost));
        printf("2a event join    : waiter starts %.1f us after the writer's end\n", us(h[1], h[2]));
    }
    // ---- 2b. cross-queue join with hipStreamWaitValue32 on signal memory ----
    {
        int can = 0;
        (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
        unsigned *flag = nullptr;
        hipError_t e = hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory);
        printf("2b hipStreamWaitValue32: device attribute %d, signal memory alloc: %s\n", can, hipGetErrorString(e));
        if (e == hipSuccess) {
            unsigned zero = 0;
            CK(hipMemcpy(flag, &zero, 4, hipMemcpyHostToDevice));
            for (int rep = 0; rep < 4; ++rep) {
                e = hipStreamWaitValue32(s2, flag, (unsigned)(rep + 1), hipStreamWaitValueGte, 0xFFFFFFFFu);
                if (e != hipSuccess) { printf("   hipStreamWaitValue32: %s\n", hipGetErrorString(e)); break; }
                hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s2, ts, 1, 100u);
                hipLaunchKernelGGL(spin_then_flag, dim3(1), dim3(64), 0, s, ts, 0, T, flag);
                CK(hipStreamSynchronize(s2)); CK(hipStreamSynchronize(s));
                CK(hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost));
                printf("2b wait-value    : waiter starts %.1f us after the writer's flag write\n", us(h[1], h[2]));
            }
        }
    }
    // ---- 3. an event record between two kernels of one stream ----
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ts, 0, T);
        CK(hipEventRecord(ev, s));
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ts, 1, 100u);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost));
        printf("3  record between: gap end->start %.1f us\n", us(h[1], h[2]));
    }
    // ---- 3b. the same with another stream waiting on the event (what the pipelined step does) ----
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ts, 0, T);
        CK(hipEventRecord(ev, s));
        CK(hipStreamWaitEvent(s2, ev, 0));
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s2, ts, 2, 100u);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ts, 1, 100u);
        CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
        CK(hipMemcpy(h, ts, sizeof(h), hipMemcpyDeviceToHost));
        printf("3b record + foreign waiter: same-stream successor gap %.1f us, foreign waiter gap %.1f us\n", us(h[1], h[2]), us(h[1], h[4]));
    }
    return 0;
}
