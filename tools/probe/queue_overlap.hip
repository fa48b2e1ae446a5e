// Probe: do two HIP streams overlap when one carries a chain of tiny kernels and the other kernels with grids far larger
// than the chip?  (Training step: data-gradient chain on one stream, weight-gradient work on a second.)
//   hipcc --offload-arch=gfx950 -O3 tools/probe/queue_overlap.hip -o tools/probe/queue_overlap_bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void spin_kernel(float* p, int iters) {      // ~iters * 8 dependent FMAs per thread
    float a = p[threadIdx.x & 63];
    for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
    if (a == 12345.678f) p[0] = a;
}
// persistent variant: `nwg` workgroups walk `total` work items
__global__ void spin_persistent(float* p, int iters, int total) {
    for (int w = blockIdx.x; w < total; w += gridDim.x) {
        float a = p[threadIdx.x & 63];
        for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
        if (a == 12345.678f) p[0] = a;
    }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char**) {
    float* p; hipMalloc(&p, 4096); hipMemset(p, 0, 4096);
    hipStream_t a, b;
    int least = 0, greatest = 0;
    hipDeviceGetStreamPriorityRange(&least, &greatest);
    const bool prio = argc > 1;       // any argument: the chain's stream gets the highest priority, the big grids the lowest
    printf("stream priority range: least %d greatest %d; %s\n", least, greatest, prio ? "A high / B low" : "default priorities");
    if (prio) { hipStreamCreateWithPriority(&a, hipStreamNonBlocking, greatest); hipStreamCreateWithPriority(&b, hipStreamNonBlocking, least); }
    else { hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking); }
    const int NA = 600, NB = 60;
    auto runA = [&]() { for (int i = 0; i < NA; ++i) hipLaunchKernelGGL(spin_kernel, dim3(16), dim3(256), 0, a, p, 1500); };
    auto runB_big = [&]() { for (int i = 0; i < NB; ++i) hipLaunchKernelGGL(spin_kernel, dim3(4096), dim3(256), 0, b, p, 4000); };
    auto runB_pers = [&](int nwg) { for (int i = 0; i < NB; ++i) hipLaunchKernelGGL(spin_persistent, dim3(nwg), dim3(256), 0, b, p, 4000, 4096); };
    auto timeit = [&](const char* name, auto fn) {
        fn(); hipDeviceSynchronize();
        double t0 = now(); fn(); hipDeviceSynchronize(); double t1 = now();
        printf("%-44s %8.3f ms\n", name, (t1 - t0) * 1e3);
    };
    timeit("A alone (600 x 16-WG kernels, dependent)", [&]() { runA(); });
    timeit("B alone (60 x 4096-WG kernels)", [&]() { runB_big(); });
    timeit("A || B big grids", [&]() { runB_big(); runA(); });
    timeit("A || B big grids (A issued first)", [&]() { runA(); runB_big(); });
    for (int nwg : {1024, 512, 256, 128}) {
        char nm[96];
        snprintf(nm, sizeof nm, "B persistent %d WGs alone", nwg);
        timeit(nm, [&]() { runB_pers(nwg); });
        snprintf(nm, sizeof nm, "A || B persistent %d WGs", nwg);
        timeit(nm, [&]() { runB_pers(nwg); runA(); });
    }
    // the same two chains as parallel branches of ONE hipGraph (the recorded training step): does replay overlap them?
    for (int order = 0; order < 3; ++order) {
        hipGraph_t graph; hipGraphExec_t exec;
        hipEvent_t e0, e1; hipEventCreateWithFlags(&e0, hipEventDisableTiming); hipEventCreateWithFlags(&e1, hipEventDisableTiming);
        hipStreamBeginCapture(a, hipStreamCaptureModeThreadLocal);
        hipEventRecord(e0, a); hipStreamWaitEvent(b, e0, 0);
        if (order == 0) { runB_big(); runA(); }
        else if (order == 1) { runA(); runB_big(); }
        else for (int i = 0; i < NB; ++i) {          // interleaved: 1 B launch per 10 A launches
            hipLaunchKernelGGL(spin_kernel, dim3(4096), dim3(256), 0, b, p, 4000);
            for (int k = 0; k < NA / NB; ++k) hipLaunchKernelGGL(spin_kernel, dim3(16), dim3(256), 0, a, p, 1500);
        }
        hipEventRecord(e1, b); hipStreamWaitEvent(a, e1, 0);
        hipStreamEndCapture(a, &graph);
        hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        const char* names[3] = {"graph: B recorded first", "graph: A recorded first", "graph: interleaved recording"};
        timeit(names[order], [&]() { hipGraphLaunch(exec, a); });
    }
    return 0;
}
