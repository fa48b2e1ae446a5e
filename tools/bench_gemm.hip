// Developer micro-benchmark (GPU box): times gate-GEMM tile configurations on synthetic data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DFWN_ABL=n] tools/bench_gemm.hip -o /tmp/bench_gemm
#include "../tf-flowavenet_amd/csrc/flow_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static void* dalloc(size_t bytes, int fill) {
    void* p; CK(hipMalloc(&p, bytes));
    std::vector<unsigned short> h(bytes / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = fill ? (unsigned short)(0x3c00 + (rand() & 0xff) - ((rand() & 1) << 15)) : 0;
    CK(hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice));
    return p;
}
static hipEvent_t e0, e1;
template <class F> static void timeit(const char* name, double flops, F fn) {
    for (int i = 0; i < 3; ++i) fn();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int it = 20;
    for (int i = 0; i < it; ++i) fn();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    hipError_t e = hipGetLastError();
    printf("  %-28s %8.2f us  %7.1f TFLOP/s %s\n", name, ms * 1e3 / it, flops / (ms * 1e-3 / it) / 1e12, e == hipSuccess ? "" : hipGetErrorString(e));
}
#define GATE_CFG(BM, BN, WM, WN, BK, D)                                                                    \
    timeit("gate " #BM "x" #BN " w" #WM "x" #WN " k" #BK " d" #D, flops, [&] {                             \
        hipLaunchKernelGGL((gemm_ring_kernel<BM, BN, WM, WN, BK, D, GateProb>),                            \
                           dim3(((M + BM - 1) / BM) * (512 / BN)), dim3(64 * WM * WN), 0, 0, p, 512 / BN); \
    })

#define GATE_CFGK(BM, BN, WM, WN, BK, D, KSP)                                                              \
    timeit("gate " #BM "x" #BN " w" #WM "x" #WN " k" #BK " d" #D " ksp" #KSP, flops, [&] {                 \
        hipLaunchKernelGGL((gemm_ring_kernel<BM, BN, WM, WN, BK, D, GateProb, KSP>),                       \
                           dim3(((M + BM - 1) / BM) * (512 / BN)), dim3(64 * WM * WN * KSP), 0, 0, p, 512 / BN); \
    })

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8;
    const int T = 16128;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blk0 = argc > 2 ? atoi(argv[2]) : 0, blk1 = argc > 3 ? atoi(argv[3]) : 7;
    for (int blk = blk0; blk <= blk1; ++blk) {
        const int Ch = 1 << blk, Ti = T / (2 * Ch), M = B * Ti, cin = 40 * 2 * Ch, kcpad = (cin + 63) / 64 * 64;
        const bool hoist = M < 4096;
        void* h = dalloc((size_t)M * 512, 1);
        void* ca = dalloc((size_t)M * cin * 2, 1);
        void* Wd = dalloc(512ull * 768 * 2, 1);
        void* Wc = dalloc(512ull * kcpad * 2, 1);
        float* bias = (float*)dalloc(512 * 4, 0);
        float* P = (float*)dalloc((size_t)M * 512 * 4, 0);
        void* o = dalloc((size_t)M * 512 * 2, 1);
        GateProb p{(const bf16*)h, hoist ? nullptr : (const bf16*)ca, hoist ? P : nullptr, (const bf16*)Wd, (const bf16*)Wc, bias, (bf16*)o, M, Ti, 1, cin, kcpad};
        const double flops = 2.0 * M * (768.0 + (hoist ? 0 : cin)) * 512;
        printf("block %d  M=%d K=%d\n", blk, M, 768 + (hoist ? 0 : cin));
#define HALO_CFG(BM, BN)                                                                                  \
    timeit("gate halo " #BM "x" #BN, flops, [&] {                                                         \
        hipLaunchKernelGGL((gate_halo_kernel<BM, BN, GateProb>), dim3(((M + BM - 1) / BM) * (512 / BN)),  \
                           dim3(BM * 4), 0, 0, p, 512 / BN); })
        if (M >= 16128) HALO_CFG(256, 256);
        HALO_CFG(256, 128);
        HALO_CFG(128, 256);
        HALO_CFG(128, 128);
        HALO_CFG(64, 128);
        if (M >= 16128) {
            GATE_CFG(256, 256, 2, 4, 64, 2);
            GATE_CFG(256, 256, 4, 4, 64, 2);
            GATE_CFG(256, 256, 4, 4, 32, 2);
            GATE_CFG(256, 256, 4, 4, 32, 3);
            GATE_CFG(128, 256, 2, 4, 32, 3);
            GATE_CFG(128, 256, 2, 4, 32, 2);
            GATE_CFG(256, 128, 4, 2, 32, 3);
            GATE_CFG(128, 128, 4, 2, 64, 2);
            GATE_CFG(256, 128, 4, 2, 64, 3);
            GATE_CFG(256, 128, 8, 2, 64, 3);
            GATE_CFG(128, 256, 2, 4, 64, 3);
            GATE_CFG(128, 256, 4, 4, 64, 3);
        }
        GATE_CFG(128, 128, 2, 2, 64, 2);
        GATE_CFG(128, 128, 4, 2, 64, 3);
        GATE_CFG(128, 128, 4, 2, 64, 2);
        GATE_CFG(64, 128, 2, 2, 64, 4);
        GATE_CFG(64, 256, 2, 4, 64, 3);
        if (M <= 4096) {
            GATE_CFG(64, 64, 2, 1, 64, 4);
            GATE_CFGK(64, 64, 2, 1, 64, 4, 2);
            GATE_CFGK(64, 64, 2, 1, 64, 8, 2);
            GATE_CFGK(64, 64, 2, 1, 128, 4, 2);
            GATE_CFGK(64, 64, 2, 1, 128, 3, 2);
            GATE_CFGK(64, 64, 2, 1, 128, 4, 4);
            GATE_CFGK(64, 64, 2, 1, 128, 4, 8);
            GATE_CFGK(32, 64, 1, 1, 128, 4, 8);
            GATE_CFGK(32, 64, 1, 1, 128, 4, 4);
            GATE_CFGK(64, 128, 2, 2, 128, 3, 4);
            GATE_CFGK(64, 64, 2, 1, 64, 4, 4);
            GATE_CFGK(64, 128, 2, 2, 64, 4, 2);
            GATE_CFGK(128, 128, 4, 2, 64, 3, 2);
        }
        {
            void* Ws = dalloc(256ull * 512 * 2, 1);
            void* Wf = dalloc(256ull * 256 * 2, 1);
            const int npt = Ch > 32 ? (Ch + 31) / 32 : 1;
            void* Wz = dalloc((size_t)npt * 64 * 256 * 2, 1);
            float* bz = (float*)dalloc(4096, 0);
            float* an = (float*)dalloc(8 * Ch * 4 + 64, 0);
            float* xa = (float*)dalloc((size_t)M * Ch * 4, 0);
            float* xb = (float*)dalloc((size_t)M * Ch * 4, 0);
            float* part = (float*)dalloc(1 << 16, 0);
            void* scr = dalloc((size_t)M * 1024, 0);
            timeit(fwn_tail_is_split(M) ? "tail (N-split, 3 launches)" : "tail (fused)", 2.0 * M * (512.0 * 256 + 256 * 256 + 256.0 * 2 * Ch), [&] {
                fwn_launch_tail(o, (long)M * 256, 2, Ws, bias, Wf, bias, Wz, bz, bz, an, xa, xb, part, M, Ch, npt, 0, scr,
                                (char*)scr + (size_t)M * 512, 0); });
            CK(hipDeviceSynchronize());
            CK(hipFree(scr));
            CK(hipDeviceSynchronize());
            for (void* q : {Ws, Wf, Wz, (void*)bz, (void*)an, (void*)xa, (void*)xb, (void*)part}) CK(hipFree(q));
        }
        CK(hipDeviceSynchronize());
        for (void* q : {h, ca, Wd, Wc, (void*)bias, (void*)P, o}) CK(hipFree(q));
    }
    return 0;
}
