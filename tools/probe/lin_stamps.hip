// Diagnostic (GPU box): where a K chunk of fwn_gemm's small tile spends its cycles (s_memtime stamps, workgroup 0).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DFWN_STAMP -I include tools/probe/lin_stamps.hip -o tools/probe/lin_stamps_bin
// The shape is the dilated data-gradient GEMM of block 4 of the training step: M = 1600, N = 256, K = 3 taps x 512.
#include "../../tf-flowavenet_amd/csrc/train_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static void* dalloc(size_t bytes) {
    void* p; CK(hipMalloc(&p, bytes));
    std::vector<unsigned short> h(bytes / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0xff) - ((rand() & 1) << 15));
    CK(hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice));
    return p;
}
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 1600, Ti = 200, N = 256, K = 1536, NBUF = 96;
    std::vector<void*> W(NBUF), X(NBUF), Y(NBUF);
    for (int i = 0; i < NBUF; ++i) { W[i] = dalloc((size_t)N * K * 2); X[i] = dalloc((size_t)M * 512 * 2); Y[i] = dalloc((size_t)M * N * 2); }
    auto desc = [&](int i) {
        fwn_gemm_desc g = fwn_gemm_desc();
        for (int tap = 0; tap < 3; ++tap) { g.seg[tap].x = X[i]; g.seg[tap].rows = M; g.seg[tap].ld = 512; g.seg[tap].k = 512; g.seg[tap].shift = -(tap - 1) * 3; g.seg[tap].koff = tap * 512; }
        g.nseg = 3; g.M = M; g.N = N; g.Ti = Ti; g.W = W[i]; g.ldw = K; g.Y = Y[i]; g.ldy = N; g.nsplit = 1; g.oscale = 1.0f;
        return g;
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {     // a different (cold) weight / activation set per launch, like a flow inside a pass
        CK(hipEventRecord(e0));
        for (int i = 0; i < NBUF; ++i) { fwn_gemm_desc g = desc(i); fwn_gemm_launch(&g, 0); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("M %d N %d K %d: %.2f us per launch (cold operands, back to back)\n", M, N, K, ms * 1e3 / NBUF);
    }
    std::vector<unsigned long long> v(16 * 64 * 4 + 64);
    CK(hipMemcpyFromSymbol(v.data(), HIP_SYMBOL(fwn_ring_stamps), v.size() * 8));
    const int nw = 4, nq = K / 128;
    const unsigned long long t0 = v[16 * 64 * 4 + 0];
    printf("wave 0: kernel start -> loop end %llu, -> epilogue start %llu, -> end %llu cycles (s_memtime ticks at 100 MHz: x%.0f core cycles)\n",
           v[16 * 64 * 4 + 1] - t0, v[16 * 64 * 4 + 2] - t0, v[16 * 64 * 4 + 3] - t0, 21.0);
    for (int q = 0; q < nq; ++q) {
        double a = 0, b = 0, c = 0, st = 0;
        for (int w = 0; w < nw; ++w) {
            const unsigned long long* s = &v[(w * 64 + q) * 4];
            a += (double)(s[1] - s[0]); b += (double)(s[2] - s[1]); c += (double)(s[3] - s[2]); st += (double)(s[0] - t0);
        }
        printf("  chunk %2d  t=%6.0f  wait %5.0f  barrier %5.0f  mma+issue %5.0f   (ticks, mean over %d waves)\n", q, st / nw, a / nw, b / nw, c / nw, nw);
    }
    return 0;
}
