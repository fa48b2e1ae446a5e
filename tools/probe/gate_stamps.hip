// Diagnostic (GPU box): where a step of the tap-sharing gate tile spends its cycles (s_memtime stamps per wave).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DFWN_STAMP tools/probe/gate_stamps.hip -o /tmp/gate_stamps
#include "../../tf-flowavenet_amd/csrc/flow_kernels.hip"
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
int main(int argc, char** argv) {
    const int fp8 = argc > 1 ? atoi(argv[1]) : 0;
    const int B = 8, T = 16128, Ti = T / 2, M = B * Ti, cin = 80, kcpad = 128;
    void* h = dalloc((size_t)M * 512, 1); void* ca = dalloc((size_t)M * cin * 2, 1);
    void* Wd = dalloc(512ull * 768 * 2, 1); void* Wc = dalloc(512ull * kcpad * 2, 1);
    float* bias = (float*)dalloc(512 * 4, 0); void* o = dalloc((size_t)M * 512, 1);
    unsigned long long* st; CK(hipMalloc(&st, 2 * 16 * 24 * 4 * 8)); CK(hipMemset(st, 0, 2 * 16 * 24 * 4 * 8));
    GateProb p{(const bf16*)h, (const bf16*)ca, nullptr, (const bf16*)Wd, (const bf16*)Wc, bias, (bf16*)o, M, Ti, 1, cin, kcpad};
    p.h8 = (const unsigned char*)h; p.Wd8 = (const unsigned char*)Wd; p.sb = 120;
    p.stamps = st;
    for (int it = 0; it < 5; ++it) {
        if (fp8) hipLaunchKernelGGL((gate_halo_kernel<256, 256, GateProb, true>), dim3(504), dim3(1024), 0, 0, p, 2);
        else hipLaunchKernelGGL((gate_halo_kernel<256, 256, GateProb, false>), dim3(504), dim3(1024), 0, 0, p, 2);
    }
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> v(2 * 16 * 24 * 4);
    CK(hipMemcpy(v.data(), st, v.size() * 8, hipMemcpyDeviceToHost));
    const int nsteps = fp8 ? 6 : 12;
    for (int wg = 0; wg < 2; ++wg) {
        printf("workgroup %d (%s): per conv step, cycles: [wait vmcnt] [barrier] [ds_read+mfma]   (mean over 16 waves; min..max of the mma part)\n", wg, fp8 ? "fp8" : "bf16");
        unsigned long long t00 = ~0ull;
        for (int w = 0; w < 16; ++w) t00 = std::min(t00, v[((wg * 16 + w) * 24 + 0) * 4 + 0]);
        for (int s = 0; s < nsteps; ++s) {
            double a = 0, b = 0, c = 0, cmin = 1e18, cmax = 0, start = 0;
            for (int w = 0; w < 16; ++w) {
                const unsigned long long* q = &v[((wg * 16 + w) * 24 + s) * 4];
                a += (double)(q[1] - q[0]); b += (double)(q[2] - q[1]); c += (double)(q[3] - q[2]);
                cmin = std::min(cmin, (double)(q[3] - q[2])); cmax = std::max(cmax, (double)(q[3] - q[2]));
                start += (double)(q[0] - t00);
            }
            printf("  step %2d  t=%7.0f  vmcnt %6.0f  barrier %6.0f  mma %6.0f (%5.0f..%5.0f)\n", s, start / 16, a / 16, b / 16, c / 16, cmin, cmax);
        }
        double tot = 0;
        for (int w = 0; w < 16; ++w) tot += (double)(v[((wg * 16 + w) * 24 + 20) * 4 + 0] - v[((wg * 16 + w) * 24 + 0) * 4 + 0]);
        printf("  K loop incl. conditioning steps: %.0f cycles (mean over waves)\n", tot / 16);
    }
    return 0;
}
