// Developer harness (GPU box): the register-streamed tail (csrc/tail_rs.h) alone on random operands - time per launch and,
// built with -DFWN_TRS_STAMP, where a wave's cycles go (clock stamps at the phase boundaries).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize [-DFWN_TRS_STAMP] tools/bench_tail_rs.hip -o /tmp/bench_tail_rs
//   /tmp/bench_tail_rs [rows M] [MT 1|2|4] [Ch] [front 0|1] [iters]
#include "../tf-flowavenet_amd/csrc/tail_rs.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static unsigned short rnd_bf16(float scale) {
    const float v = ((rand() & 0xffff) / 32768.0f - 1.0f) * scale;
    uint32_t u; memcpy(&u, &v, 4);
    return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
}
static void* dalloc_bf16(size_t n, float scale) {
    void* p; CK(hipMalloc(&p, n * 2));
    std::vector<unsigned short> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = rnd_bf16(scale);
    CK(hipMemcpy(p, h.data(), n * 2, hipMemcpyHostToDevice));
    return p;
}
static float* dalloc_f32(size_t n, float scale, float off = 0.0f) {
    float* p; CK(hipMalloc(&p, n * 4));
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = off + ((rand() & 0xffff) / 32768.0f - 1.0f) * scale;
    CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
    return p;
}

template <int MT, bool FRONT>
static void launch(const TailArgs& a, const void* Wts, int grid, unsigned long long* stamps) {
#ifdef FWN_TRS_STAMP
    hipLaunchKernelGGL((tail_rs_kernel<MT, FRONT, false>), dim3(grid), dim3(512), 0, 0, a, (const bf16*)Wts, stamps);
#else
    hipLaunchKernelGGL((tail_rs_kernel<MT, FRONT, false>), dim3(grid), dim3(512), 0, 0, a, (const bf16*)Wts);
#endif
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 32256;
    const int MT = argc > 2 ? atoi(argv[2]) : 4;
    const int Ch = argc > 3 ? atoi(argv[3]) : 2;
    const int front = argc > 4 ? atoi(argv[4]) : 1;
    const int iters = argc > 5 ? atoi(argv[5]) : 50;
    const int L = 2, NF = 6;                     // six flows' weights in rotation (cold-ish L2, as inside a pass)
    const int Ti = M / 8 > 0 ? M / 8 : M, kfn = (6 * Ch + 15) / 16 * 16;
    void* o = dalloc_bf16((size_t)L * M * 256, 0.8f);
    std::vector<void*> Ws(NF), Wf(NF), Wz(NF), Wts(NF), Wfn(NF);
    for (int f = 0; f < NF; ++f) {
        Ws[f] = dalloc_bf16(256ull * 512, 0.05f); Wf[f] = dalloc_bf16(256ull * 256, 0.06f); Wz[f] = dalloc_bf16(64ull * 256, 0.02f);
        Wfn[f] = dalloc_bf16(256ull * kfn, 0.3f);
        CK(hipMalloc(&Wts[f], 8ull * 48 * 1024));
        hipLaunchKernelGGL(tail_stream_pack_kernel, dim3(96), dim3(256), 0, 0, (const bf16*)Ws[f], (const bf16*)Wf[f], (bf16*)Wts[f]);
    }
    float* bs = dalloc_f32(256, 0.1f); float* bfin = dalloc_f32(256, 0.1f); float* bz = dalloc_f32(64, 0.01f); float* ez = dalloc_f32(64, 0.0f, 1.0f);
    float* an = dalloc_f32(8 * Ch, 0.1f, 1.0f); float* bfn = dalloc_f32(256, 0.1f); float* ann = dalloc_f32(8 * Ch, 0.1f, 1.0f);
    float* xa = dalloc_f32((size_t)M * Ch, 1.0f); float* xb = dalloc_f32((size_t)M * Ch, 1.0f);
    float* xo; CK(hipMalloc(&xo, (size_t)M * Ch * 4));
    void* h0; CK(hipMalloc(&h0, (size_t)M * 512));
    const int rows = 32 * MT - (front ? 2 : 0), grid = (M + rows - 1) / rows;
    float* partial; CK(hipMalloc(&partial, (size_t)grid * 4));
    unsigned long long* stamps = nullptr;
    CK(hipMalloc(&stamps, (size_t)grid * 8 * 16 * 8)); CK(hipMemset(stamps, 0, (size_t)grid * 8 * 16 * 8));
    auto args = [&](int f) {
        TailArgs a{(const bf16*)o, (const bf16*)Ws[f], bs, (const bf16*)Wf[f], bfin, (const bf16*)Wz[f], bz, ez, an, xa, xb, partial, (long)M * 256, L, M, Ch, 1, 0};
        a.xb_out = xo; a.S = nullptr;
        a.h0_next = front ? (bf16*)h0 : nullptr; a.Wfn = front ? (const bf16*)Wfn[f] : nullptr; a.bfn = front ? bfn : nullptr; a.an_next = front ? ann : nullptr;
        a.kfn = front ? kfn : 0; a.Ti = front ? Ti : 0; a.overlap = front ? 1 : 0;
        a.save_s = nullptr; a.save_u = nullptr; a.save_z = nullptr;
        return a;
    };
    auto run = [&](int f) {
        const TailArgs a = args(f);
        if (MT == 4) { if (front) launch<4, true>(a, Wts[f], grid, stamps); else launch<4, false>(a, Wts[f], grid, stamps); }
        else if (MT == 2) { if (front) launch<2, true>(a, Wts[f], grid, stamps); else launch<2, false>(a, Wts[f], grid, stamps); }
        else { if (front) launch<1, true>(a, Wts[f], grid, stamps); else launch<1, false>(a, Wts[f], grid, stamps); }
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 6; ++i) run(i % NF);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) run(i % NF);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double gf = 2.0 * M * (256.0 * L + 256 + 2 * Ch) * 256 / 1e9;
    printf("M %d  MT %d  Ch %d  front %d  grid %d: %.2f us per launch (back to back, %d launches), %.2f GFLOP -> %.0f TFLOP/s\n", M, MT, Ch, front, grid,
           ms * 1e3 / iters, iters, gf, gf / (ms / iters) );
#ifdef FWN_TRS_STAMP
    {
        std::vector<unsigned long long> h((size_t)grid * 8 * 16);
        CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        static const char* names[14] = {"entry", "prologue issued", "barrier 0 passed (first o slice + constants landed)", "phase 1 K loop done (32 k-steps)",
                                         "S parked + barrier", "phase 2 K loop done (16 k-steps)", "vmcnt(0)", "barrier (S reads done)", "U parked + barrier",
                                         "phase 3 MFMAs done (tile waves)", "epilogue issued", "barrier (partials / T image)", "front conv MFMAs done", "end"};
        for (int grp = 0; grp < 2; ++grp) {       // tile waves (0 .. MT-1) and the others
            printf("%s: median cycles since entry [median increment]\n", grp == 0 ? "waves 0 .. MT-1 (ZeroConv + coupling)" : "waves MT .. 7 (a plane)");
            double prev = 0;
            for (int i = 1; i < 14; ++i) {
                std::vector<double> v;
                for (int wg = 0; wg < grid; ++wg)
                    for (int w = 0; w < 8; ++w) {
                        if ((w < MT) != (grp == 0)) continue;
                        const unsigned long long* s = &h[((size_t)wg * 8 + w) * 16];
                        if (s[i] && s[0]) v.push_back((double)(s[i] - s[0]));
                    }
                if (v.empty()) continue;
                std::sort(v.begin(), v.end());
                const double med = v[v.size() / 2];
                printf("  %2d %-52s %8.0f  [%+7.0f]\n", i, names[i], med, med - prev);
                prev = med;
            }
        }
        std::vector<double> clk;
        for (int wg = 0; wg < grid; ++wg) {
            const unsigned long long* s = &h[((size_t)wg * 8) * 16];
            if (s[15] > s[14] && s[13] > s[0]) clk.push_back((double)(s[13] - s[0]) / (double)(s[15] - s[14]) * 0.1);
        }
        if (!clk.empty()) { std::sort(clk.begin(), clk.end()); printf("in-kernel clock (wave 0 of each workgroup, median): %.2f GHz\n", clk[clk.size() / 2]); }
    }
#endif
    return 0;
}
