// Developer harness (GPU box): the co-resident gate (csrc/gate_co.h: 4-wave workgroups, two per CU) against the
// register-streamed gate (csrc/gate_rs.h) on the same random operands and the same fragment stream: comparison of the
// outputs (accumulation orders differ: one-ulp ties), then interleaved timing rounds.  Compiles in seconds (only the two
// gate headers), unlike tools/bench_gate_rs.hip which carries every flow kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Itools tools/bench_gate_co.hip -o tools/gate_co_bin
//   tools/gate_co_bin [B] [first block] [last block] [dil]        (-DFWN_RS_STAMP: per-item s_memtime stamps)
#include "gate_co.h"
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

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8;
    const int blk0 = argc > 2 ? atoi(argv[2]) : 0, blk1 = argc > 3 ? atoi(argv[3]) : 1;
    const int dil = argc > 4 ? atoi(argv[4]) : 1;
    const int T = argc > 5 ? atoi(argv[5]) : 16128;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blk = blk0; blk <= blk1; ++blk) {
        const int Ch = 1 << blk, Ti = T / (2 * Ch), M = B * Ti, cin = 40 * 2 * Ch, kcpad = (cin + 63) / 64 * 64;
        void* h = dalloc_bf16((size_t)M * 256, 1.0f);
        void* ca = dalloc_bf16((size_t)M * cin, 1.0f);
        void* Wd = dalloc_bf16(512ull * 768, 0.05f);
        void* Wc = dalloc_bf16(512ull * kcpad, 0.05f);
        std::vector<float> hb(512);
        for (auto& v : hb) v = ((rand() & 0xffff) / 32768.0f - 1.0f) * 0.5f;
        float* bias; CK(hipMalloc(&bias, 2048)); CK(hipMemcpy(bias, hb.data(), 2048, hipMemcpyHostToDevice));
        void *o_ref, *o_new, *Wg;
        CK(hipMalloc(&o_ref, (size_t)M * 512)); CK(hipMalloc(&o_new, (size_t)M * 512));
        CK(hipMemset(o_ref, 0xff, (size_t)M * 512)); CK(hipMemset(o_new, 0xee, (size_t)M * 512));
        const int nkc = (cin + 15) / 16;
        CK(hipMalloc(&Wg, 16ull * (48 + nkc) * 1024));
        if (nkc == 5) hipLaunchKernelGGL(gate_stream_pack_kernel<5>, dim3(256), dim3(256), 0, 0, (const bf16*)Wd, (const bf16*)Wc, kcpad, (bf16*)Wg);
        else if (nkc == 10) hipLaunchKernelGGL(gate_stream_pack_kernel<10>, dim3(256), dim3(256), 0, 0, (const bf16*)Wd, (const bf16*)Wc, kcpad, (bf16*)Wg);
        else if (nkc == 20) hipLaunchKernelGGL(gate_stream_pack_kernel<20>, dim3(256), dim3(256), 0, 0, (const bf16*)Wd, (const bf16*)Wc, kcpad, (bf16*)Wg);
        else { printf("no instantiation for cin %d\n", cin); return 1; }
        GateRsArgs ar{(const bf16*)h, (const bf16*)ca, (const bf16*)Wg, bias, (bf16*)o_ref, M, Ti, dil, cin};
        GateRsArgs a{(const bf16*)h, (const bf16*)ca, (const bf16*)Wg, bias, (bf16*)o_new, M, Ti, dil, cin};
        const int t256 = (M + 255) / 256;
#ifdef FWN_RS_STAMP
        unsigned long long* stamps; CK(hipMalloc(&stamps, (size_t)2048 * 8 * 32 * 8)); CK(hipMemset(stamps, 0, (size_t)2048 * 8 * 32 * 8));
        a.stamps = stamps; ar.stamps = stamps + (size_t)1024 * 8 * 32;
#endif
        auto run_ref = [&] {
            const int nt = t256 * 2;
            if (nkc == 5) hipLaunchKernelGGL((gate_rs_kernel<5, 8, false>), dim3(nt), dim3(512), 0, 0, ar, nt);
            else if (nkc == 10) hipLaunchKernelGGL((gate_rs_kernel<10, 8, false>), dim3(nt), dim3(512), 0, 0, ar, nt);
            else hipLaunchKernelGGL((gate_rs_kernel<20, 8, false>), dim3(nt), dim3(512), 0, 0, ar, nt);
        };
        auto run_new = [&] {
            const int grid = t256 * 4;
            if (nkc == 5) hipLaunchKernelGGL((gate_co_kernel<5>), dim3(grid), dim3(256), 0, 0, a);
            else if (nkc == 10) hipLaunchKernelGGL((gate_co_kernel<10>), dim3(grid), dim3(256), 0, 0, a);
            else hipLaunchKernelGGL((gate_co_kernel<20>), dim3(grid), dim3(256), 0, 0, a);
        };
        run_ref();
        CK(hipDeviceSynchronize());
        run_new();
        CK(hipDeviceSynchronize());
        std::vector<unsigned short> r((size_t)M * 256), n((size_t)M * 256);
        CK(hipMemcpy(r.data(), o_ref, r.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(n.data(), o_new, n.size() * 2, hipMemcpyDeviceToHost));
        size_t bad = 0, bad2 = 0, first = (size_t)-1; double maxd = 0;
        for (size_t i = 0; i < r.size(); ++i) {
            if (r[i] != n[i]) {
                const int du = abs((int)(short)r[i] - (int)(short)n[i]);
                if (du > 1) { ++bad2; if (first == (size_t)-1) first = i; }
                ++bad;
                uint32_t ua = (uint32_t)r[i] << 16, ub = (uint32_t)n[i] << 16; float fa, fb;
                memcpy(&fa, &ua, 4); memcpy(&fb, &ub, 4);
                maxd = std::max(maxd, (double)fabsf(fa - fb));
            }
        }
        printf("block %d  M=%d K=%d dil=%d: %zu of %zu outputs differ, %zu by more than one bf16 ulp (max |d| %.3g)", blk, M, 768 + cin, dil, bad, r.size(), bad2, maxd);
        if (bad2) printf(", first at row %zu ch %zu: ref %04x new %04x", first / 256, first % 256, r[first], n[first]);
        printf("\n");
        // repeatability soak: the co-resident kernel against its own first result
        {
            std::vector<unsigned short> cur((size_t)M * 256);
            int badruns = 0;
            for (int it = 0; it < 20; ++it) {
                CK(hipMemsetAsync(o_new, 0xee, (size_t)M * 512, 0));
                run_new(); run_ref();
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(cur.data(), o_new, cur.size() * 2, hipMemcpyDeviceToHost));
                if (memcmp(cur.data(), n.data(), cur.size() * 2)) ++badruns;
            }
            printf("  repeat soak: %d of 20 runs differ from the first\n", badruns);
        }
        const double flops = 2.0 * M * (768.0 + cin) * 512;
        std::vector<float> tr, tn;
        for (int round = 0; round < 7; ++round) {
            for (int which = 0; which < 2; ++which) {
                const int it = 20;
                if (which) run_new(); else run_ref();
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                for (int i = 0; i < it; ++i) { if (which) run_new(); else run_ref(); }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                (which ? tn : tr).push_back(ms * 1e3f / it);
            }
        }
        std::sort(tr.begin(), tr.end()); std::sort(tn.begin(), tn.end());
        printf("  reg-streamed : median %.2f us (min %.2f)  %.1f TFLOP/s\n", tr[3], tr[0], flops / (tr[3] * 1e-6) / 1e12);
        printf("  co-resident  : median %.2f us (min %.2f)  %.1f TFLOP/s\n", tn[3], tn[0], flops / (tn[3] * 1e-6) / 1e12);
#ifdef FWN_RS_STAMP
        {
            CK(hipMemset(stamps, 0, (size_t)2048 * 8 * 32 * 8));
            run_new();
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> st((size_t)1024 * 4 * 32);
            CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull; for (size_t w = 0; w < (size_t)1008 * 4; ++w) if (st[w * 32]) t0 = std::min(t0, st[w * 32]);
            for (int wgi : {0, 1, 8, 100, 500, 600, 1000}) {
                if (wgi >= t256 * 4) continue;
                for (int w = 0; w < 4; w += 3) {
                    const unsigned long long* q = &st[((size_t)wgi * 4 + w) * 32];
                    const double clk = (double)(q[21] - q[0]) / ((double)(q[31] - q[30]) * 10.0);
                    printf("  wg %4d wave %d: start %7llu | barrier arrivals:", wgi, w, q[0] - t0);
                    for (int i = 1; i < 20; ++i) if (q[i]) printf(" %llu", q[i] - q[0]);
                    printf(" | loop end %llu | epilogue +%llu | clock %.2f GHz\n", q[20] - q[0], q[21] - q[20], clk);
                }
            }
        }
#endif
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) printf("  HIP error: %s\n", hipGetErrorString(e));
        for (void* q : {h, ca, Wd, Wc, (void*)bias, o_ref, o_new, Wg}) CK(hipFree(q));
    }
    return 0;
}
