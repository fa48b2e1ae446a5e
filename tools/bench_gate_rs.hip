// Developer harness (GPU box): the register-streamed gate tile (csrc/gate_rs.h) against the tap-sharing tile
// (csrc/gate_halo.h) on the same random operands: bit comparison of the outputs, then interleaved timing rounds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize tools/bench_gate_rs.hip -o /tmp/bench_gate_rs
//   /tmp/bench_gate_rs [B] [first block] [last block] [dil]
#include "../tf-flowavenet_amd/csrc/flow_kernels.hip"
#include "../tf-flowavenet_amd/csrc/gate_rs.hip"
#include "../tf-flowavenet_amd/csrc/tail_rs.hip"      // (flow_kernels.hip's tail dispatch refers to it)
#include "gate_co.h"       // the co-resident gate left the library in round 5 (tools/gate_co.h)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define RS_NW 8
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
    const int grid_max = argc > 5 ? atoi(argv[5]) : 1 << 30;     // e.g. 256: the experimental persistent form (one workgroup per CU)
    const int T = 16128;
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
        GateProb p{(const bf16*)h, (const bf16*)ca, nullptr, (const bf16*)Wd, (const bf16*)Wc, bias, (bf16*)o_ref, M, Ti, dil, cin, kcpad};
        GateRsArgs a{(const bf16*)h, (const bf16*)ca, (const bf16*)Wg, bias, (bf16*)o_new, M, Ti, dil, cin};
        const int t256 = (M + 255) / 256;
#ifdef FWN_RS_STAMP
        unsigned long long* stamps; CK(hipMalloc(&stamps, (size_t)1024 * RS_NW * 32 * 8)); CK(hipMemset(stamps, 0, (size_t)1024 * RS_NW * 32 * 8));
        a.stamps = stamps;
#endif
        const bool big = t256 * 2 >= 192;         // the library's choice: 256 x 256 tap-sharing tile / 256-row stream tile
        const int t128 = (M + 127) / 128;
        auto run_ref = [&] {
            if (big) hipLaunchKernelGGL((gate_halo_kernel<256, 256, GateProb>), dim3(t256 * 2), dim3(1024), 0, 0, p, 2);
            else hipLaunchKernelGGL((gate_halo_kernel<256, 128, GateProb>), dim3(t256 * 4), dim3(1024), 0, 0, p, 4);
        };
        auto run_rs = [&] {
            const int nt = big ? t256 * 2 : t128 * 2, grid = nt < grid_max ? nt : grid_max;
            // fewer workgroups than tiles: the persistent form (experimental); else one tile per workgroup (the product)
#define RS_GO(n, mt) do { if (grid < nt) hipLaunchKernelGGL((gate_rs_kernel<n, mt, true>), dim3(grid), dim3(512), 0, 0, a, nt); \
                          else hipLaunchKernelGGL((gate_rs_kernel<n, mt, false>), dim3(grid), dim3(512), 0, 0, a, nt); } while (0)
            if (big) {
                if (nkc == 5) RS_GO(5, 8); else if (nkc == 10) RS_GO(10, 8); else RS_GO(20, 8);
            } else {
                if (nkc == 5) RS_GO(5, 4); else if (nkc == 10) RS_GO(10, 4); else RS_GO(20, 4);
            }
        };
        auto run_co = [&] {
            const int grid = t256 * 4;
            if (nkc == 5) hipLaunchKernelGGL((gate_co_kernel<5>), dim3(grid), dim3(256), 0, 0, a);
            else if (nkc == 10) hipLaunchKernelGGL((gate_co_kernel<10>), dim3(grid), dim3(256), 0, 0, a);
            else hipLaunchKernelGGL((gate_co_kernel<20>), dim3(grid), dim3(256), 0, 0, a);
        };
        const bool use_co = getenv("RS_CO") != nullptr;     // the co-resident form (csrc/gate_co.h) as the "new" contender
        auto run_new = [&] { if (use_co) run_co(); else run_rs(); };
        if (getenv("RS_DEBUG")) printf("h %p ca %p Wd %p Wc %p Wg %p bias %p o_ref %p o_new %p\n", h, ca, Wd, Wc, Wg, (void*)bias, o_ref, o_new);
        if (!getenv("RS_SKIP_REF")) run_ref();
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
        if (getenv("RS_CONTEND")) {
            // contention soak: the kernel beside a bandwidth hog on a second stream, every output against the quiet run
            hipStream_t s2; CK(hipStreamCreate(&s2));
            void *big1, *big2; CK(hipMalloc(&big1, 1u << 30)); CK(hipMalloc(&big2, 1u << 30));
            std::vector<unsigned short> quiet((size_t)M * 256), cur((size_t)M * 256);
            run_new(); CK(hipDeviceSynchronize());
            CK(hipMemcpy(quiet.data(), o_new, quiet.size() * 2, hipMemcpyDeviceToHost));
            int badruns = 0;
            for (int it = 0; it < 40; ++it) {
                if (atoi(getenv("RS_CONTEND")) == 2) {     // the same kernel on a second stream (its own output): two launches share the chip
                    GateRsArgs a2 = a; a2.o = (bf16*)big2;
                    const int nt2 = big ? t256 * 2 : t128 * 2, grid2 = nt2 < grid_max ? nt2 : grid_max;
                    for (int k = 0; k < 6; ++k) {
                        if (big && nkc == 5 && grid2 < nt2) hipLaunchKernelGGL((gate_rs_kernel<5, 8, true>), dim3(grid2), dim3(512), 0, s2, a2, nt2);
                        else if (big && nkc == 5) hipLaunchKernelGGL((gate_rs_kernel<5, 8, false>), dim3(grid2), dim3(512), 0, s2, a2, nt2);
                    }
                } else
                for (int k = 0; k < 4; ++k) CK(hipMemcpyAsync(big2, big1, 1u << 30, hipMemcpyDeviceToDevice, s2));
                CK(hipMemsetAsync(o_new, 0xee, (size_t)M * 512, 0));
                run_new();
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(cur.data(), o_new, cur.size() * 2, hipMemcpyDeviceToHost));
                size_t nb = 0, firstb = 0; int rmin = 1 << 30, rmax = -1, cmin = 999, cmax = -1;
                for (size_t i = 0; i < cur.size(); ++i) if (cur[i] != quiet[i]) { if (!nb) firstb = i; ++nb; int r_ = (int)(i / 256), c_ = (int)(i % 256); rmin = std::min(rmin, r_); rmax = std::max(rmax, r_); cmin = std::min(cmin, c_); cmax = std::max(cmax, c_); }
                if (nb) { ++badruns; if (badruns <= 8) printf("  contended run %d: %zu outputs differ; rows %d..%d (tiles %d..%d), channels %d..%d; first row %zu ch %zu\n", it, nb, rmin, rmax, rmin / 256, rmax / 256, cmin, cmax, firstb / 256, firstb % 256); }
            }
            printf("  contention soak: %d of 40 runs differ from the quiet run\n", badruns);
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
        printf("  tap-sharing  : median %.2f us (min %.2f)  %.1f TFLOP/s\n", tr[3], tr[0], flops / (tr[3] * 1e-6) / 1e12);
        printf("  reg-streamed : median %.2f us (min %.2f)  %.1f TFLOP/s\n", tn[3], tn[0], flops / (tn[3] * 1e-6) / 1e12);
#if defined(FWN_RS_CHECK)
        {
            CK(hipMemset(stamps, 0, 4096));
            run_new();
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> st(8 + 4 * 60);
            CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
            printf("  schedule check: %llu waits with fewer younger operations than the static count\n", st[0] & 0xffffffffull);
            for (unsigned i = 0; i < std::min<unsigned long long>(st[0] & 0xffffffffull, 60); ++i)
                printf("    wave %llu tile/g %llu: actual %llu static %llu\n", st[8 + 4 * i], st[9 + 4 * i], st[10 + 4 * i], st[11 + 4 * i]);
        }
#elif defined(FWN_RS_STAMP)
        {
            std::vector<unsigned long long> st((size_t)1024 * RS_NW * 32);
            CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
            // earliest start over all waves = time zero; print a few workgroups' waves: prologue, items, K loop end, epilogue end
            unsigned long long t0 = ~0ull; for (size_t w = 0; w < (size_t)256 * RS_NW; ++w) if (st[w * 32]) t0 = std::min(t0, st[w * 32]);
            for (int wgi : {0, 1, 100, 251, 255}) {
                for (int w = 0; w < RS_NW; w += RS_NW - 1) {
                    const unsigned long long* q = &st[((size_t)wgi * RS_NW + w) * 32];
                    const double clk = (double)(q[21] - q[0]) / ((double)(q[31] - q[30]) * 10.0);   // cycles per ns: memrealtime ticks at 100 MHz
                    printf("  wg %3d wave %d: start %7llu | item waits/barriers:", wgi, w, q[0] - t0);
                    for (int i = 0; i < 7; ++i) if (q[1 + 2 * i]) printf(" [%llu +%llu]", q[1 + 2 * i] - q[0], q[2 + 2 * i] - q[1 + 2 * i]);
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
