// Yardstick (task 2): the programming guide's 256 x 256 "8-phase" bf16 GEMM template (cdna_hip_programming.md
// section 5: 8 waves as 2 (M) x 4 (N), wave tile 128 x 64, BK = 64, v_mfma_f32_16x16x32_bf16, 128 KB of LDS = two K-tiles x
// four half-tiles, global_load ... lds staging with COUNTED vmcnt across raw s_barriers, the st_16x32 swizzle, s_setprio around
// the 16-MFMA clusters, the two wave groups one barrier apart) written out from the guide's text - its example source is not in
// this image - as ONE file with no product dependency, and run on random operands at 4096^3 / 8192^3 and at the gate's own shape
// (M = 64 512, N = 512, K = 832 / 896 / 4096), next to tools/bench_gemm_yardstick.hip (the repo's ring core on the same shapes).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bench_gemm_8phase.hip -o /tmp/gemm8 && /tmp/gemm8
//
// C[M][N] (bf16) = A[M][K] . B[N][K]^T, both operands K-contiguous ("B^T input"), fp32 accumulation.  M, N multiples of 256,
// K a multiple of 64.
//
// Schedule of one K-tile t (LDS buffer t & 1), four phases; a wave reads only ITS A half (rows of wave row wr) and ITS B half
// (wave columns wc >> 1):
//   phase 1: ds_read all 8 B fragments + the 8 A fragments of rows 0-63   | stage A0(t+1) | MFMA quadrant (rows 0-63, cols 0-31)
//   phase 2:                                                               | stage A1(t+1) | MFMA (rows 0-63,  cols 32-63)
//   phase 3: ds_read the 8 A fragments of rows 64-127                      | stage B0(t+2) | MFMA (rows 64-127, cols 32-63)
//   phase 4:                                          vmcnt(4)             | stage B1(t+2) | MFMA (rows 64-127, cols 0-31)
// Every phase: [ds_reads] [2 LDS-DMA per thread] s_barrier, lgkmcnt(0), setprio 1, 16 MFMAs, setprio 0, s_barrier.  Waves 4-7
// (wave row 1: the SIMD partners of waves 0-3) run ONE BARRIER behind, so one group's MFMA cluster sits beside the other's
// ds_reads and DMA issue.  WAR: a half is restaged two phases after its last ds_read (B: read in phase 1, restaged from phase
// 3; A: read in phase 3, restaged from the next tile's phase 1).  RAW: the one counted wait per K-tile (phase 4: everything of
// tile t+1 has landed, the two B halves of tile t+2 stay in flight) comes a phase before the first read of what it retires.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define OOB 0x80000000u

// 64 lanes x 16 B -> LDS at lds_addr + lane * 16, from inline asm: invisible to hipcc's wait insertion (every wait is counted here)
__device__ __forceinline__ void dma16(u32x4 srd, uint32_t voff, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ u32x4 make_srd(const void* p, uint32_t bytes) {
    const unsigned long long b = (unsigned long long)(uintptr_t)p;
    return u32x4{(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) & 0xffffu,
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}
// st_16x32: inside a 1-KB subtile (16 rows x 32 bf16) rows 8-15 have their two 32-byte halves swapped
__device__ __forceinline__ uint32_t swz(uint32_t b) { return b ^ (((b >> 9) & 1u) << 5); }
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return ((x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

#ifndef G8_SETPRIO
#define G8_SETPRIO 1
#endif
#ifndef G8_LATEWAIT
#define G8_LATEWAIT 1
#endif

// clk (may be null): [workgroup][4] = s_memtime at start / end of wave 0, s_memrealtime (100 MHz) at start / end - the in-kernel
// clock and the cycles a workgroup lives, in a buffer nothing else reads
__global__ __launch_bounds__(512, 2) void gemm8_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, bf16* __restrict__ C, int M, int N, int K,
                                                       unsigned long long* clk) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[131072];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int ntn = N / 256;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (wg / ntn) * 256, n0 = (wg % ntn) * 256;
    const int nt = K / 64;
    unsigned long long c0 = 0, r0 = 0;
    if (clk) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds);

    // LDS: buffer b at b * 65536: A half h at + h * 16384, B half h at + 32768 + h * 16384; a half = 16 subtiles of 1 KB:
    // subtile s = (16-row group s >> 1, k half s & 1)
    const u32x4 srdA = make_srd(A, (uint32_t)((size_t)M * K * 2)), srdB = make_srd(B, (uint32_t)((size_t)N * K * 2));
    // this lane's place inside a staged subtile (the swizzle on the SOURCE side, the LDS destination stays lane-linear)
    const uint32_t sb = swz((uint32_t)lane * 16u);
    const uint32_t srow = sb >> 6, scol = sb & 63u;          // row in the subtile, byte column
    auto stage = [&](bool isB, int h, int t) {
        const u32x4 srd = isB ? srdB : srdA;
        const int row0 = (isB ? n0 : m0) + h * 128;
        const uint32_t dst = lds0 + (uint32_t)((t & 1) * 65536 + (isB ? 32768 : 0) + h * 16384);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int s = wave + 8 * j;
            const uint32_t off = (uint32_t)((row0 + (s >> 1) * 16 + (int)srow) * K + t * 64 + (s & 1) * 32) * 2u + scol;
            dma16(srd, t < nt ? off : OOB, dst + (uint32_t)s * 1024u);
        }
    };
    // fragment reads: lane l holds row l & 15, k = 8 (l >> 4) .. + 7 of a 16 x 32 subtile
    const uint32_t lofs = swz((uint32_t)((lane & 15) * 64 + (lane >> 4) * 16));
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[4][2], bfr[4][2];
    auto read_a = [&](int t, int qm) {
        const unsigned char* base = lds + (t & 1) * 65536 + wr * 16384 + lofs;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[mi][ks] = *(const bf16x8*)(base + ((qm * 4 + mi) * 2 + ks) * 1024);
    };
    auto read_b = [&](int t) {
        const unsigned char* base = lds + (t & 1) * 65536 + 32768 + (wc >> 1) * 16384 + lofs;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bfr[ni][ks] = *(const bf16x8*)(base + (((wc & 1) * 4 + ni) * 2 + ks) * 1024);
    };
    // one quadrant: rows 64 qm .. + 63, columns 32 qn .. + 31 of the wave tile; D = B-fragment x A-fragment, i.e. the C tile
    // TRANSPOSED in the accumulator (rows = n, columns = m): a lane ends up with 4 consecutive n of one m -> 8-byte stores
    auto mfma_q = [&](int qm, int qn) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int nj = 0; nj < 2; ++nj)
                    acc[qm * 4 + mi][qn * 2 + nj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[qn * 2 + nj][ks], af[mi][ks], acc[qm * 4 + mi][qn * 2 + nj], 0, 0, 0);
    };
#define BAR() __builtin_amdgcn_s_barrier()
#define LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define PRIO(x) do { if (G8_SETPRIO) __builtin_amdgcn_s_setprio(x); } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)

    // prologue: B(0), A(0), B(1) - then everything but B(1) has landed
    stage(true, 0, 0); stage(true, 1, 0); stage(false, 0, 0); stage(false, 1, 0); stage(true, 0, 1); stage(true, 1, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    BAR();
    if (wr == 1) BAR();          // the second wave row runs one barrier behind from here on
    for (int t = 0; t < nt; ++t) {
        // ---- phase 1
        read_b(t); SB(); read_a(t, 0); SB();
        stage(false, 0, t + 1);
        BAR(); LGKM0(); SB();
        PRIO(1); mfma_q(0, 0); PRIO(0); SB();
        BAR();
        // ---- phase 2
        stage(false, 1, t + 1);
        BAR(); SB();
        PRIO(1); mfma_q(0, 1); PRIO(0); SB();
        BAR();
        // ---- phase 3
        read_a(t, 1); SB();
        stage(true, 0, t + 2);
        BAR(); LGKM0(); SB();
        PRIO(1); mfma_q(1, 1); PRIO(0); SB();
        BAR();
        // ---- phase 4: tile t + 1 complete (its A halves were issued in phases 1 / 2; B(t + 2) stays in flight)
        stage(true, 1, t + 2);
        // The wait must precede the barrier EVENT after which the first wave row reads tile t + 1: that event is this phase's
        // SECOND barrier of row 0 = its FIRST barrier of row 1 (one barrier behind) - row 0 waits behind its MFMA cluster
        // (G8_LATEWAIT), which gives the youngest half-tile of tile t + 1 another half phase to land
        if (!G8_LATEWAIT || wr == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        BAR(); SB();
        PRIO(1); mfma_q(1, 0); PRIO(0); SB();
        if (G8_LATEWAIT && wr == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        BAR();
    }
    if (wr == 0) BAR();          // pair the second row's last barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- epilogue: lane (l & 15 = m in its 16-row tile, 4 (l >> 4) + r = n in its 16-column tile): 4 consecutive bf16 of one row
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const size_t row = (size_t)(m0 + wr * 128 + mi * 16 + (lane & 15));
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const f32x4 v = acc[mi][ni];
            const uint2 o = make_uint2(__builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2)),
                                       __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2)));
            *(uint2*)(C + row * N + n0 + wc * 64 + ni * 16 + (lane >> 4) * 4) = o;
        }
    }
    if (clk && tid == 0) {
        unsigned long long* q = clk + (size_t)blockIdx.x * 4;
        q[0] = c0; q[1] = __builtin_amdgcn_s_memtime(); q[2] = r0; q[3] = __builtin_amdgcn_s_memrealtime();
    }
}

static unsigned short f2bf(float v) { uint32_t u; memcpy(&u, &v, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
static float bf2f(unsigned short h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
    struct Shape { int M, N, K; const char* what; };
    std::vector<Shape> shapes = {{4096, 4096, 4096, "4096^3"}, {8192, 8192, 8192, "8192^3"},
                                 {64512, 512, 832, "gate rows x 512 x K 832 (768 + 64)"}, {64512, 512, 896, "gate rows x 512 x K 896"},
                                 {64512, 512, 4096, "gate rows x 512 x K 4096"}, {32256, 512, 960, "block-1 gate (K 768 + 160 -> 960)"}};
    if (argc > 3) shapes = {{atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), "argv"}};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Shape& s : shapes) {
        const size_t na = (size_t)s.M * s.K, nb = (size_t)s.N * s.K, nc = (size_t)s.M * s.N;
        std::vector<unsigned short> ha(na), hb(nb);
        srand(1234);
        for (auto& v : ha) v = f2bf((rand() & 0xffff) / 32768.0f - 1.0f);        // uniform [-1, 1): the guide's rule 25
        for (auto& v : hb) v = f2bf((rand() & 0xffff) / 32768.0f - 1.0f);
        void *dA, *dB, *dC;
        CK(hipMalloc(&dA, na * 2)); CK(hipMalloc(&dB, nb * 2)); CK(hipMalloc(&dC, nc * 2));
        CK(hipMemcpy(dA, ha.data(), na * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hb.data(), nb * 2, hipMemcpyHostToDevice));
        CK(hipMemset(dC, 0xff, nc * 2));
        const int grid = (s.M / 256) * (s.N / 256);
        unsigned long long* dclk; CK(hipMalloc(&dclk, (size_t)grid * 32)); CK(hipMemset(dclk, 0, (size_t)grid * 32));
        unsigned long long* clkarg = nullptr;
        auto run = [&] { hipLaunchKernelGGL(gemm8_kernel, dim3(grid), dim3(512), 0, 0, (const bf16*)dA, (const bf16*)dB, (bf16*)dC, s.M, s.N, s.K, clkarg); };
        run(); CK(hipDeviceSynchronize());
        // check 4096 sampled outputs against fp64 dot products of the bf16 operands (asymmetric random B: a transposed or
        // permuted store cannot pass), plus one whole 256 x 256 tile's corner rows
        std::vector<unsigned short> hc(nc);
        CK(hipMemcpy(hc.data(), dC, nc * 2, hipMemcpyDeviceToHost));
        double maxrel = 0; int bad = 0;
        for (int i = 0; i < 4096; ++i) {
            const int m = (int)(((unsigned)rand() * 2654435761u) % (unsigned)s.M), n = (int)(((unsigned)rand() * 40503u) % (unsigned)s.N);
            double ref = 0;
            for (int k = 0; k < s.K; ++k) ref += (double)bf2f(ha[(size_t)m * s.K + k]) * (double)bf2f(hb[(size_t)n * s.K + k]);
            const double got = bf2f(hc[(size_t)m * s.N + n]);
            const double err = fabs(got - ref) / (fabs(ref) + sqrt((double)s.K) * 0.05);
            maxrel = std::max(maxrel, err);
            if (err > 2e-2) ++bad;
        }
        std::vector<float> ts;
        for (int round = 0; round < 5; ++round) {
            const int it = s.M * (size_t)s.N * s.K > (1ull << 36) ? 5 : 20;
            run(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < it; ++i) run();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ts.push_back(ms * 1e3f / it);
        }
        std::sort(ts.begin(), ts.end());
        // one stamped launch behind the timed ones (the chip is warm): in-kernel clock and cycles per workgroup / per K-tile
        clkarg = dclk; run(); CK(hipDeviceSynchronize()); clkarg = nullptr;
        std::vector<unsigned long long> hk((size_t)grid * 4);
        CK(hipMemcpy(hk.data(), dclk, hk.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> ghz, cyc;
        for (int g = 0; g < grid; ++g) if (hk[4 * g + 3] > hk[4 * g + 2]) { ghz.push_back((double)(hk[4 * g + 1] - hk[4 * g]) / (double)(hk[4 * g + 3] - hk[4 * g + 2]) * 0.1); cyc.push_back((double)(hk[4 * g + 1] - hk[4 * g])); }
        std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
        const double g_med = ghz.empty() ? 0 : ghz[ghz.size() / 2], c_med = cyc.empty() ? 0 : cyc[cyc.size() / 2];
        const double fl = 2.0 * s.M * s.N * s.K;
        printf("%-44s M %6d N %5d K %5d  %4d workgroups: median %9.2f us (min %9.2f)  %7.1f TFLOP/s (best %7.1f)   clock %.2f GHz, %.0f cycles per workgroup = %.0f per K-tile (2048 = the matrix pipe alone)   check: %d of 4096 sampled outputs off (max scaled err %.2e)\n",
               s.what, s.M, s.N, s.K, grid, ts[2], ts[0], fl / ts[2] / 1e6, fl / ts[0] / 1e6, g_med, c_med, c_med / (s.K / 64), bad, maxrel);
        CK(hipFree(dclk));
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
    }
    return 0;
}
