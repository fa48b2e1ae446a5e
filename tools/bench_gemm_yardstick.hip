// Developer micro-benchmark (GPU box): the yardstick question of VERDICT r4 (weak 4 / task 6).
// A PLAIN bf16 GEMM C[M][N] = A[M][K] B[N][K]^T (no epilogue arithmetic, bf16 out) on this repo's LDS-DMA ring core
// (csrc/gemm_ring.h) in the geometry of the programming guide's 256 x 256 template - 8 waves as 2 x 4, 128 x 64 per wave - and
// in the 16-wave geometry the gate's tap-sharing tile uses, on uniform random operands, at
//   (a) 4096^3 and 8192^3: what the guide quotes its template at (1 320 - 1 340 TF / ~1 470 TF = 0.53 - 0.59 of 2.5 PF), and
//   (b) the gate's own shape, M = 64 512, N = 512, K = 832 and 896 (the gate's K = 848 = 768 + 80 sits between: 13.25 chunks of 64),
// so that "what does K = 848 cost a GEMM-shaped kernel on THIS box at THIS clock" has a number next to the gate's.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize tools/bench_gemm_yardstick.hip -o tools/gemm_yardstick_bin
#include "../tf-flowavenet_amd/csrc/common.h"
#include "../tf-flowavenet_amd/csrc/gemm_ring.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct PlainProb {
    const bf16* A;        // [M][K]
    const bf16* B;        // [N][K]
    bf16* C;              // [M][N]
    int M, N, K;
    struct RowCtx { int row; };
    struct ChunkCtx { int k0; };
    template <int BK> __device__ int nchunks() const { return K / BK; }
    __device__ RowCtx row_ctx(int row) const { return RowCtx{row}; }
    template <int BK> __device__ ChunkCtx chunk_ctx(int q) const { return ChunkCtx{q * BK}; }
    __device__ srd_t a_srd(const ChunkCtx&) const { return make_srd(A, (uint32_t)((size_t)M * K * 2)); }
    __device__ uint32_t a_voff(const ChunkCtx& cc, const RowCtx& rc, int c8) const {
        return rc.row < M ? (uint32_t)(rc.row * K + cc.k0 + c8 * 8) * 2u : FWN_OOB;
    }
    __device__ srd_t b_srd(const ChunkCtx&) const { return make_srd(B, (uint32_t)((size_t)N * K * 2)); }
    __device__ uint32_t b_voff(const ChunkCtx& cc, int n, int c8) const { return (uint32_t)(n * K + cc.k0 + c8 * 8) * 2u; }
    __device__ float acc_init(int) const { return 0.0f; }
    template <int MI>
    __device__ void epilogue(const f32x16 (&acc)[MI][2], int mrow0, int ncol0, int lane) const {
        const int lr = lane & 31;
        const srd_t so = make_srd(C, (uint32_t)((size_t)M * N * 2));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const uint32_t voff = (uint32_t)((mrow0 + 4 * (lane >> 5)) * N + ncol0 + ni * 32 + lr) * 2u;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    buf_store_bf16(so, voff, (uint32_t)((mi * 32 + acc_row_c(r)) * N * 2), acc[mi][ni][r]);
        }
    }
};

template <int BM, int BN, int WM, int WN, int BK, int D>
__global__ __launch_bounds__(64 * WM * WN) void plain_kernel(PlainProb p, int ntn) {
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    gemm_ring_body<BM, BN, WM, WN, BK, D, PlainProb>(p, wg / ntn, wg % ntn);
}

static void* dalloc_rand(size_t elems) {
    void* p; CK(hipMalloc(&p, elems * 2));
    std::vector<unsigned short> h(elems);
    for (size_t i = 0; i < elems; ++i) {                       // uniform in [-1, 1): bf16 of a float
        const float f = (float)rand() / (float)RAND_MAX * 2.0f - 1.0f;
        unsigned int u; memcpy(&u, &f, 4);
        h[i] = (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
    }
    CK(hipMemcpy(p, h.data(), elems * 2, hipMemcpyHostToDevice));
    return p;
}

static hipEvent_t e0, e1;
template <class F> static double timeit(F fn, int it) {
    for (int i = 0; i < 5; ++i) fn();
    CK(hipDeviceSynchronize());
    double best = 1e30, sum = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < it; ++i) fn();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / it;
        best = us < best ? us : best; sum += us;
    }
    CK(hipGetLastError());
    return sum / 5;
}

#define RUN(BM, BN, WM, WN, BK, D)                                                                                         \
    do {                                                                                                                   \
        const double us = timeit([&] { hipLaunchKernelGGL((plain_kernel<BM, BN, WM, WN, BK, D>), dim3(((M + BM - 1) / BM) * (N / BN)), \
                                                          dim3(64 * WM * WN), 0, 0, p, N / BN); }, it);                    \
        printf("  tile %3dx%3d  %2d waves (%dx%d)  BK %d  ring %d : %9.2f us  %7.1f TFLOP/s = %.3f of 2.5 PF\n", BM, BN, WM * WN, WM, WN, \
               BK, D, us, flops / us / 1e6, flops / us / 1e6 / 2500.0);                                                   \
    } while (0)

int main() {
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int shapes[][3] = {{4096, 4096, 4096}, {8192, 8192, 8192}, {64512, 512, 832}, {64512, 512, 896}, {64512, 512, 4096}};
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        void* A = dalloc_rand((size_t)M * K);
        void* B = dalloc_rand((size_t)N * K);
        void* Cc = dalloc_rand((size_t)M * N);
        PlainProb p{(const bf16*)A, (const bf16*)B, (bf16*)Cc, M, N, K};
        const double flops = 2.0 * M * N * K;
        const int it = flops > 5e11 ? 5 : 20;
        printf("M = %d, N = %d, K = %d (%.1f GFLOP), uniform random [-1, 1) bf16 operands\n", M, N, K, flops / 1e9);
        RUN(256, 256, 2, 4, 64, 2);
        RUN(256, 256, 4, 4, 64, 2);
        RUN(256, 128, 8, 2, 64, 3);
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(Cc));
    }
    return 0;
}
