// Training-side primitives (work in progress): a generic
// multi-segment GEMM on the LDS-DMA ring core, used for the un-fused training forward of the tail,
// every data gradient (transposed packed weights, tap shifts with clip-edge masks, ReLU masks,
// residual adds) and every weight gradient (transposed activation copies, K = rows, split over
// workgroups into fp32 partials that a second pass sums in a fixed order).
#include "common.h"
#include "gemm_ring.h"
#include "fwn_internal.h"
#include "../../include/fwn.h"

struct LinProb {
    static constexpr bool A_DMA = true;
    static constexpr bool ALLOW_256 = false;
    const fwn_gemm_desc& g;   // the kernel argument itself (never copied: segments are indexed at run time)
    int q0, nq;               // this workgroup's chunk range (all chunks unless split)
    float* y32;               // fp32 output of this split
    struct RowCtx { int row, t; };
    struct ChunkCtx { int s, k0; };
    template <int BK> __device__ int nchunks() const { return nq; }
    __device__ RowCtx row_ctx(int row) const { return RowCtx{row, g.Ti > 0 ? row % g.Ti : row}; }
    template <int BK> __device__ ChunkCtx chunk_ctx(int q) const {
        int gq = q0 + q, s = 0;
        for (; s < g.nseg - 1; ++s) {
            const int cs = (g.seg[s].k + BK - 1) / BK;
            if (gq < cs) break;
            gq -= cs;
        }
        return ChunkCtx{s, gq * BK};
    }
    __device__ srd_t a_srd(const ChunkCtx& cc) const {
        return make_srd(g.seg[cc.s].x, (uint32_t)((size_t)g.seg[cc.s].rows * g.seg[cc.s].ld * 2));
    }
    __device__ uint32_t a_voff(const ChunkCtx& cc, const RowCtx& rc, int c8) const {
        const fwn_gemm_seg& sg = g.seg[cc.s];
        const int kk = cc.k0 + c8 * 8;
        bool ok = rc.row < g.M && kk < sg.k;
        if (g.Ti > 0) ok = ok && (unsigned)(rc.t + sg.shift) < (unsigned)g.Ti;
        else ok = ok && (unsigned)(rc.row + sg.shift) < (unsigned)sg.rows;
        return ok ? (uint32_t)((rc.row + sg.shift) * sg.ld + kk) * 2u : FWN_OOB;
    }
    __device__ srd_t b_srd(const ChunkCtx&) const { return make_srd(g.W, (uint32_t)((size_t)g.N * g.ldw * 2)); }
    __device__ uint32_t b_voff(const ChunkCtx& cc, int n, int c8) const {
        const fwn_gemm_seg& sg = g.seg[cc.s];
        const int kk = cc.k0 + c8 * 8;
        return (n < g.N && kk < sg.k) ? (uint32_t)(n * g.ldw + sg.koff + kk) * 2u : FWN_OOB;
    }
    __device__ float acc_init(int) const { return 0.0f; }
    template <int MI>
    __device__ void epilogue(const f32x16 (&acc)[MI][2], int mrow0, int ncol0, int lane) const {
        const int lr = lane & 31;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = ncol0 + ni * 32 + lr;
            if (col >= g.N) continue;
            const float b = g.bias ? g.bias[col] : 0.0f;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mrow0 + mi * 32 + acc_row_c(r) + 4 * (lane >> 5);
                    if (row >= g.M) continue;
                    float v = acc[mi][ni][r] + b;
                    if (g.R) v += g.rscale * (float)((const bf16*)g.R)[(size_t)row * g.ldr + col];
                    if (g.mask && !((float)((const bf16*)g.mask)[(size_t)row * g.ldmask + col] > 0.0f)) v = 0.0f;
                    if (g.relu) v = fmaxf(v, 0.0f);
                    v *= g.oscale;
                    if (g.out_f32) {
                        float* dst = y32 + (size_t)row * g.ldy + col;
                        *dst = g.accumulate ? *dst + v : v;
                    } else {
                        ((bf16*)g.Y)[(size_t)row * g.ldy + col] = (bf16)v;
                    }
                }
        }
    }
};

template <int BM, int BN, int WM, int WN, int D>
__global__ __launch_bounds__(64 * WM * WN) void lin_kernel(const fwn_gemm_desc g, int ntn, int nq_all) {
    const int per = (nq_all + g.nsplit - 1) / g.nsplit;
    const int q0 = (int)blockIdx.z * per;
    const int nq = max(0, min(per, nq_all - q0));      // an empty split still writes its zero partial
    const LinProb p{g, q0, nq, (float*)g.Y + (size_t)blockIdx.z * g.split_stride};
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    gemm_ring_body<BM, BN, WM, WN, 64, D, LinProb>(p, wg / ntn, wg % ntn);
}

int fwn_gemm_launch(const fwn_gemm_desc* g, hipStream_t st) {
    const fwn_gemm_desc& p = *g;
    int nq_all = 0;
    for (int s = 0; s < g->nseg; ++s) nq_all += (g->seg[s].k + 63) / 64;
    const int M = g->M, ns = g->nsplit;
    const int n128 = (g->N + 127) / 128;
    // the same fill rule as the inference GEMMs: the largest tile that still gives about a workgroup per CU
    if (((M + 255) / 256) * n128 * ns >= 192)
        hipLaunchKernelGGL((lin_kernel<256, 128, 8, 2, 3>), dim3(((M + 255) / 256) * n128, 1, ns), dim3(1024), 0, st, p, n128, nq_all);
    else if (((M + 127) / 128) * n128 * ns >= 192)
        hipLaunchKernelGGL((lin_kernel<128, 128, 4, 2, 3>), dim3(((M + 127) / 128) * n128, 1, ns), dim3(512), 0, st, p, n128, nq_all);
    else
        hipLaunchKernelGGL((lin_kernel<64, 128, 2, 2, 4>), dim3(((M + 63) / 64) * n128, 1, ns), dim3(256), 0, st, p, n128, nq_all);
    return 0;
}

// ---- dst[c][m] = valid(m) ? src[m + shift][c] : 0, zero padded to ldd columns; row C of dst (if
// ones_row) is 1 for m < M - so the weight-gradient GEMM over the transposed copy also yields the
// bias gradient (column sums) in its extra output row.
__global__ __launch_bounds__(256) void transpose_shift_kernel(const bf16* __restrict__ src, int M, int C, int lds_,
                                                              int shift, int Ti, bf16* __restrict__ dst, int ldd,
                                                              int ones_row) {
    __shared__ bf16 tile[64][66];
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int mm = i >> 6, cc = i & 63;
        const int m = m0 + mm, c = c0 + cc;
        bool ok = m < M && c < C;
        if (ok) ok = Ti > 0 ? (unsigned)(m % Ti + shift) < (unsigned)Ti : (unsigned)(m + shift) < (unsigned)M;
        tile[mm][cc] = ok ? src[(size_t)(m + shift) * lds_ + c] : (bf16)0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int cc = i >> 6, mm = i & 63;
        const int m = m0 + mm, c = c0 + cc;
        if (c < C && m < ldd) dst[(size_t)c * ldd + m] = tile[mm][cc];
    }
    if (ones_row && blockIdx.y == 0 && threadIdx.x < 64) {
        const int m = m0 + threadIdx.x;
        if (m < ldd) dst[(size_t)C * ldd + m] = (bf16)(m < M ? 1.0f : 0.0f);
    }
}
void fwn_transpose_launch(const void* src, int M, int C, int ld_src, int shift, int Ti, void* dst, int ld_dst,
                          int ones_row, hipStream_t st) {
    hipLaunchKernelGGL(transpose_shift_kernel, dim3((ld_dst + 63) / 64, (C + 63) / 64), dim3(256), 0, st,
                       (const bf16*)src, M, C, ld_src, shift, Ti, (bf16*)dst, ld_dst, ones_row);
}

// ---- out[i] = scale * sum_s partial[s][i], fixed order (deterministic split-K second pass) -------
__global__ __launch_bounds__(256) void reduce_splits_kernel(const float* __restrict__ partial, int nsplit, long stride,
                                                            long n, float scale, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float a = 0.0f;
        for (int s = 0; s < nsplit; ++s) a += partial[(size_t)s * stride + i];
        out[i] = a * scale;
    }
}
void fwn_reduce_splits_launch(const float* partial, int nsplit, long stride, long n, float scale, float* out,
                              hipStream_t st) {
    long nb = (n + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)nb), dim3(256), 0, st, partial, nsplit, stride, n, scale, out);
}
