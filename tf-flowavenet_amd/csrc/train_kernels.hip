// Training-side primitives (work in progress): a generic
// multi-segment GEMM on the LDS-DMA ring core, used for every data gradient (transposed packed weights, tap shifts with clip-edge masks, ReLU masks,
// residual adds) and every weight gradient (transposed activation copies, K = rows, split over
// workgroups into fp32 partials that a second pass sums in a fixed order).
#include "common.h"
#include "gemm_ring.h"
#include <string.h>
#include "fwn_internal.h"
#include "../../include/fwn.h"

struct LinProb {
    static constexpr bool A_DMA = true;
    static constexpr bool ALLOW_256 = false;
    const fwn_gemm_desc& g;   // the kernel argument itself (never copied: segments are indexed at run time)
    int q0, nq;               // this workgroup's chunk range (all chunks unless split)
    float* y32;               // fp32 output of this split
    struct RowCtx { int row, t; };
    // The segment's fields are read from the kernel arguments ONCE per chunk (the segment index is a run-time value:
    // looked up per DMA piece, every piece would wait for its own scalar loads).
    struct ChunkCtx { const void* x; int rows, ld, k, shift, koff, k0; };
    template <int BK> __device__ int nchunks() const { return nq; }
    __device__ RowCtx row_ctx(int row) const { return RowCtx{row, g.Ti > 0 ? row % g.Ti : row}; }
    template <int BK> __device__ ChunkCtx chunk_ctx(int q) const {
        int gq = q0 + q, s = 0;
        for (; s < g.nseg - 1; ++s) {
            const int cs = (g.seg[s].k + BK - 1) / BK;
            if (gq < cs) break;
            gq -= cs;
        }
        const fwn_gemm_seg sg = g.seg[s];
        return ChunkCtx{sg.x, sg.rows, sg.ld, sg.k, sg.shift, sg.koff, gq * BK};
    }
    __device__ srd_t a_srd(const ChunkCtx& cc) const { return make_srd(cc.x, (uint32_t)((size_t)cc.rows * cc.ld * 2)); }
    __device__ uint32_t a_voff(const ChunkCtx& cc, const RowCtx& rc, int c8) const {
        const int kk = cc.k0 + c8 * 8;
        bool ok = rc.row < g.M && kk < cc.k;
        if (g.Ti > 0) ok = ok && (unsigned)(rc.t + cc.shift) < (unsigned)g.Ti;
        else ok = ok && (unsigned)(rc.row + cc.shift) < (unsigned)cc.rows;
        return ok ? (uint32_t)((rc.row + cc.shift) * cc.ld + kk) * 2u : FWN_OOB;
    }
    __device__ srd_t b_srd(const ChunkCtx&) const { return make_srd(g.W, (uint32_t)((size_t)g.N * g.ldw * 2)); }
    __device__ uint32_t b_voff(const ChunkCtx& cc, int n, int c8) const {
        const int kk = cc.k0 + c8 * 8;
        return (n < g.N && kk < cc.k) ? (uint32_t)(n * g.ldw + cc.koff + kk) * 2u : FWN_OOB;
    }
    __device__ float acc_init(int) const { return 0.0f; }
    // ---- row-major epilogue (gemm_ring.h LDS_EPI): 8 consecutive columns of one row per lane, 16-byte loads / stores ----
    // With one column per lane the direct epilogue below issues up to 64 two- / four-byte VMEM instructions per lane and
    // operand (absent operands included: their zero-sized descriptors still cost the issue); the training GEMMs with 256-row
    // tiles spent more time there than in the K loop.  Same arithmetic in the same order: results are bit-identical.
    // Needs 8-column groups that are whole and 16-byte aligned.  The gate-derivative tiles (round 3) too: a lane's 8
    // columns take their tanh / sigmoid factors as two 16-byte loads and leave as two 16-byte stores (the accumulator
    // layout made the skip-gradient GEMM of block 0 a 49 us launch for 6.7 GFLOP).
    static constexpr bool LDS_EPI = true;
    __device__ bool rows_launch() const {
        const bool al = ((uintptr_t)g.Y & 15) == 0 && (!g.R || ((uintptr_t)g.R & 15) == 0) && (!g.mask || ((uintptr_t)g.mask & 15) == 0) &&
                        (!g.bias || ((uintptr_t)g.bias & 15) == 0) &&
                        (!g.gate_aux || ((((uintptr_t)g.gate_aux | (uintptr_t)g.gate_out) & 15) == 0 && !g.out_f32 && g.gate_col0 % 64 == 0));
        return al && g.N % 8 == 0 && g.ldy % 8 == 0 && (!g.R || g.ldr % 8 == 0) && (!g.mask || g.ldmask % 8 == 0) &&
               (!g.out_f32 || (size_t)g.split_stride % 4 == 0);
    }
    __device__ bool rows_tile(int) const { return true; }
    template <int MI>
    __device__ void epilogue_rows(const float* wt, int mrow0, int ncol0, int lane) const {
        // the 64 columns of a wave tile lie inside or outside the gated range together (gate_col0 is a multiple of 64); two
        // instances of the body so that the registers of either (fp32 accumulate | gate factors) do not add up
        const int c0 = ncol0 - g.gate_col0;
        if (g.gate_aux && c0 >= 0 && c0 < 256) rows_body<MI, true>(wt, mrow0, ncol0, lane);
        else rows_body<MI, false>(wt, mrow0, ncol0, lane);
    }
    template <int MI, bool GATED>
    __device__ __forceinline__ void rows_body(const float* wt, int mrow0, int ncol0, int lane) const {
        const int col = ncol0 + (lane & 7) * 8;
        const bool cok = col < g.N;                              // the whole group of 8 (N % 8 == 0)
        const int cc = cok ? col : 0;                            // clamped: loads stay inside, stores of such lanes are dropped
        float bb[8];
        if (g.bias) {
            const float4 b0 = *(const float4*)(g.bias + cc), b1 = *(const float4*)(g.bias + cc + 4);
            bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) bb[e] = 0.0f;
        }
        const bool relu_on = g.relu != 0, acc_on = !GATED && g.out_f32 && g.accumulate;
        constexpr bool gated = GATED;
        const int gc = ncol0 - g.gate_col0 + (lane & 7) * 8;     // column inside the gate's 256 channels
        const float mdef = g.mask ? 0.0f : 1.0f;
        const uint32_t ybytes = (uint32_t)((size_t)g.M * g.ldy * (g.out_f32 ? 4 : 2));
        const srd_t sY = make_srd(g.out_f32 ? (const void*)y32 : (const void*)g.Y, ybytes);
        constexpr int NIT = 4 * MI;
        Pack16 rv[NIT], mv[NIT], tfv[GATED ? NIT : 1], sgv[GATED ? NIT : 1];
        float4 ya[GATED ? 1 : NIT][2];
        int rowc[NIT];
        bool ok[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = mrow0 + it * 8 + (lane >> 3);
            ok[it] = cok && row < g.M;
            rowc[it] = row < g.M ? row : 0;
        }
        // one uniform branch per operand, not one per load: the loads inside issue back to back
        if (g.R) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) rv[it].u = *(const uint4*)((const bf16*)g.R + (size_t)rowc[it] * g.ldr + cc);
        } else {
#pragma unroll
            for (int it = 0; it < NIT; ++it) rv[it].u = zero16();
        }
        if (g.mask) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) mv[it].u = *(const uint4*)((const bf16*)g.mask + (size_t)rowc[it] * g.ldmask + cc);
        } else {
#pragma unroll
            for (int it = 0; it < NIT; ++it) mv[it].u = zero16();
        }
        if constexpr (GATED) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const bf16* ap = (const bf16*)g.gate_aux + (size_t)rowc[it] * 512 + (cok ? gc : 0);
                tfv[it].u = *(const uint4*)ap;
                sgv[it].u = *(const uint4*)(ap + 256);
            }
        } else {
            if (acc_on) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const float* yp = y32 + (size_t)rowc[it] * g.ldy + cc;
                    ya[it][0] = *(const float4*)yp;
                    ya[it][1] = *(const float4*)(yp + 4);
                }
            } else {
#pragma unroll
                for (int it = 0; it < NIT; ++it) ya[it][0] = ya[it][1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            float a[8], out[8];
            lds_epi_take(wt, it, lane, a);
            float yv[8];
            if constexpr (GATED) {
#pragma unroll
                for (int e = 0; e < 8; ++e) yv[e] = 0.0f;
            } else {
                yv[0] = ya[it][0].x; yv[1] = ya[it][0].y; yv[2] = ya[it][0].z; yv[3] = ya[it][0].w;
                yv[4] = ya[it][1].x; yv[5] = ya[it][1].y; yv[6] = ya[it][1].z; yv[7] = ya[it][1].w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = a[e] + bb[e] + g.rscale * (float)rv[it].e[e];
                v = ((float)mv[it].e[e] + mdef) > 0.0f ? v : 0.0f;
                v = relu_on ? fmaxf(v, 0.0f) : v;
                out[e] = v * g.oscale + yv[e];
            }
            const int row = mrow0 + it * 8 + (lane >> 3);
            if (!GATED && g.out_f32) {
                const uint32_t voff = ok[it] ? (uint32_t)(row * g.ldy + col) * 4u : FWN_OOB;
                const u32x4 o0 = {__builtin_bit_cast(unsigned int, out[0]), __builtin_bit_cast(unsigned int, out[1]),
                                  __builtin_bit_cast(unsigned int, out[2]), __builtin_bit_cast(unsigned int, out[3])};
                const u32x4 o1 = {__builtin_bit_cast(unsigned int, out[4]), __builtin_bit_cast(unsigned int, out[5]),
                                  __builtin_bit_cast(unsigned int, out[6]), __builtin_bit_cast(unsigned int, out[7])};
                __builtin_amdgcn_raw_buffer_store_b128(o0, sY, voff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(o1, sY, voff == FWN_OOB ? FWN_OOB : voff + 16u, 0, 0);
            } else if constexpr (gated) {
                // the gate's derivative instead of a store + fwn_gate_bwd: d (rounded to bf16 as it would have been stored)
                // times the kept factors; same expressions as gate_bwd_kernel and the direct epilogue below
                Pack16 of, og;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = (float)(bf16)out[e], tf = (float)tfv[it].e[e], sg = (float)sgv[it].e[e];
                    of.e[e] = (bf16)(d * sg * (1.0f - tf * tf));
                    og.e[e] = (bf16)(d * tf * sg * (1.0f - sg));
                }
                const srd_t sGo = make_srd(g.gate_out, (uint32_t)((size_t)g.M * 512 * 2));
                const uint32_t vo = ok[it] ? (uint32_t)(row * 512 + gc) * 2u : FWN_OOB;
                __builtin_amdgcn_raw_buffer_store_b128(of.w, sGo, vo, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(og.w, sGo, vo, 512, 0);
            } else {
                Pack16 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o.e[e] = (bf16)out[e];
                __builtin_amdgcn_raw_buffer_store_b128(o.w, sY, ok[it] ? (uint32_t)(row * g.ldy + col) * 2u : FWN_OOB, 0, 0);
            }
        }
    }
    template <int MI>
    __device__ void epilogue(const f32x16 (&acc)[MI][2], int mrow0, int ncol0, int lane) const {
        // Branch-free: an absent operand (no residual / mask / accumulate / bias) gets a zero-sized buffer descriptor, rows
        // past M and columns past N an out-of-range offset - loads read 0, stores are dropped - so the loads of a tile
        // issue back to back.  (Written as `ptr ? load : 0` every load sat in its own branch behind an s_waitcnt
        // vmcnt(0): 32 serial round trips per operand, +8 us on a 64 x 128 tile.)
        const int lr = lane & 31;
        const int rbase = mrow0 + 4 * (lane >> 5);
        const srd_t sR = make_srd(g.R ? g.R : g.W, g.R ? (uint32_t)((size_t)g.M * g.ldr * 2) : 0u);
        const srd_t sM = make_srd(g.mask ? g.mask : g.W, g.mask ? (uint32_t)((size_t)g.M * g.ldmask * 2) : 0u);
        const srd_t sB = make_srd(g.bias ? (const void*)g.bias : g.W, g.bias ? (uint32_t)g.N * 4u : 0u);
        const uint32_t ybytes = (uint32_t)((size_t)g.M * g.ldy * (g.out_f32 ? 4 : 2));
        const srd_t sY = make_srd(g.out_f32 ? (const void*)y32 : (const void*)g.Y, ybytes);
        const srd_t sYa = make_srd(g.out_f32 ? (const void*)y32 : (const void*)g.Y, (g.out_f32 && g.accumulate) ? ybytes : 0u);
        const uint32_t gbytes = g.gate_aux ? (uint32_t)((size_t)g.M * 512 * 2) : 0u;
        const srd_t sGa = make_srd(g.gate_aux ? g.gate_aux : g.W, gbytes), sGo = make_srd(g.gate_aux ? (const void*)g.gate_out : g.W, gbytes);
        const float mdef = g.mask ? 0.0f : 1.0f;                    // no mask: every element passes
        const bool relu_on = g.relu != 0;                          // wave-uniform: a scalar select, no branch around loads
        const uint32_t ysz = g.out_f32 ? 4u : 2u;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = ncol0 + ni * 32 + lr;
            const bool cok = col < g.N;
            const float b = buf_load_f32(sB, cok ? (uint32_t)col * 4u : FWN_OOB, 0u);
            const uint32_t vR = cok ? (uint32_t)(rbase * g.ldr + col) * 2u : FWN_OOB;
            const uint32_t vM = cok ? (uint32_t)(rbase * g.ldmask + col) * 2u : FWN_OOB;
            const uint32_t vY = cok ? (uint32_t)(rbase * g.ldy + col) * ysz : FWN_OOB;
            // the 32 columns of this group lie inside or outside the gated range together (gate_col0 is a multiple of 256)
            const int c0 = ncol0 + ni * 32 - g.gate_col0;
            const bool gated = g.gate_aux && c0 >= 0 && c0 < 256;
            const uint32_t vG = cok ? (uint32_t)(rbase * 512 + c0 + lr) * 2u : FWN_OOB;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                float rv[16], mv[16], yv[16], out[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = mi * 32 + acc_row_c(r);
                    rv[r] = buf_load_bf16(sR, vR, (uint32_t)(ro * g.ldr * 2));
                    mv[r] = buf_load_bf16(sM, vM, (uint32_t)(ro * g.ldmask * 2));
                    yv[r] = buf_load_f32(sYa, vY, (uint32_t)(ro * g.ldy * 4));
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[mi][ni][r] + b + g.rscale * rv[r];
                    v = (mv[r] + mdef) > 0.0f ? v : 0.0f;
                    v = relu_on ? fmaxf(v, 0.0f) : v;       // off: v untouched, so a NaN stays a NaN (fmaxf(NaN, -inf) would be -inf)
                    out[r] = v * g.oscale + yv[r];
                }
                if (g.out_f32) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) buf_store_f32(sY, vY, (uint32_t)((mi * 32 + acc_row_c(r)) * g.ldy * 4), out[r]);
                } else if (gated) {
                    // the gate's derivative instead of a store + fwn_gate_bwd: d (rounded to bf16 as it would have been
                    // stored) times the kept factors; same expressions as gate_bwd_kernel
                    float tf[16], sg[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t ro = (uint32_t)((mi * 32 + acc_row_c(r)) * 512 * 2);
                        tf[r] = buf_load_bf16(sGa, vG, ro);
                        sg[r] = buf_load_bf16(sGa, vG, ro + 512u);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t ro = (uint32_t)((mi * 32 + acc_row_c(r)) * 512 * 2);
                        const float d = (float)(bf16)out[r];
                        buf_store_bf16(sGo, vG, ro, d * sg[r] * (1.0f - tf[r] * tf[r]));
                        buf_store_bf16(sGo, vG, ro + 512u, d * tf[r] * sg[r] * (1.0f - sg[r]));
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) buf_store_bf16(sY, vY, (uint32_t)((mi * 32 + acc_row_c(r)) * g.ldy * 2), out[r]);
                }
            }
        }
    }
};

template <int BM, int BN, int WM, int WN, int D, int KSP = 1, int BK = 64>
__global__ __launch_bounds__(64 * WM * WN * KSP) void lin_kernel(const fwn_gemm_desc g, int ntn, int nq_all) {
    const int per = (nq_all + g.nsplit - 1) / g.nsplit;
    const int q0 = (int)blockIdx.z * per;
    const int nq = max(0, min(per, nq_all - q0));      // an empty split still writes its zero partial
    const LinProb p{g, q0, nq, (float*)g.Y + (size_t)blockIdx.z * g.split_stride};
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    gemm_ring_body<BM, BN, WM, WN, BK, D, LinProb, KSP>(p, wg / ntn, wg % ntn);
}

int fwn_gemm_launch(const fwn_gemm_desc* g, hipStream_t st) {
    const fwn_gemm_desc& p = *g;
    int nq_all = 0;
    for (int s = 0; s < g->nseg; ++s) nq_all += (g->seg[s].k + 63) / 64;
    const int M = g->M, ns = g->nsplit;
    const int n128 = (g->N + 127) / 128;
    // Tile by the operand-stream cost of the launch (one workgroup per CU pulls (BM + BN) K 2 bytes through it):
    // ceil(workgroups / 256) x (BM + BN), the larger tile on ties.  For most shapes this is the old "largest tile that
    // still gives ~a workgroup per CU"; it differs where 64 x 128 tiles would run a second, mostly empty round (the
    // conditioning-gradient GEMMs of the late blocks: N = cin up to 10240 against a few hundred rows).
    const int w256 = ((M + 255) / 256) * n128 * ns, w128 = ((M + 127) / 128) * n128 * ns, w64 = ((M + 63) / 64) * n128 * ns;
    const int c256 = ((w256 + 255) / 256) * 384, c128 = ((w128 + 255) / 256) * 256, c64 = ((w64 + 255) / 256) * 192;
    if (c256 <= c128 && c256 <= c64)
        hipLaunchKernelGGL((lin_kernel<256, 128, 8, 2, 3>), dim3(((M + 255) / 256) * n128, 1, ns), dim3(1024), 0, st, p, n128, nq_all);
    else if (c128 <= c64)
        hipLaunchKernelGGL((lin_kernel<128, 128, 4, 2, 3>), dim3(((M + 127) / 128) * n128, 1, ns), dim3(512), 0, st, p, n128, nq_all);
    else {   // small M: 128-wide K chunks (256-byte LDS rows) - a third less time per unit of K on these latency chains
        int nq128 = 0;
        for (int s = 0; s < g->nseg; ++s) nq128 += (g->seg[s].k + 127) / 128;
        // A workgroup streams (BM + BN) K 2 bytes through ONE CU at ~50 GB/s: a launch of a few dozen 64 x 128 tiles is
        // bound by that, not by the chip.  Below FWN_LIN_TINY such workgroups, 32 x 64 tiles (4-way split-K inside the
        // workgroup): four times the CUs, half the bytes each.
        if (((M + 63) / 64) * n128 * ns < FWN_TUNE(FWN_LIN_TINY, 40)) {
            const int n64 = (g->N + 63) / 64;
            hipLaunchKernelGGL((lin_kernel<32, 64, 1, 1, 6, 4, 128>), dim3(((M + 31) / 32) * n64, 1, ns), dim3(256), 0, st, p, n64, nq128);
        } else
            hipLaunchKernelGGL((lin_kernel<64, 128, 2, 2, 3, 1, 128>), dim3(((M + 63) / 64) * n128, 1, ns), dim3(256), 0, st, p, n128, nq128);
    }
    return 0;
}

// ---- dst[c][m] = valid(m) ? src[m + shift][c] : 0, zero padded to ldd columns; row C of dst (if
// ones_row) is 1 for m < M - so the weight-gradient GEMM over the transposed copy also yields the
// bias gradient (column sums) in its extra output row.
__global__ __launch_bounds__(256) void transpose_shift_kernel(const bf16* __restrict__ src, int M, int C, int lds_,
                                                              int shift0, int dshift, int Ti, bf16* __restrict__ dst,
                                                              int ldd, int ones_row) {
    // blockIdx.z = tap: shift = shift0 + z * dshift, written to dst rows [z*C, (z+1)*C)
    __shared__ bf16 tile[64][66];
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int shift = shift0 + (int)blockIdx.z * dshift;
    dst += (size_t)blockIdx.z * C * ldd;
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
    if (ones_row && blockIdx.z == gridDim.z - 1 && blockIdx.y == 0 && threadIdx.x < 64) {
        const int m = m0 + threadIdx.x;
        if (m < ldd) dst[(size_t)C * ldd + m] = (bf16)(m < M ? 1.0f : 0.0f);
    }
}
void fwn_transpose_launch(const void* src, int M, int C, int ld_src, int shift0, int dshift, int ntap, int Ti, void* dst,
                          int ld_dst, int ones_row, hipStream_t st) {
    hipLaunchKernelGGL(transpose_shift_kernel, dim3((ld_dst + 63) / 64, (C + 63) / 64, ntap), dim3(256), 0, st,
                       (const bf16*)src, M, C, ld_src, shift0, dshift, Ti, (bf16*)dst, ld_dst, ones_row);
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

// ---------------------------------------------------------------------------------------------
// Element-wise pieces of the training forward / backward.  Planes are fp32 [M][Ch] (device channel
// order), Z = [log_s | t] is fp32 [M][2*Ch] in the same order.
// ---------------------------------------------------------------------------------------------
// ActNorm forward on one plane, in place: y = (x + shift[c]) * scale[c]   (model.py:86-94)
__global__ __launch_bounds__(256) void actnorm_fwd_kernel(float* __restrict__ x, const float* __restrict__ an,
                                                          long n, int Ch) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i & (Ch - 1));
        x[i] = (x[i] + an[c]) * an[Ch + c];
    }
}
// both planes of a flow in one launch: an2 = [2][4][Ch] (the flow's table: plane a, plane b)
__global__ __launch_bounds__(256) void actnorm_fwd2_kernel(float* __restrict__ xa, float* __restrict__ xb,
                                                           const float* __restrict__ an2, long n, int Ch) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < 2 * n; i += (long)gridDim.x * 256) {
        const bool second = i >= n;
        float* x = second ? xb : xa;
        const float* an = an2 + (second ? 4 * Ch : 0);
        const long e = second ? i - n : i;
        const int c = (int)(e & (Ch - 1));
        x[e] = (x[e] + an[c]) * an[Ch + c];
    }
}
// Z here is the ZeroConv output BEFORE its exp(3 scale) factor ez (modules.py:51-56): (log_s | t) = Z * ez.
// coupling forward (model.py:124-141): out_b = (y_b - t) exp(-log_s), in place over y_b;
// partial[block] = sum(-log_s) of the block's elements (fixed order).
// (A stage entry point since round 3: the training forward runs the inference tail, whose epilogue holds the coupling.)
__global__ __launch_bounds__(256) void coupling_fwd_kernel(float* __restrict__ yb, const float* __restrict__ Z,
                                                           const float* __restrict__ ez, long n, int Ch,
                                                           float* __restrict__ partial) {
    __shared__ float red[256];
    float acc = 0.0f;
    const long per = (n + gridDim.x - 1) / gridDim.x;
    const long i0 = (long)blockIdx.x * per, i1 = min(n, i0 + per);
    for (long i = i0 + threadIdx.x; i < i1; i += 256) {
        const long m = i / Ch;
        const int c = (int)(i - m * Ch);
        const float ls = Z[m * 2 * Ch + c] * ez[c], t = Z[m * 2 * Ch + Ch + c] * ez[Ch + c];
        yb[i] = (yb[i] - t) * __expf(-ls);
        acc -= ls;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// coupling backward.  In: g = dL/d out_b (fp32, becomes dL/d y_b in place), ob = out_b (becomes y_b in
// place), Z.  Out: dZ bf16 [M][ldz] = gradient wrt Z (already times ez), dzz fp32 [M][2Ch] = d(log_s|t) * (log_s|t)
// (ZeroConv scale gradient = 3 * column sums).  cls = d(-logdet)/dlog_s = +1 / (2 M Ch)   (model.py:135: mean(-log_s)/2).
__global__ __launch_bounds__(256) void coupling_bwd_kernel(float* __restrict__ g, float* __restrict__ ob,
                                                           const float* __restrict__ Z, const float* __restrict__ ez,
                                                           long n, int Ch, float cls, bf16* __restrict__ dZ, int ldz,
                                                           float* __restrict__ dzz, const float* __restrict__ ya,
                                                           bf16* __restrict__ ya_bf, int ldya) {
    // ya / ya_bf (optional): also the bf16 copy of the other plane that the front conv's weight gradient reads, and the
    // zero padding of dZ's rows (ldz > 2 Ch) - two launches less per flow
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long m = i / Ch;
        const int c = (int)(i - m * Ch);
        if (ya_bf) {        // rows of ldya >= Ch channels, zero padded (16-byte rows for the TN GEMM)
            ya_bf[m * ldya + c] = (bf16)ya[i];
            if (c == 0) {
                for (int k = Ch; k < ldya; ++k) ya_bf[m * ldya + k] = (bf16)0.0f;
                for (int k = 2 * Ch; k < ldz; ++k) dZ[m * ldz + k] = (bf16)0.0f;
            }
        }
        const float ls = Z[m * 2 * Ch + c] * ez[c], t = Z[m * 2 * Ch + Ch + c] * ez[Ch + c];
        const float e = __expf(-ls), gb = g[i], o = ob[i];
        const float dls = -gb * o + cls, dt = -gb * e;
        g[i] = gb * e;
        ob[i] = o / e + t;
        dZ[m * ldz + c] = (bf16)(dls * ez[c]);
        dZ[m * ldz + Ch + c] = (bf16)(dt * ez[Ch + c]);
        dzz[m * 2 * Ch + c] = dls * ls;
        dzz[m * 2 * Ch + Ch + c] = dt * t;
    }
}
// gate backward: aux = [tanh f | sigmoid g] (bf16 [M][512]), do_ (bf16 [M][256]) ->
// dpre = [do * sg * (1 - tf^2) | do * tf * sg * (1 - sg)]  (bf16 [M][512], natural channel order)
__global__ __launch_bounds__(256) void gate_bwd_kernel(const bf16* __restrict__ do_, int ld_do, const bf16* __restrict__ aux,
                                                       long n, bf16* __restrict__ dpre) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long m = i >> 8;
        const int c = (int)(i & 255);
        const float d = (float)do_[m * ld_do + c], tf = (float)aux[m * 512 + c], sg = (float)aux[m * 512 + 256 + c];
        dpre[m * 512 + c] = (bf16)(d * sg * (1.0f - tf * tf));
        dpre[m * 512 + 256 + c] = (bf16)(d * tf * sg * (1.0f - sg));
    }
}
// out[c] = scale * sum_m A[m][c] * (B ? B[m][c] : 1), fp32 [M][C].  Two passes, both in a fixed order:
// blockIdx.y owns a contiguous range of rows and writes partial[blockIdx.y][c]; colsum_final sums them.
__global__ __launch_bounds__(256) void colsum_prod_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                          long M, int C, float* __restrict__ partial) {
    __shared__ double red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    const long per = (M + gridDim.y - 1) / gridDim.y;
    const long m0 = (long)blockIdx.y * per, m1 = min(M, m0 + per);
    double acc = 0.0;
    if (c < C)
        for (long m = m0 + part; m < m1; m += 4) acc += (double)A[m * C + c] * (B ? (double)B[m * C + c] : 1.0);
    red[part][threadIdx.x & 63] = acc;
    __syncthreads();
    if (part == 0 && c < C)
        partial[(size_t)blockIdx.y * C + c] =
            (float)((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int nblk, int C, float scale,
                                                           float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double a = 0.0;
    for (int b = 0; b < nblk; ++b) a += (double)partial[(size_t)b * C + c];
    out[c] = (float)(a * scale);
}
// ActNorm backward on one plane (model.py:86-94): in dy (becomes dx = dy * scale in place), y (becomes
// x = y / scale - shift in place).
__global__ __launch_bounds__(256) void actnorm_bwd_kernel(float* __restrict__ dy, float* __restrict__ y,
                                                          const float* __restrict__ an, long n, int Ch) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i & (Ch - 1));
        dy[i] *= an[Ch + c];
        y[i] = y[i] * an[2 * Ch + c] - an[c];
    }
}
// ---- the parameter-sized gradients around a flow's ActNorm in two launches (model.py:86-94, modules.py:51-56) ----
// s1[role][c] = sum_m g[m][c], s2[role][c] = sum_m g[m][c] y[m][c] (y = ActNorm OUTPUT of plane role), then in
// place g <- g * scale, y <- y * iscale - shift (back to the flow's input);  z[j] = sum_m dzz[m][j] (ZeroConv
// scale).  Pass 1: a workgroup owns a contiguous row range; 256 is a multiple of Ch and 2 Ch, so a thread stays on
// one column; fp64 sums, fixed order.  Pass 2 totals the row ranges and scatters into the parameters' order:
// db[role Ch + br[c]] = s1 scale[c], dlogs[role Ch + br[c]] = 3 s2 - 3 / (2 Ch), dzscale[zc[j]] = 3 z[j].
__global__ __launch_bounds__(256) void flow_small_grads_kernel(float* __restrict__ ga, float* __restrict__ ya,
                                                               float* __restrict__ gb, float* __restrict__ yb,
                                                               const float* __restrict__ dzz, const float* __restrict__ an,
                                                               long M, int Ch, double* __restrict__ partial) {
    __shared__ double red[5][256];
    const int t = threadIdx.x;
    const long per = (M + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * per, r1 = min(M, r0 + per);
    const int c = t & (Ch - 1);
    const float sh_a = an[c], sc_a = an[Ch + c], is_a = an[2 * Ch + c];
    const float sh_b = an[4 * Ch + c], sc_b = an[5 * Ch + c], is_b = an[6 * Ch + c];
    double s1a = 0.0, s2a = 0.0, s1b = 0.0, s2b = 0.0, sz = 0.0;
    for (long i = r0 * Ch + t; i < r1 * Ch; i += 256) {
        const float g0 = ga[i], y0 = ya[i], g1 = gb[i], y1 = yb[i];
        s1a += g0; s2a += (double)g0 * y0;
        s1b += g1; s2b += (double)g1 * y1;
        ga[i] = g0 * sc_a; ya[i] = y0 * is_a - sh_a;
        gb[i] = g1 * sc_b; yb[i] = y1 * is_b - sh_b;
    }
    for (long i = r0 * 2 * Ch + t; i < r1 * 2 * Ch; i += 256) sz += dzz[i];
    red[0][t] = s1a; red[1][t] = s2a; red[2][t] = s1b; red[3][t] = s2b; red[4][t] = sz;
    __syncthreads();
    // thread k holds the sums of channel k mod Ch (k mod 2 Ch for the ZeroConv scale): fold the 256 entries down to one
    // period by halving - a fixed order, and 8 steps where the early blocks (Ch = 1: one thread adding 256 doubles
    // from LDS one after the other) spent 6 us of a 14 us launch on the flow's chain
    for (int st = 128; st >= Ch; st >>= 1) {
        if (t < st) {
#pragma unroll
            for (int q = 0; q < 4; ++q) red[q][t] += red[q][t + st];
            if (st >= 2 * Ch) red[4][t] += red[4][t + st];
        }
        __syncthreads();
    }
    double* o = partial + (size_t)blockIdx.x * 6 * Ch;
    for (int idx = t; idx < 6 * Ch; idx += 256) {
        const int q = idx < 4 * Ch ? idx / Ch : 4, col = idx < 4 * Ch ? idx % Ch : idx - 4 * Ch;
        o[idx] = red[q][col];
    }
}
__global__ __launch_bounds__(256) void flow_small_grads_final_kernel(const double* __restrict__ partial, int nb, int Ch,
                                                                     const float* __restrict__ an,
                                                                     const long long* __restrict__ br,
                                                                     const long long* __restrict__ zc, float* __restrict__ db,
                                                                     float* __restrict__ dlogs, float* __restrict__ dzscale) {
    for (int idx = threadIdx.x; idx < 6 * Ch; idx += 256) {
        double a = 0.0;
        for (int b0 = 0; b0 < nb; b0 += 16) {         // 16 loads in flight, added in block order
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = partial[(size_t)min(b0 + u, nb - 1) * 6 * Ch + idx];
#pragma unroll
            for (int u = 0; u < 16; ++u) a += b0 + u < nb ? v[u] : 0.0;
        }
        if (idx < 4 * Ch) {
            const int role = idx / (2 * Ch), which = (idx / Ch) & 1, c = idx % Ch;
            const int dst = role * Ch + (int)br[c];
            if (which == 0) db[dst] = (float)a * an[role * 4 * Ch + Ch + c];
            else dlogs[dst] = (float)(3.0 * a) - 3.0f / (2.0f * Ch);
        } else {
            dzscale[zc[idx - 4 * Ch]] = (float)(3.0 * a);
        }
    }
}
int fwn_small_grads_blocks(long M, int Ch) {
    long nb = M * Ch / 512;           // two passes of a workgroup over its rows (a latency chain, not bytes); the final
    return (int)(nb < 1 ? 1 : nb > 96 ? 96 : nb);      // pass adds the blocks serially: keep them few
}
void fwn_small_grads_launch(float* ga, float* ya, float* gb, float* yb, const float* dzz, const float* an, long M, int Ch,
                            const long long* br, const long long* zc, double* partial, float* db, float* dlogs,
                            float* dzscale, hipStream_t st) {
    fwn_small_grads_main(ga, ya, gb, yb, dzz, an, M, Ch, partial, st);
    fwn_small_grads_final(an, M, Ch, br, zc, partial, db, dlogs, dzscale, st);
}
// the two halves separately: the first is on the flow's chain (it restores the planes), the second only totals the
// parameter gradients and may run elsewhere
void fwn_small_grads_main(float* ga, float* ya, float* gb, float* yb, const float* dzz, const float* an, long M, int Ch,
                          double* partial, hipStream_t st) {
    hipLaunchKernelGGL(flow_small_grads_kernel, dim3(fwn_small_grads_blocks(M, Ch)), dim3(256), 0, st, ga, ya, gb, yb, dzz, an, M, Ch, partial);
}
void fwn_small_grads_final(const float* an, long M, int Ch, const long long* br, const long long* zc, const double* partial, float* db,
                           float* dlogs, float* dzscale, hipStream_t st) {
    hipLaunchKernelGGL(flow_small_grads_final_kernel, dim3(1), dim3(256), 0, st, partial, fwn_small_grads_blocks(M, Ch), Ch, an, br, zc, db,
                       dlogs, dzscale);
}

// Weight-norm backward (convolutional.py:73-80): W = V g / ||V||_col, straight from the split-K partials.
// ---- weight-norm backward of a GROUP of convolutions (all weight gradients of one flow) in two launches ----
// part: split-K partials fp32 [S][rows][ldp] of a weight-gradient GEMM.  dW[k][n] = scale * sum_s part[s][row_src ?
// row_src[k] : k][col0 + n] (fixed order), bias gradient = the same sum over row `bias_row` (< 0: none).
// dg = sum_k dW V / nrm, dV = (g / nrm) (dW - V dg / nrm).  g == NULL: no weight norm, dV = dW.
// Pass 1 (workgroup = 32 rows x 64 columns of one job; a wave reads 256 contiguous bytes of a row): reduce the
// splits, stage dW in dV, per-workgroup column sums of V^2 and V dW (fp64) to scratch.  Pass 2, same grid: total
// the column sums over the row chunks in a fixed order, finish dg and dV.
struct WnGroup {
    fwn_wn_job job[FWN_MAX_GROUP];
    int first[FWN_MAX_GROUP + 1];
    int first_mid[FWN_MAX_GROUP + 1];   // workgroup prefix of the middle pass: one per (job, 64 columns)
    long soff[FWN_MAX_GROUP];           // scratch offset (doubles) of each job: [row chunk][N][2], then totals [N][2]
    int njobs;
};
__device__ __forceinline__ float wn_sum_splits(const float* __restrict__ p, int nsplit, long stride) {
    float a = 0.0f;
    int s = 0;
    for (; s + 4 <= nsplit; s += 4) {
        const float t0 = p[(size_t)s * stride], t1 = p[(size_t)(s + 1) * stride], t2 = p[(size_t)(s + 2) * stride],
                    t3 = p[(size_t)(s + 3) * stride];
        a = (((a + t0) + t1) + t2) + t3;
    }
    for (; s < nsplit; ++s) a += p[(size_t)s * stride];
    return a;
}
template <int PASS>
__global__ __launch_bounds__(256) void wn_group_kernel(const WnGroup grp, double* __restrict__ scratch) {
    __shared__ double red[2][4][64];
    int jn = 0;
    while (jn + 1 < grp.njobs && (int)blockIdx.x >= grp.first[jn + 1]) ++jn;
    const fwn_wn_job q = grp.job[jn];
    if (PASS == 2 && !q.g) return;
    const int ncol = (q.N + 63) / 64, local = (int)blockIdx.x - grp.first[jn];
    const int nc = local % ncol, kc = local / ncol, nkc = (q.K + 31) / 32;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = nc * 64 + lane;
    double* sc = scratch + grp.soff[jn];
    if (PASS == 1) {
        double ss = 0.0, dot = 0.0;
        if (n < q.N) {
            const int sn = q.col0 + (q.col_src ? q.col_src[n] : n);          // source column of output column n
            if (kc == 0 && w == 0 && q.db && q.bias_row >= 0)
                q.db[n] = q.scale * wn_sum_splits(q.part + (size_t)q.bias_row * q.ldp + sn, q.nsplit, (long)q.split_stride);
            // the 8 rows of this thread together: every load of a pass over the splits is independent of the others
            // (one memory round trip per split instead of one per row and split); each element still sums its
            // splits in ascending order
            size_t off[8];
            float d[8], v[8];
            bool ok[8];
            int kk[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = kc * 32 + w + 4 * i;
                ok[i] = k < q.K;
                kk[i] = ok[i] ? k : q.K - 1;
                d[i] = 0.0f;
            }
            // one uniform branch per table / operand, not one per load: the loads inside issue back to back
            if (q.row_src) {
#pragma unroll
                for (int i = 0; i < 8; ++i) off[i] = (size_t)q.row_src[kk[i]] * q.ldp + sn;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) off[i] = (size_t)kk[i] * q.ldp + sn;
            }
            if (q.g) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = q.V[(size_t)kk[i] * q.N + n];
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = 0.0f;
            }
            // splits four at a time: 32 loads in flight, added in ascending split order as before
            int s_ = 0;
            for (; s_ + 4 <= q.nsplit; s_ += 4) {
                float t[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* ps = q.part + (size_t)(s_ + u) * q.split_stride;
#pragma unroll
                    for (int i = 0; i < 8; ++i) t[u][i] = ps[off[i]];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < 8; ++i) d[i] += t[u][i];
            }
            if (s_ < q.nsplit) {        // the last 1..3 splits in one round trip as well
                float t[3][8];
                const int rem = q.nsplit - s_;
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const float* ps = q.part + (size_t)(s_ + (u < rem ? u : 0)) * q.split_stride;
#pragma unroll
                    for (int i = 0; i < 8; ++i) t[u][i] = ps[off[i]];
                }
#pragma unroll
                for (int u = 0; u < 3; ++u)
#pragma unroll
                    for (int i = 0; i < 8; ++i) d[i] = u < rem ? d[i] + t[u][i] : d[i];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = kc * 32 + w + 4 * i;
                const float di = q.scale * d[i];
                if (ok[i]) {
                    q.dV[(size_t)k * q.N + n] = di;
                    ss += (double)v[i] * v[i];
                    dot += (double)v[i] * (double)di;
                }
            }
        }
        if (!q.g) return;
        red[0][w][lane] = ss;
        red[1][w][lane] = dot;
        __syncthreads();
        if (w == 0 && n < q.N) {
            double* o = sc + ((size_t)kc * q.N + n) * 2;
            o[0] = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
            o[1] = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
        }
    } else {
        if (n >= q.N) return;
        const double* tot = sc + ((size_t)nkc * q.N + n) * 2;      // column totals of the middle pass
        const double ss = tot[0], dot = tot[1];
        const double nrm = sqrt(fmax(ss, 1e-12)), dgn = dot / nrm;
        if (kc == 0 && w == 0) q.dg[n] = (float)dgn;
        const double gn = (double)q.g[n] / nrm, dn = dgn / nrm;        // per column: no division per element
        float dv[8], vv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {                 // all loads first
            const size_t e = (size_t)min(kc * 32 + w + 4 * i, q.K - 1) * q.N + n;
            dv[i] = q.dV[e];
            vv[i] = q.V[e];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = kc * 32 + w + 4 * i;
            if (k < q.K) q.dV[(size_t)k * q.N + n] = (float)(gn * ((double)dv[i] - (double)vv[i] * dn));
        }
    }
}
// Middle pass: column totals of the per-row-chunk sums, once per column (a conditioning conv of the last block has
// K = 10240 = 320 row chunks: left to pass 2, every one of its workgroups would re-read all of them).  One workgroup
// per (job, 64 columns); wave w adds chunks w, w+4, .. (8 loads in flight), the four waves are added in order.
__global__ __launch_bounds__(256) void wn_group_mid_kernel(const WnGroup grp, double* __restrict__ scratch) {
    __shared__ double red[2][4][64];
    int jn = 0;
    while (jn + 1 < grp.njobs && (int)blockIdx.x >= grp.first_mid[jn + 1]) ++jn;
    const fwn_wn_job q = grp.job[jn];
    const int nc = (int)blockIdx.x - grp.first_mid[jn], nkc = (q.K + 31) / 32;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = min(nc * 64 + lane, q.N - 1);
    const double* sc = scratch + grp.soff[jn];
    double ss = 0.0, dot = 0.0;
    for (int c0 = w; c0 < nkc; c0 += 32) {
        double a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double* o = sc + ((size_t)min(c0 + 4 * u, nkc - 1) * q.N + n) * 2;
            a[u] = o[0];
            b[u] = o[1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c0 + 4 * u < nkc) { ss += a[u]; dot += b[u]; }
    }
    red[0][w][lane] = ss;
    red[1][w][lane] = dot;
    __syncthreads();
    if (w == 0 && nc * 64 + lane < q.N) {
        double* o = scratch + grp.soff[jn] + ((size_t)nkc * q.N + n) * 2;
        o[0] = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
        o[1] = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
    }
}
long fwn_wn_group_scratch_doubles(const fwn_wn_job* jobs, int njobs) {
    long tot = 0;
    for (int j = 0; j < njobs; ++j)
        if (jobs[j].g) tot += (long)((jobs[j].K + 31) / 32 + 1) * jobs[j].N * 2;
    return tot > 0 ? tot : 1;
}
void fwn_wn_group_launch(const fwn_wn_job* jobs, int njobs, double* scratch, hipStream_t st) {
    WnGroup g;
    g.njobs = njobs;
    int total = 0, total_mid = 0;
    long off = 0;
    for (int j = 0; j < njobs; ++j) {
        g.job[j] = jobs[j];
        g.first[j] = total;
        g.first_mid[j] = total_mid;
        g.soff[j] = off;
        total += ((jobs[j].K + 31) / 32) * ((jobs[j].N + 63) / 64);
        if (jobs[j].g) {
            off += (long)((jobs[j].K + 31) / 32 + 1) * jobs[j].N * 2;
            total_mid += (jobs[j].N + 63) / 64;
        }
    }
    for (int j = njobs; j <= FWN_MAX_GROUP; ++j) { g.first[j] = total; g.first_mid[j] = total_mid; }
    hipLaunchKernelGGL(wn_group_kernel<1>, dim3(total), dim3(256), 0, st, g, scratch);
    if (total_mid > 0) {
        hipLaunchKernelGGL(wn_group_mid_kernel, dim3(total_mid), dim3(256), 0, st, g, scratch);
        hipLaunchKernelGGL(wn_group_kernel<2>, dim3(total), dim3(256), 0, st, g, scratch);
    }
}

static inline unsigned ew_grid(long n) { long b = (n + 255) / 256; return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }
void fwn_ew_actnorm_fwd(float* x, const float* an, long n, int Ch, hipStream_t st) {
    hipLaunchKernelGGL(actnorm_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x, an, n, Ch);
}
void fwn_ew_actnorm_fwd2(float* xa, float* xb, const float* an2, long n, int Ch, hipStream_t st) {
    hipLaunchKernelGGL(actnorm_fwd2_kernel, dim3(ew_grid(2 * n)), dim3(256), 0, st, xa, xb, an2, n, Ch);
}
void fwn_ew_coupling_fwd(float* yb, const float* Z, const float* ez, long n, int Ch, float* partial, int nblocks,
                         hipStream_t st) {
    hipLaunchKernelGGL(coupling_fwd_kernel, dim3(nblocks), dim3(256), 0, st, yb, Z, ez, n, Ch, partial);
}
void fwn_ew_coupling_bwd(float* g, float* ob, const float* Z, const float* ez, long n, int Ch, float cls, void* dZ,
                         int ldz, float* dzz, hipStream_t st) {
    hipLaunchKernelGGL(coupling_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, st, g, ob, Z, ez, n, Ch, cls, (bf16*)dZ, ldz, dzz,
                       (const float*)nullptr, (bf16*)nullptr, 0);
}
void fwn_ew_coupling_bwd_ex(float* g, float* ob, const float* Z, const float* ez, long n, int Ch, float cls, void* dZ,
                            int ldz, float* dzz, const float* ya, void* ya_bf, int ldya, hipStream_t st) {
    hipLaunchKernelGGL(coupling_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, st, g, ob, Z, ez, n, Ch, cls, (bf16*)dZ, ldz, dzz, ya,
                       (bf16*)ya_bf, ldya);
}
void fwn_ew_gate_bwd(const void* do_, int ld_do, const void* aux, long n, void* dpre, hipStream_t st) {
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, st, (const bf16*)do_, ld_do, (const bf16*)aux, n, (bf16*)dpre);
}
int fwn_colsum_blocks(long M, int C) {
    long nb = M / 256;                                    // >= 256 rows per block
    const long cap = 1024 / ((C + 63) / 64);
    if (nb > cap) nb = cap;
    return (int)(nb < 1 ? 1 : nb);
}
void fwn_ew_colsum_prod(const float* A, const float* B, long M, int C, float scale, float* partial, float* out,
                        hipStream_t st) {
    const int nb = fwn_colsum_blocks(M, C);
    hipLaunchKernelGGL(colsum_prod_kernel, dim3((C + 63) / 64, nb), dim3(256), 0, st, A, B, M, C, partial);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, st, partial, nb, C, scale, out);
}
void fwn_ew_actnorm_bwd(float* dy, float* y, const float* an, long n, int Ch, hipStream_t st) {
    hipLaunchKernelGGL(actnorm_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, st, dy, y, an, n, Ch);
}

// ---- backward of one up-sampling stage: Conv2DTranspose((2s,3),(s,1),'same') + LeakyReLU(0.4) -----
// (model.py:301-311; forward in aux_kernels.hip upsample_kernel).  y, dy: [B][H*s][W]; x: [B][H][W].
// dpre = dy * (y > 0 ? 1 : 0.4), in place over dy.
__global__ __launch_bounds__(256) void up_dpre_kernel(float* __restrict__ dy, const float* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        dy[i] *= y[i] > 0.0f ? 1.0f : 0.4f;
}
// dx[b,i,w] = sum_{k,kw} dpre[b, i*s + k - s/2, w + kw - 1] * wk[k][kw]
__global__ __launch_bounds__(256) void up_dx_kernel(const float* __restrict__ dpre, int B, int H, int W, int s,
                                                    const float* __restrict__ wk, float* __restrict__ dx) {
    const long total = (long)B * H * W;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int w = (int)(idx % W), i = (int)((idx / W) % H), b = (int)(idx / ((long)W * H));
        float acc = 0.0f;
        for (int k = 0; k < 2 * s; ++k) {
            const int tau = i * s + k - s / 2;
            if (tau < 0 || tau >= H * s) continue;
            const float* row = dpre + ((long)b * H * s + tau) * W;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ww = w + kw - 1;
                if (ww >= 0 && ww < W) acc = fmaf(row[ww], wk[k * 3 + kw], acc);
            }
        }
        dx[idx] = acc;
    }
}
// blockIdx.x = kernel row k (its three kw taps together: one x load serves all three), the last one the bias;
// blockIdx.y = a contiguous chunk of the (b, i) rows.  partial[chunk][tap = 3 k + kw | 6 s] in a fixed order, summed by
// colsum_final_kernel.  The chunk's (row, w) pairs are flattened over the threads, four independent pairs in flight per
// thread; addresses clamped and validity applied as a factor (a branch per load would serialise the round trips).
__global__ __launch_bounds__(256) void up_dw_kernel(const float* __restrict__ dpre, const float* __restrict__ x, int B,
                                                    int H, int W, int s, float* __restrict__ partial) {
    __shared__ double red[3][256];
    const int k = blockIdx.x, ntap = 6 * s;
    const long rows = (long)B * H, per = (rows + gridDim.y - 1) / gridDim.y;
    const long r0 = (long)blockIdx.y * per, r1 = min(rows, r0 + per);
    double acc[3] = {0.0, 0.0, 0.0};
    if (k < 2 * s) {
        const long n = (r1 - r0) * W;
        for (long e0 = threadIdx.x; e0 < n; e0 += 4 * 256) {
            float xv[4], dv[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long e = min(e0 + 256L * u, n - 1);
                const long r = r0 + e / W;
                const int w = (int)(e % W);
                const int i = (int)(r % H), b = (int)(r / H);
                const int tau = i * s + k - s / 2;
                const bool ok = e0 + 256L * u < n && tau >= 0 && tau < H * s;
                const long dr = ((long)b * H * s + min(max(tau, 0), H * s - 1)) * W;
                xv[u] = x[r * W + w] * (ok ? 1.0f : 0.0f);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int ww = w + kw - 1;
                    dv[u][kw] = dpre[dr + min(max(ww, 0), W - 1)] * ((ww >= 0 && ww < W) ? 1.0f : 0.0f);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc[kw] += (double)xv[u] * (double)dv[u][kw];
        }
    } else {        // bias: the sum of the chunk's dpre, eight loads in flight
        const long e0 = r0 * s * W, e1 = r1 * s * W;
        for (long idx = e0 + threadIdx.x; idx < e1; idx += 8 * 256) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = dpre[min(idx + 256L * u, e1 - 1)] * (idx + 256L * u < e1 ? 1.0f : 0.0f);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[0] += (double)v[u];
        }
    }
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) red[kw][threadIdx.x] = acc[kw];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) red[kw][threadIdx.x] += red[kw][threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x < 3) {
        float* o = partial + (size_t)blockIdx.y * (ntap + 1);
        if (k < 2 * s) o[3 * k + threadIdx.x] = (float)red[threadIdx.x][0];
        else if (threadIdx.x == 0) o[ntap] = (float)red[0][0];
    }
}
int fwn_up_bwd_chunks(int B, int H) { const long r = (long)B * H; return (int)(r < 64 ? 1 : (r / 16 > 64 ? 64 : r / 16)); }
void fwn_up_bwd_launch(float* dy, const float* y, const float* x, int B, int H, int W, int s, const float* wk,
                       float* dx, float* dwk_bias, float* partial, hipStream_t st) {
    const long n = (long)B * H * s * W;
    const int nc = fwn_up_bwd_chunks(B, H);
    hipLaunchKernelGGL(up_dpre_kernel, dim3(ew_grid(n)), dim3(256), 0, st, dy, y, n);
    if (dx) hipLaunchKernelGGL(up_dx_kernel, dim3(ew_grid((long)B * H * W)), dim3(256), 0, st, dy, B, H, W, s, wk, dx);
    hipLaunchKernelGGL(up_dw_kernel, dim3(2 * s + 1, nc), dim3(256), 0, st, dy, x, B, H, W, s, partial);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((6 * s + 1 + 255) / 256), dim3(256), 0, st, partial, nc, 6 * s + 1, 1.0f, dwk_bias);
}

// ---------------------------------------------------------------------------------------------
// TN GEMM for weight gradients: dW[tap*Kx + i][j] = sum_m X[m + shift_tap][i] * dY[m][j], both operands
// read as they lie in memory (rows = m) - no transposed copies.  The contraction index is the ROW of
// both LDS tiles, so the MFMA fragments come from ds_read_b64_tr_b16 (4 rows x 16 columns delivered
// column-major): lane i of a 16-lane group receives column i of 4 consecutive rows; two such reads
// give the 8 consecutive k of a v_mfma_f32_32x32x16_bf16 operand.  Tile 128 (i) x 128 (j), 4 waves
// (wave tile 64 x 64), chunks of 64 rows through a 3-slot LDS-DMA ring; LDS image (b) of the
// programming guide (256-byte rows, chunk XOR ((row&3)<<2 | (row>>2)&3)) - conflict-free for the
// transposed reads.  blockIdx.z = split of the row range; partials fp32 [S][R][N] for fwn_wn_backward.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4;
struct TnArgs {
    const bf16* x;   int ldx, Kx, ntap, shift0, dshift;
    const bf16* dy;  int ldy, N;
    int M, Ti, nsplit;
    float* part;     long split_stride;      // [S][ntap*Kx (+1)][N]
    int bias_row;                            // != 0: row ntap*Kx of every partial = column sums of dY (bias gradient)
};
__device__ __forceinline__ int tn_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// A launch covers a GROUP of such GEMMs (the weight gradients of one flow share M and Ti): the job table rides in
// the kernel arguments, workgroup -> (job, row tile, column tile, split) by a scan of the prefix counts, so that
// every GEMM needs only a few splits for the group to fill the chip (fewer partials to write and re-read).
struct TnGroup {
    TnArgs job[FWN_MAX_GROUP];
    int first[FWN_MAX_GROUP + 1];
    int njobs;
};
// WI x WJ waves of 64 x 64 each: <2,2,3> = 128 x 128 tile, 96 KB of LDS; <4,4,2> = 256 x 256 tile (half the
// L2 -> LDS bytes per flop: the long contractions of the first blocks are bound by exactly that), 128 KB.
// X and dY chunks are stored as panels of 128 columns (16 KB images).
// Several groups in ONE launch (round 4: the weight gradients of all flows of a block - their operands exist together once
// the block's chain has run): the group tables live in device memory (tn_table_put_kernel writes one from its kernel
// arguments), the launch carries only the workgroup prefix of every group.  With the whole block in one grid a handful of
// splits fills the chip where every flow on its own needed a dozen - a sixth of the fp32 partials written and read back.
#define FWN_TN_MAXGROUPS 16
struct TnMulti {
    int gfirst[FWN_TN_MAXGROUPS + 1];
    int ngroups;
};
__global__ void tn_table_put_kernel(const TnGroup g, TnGroup* __restrict__ dst) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = g;
}
template <int WI, int WJ, int D>
__device__ __forceinline__ void tn_gemm_body(const TnGroup& grp, const int bid) {
    constexpr int TILE = 64 * 256, XP = WI / 2, YP = WJ / 2, SLOT = (XP + YP) * TILE, NW = WI * WJ;
    constexpr int BMX = 64 * WI, BNY = 64 * WJ;
    constexpr int PX = 16 * XP / NW, PY = 16 * YP / NW;          // DMA pieces per wave per chunk
    static_assert(PX * NW == 16 * XP && PY * NW == 16 * YP, "pieces must divide over the waves");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[D * SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave / WJ, wj = wave % WJ;
    int first[FWN_MAX_GROUP + 1];                                  // (all at once: the table may live in device memory)
#pragma unroll
    for (int j = 0; j <= FWN_MAX_GROUP; ++j) first[j] = grp.first[j];
    int jn = 0;
#pragma unroll
    for (int j = 1; j < FWN_MAX_GROUP; ++j) jn += (j < grp.njobs && bid >= first[j]) ? 1 : 0;
    const TnArgs a = grp.job[jn];
    const int kxt = (a.Kx + BMX - 1) / BMX;                        // row tiles per tap
    const int nbx = a.ntap * kxt, nby = (a.N + BNY - 1) / BNY;
    int first_jn = 0;
#pragma unroll
    for (int j = 0; j < FWN_MAX_GROUP; ++j) first_jn = j == jn ? first[j] : first_jn;
    const int local = bid - first_jn;
    const int bx = local % nbx, by = (local / nbx) % nby, bz = local / (nbx * nby);     // splits slowest: tiles of one row range run together
    const int tap = bx / kxt, kx0 = (bx % kxt) * BMX, n0 = by * BNY;
    const int shift = a.shift0 + tap * a.dshift;
    const int nchunk_all = (a.M + 63) / 64, per = (nchunk_all + a.nsplit - 1) / a.nsplit;
    const int c0 = bz * per, nq = max(0, min(per, nchunk_all - c0));
    // DMA: a 1 KB piece = 4 rows x 256 B; wave w issues pieces w, w+4, .. (16 per tile)
    const int prow = lane >> 4, pch = lane & 15;
    const srd_t sx = make_srd(a.x, (uint32_t)((size_t)a.M * a.ldx * 2)), sy = make_srd(a.dy, (uint32_t)((size_t)a.M * a.ldy * 2));
    auto issue = [&](int q) {
        unsigned char* base = lds + (q % D) * SLOT;
        const int m0 = (c0 + q) * 64;
#pragma unroll
        for (int j = 0; j < PX + PY; ++j) {
            const bool isx = j < PX;
            const int pidx = wave + NW * (isx ? j : j - PX);        // piece of the X (dY) part: panel pidx / 16, 4 rows each
            const int panel = pidx >> 4, pp = pidx & 15;
            const int row = 4 * pp + prow;                          // tile row 0..63
            const int ch = pch ^ (((row & 3) << 2) | ((row >> 2) & 3));   // source chunk for this lane's linear slot
            const int m = m0 + row;
            if (isx) {
                const int t = a.Ti > 0 ? m % a.Ti : m;
                const int col = kx0 + 128 * panel + ch * 8;
                const bool ok = m < a.M && (unsigned)(t + shift) < (unsigned)(a.Ti > 0 ? a.Ti : a.M) && col < a.Kx;
                buf_load16_lds(sx, ok ? (uint32_t)((m + shift) * a.ldx + col) * 2u : FWN_OOB, base + panel * TILE + pp * 1024);
            } else {
                const int col = n0 + 128 * panel + ch * 8;
                const bool ok = m < a.M && col < a.N;
                buf_load16_lds(sy, ok ? (uint32_t)(m * a.ldy + col) * 2u : FWN_OOB, base + (XP + panel) * TILE + pp * 1024);
            }
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    // bias gradient = column sums of dY: the wi == 0 waves of the first row tile add up the dY fragments they hold
    // anyway (fp32 VALU adds next to the MFMAs; two registers instead of a second accumulator tile)
    const bool do_bias = a.bias_row && bx == 0 && wi == 0;
    float bsum[2] = {0.0f, 0.0f};
    // transposed-read addresses: group g = lane>>4 = (khalf, chalf); lane 4q+p of a group supplies row q,
    // 8-byte half (p&1) of chunk c0 + (p>>1)
    const int kh = lane >> 5, chalf = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
    auto tr_addr = [&](int rbase, int cblk) {      // rows rbase + q4, columns 16*cblk ..: chunks 2*cblk + (p4>>1)
        return tn_off(rbase + q4, 2 * cblk + (p4 >> 1)) + 8 * (p4 & 1);
    };
    for (int q = 0; q < D - 1; ++q)
        if (q < nq) issue(q);
    for (int q = 0; q < nq; ++q) {
        const int pending = min(nq, q + D - 1) - (q + 1);
        if (pending >= 1) FWN_WAIT_VMCNT(PX + PY); else FWN_WAIT_VMCNT(0);
        FWN_RING_BARRIER();      // the slot read by the previous chunk is refilled behind it: retire this wave's LDS reads first (gemm_ring.h)
        if (q + D - 1 < nq) issue(q + D - 1);
        const unsigned char* xa = lds + (q % D) * SLOT + (wi >> 1) * TILE;
        const unsigned char* ya = lds + (q % D) * SLOT + (XP + (wj >> 1)) * TILE;
#pragma unroll
        for (int s = 0; s < 4; ++s) {                 // k-steps of 16 rows
            const int rb = 16 * s + 8 * kh;
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {          // operand columns: 32*(2*wi + ib) + 16*chalf + lane-in-group
                const int cb = 2 * (2 * (wi & 1) + ib) + chalf;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xa + tr_addr(rb, cb)));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xa + tr_addr(rb + 4, cb)));
                const short v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                af[ib] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const int cb = 2 * (2 * (wj & 1) + jb) + chalf;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ya + tr_addr(rb, cb)));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ya + tr_addr(rb + 4, cb)));
                const short v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bfr[jb] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) acc[ib][jb] = mfma32(af[ib], bfr[jb], acc[ib][jb]);
            if (do_bias) {
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    float t = 0.0f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) t += (float)bfr[jb][e];
                    bsum[jb] += t;
                }
            }
        }
    }
    // C layout: col = lane & 31 (j), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (i)
    float* out = a.part + (size_t)bz * a.split_stride;
    const int R = a.ntap * a.Kx;
    const uint32_t obytes = (uint32_t)((size_t)(R + (a.bias_row ? 1 : 0)) * a.N * 4);
    const srd_t so = make_srd(out, obytes);
    if (do_bias) {                     // lane and lane + 32 hold the two k halves of column lane & 31
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const float tot = bsum[jb] + __shfl_xor(bsum[jb], 32);
            const int col = n0 + 64 * wj + 32 * jb + lane;
            if (lane < 32) buf_store_f32(so, col < a.N ? (uint32_t)(R * a.N + col) * 4u : FWN_OOB, 0, tot);
        }
    }
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const int col = n0 + 64 * wj + 32 * jb + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kx = kx0 + 64 * wi + 32 * ib + acc_row(r, lane);
                const bool ok = kx < a.Kx && col < a.N;
                buf_store_f32(so, ok ? (uint32_t)((tap * a.Kx + kx) * a.N + col) * 4u : FWN_OOB, 0, acc[ib][jb][r]);
            }
        }
}
template <int WI, int WJ, int D>
__global__ __launch_bounds__(64 * WI * WJ) void tn_gemm_kernel(const TnGroup grp) {
    tn_gemm_body<WI, WJ, D>(grp, (int)blockIdx.x);
}
template <int WI, int WJ, int D>
__global__ __launch_bounds__(64 * WI * WJ) void tn_gemm_multi_kernel(const TnGroup* __restrict__ tab, const TnMulti hdr) {
    int g = 0;
#pragma unroll
    for (int k = 1; k < FWN_TN_MAXGROUPS; ++k) g += (k < hdr.ngroups && (int)blockIdx.x >= hdr.gfirst[k]) ? 1 : 0;
    int gf = 0;
#pragma unroll
    for (int k = 0; k < FWN_TN_MAXGROUPS; ++k) gf = k == g ? hdr.gfirst[k] : gf;
    tn_gemm_body<WI, WJ, D>(tab[g], (int)blockIdx.x - gf);
}
// column sums of a bf16 matrix (bias gradients): out[c] = scale * sum_m dy[m][c]; two fixed-order passes
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16* __restrict__ dy, long M, int C, int ld,
                                                          float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    const long per = (M + gridDim.y - 1) / gridDim.y;
    const long m0 = (long)blockIdx.y * per, m1 = min(M, m0 + per);
    float acc = 0.0f;
    if (c < C)
        for (long m = m0 + part; m < m1; m += 4) acc += (float)dy[m * ld + c];
    red[part][threadIdx.x & 63] = acc;
    __syncthreads();
    if (part == 0 && c < C)
        partial[(size_t)blockIdx.y * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
#ifndef FWN_TN256_MIN
// 256 x 256 output tiles from 128 rows on (2048 until round 2): the weight-gradient GEMMs of the late blocks - K = a few
// hundred rows against cin up to 10240 output rows - are bound by writing their output; a quarter of the workgroups does
// that in fewer rounds (training step -0.18 ms).
#define FWN_TN256_MIN 128
#endif
int fwn_tn_tile(int M) { return M >= FWN_TUNE(FWN_TN256_MIN, FWN_TN256_MIN) ? 256 : 128; }     // output tile edge of the weight-gradient GEMM
static int tn_fill_group(TnGroup& g, const fwn_tn_job* jobs, int njobs, int M, int Ti) {       // -> workgroups of the group
    memset(&g, 0, sizeof(g));
    g.njobs = njobs;
    int total = 0;
    for (int j = 0; j < njobs; ++j) {
        const fwn_tn_job& q = jobs[j];
        g.job[j] = TnArgs{(const bf16*)q.x, q.ldx, q.Kx, q.ntap, q.shift0, q.dshift, (const bf16*)q.dy, q.ldy, q.N, M, Ti,
                          q.nsplit, q.part, (long)q.split_stride, q.bias_row};
        g.first[j] = total;
        const int tile = fwn_tn_tile(M);
        total += q.ntap * ((q.Kx + tile - 1) / tile) * ((q.N + tile - 1) / tile) * q.nsplit;
    }
    for (int j = njobs; j <= FWN_MAX_GROUP; ++j) g.first[j] = total;
    return total;
}
void fwn_tn_group_launch(const fwn_tn_job* jobs, int njobs, int M, int Ti, hipStream_t st) {
    TnGroup g;
    const int total = tn_fill_group(g, jobs, njobs, M, Ti);
    if (fwn_tn_tile(M) == 256) hipLaunchKernelGGL((tn_gemm_kernel<4, 4, 2>), dim3(total), dim3(1024), 0, st, g);
    else hipLaunchKernelGGL((tn_gemm_kernel<2, 2, 3>), dim3(total), dim3(256), 0, st, g);
}
// ngroups (<= fwn_tn_multi_max()) groups of <= FWN_MAX_GROUP jobs each, all at the same M / Ti, in ONE launch; `table`: device
// memory of fwn_tn_table_bytes() bytes that stays untouched until the launch has run (the group tables are written into
// it by ngroups one-thread launches in front of it).
size_t fwn_tn_table_bytes(void) { return FWN_TN_MAXGROUPS * sizeof(TnGroup); }
int fwn_tn_multi_max(void) { return FWN_TN_MAXGROUPS; }
void fwn_tn_multi_launch(const fwn_tn_job* const* jobs, const int* njobs, int ngroups, int M, int Ti, void* table, hipStream_t st) {
    TnMulti hdr;
    memset(&hdr, 0, sizeof(hdr));
    hdr.ngroups = ngroups;
    TnGroup* tab = (TnGroup*)table;
    int total = 0;
    for (int k = 0; k < ngroups; ++k) {
        TnGroup g;
        hdr.gfirst[k] = total;
        total += tn_fill_group(g, jobs[k], njobs[k], M, Ti);
        hipLaunchKernelGGL(tn_table_put_kernel, dim3(1), dim3(64), 0, st, g, tab + k);
    }
    for (int k = ngroups; k <= FWN_TN_MAXGROUPS; ++k) hdr.gfirst[k] = total;
    if (fwn_tn_tile(M) == 256) hipLaunchKernelGGL((tn_gemm_multi_kernel<4, 4, 2>), dim3(total), dim3(1024), 0, st, tab, hdr);
    else hipLaunchKernelGGL((tn_gemm_multi_kernel<2, 2, 3>), dim3(total), dim3(256), 0, st, tab, hdr);
}
void fwn_colsum_bf16_launch(const void* dy, long M, int C, int ld, float scale, float* partial, float* out, hipStream_t st) {
    const int nb = fwn_colsum_blocks(M, C);
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3((C + 63) / 64, nb), dim3(256), 0, st, (const bf16*)dy, M, C, ld, partial);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, st, partial, nb, C, scale, out);
}
