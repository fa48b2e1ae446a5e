// ZeroConv + coupling epilogue problem of the N-split tail, shared by flow_kernels.hip (a launch of its own) and
// flow_persist.hip (a ticket of the one-launch flow).
#pragma once
#include "common.h"
#include "gemm_ring.h"

__device__ __forceinline__ int swap_bits23(int c) { return (c & ~12) | ((c & 4) << 1) | ((c & 8) >> 1); }

struct TailZeroProb {     // (log_s | t) = U' Wz, then the coupling + ActNorm of one 32-channel pair tile (tile_n = pt)
    static constexpr bool A_DMA = true;
    static constexpr bool ALLOW_256 = false;
    const bf16* U;        // [M][256]
    const bf16* Wz;       // [npt*64][256]
    const float* bz;      // [npt*64]
    const float* ez;      // [npt*64]
    const float* an;      // [2][4][Ch]
    float* xa;
    float* xb;
    float* partial;       // [mtiles*8] (forward) or nullptr
    int M, Ch, npt, inverse;
    float* save_z;        // optional (training): Z = U Wz + bz, fp32 [M][2 Ch] (log_s channels, then t channels)
    struct RowCtx { int row; };
    struct ChunkCtx { int k0; };
    template <int BK> __device__ int nchunks() const { return FWN_HID / BK; }
    __device__ RowCtx row_ctx(int row) const { return RowCtx{row}; }
    template <int BK> __device__ ChunkCtx chunk_ctx(int q) const { return ChunkCtx{q * BK}; }
    __device__ srd_t a_srd(const ChunkCtx&) const { return make_srd(U, (uint32_t)((size_t)M * FWN_HID * 2)); }
    __device__ uint32_t a_voff(const ChunkCtx& cc, const RowCtx& rc, int c8) const {
        return rc.row < M ? (uint32_t)(rc.row * FWN_HID + cc.k0 + c8 * 8) * 2u : FWN_OOB;
    }
    __device__ srd_t b_srd(const ChunkCtx&) const { return make_srd(Wz, (uint32_t)(npt * 64u * FWN_HID * 2u)); }
    __device__ uint32_t b_voff(const ChunkCtx& cc, int n, int c8) const {
        return (uint32_t)(n * FWN_HID + cc.k0 + c8 * 8) * 2u;
    }
    __device__ float acc_init(int col) const { return bz[col]; }     // ZeroConv bias: (acc + b) * exp(3 scale)
    // gemm_ring.h PREFETCH (small tiles): the two planes' elements this lane transforms in place (no other lane touches them)
    static constexpr bool PREFETCH = true;
    template <int MI>
    __device__ void prefetch(float (&pre)[MI][32], int mrow0, int ncol0, int lane) const {
        const int tau = (ncol0 >> 6) * 32 + (lane & 31);
        const uint32_t plane_bytes = (uint32_t)((size_t)M * Ch * 4);
        const srd_t sxa = make_srd(xa, plane_bytes), sxb = make_srd(xb, plane_bytes);
        const uint32_t voff = tau < Ch ? (uint32_t)((mrow0 + 4 * (lane >> 5)) * Ch + tau) * 4u : FWN_OOB;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t so = (uint32_t)((mi * 32 + acc_row_c(r)) * Ch * 4);
                pre[mi][r] = buf_load_f32(sxb, voff, so);
                pre[mi][16 + r] = buf_load_f32(sxa, voff, so);
            }
    }
    template <int MI>
    __device__ void epilogue_pre(const f32x16 (&acc)[MI][2], int mrow0, int ncol0, int lane, const float (&pre)[MI][32]) const {
        epilogue_impl<MI, true>(acc, mrow0, ncol0, lane, pre);
    }
    template <int MI>
    __device__ void epilogue(const f32x16 (&acc)[MI][2], int mrow0, int ncol0, int lane) const {
        const float none[MI][32] = {};
        epilogue_impl<MI, false>(acc, mrow0, ncol0, lane, none);
    }
    // The one-launch flow (flow_persist.h) stages everything this epilogue reads in LDS before it waits for its producers
    // (stage_lds: LDS-DMA, no registers): the two planes' tile as [2][64 rows][CW = min(Ch, 32)] fp32 (rows past M zeros)
    // at `lt`, behind it exp(3 scale) of the pair tile [64] and the eight ActNorm rows [8][32].  MI = 1 only.
    static constexpr int LDS_BYTES = 2 * 64 * 32 * 4 + 256 + 8 * 32 * 4;
    __device__ __forceinline__ void stage_lds(unsigned char* lt, int m0, int ncol0, int wave, int lane) const {
        const int pt = ncol0 >> 6, cw = Ch < 32 ? Ch : 32, cl = 31 - __clz(cw);
        const uint32_t plane_bytes = (uint32_t)((size_t)M * Ch * 4);
        const srd_t sxa = make_srd(xa, plane_bytes), sxb = make_srd(xb, plane_bytes);
        for (int q = wave; q < 2 * cw; q += 8) {               // 64 elements of 4 bytes per piece
            const int pl = q >= cw, idx = (q - pl * cw) * 64 + lane, row = idx >> cl, ch = idx & (cw - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(pl ? sxa : sxb, (lds_ptr_t)(lt + pl * 8192 + (q - pl * cw) * 256), 4,
                                                     (uint32_t)((m0 + row) * Ch + pt * 32 + ch) * 4u, 0, 0, 0);
        }
        if (wave == 7) __builtin_amdgcn_raw_ptr_buffer_load_lds(make_srd(ez, (uint32_t)(npt * 64 * 4)), (lds_ptr_t)(lt + 16384), 4,
                                                                 (uint32_t)(pt * 64 + lane) * 4u, 0, 0, 0);
        if (wave >= 3 && wave < 7) {
            const int k = (wave - 3) * 2 + (lane >> 5), tau = pt * 32 + (lane & 31);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(make_srd(an, (uint32_t)(8 * Ch * 4)), (lds_ptr_t)(lt + 16384 + 256 + (wave - 3) * 256), 4,
                                                     (uint32_t)(k * Ch + (tau < Ch ? tau : 0)) * 4u, 0, 0, 0);
        }
    }
    // bad: a producer of this ticket gave up waiting (flow_persist.h): every output of the ticket becomes NaN
    __device__ void epilogue_lds(const f32x16 (&acc)[1][2], int mrow0, int ncol0, int lane, const unsigned char* lt, bool bad) const {
        const float none[1][32] = {};
        epilogue_impl<1, false, true>(acc, mrow0, ncol0, lane, none, lt, bad);
    }
    template <int MI, bool PRE, bool LDS = false>
    __device__ __forceinline__ void epilogue_impl(const f32x16 (&acc)[MI][2], int mrow0, int ncol0, int lane, const float (&pre)[MI][32],
                                                  const unsigned char* lt = nullptr, bool bad = false) const {
        const int lr = lane & 31, pt = ncol0 >> 6;
        const int tau = pt * 32 + lr;
        const bool chok = tau < Ch;
        const int tc = chok ? tau : 0;
        const float* an_a = an;
        const float* an_b = an + 4 * Ch;
        float els, et, a_sh, a_sc, a_isc, a_l3, b_sh, b_sc, b_isc, b_l3;
        if constexpr (LDS) {
            const float* cz = (const float*)(lt + 16384);
            const float* ca = cz + 64 + lr;
            els = cz[lr]; et = cz[32 + lr];
            a_sh = ca[0]; a_sc = ca[32]; a_isc = ca[64]; a_l3 = ca[96];
            b_sh = ca[128]; b_sc = ca[160]; b_isc = ca[192]; b_l3 = ca[224];
        } else {
            els = ez[pt * 64 + lr]; et = ez[pt * 64 + 32 + lr];
            a_sh = an_a[tc]; a_sc = an_a[Ch + tc]; a_isc = an_a[2 * Ch + tc]; a_l3 = an_a[3 * Ch + tc];
            b_sh = an_b[tc]; b_sc = an_b[Ch + tc]; b_isc = an_b[2 * Ch + tc]; b_l3 = an_b[3 * Ch + tc];
        }
        const uint32_t plane_bytes = (uint32_t)((size_t)M * Ch * 4);
        const srd_t sxa = make_srd(xa, plane_bytes), sxb = make_srd(xb, plane_bytes);
        const int rbase = mrow0 + 4 * (lane >> 5);
        const uint32_t voff = chok ? (uint32_t)(rbase * Ch + tau) * 4u : FWN_OOB;     // rows past M fall off the descriptor
        const srd_t sz = make_srd(save_z ? save_z : xb, save_z ? (uint32_t)((size_t)M * 2 * Ch * 4) : 0u);
        const uint32_t zoff = chok ? (uint32_t)(rbase * 2 * Ch + tau) * 4u : FWN_OOB;
        float lsum = 0.0f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            float xbv[16], xav[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t so = (uint32_t)((mi * 32 + acc_row_c(r)) * Ch * 4);
                if constexpr (LDS) {
                    const int cw = Ch < 32 ? Ch : 32;
                    const float* e = (const float*)lt + ((mrow0 & 63) + 4 * (lane >> 5) + acc_row_c(r)) * cw + (lr < cw ? lr : 0);
                    xbv[r] = e[0];
                    xav[r] = e[2048];
                } else {
                    xbv[r] = PRE ? pre[mi][r] : buf_load_f32(sxb, voff, so);
                    xav[r] = PRE ? pre[mi][16 + r] : buf_load_f32(sxa, voff, so);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t so = (uint32_t)((mi * 32 + acc_row_c(r)) * Ch * 4);
                const bool ok = chok && rbase + mi * 32 + acc_row_c(r) < M;
                const float ls = acc[mi][0][r] * els, t = acc[mi][1][r] * et;
                float ob, oa;
                if (!inverse) {
                    const float yb = (xbv[r] + b_sh) * b_sc;                  // ActNorm (model.py:86-94)
                    ob = (yb - t) * __expf(-ls);                               // model.py:134
                    oa = (xav[r] + a_sh) * a_sc;
                    lsum += ok ? (a_l3 + b_l3 - ls) : 0.0f;                    // model.py:135 + :80
                } else {
                    const float yb = xbv[r] * __expf(ls) + t;                  // model.py:156
                    ob = yb * b_isc - b_sh;                                    // ActNorm^-1 (model.py:97-102)
                    oa = xav[r] * a_isc - a_sh;
                }
                if constexpr (LDS) {
                    ob = bad ? __builtin_nanf("") : ob;
                    oa = bad ? __builtin_nanf("") : oa;
                }
                buf_store_f32(sxb, voff, so, ob);
                buf_store_f32(sxa, voff, so, oa);
                if (save_z) {
                    buf_store_f32(sz, zoff, 2 * so, acc[mi][0][r]);
                    buf_store_f32(sz, zoff, 2 * so + (uint32_t)(Ch * 4), acc[mi][1][r]);
                }
            }
        }
        if (partial) {          // one slot per (row tile, pair tile, wave row): fixed order, summed by prior_kernel
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) lsum += __shfl_xor(lsum, s);
            if constexpr (LDS) lsum = bad ? __builtin_nanf("") : lsum;
            const int tile_m = mrow0 >> 6, wm = (mrow0 >> 5) & 1;
            if (lane == 0) {
                partial[(tile_m * 4 + pt) * 2 + wm] = lsum;
                if (pt == 0)
                    for (int p2 = npt; p2 < 4; ++p2) partial[(tile_m * 4 + p2) * 2 + wm] = 0.0f;
            }
        }
    }
};

