// Shared device helpers for the gfx950 (MI355X / CDNA4) FloWaveNet kernels.
// wave = 64 lanes, MFMA = v_mfma_f32_32x32x16_bf16, LDS tiles are 64 bf16 (128 B)
// wide and XOR-swizzled so that ds_read_b128 fragment reads are conflict-free.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define FWN_HID 256          // residual/gate/skip channels (reference model.py:217)
#define FWN_BK 64            // K elements per staged chunk (128 B rows in LDS)

union Pack16 {               // 16 bytes = 8 bf16 = one MFMA A/B fragment
    uint4 u;
    bf16x8 v;
    bf16 e[8];
};

__device__ __forceinline__ uint4 zero16() { return make_uint4(0u, 0u, 0u, 0u); }

// Byte offset of 16-byte chunk `c8` (0..7) of row `row` in a [rows][64] bf16 LDS tile.
// Two 128-B rows share one 256-B bank row; XOR with (row>>1)&7 spreads 16 rows over the
// 16 slots of a bank row ("LDS image").
__device__ __forceinline__ int lds_off64(int row, int c8) {
    return row * 128 + ((c8 ^ ((row >> 1) & 7)) << 4);
}

// Byte offset of 16-byte chunk `c` (0..31) of row `row` in a [rows][256] bf16 LDS tile
// (512-B rows: every row starts on the same bank, so XOR with row&15).
__device__ __forceinline__ int lds_off256(int row, int c) {
    return row * 512 + (((c & 15) ^ (row & 15)) << 4) + ((c >> 4) << 8);
}

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// Row of accumulator register `r` (0..15) for this lane inside a 32x32 tile
// (C/D layout: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).
__device__ __forceinline__ int acc_row(int r, int lane) {
    return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

__device__ __forceinline__ float fast_sigmoid(float x) {
    return __frcp_rn(1.0f + __expf(-x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    // tanh(x) = 1 - 2 / (1 + exp(2x)); saturates correctly for |x| large.
    return 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * x));
}

// XCD-aware, bijective remap of the linear workgroup id: workgroups that the
// dispatcher deals to the same XCD (id % 8) receive consecutive tile ids, so tiles that
// share A rows (the N-tiles of one M-tile) hit the same L2.  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// ---------------------------------------------------------------------------
// Generic bf16 MFMA GEMM core: block tile (32*MI*2) x 128, 256 threads = 4 waves
// laid out 2 (M) x 2 (N); each wave owns MI x 2 tiles of 32x32.
// `Prob` supplies the K-chunk sources (A may be a shifted / zero-padded / converted
// view) and the fused epilogue.  Register-staged, LDS double-buffered, one barrier
// per K-chunk.
// ---------------------------------------------------------------------------
template <int MI, class Prob>
__device__ __forceinline__ void gemm128_body(const Prob& p, int tile_m, int tile_n) {
    constexpr int BM = 64 * MI;
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = 128 * 128;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (A_BYTES + B_BYTES)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int c8 = tid & 7, r0 = tid >> 3;
    const int m0 = tile_m * BM, n0 = tile_n * 128;

    typename Prob::RowCtx rc[2 * MI];
#pragma unroll
    for (int i = 0; i < 2 * MI; ++i) rc[i] = p.row_ctx(m0 + r0 + 32 * i);

    uint4 ra[2 * MI], rb[4];
    const int nq = p.nchunks();

    auto gload = [&](int q) {
        typename Prob::ChunkCtx cc = p.chunk_ctx(q);
#pragma unroll
        for (int i = 0; i < 2 * MI; ++i) ra[i] = p.load_a(cc, rc[i], c8);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = p.load_b(cc, n0 + r0 + 32 * i, c8);
    };
    auto lwrite = [&](int buf) {
        unsigned char* la = lds + buf * (A_BYTES + B_BYTES);
        unsigned char* lb = la + A_BYTES;
#pragma unroll
        for (int i = 0; i < 2 * MI; ++i) *(uint4*)(la + lds_off64(r0 + 32 * i, c8)) = ra[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *(uint4*)(lb + lds_off64(r0 + 32 * i, c8)) = rb[i];
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    gload(0);
    lwrite(0);
    __syncthreads();
    for (int q = 0; q < nq; ++q) {
        const bool more = (q + 1 < nq);
        if (more) gload(q + 1);
        const unsigned char* la = lds + (q & 1) * (A_BYTES + B_BYTES);
        const unsigned char* lb = la + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 af[MI], bfr[2];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                af[mi] = *(const bf16x8*)(la + lds_off64(wm * 32 * MI + mi * 32 + lr, kk * 2 + lh));
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                bfr[ni] = *(const bf16x8*)(lb + lds_off64(wn * 64 + ni * 32 + lr, kk * 2 + lh));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma32(af[mi], bfr[ni], acc[mi][ni]);
        }
        if (more) lwrite((q + 1) & 1);
        __syncthreads();
    }
    p.template epilogue<MI>(acc, m0 + wm * 32 * MI, n0 + wn * 64, lane);
}
