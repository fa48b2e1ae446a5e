// Shared device helpers for the gfx950 (MI355X / CDNA4) FloWaveNet kernels.
// wave = 64 lanes, MFMA = v_mfma_f32_32x32x16_bf16, LDS tiles are 64 bf16 (128 B) wide and
// XOR-swizzled so that ds_read_b128 fragment reads are conflict-free.  Tiles are staged
// HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no VGPR round trip, no branches;
// rows that must read as zero (clip edges of the dilated taps, M / K padding) get an
// out-of-range buffer offset, which the hardware range check turns into zeros.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __amdgpu_buffer_rsrc_t srd_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define FWN_HID 256          // residual/gate/skip channels (reference model.py:217)
#define FWN_BK 64            // K elements per staged chunk (128 B rows in LDS)
#define FWN_OOB 0x80000000u  // buffer offset beyond every descriptor's range (buffers < 2 GiB)

union Pack16 {               // 16 bytes = 8 bf16 = one MFMA A/B fragment
    uint4 u;
    u32x4 w;
    bf16x8 v;
    bf16 e[8];
};

__device__ __forceinline__ uint4 zero16() { return make_uint4(0u, 0u, 0u, 0u); }

// Raw buffer descriptor over [p, p + bytes): stride 0, hardware range check on the byte offset.
__device__ __forceinline__ srd_t make_srd(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
// (No VGPR-destination 16-byte buffer load helper on purpose: hipcc / ROCm 7.2 miscompiles element
// extraction from __builtin_amdgcn_raw_buffer_load_b128 - see DESIGN.md "Toolchain pitfalls".)
// 64 lanes x 16 B straight into LDS at (wave-uniform) dst + lane*16.
__device__ __forceinline__ void buf_load16_lds(srd_t s, uint32_t voff, unsigned char* dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
}
// The same with the non-temporal cache policy (aux = 2: nt).  For weight streams that ONE workgroup (or a handful) reads
// once per pass - the small tiles of the late blocks: the guide's nt-weights row (issue -> landed -18 % from cold caches);
// never for slices every CU re-reads from L2 (the big tiles).  FWN_NT_SMALL (developer switch) turns it on.
#ifndef FWN_NT_SMALL
#define FWN_NT_SMALL 0
#endif
__device__ __forceinline__ void buf_load16_lds_nt(srd_t s, uint32_t voff, unsigned char* dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s, (lds_ptr_t)dst, 16, voff, 0, 0, FWN_NT_SMALL ? 2 : 0);
}

// Epilogue stores / loads through a buffer descriptor: rows past the end of the matrix (last
// tile) fall outside the descriptor and are dropped / read as zero by the hardware range check,
// so no per-element branch; the per-register row offset rides in the SGPR soffset operand.
__device__ __forceinline__ void buf_store_bf16(srd_t s, uint32_t voff, uint32_t soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (bf16)v), s, voff, soff, 0);
}
__device__ __forceinline__ void buf_store_u8(srd_t s, uint32_t voff, uint32_t soff, unsigned int v) {
    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)v, s, voff, soff, 0);
}
__device__ __forceinline__ void buf_store_f32(srd_t s, uint32_t voff, uint32_t soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), s, voff, soff, 0);
}
__device__ __forceinline__ float buf_load_f32(srd_t s, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(s, voff, soff, 0));
}
__device__ __forceinline__ float buf_load_bf16(srd_t s, uint32_t voff, uint32_t soff) {
    const unsigned int u = (unsigned int)__builtin_amdgcn_raw_buffer_load_b16(s, voff, soff, 0) << 16;
    return __builtin_bit_cast(float, u);
}
// row offset (in rows) of accumulator register r, without the lane-dependent 4*(lane>>5) part
__device__ __forceinline__ constexpr int acc_row_c(int r) { return (r & 3) + 8 * (r >> 2); }

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

// tanh(f) * sigmoid(g) with two v_exp_f32 and one v_rcp_f32:
//   a = e^-2f, b = e^-g  ->  (1 - a) / ((1 + a) (1 + b)).
// a is capped at 2^40 so it stays finite (tanh is -1 to fp32 long before that);
// large positive f, g underflow a, b to 0, which is the right limit.
// The gate weights and biases arrive pre-multiplied (packing.py) so the accumulators already hold
// the exponents: u = -2 log2(e) f, v = -log2(e) g.  Two outputs per call so the adds and
// multiplies pair up into v_pk_*_f32 (the epilogue is VALU-bound: 3 quarter-rate ops per output).
// b may overflow to +inf (-> rcp 0 -> 0, the right limit while a is finite), so only u is capped.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gated_unit2(f32x2 u, f32x2 v) {
    f32x2 a, b;
    a.x = __builtin_amdgcn_exp2f(fminf(u.x, 40.0f));
    a.y = __builtin_amdgcn_exp2f(fminf(u.y, 40.0f));
    b.x = __builtin_amdgcn_exp2f(v.x);
    b.y = __builtin_amdgcn_exp2f(v.y);
    const f32x2 den = (1.0f + a) * (1.0f + b);
    f32x2 r;
    r.x = __builtin_amdgcn_rcpf(den.x);
    r.y = __builtin_amdgcn_rcpf(den.y);
    return (1.0f - a) * r;
}

// Two fp32 -> two OCP e4m3 bytes (v_cvt_pk_fp8_f32, round to nearest even), saturating at +-448 (e4m3fn has no
// infinity: without the clamp an overflow would become NaN).  Low byte = a.
__device__ __forceinline__ unsigned int pack_e4m3x2(float a, float b) {
    a = fminf(fmaxf(a, -448.0f), 448.0f);
    b = fminf(fmaxf(b, -448.0f), 448.0f);
    return (unsigned int)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
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
// Generic bf16 MFMA GEMM core: block tile (64*MI) x 128, 256 threads = 4 waves laid out
// 2 (M) x 2 (N); each wave owns MI x 2 tiles of 32x32.
// `Prob` supplies, per K-chunk, a buffer descriptor + per-lane byte offset for the A rows
// (possibly a shifted / zero-padded view) and the B rows, and the fused epilogue.
// A problem whose A operand needs arithmetic on the way in (front conv: fp32 -> ActNorm ->
// bf16) sets A_DMA = false and returns the 16 packed bytes instead.
// Pipeline: 2 LDS buffers, the DMA of chunk q+1 is in flight while chunk q is multiplied,
// one barrier per chunk.
// ---------------------------------------------------------------------------
template <int MI, class Prob>
__device__ __forceinline__ void gemm128_body(const Prob& p, int tile_m, int tile_n) {
    constexpr int BM = 64 * MI;
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = 128 * 128;
    constexpr int NA = BM / 32;      // A rows handled per lane per chunk
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * (A_BYTES + B_BYTES)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int m0 = tile_m * BM, n0 = tile_n * 128;

    // DMA slot j of this wave covers tile rows [8*(wave + 4j), +8); lane -> (row, LDS slot);
    // the global chunk it fetches is the slot XOR-ed back through the swizzle.
    int trow[4], tc8[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        trow[j] = 8 * (wave + 4 * j) + (lane >> 3);
        tc8[j] = (lane & 7) ^ ((trow[j] >> 1) & 7);
    }
    typename Prob::RowCtx rc[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) rc[j] = p.row_ctx(m0 + trow[j]);

    const int nq = p.nchunks();
    uint4 ra[NA];   // only used when !Prob::A_DMA

    auto stage = [&](int q, int buf) {
        typename Prob::ChunkCtx cc = p.chunk_ctx(q);
        unsigned char* la = lds + buf * (A_BYTES + B_BYTES);
        unsigned char* lb = la + A_BYTES;
        if constexpr (Prob::A_DMA) {
            const srd_t sa = p.a_srd(cc);
#pragma unroll
            for (int j = 0; j < NA; ++j)
                buf_load16_lds(sa, p.a_voff(cc, rc[j], tc8[j]), la + (wave + 4 * j) * 1024);
        } else {
#pragma unroll
            for (int j = 0; j < NA; ++j) ra[j] = p.load_a(cc, rc[j], tc8[j]);
        }
        const srd_t sb = p.b_srd(cc);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            buf_load16_lds(sb, p.b_voff(cc, n0 + trow[j], tc8[j]), lb + (wave + 4 * j) * 1024);
    };
    auto write_a = [&](int buf) {
        if constexpr (!Prob::A_DMA) {
            unsigned char* la = lds + buf * (A_BYTES + B_BYTES);
#pragma unroll
            for (int j = 0; j < NA; ++j) *(uint4*)(la + (wave + 4 * j) * 1024 + lane * 16) = ra[j];
        }
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    stage(0, 0);
    write_a(0);
    __syncthreads();
    for (int q = 0; q < nq; ++q) {
        const bool more = (q + 1 < nq);
        if (more) stage(q + 1, (q + 1) & 1);
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
        if (more) write_a((q + 1) & 1);
        __syncthreads();   // drains the LDS-DMA of chunk q+1 (vmcnt) and fences the buffer swap
    }
    p.template epilogue<MI>(acc, m0 + wm * 32 * MI, n0 + wn * 64, lane);
}
