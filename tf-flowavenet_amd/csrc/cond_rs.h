// Hoisted conditioning projection, register-streamed form (round 6): P[z] = c_a @ Wc[z]^T for all (flow, layer) matrices z of a
// block, fp32 [M][512] each (modules.py:120-121 lifted out of the flow chain: DESIGN.md section 3.4).
//
// The ring tile this replaces from 384 rows on (cond_batch_kernel, gemm_ring.h) reads BOTH operands from LDS - 32 FLOP per LDS
// byte with 64 x 64 wave tiles, i.e. the LDS read port at its peak when the matrix pipe is at its peak - and measured 0.67 - 0.85
// PFLOP/s on these shapes (63 GFLOP in 76 - 93 us).  Here, as in gate_rs.h / tail_rs.h:
//   * a workgroup = 32 MT rows (MT = 4, 3 or 2: whichever fills the chip's CUs most evenly at the shape) x all 512 columns of ONE
//     matrix; a wave owns 64 COLUMNS (two 32-column MFMA operands) x all rows: 2 MT accumulator tiles, 2 MT MFMAs per k-step and wave;
//   * the weights of a wave are its own: packed once in fragment order (cond_stream_pack_kernel: [z][wave][k-step][2][lane][8 bf16],
//     2 KiB per k-step and wave = two fully coalesced 16-byte loads per lane) and streamed straight into a ring of R k-steps of
//     registers by buffer loads issued from inline asm, waits counted by hand (CrsCount);
//   * the activations are the operand all waves share: each 64-column chunk of the tile's 128 rows is staged ONCE by LDS-DMA
//     (2 one-KiB pieces per wave) into one of four 16 KiB slots, three items ahead; every activation fragment a wave reads feeds
//     TWO MFMAs (its two column groups): 64 FLOP per LDS byte;
//   * one workgroup barrier per item (4 k-steps), behind each wave's own wait for its pieces of the item;
//   * few rows against a long K: the K range is dealt over `nsplit` workgroups per tile (split 0 writes P, the others partials
//     that fwn_launch_cond_reduce adds in ascending order: bit-reproducible, like the ring form's split).
// The product is NOT transposed (rows on accumulator rows, channels on lanes): a store instruction writes 128 contiguous bytes
// of two P rows.
#pragma once
#include "gate_rs.h"        // rs_wload, rs_vmwait, rs_static_for

// one 16-byte LDS-DMA per lane issued from inline asm (tail_rs.h's trs_dma16: issued through the builtin, hipcc drains vmcnt
// in front of the next LDS access): M0 = the LDS address, saved and restored inside the statement
__device__ __forceinline__ void crs_dma16(u32x4 srd, uint32_t voff, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ u32x4 crs_srd(const void* p, uint32_t bytes) {
    const unsigned long long b = (unsigned long long)(uintptr_t)p;
    return u32x4{(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) & 0xffffu,
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}

struct CondRsArgs {
    const bf16* ca;         // [M][cin]
    const bf16* ca_odd;     // NULL, or the plane the flows with an odd index read
    const bf16* Ws;         // fragment streams of this launch's matrices: [nz][8 waves][kcpad / 16][2][64][8]
    float* P;               // [nz][M][512]
    float* part;            // split s > 0 writes part + (s - 1) part_stride (laid out like P)
    long part_stride;
    int M, cin, kcpad, L, nsplit, nrt;      // nrt = ceil(M / (32 MT)) row tiles
};

// Vector-memory operations a wave issues AFTER a given one until it waits for it: the wave's program order -
//   pieces of items 0 .. 2, ring loads W(0 .. R-2) (two per k-step); then per item i, per k-step g = 4 i + l:
//   [l = 0: wait for the own pieces of item i, barrier] [wait for W(g)] [l = 0: the 2 pieces of item i + 3] the 2 loads of W(g + R - 1)
// walked at compile time; the kernel's constants are checked against it.
template <int R>
struct CrsCount {
    static constexpr int AHEAD = 3;
    // ops issued after the second load of W(g) when k-step g waits for it (kind 0) / after the second piece of item i at its barrier (kind 1)
    static constexpr int after(int kind, int key) {
        int count = -1;
#define CRS_OP(k_, a_) do { if (count >= 0) ++count; if (kind == (k_) && key == (a_)) count = 0; } while (0)
        for (int i = 0; i < AHEAD; ++i) { CRS_OP(2, -1); CRS_OP(1, i); }
        for (int g = 0; g < R - 1; ++g) { CRS_OP(2, -1); CRS_OP(0, g); }
        for (int i = 0; i < 64; ++i)
            for (int l = 0; l < 4; ++l) {
                const int g = 4 * i + l;
                if (l == 0 && kind == 1 && key == i) return count;
                if (kind == 0 && key == g) return count;
                if (l == 0) { CRS_OP(2, -1); CRS_OP(1, i + AHEAD); }
                CRS_OP(2, -1); CRS_OP(0, g + R - 1);
            }
#undef CRS_OP
        return -1;
    }
    // the count the kernel uses at position pos (0 .. 7) of a pair of items: one statement per position whatever the pair (a
    // branch on "first pair" around an asm wait that names ring registers is the persistent gate's race: gate_rs.h), so the
    // smaller of the first pair's and the steady state's value
    static constexpr int wait_kstep(int pos) {
        int n = after(0, pos);
        for (int pr = 1; pr < 6; ++pr) { const int m = after(0, 8 * pr + pos); if (m < n) n = m; }
        return n;
    }
    static constexpr int wait_item0 = after(1, 0);      // the first item's pieces: only the prologue behind them
    static constexpr int wait_item = after(1, 1) < after(1, 5) ? after(1, 1) : after(1, 5);
    static constexpr bool steady() {                    // the schedule repeats with the pair
        for (int pos = 0; pos < 8; ++pos)
            for (int pr = 2; pr < 6; ++pr) if (after(0, 8 * pr + pos) != after(0, 8 + pos)) return false;
        for (int i = 3; i < 12; ++i) if (after(1, i) != after(1, 3)) return false;
        return true;
    }
};

template <int N>
__device__ __forceinline__ void crs_wwait2(bf16x8& a, bf16x8& b) {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int I, int R>
__device__ __forceinline__ void crs_touch_all(bf16x8 (&wq)[R][2]) {
    if constexpr (I < R) {
        asm volatile("" : "+v"(wq[I][0]), "+v"(wq[I][1]));
        crs_touch_all<I + 1, R>(wq);
    }
}

template <int MT, int R = 8>
__global__ __launch_bounds__(512) void cond_rs_kernel(CondRsArgs p) {
    using C = CrsCount<R>;
    static_assert(R == 8, "a pair of items = 8 k-steps = one turn of the ring");
    static_assert(MT >= 2 && MT <= 4, "64-, 96- or 128-row tiles (the two pieces and the two loads of a k-step ride its first four MFMAs)");
    static_assert(C::steady(), "the vmcnt schedule must repeat from the second pair of items on");
    static_assert(C::wait_item0 >= 0 && C::wait_item >= C::wait_item0 && C::wait_item < 64, "piece waits");
    constexpr int BM = 32 * MT, SLOT = 128 * 128, AHEAD = C::AHEAD;      // a slot holds 16 pieces (2 per wave) whatever MT: rows past BM are zero rows
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * SLOT];
    const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_ptr_t)lds);
    const int wg = xcd_remap(blockIdx.x, gridDim.x);      // row tiles of a (matrix, split) are neighbours: one XCD's L2 serves their weights
    const int rt = wg % p.nrt, t1 = wg / p.nrt, sp = t1 % p.nsplit, z = t1 / p.nsplit;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int M = p.M, cin = p.cin, m0 = rt * BM;
    const bf16* ca = (p.ca_odd && ((z / p.L) & 1)) ? p.ca_odd : p.ca;
    // this workgroup's K range in 64-wide chunks (every split gets ceil(nch / nsplit) or the rest)
    const int nch = p.kcpad / 64, per = (nch + p.nsplit - 1) / p.nsplit;
    const int q0 = min(sp * per, nch), q1 = min(q0 + per, nch);
    const int npairs = (q1 - q0 + 1) >> 1;               // items past q1 are phantoms: zero pieces, out-of-range (zero) weights

    // ---- DMA pieces: piece j of this wave = slot rows 8 (wave + 8 j) .. + 7 of item i (chunk q0 + i) in slot i % 4
    const u32x4 srd_a = crs_srd(ca, (uint32_t)((size_t)M * cin * 2));
    auto issue_piece = [&](int i, int j) {
        const int pi = wave + 8 * j;
        const int jrow = 8 * pi + (lane >> 3);
        const int c = (lane & 7) ^ ((jrow >> 1) & 7);
        const int col = (q0 + i) * 64 + c * 8;
        const bool ok = (jrow < BM) & (m0 + jrow < M) & (col < cin) & (q0 + i < q1);
        crs_dma16(srd_a, ok ? (uint32_t)((m0 + jrow) * cin + col) * 2u : FWN_OOB, lds0 + (uint32_t)((i & 3) * SLOT + pi * 1024));
    };
    // ---- this wave's weight stream from k-step 4 q0 on: k-step g (relative) -> ring stage g % 8, two fragments
    const int nks = p.kcpad / 16;
    const unsigned long long wbase = (unsigned long long)(uintptr_t)p.Ws + ((unsigned long long)(z * 8 + wave) * nks + (unsigned long long)q0 * 4) * 2048ull;
    const u32x4 wsrd = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wbase),
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(wbase >> 32)) & 0xffffu,
                        (uint32_t)__builtin_amdgcn_readfirstlane((q1 - q0) * 4 * 2048), 0x00020000u};
    const uint32_t wl = (uint32_t)lane * 16u;
    bf16x8 wq[R][2];
    // k-step (8 pr + G): its byte offset = pr * 16384 + G * 2048: a 4-KiB window in the scalar offset, the rest an immediate
    auto issue_w = [&](auto G, auto F, int pairbase) {
        constexpr int g = decltype(G)::value, f = decltype(F)::value;
        constexpr int off = g * 2048 + f * 1024;
        rs_wload<off % 4096>(wq[g % R][f], wsrd, (uint32_t)(pairbase + (off / 4096) * 4096), wl);
    };

    // ---- prologue (order = CrsCount::after)
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) {
        issue_piece(i, 0);
        issue_piece(i, 1);
    }
    rs_static_for<R - 1>([&](auto G) {
        issue_w(G, std::integral_constant<int, 0>{}, 0);
        issue_w(G, std::integral_constant<int, 1>{}, 0);
    });

    f32x16 acc[MT][2];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int cg = 0; cg < 2; ++cg)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][cg][r] = 0.0f;
    // activation fragment of k-step ki of the item, time tile mi: slot row mi * 32 + lr, 16-byte piece 2 ki + lh (XOR-swizzled)
    const int xv = (lh ^ ((lr >> 1) & 7)) << 4;
    const int rb = lr * 128;
    bf16x8 hf[MT];

    for (int pr = 0; pr < npairs; ++pr) {
        const int pairbase = pr * (8 * 2048);
        rs_static_for<8>([&](auto POS) {
            constexpr int pos = decltype(POS)::value, e = pos >> 2, l = pos & 3;
            const int i = 2 * pr + e;
            const unsigned char* la = lds + (i & 3) * SLOT + rb;
            if constexpr (l == 0) {
                // own pieces of item i landed, then everybody's; behind the barrier every wave has left item i - 1 (its slot is item i + 3's)
                if (i == 0) rs_vmwait<C::wait_item0>(); else rs_vmwait<C::wait_item>();
                FWN_RING_BARRIER();
            }
            crs_wwait2<C::wait_kstep(pos)>(wq[pos][0], wq[pos][1]);
            if constexpr (l == 0) {
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) hf[mi] = *(const bf16x8*)(la + mi * 4096 + xv);
            }
            int kon = ((l + 1) * 32) ^ xv;
            asm volatile("" : "+v"(kon));
            __builtin_amdgcn_sched_barrier(0);
            // one MFMA per slot, each followed by at most one other instruction group
            rs_static_for<2 * MT>([&](auto S) {
                constexpr int s = decltype(S)::value, mi = s >> 1, cg = s & 1;
                acc[mi][cg] = mfma32(hf[mi], wq[pos][cg], acc[mi][cg]);
                if constexpr (cg == 1 && l < 3) hf[mi] = *(const bf16x8*)(la + mi * 4096 + kon);
                if constexpr (l == 0 && s < 2) issue_piece(i + AHEAD, s);
                if constexpr (s == 2) issue_w(std::integral_constant<int, pos + R - 1>{}, std::integral_constant<int, 0>{}, pairbase);
                if constexpr (s == 3) issue_w(std::integral_constant<int, pos + R - 1>{}, std::integral_constant<int, 1>{}, pairbase);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    }
    // the look-ahead loads and pieces past the K range (out of range: zeros) must have landed before the registers / the LDS
    // are anyone else's
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    crs_touch_all<0, R>(wq);

    // ---- epilogue: register r of tile (mi, cg) = row m0 + 32 mi + acc_row_c(r) + 4 lh, column 64 wave + 32 cg + lr
    float* out = (sp == 0 ? p.P : p.part + (size_t)(sp - 1) * p.part_stride) + (size_t)z * M * 512;
    const srd_t so = make_srd(out, (uint32_t)((size_t)M * 512 * 4));
#pragma unroll
    for (int cg = 0; cg < 2; ++cg) {
        const uint32_t voff = (uint32_t)((m0 + 4 * lh) * 512 + wave * 64 + cg * 32 + lr) * 4u;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                buf_store_f32(so, voff, (uint32_t)((mi * 32 + acc_row_c(r)) * 512 * 4), acc[mi][cg][r]);
    }
}

// Wc [512][kcpad] of matrix z (rows = packed-N columns of P) -> its fragment stream: out[((z 8 + wave) nks + g) 2 + cg][lane][8]:
// lane l holds row 64 wave + 32 cg + (l & 31), k 16 g + 8 (l >> 5) .. + 7.
__global__ void cond_stream_pack_kernel(const bf16* __restrict__ Wc, long w_stride, int kcpad, int nz, bf16* __restrict__ out) {
    const int nks = kcpad / 16;
    const long total = (long)nz * 8 * nks * 2 * 64;          // 16-byte pieces
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        long t = i >> 6;
        const int cg = (int)(t & 1); t >>= 1;
        const int g = (int)(t % nks); t /= nks;
        const int wave = (int)(t & 7), z = (int)(t >> 3);
        const int row = wave * 64 + cg * 32 + (lane & 31);
        ((uint4*)out)[i] = *(const uint4*)(Wc + (size_t)z * w_stride + (size_t)row * kcpad + g * 16 + 8 * (lane >> 5));
    }
}
