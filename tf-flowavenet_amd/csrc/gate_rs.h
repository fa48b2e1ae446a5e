// Gated dilated layer, blocks 0-1 form (round 4): activations RESIDENT in LDS, weights streamed to registers, the two
// waves of a SIMD one item apart.
//
// gate_halo.h runs 16 waves of 64 x 64 per 256 x 256 tile: every (tap, slice) step ends in a workgroup barrier, the
// four waves of a SIMD run their MFMA groups one after the other and the early finishers wait (2 800 cycles per
// 2 048-cycle step), both operands are read from LDS by every wave, and the matrix pipe idles
// through the whole tanh / sigmoid epilogue.  This kernel turns the tile around:
//   * EIGHT waves, two per SIMD.  A wave owns ALL 256 time rows of the tile x 16 channels (filter + gate rows = one
//     32-row MFMA operand): 8 accumulator tiles of 32 x 32 = 128 registers.  The product is computed transposed
//     (channels on accumulator rows, time on lanes, like tail_chain.h): a lane ends up with 4 consecutive channels of one
//     time row per register group -> 8-byte output stores.
//   * The weights of a wave are its own (no other wave of the workgroup multiplies by them), so they do not go through
//     LDS at all: they are packed once in FRAGMENT order (gate_stream_pack_kernel: [channel group][k-step][lane][8 bf16],
//     1 KiB per MFMA operand = one fully coalesced 16-byte load per lane) and streamed straight into a ring of R k-steps
//     of registers by global loads issued from inline asm (hipcc would drain the whole queue at the first use of an
//     ordinary load while an LDS-DMA is in flight, guide section 5 trap (b)); waits are counted by hand (RsCount: a
//     compile-time walk over the wave's own instruction order).  No barrier orders them.
//   * The activations are the operand all waves share: each 64-channel slice of rows [m0 - dil, m0 + 256 + dil) is staged
//     ONCE by LDS-DMA (tap sharing as in gate_halo.h) into one of four 40 KiB slots; the conditioning chunks go through
//     the same slots.  One workgroup barrier per item (slice or conditioning chunk), 7 per block-0 tile instead of 14.
//   * Waves 4-7 (the SIMD partners of waves 0-3) run ONE ITEM BEHIND waves 0-3: between barriers n and n + 1 the leaders
//     multiply item n and the laggers item n - 1.  The leaders' epilogue (transcendental-bound VALU work) runs under the
//     laggers' last slice and the leaders' first slice under the laggers' DMA issue: the matrix pipe of a SIMD always has
//     a wave that feeds it.  Item order: slice 0, the conditioning chunks, slices 1-3 (a long item at either end).
// LDS ring: item i lives in slot i % 4; its pieces (5 per wave, 8 rows each; pieces past the tile are out of range =
// zero rows, the zero row of the clip mask among them) are issued behind barrier i - 2 (which proves the laggers have
// left item i - 4) and every wave waits for its own pieces of item i before it arrives at barrier i.
#pragma once
#include "gemm_ring.h"
#include <type_traits>

// MT: 32-row time tiles per wave (8: 256-row workgroup tiles; 4: 128-row tiles for row counts at which 256-row tiles would
// leave CUs empty; 2: 64-row tiles, round 6 - block 3 of the 8-clip pass, 8 064 rows x cin 640, where the 128 x 128 tap-sharing
// tile ran at 0.22 of the MFMA peak).  DMA pieces per wave and item: 8 waves x PP x 8 rows must hold the tile, its halo and a zero row.
__host__ __device__ constexpr int rs_pp(int mt) { return mt == 8 ? 5 : mt == 4 ? 3 : 2; }      // 320- / 192- / 128-row slots
__host__ __device__ constexpr int rs_zrow(int mt) { return 32 * mt + 24; }       // a slot row no item ever stages

#ifndef FWN_RABL
#define FWN_RABL 0               // developer ablation (wrong results): 1 no weight loads after the prologue, 2 no epilogue, 3 no item barriers,
                                 // 4 no DMA pieces after the prologue, 5 no fragment reads after the first
#endif

struct GateRsArgs {
    const bf16* h;        // [M][256]
    const bf16* ca;       // [M][cin]
    const bf16* Wg;       // fragment stream [16 channel groups][48 + NKC k-steps in plan order][64][8]
    const float* bias;    // [512] packed-N order
    bf16* o;              // [M][256]
    int M, Ti, dil, cin;
#ifdef FWN_RS_STAMP
    unsigned long long* stamps;   // diagnostic build (tools/bench_gate_rs.hip): [workgroup][wave][32] s_memtime / s_memrealtime
#endif
    // fwn_gate_clock (the CLK = true instantiation; the product launches never read it): [workgroup][8 waves][4] =
    // s_memtime at wave start / end, s_memrealtime at wave start / end.  Nothing else reads these words.
    unsigned long long* clk = nullptr;
};
#if defined(FWN_RS_STAMP) && !defined(FWN_RS_CHECK)
#define RS_STAMP(i) do { if (lane == 0) p.stamps[((size_t)blockIdx.x * 8 + wave) * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define RS_STAMP_RT(i) do { if (lane == 0) p.stamps[((size_t)blockIdx.x * 8 + wave) * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RS_STAMP(i) do { } while (0)
#define RS_STAMP_RT(i) do { } while (0)
#endif

// one 16-byte load per lane, hidden from hipcc's wait bookkeeping; the caller counts vmcnt.
// * A BUFFER load (MUBUF), like the LDS-DMA pieces and the epilogue stores: a counted s_waitcnt vmcnt(N) is only a wait
//   for a particular operation if the operations retire in issue order, and that holds among MUBUF operations.  The first
//   persistent form of this kernel loaded the weights with global_load_dwordx4 (the FLAT family, segment global): with
//   out-of-range pieces (which complete at once) and stores between the loads, ~40 % of the overlapped steps came back
//   wrong while every wait had at least its static count of younger operations in program order (FWN_RS_CHECK build):
//   younger MUBUF operations were retiring ahead of older global loads.  (The guide's "flat_*: out of order".)
// * s_nop 4: the descriptor and the offset are scalar arithmetic on kernel arguments, but under SGPR pressure hipcc parks
//   scalars in VGPR lanes and brings them back with v_readlane_b32 right in front of the statement - a VALU write of an
//   SGPR that a VMEM instruction reads needs 5 wait states, and hipcc pads nothing inside an asm (guide section 5.7 item 2):
//   without the pad the NKC = 10 instantiation loaded through garbage bases (memory access faults at random addresses).
template <int OFF>
__device__ __forceinline__ void rs_wload(bf16x8& dst, u32x4 srd, uint32_t soff, uint32_t voff) {
    static_assert(OFF >= 0 && OFF < 4096, "12-bit unsigned immediate");
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(dst) : "v"(voff), "s"(srd), "s"(soff), "i"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void rs_wwait(bf16x8& a) {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void rs_vmwait() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

// tanh(f) sigmoid(g) from the exponents u = -2 log2(e) f, v = -log2(e) g the accumulators hold (common.h gated_unit2):
// a = 2^u, b = 2^v -> (1 - a) / ((1 + a)(1 + b)).  Scalar fp32 on purpose: this epilogue is paid for at one or two waves
// per SIMD, where every v_pk_*_f32 costs the issue time of about four plain VALU instructions (guide, price of fillers),
// and the cap is one v_med3_f32 (fminf on a value fresh from an accumulator costs a canonicalising v_max_f32 as well).
// Same operations in the same order as gated_unit2: identical results.
__device__ __forceinline__ float rs_gated1(float u, float v) {
    const float a = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(u, 40.0f, -3.0e38f));
    const float b = __builtin_amdgcn_exp2f(v);
    const float r = __builtin_amdgcn_rcpf((1.0f + a) * (1.0f + b));
    return (1.0f - a) * r;
}

// every ring stage made opaque at this point (behind a wait: no use of a stage is scheduled above it)
template <int I, int R>
__device__ __forceinline__ void rs_touch_all(bf16x8 (&wq)[R]) {
    if constexpr (I < R) {
        asm volatile("" : "+v"(wq[I]));
        rs_touch_all<I + 1, R>(wq);
    }
}

template <int I, int N, class F>
__device__ __forceinline__ void rs_static_for_impl(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        rs_static_for_impl<I + 1, N>(f);
    }
}
template <int N, class F>
__device__ __forceinline__ void rs_static_for(F&& f) { rs_static_for_impl<0, N>(f); }

// Packed row (packing.py gate order: row cg*64 + kind*32 + c of channel cg*32 + c) of accumulator row rho of channel
// group grp (16 channels): a lane's registers 8 pp + j (filter) and 8 pp + 4 + j (gate) belong to the same channel
// j + 4 lh + 8 pp, i.e. fragment rows 0-7 are the filter rows of channels 0-7, 8-15 their gate rows, 16-23 the filter
// rows of channels 8-15, 24-31 their gate rows.
__device__ __host__ __forceinline__ constexpr int rs_packed_row(int grp, int rho) {
    const int ch = grp * 16 + (rho & 7) + 8 * (rho >> 4), kind = (rho >> 3) & 1;
    return (ch >> 5) * 64 + kind * 32 + (ch & 31);
}

// The K loop of one tile as a list of items.  NKC: conditioning k-steps of 16 (cin / 16 rounded up).
template <int NKC>
struct RsPlan {
    static constexpr int NC = (NKC + 3) / 4;                 // conditioning chunks of up to 64 columns
    static constexpr int NI = 4 + NC;                        // items: slice 0, chunks 0 .. NC-1, slices 1 .. 3
    static constexpr int NK = 48 + NKC;                      // k-steps of 16
    static constexpr bool is_slice(int i) { return i == 0 || i > NC; }
    static constexpr int slice_of(int i) { return i == 0 ? 0 : i - NC; }
    static constexpr int chunk_of(int i) { return i - 1; }
    static constexpr int item_ks(int i) { return is_slice(i) ? 12 : (chunk_of(i) < NC - 1 ? 4 : NKC - 4 * (NC - 1)); }
    static constexpr int item_first(int i) { int g = 0; for (int j = 0; j < i; ++j) g += item_ks(j); return g; }
    static constexpr int item_of(int g) { int i = 0; while (g >= item_ks(i)) { g -= item_ks(i); ++i; } return i; }
    // column of k-step g in the packed weights: < 768: Wd column tap*256 + kc*64 + ki*16; else 768 + Wc column
    static constexpr int kcol(int g) {
        const int i = item_of(g), l = g - item_first(i);
        return is_slice(i) ? (l >> 2) * 256 + slice_of(i) * 64 + (l & 3) * 16 : 768 + chunk_of(i) * 64 + l * 16;
    }
};

// Vector-memory operations a wave has issued AFTER a given one, at the point where it waits for that one: the wave's
// program order walked at compile time, over the prologue and three tiles of a persistent workgroup (tile 0 = the first,
// whose operands come from the prologue; tile 1 = every later one; tile 2 only supplies tile 1's look-ahead).
// Global item J = tile * NI + item lives in slot J % 4; global barrier J is the one behind which the LEADERS open item J.
//   every wave      : pieces of item 0 first.
//   leaders (LAG 0) : ring loads W(0, 0 .. R-2), [barrier 0]; per tile: before item i > 0 [barrier J]; under the first k-step
//                     of item J the pieces of item J + 2 (slots 0 .. PP-1, one per MFMA; in tile 0 under k-step 0 those of
//                     item 1 in front of them); behind the K loop [barrier (tile + 1) NI] - it lets the laggers into their
//                     last item and is the barrier in front of the leaders' next item 0 -, then the epilogue.
//   laggers (LAG 1) : [barrier 0], pieces of items 1 and 2, ring loads; per tile: before item J [barrier J + 1]; under its first
//                     k-step the pieces of item J + 3.
//   weights         : under k-step g (slot 1, behind that slot's pieces) the load into the ring stage k-step g - 1 has left:
//                     W(tile, g + R - 1) while that exists, else the NEXT tile's W(tile + 1, (g - 1) % R) if that stage is one
//                     of the R - 1 the next tile starts from; the stage the last k-step leaves is loaded behind the K loop.
// Pieces and loads beyond the last tile are issued all the same (out-of-range pieces write zeros into a free slot, the loads
// re-read this tile's stream): the counts must not depend on whether a next tile exists.
// PERSIST = false: one tile per workgroup - nothing is issued for a next tile (the product; the walk then only ever looks at tile 0).
template <int NKC, int R, bool LAG, int MT, bool PERSIST>
struct RsCount {
    using P = RsPlan<NKC>;
    static constexpr int PP = rs_pp(MT), NI = P::NI, NK = P::NK, AHEAD = LAG ? 3 : 2;
    static_assert(PP <= MT, "one piece per MFMA slot");
    // target: tk 0 = weights W(ta, tb) / 1 = last piece of global item ta; query: qk 0 = the wait in front of k-step (qa, qb) /
    // 1 = the arrival at global barrier qa
    static constexpr int walk(int tk, int ta, int tb, int qk, int qa, int qb) {
        int count = -1, result = -1;
        bool done = false;
#define RS_W(tile, k) do { if (count >= 0) ++count; if (tk == 0 && ta == (tile) && tb == (k)) count = 0; } while (0)
#define RS_P(J) do { for (int j_ = 0; j_ < PP; ++j_) { if (count >= 0) ++count; if (tk == 1 && ta == (J) && j_ == PP - 1) count = 0; } } while (0)
#define RS_P1(J, j_) do { if (count >= 0) ++count; if (tk == 1 && ta == (J) && (j_) == PP - 1) count = 0; } while (0)
#define RS_QB(b) do { if (!done && qk == 1 && qa == (b)) { result = count; done = true; } } while (0)
#define RS_QK(tile, g) do { if (!done && qk == 0 && qa == (tile) && qb == (g)) { result = count; done = true; } } while (0)
        RS_P(0);
        if (!LAG) {
            for (int k = 0; k < R - 1; ++k) RS_W(0, k);
            RS_QB(0);
        } else {
            RS_QB(0);
            RS_P(1);
            RS_P(2);
            for (int k = 0; k < R - 1; ++k) RS_W(0, k);
        }
        for (int tile = 0; tile < 3; ++tile) {
            for (int i = 0; i < NI; ++i) {
                const int J = tile * NI + i;
                if (LAG) RS_QB(J + 1);
                else if (i > 0) RS_QB(J);
                for (int l = 0; l < P::item_ks(i); ++l) {
                    const int g = P::item_first(i) + l;
                    RS_QK(tile, g);
                    for (int slot = 0; slot < MT; ++slot) {
                        if (!LAG && tile == 0 && g == 0 && slot < PP) RS_P1(1, slot);
                        if (l == 0 && slot < PP && (PERSIST || i + AHEAD < NI)) RS_P1(J + AHEAD, slot);
                        if (slot == 1) {
                            if (g + R - 1 < NK) RS_W(tile, g + R - 1);
                            else if (PERSIST && (g - 1) % R <= R - 2) RS_W(tile + 1, (g - 1) % R);
                        }
                    }
                }
            }
            if (PERSIST && (NK - 1) % R <= R - 2) RS_W(tile + 1, (NK - 1) % R);
            if (!LAG) RS_QB((tile + 1) * NI);
        }
#undef RS_W
#undef RS_P
#undef RS_P1
#undef RS_QB
#undef RS_QK
        return result;
    }
    // tile: 0 = the workgroup's first tile, 1 = any later one
    static constexpr int wait_kstep(int tile, int g) { return walk(0, tile, g, 0, tile, g); }
    // own pieces of global item b (one tile per workgroup: the barriers past the last item have nothing to wait for)
    static constexpr int wait_barrier(int b) { return !PERSIST && b >= NI ? -1 : walk(1, b, 0, 1, b, 0); }
    static constexpr bool steady_ok(int g) { return !PERSIST || walk(0, 1, g, 0, 1, g) == walk(0, 2, g, 0, 2, g); }   // the schedule is periodic
};

// One wave of a PERSISTENT workgroup: tiles t0, t0 + tstep, ... < ntiles (tile_n = tile & 1 is the same for all of them:
// tstep is even, so the wave's weight stream is too).
template <int NKC, int R, bool LAG, int MT, bool PERSIST, class Stamp>
__device__ __forceinline__ void gate_rs_wave(const GateRsArgs& p, unsigned char* lds, int wave, int lane, int t0, int tstep, int ntiles,
                                             Stamp&& stamp) {
    using P = RsPlan<NKC>;
    using C = RsCount<NKC, R, LAG, MT, PERSIST>;
    constexpr int NK = P::NK, NI = P::NI, PP = rs_pp(MT), BM = 32 * MT, ZROW = rs_zrow(MT), AHEAD = C::AHEAD;
    constexpr int SLOT = 8 * PP * 1024;
    const int lr = lane & 31, lh = lane >> 5;
    const int dil = p.dil, M = p.M, cin = p.cin;
    const int grp = (t0 & 1) * 8 + wave;              // channel group: 16 channels
    constexpr int FAR = 0x3f000000;                   // "m0" of a tile that does not exist: every row out of range

#ifdef FWN_RS_CHECK       // diagnostic build: dynamic count of vector-memory issues against the static schedule (RsCount)
    int vm_issued = 0, w_idx[R] = {}, chk_tile = 0;
#define RS_CHK_ISSUE() (++vm_issued)
#define RS_CHK_W(st) (w_idx[st] = vm_issued++)
#define RS_CHK_WAIT(st, g_, N) do { const int young_ = vm_issued - 1 - w_idx[st]; \
        if (young_ < (N) && lane == 0) { const unsigned i_ = atomicAdd((unsigned*)p.stamps, 1u); if (i_ < 60) { unsigned long long* r_ = p.stamps + 8 + 4 * i_; \
            r_[0] = (unsigned long long)(LAG ? 1000 : 0) + wave; r_[1] = (unsigned long long)chk_tile * 1000 + (g_); r_[2] = (unsigned long long)young_; r_[3] = (unsigned long long)(N); } } } while (0)
#else
#define RS_CHK_ISSUE() do { } while (0)
#define RS_CHK_W(st) do { } while (0)
#define RS_CHK_WAIT(st, g_, N) do { } while (0)
#endif
    // ---- DMA pieces: piece j of this wave = slot rows 8 (wave + 8 j) .. + 7; offsets are computed at issue time (a few
    // VALU operations per piece under the MFMAs) rather than kept in registers across the K loop.  ITEM: item of the tile
    // whose first row is m0_ (this tile's or the next one's); slot: (global item) % 4.
    const uint32_t hbytes = (uint32_t)((size_t)M * FWN_HID * 2);
    const uint32_t cbytes = (uint32_t)((size_t)M * cin * 2);
    const int prow = 8 * wave + (lane >> 3);          // slot row of piece 0
    auto issue_piece = [&](auto ITEM, int slot, int m0_, int j) {
        constexpr int item = decltype(ITEM)::value;
        static_assert(item < NI, "no such item");
        unsigned char* dst = lds + slot * SLOT + (wave + 8 * j) * 1024;
        RS_CHK_ISSUE();
        const int jrow = prow + 64 * j;
        const int c = (lane & 7) ^ ((jrow >> 1) & 7);
        if constexpr (P::is_slice(item)) {
            const int g = m0_ - dil + jrow;
            const bool ok = (jrow < BM + 2 * dil) & ((unsigned)g < (unsigned)M);
            buf_load16_lds(make_srd(p.h, hbytes), ok ? (uint32_t)(g * (FWN_HID * 2) + c * 16 + P::slice_of(item) * 128) : FWN_OOB, dst);
        } else {
            const int col = P::chunk_of(item) * 64 + c * 8;
            const bool ok = (jrow < BM) & (m0_ + jrow < M) & (col < cin);
            buf_load16_lds(make_srd(p.ca, cbytes), ok ? (uint32_t)((m0_ + jrow) * cin + col) * 2u : FWN_OOB, dst);
        }
    };

    // ---- weight stream of this wave (the same for every tile): k-step g -> ring stage g % R
    const unsigned long long wbase = (unsigned long long)(uintptr_t)p.Wg + (unsigned long long)grp * NK * 1024;
    const u32x4 wsrd = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wbase),
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(wbase >> 32)) & 0xffffu,      // stride 0
                        (uint32_t)(NK * 1024), 0x00020000u};
    const uint32_t wl = (uint32_t)lane * 16u;
    bf16x8 wq[R];
    auto issue_w = [&](auto G) {                      // 4 KiB windows: the window in the scalar offset, the rest an immediate
        constexpr int g = decltype(G)::value;
        rs_wload<(g * 1024) % 4096>(wq[g % R], wsrd, (uint32_t)(((g * 1024) / 4096) * 4096), wl);
        RS_CHK_W(g % R);
    };

    // barrier number b (mod the tile period; `first`: this is the workgroup's first tile): every wave first waits for its own
    // pieces of global item b, the item the LEADERS open behind it.  B = barrier index within tiles 0 / 1 of RsCount::walk.
    auto barrier = [&](auto B0, auto B1, bool first, int sid) {
        constexpr int b0 = decltype(B0)::value, b1 = decltype(B1)::value;
        constexpr int n0 = b0 >= 0 ? C::wait_barrier(b0) : -2, n1 = PERSIST || b0 < 0 ? C::wait_barrier(b1) : n0;
#if defined(FWN_RS_SAFE) && (FWN_RS_SAFE & 2)
        rs_vmwait<0>();
#else
        if constexpr (n1 < 0 && (n0 < 0 || !PERSIST)) { }                          // nothing to wait for
        else if constexpr (n0 == n1 || b0 < 0) rs_vmwait<n1>();
        else { if (first) rs_vmwait<n0 >= 0 ? n0 : 0>(); else rs_vmwait<n1>(); }
#endif
        stamp(sid);
        if (FWN_RABL != 3) FWN_RING_BARRIER();       // (nothing is in flight here by construction: the last k-step of an item prefetches nothing)
        stamp(sid + 1);
    };

    // ---- prologue (order = RsCount::walk): the pieces of item 0 first, everywhere; the laggers' other issues wait until
    // barrier 0 has let the leaders start
    int m0 = (t0 >> 1) * BM;
#pragma unroll
    for (int j = 0; j < PP; ++j) issue_piece(std::integral_constant<int, 0>{}, 0, m0, j);
    if constexpr (!LAG) {
        rs_static_for<R - 1>([&](auto G) { issue_w(G); });
        barrier(std::integral_constant<int, -1>{}, std::integral_constant<int, 0>{}, true, 1);
    } else {
        barrier(std::integral_constant<int, -1>{}, std::integral_constant<int, 0>{}, true, 1);
        rs_static_for<2>([&](auto I) {
            constexpr int it = decltype(I)::value + 1;
            static_assert(it < NI, "at least three items");
#pragma unroll
            for (int j = 0; j < PP; ++j) issue_piece(std::integral_constant<int, it>{}, it, m0, j);
        });
        rs_static_for<R - 1>([&](auto G) { issue_w(G); });
    }

    bool first = true;
    int jb = 0;                                       // (tile index within this workgroup * NI) % 4: slot of the tile's item 0
    for (int t = t0;; t += tstep) {
        const bool has_next = PERSIST && t + tstep < ntiles;
#ifdef FWN_RS_REALDUMMY
        const int m0n = has_next ? ((t + tstep) >> 1) * BM : m0;
#else
        const int m0n = has_next ? ((t + tstep) >> 1) * BM : FAR;
#endif
        // ---- activation fragment addresses: view v = tap 0..2 (slot row i + tap*dil, clip mask) or 3 (conditioning: row i).
        // The centre tap never leaves its clip: one base register + immediates, like the conditioning view; one integer
        // division per lane and tile: a tile crosses at most one clip edge - the launcher requires Ti >= 256.
        int rbe[2][MT], xv[4];
        {
            const int t0r = (m0 + lr) % p.Ti;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int sh = v < 3 ? v * dil : 0;
                const int row = lr + sh;
                xv[v] = (lh ^ ((row >> 1) & 7)) << 4;
                if (v == 0 || v == 2) {
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi) {
                        int tt = t0r + mi * 32;
                        tt = (tt >= p.Ti ? tt - p.Ti : tt) + (v - 1) * dil;
                        rbe[v >> 1][mi] = ((unsigned)tt < (unsigned)p.Ti ? row + mi * 32 : ZROW) * 128;
                    }
                }
            }
        }
        const int rb1 = (lr + dil) * 128;             // centre tap: slot row i + dil, tile mi at + mi * 4096 (immediate)
        const int rb3 = lr * 128;                     // conditioning view: row i, no mask
        // activation fragments of the running k-step, refilled tile by tile for the next one right behind the MFMA that
        // reads them (single buffer: the next use of hf[mi] is a whole k-step away)
        bf16x8 hf[MT];
        // (ko: the k-step's column offset, made opaque once per k-step - left to itself hipcc keeps every (view, ki, tile)
        // address of the whole unrolled loop in registers and spills)
        auto kofs = [&](int v, int ki) {
            int x = xv[v];
            asm volatile("" : "+v"(x));
            return (ki * 32) ^ x;
        };
        auto ldfrag1 = [&](const unsigned char* la, int v, int ko, int mi) {
            hf[mi] = *(const bf16x8*)(la + (v == 1 ? rb1 + mi * 4096 : v == 3 ? rb3 + mi * 4096 : rbe[(v >> 1) & 1][mi]) + ko);
        };
        // bias -> accumulators, through the scalar cache (uniform addresses; a vector load here would make hipcc wait for
        // the whole DMA queue at its first use): register r is row (r & 3) + 8 (r >> 2) + 4 lh of the fragment
        f32x16 acc[MT];
        {
            const float* __restrict__ bias = p.bias;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float b0 = bias[rs_packed_row(grp, acc_row_c(r))];
                const float b1 = bias[rs_packed_row(grp, acc_row_c(r) + 4)];
                const float b = lh ? b1 : b0;
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) acc[mi][r] = b;
            }
        }

        // ---- the K loop, every position a compile-time constant (the ring stages and fragment buffers must be registers,
        // never an indexed array).  Under k-step g: the fragments of k-step g + 1, one weight load into the stage k-step g - 1
        // has left and, when g opens an item, this wave's pieces of the item 2 (3) ahead - of this tile or of the next.
        rs_static_for<NK>([&](auto G) {
            constexpr int g = decltype(G)::value;
            constexpr int item = P::item_of(g), l = g - P::item_first(item);
            constexpr bool slice = P::is_slice(item);
            constexpr int view = slice ? l >> 2 : 3, ki = slice ? l & 3 : l;
            constexpr bool firstk = l == 0, last = l + 1 == P::item_ks(item);
            constexpr int pit = (item + AHEAD) % NI;                          // the item whose pieces ride this k-step ...
            constexpr bool pnext = item + AHEAD >= NI;                        // ... of the next tile?
            const unsigned char* la = lds + ((jb + item) & 3) * SLOT;
            {
                constexpr int n0 = C::wait_kstep(0, g), n1 = PERSIST ? C::wait_kstep(1, g) : n0;
                static_assert(C::steady_ok(g), "the vmcnt schedule must repeat from the second tile on");
                if (first) RS_CHK_WAIT(g % R, g, n0); else RS_CHK_WAIT(g % R, g, n1);
#if defined(FWN_RS_SAFE) && (FWN_RS_SAFE & 1)        // developer build: drain in front of every k-step
                rs_wwait<0>(wq[g % R]);
#else
                // ONE statement whatever the tile: under `if (first) wait<n0> else wait<n1>` hipcc merges the two arms' "+v"
                // operand in a new register and fills it with a v_mov IN FRONT of the s_waitcnt of one arm - a copy of a ring
                // stage whose load is still in flight (found in the ISA by tools/check_async_loads.py; it was the persistent
                // form's race: stale weights in the k-steps whose counts differ between the first and the later tiles,
                // whenever the load had not landed by then).  The smaller count is safe for both (it waits for more).
                rs_wwait<(n0 < n1 ? n0 : n1)>(wq[g % R]);
#endif
            }
            if constexpr (firstk) {
                // leaders: the barrier in front of a tile's item 0 was the one behind the previous tile's K loop (or barrier 0)
                if constexpr (LAG) barrier(std::integral_constant<int, item + 1>{}, std::integral_constant<int, NI + item + 1>{}, first, 3 + 2 * item);
                else if constexpr (item > 0) barrier(std::integral_constant<int, item>{}, std::integral_constant<int, NI + item>{}, first, 1 + 2 * item);
                if (FWN_RABL != 5 || g == 0) {
                    const int ko = kofs(view, ki);
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi) ldfrag1(la, view, ko, mi);
                }
            }
            constexpr int nview = slice ? (l + 1) >> 2 : 3, nki = slice ? (l + 1) & 3 : l + 1;
            const int kon = last ? 0 : kofs(nview, nki);
            __builtin_amdgcn_sched_barrier(0);
            // one MFMA per slot, each followed by at most one other instruction group (an in-order wave can only fill the
            // issue cycles behind the MFMA it has just issued)
            rs_static_for<MT>([&](auto MI) {
                constexpr int mi = decltype(MI)::value;
                acc[mi] = mfma32(wq[g % R], hf[mi], acc[mi]);
                if constexpr (!last && FWN_RABL != 5) ldfrag1(la, nview, kon, mi);
                if constexpr (!LAG && g == 0 && mi < PP) {
                    if (first) issue_piece(std::integral_constant<int, 1>{}, 1, m0, mi);
                }
                if constexpr (firstk && mi < PP && FWN_RABL != 4 && (PERSIST || !pnext))
                    issue_piece(std::integral_constant<int, pit>{}, (jb + item + AHEAD) & 3, pnext ? m0n : m0, mi);
                if constexpr (mi == 1 && FWN_RABL != 1) {
                    if constexpr (g + R - 1 < NK) issue_w(std::integral_constant<int, g + R - 1>{});
                    else if constexpr (PERSIST && (g - 1) % R <= R - 2) issue_w(std::integral_constant<int, (g - 1) % R>{});
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        if constexpr (PERSIST && (NK - 1) % R <= R - 2) issue_w(std::integral_constant<int, (NK - 1) % R>{});
        // leaders: the barrier that lets the laggers into their last item - and stands in front of the leaders' next item 0
        if constexpr (!LAG) barrier(std::integral_constant<int, NI>{}, std::integral_constant<int, 2 * NI>{}, first, 1 + 2 * NI);
        stamp(20);

        bool skip_epilogue = false;
        if (FWN_RABL == 2) {
            float s = 0.0f;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[mi][r];
            skip_epilogue = s != 12345.678f;
        }
        // ---- epilogue: tanh(f) sigmoid(g); registers 8 q + j (filter) and 8 q + 4 + j (gate) of channel j + 4 lh + 8 q, i.e.
        // a lane holds channels 4 lh .. + 3 (q = 0) and 8 + 4 lh .. + 3 (q = 1) of its time row as two 8-byte packs.  One
        // v_permlane32_swap per dword (guide T21) gives the lower half-wave channels 0-7 and the upper one channels 8-15 of
        // the row: ONE 16-byte store per lane and time tile instead of two 8-byte ones (the store tail is issue-bound).
        if (!skip_epilogue) {
            const srd_t so = make_srd(p.o, hbytes);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                const int row = m0 + mi * 32 + lr;
                const uint32_t voff = row < M ? (uint32_t)(row * FWN_HID + grp * 16 + 8 * lh) * 2u : FWN_OOB;
                uint32_t w[2][2];
                typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x16& a = acc[mi];
#pragma unroll
                    for (int d = 0; d < 2; ++d) {            // channels 2 d, 2 d + 1 of the group of four -> one v_cvt_pk_bf16_f32
                        const f32x2 y = {rs_gated1(a[8 * q + 2 * d], a[8 * q + 4 + 2 * d]), rs_gated1(a[8 * q + 2 * d + 1], a[8 * q + 4 + 2 * d + 1])};
                        w[q][d] = __builtin_bit_cast(uint32_t, __builtin_convertvector(y, bf16x2));
                    }
                }
                // vdst = the q = 0 pack, src = the q = 1 pack: lanes 32-63 of vdst swap with lanes 0-31 of src
                u32x4 out;
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(w[0][d], w[1][d], false, false);
                    // lower half: [own q0 | upper's q0] = channels 0-3 | 4-7; upper half: [lower's q1 | own q1] = channels 8-11 | 12-15
                    out[d] = sw[0];
                    out[2 + d] = sw[1];
                }
                __builtin_amdgcn_raw_buffer_store_b128(out, so, voff, 0, 0);
                RS_CHK_ISSUE();
            }
        }
#ifdef FWN_RS_CHECK
        ++chk_tile;
#endif
#if defined(FWN_RS_SAFE) && (FWN_RS_SAFE & 32)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        if (!has_next) break;
        m0 = m0n;
        jb = (jb + NI) & 3;
        first = false;
    }
    // behind the last tile: the look-ahead loads of a tile that does not exist (they re-read this wave's stream into the ring
    // registers) and its out-of-range pieces must have landed before the registers / the LDS are anyone else's
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    rs_touch_all<0, R>(wq);           // every ring stage named behind the wait (any R)
}

// NKC: conditioning k-steps of 16 (cin / 16 rounded up); MT: 32-row time tiles per wave; R: ring stages of weight fragments
// per wave (R - 1 k-steps in flight).  Persistent: gridDim.x (even) workgroups share the ntiles = 2 ceil(M / 32 MT) tiles.
#ifndef FWN_RS_R
#define FWN_RS_R 6
#endif
template <int NKC, int MT = 8, bool PERSIST = false, int R = FWN_RS_R, bool CLK = false>
__global__ __launch_bounds__(512) void gate_rs_kernel(GateRsArgs p, int ntiles) {
    static_assert(R >= 3 && R <= 12, "ring depth");
    static_assert(MT == 8 || MT == 4 || MT == 2, "256-, 128- or 64-row tiles");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * 8 * rs_pp(MT) * 1024];
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    RS_STAMP(0); RS_STAMP_RT(30);
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if constexpr (CLK) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
#ifdef FWN_RS_PRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(FWN_RS_PRIO);
#endif
    auto stamp = [&](int i) { RS_STAMP(i); };
    if (wave >= 4) gate_rs_wave<NKC, R, true, MT, PERSIST>(p, lds, wave, lane, wg, (int)gridDim.x, ntiles, stamp);
    else gate_rs_wave<NKC, R, false, MT, PERSIST>(p, lds, wave, lane, wg, (int)gridDim.x, ntiles, stamp);
    RS_STAMP(21); RS_STAMP_RT(31);
    if constexpr (CLK) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long* q = p.clk + ((size_t)blockIdx.x * 8 + wave) * 4;
            q[0] = clk_c0; q[1] = c1; q[2] = clk_r0; q[3] = r1;
        }
    }
}

// Packed gate weights -> fragment stream of gate_rs_kernel<NKC>: out[grp][g][lane][8], k-step g in plan order (RsPlan).
// Wd [512][768] and Wc [512][kcpad] are the gate-packed rows (packing.py).  Lane l holds fragment row l & 31,
// k 8 (l >> 5) .. + 7 of the k-step's 16 columns.
template <int NKC>
__global__ void gate_stream_pack_kernel(const bf16* __restrict__ Wd, const bf16* __restrict__ Wc, int kcpad, bf16* __restrict__ out) {
    using P = RsPlan<NKC>;
    const long total = 16L * P::NK * 64;               // 16-byte pieces
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const long t = i >> 6;
        const int g = (int)(t % P::NK), grp = (int)(t / P::NK);
        const int row = rs_packed_row(grp, lane & 31);
        const int col = P::kcol(g) + 8 * (lane >> 5);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (col < 768) v = *(const uint4*)(Wd + (size_t)row * 768 + col);
        else if (col - 768 < kcpad) v = *(const uint4*)(Wc + (size_t)row * kcpad + (col - 768));
        ((uint4*)out)[i] = v;
    }
}
