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
// leave CUs empty).  DMA pieces per wave and item: 8 waves x PP x 8 rows must hold the tile, its halo and a zero row.
__host__ __device__ constexpr int rs_pp(int mt) { return mt == 8 ? 5 : 3; }      // 320- / 192-row slots
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
};
#ifdef FWN_RS_STAMP
#define RS_STAMP(i) do { if (lane == 0) p.stamps[((size_t)blockIdx.x * 8 + wave) * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define RS_STAMP_RT(i) do { if (lane == 0) p.stamps[((size_t)blockIdx.x * 8 + wave) * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RS_STAMP(i) do { } while (0)
#define RS_STAMP_RT(i) do { } while (0)
#endif

// one 16-byte load per lane, hidden from hipcc's wait bookkeeping; the caller counts vmcnt
// (sbase is scalar arithmetic on kernel arguments: an SALU result needs no wait states in front of a VMEM read)
template <int OFF>
__device__ __forceinline__ void rs_wload(bf16x8& dst, const unsigned char* sbase, uint32_t voff) {
    static_assert(OFF >= 0 && OFF < 4096, "13-bit signed immediate");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "i"(OFF) : "memory");
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
// program order walked at compile time.  Every wave opens with its pieces of item 0.  Leaders (LAG = false): ring loads
// W[0 .. R-2], [barrier 0]; under the first k-step of item i the pieces of item i + 2 (slots 0 .. 4, one per MFMA; under
// k-step 0 those of item 1 in front of them), under every k-step g the load of W[g + R - 1] (slot 1, behind that slot's
// pieces).  Laggers: [barrier 0], pieces of items 1 and 2, ring loads, [barrier 1]; under the first k-step of item i the
// pieces of item i + 3.
template <int NKC, int R, bool LAG, int MT>
struct RsCount {
    using P = RsPlan<NKC>;
    static constexpr int PP = rs_pp(MT);
    static_assert(PP <= MT, "one piece per MFMA slot");
    // target: weight k-step tw (or -1) / last piece of item ti (or -1); query: the wait in front of k-step qg (or -1) /
    // the arrival at barrier qb (or -1)
    static constexpr int walk(int tw, int ti, int qg, int qb) {
        int count = -1, result = -1;
        bool done = false;
        for (int j = 0; j < PP; ++j) { if (count >= 0) ++count; if (0 == ti && j == PP - 1) count = 0; }
        if (!LAG) for (int k = 0; k < R - 1; ++k) { if (count >= 0) ++count; if (k == tw) count = 0; }
        if (!done && qb == 0) { result = count; done = true; }
        if (LAG) {
            for (int it = 1; it <= 2 && it < P::NI; ++it)
                for (int j = 0; j < PP; ++j) { if (count >= 0) ++count; if (it == ti && j == PP - 1) count = 0; }
            for (int k = 0; k < R - 1; ++k) { if (count >= 0) ++count; if (k == tw) count = 0; }
            if (!done && qb == 1) { result = count; done = true; }
        }
        for (int i = 0; i < P::NI; ++i) {
            const int ahead = LAG ? 3 : 2;
            for (int l = 0; l < P::item_ks(i); ++l) {
                const int g = P::item_first(i) + l;
                if (!done && g == qg) { result = count; done = true; }
                const bool pieces = l == 0 && i + ahead < P::NI;
                for (int slot = 0; slot < MT; ++slot) {
                    if (!LAG && g == 0 && slot < PP) { if (count >= 0) ++count; if (1 == ti && slot == PP - 1) count = 0; }
                    if (pieces && slot < PP) { if (count >= 0) ++count; if (i + ahead == ti && slot == PP - 1) count = 0; }
                    if (slot == 1 && g + R - 1 < P::NK) { if (count >= 0) ++count; if (g + R - 1 == tw) count = 0; }
                }
            }
            if (!done && qb == i + (LAG ? 2 : 1)) { result = count; done = true; }
        }
        return result;
    }
    static constexpr int wait_kstep(int g) { return walk(g, -1, g, -1); }
    static constexpr int wait_barrier(int b) { return walk(-1, b, -1, b); }   // own pieces of item b (b < NI)
};

template <int NKC, int R, bool LAG, int MT, class Stamp>
__device__ __forceinline__ void gate_rs_wave(const GateRsArgs& p, unsigned char* lds, int wave, int lane, int m0, int grp, Stamp&& stamp) {
    using P = RsPlan<NKC>;
    using C = RsCount<NKC, R, LAG, MT>;
    constexpr int NK = P::NK, NI = P::NI, PP = rs_pp(MT), BM = 32 * MT, ZROW = rs_zrow(MT);
    constexpr int SLOT = 8 * PP * 1024;
    const int lr = lane & 31, lh = lane >> 5;
    const int dil = p.dil, M = p.M, cin = p.cin;

    // bias -> accumulators, through the scalar cache (uniform addresses; a vector load here would make hipcc wait for
    // the whole prologue queue at its first use): register r is row (r & 3) + 8 (r >> 2) + 4 lh of the fragment
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

    // ---- DMA pieces: piece j of this wave = slot rows 8 (wave + 8 j) .. + 7; offsets are computed at issue time (a few
    // VALU operations per piece under the MFMAs) rather than kept in registers across the K loop
    const uint32_t hbytes = (uint32_t)((size_t)M * FWN_HID * 2);
    const uint32_t cbytes = (uint32_t)((size_t)M * cin * 2);
    const int prow = 8 * wave + (lane >> 3);          // slot row of piece 0
    auto issue_piece = [&](auto ITEM, int j) {
        constexpr int item = decltype(ITEM)::value;
        static_assert(item < NI, "no such item");
        unsigned char* dst = lds + (item & 3) * SLOT + (wave + 8 * j) * 1024;
        const int jrow = prow + 64 * j;
        const int c = (lane & 7) ^ ((jrow >> 1) & 7);
        if constexpr (P::is_slice(item)) {
            const int g = m0 - dil + jrow;
            const bool ok = (jrow < BM + 2 * dil) & ((unsigned)g < (unsigned)M);
            buf_load16_lds(make_srd(p.h, hbytes), ok ? (uint32_t)(g * (FWN_HID * 2) + c * 16 + P::slice_of(item) * 128) : FWN_OOB, dst);
        } else {
            const int col = P::chunk_of(item) * 64 + c * 8;
            const bool ok = (jrow < BM) & (m0 + jrow < M) & (col < cin);
            buf_load16_lds(make_srd(p.ca, cbytes), ok ? (uint32_t)((m0 + jrow) * cin + col) * 2u : FWN_OOB, dst);
        }
    };

    // ---- weight stream of this wave: k-step g -> ring stage g % R
    const unsigned char* wbase = (const unsigned char*)p.Wg + (size_t)grp * NK * 1024;
    const uint32_t wl = (uint32_t)lane * 16u;
    bf16x8 wq[R];
    auto issue_w = [&](auto G) {                      // 4 KiB windows: base = scalar add, the rest an immediate
        constexpr int g = decltype(G)::value;
        rs_wload<(g * 1024) % 4096>(wq[g % R], wbase + ((g * 1024) / 4096) * 4096, wl);
    };

    // ---- prologue (order = RsCount::walk): the pieces of item 0 first, everywhere; the laggers' other issues wait until
    // barrier 0 has let the leaders start
#pragma unroll
    for (int j = 0; j < PP; ++j) issue_piece(std::integral_constant<int, 0>{}, j);
    if constexpr (!LAG) rs_static_for<R - 1>([&](auto G) { issue_w(G); });

    // ---- activation fragment addresses: view v = tap 0..2 (slot row i + tap*dil, clip mask) or 3 (conditioning: row i).
    // The centre tap never leaves its clip: one base register + immediates, like the conditioning view; one integer
    // division per lane: a tile of 256 rows crosses at most one clip edge - the launcher requires Ti >= 256.
    int rbe[2][MT], xv[4];
    {
        const int t0 = (m0 + lr) % p.Ti;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int sh = v < 3 ? v * dil : 0;
            const int row = lr + sh;
            xv[v] = (lh ^ ((row >> 1) & 7)) << 4;
            if (v == 0 || v == 2) {
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    int t = t0 + mi * 32;
                    t = (t >= p.Ti ? t - p.Ti : t) + (v - 1) * dil;
                    rbe[v >> 1][mi] = ((unsigned)t < (unsigned)p.Ti ? row + mi * 32 : ZROW) * 128;
                }
            }
        }
    }
    const int rb1 = (lr + dil) * 128;                 // centre tap: slot row i + dil, tile mi at + mi * 4096 (immediate)
    const int rb3 = lr * 128;                         // conditioning view: row i, no mask
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

    // barrier number b: every wave first waits for its own pieces of item b, the item the LEADERS open behind it
    auto barrier = [&](auto B) {
        constexpr int b = decltype(B)::value;
        if constexpr (b < NI) rs_vmwait<C::wait_barrier(b)>();
        stamp(1 + 2 * b);
        if (FWN_RABL != 3) FWN_RING_BARRIER();       // (nothing is in flight here by construction: the last k-step of an item prefetches nothing)
        stamp(2 + 2 * b);
    };
    if constexpr (LAG) {
        barrier(std::integral_constant<int, 0>{});
        rs_static_for<2>([&](auto I) {
            constexpr int it = decltype(I)::value + 1;
            if constexpr (it < NI) {
#pragma unroll
                for (int j = 0; j < PP; ++j) issue_piece(std::integral_constant<int, it>{}, j);
            }
        });
        rs_static_for<R - 1>([&](auto G) { issue_w(G); });
    }

    // ---- the K loop, every position a compile-time constant (the ring stages and fragment buffers must be registers,
    // never an indexed array).  Under k-step g: the fragments of k-step g + 1, the weights of k-step g + R - 1 (into the
    // stage k-step g - 1 has just left) and, when g opens an item, this wave's pieces of the item 2 (3) ahead.
    rs_static_for<NK>([&](auto G) {
        constexpr int g = decltype(G)::value;
        constexpr int item = P::item_of(g), l = g - P::item_first(item);
        constexpr bool slice = P::is_slice(item);
        constexpr int view = slice ? l >> 2 : 3, ki = slice ? l & 3 : l;
        constexpr bool first = l == 0, last = l + 1 == P::item_ks(item);
        constexpr int ahead = LAG ? 3 : 2;
        constexpr bool issuing = first && item + ahead < NI;
        const unsigned char* la = lds + (item & 3) * SLOT;
        rs_wwait<C::wait_kstep(g)>(wq[g % R]);
        if constexpr (first) {
            barrier(std::integral_constant<int, item + (LAG ? 1 : 0)>{});
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
            if constexpr (!LAG && g == 0 && mi < PP) issue_piece(std::integral_constant<int, 1>{}, mi);
            if constexpr (issuing && mi < PP && FWN_RABL != 4) issue_piece(std::integral_constant<int, issuing ? item + ahead : 0>{}, mi);
            if constexpr (mi == 1 && g + R - 1 < NK && FWN_RABL != 1) issue_w(std::integral_constant<int, g + R - 1>{});
            __builtin_amdgcn_sched_barrier(0);
        });
    });
    // the barrier that lets the laggers into their last item (leaders), nothing for the laggers: every wave has executed
    // NI + 1 barriers
    if constexpr (!LAG) barrier(std::integral_constant<int, NI>{});
    stamp(20);

    if (FWN_RABL == 2) {
        float s = 0.0f;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[mi][r];
        if (s != 12345.678f) return;
    }
    // ---- epilogue: tanh(f) sigmoid(g); registers 8 q + j (filter) and 8 q + 4 + j (gate) of channel j + 4 lh + 8 q, i.e. a
    // lane holds channels 4 lh .. + 3 (q = 0) and 8 + 4 lh .. + 3 (q = 1) of its time row as two 8-byte packs.  One
    // v_permlane32_swap per dword (guide T21) gives the lower half-wave channels 0-7 and the upper one channels 8-15 of the
    // row: ONE 16-byte store per lane and time tile instead of two 8-byte ones (the store tail is issue-bound).
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
    }
}

// NKC: conditioning k-steps of 16 (cin / 16 rounded up); R: ring stages of weight fragments per wave (R - 1 k-steps in
// flight)
#ifndef FWN_RS_R
#define FWN_RS_R 6
#endif
template <int NKC, int MT = 8, int R = FWN_RS_R>
__global__ __launch_bounds__(512) void gate_rs_kernel(GateRsArgs p) {
    static_assert(R >= 3 && R <= 12, "ring depth");
    static_assert(MT == 8 || MT == 4, "256- or 128-row tiles");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * 8 * rs_pp(MT) * 1024];
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = wg >> 1, tile_n = wg & 1;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = tile_m * (32 * MT);
    const int grp = tile_n * 8 + wave;                // channel group: 16 channels
    RS_STAMP(0); RS_STAMP_RT(30);
#ifdef FWN_RS_PRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(FWN_RS_PRIO);
#endif
    auto stamp = [&](int i) { RS_STAMP(i); };
    if (wave >= 4) gate_rs_wave<NKC, R, true, MT>(p, lds, wave, lane, m0, grp, stamp);
    else gate_rs_wave<NKC, R, false, MT>(p, lds, wave, lane, m0, grp, stamp);
    RS_STAMP(21); RS_STAMP_RT(31);
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
