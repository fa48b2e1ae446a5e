// Tail of a flow, register-streamed form (round 6): the same arithmetic as tail_chain.h - skip-sum GEMM -> ReLU -> final
// 1x1 -> ReLU -> ZeroConv1d -> affine coupling + ActNorm (+ log-det partials) and, chained, the NEXT flow's front conv
// (modules.py:175-180,51-56,144,164-165; model.py:86-102,124-141,146-161,166-174) - with the workgroup turned around the
// way gate_rs.h turned the gate around.
//
// tail_chain.h gives a wave 32 time rows x ALL 256 channels, so S and U never leave the registers - but every wave then
// multiplies by every weight: all 0.4 MB of a flow's tail weights go through the LDS of every workgroup (LDS-DMA, ~41 GB/s
// per CU: 10 - 13 us per launch whatever the rows) and are read back from LDS by each of its waves (at one wave per SIMD:
// the 128-row form runs its MFMA phases at 67 % of the at-clock rate and every epilogue in sequence behind them;
// profiles/r05_kernel_summary_B8.txt: 0.26 / 0.18 / 0.095 / 0.05 of the MFMA peak at blocks 0 - 3).  Here:
//   * EIGHT waves, two per SIMD; a wave owns 32 OUTPUT CHANNELS x all BM = 32 MT time rows of the tile (MT accumulator tiles).
//     Its weights are its own: they never touch LDS.  Wskip | Wfinal are packed once in MFMA-fragment order
//     (tail_stream_pack_kernel: [wave][k-step][lane][8 bf16]) and streamed straight into a ring of R register stages by
//     buffer loads issued from inline asm, waits counted by hand (TrsCount: a compile-time walk over the wave's own issue
//     order) - gate_rs.h's weight path.  Each fragment feeds MT MFMAs.
//   * The shared operand is the activations: the o rows arrive by LDS-DMA as 64-column slices in a 4-slot ring (three in
//     flight), S and U cross the workgroup through ONE LDS image [BM][256] (8-byte writes from the accumulator layout,
//     16-byte fragment reads, XOR-swizzled: lds_off256) between the phases - two barriers per exchange, six more for the
//     eight o slices.
//   * ZeroConv + coupling run on MT of the waves (one 32-row tile each, Wz from an LDS image that arrives under phase 2)
//     while the others apply the a-plane's ActNorm; the chained front conv is one more 32-channel MFMA chain on every wave.
//   * Accumulation order, MFMA shape and operand placement are those of tail_chain.h: results are the same bits
//     (tests/test_gpu_parity.py compares the two kernels with ==).
// LDS: region A = o ring, then the Wz image; region B = S, then U; 8 KB of per-flow constants; MT = 4: 140 KB (one workgroup
// per CU), MT = 2: 74 KB (two), MT = 1: 57 KB.
#pragma once
#include "gate_rs.h"
#include "tail_chain.h"

#ifndef FWN_TRS_SAFE
#define FWN_TRS_SAFE 0           // developer build: 1 = drain the vector-memory queue in front of every k-step and barrier
#endif
#ifndef FWN_TRS_ABL
#define FWN_TRS_ABL 0            // developer ablation (wrong results): 1 no weight loads after the prologue, 2 no epilogue
#endif

// diagnostic build (tools/bench_tail_rs.hip, -DFWN_TRS_STAMP): s_memtime per wave at the phase boundaries, into a buffer nothing
// else reads: stamps[(workgroup * 8 + wave) * 16 + i]; i = 14 / 15: s_memrealtime at wave start / end
#ifdef FWN_TRS_STAMP
#define TRS_STAMP(i) do { if (lane == 0) stamps[((size_t)blockIdx.x * 8 + wave) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define TRS_STAMP_RT(i) do { if (lane == 0) stamps[((size_t)blockIdx.x * 8 + wave) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define TRS_STAMP_ARG , unsigned long long* stamps
#else
#define TRS_STAMP(i) do { } while (0)
#define TRS_STAMP_RT(i) do { } while (0)
#define TRS_STAMP_ARG
#endif

__host__ __device__ constexpr int trs_default_r(int mt) { return mt >= 4 ? 8 : 12; }

// Vector-memory operations a wave has issued AFTER a given one at the point where it waits for that one (every wave of the
// workgroup issues the same sequence):
//   prologue : one constant piece, then W(0 .. R-2) with the o pieces of item k / 2 (PP each) in front of W(k), k = 0, 2, 4
//   phase 1  : k-step g = 0 .. 31 (item g / 4): [wait W(g)] [g % 4 == 0: wait own pieces of the item, barrier]; behind MFMA
//              slot j < PP of an item's first k-step piece j of item + 3; behind slot WSLOT W(g + R - 1)
//   S        : SAVE: 2 MT stores; barrier; 4 Wz pieces; 4 plane loads (the coupling's in_b values, registers 0 - 3)
//   phase 2  : k-step g = 32 .. 47: [wait W(g)]; behind slot WSLOT W(g + R - 1) (FRONT: three more k-steps = the next flow's
//              front weights)
template <int MT, int R, bool FRONT, bool SAVE>
struct TrsCount {
    static constexpr int PP = MT >= 2 ? MT / 2 : 1, WSLOT = MT > 1 ? 1 : 0;
    static constexpr int NKF = FRONT ? 3 : 0, NKW = 48 + NKF, NI = 8, LA = 3, NSV = SAVE ? 2 * MT : 0, NZ = 4, NX = 4;
    static_assert(R >= 3 && R - 1 <= 32, "ring depth");
    // target: tk 0 = W(ta) / 1 = the last own piece of item ta; query: qk 0 = the wait in front of k-step qa / 1 = the wait in
    // front of item barrier qa
    static constexpr int walk(int tk, int ta, int qk, int qa) {
        int count = -1, result = -1;
        bool done = false;
#define T_OP() do { if (count >= 0) ++count; } while (0)
#define T_W(k) do { T_OP(); if (tk == 0 && ta == (k)) count = 0; } while (0)
#define T_P1(i, j) do { T_OP(); if (tk == 1 && ta == (i) && (j) == PP - 1) count = 0; } while (0)
#define T_QK(g) do { if (!done && qk == 0 && qa == (g)) { result = count; done = true; } } while (0)
#define T_QB(i) do { if (!done && qk == 1 && qa == (i)) { result = count; done = true; } } while (0)
        T_OP();
        for (int k = 0; k < R - 1; ++k) {
            if (k % 2 == 0 && k / 2 < LA)
                for (int j = 0; j < PP; ++j) T_P1(k / 2, j);
            T_W(k);
        }
        for (int g = 0; g < 32; ++g) {
            T_QK(g);
            if ((g & 3) == 0) T_QB(g >> 2);
            for (int slot = 0; slot < MT; ++slot) {
                if ((g & 3) == 0 && slot < PP && (g >> 2) + LA < NI) T_P1((g >> 2) + LA, slot);
                if (slot == WSLOT && g + R - 1 < NKW) T_W(g + R - 1);
            }
        }
        for (int s = 0; s < NSV + NZ + NX; ++s) T_OP();
        for (int g = 32; g < 48; ++g) {
            T_QK(g);
            for (int slot = 0; slot < MT; ++slot)
                if (slot == WSLOT && g + R - 1 < NKW) T_W(g + R - 1);
        }
#undef T_OP
#undef T_W
#undef T_P1
#undef T_QK
#undef T_QB
        return result;
    }
    static constexpr int wait_kstep(int g) { return walk(0, g, 0, g); }
    static constexpr int wait_barrier(int i) { return walk(1, i, 1, i); }
};

// bf16 pack of four accumulator registers (ReLU applied): two dwords
typedef __attribute__((ext_vector_type(2))) __bf16 trs_bf16x2;
// ReLU AFTER the rounding, on the packed pair, as ONE v_pk_max_i16 against zero: a negative bf16 is a negative int16, -0 the
// most negative one; a value that rounds to zero is zero either way - the same bits as (bf16)fmaxf(x, 0) for every x that is
// not a NaN.  fmaxf on a register fresh from an MFMA costs a canonicalising v_max_f32 besides the real one: 128 + 32 VALU
// instructions per parked tile set against 32 + 32 here, at two waves per SIMD (the exchange was VALU-bound).
typedef __attribute__((ext_vector_type(2))) short trs_i16x2;
__device__ __forceinline__ uint32_t trs_relu_pk(f32x2 v) {
    const trs_i16x2 w = __builtin_bit_cast(trs_i16x2, __builtin_convertvector(v, trs_bf16x2));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(w, trs_i16x2{0, 0}));
}
__device__ __forceinline__ uint2 trs_pack4_relu(const f32x16& a, int q) {
    return make_uint2(trs_relu_pk(f32x2{a[4 * q], a[4 * q + 1]}), trs_relu_pk(f32x2{a[4 * q + 2], a[4 * q + 3]}));
}
// A lane's four packs of one time tile (channels 8 q + 4 lh .. + 3 of its row, q = 0 .. 3) as two 16-byte pieces: one
// v_permlane32_swap per dword (guide T21) gives the lower half-wave channels 16 p .. + 7 and the upper one 16 p + 8 .. + 15.
__device__ __forceinline__ void trs_store_row32(const uint2 (&pk)[4], srd_t dst, uint32_t voff) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * p].x, pk[2 * p + 1].x, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * p].y, pk[2 * p + 1].y, false, false);
        const u32x4 out = {s0[0], s1[0], s0[1], s1[1]};
        __builtin_amdgcn_raw_buffer_store_b128(out, dst, voff, p * 32, 0);
    }
}

// LDS-DMA piece (64 lanes x 16 bytes -> LDS at the wave-uniform byte address lds_addr + lane * 16) issued from inline asm.
// Through the builtin (buf_load16_lds) hipcc knows that LDS writes are in flight and puts an s_waitcnt vmcnt(0) in front of the
// next ds_read / ds_write it emits (it cannot tell that the access is to another slot): in this kernel that drained the whole
// prologue - two more o slices and R - 1 weight fragments - in front of the first k-step, and the Wz image in front of phase
// 2 (stamps: 3.6 k cycles at barrier 0).  Issued from asm the pieces are invisible to that pass; every wait for them is one of
// TrsCount's.  M0 (the LDS address) is compiler-reserved: saved and restored inside the statement (guide section 5.7).
__device__ __forceinline__ void trs_dma16(u32x4 srd, uint32_t voff, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(srd), "s"(lds_addr) : "memory");
}
// raw buffer descriptor (stride 0, range check on the byte offset) in SGPRs an asm statement can name
__device__ __forceinline__ u32x4 trs_srd(const void* p, uint32_t bytes) {
    const unsigned long long b = (unsigned long long)(uintptr_t)p;
    return u32x4{(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) & 0xffffu,
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}

// one dword per lane, hidden from hipcc's wait bookkeeping like rs_wload (the caller counts vmcnt)
__device__ __forceinline__ void trs_xload(float& dst, u32x4 srd, uint32_t voff) {
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(srd) : "memory");
}

// every ring stage made opaque at this point (behind a wait: no consumer of a stage is scheduled above it)
template <int I, int R>
__device__ __forceinline__ void trs_touch(bf16x8 (&wq)[R]) {
    if constexpr (I < R) {
        asm volatile("" : "+v"(wq[I]));
        trs_touch<I + 1, R>(wq);
    }
}

// MT: 32-row time tiles per workgroup (tile = 32 MT rows); FRONT: the chained front conv (Ch <= 8); SAVE: keep S, U, Z for
// the training backward; R: ring stages of weight fragments per wave.  L = 2 layers, one ZeroConv pair tile (Ch <= 32).
template <int MT, bool FRONT, bool SAVE, int R = trs_default_r(MT)>
__global__ __launch_bounds__(512, (MT <= 2 ? 4 : 2)) void tail_rs_kernel(TailArgs a, const bf16* __restrict__ Wts TRS_STAMP_ARG) {
    using C = TrsCount<MT, R, FRONT, SAVE>;
    static_assert(MT == 1 || MT == 2 || MT == 4, "32-, 64- or 128-row tiles");
    static_assert(32 * MT * 512 >= 3 * MT * 1024, "the front conv's B fragments live in the U region");
    constexpr int BM = 32 * MT, PP = C::PP, WSLOT = C::WSLOT, NKW = C::NKW, NI = C::NI, LA = C::LA;
    constexpr int SLOT = BM * 128 > 8192 ? BM * 128 : 8192;      // one o slice [rows][64] (at least the 8 pieces of one per wave)
    constexpr int A_BYTES = 4 * SLOT > 32768 ? 4 * SLOT : 32768; // o ring, then the Wz image [4][64][64]
    constexpr int B_BYTES = BM * 512;                            // S, then U: [BM][256] bf16 (lds_off256)
    constexpr int C_BS = 0, C_BF = 256, C_BZ = 512, C_EZ = 768, C_AN = 1024, C_BFN = 1280, C_ANN = 1536, CST = 2048;   // 1-KB pieces
    constexpr int T_FLOATS = FRONT ? BM * 8 : 0;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[A_BYTES + B_BYTES + CST * 4 + T_FLOATS * 4 + 64];
    unsigned char* const lA = lds;
    const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_ptr_t)lds);   // LDS byte address of the array (0: the kernel's only LDS object)
    unsigned char* const lB = lds + A_BYTES;
    float* const cst = (float*)(lds + A_BYTES + B_BYTES);
    float* const Tt = cst + CST;
    float* const red = Tt + T_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int g0 = a.overlap ? (int)blockIdx.x * (BM - 2) - 1 : (int)blockIdx.x * BM;
    const int M = a.M, Ch = a.Ch;
    constexpr bool front = FRONT;
    TRS_STAMP(0); TRS_STAMP_RT(14);

    // ---- constants -> LDS, one piece per wave, ahead of everything else in the queue (the first barrier's wait covers it)
    {
        const float* src = a.bs;
        uint32_t bytes = 0;
        if (wave == 0) { src = a.bs; bytes = 1024u; }
        else if (wave == 1) { src = a.bfin; bytes = 1024u; }
        else if (wave == 2) { src = a.bz; bytes = 256u; }
        else if (wave == 3) { src = a.ez; bytes = 256u; }
        else if (wave == 4) { src = a.an; bytes = (uint32_t)Ch * 32u; }
        else if (wave == 5 && front) { src = a.bfn; bytes = 1024u; }
        else if (wave == 6 && front && a.an_next) { src = a.an_next; bytes = (uint32_t)Ch * 32u; }
        trs_dma16(trs_srd(src, bytes), (uint32_t)lane * 16u, lds0 + (uint32_t)(A_BYTES + B_BYTES + 1024 * wave));
    }

    // ---- o slices: item i = (layer i / 4, columns 64 (i % 4) ..), image [rows][64] in slot i % 4; piece j of this wave =
    // image rows 8 (wave + 8 j) .. + 7 (rows past the tile or the matrix: out of range = zeros)
    const u32x4 srd_o = trs_srd(a.o, (uint32_t)(((size_t)(a.L - 1) * a.o_stride + (size_t)M * FWN_HID) * 2));
    auto issue_piece = [&](int item, int j) {
        const int pi = wave + 8 * j;
        const int jrow = 8 * pi + (lane >> 3);
        const int c = (lane & 7) ^ ((jrow >> 1) & 7);
        const int gr = g0 + jrow;
        const bool ok = (jrow < BM) & ((unsigned)gr < (unsigned)M);
        const uint32_t off = (uint32_t)((item >> 2) * a.o_stride + (long)gr * FWN_HID + (item & 3) * 64 + c * 8) * 2u;
        trs_dma16(srd_o, ok ? off : FWN_OOB, lds0 + (uint32_t)((item & 3) * SLOT + (wave + 8 * j) * 1024));
    };

    // ---- this wave's weight stream: k-step g -> ring stage g % R; g >= 48: the next flow's front weights [256][kfn]
    const unsigned long long wbase = (unsigned long long)(uintptr_t)Wts + (unsigned long long)wave * (48 * 1024);
    const u32x4 wsrd = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wbase),
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(wbase >> 32)) & 0xffffu, (uint32_t)(48 * 1024), 0x00020000u};
    const unsigned long long nbase = (unsigned long long)(uintptr_t)(front ? (const void*)a.Wfn : (const void*)Wts);
    const u32x4 nsrd = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)nbase),
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(nbase >> 32)) & 0xffffu,
                        front ? (uint32_t)(256 * a.kfn * 2) : 0u, 0x00020000u};
    const uint32_t wl = (uint32_t)lane * 16u;
    bf16x8 wq[R];
    auto issue_w = [&](auto G) {
        constexpr int g = decltype(G)::value;
        if constexpr (g < 48) {
            rs_wload<(g * 1024) % 4096>(wq[g % R], wsrd, (uint32_t)(((g * 1024) / 4096) * 4096), wl);
        } else {
            constexpr int kk = g - 48;
            const int col = kk * 16 + lh * 8;
            rs_wload<0>(wq[g % R], nsrd, 0u, col < a.kfn ? (uint32_t)(((wave * 32 + lr) * a.kfn + col) * 2) : FWN_OOB);
        }
    };
    auto wait_w = [&](auto G) {
        constexpr int g = decltype(G)::value;
        if (FWN_TRS_SAFE) rs_wwait<0>(wq[g % R]);
        else rs_wwait<C::wait_kstep(g)>(wq[g % R]);
    };

    // prologue order: what the first k-steps need first - slice 0 and two weight fragments, then the other slices between the
    // rest of the ring's first R - 1 fragments (TrsCount::walk mirrors it)
    rs_static_for<R - 1>([&](auto G) {
        constexpr int g = decltype(G)::value;
        if constexpr (g % 2 == 0 && g / 2 < LA) {
#pragma unroll
            for (int j = 0; j < PP; ++j) issue_piece(g / 2, j);
        }
        issue_w(G);
    });
    static_assert((R - 2) / 2 >= LA - 1, "the prologue issues the first LA slices among its weight loads");
    TRS_STAMP(1);

    // bias tables are in accumulator order (packing.acc_k_perm swaps bits 2 and 3 of the channel index): register r of this
    // lane is channel 32 wave + (r & 3) + 4 lh + 8 (r >> 2), stored at 32 wave + (r & 3) + 4 ((r >> 2) & 1) + 8 lh + 16 (r >> 3)
    f32x16 acc[MT];
    auto init_acc = [&](const float* bias, bool perm) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int at = perm ? wave * 32 + 4 * (q & 1) + 8 * lh + 16 * (q >> 1) : wave * 32 + 8 * q + 4 * lh;
            const float4 v = *(const float4*)(bias + at);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) { acc[mi][4 * q] = v.x; acc[mi][4 * q + 1] = v.y; acc[mi][4 * q + 2] = v.z; acc[mi][4 * q + 3] = v.w; }
        }
    };
    bf16x8 hf[MT];
    // fragment addresses.  o slice: row mi * 32 + lr, 16-byte chunk 2 ki + lh of a 128-byte row (lds_off64);
    // S / U image: row mi * 32 + lr, chunk 2 g2 + lh of a 512-byte row (lds_off256)
    const int rb1 = lr * 128, xv1 = (lh ^ ((lr >> 1) & 7)) << 4;
    const int rb2 = lr * 512, xv2 = (lh ^ (lr & 15)) << 4;
    auto kofs = [&](int x, int k32) {            // made opaque once per k-step (gate_rs.h: keeps hipcc from materialising every address)
        asm volatile("" : "+v"(x));
        return k32 ^ x;
    };
    // where this lane's pack q of tile mi goes in the S / U image: row mi * 32 + lr, channels 32 wave + 8 q + 4 lh .. + 3
    auto su_off = [&](int mi, int q) { return lds_off256(mi * 32 + lr, wave * 4 + q) + 8 * lh; };
    const uint32_t su_bytes = (uint32_t)((size_t)M * FWN_HID * 2);
    auto row_of = [&](int mi) { return g0 + mi * 32 + lr; };
    auto owned_row = [&](int mi) {
        const int rl = mi * 32 + lr, row = g0 + rl;
        return (unsigned)row < (unsigned)M && (!a.overlap || (rl >= 1 && rl <= BM - 2));
    };
    // accumulators -> ReLU -> bf16 -> the S / U image (and the training copy)
    auto park = [&](bf16* save) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            uint2 pk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                pk[q] = trs_pack4_relu(acc[mi], q);
                *(uint2*)(lB + su_off(mi, q)) = pk[q];
            }
            if constexpr (SAVE)
                trs_store_row32(pk, make_srd(save, su_bytes), owned_row(mi) ? (uint32_t)(row_of(mi) * FWN_HID + wave * 32 + 8 * lh) * 2u : FWN_OOB);
        }
    };

    // ---------------- phase 1: S^T[32 channels of this wave][BM rows] = Ws @ [o_0 | o_1]^T + bs ----------------
    // (the constants arrive with item 0: the accumulators are initialised behind barrier 0)
    rs_static_for<32>([&](auto G) {
        constexpr int g = decltype(G)::value;
        constexpr int item = g >> 2, ki = g & 3;
        const unsigned char* la = lA + (item & 3) * SLOT;
        wait_w(G);
        if constexpr (ki == 0) {
            if (FWN_TRS_SAFE) rs_vmwait<0>(); else rs_vmwait<C::wait_barrier(item)>();
            FWN_RING_BARRIER();
            if constexpr (item == 0) { TRS_STAMP(2); init_acc(cst + C_BS, true); }
            const int ko = kofs(xv1, 0);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) hf[mi] = *(const bf16x8*)(la + rb1 + mi * 4096 + ko);
        }
        const int kon = ki < 3 ? kofs(xv1, (ki + 1) * 32) : 0;
        __builtin_amdgcn_sched_barrier(0);
        rs_static_for<MT>([&](auto MI) {
            constexpr int mi = decltype(MI)::value;
            acc[mi] = mfma32(wq[g % R], hf[mi], acc[mi]);
            if constexpr (ki < 3) hf[mi] = *(const bf16x8*)(la + rb1 + mi * 4096 + kon);
            if constexpr (ki == 0 && mi < PP && item + LA < NI) issue_piece(item + LA, mi);
            if constexpr (mi == WSLOT && g + R - 1 < NKW && FWN_TRS_ABL != 1) issue_w(std::integral_constant<int, g + R - 1>{});
            __builtin_amdgcn_sched_barrier(0);
        });
    });
    TRS_STAMP(3);
    park(a.save_s);
    FWN_RING_BARRIER();              // S is complete; every wave has left the o ring
    TRS_STAMP(4);
    // ---- the ZeroConv weights [64][256] as 4 sub-tiles [64][64] of 8 KB into region A (tail_chain.h's image): 4 pieces per wave
    {
        const u32x4 srd_z = trs_srd(a.Wz, 64u * 256u * 2u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pi = wave + 8 * j, q = pi >> 3;
            const int r = (pi & 7) * 8 + (lane >> 3);
            trs_dma16(srd_z, (uint32_t)(r * FWN_HID + q * 64 + ((lane & 7) ^ ((r >> 1) & 7)) * 8) * 2u, lds0 + (uint32_t)(pi * 1024));
        }
    }
    // ---- the coupling's in_b values of the tile waves' rows (accumulator registers 0 - 3: channels 4 lh .. + 3, all there are for
    // Ch <= 8), requested now so that they land under phase 2: loaded in the epilogue they were a dependent round trip of ~3 us
    // at the end of every workgroup.  Every wave issues the four loads (the others out of range): one wait schedule.
    float xpre[4];
    {
        const unsigned long long xbase = (unsigned long long)(uintptr_t)a.xb;
        const u32x4 xsrd = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)xbase),
                            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(xbase >> 32)) & 0xffffu,
                            (uint32_t)((size_t)M * Ch * 4), 0x00020000u};
        const int row = g0 + wave * 32 + lr;
        const bool rv = (wave < MT) & ((unsigned)row < (unsigned)M);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tau = r + 4 * lh;
            trs_xload(xpre[r], xsrd, (rv && tau < Ch) ? (uint32_t)(row * Ch + tau) * 4u : FWN_OOB);
        }
    }

    // ---------------- phase 2: U^T = Wf @ S^T + bfin ----------------
    init_acc(cst + C_BF, true);
    rs_static_for<16>([&](auto G2) {
        constexpr int g2 = decltype(G2)::value, g = 32 + g2;
        wait_w(std::integral_constant<int, g>{});
        if constexpr (g2 == 0) {
            const int ko = kofs(xv2, 0);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) hf[mi] = *(const bf16x8*)(lB + rb2 + mi * 16384 + ko);
        }
        const int kon = g2 < 15 ? kofs(xv2, ((g2 + 1) & 7) * 32) + ((g2 + 1) >> 3) * 256 : 0;
        __builtin_amdgcn_sched_barrier(0);
        rs_static_for<MT>([&](auto MI) {
            constexpr int mi = decltype(MI)::value;
            acc[mi] = mfma32(wq[g % R], hf[mi], acc[mi]);
            if constexpr (g2 < 15) hf[mi] = *(const bf16x8*)(lB + rb2 + mi * 16384 + kon);
            if constexpr (mi == WSLOT && g + R - 1 < NKW && FWN_TRS_ABL != 1) issue_w(std::integral_constant<int, g + R - 1>{});
            __builtin_amdgcn_sched_barrier(0);
        });
    });
    TRS_STAMP(5);
    // everything this wave has requested has landed: the Wz pieces, the front weights (they stay in their ring stages)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    trs_touch<0, R>(wq);
    asm volatile("" : "+v"(xpre[0]), "+v"(xpre[1]), "+v"(xpre[2]), "+v"(xpre[3]));
    TRS_STAMP(6);
    FWN_RING_BARRIER();              // every wave has read its last S fragments
    TRS_STAMP(7);
    park(a.save_u);
    FWN_RING_BARRIER();              // U is complete
    TRS_STAMP(8);

    if (FWN_TRS_ABL == 2) return;
    const float* an_a = cst + C_AN;
    const float* an_b = an_a + 4 * Ch;
    const uint32_t plane_bytes = (uint32_t)((size_t)M * Ch * 4);
    float lsum = 0.0f;
    if (wave < MT) {
        // ---------------- phase 3 (this wave: time tile `wave`): [log_s | t]^T = Wz @ U^T ----------------
        f32x16 z0, z1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { z0[r] = 0.0f; z1[r] = 0.0f; }
        const unsigned char* ub = lB + rb2 + wave * 16384;
        const unsigned char* zb = lA + lr * 128;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int g2 = 4 * q + kk;
                const bf16x8 b = *(const bf16x8*)(ub + (((g2 & 7) * 32) ^ xv2) + (g2 >> 3) * 256);
                const int wo = q * 8192 + (((kk * 2 + lh) ^ ((lr >> 1) & 7)) << 4);
                z0 = mfma32(*(const bf16x8*)(zb + wo), b, z0);
                z1 = mfma32(*(const bf16x8*)(zb + wo + 4096), b, z1);
            }
        TRS_STAMP(9);
        // ---------------- affine coupling + ActNorm on the b plane (tail_chain.h's epilogue, one pair tile) ----------------
        const int rl = wave * 32 + lr, row = g0 + rl;
        const bool rvalid = (unsigned)row < (unsigned)M;
        const bool owned = rvalid && (!a.overlap || (rl >= 1 && rl <= BM - 2));
        const float* bzl = cst + C_BZ;
        const float* ezl = cst + C_EZ;
        const float* ann = cst + C_ANN;
        const srd_t sxo = make_srd(a.xb_out ? a.xb_out : a.xb, plane_bytes);
        const bool vec4 = Ch >= 4;
        // One group = accumulator registers 4 q .. 4 q + 3 = channels 8 q + 4 lh .. + 3 of the lane's row.  Ch <= 8 (blocks 0 - 3)
        // has group 0 only: the others are skipped by wave-uniform branches (tail_chain.h evaluates all 16 registers under
        // masks - ~1 500 instructions and 160 dependent LDS waits per wave, 3 us at the end of every workgroup; the masked
        // terms add exact zeros to lsum, so skipping them leaves the bits alone).  A group's constants are read as a batch.
        const srd_t sz = make_srd(a.save_z, SAVE && a.save_z ? (uint32_t)((size_t)M * 2 * Ch * 4) : 0u);
        float ov0[4];
        auto group = [&](auto Q, const float (&xq)[4]) {
            constexpr int q = decltype(Q)::value;
            const int j0 = 8 * q + 4 * lh;
            float bl[4], el[4], bt[4], et[4], s0[4], s1[4], s2[4], l3[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = j0 + e, tc = j < Ch ? j : 0;
                bl[e] = bzl[j]; el[e] = ezl[j]; bt[e] = bzl[j + 32]; et[e] = ezl[j + 32];
                s0[e] = an_b[tc];
                s1[e] = a.inverse ? an_b[2 * Ch + tc] : an_b[Ch + tc];
                l3[e] = a.inverse ? 0.0f : an_a[3 * Ch + tc] + an_b[3 * Ch + tc];
                s2[e] = 0.0f;
            }
            float ov[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * q + e, j = j0 + e;
                const bool ok = j < Ch && owned;
                const float ls = (z0[r] + bl[e]) * el[e];
                const float t = (z1[r] + bt[e]) * et[e];
                if (!a.inverse) {
                    const float yb = (xq[e] + s0[e]) * s1[e];                          // ActNorm (model.py:86-94)
                    ov[e] = (yb - t) * __expf(-ls);                                     // model.py:134
                    lsum += ok ? (l3[e] - ls) : 0.0f;                                   // model.py:135 + :80
                } else {
                    const float yb = xq[e] * __expf(ls) + t;                            // model.py:156
                    ov[e] = yb * s1[e] - s0[e];                                         // ActNorm^-1 (model.py:97-102)
                }
            }
            if (vec4) {
                const uint32_t vo = (j0 < Ch && owned) ? (uint32_t)(row * Ch + j0) * 4u : FWN_OOB;
                const u32x4 o4 = {__builtin_bit_cast(unsigned int, ov[0]), __builtin_bit_cast(unsigned int, ov[1]),
                                  __builtin_bit_cast(unsigned int, ov[2]), __builtin_bit_cast(unsigned int, ov[3])};
                __builtin_amdgcn_raw_buffer_store_b128(o4, sxo, vo, 0, 0);
                if constexpr (SAVE) {         // Z = U Wz + bz: [row][tau] = log_s, [row][Ch + tau] = t (before exp(3 scale))
                    const uint32_t zoff = vo != FWN_OOB ? (uint32_t)(row * 2 * Ch + j0) * 4u : FWN_OOB;
                    u32x4 zl, zt;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        zl[e] = __builtin_bit_cast(unsigned int, z0[4 * q + e] + bl[e]);
                        zt[e] = __builtin_bit_cast(unsigned int, z1[4 * q + e] + bt[e]);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(zl, sz, zoff, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(zt, sz, zoff, Ch * 4, 0);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = j0 + e;
                    const uint32_t vo = (j < Ch && owned) ? (uint32_t)(row * Ch + j) * 4u : FWN_OOB;
                    buf_store_f32(sxo, vo, 0, ov[e]);
                    if constexpr (SAVE) {
                        const uint32_t zoff = vo != FWN_OOB ? (uint32_t)(row * 2 * Ch + j) * 4u : FWN_OOB;
                        buf_store_f32(sz, zoff, 0, z0[4 * q + e] + bl[e]);
                        buf_store_f32(sz, zoff, Ch * 4, z1[4 * q + e] + bt[e]);
                    }
                }
            }
            if constexpr (q == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) ov0[e] = ov[e];
            }
        };
        // group 0: its in_b values were requested behind barrier S (xpre); the other groups exist for Ch >= 16 only
        group(std::integral_constant<int, 0>{}, xpre);
        if (Ch > 8) {
            auto later = [&](auto Q) {
                constexpr int q = decltype(Q)::value;
                const int tau0 = 8 * q + 4 * lh;
                const float4 v = *(const float4*)(a.xb + ((tau0 < Ch && rvalid) ? (size_t)row * Ch + tau0 : 0));   // clamped
                const float xq[4] = {v.x, v.y, v.z, v.w};
                group(Q, xq);
            };
            later(std::integral_constant<int, 1>{});
            if (Ch > 16) {
                later(std::integral_constant<int, 2>{});
                later(std::integral_constant<int, 3>{});
            }
        }
        if constexpr (FRONT) {
            // the tile's out_b (every row, halo included) as the next flow's network input: that flow's ActNorm applied in
            // the forward direction (model.py:86-94 ahead of its coupling), raw in the inverse direction
#pragma unroll
            for (int r = 0; r < 4; ++r) {                // channels < 8 live in registers 0..3 (tau = r + 4 lh)
                const int tau = r + 4 * lh;
                if (tau < Ch) {
                    const float v = a.an_next ? (ov0[r] + ann[tau]) * ann[Ch + tau] : ov0[r];
                    Tt[rl * 8 + tau] = rvalid ? v : 0.0f;
                }
            }
        }
        if (a.partial) {
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) lsum += __shfl_xor(lsum, s);
            if (lane == 0) red[wave] = lsum;
        }
    } else {
        // ---------------- a plane: ActNorm only (the coupling passes in_a through unchanged), owned rows [ra, rb) ----------------
        constexpr int NT = 512 - 64 * MT;
        const int t2 = tid - 64 * MT;
        const srd_t sxa = make_srd(a.xa, plane_bytes);
        const int ra = max(a.overlap ? g0 + 1 : g0, 0), rb = min(a.overlap ? g0 + BM - 1 : g0 + BM, M);
        const int total = (rb - ra) * Ch;
        const uint32_t base = (uint32_t)(ra * Ch) * 4u;
        const int chmask = Ch - 1;
        if (Ch >= 4) {       // (plain clamped float4 loads: tail_chain.h's note on raw_buffer_load_b128)
            for (int idx = t2 * 4; idx < total; idx += NT * 4) {
                const float4 v = *(const float4*)(a.xa + (size_t)ra * Ch + idx);
                float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int tau = (idx + e) & chmask;
                    f[e] = a.inverse ? (f[e] * an_a[2 * Ch + tau] - an_a[tau]) : ((f[e] + an_a[tau]) * an_a[Ch + tau]);
                }
                const u32x4 o4 = {__builtin_bit_cast(unsigned int, f[0]), __builtin_bit_cast(unsigned int, f[1]),
                                  __builtin_bit_cast(unsigned int, f[2]), __builtin_bit_cast(unsigned int, f[3])};
                __builtin_amdgcn_raw_buffer_store_b128(o4, sxa, base + (uint32_t)idx * 4u, 0, 0);
            }
        } else {
            for (int idx = t2; idx < total; idx += NT) {
                const int tau = idx & chmask;
                const uint32_t off = base + (uint32_t)idx * 4u;
                const float v = buf_load_f32(sxa, off, 0);
                buf_store_f32(sxa, off, 0, a.inverse ? (v * an_a[2 * Ch + tau] - an_a[tau]) : ((v + an_a[tau]) * an_a[Ch + tau]));
            }
        }
    }
    TRS_STAMP(10);
    if (a.partial == nullptr && !FRONT) { TRS_STAMP_RT(15); return; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS only: __syncthreads() would also wait for the plane stores
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    TRS_STAMP(11);
    if (a.partial && tid == 0) {
        float t = 0.0f;
        for (int w = 0; w < MT; ++w) t += red[w];
        a.partial[blockIdx.x] = t;
    }
    if constexpr (FRONT) {
        // ---------------- the next flow's front conv: h0^T = Wfn @ [taps of out_b as hi | lo]^T + bfn, ReLU ----------------
        // The B operand (taps of out_b as hi | lo bf16 pairs, clip edges masked) does not depend on the wave's channels: each of
        // its MT * nks fragments is built ONCE (fragment f = kk * MT + mi by wave f % 8) into the U region - every wave building
        // all of them was 3 - 5 k cycles of VALU per workgroup (tools/bench_tail_rs.hip stamps) - and read back by everyone.
        init_acc(cst + C_BFN, false);
        const int chlog = 31 - __builtin_clz(Ch);
        const int nks = a.kfn >> 4;
        for (int f = wave; f < MT * nks; f += 8) {
            const int mi = f & (MT - 1), kk = f / MT;
            const int rl = mi * 32 + lr, row = g0 + rl;
            const bool rvalid = (unsigned)row < (unsigned)M;
            const int t_in = rvalid ? row % a.Ti : 0;
            Pack16 b;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = kk * 8 + lh * 4 + e;     // (tap, tau) of this element pair
                const int tap = idx >> chlog, tau = idx & (Ch - 1);
                const int rr = rl + tap - 1;
                const bool ok = idx < 3 * Ch && (unsigned)(t_in + tap - 1) < (unsigned)a.Ti && (unsigned)rr < (unsigned)BM;
                const float v = ok ? Tt[(ok ? rr : 0) * 8 + tau] : 0.0f;
                const bf16 hi = (bf16)v;
                b.e[2 * e] = hi;
                b.e[2 * e + 1] = (bf16)(v - (float)hi);
            }
            *(uint4*)(lB + f * 1024 + lane * 16) = b.u;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        rs_static_for<3>([&](auto KK) {
            constexpr int kk = decltype(KK)::value;
            if (kk < nks) {
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
                    acc[mi] = mfma32(wq[(48 + kk) % R], *(const bf16x8*)(lB + (kk * MT + mi) * 1024 + lane * 16), acc[mi]);
            }
        });
        TRS_STAMP(12);
        const srd_t sh = make_srd(a.h0_next, su_bytes);
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            uint2 pk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) pk[q] = trs_pack4_relu(acc[mi], q);
            trs_store_row32(pk, sh, owned_row(mi) ? (uint32_t)(row_of(mi) * FWN_HID + wave * 32 + 8 * lh) * 2u : FWN_OOB);
        }
    }
    TRS_STAMP(13); TRS_STAMP_RT(15);
}

// Wskip [256][512] | Wfinal [256][256] (rows in accumulator order: packing.acc_k_perm) -> fragment stream
// out[wave][g][lane][8]: lane l holds channel 32 wave + (l & 31), k 16 g' + 8 (l >> 5) .. + 7 of k-step g (g < 32: Wskip
// column 16 g; else Wfinal column 16 (g - 32)).
// jobs != NULL: the streams of gridDim.y flows in one launch (a training step re-packs every flow's stream from the freshly
// packed Wskip | Wfinal: packing.PackPlan) - job blockIdx.y = {Wskip, Wfinal, out} device pointers
struct TailStreamJob { const bf16* Ws; const bf16* Wf; bf16* out; };
__global__ void tail_stream_pack_kernel(const bf16* __restrict__ Ws, const bf16* __restrict__ Wf, bf16* __restrict__ out,
                                        const TailStreamJob* __restrict__ jobs) {
    if (jobs) {
        const TailStreamJob j = jobs[blockIdx.y];
        Ws = j.Ws; Wf = j.Wf; out = j.out;
    }
    const long total = 8L * 48 * 64;                   // 16-byte pieces
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const long t = i >> 6;
        const int g = (int)(t % 48), w = (int)(t / 48);
        const int ch = w * 32 + (lane & 31);
        const int prow = (ch & ~12) | ((ch & 4) << 1) | ((ch & 8) >> 1);        // acc_k_perm: bits 2 and 3 swapped
        const int col = (g < 32 ? g : g - 32) * 16 + 8 * (lane >> 5);
        ((uint4*)out)[i] = g < 32 ? *(const uint4*)(Ws + (size_t)prow * 512 + col) : *(const uint4*)(Wf + (size_t)prow * 256 + col);
    }
}
