// Gated dilated layer with the three taps sharing one staged activation tile.
//
// The ring GEMM (gemm_ring.h) treats the k=3 dilated conv as K = 3 x 256 and stages the same h
// rows three times, shifted by the dilation.  At 256-row tiles the gate is bound by L2 -> LDS
// bytes (all 256 CUs streaming ~11 TB/s), so this kernel stages each 64-channel slice of
// h rows [m0 - dil, m0 + 256 + dil) ONCE and lets the taps read it at row offsets 0, dil, 2 dil:
// 606 KB instead of 868 KB per tile.  A clip edge inside the tile cannot be zero-filled by the
// DMA any more (the neighbouring clip's rows are live data for the centre tap): lanes whose
// tap row falls outside their clip read a zero row of the slot instead (row 263, which every
// staged slice leaves out of range).
//
// LDS: 2 activation slots x 33 KB (264 rows) + 2 weight slots x 32 KB.  One step = one weight
// chunk (tap, slice) = 16 MFMAs per wave; the next weight chunk and a third of the next
// activation slice are issued under the MFMAs of the current step.
#pragma once
#include "gemm_ring.h"

#define FWN_HALO_MAXDIL 3
#ifndef FWN_HABL
#define FWN_HABL 0   // developer ablation (tools/bench_gemm.hip): 1 no weight refills, 2 no activation refills, 3 neither, 4 no epilogue
#endif

// Tile BM x BN with BM / 16 waves (so every wave stages 2 activation pieces per slice, wave 0 the
// odd one) laid out (NWV / WN) x WN, WN = BN / 64.
//
// FP8 (BASELINE configs[4]: "fp8 MFMA dilated-conv path"): the three dilated taps read an e4m3 copy of h (p.h8, [M][256]
// bytes, written by the front / res epilogues) against e4m3 weights (p.Wd8, [512][768] bytes, one power-of-two scale
// per matrix) with v_mfma_scale_f32_32x32x64_f8f6f4 - twice the bf16 MFMA rate at half the L2 -> LDS bytes: a slice is
// 128 channels (the same 128-byte LDS rows and swizzle), 2 slices x 3 taps = 6 steps of two 64-deep MFMA k-steps
// instead of 12 steps of four 16-deep ones.  The weight scale rides in the instruction's E8M0 scale operand
// (scale_b = 127 - e for weights stored as W 2^e), so the accumulators hold the same quantity as in the bf16 path
// and the bf16 conditioning steps and the epilogue are unchanged.  Operand layout (probed on the device,
// tools/probe/fp8_layout.hip): lane l holds row l & 31, k = 32 (l >> 5) .. + 31 as 32 consecutive bytes.
typedef __attribute__((ext_vector_type(8))) int i32x8;

template <int BM, int BN, class Prob, bool FP8 = false>
__global__ __launch_bounds__(BM * 4) void gate_halo_kernel(Prob p, int ntn) {
    using G = RingGeom<64>;
    constexpr int NSL = FP8 ? 2 : 4;                         // activation slices of 128 bytes per row
    constexpr int NCONV = 3 * NSL;                           // conv steps
    constexpr int NWV = BM / 16, WN = BN / 64, WM = NWV / WN, MI = BM / (32 * WM);
    constexpr int PB = (BN / 8) / NWV;                       // weight pieces per wave per chunk
    static_assert(WM * WN == NWV && MI * 32 * WM == BM && PB * NWV * 8 == BN && PB <= 4, "bad tile");
    constexpr int AP = BM / 8 + 1;                           // 8-row pieces per activation slice
    static_assert(AP * 8 > BM + 2 * FWN_HALO_MAXDIL, "the zero row must lie beyond the halo");
    constexpr int A_BYTES = AP * 1024, B_BYTES = BN * G::RB;
    constexpr int ZROW = AP * 8 - 1;                         // never staged: reads as zero
#ifdef FWN_STAMP      // diagnostic build (tools/bench_gemm.hip): s_memtime stamps per wave and step, dumped to p.stamps
    constexpr int STAMP_BYTES = 16 * 24 * 4 * 8;            // 16 waves x 24 steps x 4 stamps
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * A_BYTES + 2 * B_BYTES + STAMP_BYTES];
    unsigned long long* const stamp = (unsigned long long*)(lds + 2 * A_BYTES + 2 * B_BYTES);
#define FWN_STAMP_AT(step, k) do { if ((threadIdx.x & 63) == 0) stamp[((threadIdx.x >> 6) * 24 + (step)) * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FWN_STAMP_AT(step, k) do { } while (0)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * A_BYTES + 2 * B_BYTES];
#endif
    unsigned char* const ldsA = lds;
    unsigned char* const ldsB = lds + 2 * A_BYTES;

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = wg / ntn, tile_n = wg % ntn;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int dil = p.dil, M = p.M, cin = p.cin, kcpad = p.kcpad;
    const int ncond = p.ca ? kcpad / 64 : 0;
    const int S = NCONV + ncond;

    // ---- DMA plan: activation pieces wave + 16 j (j = 2: piece 32, wave 0 only), weight pieces
    // wave + 16 j (j < 2)
    // (conditioning addresses are recomputed at each issue: only 2-5 of the steps use them and the
    // conv loop has no registers to spare at 128 VGPRs)
    uint32_t ah[3], bd[PB];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int jrow = 8 * (wave + NWV * j) + (lane >> 3);
        const int c = (lane & 7) ^ ((jrow >> 1) & 7);
        const int g = m0 - dil + jrow;
        const bool ok = jrow < BM + 2 * dil && (unsigned)g < (unsigned)M;
        ah[j] = ok ? (uint32_t)(g * FWN_HID * (FP8 ? 1 : 2) + c * 16) : FWN_OOB;     // byte offset: rows of 256 (e4m3) or 512 bytes
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int jrow = 8 * (wave + NWV * j) + (lane >> 3);
        bd[j] = (uint32_t)((n0 + jrow) * 3 * FWN_HID * (FP8 ? 1 : 2) + ((lane & 7) ^ ((jrow >> 1) & 7)) * 16);
    }
    const uint32_t hbytes = (uint32_t)((size_t)M * FWN_HID * (FP8 ? 1 : 2));
    const uint32_t cbytes = (uint32_t)((size_t)M * cin * 2);
    auto issueA_conv = [&](int kc, int j) {          // slice kc of h -> slot kc & 1
        if (j == 2 && wave != 0) return;
        if ((FWN_HABL == 2 || FWN_HABL == 3) && kc > 0) return;
        buf_load16_lds(make_srd(FP8 ? (const void*)p.h8 : (const void*)p.h, hbytes), ah[j] + (uint32_t)kc * 128u,
                       ldsA + (kc & 1) * A_BYTES + (wave + NWV * j) * 1024);
    };
    auto issueA_cond = [&](int qc, int j) {          // conditioning chunk qc -> slot qc & 1
        if (FWN_HABL == 2 || FWN_HABL == 3) return;
        const int jrow = 8 * (wave + NWV * j) + (lane >> 3);
        const int col = qc * 64 + ((lane & 7) ^ ((jrow >> 1) & 7)) * 8;
        const bool ok = col < cin && m0 + jrow < M;
        buf_load16_lds(make_srd(p.ca, cbytes), ok ? (uint32_t)((m0 + jrow) * cin + col) * 2u : FWN_OOB,
                       ldsA + (qc & 1) * A_BYTES + (wave + NWV * j) * 1024);
    };
    auto issueB = [&](int s, int j) {                // weight chunk of step s -> slot s & 1
        if ((FWN_HABL == 1 || FWN_HABL == 3) && s > 0) return;
        unsigned char* dst = ldsB + (s & 1) * B_BYTES + (wave + NWV * j) * 1024;
        if (s < NCONV) {
            const int kc = s / 3, tap = s - 3 * kc;
            if constexpr (FP8)
                buf_load16_lds(make_srd(p.Wd8, 512u * 768u), bd[j] + (uint32_t)(tap * FWN_HID + kc * 128), dst);
            else
                buf_load16_lds(make_srd(p.Wd, 512u * 768u * 2u), bd[j] + (uint32_t)(tap * FWN_HID + kc * 64) * 2u, dst);
        } else {
            const int jrow = 8 * (wave + NWV * j) + (lane >> 3);
            const int col = (s - NCONV) * 64 + ((lane & 7) ^ ((jrow >> 1) & 7)) * 8;
            buf_load16_lds(make_srd(p.Wc, (uint32_t)(512u * kcpad * 2u)),
                           (uint32_t)((n0 + jrow) * kcpad + col) * 2u, dst);
        }
    };

    // ---- fragment addresses.  Activation rows: tile row i sits at slot row i + tap*dil; lanes whose
    // tap row leaves the clip read the zero row.  The conditioning chunks sit at slot row i.
    int rb[3][MI], xv[3];
    {
        const int i0 = wm * 32 * MI + lr;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int row = i0 + tap * dil;
            xv[tap] = (lh ^ ((row >> 1) & 7)) << 4;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int t = (m0 + i0 + mi * 32) % p.Ti + (tap - 1) * dil;
                rb[tap][mi] = ((unsigned)t < (unsigned)p.Ti ? row + mi * 32 : ZROW) * G::RB;
            }
        }
    }
    int bfr[G::KS];
#pragma unroll
    for (int ki = 0; ki < G::KS; ++ki) bfr[ki] = G::off(wn * 64 + lr, ki * 2 + lh);

    f32x16 acc[MI][2];

    // one step: 4 k-steps of 4 MFMAs, fragments double-buffered, `issue(ki)` slipped under them
    // nks: k-steps of 16 this step multiplies (4, or fewer for the last conditioning chunk: cin = 80 ends 16 columns
    // into its second chunk - 3 of block 0's 56 k-steps are pure zero padding); the issue() hooks run for all four slots
    auto mma_step = [&](const unsigned char* la, const unsigned char* lb, int rb0, int rb1, int x, int nks, auto&& issue) {
        // opaque copy: keeps the 24 (tap, mi, ki) fragment addresses from being hoisted out of the
        // slice loop into registers the kernel does not have (they cost 3 VALU ops per k-step here)
        asm volatile("" : "+v"(x));
        bf16x8 af[2][MI], bf_[2][2];
        auto ldfrag = [&](int ki, int sb) {
            const int ko = (ki * 32) ^ x;
            af[sb][0] = *(const bf16x8*)(la + rb0 + ko);
            if constexpr (MI > 1) af[sb][1] = *(const bf16x8*)(la + rb1 + ko);
            bf_[sb][0] = *(const bf16x8*)(lb + bfr[ki]);
            bf_[sb][1] = *(const bf16x8*)(lb + bfr[ki] + 32 * G::RB);
        };
        ldfrag(0, 0);
#pragma unroll
        for (int ki = 0; ki < G::KS; ++ki) {
            if (ki >= nks) {            // wave-uniform: a trimmed chunk only runs its DMA hooks here
                issue(ki);
                continue;
            }
            if (ki + 1 < G::KS && ki + 1 < nks) ldfrag(ki + 1, (ki + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = mfma32(af[ki & 1][mi], bf_[ki & 1][ni], acc[mi][ni]);
                    if (mi == 0 && ni == 0) issue(ki);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // fp8 step: two k-steps of 64 (= 2 x 16-byte pieces per lane and operand), 4 scaled MFMAs each; x8 = swizzle << 4
    const int xb8 = (((wn * 64 + lr) >> 1) & 7) << 4, brow8 = (wn * 64 + lr) * G::RB;
    const int sc_a = 127, sc_b = p.sb;              // E8M0 scale operands: activations as stored, weights 2^-e
    auto mma_step8 = [&](const unsigned char* la, const unsigned char* lb, int rb0, int rb1, int x8, auto&& issue) {
        asm volatile("" : "+v"(x8));
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int c0 = ks * 64 + lh * 32;
            const int oa0 = c0 ^ x8, oa1 = (c0 + 16) ^ x8, ob0 = c0 ^ xb8, ob1 = (c0 + 16) ^ xb8;
            auto frag = [&](const unsigned char* base, int o0, int o1) {
                const u32x4 lo = *(const u32x4*)(base + o0), hi = *(const u32x4*)(base + o1);
                return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            };
            i32x8 a[MI], b[2];
            a[0] = frag(la + rb0, oa0, oa1);
            if constexpr (MI > 1) a[1] = frag(la + rb1, oa0, oa1);
            b[0] = frag(lb + brow8, ob0, ob1);
            b[1] = frag(lb + brow8 + 32 * G::RB, ob0, ob1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    // inline asm: hipcc (ROCm 7.2) gives the builtin a destination distinct from its C operand (an
                    // early-clobber form), which doubles the accumulator registers and spills 264 VGPRs at the 128 this
                    // 16-wave workgroup has; "+v" ties them.  s_nop 1 covers a scale operand re-materialised by a VALU
                    // move just ahead of the statement (hipcc pads nothing inside or before an asm, guide section 5.7).
                    asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]"
                                 : "+v"(acc[mi][ni]) : "v"(a[mi]), "v"(b[ni]), "v"(sc_a), "v"(sc_b));
                    if (ni == 0) issue(2 * ks + mi);
                }
            if constexpr (MI == 1) issue(2 * ks + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // prologue: slice 0 and the first weight chunk
#pragma unroll
    for (int j = 0; j < 3; ++j) issueA_conv(0, j);
#pragma unroll
    for (int j = 0; j < PB; ++j) issueB(0, j);
    // bias -> accumulators, after the prologue DMAs are queued (see gemm_ring.h)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const float c0 = p.acc_init(n0 + wn * 64 + ni * 32 + lr);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = c0;
    }

    // Per step the wave issues [weight piece, weight piece, activation piece] in that order, so
    // "all but the newest one" (vmcnt(1)) leaves only the activation piece of a later slice in flight.
    int s = 0;
    for (int kc = 0; kc < NSL; ++kc) {
        const unsigned char* la = ldsA + (kc & 1) * A_BYTES;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap, ++s) {
            // in flight and not needed yet: the activation piece issued under the previous step
            // (tap 1, 2 of this slice; none when that step issued nothing or this step opens a slice)
            const bool a_next_conv = kc + 1 < NSL, a_next_cond = kc + 1 == NSL && ncond > 0;
            const bool prev_a = tap > 0 && (a_next_conv || (a_next_cond && tap - 1 < 2)) && (tap - 1 < 2 || wave == 0);
            FWN_STAMP_AT(s, 0);
            if (prev_a) FWN_WAIT_VMCNT(1);
            else FWN_WAIT_VMCNT(0);
            FWN_STAMP_AT(s, 1);
            FWN_RING_BARRIER();
            FWN_STAMP_AT(s, 2);
            const unsigned char* lb = ldsB + (s & 1) * B_BYTES;
            auto hooks = [&](int ki) {
                if (ki < PB && s + 1 < S) issueB(s + 1, ki);
                if (ki == (PB < 3 ? 2 : 3)) {        // after this step's weight pieces
                    if (a_next_conv) issueA_conv(kc + 1, tap);
                    else if (a_next_cond && tap < 2) issueA_cond(0, tap);
                }
            };
            if constexpr (FP8) mma_step8(la, lb, rb[tap][0], rb[tap][MI - 1], xv[tap] ^ (lh << 4), hooks);   // xv = (lh ^ swizzle) << 4
            else mma_step(la, lb, rb[tap][0], rb[tap][MI - 1], xv[tap], 4, hooks);
            FWN_STAMP_AT(s, 3);
        }
    }
    if constexpr (FP8) {
        // the last asm MFMA's result -> the compiler's next reader / writer of the accumulators (16-pass XDL: 18 states)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[mi][ni]));
    }
    const int ic = wm * 32 * MI + lr;                // conditioning chunks sit at slot row i, no mask
    const int xc = (lh ^ ((ic >> 1) & 7)) << 4;
    const int rbc[2] = {ic * G::RB, (ic + 32) * G::RB};
    for (int qc = 0; qc < ncond; ++qc, ++s) {
        FWN_WAIT_VMCNT(0);
        FWN_RING_BARRIER();
        const unsigned char* la = ldsA + (qc & 1) * A_BYTES;
        const unsigned char* lb = ldsB + (s & 1) * B_BYTES;
        const int kleft = cin - qc * 64;             // valid K columns of this chunk (the rest is zero padding)
        mma_step(la, lb, rbc[0], rbc[MI - 1], xc, kleft >= 64 ? 4 : (kleft + 15) >> 4, [&](int ki) {
            if (ki < PB && s + 1 < S) issueB(s + 1, ki);
            if (ki == 3 && qc + 1 < ncond) {
                issueA_cond(qc + 1, 0);
                issueA_cond(qc + 1, 1);
            }
        });
    }
#ifdef FWN_STAMP
    FWN_STAMP_AT(20, 0);                                  // end of the K loop
    if (p.stamps && blockIdx.x < 2) {
        __syncthreads();
        for (int i = threadIdx.x; i < 16 * 24 * 4; i += blockDim.x) p.stamps[blockIdx.x * 16 * 24 * 4 + i] = stamp[i];
    }
#endif
    if (FWN_HABL == 4) {
        float sacc = 0.0f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[mi][ni][r];
        if (sacc != 12345.678f) return;
    }
    p.template epilogue<MI>(acc, m0 + wm * 32 * MI, n0 + wn * 64, lane);
}
