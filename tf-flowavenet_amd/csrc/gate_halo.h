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
template <int BM, int BN, class Prob>
__global__ __launch_bounds__(BM * 4) void gate_halo_kernel(Prob p, int ntn) {
    using G = RingGeom<64>;
    constexpr int NWV = BM / 16, WN = BN / 64, WM = NWV / WN, MI = BM / (32 * WM);
    constexpr int PB = (BN / 8) / NWV;                       // weight pieces per wave per chunk
    static_assert(WM * WN == NWV && MI * 32 * WM == BM && PB * NWV * 8 == BN && PB <= 4, "bad tile");
    constexpr int AP = BM / 8 + 1;                           // 8-row pieces per activation slice
    static_assert(AP * 8 > BM + 2 * FWN_HALO_MAXDIL, "the zero row must lie beyond the halo");
    constexpr int A_BYTES = AP * 1024, B_BYTES = BN * G::RB;
    constexpr int ZROW = AP * 8 - 1;                         // never staged: reads as zero
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * A_BYTES + 2 * B_BYTES];
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
    const int S = 12 + ncond;

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
        ah[j] = ok ? (uint32_t)(g * FWN_HID + c * 8) * 2u : FWN_OOB;
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int jrow = 8 * (wave + NWV * j) + (lane >> 3);
        bd[j] = (uint32_t)((n0 + jrow) * 3 * FWN_HID + ((lane & 7) ^ ((jrow >> 1) & 7)) * 8) * 2u;
    }
    const uint32_t hbytes = (uint32_t)((size_t)M * FWN_HID * 2);
    const uint32_t cbytes = (uint32_t)((size_t)M * cin * 2);
    auto issueA_conv = [&](int kc, int j) {          // slice kc of h -> slot kc & 1
        if (j == 2 && wave != 0) return;
        if ((FWN_HABL == 2 || FWN_HABL == 3) && kc > 0) return;
        buf_load16_lds(make_srd(p.h, hbytes), ah[j] + (uint32_t)kc * 128u,
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
        if (s < 12) {
            const int kc = s / 3, tap = s - 3 * kc;
            buf_load16_lds(make_srd(p.Wd, 512u * 768u * 2u), bd[j] + (uint32_t)(tap * FWN_HID + kc * 64) * 2u, dst);
        } else {
            const int jrow = 8 * (wave + NWV * j) + (lane >> 3);
            const int col = (s - 12) * 64 + ((lane & 7) ^ ((jrow >> 1) & 7)) * 8;
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
    auto mma_step = [&](const unsigned char* la, const unsigned char* lb, int rb0, int rb1, int x, auto&& issue) {
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
            if (ki + 1 < G::KS) ldfrag(ki + 1, (ki + 1) & 1);
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
    for (int kc = 0; kc < 4; ++kc) {
        const unsigned char* la = ldsA + (kc & 1) * A_BYTES;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap, ++s) {
            // in flight and not needed yet: the activation piece issued under the previous step
            // (tap 1, 2 of this slice; none when that step issued nothing or this step opens a slice)
            const bool a_next_conv = kc + 1 < 4, a_next_cond = kc + 1 == 4 && ncond > 0;
            const bool prev_a = tap > 0 && (a_next_conv || (a_next_cond && tap - 1 < 2)) && (tap - 1 < 2 || wave == 0);
            if (prev_a) FWN_WAIT_VMCNT(1);
            else FWN_WAIT_VMCNT(0);
            __builtin_amdgcn_s_barrier();
            const unsigned char* lb = ldsB + (s & 1) * B_BYTES;
            mma_step(la, lb, rb[tap][0], rb[tap][MI - 1], xv[tap], [&](int ki) {
                if (ki < PB && s + 1 < S) issueB(s + 1, ki);
                if (ki == (PB < 3 ? 2 : 3)) {        // after this step's weight pieces
                    if (a_next_conv) issueA_conv(kc + 1, tap);
                    else if (a_next_cond && tap < 2) issueA_cond(0, tap);
                }
            });
        }
    }
    const int ic = wm * 32 * MI + lr;                // conditioning chunks sit at slot row i, no mask
    const int xc = (lh ^ ((ic >> 1) & 7)) << 4;
    const int rbc[2] = {ic * G::RB, (ic + 32) * G::RB};
    for (int qc = 0; qc < ncond; ++qc, ++s) {
        FWN_WAIT_VMCNT(0);
        __builtin_amdgcn_s_barrier();
        const unsigned char* la = ldsA + (qc & 1) * A_BYTES;
        const unsigned char* lb = ldsB + (s & 1) * B_BYTES;
        mma_step(la, lb, rbc[0], rbc[MI - 1], xc, [&](int ki) {
            if (ki < PB && s + 1 < S) issueB(s + 1, ki);
            if (ki == 3 && qc + 1 < ncond) {
                issueA_cond(qc + 1, 0);
                issueA_cond(qc + 1, 1);
            }
        });
    }
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
