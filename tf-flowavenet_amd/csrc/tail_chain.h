// Tail of a flow, register-chained:  skip-sum GEMM -> ReLU -> final 1x1 -> ReLU -> ZeroConv1d -> affine coupling +
// ActNorm (+ log-det partials), and - when the flows of a block are chained - the NEXT flow's front conv on the plane
// the coupling has just produced.  modules.py:175-180,51-56,144,164-165; model.py:86-102,124-141,146-161.
//
// Weight-streaming: each wave owns 32 time rows and ALL 256 hidden channels.  The GEMMs are computed transposed
// (channels on the accumulator registers, time on the lanes), so the fp32 accumulator tile of one GEMM, after bias +
// ReLU + bf16 packing, IS the B operand of the next MFMA chain: no LDS round trip for S and U.  The ROWS of Wskip /
// Wfinal (and their biases) are packed in accumulator-register order (packing.acc_k_perm), so the 8 packed elements of a
// lane are 8 consecutive channels: S and U exist in natural channel order, the K axes of Wfinal / Wzero are natural.
// Weights (and in phase 1 the o rows) stream through a D-slot LDS ring by LDS-DMA with counted vmcnt waits, shared by
// the NW waves of the workgroup.
//
// Training (a.save_*): the backward needs S, U (bf16 [M][256]) and Z = U Wz + bz (fp32 [M][2 Ch], log_s | t in plane
// channel order).  A lane stores its packed operands as 16-byte pieces right after each pack (32 rows x 32 bytes per
// wave instruction) and Z beside the coupling's plane stores.  The stores count in vmcnt like the DMA pieces, in issue
// order: the ring waits of the next LA chunks add them (step()).
//
// Ring sizing (round 3).  The 256-row form (8 waves) used to run two 64 KB slots - ONE chunk in flight: every chunk
// paid its DMA latency (16 x ~2 us per launch, 37 us at block 0).  Phase-1 chunks are now BK1 = 32 columns wide there
// (16 KB of weights + 16 KB of o rows), four slots, three chunks in flight.  Phases 2 / 3 use [256][64] weight chunks
// in the same slots; a ZeroConv of one pair tile (Ch <= 32: all its weights are 32 KB) is ONE chunk.
//
// HAS_P1 = false (small M, N-split tail): S = ReLU(skip sum) comes from a ring GEMM that splits the skip weights over
// workgroups (flow_kernels.hip TailLinProb); its rows arrive here as the first chunk and phases 2 / 3, the coupling and
// the chained front conv run as above - one launch instead of two.
//
// Chained front conv (a.h0_next): the coupling's out_b is the next flow's in_a (change_order, model.py:166-174,190).
// The conv needs rows n - 1, n, n + 1, so a workgroup computes its tile with ONE overlap row on either side
// (a.overlap: tiles of RW - 2 owned rows) and writes out_b to a third plane buffer (a.xb_out) - the neighbour still
// reads the old value of its halo row.  The conv itself is one more MFMA chain in the same transposed form:
// K = (tap, channel, hi | lo bf16 half of the fp32 state) <= 48 against a copy of the front weights packed in that
// order (packing.front3_src_k), B operand built from an LDS image of the tile's out_b; the 256 output channels of a
// lane's row leave through an LDS transposition as 16-byte stores.
#pragma once
#include "gemm_ring.h"

#ifndef FWN_TABL
#define FWN_TABL 0   // developer ablation (tools/diag/ab_tail.sh): 1 no MFMA / fragment reads, 2 no DMA after the prologue, 3 no epilogue
#endif

struct TailArgs {
    const bf16* o;        // [L][M][256]
    const bf16* Ws;       // [256][L*256]   rows in acc order (row n' = channel acc_k_perm(n'))
    const float* bs;      // [256]  (sum of the L skip biases)
    const bf16* Wf;       // [256][256]   rows in acc order, K natural
    const float* bfin;    // [256]
    const bf16* Wz;       // [npt*64][256] K natural; pair tiles: 32 log_s rows then 32 t rows
    const float* bz;      // [npt*64]
    const float* ez;      // [npt*64]  exp(3*scale)
    const float* an;      // [2][4][Ch]: (a|b) x (shift, scale, iscale, logs3)
    float* xa;            // plane holding in_a / out_a  [M][Ch]
    float* xb;            // plane holding in_b (and out_b unless xb_out)  [M][Ch]
    float* partial;       // [gridDim.x] log-det partial sums (forward) or nullptr
    long o_stride;        // elements between layers of o
    int L, M, Ch, npt, inverse;
    // ---- chain extensions (all optional) ----
    float* xb_out;        // where out_b goes (nullptr: in place)
    const bf16* S;        // HAS_P1 == false: S [M][256] = ReLU(skip sum)
    bf16* h0_next;        // != nullptr: also write the next flow's h0 = ReLU(front conv(out_b)) [M][256]
    const bf16* Wfn;      // next flow's front weights [256][kfn], k = (tap*Ch + tau)*2 + (hi|lo)
    const float* bfn;     // its bias [256]
    const float* an_next; // forward: the next flow's ActNorm table [2][4][Ch] (its a rows are applied); inverse: nullptr
    int kfn, Ti, overlap;
    // ---- what the training backward keeps (all optional) ----
    bf16* save_s;         // S = ReLU(skip sum)  [M][256]   (HAS_P1 only: otherwise S is a.S already)
    bf16* save_u;         // U = ReLU(final conv)  [M][256]
    float* save_z;        // Z = U Wz + bz  [M][2 Ch]: log_s channels, then t channels, plane order
};


// NW waves x 32 rows per workgroup, D ring slots, BK1 = phase-1 chunk width, WDB = double-buffered weight fragments
// (worth it at one wave per SIMD), NPT = ZeroConv pair tiles (Ch <= 32 NPT), HAS_P1 = the skip GEMM runs here.
template <int NW, int D, int BK1, bool WDB, int NPT, bool HAS_P1, bool FRONT = false>
__global__ __launch_bounds__(64 * NW) void tail_kernel(TailArgs a) {
    static_assert(!FRONT || NPT == 1, "the chained front conv serves Ch <= 8 (one ZeroConv pair tile)");
    using G1 = RingGeom<BK1>;
    constexpr int RW = 32 * NW;
    constexpr int RB1 = BK1 * 2;
    constexpr int W1_BYTES = 256 * RB1, O1_BYTES = RW * RB1;
    constexpr int W2_BYTES = 256 * 128;                        // phase 2 / 3 / front chunk: [256][64] bf16
    constexpr int S_BYTES = RW * 512;                          // HAS_P1 == false: the S' rows of this tile, 4 sub-tiles [RW][64]
    constexpr int SLOT0 = HAS_P1 ? W1_BYTES + O1_BYTES : S_BYTES;
    constexpr int SLOT = SLOT0 > W2_BYTES ? SLOT0 : W2_BYTES;
    constexpr bool FRONT_OK = FRONT;                           // the chained front conv (Ch <= 8 only)
    // constants, each table in 1-KB pieces of its own (an LDS-DMA piece always writes 1 KB):
    // bs | bfin | bz | ez | an (NPT pieces) | [bfn | an_next]
    constexpr int C_BS = 0, C_BF = 256, C_BZ = 512, C_EZ = 768, C_AN = 1024, C_BFN = C_AN + 256 * NPT, C_ANN = C_BFN + 256,
                  CST = FRONT_OK ? C_ANN + 256 : C_BFN;
    constexpr int NCP = 4 + NPT + (FRONT_OK ? 2 : 0);          // constant pieces
    constexpr int T_FLOATS = FRONT_OK ? RW * 8 : 0;            // front-conv input image [RW][8] fp32
    static_assert(D * SLOT >= RW * 512, "the h0 transposition tiles must fit the ring");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[D * SLOT + CST * 4 + T_FLOATS * 4 + 64];
    float* cst = (float*)(lds + D * SLOT);
    float* Tt = cst + CST;
    float* red = Tt + T_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    // tile rows [g0, g0 + RW); with a.overlap the first and the last row belong to the neighbours (read-only halo)
    const int g0 = a.overlap ? (int)blockIdx.x * (RW - 2) - 1 : (int)blockIdx.x * RW;
    const int rl = wave * 32 + lr;                   // this lane's row inside the tile
    const int row = g0 + rl;                         // ... and in the plane (may be -1 or >= M)
    const bool rvalid = (unsigned)row < (unsigned)a.M;
    const bool owned = rvalid && (!a.overlap || (rl >= 1 && rl <= RW - 2));
    const int KS = a.L * FWN_HID;
    const int Ch = a.Ch;
    constexpr bool front = FRONT;                    // (a.h0_next is set)

    // chunk sequence: [S rows] | phase 1 | phase 2 (4) | phase 3 (1 or 4) | [front weights]
    const int n0 = HAS_P1 ? 0 : 1;
    const int n1 = HAS_P1 ? KS / BK1 : 0;
    const int c2 = n0 + n1, c3 = c2 + 4;
    constexpr int N3 = NPT == 1 ? 1 : 4;
    const int c4 = c3 + N3;
    const int NC = c4 + (front ? 1 : 0);

    // ---- constants -> LDS by LDS-DMA, ahead of chunk 0 in the same queue (the first wait covers them): ordinary loads
    // here would have to retire before the first DMA is issued - a cold-miss stall at the head of every launch ----
    for (int pc = wave; pc < NCP; pc += NW) {
        const float* src;
        uint32_t bytes, off = 0;
        int dst;
        if (pc == 0) { src = a.bs; bytes = 1024u; dst = C_BS; }
        else if (pc == 1) { src = a.bfin; bytes = 1024u; dst = C_BF; }
        else if (pc == 2) { src = a.bz; bytes = (uint32_t)a.npt * 256u; dst = C_BZ; }
        else if (pc == 3) { src = a.ez; bytes = (uint32_t)a.npt * 256u; dst = C_EZ; }
        else if (pc < 4 + NPT) { src = a.an; bytes = (uint32_t)Ch * 32u; off = (uint32_t)(pc - 4) * 1024u; dst = C_AN + (pc - 4) * 256; }
        else if (pc == 4 + NPT) { src = front ? a.bfn : a.bs; bytes = front ? 1024u : 0u; dst = C_BFN; }
        else { src = (front && a.an_next) ? a.an_next : a.bs; bytes = (front && a.an_next) ? (uint32_t)Ch * 32u : 0u; dst = C_ANN; }
        buf_load16_lds(make_srd(src, bytes), off + (uint32_t)lane * 16u, (unsigned char*)(cst + dst));
    }
    // (NZ bz / ez use only the first npt*64 floats of their piece; C_BZ / C_EZ regions are 256 floats apart)
    static_assert(NPT * 64 <= 256, "bz / ez piece");

    // ---- DMA plans ----
    constexpr int PW1 = HAS_P1 ? (256 / G1::PR) / NW : 0;       // phase-1 weight pieces per wave
    constexpr int PO1 = HAS_P1 ? (RW / G1::PR) / NW : 0;        // phase-1 o pieces per wave
    constexpr int PS0 = HAS_P1 ? 0 : 16;                        // S pieces per wave (RW*512/1024 = 16 NW pieces)
    constexpr int PW2 = 32 / NW;                                // [256][64] chunk pieces per wave
    static_assert(!HAS_P1 || (PW1 * G1::PR * NW == 256 && PO1 * G1::PR * NW == RW), "phase-1 pieces must divide over the waves");
    const srd_t srd_s = make_srd(a.Ws, (uint32_t)(256u * KS * 2u));
    const srd_t srd_f = make_srd(a.Wf, 256u * 256u * 2u);
    const srd_t srd_z = make_srd(a.Wz, (uint32_t)(a.npt * 64u * 256u * 2u));
    const srd_t srd_o = make_srd(a.o, HAS_P1 ? (uint32_t)(((size_t)(a.L - 1) * a.o_stride + (size_t)a.M * FWN_HID) * 2) : 0u);
    const srd_t srd_S = make_srd(HAS_P1 ? (const void*)a.Ws : (const void*)a.S, HAS_P1 ? 0u : (uint32_t)((size_t)a.M * FWN_HID * 2));
    const srd_t srd_n = make_srd(front ? (const void*)a.Wfn : (const void*)a.Ws, front ? (uint32_t)(256u * a.kfn * 2u) : 0u);

    // pieces of a [256][64] chunk: piece j of this wave = rows 8 (wave + NW j) + lane / 8
    auto issue_w64 = [&](const srd_t s, int ld, int col, int kmax, int c, int part, int nparts) {
        unsigned char* dst = lds + (c % D) * SLOT;
#pragma unroll
        for (int j = 0; j < PW2; ++j)
            if (part < 0 || j * nparts / PW2 == part) {
                const int r = 8 * (wave + NW * j) + (lane >> 3);
                const int k = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
                buf_load16_lds(s, k < kmax ? (uint32_t)(r * ld + col + k) * 2u : FWN_OOB, dst + (wave + NW * j) * 1024);
            }
    };
    auto issue_chunk = [&](int c, int part, int nparts) {        // part < 0: the whole chunk
        if (c >= NC) return;
        if (FWN_TABL == 2 && c >= D - 1) return;
        unsigned char* dst = lds + (c % D) * SLOT;
        if (!HAS_P1 && c == 0) {                                 // S' rows: sub-tile q = [RW][64] at q * RW * 128
#pragma unroll
            for (int j = 0; j < PS0; ++j)
                if (part < 0 || j * nparts / (PS0 ? PS0 : 1) == part) {
                    const int pi = wave + NW * j;                // 0 .. 16 NW - 1: (q, 8-row group)
                    const int q = pi / (4 * NW), rr = (pi % (4 * NW)) * 8 + (lane >> 3);
                    const int k = q * 64 + ((lane & 7) ^ ((rr >> 1) & 7)) * 8;
                    const int gr = g0 + rr;
                    buf_load16_lds(srd_S, (unsigned)gr < (unsigned)a.M ? (uint32_t)(gr * FWN_HID + k) * 2u : FWN_OOB, dst + pi * 1024);
                }
        } else if (c < c2) {                                     // phase 1: Ws columns [q BK1, +BK1) and the o rows of (layer, k)
            if constexpr (HAS_P1) {
                const int q = c - n0;
                constexpr int CPL = FWN_HID / BK1;
                const uint32_t obase = (uint32_t)((q / CPL) * a.o_stride + (q % CPL) * BK1);
#pragma unroll
                for (int j = 0; j < PW1 + PO1; ++j)
                    if (part < 0 || j * nparts / (PW1 + PO1) == part) {
                        if (j < PW1) {
                            const int r = G1::PR * (wave + NW * j) + G1::piece_row(lane);
                            buf_load16_lds(srd_s, (uint32_t)(r * KS + q * BK1 + G1::piece_c(lane, r) * 8) * 2u, dst + (wave + NW * j) * 1024);
                        } else {
                            const int jo = j - PW1;
                            const int r = G1::PR * (wave + NW * jo) + G1::piece_row(lane);
                            const int gr = g0 + r;
                            buf_load16_lds(srd_o, (unsigned)gr < (unsigned)a.M ? (obase + (uint32_t)(gr * FWN_HID + G1::piece_c(lane, r) * 8)) * 2u : FWN_OOB,
                                           dst + W1_BYTES + (wave + NW * jo) * 1024);
                        }
                    }
            }
        } else if (c < c3) {
            issue_w64(srd_f, FWN_HID, (c - c2) * 64, 64, c, part, nparts);
        } else if (c < c4) {
            if constexpr (NPT == 1) {                            // all of Wz [64][256] as 4 sub-tiles [64][64] of 8 KB
#pragma unroll
                for (int j = 0; j < PW2; ++j)
                    if (part < 0 || j * nparts / PW2 == part) {
                        const int pi = wave + NW * j, q = pi >> 3;
                        const int r = (pi & 7) * 8 + (lane >> 3);
                        buf_load16_lds(srd_z, (uint32_t)(r * FWN_HID + q * 64 + ((lane & 7) ^ ((r >> 1) & 7)) * 8) * 2u, dst + pi * 1024);
                    }
            } else {
                issue_w64(srd_z, FWN_HID, (c - c3) * 64, 64, c, part, nparts);
            }
        } else {
            issue_w64(srd_n, a.kfn, 0, a.kfn, c, part, nparts);
        }
    };
    auto pieces = [&](int c) -> int { return c >= NC ? 0 : (!HAS_P1 && c == 0) ? PS0 : c < c2 ? PW1 + PO1 : PW2; };
    // wait for chunk c (issued so far: chunks .. c + D - 2; those after c may stay in flight) and cross the barrier
    constexpr int LA = D - 1;                                    // refill distance
    // saved operands (training): 16 stores per lane, issued when chunks .. sv_upto had been issued - they are younger than
    // those chunks' pieces and older than every later chunk's
    constexpr int NSV = 16;
    int sv_upto = -1;
    auto step = [&](int c) {
        if (FWN_TABL == 2) { if (c == 0) FWN_WAIT_VMCNT(0); FWN_RING_BARRIER(); return; }
        int pend = c <= sv_upto ? NSV : 0;
#pragma unroll
        for (int i = 1; i <= D - 2; ++i) pend += pieces(c + i);
        fwn_wait_vm_le(pend);
        FWN_RING_BARRIER();
    };
    auto save_pk = [&](bf16* dst, const bf16x8 (&pk_)[8][2], int issued_upto) {
        const srd_t sd = make_srd(dst, (uint32_t)((size_t)a.M * FWN_HID * 2));
        const uint32_t base = owned ? (uint32_t)(row * FWN_HID + lh * 8) * 2u : FWN_OOB;
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pk_[ct][s]), sd, base, (ct * 32 + s * 16) * 2, 0);
        sv_upto = issued_upto;
    };

    // fragment offsets inside a 32-row tile: the swizzles depend on the row only through lr
    int wfrag[4], wfrag1[G1::KS];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) wfrag[kk] = lr * 128 + (((kk * 2 + lh) ^ ((lr >> 1) & 7)) << 4);
#pragma unroll
    for (int kk = 0; kk < G1::KS; ++kk) wfrag1[kk] = G1::off(lr, kk * 2 + lh);
#define WFRAG(wb, t, kk) (*(const bf16x8*)((wb) + wfrag[kk] + (t) * 4096))
#define WFRAG1(wb, t, kk) (*(const bf16x8*)((wb) + wfrag1[kk] + (t) * (32 * RB1)))

    f32x16 acc[8];
    auto init_acc = [&](const float* bias) {         // acc[ct][r] = bias[ct*32 + acc_row(r)]
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = *(const float4*)(bias + ct * 32 + 8 * g + 4 * lh);
                acc[ct][4 * g + 0] = v.x; acc[ct][4 * g + 1] = v.y; acc[ct][4 * g + 2] = v.z; acc[ct][4 * g + 3] = v.w;
            }
    };
    bf16x8 pk[8][2];                                  // packed activations: B operands of the next chain
    auto pack_relu = [&]() {
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                Pack16 t;
#pragma unroll
                for (int j = 0; j < 8; ++j) t.e[j] = (bf16)fmaxf(acc[ct][8 * s + j], 0.0f);
                pk[ct][s] = t.v;
            }
    };
    bf16x8 wf[2][8];
    auto ldw = [&](const unsigned char* wb, int kk, int s) {
#pragma unroll
        for (int t = 0; t < 8; ++t) wf[s][t] = WFRAG(wb, t, kk);
    };

#pragma unroll
    for (int c = 0; c < LA; ++c) issue_chunk(c, -1, 1);
    step(0);

    if constexpr (HAS_P1) {
        // ---------------- phase 1: S^T = Ws @ [o_0 | o_1 | ..]^T + bs ----------------
        init_acc(cst + C_BS);
        for (int c = 0; c < n1; ++c) {
            if (c > 0) step(c);
            const unsigned char* wb = lds + (c % D) * SLOT;
            const unsigned char* ob = wb + W1_BYTES + wave * (32 * RB1);
            if (FWN_TABL == 1) { issue_chunk(c + LA, -1, 1); continue; }
            bf16x8 wf1[2][8], bq[G1::KS];
            if (WDB) {
                // the chunk's o fragments up front (one exposed LDS latency per chunk instead of one per k-step: at one wave per
                // SIMD nothing else covers it), then the first weight fragments
#pragma unroll
                for (int kk = 0; kk < G1::KS; ++kk) bq[kk] = *(const bf16x8*)(ob + wfrag1[kk]);
#pragma unroll
                for (int t = 0; t < 8; ++t) wf1[0][t] = WFRAG1(wb, t, 0);
            }
#pragma unroll
            for (int kk = 0; kk < G1::KS; ++kk) {
                const bf16x8 b = WDB ? bq[kk] : *(const bf16x8*)(ob + wfrag1[kk]);
                if (WDB) {
                    if (kk + 1 < G1::KS) {
#pragma unroll
                        for (int t = 0; t < 8; ++t) wf1[(kk + 1) & 1][t] = WFRAG1(wb, t, kk + 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);   // keep the next step's reads ahead of these MFMAs
                }
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) {
                    acc[ct] = mfma32(WDB ? wf1[kk & 1][ct] : WFRAG1(wb, ct, kk), b, acc[ct]);
                    if (ct == 0) issue_chunk(c + LA, kk, G1::KS);
                }
                if (WDB) __builtin_amdgcn_sched_barrier(0);
            }
        }
        pack_relu();
        if (a.save_s) save_pk(a.save_s, pk, c2 - 1 + LA);
    } else {
        // S rows of this wave straight into the operand registers: sub-tile q, k-step kk -> pk[2 q + kk / 2][kk % 2]
        const unsigned char* sb = lds + wave * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) pk[2 * q + (kk >> 1)][kk & 1] = *(const bf16x8*)(sb + q * (RW * 128) + wfrag[kk]);
        issue_chunk(LA, -1, 1);                          // slot D - 1 has not been used yet
    }
    init_acc(cst + C_BF);

    // ---------------- phase 2: U^T = Wf @ S^T + bfin ----------------
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
        const int c = c2 + kc;
        if (HAS_P1 || kc > 0) step(c);
        else {                                           // HAS_P1 == false, first chunk: the S reads must have left the slot first
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            step(c);
        }
        const unsigned char* wb = lds + (c % D) * SLOT;
        if (FWN_TABL == 1) { issue_chunk(c + LA, -1, 1); continue; }
        if (WDB) ldw(wb, 0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (WDB) {
                if (kk < 3) ldw(wb, kk + 1, (kk + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ot = 0; ot < 8; ++ot) {
                acc[ot] = mfma32(WDB ? wf[kk & 1][ot] : WFRAG(wb, ot, kk), pk[2 * kc + (kk >> 1)][kk & 1], acc[ot]);
                if (ot == 0) issue_chunk(c + LA, kk, 4);
            }
            if (WDB) __builtin_amdgcn_sched_barrier(0);
        }
    }
    pack_relu();
    if (a.save_u) save_pk(a.save_u, pk, c3 - 1 + LA);

    // ---------------- phase 3: [log_s | t]^T = Wz @ U^T ----------------
    constexpr int NTZ = 2 * NPT;
#pragma unroll
    for (int tz = 0; tz < NTZ; ++tz)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tz][r] = 0.0f;
    if constexpr (NPT == 1) {
        step(c3);
        issue_chunk(c3 + LA, -1, 1);
        const unsigned char* wb = lds + (c3 % D) * SLOT;
        // the 8 fragments of sub-tile q + 1 are read while the 8 MFMAs of sub-tile q issue
        bf16x8 zf[2][4][2];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int tz = 0; tz < 2; ++tz) zf[0][kk][tz] = WFRAG(wb, tz, kk);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q + 1 < 4) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int tz = 0; tz < 2; ++tz) zf[(q + 1) & 1][kk][tz] = WFRAG(wb + (q + 1) * 8192, tz, kk);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int tz = 0; tz < 2; ++tz)
                    acc[tz] = mfma32(zf[q & 1][kk][tz], pk[2 * q + (kk >> 1)][kk & 1], acc[tz]);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            step(c3 + kc);
            issue_chunk(c3 + kc + LA, -1, 1);
            const unsigned char* wb = lds + ((c3 + kc) % D) * SLOT;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int tz = 0; tz < NTZ; ++tz)
                    acc[tz] = mfma32(WFRAG(wb, tz, kk), pk[2 * kc + (kk >> 1)][kk & 1], acc[tz]);
        }
    }

    // The chained front conv's weights were issued chunks ago: retire them NOW, while loads are the only vector-memory
    // operations outstanding.  After the coupling a vmcnt wait would also wait for the epilogue's plane stores (vmcnt
    // counts stores: ~2 us of store latency in front of the conv).
    if constexpr (FRONT) FWN_WAIT_VMCNT(0);
    if (FWN_TABL == 3) {
        float sacc = 0.0f;
#pragma unroll
        for (int tz = 0; tz < 2 * NPT; ++tz)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc += acc[tz][r];
        if (sacc != 12345.678f) return;
    }
    // ---------------- affine coupling + ActNorm on the b plane ----------------
    // Buffer loads / stores: elements outside the plane (tau >= Ch, rows not owned) get an out-of-range offset, read
    // as 0 and are dropped on store - no branches.
    const float* bzl = cst + C_BZ;
    const float* ezl = cst + C_EZ;
    const float* an_a = cst + C_AN;
    const float* an_b = an_a + 4 * Ch;
    const float* ann = cst + C_ANN;                  // next flow's table: a rows = shift[Ch] | scale[Ch]
    const uint32_t plane_bytes = (uint32_t)((size_t)a.M * Ch * 4);
    const srd_t sxb = make_srd(a.xb, plane_bytes), sxa = make_srd(a.xa, plane_bytes);
    const srd_t sxo = make_srd(a.xb_out ? a.xb_out : a.xb, plane_bytes);
    float lsum = 0.0f;
    const bool vec4 = Ch >= 4;          // 4 consecutive channels per accumulator register group
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
        float xv[16];
        uint32_t voff[16];
        if (vec4) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int tau0 = pt * 32 + 8 * g + 4 * lh;
                const bool ok = tau0 < Ch && rvalid;
                const float4 q = *(const float4*)(a.xb + (ok ? (size_t)row * Ch + tau0 : 0));   // clamped
                xv[4 * g + 0] = q.x; xv[4 * g + 1] = q.y; xv[4 * g + 2] = q.z; xv[4 * g + 3] = q.w;
                voff[4 * g] = (tau0 < Ch && owned) ? (uint32_t)(row * Ch + tau0) * 4u : FWN_OOB;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tau = pt * 32 + acc_row(r, lane);
                xv[r] = buf_load_f32(sxb, (tau < Ch && rvalid) ? (uint32_t)(row * Ch + tau) * 4u : FWN_OOB, 0);
                voff[r] = (tau < Ch && owned) ? (uint32_t)(row * Ch + tau) * 4u : FWN_OOB;
            }
        }
        float ov[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = acc_row(r, lane);
            const int tau = pt * 32 + j;
            const bool ok = tau < Ch && owned;
            const int tc = tau < Ch ? tau : 0;
            const int nls = pt * 64 + j, nt = nls + 32;
            const float ls = (acc[2 * pt][r] + bzl[nls]) * ezl[nls];
            const float t = (acc[2 * pt + 1][r] + bzl[nt]) * ezl[nt];
            if (!a.inverse) {
                const float yb = (xv[r] + an_b[tc]) * an_b[Ch + tc];              // ActNorm (model.py:86-94)
                ov[r] = (yb - t) * __expf(-ls);                                     // model.py:134
                lsum += ok ? (an_a[3 * Ch + tc] + an_b[3 * Ch + tc] - ls) : 0.0f;  // model.py:135 + :80
            } else {
                const float yb = xv[r] * __expf(ls) + t;                            // model.py:156
                ov[r] = yb * an_b[2 * Ch + tc] - an_b[tc];                         // ActNorm^-1 (model.py:97-102)
            }
        }
        if (vec4) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u32x4 o4 = {__builtin_bit_cast(unsigned int, ov[4 * g]), __builtin_bit_cast(unsigned int, ov[4 * g + 1]),
                                  __builtin_bit_cast(unsigned int, ov[4 * g + 2]), __builtin_bit_cast(unsigned int, ov[4 * g + 3])};
                __builtin_amdgcn_raw_buffer_store_b128(o4, sxo, voff[4 * g], 0, 0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) buf_store_f32(sxo, voff[r], 0, ov[r]);
        }
        if (a.save_z) {                   // Z = U Wz + bz for the training backward: [row][tau] = log_s, [row][Ch + tau] = t (before exp(3 scale))
            const srd_t sz = make_srd(a.save_z, (uint32_t)((size_t)a.M * 2 * Ch * 4));
            const int tsoff = Ch * 4;
            if (vec4) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int j0 = 8 * g + 4 * lh;
                    const uint32_t zoff = voff[4 * g] != FWN_OOB ? (uint32_t)(row * 2 * Ch + pt * 32 + j0) * 4u : FWN_OOB;
                    u32x4 zl, zt;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        zl[e] = __builtin_bit_cast(unsigned int, acc[2 * pt][4 * g + e] + bzl[pt * 64 + j0 + e]);
                        zt[e] = __builtin_bit_cast(unsigned int, acc[2 * pt + 1][4 * g + e] + bzl[pt * 64 + 32 + j0 + e]);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(zl, sz, zoff, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(zt, sz, zoff, tsoff, 0);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = acc_row(r, lane);
                    const uint32_t zoff = voff[r] != FWN_OOB ? (uint32_t)(row * 2 * Ch + pt * 32 + j) * 4u : FWN_OOB;
                    buf_store_f32(sz, zoff, 0, acc[2 * pt][r] + bzl[pt * 64 + j]);
                    buf_store_f32(sz, zoff, tsoff, acc[2 * pt + 1][r] + bzl[pt * 64 + 32 + j]);
                }
            }
        }
        if (pt == 0 && front) {
            // the tile's out_b (every row, halo included) as the next flow's network input: ActNorm of that flow applied
            // in the forward direction (model.py:86-94 ahead of its coupling), raw in the inverse direction
#pragma unroll
            for (int r = 0; r < 4; ++r) {                // channels < 8 live in registers 0..3 (tau = r + 4 lh)
                const int tau = r + 4 * lh;
                if (tau < Ch) {
                    const float v = a.an_next ? (ov[r] + ann[tau]) * ann[Ch + tau] : ov[r];
                    Tt[rl * 8 + tau] = rvalid ? v : 0.0f;
                }
            }
        }
    }
    // a-plane: ActNorm only (the coupling passes in_a through unchanged), owned rows [ra, rb)
    {
        const int ra = max(a.overlap ? g0 + 1 : g0, 0), rb = min(a.overlap ? g0 + RW - 1 : g0 + RW, a.M);
        const int total = (rb - ra) * Ch;
        const uint32_t base = (uint32_t)(ra * Ch) * 4u;
        const int chmask = Ch - 1;
        // ra * Ch is a multiple of 4 whenever Ch >= 4
        if (Ch >= 4) {
            // NOTE: the 16-byte load is a plain (clamped) float4 load, not raw_buffer_load_b128: hipcc (ROCm 7.2) lowers
            // element extracts of that builtin's result to ONE buffer_load_dword reused for all four lanes.
#pragma unroll 4
            for (int idx = tid * 4; idx < total; idx += 64 * NW * 4) {
                const float4 q = *(const float4*)(a.xa + (size_t)ra * Ch + idx);
                float f[4] = {q.x, q.y, q.z, q.w};
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
            for (int idx = tid; idx < total; idx += 64 * NW) {
                const int tau = idx & chmask;
                const uint32_t off = base + (uint32_t)idx * 4u;
                const float v = buf_load_f32(sxa, off, 0);
                buf_store_f32(sxa, off, 0, a.inverse ? (v * an_a[2 * Ch + tau] - an_a[tau]) : ((v + an_a[tau]) * an_a[Ch + tau]));
            }
        }
    }
    if (a.partial) {
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) lsum += __shfl_xor(lsum, s);
        if (lane == 0) red[wave] = lsum;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS only: __syncthreads() would also wait for the plane stores
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tid == 0) {
            float t = 0.0f;
            for (int w = 0; w < NW; ++w) t += red[w];
            a.partial[blockIdx.x] = t;
        }
    }
    if constexpr (!FRONT) return;
    else {
    // ---------------- the next flow's front conv: h0^T = Wfn @ [taps of out_b as hi | lo]^T + bfn, ReLU ----------------
    init_acc(cst + C_BFN);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's T rows are written
    __builtin_amdgcn_s_barrier();                         // every wave's T rows are visible (the weights landed before the coupling)
    asm volatile("" ::: "memory");
    {
        const unsigned char* wb = lds + (c4 % D) * SLOT;
        const int chlog = 31 - __builtin_clz(Ch);
        const int t_in = rvalid ? row % a.Ti : 0;
        const int nks = a.kfn >> 4;
        for (int kk = 0; kk < nks; ++kk) {
            Pack16 b;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = kk * 8 + lh * 4 + e;     // (tap, tau) of this element pair
                const int tap = idx >> chlog, tau = idx & (Ch - 1);
                const int rr = rl + tap - 1;
                const bool ok = idx < 3 * Ch && (unsigned)(t_in + tap - 1) < (unsigned)a.Ti && (unsigned)rr < (unsigned)RW;
                const float v = ok ? Tt[(ok ? rr : 0) * 8 + tau] : 0.0f;
                const bf16 hi = (bf16)v;
                b.e[2 * e] = hi;
                b.e[2 * e + 1] = (bf16)(v - (float)hi);
            }
#pragma unroll
            for (int ct = 0; ct < 8; ++ct)
                acc[ct] = mfma32(*(const bf16x8*)(wb + lr * 128 + ((((kk * 2 + lh) ^ ((lr >> 1) & 7))) << 4) + ct * 4096), b.v, acc[ct]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // every wave is done with the weights and T: the ring is free
    asm volatile("" ::: "memory");
    {
        // this wave's 32 rows x 256 channels as bf16 [32][256] in LDS (lds_off256 image), then 16-byte row pieces out
        unsigned char* tw = lds + wave * (32 * 512);
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                union { bf16 e[4]; uint2 u; } o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) o4.e[e] = (bf16)fmaxf(acc[ct][4 * g + e], 0.0f);
                const int chn = ct * 32 + 8 * g + 4 * lh;            // 4 consecutive channels of row lr
                *(uint2*)(tw + lds_off256(lr, chn >> 3) + (chn & 7) * 2) = o4.u;
            }
        const srd_t sh = make_srd(a.h0_next, (uint32_t)((size_t)a.M * FWN_HID * 2));
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int r2 = it * 2 + lh;                              // row of the wave's tile, 16-byte piece lr of it
            const u32x4 v = *(const u32x4*)(tw + lds_off256(r2, lr));
            const int trow = wave * 32 + r2, grow = g0 + trow;
            const bool own = (unsigned)grow < (unsigned)a.M && (!a.overlap || (trow >= 1 && trow <= RW - 2));
            __builtin_amdgcn_raw_buffer_store_b128(v, sh, own ? (uint32_t)(grow * FWN_HID + lr * 8) * 2u : FWN_OOB, 0, 0);
        }
    }
    }
#undef WFRAG
#undef WFRAG1
}
