// Deep-pipelined bf16 MFMA GEMM core for gfx950: LDS-DMA ring with counted vmcnt waits.
//
// Why a ring: on MI355X an LDS-DMA piece lands ~1.1-1.5 us after issue, and one CU only reaches
// its L2->LDS rate (~66 GB/s) with ~100 KB in flight.  A workgroup that waits for its single
// outstanding K-chunk every iteration (vmcnt(0) + barrier) therefore runs at one chunk per DMA
// latency whatever the tile.  Here D ring slots hold D-1 chunks in flight; iteration q waits only
// for chunk q (s_waitcnt vmcnt((D-2)*PW), PW = DMA pieces per wave per chunk), crosses ONE raw
// s_barrier (which also proves every wave is done with slot (q-1)%D), re-issues chunk q+D-1 into
// that slot and multiplies chunk q.  No ordinary VGPR-destination loads appear inside the loop
// (hipcc would drain the DMA queue at their first use), and all LDS lives in one array.
//
// Tile: BM x BN, waves laid out WM x WN, wave tile (32*MI) x 64 with MI = BM/(32*WM); BN = 64*WN.
// K-chunk BK = 32 or 64 bf16 (64- or 128-byte LDS rows, XOR-swizzled on the DMA source side).
#pragma once
#include "common.h"

template <int BK>
struct RingGeom {
    static constexpr int RB = BK * 2;            // row bytes
    static constexpr int PR = 1024 / RB;         // rows per 1-KiB DMA piece
    static constexpr int CPR = RB / 16;          // 16-byte pieces per row
    static constexpr int KS = BK / 16;           // MFMA k-steps per chunk
    // byte offset of 16-byte piece c of row `row` in a [rows][BK] tile
    static __device__ __forceinline__ int off(int row, int c) {
        if constexpr (BK == 128) return row * 256 + ((c ^ (row & 15)) << 4);
        else if constexpr (BK == 64) return row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
        else return row * 64 + ((c ^ ((row >> 2) & 3)) << 4);
    }
    // lane -> (row within piece, global 16-byte piece index it must fetch for its linear LDS slot)
    static __device__ __forceinline__ int piece_row(int lane) { return lane / CPR; }
    static __device__ __forceinline__ int piece_c(int lane, int row) {
        if constexpr (BK == 128) return (lane & 15) ^ (row & 15);
        else if constexpr (BK == 64) return (lane & 7) ^ ((row >> 1) & 7);
        else return (lane & 3) ^ ((row >> 2) & 3);
    }
};

// Developer ablation switch (tools/bench_gemm.hip builds with -DFWN_ABL=n; the product is 0):
//   1 = no MFMA/ds_read (DMA + waits + barriers only), 2 = DMA issued only in the prologue
//   (MFMA on stale LDS), 3 = no DMA at all and no waits, 4 = 3 without the epilogue,
//   5 = 4 without the per-chunk barrier.
#ifndef FWN_ABL
#define FWN_ABL 0
#endif
// 6 = everything but the epilogue
#define FWN_ABL_DMA (FWN_ABL < 2 || FWN_ABL == 6)
#ifndef FWN_SETPRIO
#define FWN_SETPRIO 0
#endif

#define FWN_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
// The barrier of a ring: behind it a slot that waves may still have been READING is refilled (LDS-DMA) or reused
// (ds_write).  An s_barrier orders nothing by itself, and hipcc is free to sink the consumers of a ds_read (MFMAs are
// register-only instructions) together with the s_waitcnt lgkmcnt that retires it BELOW a raw s_barrier: the wave then
// arrives with reads of the old slot contents in flight and the refill can overtake them.  That was the round-3
// front_mfma_kernel finding (wrong weights in ~1 overlapped step in 50, only with another kernel on
// the CU slowing the LDS); tools/check_barrier_lgkm.py finds the pattern in the ISA.  Every ring barrier therefore retires
// the wave's own LDS operations first - free where nothing is in flight, and exactly the missing wait where something is.
#define FWN_RING_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
// n wave-uniform (run time): wait until at most n vector-memory operations are outstanding (waiting for fewer is safe)
__device__ __forceinline__ void fwn_wait_vm_le(int n) {
    // n wave-uniform: wait until at most n vector-memory operations are outstanding (waiting for fewer is safe)
    if (n >= 48) FWN_WAIT_VMCNT(48);
    else if (n >= 40) FWN_WAIT_VMCNT(40);
    else if (n >= 32) FWN_WAIT_VMCNT(32);
    else if (n >= 28) FWN_WAIT_VMCNT(28);
    else if (n >= 24) FWN_WAIT_VMCNT(24);
    else if (n >= 20) FWN_WAIT_VMCNT(20);
    else if (n >= 16) FWN_WAIT_VMCNT(16);
    else if (n >= 12) FWN_WAIT_VMCNT(12);
    else if (n >= 8) FWN_WAIT_VMCNT(8);
    else if (n >= 6) FWN_WAIT_VMCNT(6);
    else if (n >= 4) FWN_WAIT_VMCNT(4);
    else if (n >= 2) FWN_WAIT_VMCNT(2);
    else if (n >= 1) FWN_WAIT_VMCNT(1);
    else FWN_WAIT_VMCNT(0);
}

#ifdef FWN_STAMP      // tools/probe/lin_stamps.hip: s_memtime per wave of workgroup 0: [chunk][before wait | after wait | after barrier | after MFMAs]
__device__ unsigned long long fwn_ring_stamps[16 * 64 * 4 + 64];
#define FWN_RING_STAMP(q, k) do { if (blockIdx.x == 0 && blockIdx.z == 0 && (q) < 64 && lane == 0) fwn_ring_stamps[((wave) * 64 + (q)) * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define FWN_RING_STAMP_X(i) do { if (blockIdx.x == 0 && blockIdx.z == 0 && lane == 0) fwn_ring_stamps[16 * 64 * 4 + (wave) * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FWN_RING_STAMP(q, k)
#define FWN_RING_STAMP_X(i)
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt_le(int pending_chunks) {
    // pending_chunks (wave-uniform) in {0, 1, .., 7}: wait until at most pending*N DMAs remain (vmcnt is a 6-bit
    // counter: depths whose product exceeds 63 wait for the deepest count that still fits)
#define FWN_VM_CASE(c)                                                                       \
    if constexpr ((c) * N <= 63) { if (pending_chunks >= (c)) { FWN_WAIT_VMCNT((c) * N <= 63 ? (c) * N : 0); return; } }
    FWN_VM_CASE(7) FWN_VM_CASE(6) FWN_VM_CASE(5) FWN_VM_CASE(4) FWN_VM_CASE(3) FWN_VM_CASE(2) FWN_VM_CASE(1)
#undef FWN_VM_CASE
    FWN_WAIT_VMCNT(0);
}

// ---- row-major epilogue helpers (LDS_EPI problems) ----
// A wave's (32 MI) x 64 fp32 tile in LDS: 256-byte rows, the 16-byte granule g of a row stored at g ^ ((row >> 1) & 1)
// so that both the 4-byte column writes (32 consecutive floats of one row per half wave) and the 16-byte row reads
// (ds_read_b128 is served in groups of 16 lanes that span four rows) are bank-conflict-free.
template <class P, class = void> struct prob_lds_epi { static constexpr bool value = false; };
template <class P> struct prob_lds_epi<P, decltype((void)P::LDS_EPI)> { static constexpr bool value = P::LDS_EPI; };

// Small tiles (KSP > 1: launches of a handful of workgroups - latency chains from end to end): a problem that declares
// PREFETCH requests the operands of its epilogue (hoisted conditioning projection, residual rows, flow state) right behind
// the prologue DMAs, 32 floats per 32-row tile and lane, and gets them back in epilogue_pre(): their round trip runs under
// the K loop instead of after it (~0.7 us of a 5 - 8 us launch).  Ordinary loads: they retire in issue order with the DMA
// pieces, so the first ring wait also covers them (everything of the prologue lands together); the compiler's own wait
// sits at their first use, after the loop.  Big tiles keep their loads in the epilogue: ahead of the K loop of an HBM-bound
// launch they compete with its LDS-DMA stream (res layer of block 0: 21.0 -> 24.7 us).
// A problem whose B operand (weights) is a long stream that few workgroups read once declares NT_B (hoisted conditioning)
template <class P, class = void> struct prob_nt_b { static constexpr bool value = false; };
template <class P> struct prob_nt_b<P, decltype((void)P::NT_B)> { static constexpr bool value = P::NT_B; };
template <class P, class = void> struct prob_prefetch { static constexpr bool value = false; };
template <class P> struct prob_prefetch<P, decltype((void)P::PREFETCH)> { static constexpr bool value = P::PREFETCH; };

template <int MI>
__device__ __forceinline__ void lds_epi_park(const f32x16 (&acc)[MI][2], float* wt, int lane) {
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mi * 32 + acc_row_c(r) + 4 * lh;            // (row >> 1) & 1 == (r >> 1) & 1
                wt[row * 64 + ((ni * 32 + lr) ^ (((r >> 1) & 1) << 2))] = acc[mi][ni][r];
            }
}
// Row piece `it` (0 .. 4 MI - 1) of the parked tile: this lane's row (it * 8 + lane / 8) and 8 consecutive columns
// (lane % 8) * 8 .. + 7, as two float4.  The wave's own writes are complete once lgkmcnt drains (same wave: no barrier).
__device__ __forceinline__ void lds_epi_take(const float* wt, int it, int lane, float (&v)[8]) {
    const int row = it * 8 + (lane >> 3), c = lane & 7, f = (lane >> 4) & 1;
    const float4 a = *(const float4*)(wt + row * 64 + (((2 * c) ^ f) << 2));
    const float4 b = *(const float4*)(wt + row * 64 + (((2 * c + 1) ^ f) << 2));
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// KSP > 1: intra-workgroup split-K for small tiles - KSP wave groups take alternate k-steps of
// every chunk (more waves per CU to hide LDS latency on an otherwise 2-wave tile) and their
// partial accumulators are summed through LDS before the epilogue.
template <int BM, int BN, int WM, int WN, int BK, int D, class Prob, int KSP = 1>
__device__ __forceinline__ void gemm_ring_body(const Prob& p, int tile_m, int tile_n) {
    using G = RingGeom<BK>;
    constexpr int NWV = WM * WN * KSP;
    static_assert(G::KS % KSP == 0, "k-steps must divide over the split");
    constexpr int MI = BM / (32 * WM);
    static_assert(BN == 64 * WN, "wave tile is 64 columns wide");
    static_assert(MI >= 1 && BM == 32 * MI * WM, "bad BM");
    constexpr int A_BYTES = BM * G::RB, B_BYTES = BN * G::RB, SLOT = A_BYTES + B_BYTES;
    constexpr int PA = BM / (G::PR * NWV);       // A pieces per wave per chunk
    constexpr int PB = BN / (G::PR * NWV);       // B pieces per wave per chunk
    static_assert(PA * G::PR * NWV == BM && PB * G::PR * NWV == BN, "pieces must divide evenly over waves");
    constexpr int PW = PA + PB;
    static_assert(D >= 2 && D <= 8, "ring depth");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[D * SLOT];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave / (WM * WN), wmn = wave % (WM * WN);
    const int wm = wmn / WN, wn = wmn % WN;
    const int lr = lane & 31, lh = lane >> 5;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // DMA piece j of this wave: tile rows [PR*(wave + NWV*j), +PR)
    int arow[PA], ac[PA], brow[PB], bc[PB];
    typename Prob::RowCtx rc[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        arow[j] = G::PR * (wave + NWV * j) + G::piece_row(lane);
        ac[j] = G::piece_c(lane, arow[j]);
        rc[j] = p.row_ctx(m0 + arow[j]);
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        brow[j] = G::PR * (wave + NWV * j) + G::piece_row(lane);
        bc[j] = G::piece_c(lane, brow[j]);
    }
    const int nq = p.template nchunks<BK>();
    FWN_RING_STAMP_X(0);

    // piece j (0..PW-1) of chunk q: A pieces first, then B pieces
    auto issue_piece = [&](const typename Prob::ChunkCtx& cc, int q, int j) {
        unsigned char* la = lds + (q % D) * SLOT;
        if (j < PA) {
            buf_load16_lds(p.a_srd(cc), p.a_voff(cc, rc[j < PA ? j : 0], ac[j < PA ? j : 0]),
                           la + (wave + NWV * j) * 1024);
        } else {
            const int jb = j - PA;
            // small tiles (KSP > 1: a handful of workgroups per launch): the weight rows are read once -> nt (FWN_NT_SMALL)
            if constexpr (KSP > 1 || prob_nt_b<Prob>::value)
                buf_load16_lds_nt(p.b_srd(cc), p.b_voff(cc, n0 + brow[jb < PB ? jb : 0], bc[jb < PB ? jb : 0]),
                                  la + A_BYTES + (wave + NWV * jb) * 1024);
            else
                buf_load16_lds(p.b_srd(cc), p.b_voff(cc, n0 + brow[jb < PB ? jb : 0], bc[jb < PB ? jb : 0]),
                               la + A_BYTES + (wave + NWV * jb) * 1024);
        }
    };
    auto issue = [&](int q) {
        typename Prob::ChunkCtx cc = p.template chunk_ctx<BK>(q);
#pragma unroll
        for (int j = 0; j < PW; ++j) issue_piece(cc, q, j);
    };


    // fragment addresses: the swizzle sees only lr (32-row tiles preserve the low row bits)
    // (indexed by this wave's own k-step counter ki: k-step wk + ki*KSP of the chunk)
    constexpr int KSW = G::KS / KSP;
    int afr[KSW], bfr_[KSW];
#pragma unroll
    for (int ki = 0; ki < KSW; ++ki) {
        afr[ki] = G::off(wm * 32 * MI + lr, (wk + ki * KSP) * 2 + lh);
        bfr_[ki] = G::off(wn * 64 + lr, (wk + ki * KSP) * 2 + lh);
    }

#pragma unroll
    for (int q = 0; q < D - 1; ++q)
        if (q < nq) issue(q);

    // Accumulators start from the problem's per-column constant (the bias, where the epilogue would
    // otherwise add it per element); with split-K only the first wave group carries it.  Loaded
    // AFTER the prologue DMAs are in the queue: ahead of them, the wait for these (cold) loads would
    // hold back the first DMA issue of every launch.
    f32x16 acc[MI][2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const float c0 = wk == 0 ? p.acc_init(n0 + wn * 64 + ni * 32 + lr) : 0.0f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = c0;
    }

    constexpr bool PF = KSP > 1 && prob_prefetch<Prob>::value;
    float pre[PF ? MI : 1][32];
    if constexpr (PF) {
        if (wk == 0) p.template prefetch<MI>(pre, m0 + wm * 32 * MI, n0 + wn * 64, lane);
    }

    // context of the chunk the NEXT refill loads: looked up one iteration ahead (a problem whose chunk_ctx reads a
    // descriptor table - fwn_gemm's segments - then has its scalar loads in flight under the previous chunk's MFMAs)
    typename Prob::ChunkCtx ccn = p.template chunk_ctx<BK>(D - 1 < nq ? D - 1 : 0);
    for (int q = 0; q < nq; ++q) {
        // chunks issued so far: min(nq, q + D - 1); those after q may stay in flight
        const int pending = min(nq, q + D - 1) - (q + 1);
        FWN_RING_STAMP(q, 0);
        if (FWN_ABL_DMA) wait_vmcnt_le<PW>(pending);
        else if (q == 0) FWN_WAIT_VMCNT(0);
        FWN_RING_STAMP(q, 1);
        if (FWN_ABL != 5) FWN_RING_BARRIER();
        FWN_RING_STAMP(q, 2);
        // The refill of the slot freed by this barrier (chunk q+D-1) is spread over the k-steps,
        // so every DMA issue (~100 cycles of this wave's issue time) hides under MFMAs in flight.
        const bool refill = FWN_ABL_DMA && q + D - 1 < nq;
        if (FWN_ABL == 1) {
            if (refill) issue(q + D - 1);
            continue;
        }
        const unsigned char* la = lds + (q % D) * SLOT;
        const unsigned char* lb = la + A_BYTES;
        // k-steps with explicitly double-buffered fragments: the ds_reads of step kk+1 are in
        // flight while the MFMAs of step kk issue.
        bf16x8 af[2][MI], bf_[2][2];
        auto ldfrag = [&](int ki, int s) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) af[s][mi] = *(const bf16x8*)(la + afr[ki] + mi * 32 * G::RB);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) bf_[s][ni] = *(const bf16x8*)(lb + bfr_[ki] + ni * 32 * G::RB);
        };
        constexpr int PPK = (PW + KSW - 1) / KSW;          // DMA pieces issued per k-step
        ldfrag(0, 0);
#pragma unroll
        for (int ki = 0; ki < KSW; ++ki) {
            if (ki + 1 < KSW) ldfrag(ki + 1, (ki + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);       // keep the next step's reads ahead of these MFMAs
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = mfma32(af[ki & 1][mi], bf_[ki & 1][ni], acc[mi][ni]);
                    if (mi == 0 && ni == 0 && refill) {
#pragma unroll
                        for (int j = ki * PPK; j < (ki + 1) * PPK && j < PW; ++j) issue_piece(ccn, q + D - 1, j);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        ccn = p.template chunk_ctx<BK>(q + D < nq ? q + D : 0);
            FWN_RING_STAMP(q, 3);
    }
    FWN_RING_STAMP_X(1);
    if constexpr (KSP > 1) {
        // sum the KSP partial accumulators: groups 1.. park theirs in LDS (the ring is drained)
        constexpr int TILE_F = MI * 2 * 16 * 64;           // floats per wave
        static_assert((KSP - 1) * WM * WN * TILE_F * 4 <= D * SLOT, "reduction scratch must fit the ring");
        // 16-byte LDS accesses, lane-contiguous: [wave][register quad][lane][4 floats] - the same adds in the same order as
        // the 4-byte form of rounds 2 - 4, which took 1.5 us of a 6 us launch (round 5: clock stamps of the one-launch flow,
        // profiles/r05_persist_stamps.txt; 96 ds_read_b32 + waits per summing wave against 24 ds_read_b128)
        float4* red = (float4*)lds;
        FWN_RING_BARRIER();      // the partial sums are parked in the ring's slots: the last fragment reads must have returned
        if (wk > 0) {
            float4* dst = red + ((wk - 1) * WM * WN + wmn) * (TILE_F / 4) + lane;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        dst[((mi * 2 + ni) * 4 + q) * 64] = make_float4(acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]);
        }
        __syncthreads();
        if (wk > 0) return;
#pragma unroll
        for (int g = 1; g < KSP; ++g) {
            const float4* src = red + ((g - 1) * WM * WN + wmn) * (TILE_F / 4) + lane;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 t = src[((mi * 2 + ni) * 4 + q) * 64];
                        acc[mi][ni][4 * q] += t.x; acc[mi][ni][4 * q + 1] += t.y; acc[mi][ni][4 * q + 2] += t.z; acc[mi][ni][4 * q + 3] += t.w;
                    }
        }
    }
    if (FWN_ABL >= 4) {     // ablation: keep the accumulators live without the epilogue
        float sacc = 0.0f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[mi][ni][r];
        if (sacc == 12345.678f) p.template epilogue<MI>(acc, m0 + wm * 32 * MI, n0 + wn * 64, lane);
        return;
    }
    FWN_RING_STAMP_X(2);
    if constexpr (PF) {
        p.template epilogue_pre<MI>(acc, m0 + wm * 32 * MI, n0 + wn * 64, lane, pre);
    } else if constexpr (prob_lds_epi<Prob>::value && KSP == 1 && NWV * 32 * MI * 64 * 4 <= D * SLOT) {
        // Row-major epilogue through LDS (problems that declare LDS_EPI): the accumulator layout gives a lane ONE column of
        // 16 rows, so a direct epilogue moves 2 bytes per lane and instruction (32 loads + 32 stores per lane for a
        // residual layer).  Each wave parks its (32 MI) x 64 fp32 tile in the drained ring and takes it back as rows of
        // 8 consecutive columns per lane: 16-byte loads / stores, 8 rows x 128 bytes per wave instruction.
        // (instantiations whose tiles would not fit the drained ring keep the direct epilogue)
        // rows_launch(): the same answer in every wave of the workgroup (the barrier below); rows_tile(): per wave tile -
        // a tile whose epilogue needs the accumulator layout (the gate derivative of fwn_gemm) keeps the direct form
        if (p.rows_launch()) {
            __syncthreads();             // every wave has read its last fragments (no DMA is in flight any more)
            if (p.rows_tile(n0 + wn * 64)) {
                float* wt = (float*)lds + wave * (32 * MI * 64);
                lds_epi_park<MI>(acc, wt, lane);
                p.template epilogue_rows<MI>(wt, m0 + wm * 32 * MI, n0 + wn * 64, lane);
            } else {
                p.template epilogue<MI>(acc, m0 + wm * 32 * MI, n0 + wn * 64, lane);
            }
        } else {
            p.template epilogue<MI>(acc, m0 + wm * 32 * MI, n0 + wn * 64, lane);
        }
    } else {
        p.template epilogue<MI>(acc, m0 + wm * 32 * MI, n0 + wn * 64, lane);
    }
    FWN_RING_STAMP_X(3);
}
