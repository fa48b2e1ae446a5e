// One flow of the small-M chain as ONE launch (round 5): front conv -> [gate -> res] x (L - 1) -> gate -> skip sum -> final
// conv -> ZeroConv + coupling, modules.py:161-186 / model.py:121-161, for the row counts where every stage is a launch of a
// handful of workgroups (~7 dependent launches of 5 - 9 us per flow).  The whole-model calls take it up
// to 512 rows by default (fwn_model_desc.persist_mode: blocks 4 - 7 of one clip, block 7 of the 8-clip pass; section 3.7 has the
// per-block table of where it wins); the form exists up to 4096 rows.
//
// Shape of the thing
//  * The flow is cut into TICKETS = (stage, 64-row tile i, 64-column tile j), numbered stage-major.  Workgroups (8 waves,
//    one per CU) take tickets from an atomic counter - one AHEAD of the one they work on - so the result never depends on how
//    many workgroups are resident or where they run: a ticket only ever waits for tickets with smaller numbers, and those
//    were taken by workgroups that are running (no spinning grid can starve another; any number of lanes may overlap).
//  * Dependencies are per row tile: done[stage][i] counts the column tiles of (stage, i) that have published.  A gate tile
//    waits for the previous stage's row tiles i - 1 .. i + 1 (dilation halo), the others for row tile i.  No grid barrier.
//  * EVERYTHING a ticket needs that no other workgroup of the launch writes is requested by LDS-DMA BEFORE the wait for its
//    dependencies: its weights (64 output columns x K <= 768: <= 96 KB), the gate's hoisted conditioning tile, the res bias,
//    the ZeroConv's plane tiles and tables.  Behind the wait only the activation rows are fetched (<= 64 KB, one round trip)
//    and the K loop runs out of LDS without a single wait or barrier.
//  * Hand-offs between workgroups follow the programming guide's Guideline 16, first row of its table: every handed-off
//    byte is stored write-through (16-byte sc1 stores from a row-major epilogue), every storing wave drains (vmcnt(0)), the
//    workgroup's barrier, ONE lane adds to the row tile's counter (agent scope); the consumer's wave 0 polls that word with
//    relaxed agent-scope loads, a workgroup barrier, then EVERY load of handed-off bytes is a 16-byte sc1 buffer load to
//    registers (the descriptor's bounds return the zero rows), written whole into the swizzled LDS image.  No fence anywhere.
//  * Arithmetic = the launch-per-stage path's, bit for bit: the same MFMA per k-step, the k-steps dealt to the same KSP
//    accumulation groups (k-step mod 4; mod 2 for the ZeroConv; one group for the front conv), partial sums added in the
//    same order, the same epilogue expressions.  tests/test_gpu_parity.py compares the two paths with ==.
//  * Every spin is bounded (2 s of the 100 MHz reference clock; fwn_set_option("persist_spin_us")): a give-up sets sync[1],
//    the ticket goes on with whatever is there, and the ZeroConv tickets down the chain turn their outputs into NaN - the
//    call's log-p / log-det / waveform are NaN, never silently wrong (fwn_flow_persist_status reads the word).
//
// sync block (caller zeroes it before the launch; the whole-model calls zero all of a pass's blocks with one memset):
//   sync[0] ticket counter, sync[1] give-up code, sync[8 + s * RT + i] = done[s][i].
#pragma once
#include "gemm_ring.h"
#include <string.h>
#include <type_traits>
#include "../../include/fwn.h"

#define FWN_PS_MAXL 2
#ifndef FWN_PERSIST_MAX_ROWS
#define FWN_PERSIST_MAX_ROWS 4096
#endif
#define FWN_PS_HDR 8

struct PersistArgs {
    float* xa; float* xb; const float* an;
    const bf16* W2; const float* bfront;
    const bf16* Wd[FWN_PS_MAXL]; const float* bgate[FWN_PS_MAXL];
    const bf16* Wres[FWN_PS_MAXL]; const float* bres[FWN_PS_MAXL];
    const bf16* Wskip; const float* bskip;
    const bf16* Wfinal; const float* bfinal;
    const bf16* Wzero; const float* bzero; const float* ezero;
    bf16* hA;             // [M][256]: the front conv's output (h0)
    bf16* hB;             // [M][256]: the other h buffer
    bf16* o;              // [L][M][256]
    const float* P;       // [L][M][512] hoisted conditioning projections (packed-N columns)
    float* partial;       // forward: log-det partial slots ((M + 63) / 64 * 8), inverse: nullptr
    unsigned* sync;
    int M, Ti, Ch, npt, L, inverse, has_front;
    unsigned spin_ticks;  // bound of every spin in ticks of the 100 MHz reference clock (fwn_set_option("persist_spin_us"); default 2 s)
#ifdef FWN_PS_STAMP
    unsigned long long* stamps;   // diagnostic build: [ticket][8] s_memrealtime stamps of wave 0
#endif
};

typedef __attribute__((address_space(1))) unsigned ps_gu32;
__device__ __forceinline__ unsigned ps_ld(const unsigned* p) {
    return __hip_atomic_load((const ps_gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned ps_add(unsigned* p, unsigned v) {
    return __hip_atomic_fetch_add((ps_gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wave 0 only: until *p >= need (relaxed agent-scope polls + s_sleep); a give-up after 2 s sets *err and returns 1 (the
// caller hands that to its own workgroup through LDS: the global store is not drained at the barrier that follows, so the
// other waves' reload of *err need not see it - ADVICE r5)
__device__ __forceinline__ unsigned ps_wait_ge(const unsigned* p, unsigned need, unsigned* err, unsigned code, unsigned spin_ticks) {
    if (ps_ld(p) >= need) return 0u;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        __builtin_amdgcn_s_sleep(1);
        if (ps_ld(p) >= need) return 0u;
        if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)spin_ticks) {
            __hip_atomic_store((ps_gu32*)err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return 1u;
        }
    }
}

// every wave's LDS and vector-memory operations retired, then the workgroup's barrier
#define PS_BARRIER() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

// LDS: [image(s) | weight sub-chunks]; the split-K reduction scratch (48 KB) and the epilogue park (16 KB) reuse the front
// of it once the K loop is over.  Largest stage: a gate = 71 rows x 512 B + 64 columns x 768 x 2 B.
#define FWN_PS_IMG_MAX (72 * 512)                  // 70 halo rows + the zero row, 1 KiB aligned
#define FWN_PS_EPI (FWN_PS_IMG_MAX + 64 * 768 * 2)   // the epilogue's operands by LDS-DMA (gate: the P tile, res: the bias, ZeroConv: plane tiles + tables)
#define FWN_PS_LDS (FWN_PS_EPI + 18432 + 256)
#define FWN_PS_PARK (48 * 1024)

// weights: rows [n0, n0 + 64) x columns [0, 64 nsub) of W[..][ldb] -> nsub sub-chunks of [64][64] bf16 in the ring's
// swizzled image (RingGeom<64>), 8 one-KiB pieces each, dealt over the 8 waves
__device__ __forceinline__ void ps_issue_weights(unsigned char* bw, const bf16* W, int nrows_total, int ldb, int n0, int nsub,
                                                 int wave, int lane) {
    const srd_t sw = make_srd(W, (uint32_t)((size_t)nrows_total * ldb * 2));
    const int npieces = nsub * 8;
    for (int pid = wave; pid < npieces; pid += 8) {
        const int sub = pid >> 3, r = (pid & 7) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        buf_load16_lds(sw, (uint32_t)((n0 + r) * ldb + sub * 64 + c * 8) * 2u, bw + sub * 8192 + (pid & 7) * 1024);
    }
}

// Rows g0 + (tid >> 5) + 16 j (j < NP), 16-byte piece tid & 31, of a row-major [M][256] bf16 matrix that other workgroups of
// this launch wrote: 16-byte sc1 buffer loads to registers - rows outside [0, M) read as zeros by the buffer's bounds (a negative
// row is a huge unsigned offset) - and from there into the swizzled LDS image (lds_off256: the XOR term is the same for
// every j, so both sides are one per-lane base plus j * 8 KiB).  Behind the wait for the producers this is the critical
// path of the level: one add per load, no selects.
template <int NP, int OFF, int NV>
__device__ __forceinline__ void ps_rows_issue(u32x4 (&v)[NV], const srd_t& src, int g0, int tid) {
    const uint32_t voff = (uint32_t)((g0 + (tid >> 5)) * 512 + (tid & 31) * 16);
#pragma unroll
    for (int j = 0; j < NP; ++j) v[OFF + j] = __builtin_amdgcn_raw_buffer_load_b128(src, voff + (uint32_t)(j * 8192), 0, 16);   // aux 16 = sc1
}
template <int NP, int OFF, int NV>
__device__ __forceinline__ void ps_rows_write(const u32x4 (&v)[NV], unsigned char* img, int nrows, int tid) {
    unsigned char* const p = img + lds_off256(tid >> 5, tid & 31);
#pragma unroll
    for (int j = 0; j < NP; ++j)
        if (j * 16 + 16 <= nrows || (tid >> 5) + j * 16 < nrows) *(u32x4*)(p + j * 8192) = v[OFF + j];
}

// 8 consecutive columns c8 * 8 .. + 7 of row `row` (0 .. 63) of the parked 64 x 64 fp32 tile (two wave tiles in
// lds_epi_park's layout)
__device__ __forceinline__ void ps_take8(const float* park, int row, int g0, int g1, float (&v)[8]) {
    const float* wt = park + (row >> 5) * 2048 + (row & 31) * 64;
    const int f = (row >> 1) & 1;
    const float4 a = *(const float4*)(wt + ((g0 ^ f) << 2)), b = *(const float4*)(wt + ((g1 ^ f) << 2));
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// Diagnostic build (-DFWN_PS_STAMP, tools/probe/persist_stamps.py): wave 0 stamps the 100 MHz reference clock at eight points
// of every ticket into the tail of the sync block: [ticket][8] = start | weights issued | producers done | rows in LDS |
// K loop + reduction done | stores issued | stores drained | (stage, row tile, column tile, workgroup, XCC id)
#ifdef FWN_PS_STAMP
#define PS_STAMP(k) do { if (wave == 0 && lane == 0) stamp_base[(size_t)t_cur * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PS_STAMP(k) do { } while (0)
#endif

enum { PS_FRONT = 0, PS_GATE = 1, PS_RES = 2, PS_SKIP = 3, PS_FINAL = 4, PS_ZERO = 5 };

// K loop of one wave out of LDS.  Segment sg reads this lane's image row r = so<sg> (rows of 512 bytes, 16-byte piece c at
// ((c & 15) ^ (r & 15)) * 16 + (c >> 4) * 256: lds_off256); bq = this lane's weight row in the first 64-wide sub-chunk.
// The launch-per-stage kernels deal the k-steps over KSP wave groups (k-step mod KSP) and add the groups' partial sums in
// ascending order through LDS.  Here ONE wave takes every k-step of its output quadrant and keeps the NACC = KSP partial sums
// in accumulators of its own (k-step mod NACC): the same products into the same sums, the groups added in the same order -
// without the two barriers and the LDS round trip (1.5 us of a 6 us ticket), and the chains are independent, so the MFMAs
// issue back to back.
// The k-step count NSEG * KPS is a template parameter and the loop is unrolled in full: every fragment address is then a
// per-lane base (one of 8 per segment for A, one of 4 for B: the XOR half of the swizzle,
// (2 kc + lh) ^ (row & 15) = 2 (kc ^ (row >> 1 & 7)) + (lh ^ (row & 1))) plus an IMMEDIATE offset, so the loop body is one or
// two ds_reads and one MFMA per k-step with no address arithmetic (computed per k-step, the arithmetic of ONE wave per SIMD
// set the pace: 105 cycles per 32-cycle MFMA), and the reads run PS_D k-steps ahead of the MFMAs in a ring of fragment
// registers; hipcc counts the lgkmcnt waits of straight-line code exactly.
// TWO: both column halves of the row half (two waves: the ZeroConv, whose epilogue pairs columns n and n + 32 in a lane).
#ifndef PS_D
#define PS_D 3
#endif
template <int NACC, bool TWO, int NSEG, int KPS>
__device__ __forceinline__ void ps_kloop(f32x16 (&tot)[2], float c0, float c1, const unsigned char* bq, const unsigned char* img,
                                         int so0, int so1, int so2, int lr, int lh) {
    constexpr int NH = TWO ? 2 : 1, NK = NSEG * KPS;
    f32x16 acc[NACC][NH];
#pragma unroll
    for (int g = 0; g < NACC; ++g)
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][h][r] = g == 0 ? (h == 0 ? c0 : c1) : 0.0f;
    const int bsw = (lr >> 1) & 7;
    const unsigned ab0 = (unsigned)(so0 * 512 + ((lh ^ (so0 & 1)) << 4) + (((so0 >> 1) & 7) << 5));
    const unsigned ab1 = (unsigned)(so1 * 512 + ((lh ^ (so1 & 1)) << 4) + (((so1 >> 1) & 7) << 5));
    const unsigned ab2 = (unsigned)(so2 * 512 + ((lh ^ (so2 & 1)) << 4) + (((so2 >> 1) & 7) << 5));
    const unsigned bb = (unsigned)(((lh ^ (bsw & 1)) << 4) + ((bsw >> 1) << 5));
    bf16x8 fa[PS_D], fb[PS_D], fc[PS_D];
#define PS_RD(KS, SLOT)                                                                                                \
    do {                                                                                                               \
        const int ks_ = (KS), sg_ = ks_ / KPS, kc_ = ks_ % KPS;                                                        \
        const unsigned ab_ = sg_ == 0 ? ab0 : sg_ == 1 ? ab1 : ab2;                                                    \
        fa[SLOT] = *(const bf16x8*)(img + (ab_ ^ (unsigned)((kc_ & 7) << 5)) + ((kc_ >> 3) << 8));                     \
        const unsigned char* bs_ = bq + (bb ^ (unsigned)((ks_ & 3) << 5)) + (ks_ >> 2) * 8192;                         \
        fb[SLOT] = *(const bf16x8*)bs_;                                                                                \
        if constexpr (TWO) fc[SLOT] = *(const bf16x8*)(bs_ + 32 * 128);                                                \
    } while (0)
#pragma unroll
    for (int ks = 0; ks < (PS_D < NK ? PS_D : NK); ++ks) PS_RD(ks, ks);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        __builtin_amdgcn_sched_barrier(0);
        acc[ks % NACC][0] = mfma32(fa[ks % PS_D], fb[ks % PS_D], acc[ks % NACC][0]);
        if constexpr (TWO) acc[ks % NACC][NH - 1] = mfma32(fa[ks % PS_D], fc[ks % PS_D], acc[ks % NACC][NH - 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + PS_D < NK) PS_RD(ks + PS_D, ks % PS_D);
    }
#undef PS_RD
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        tot[h] = acc[0][h];
#pragma unroll
        for (int g = 1; g < NACC; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[h][r] += acc[g][h][r];
    }
}

// ONE kernel for every shape: the stages differ in parameters (weights, K segments, accumulation groups) and in two short
// type switches (how the activation rows get into LDS; the epilogue).  a.has_front: the front conv is stage 0 (Ch >= 16:
// front_mfma_kernel's arithmetic - the a-plane as a (hi | lo) bf16 image, one accumulation group, quadrants of 32 x 32);
// otherwise the caller has launched it (Ch <= 8: the fp32 VALU kernel).
__global__ __launch_bounds__(512) void flow_persist_kernel(PersistArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[FWN_PS_LDS];
    unsigned* const misc = (unsigned*)(lds + FWN_PS_LDS - 256);
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int wm = wave & 1;
    const int M = a.M, Ti = a.Ti, L = a.L;
    const int RT = (M + 63) >> 6;
    const int FRONT = a.has_front ? 1 : 0;
    const int CHP = a.Ch, CHE = a.Ch < 32 ? 32 : a.Ch;        // channels of the plane / of the (zero padded) front image

    // ---- stage table: [front] gate0 [res0 gate1 ..] skip final zero; tickets are numbered stage-major ----
    const int nst = FRONT + 2 * L - 1 + 3;
    auto st_type = [&](int s) -> int {
        if (FRONT) { if (s == 0) return PS_FRONT; s -= 1; }
        if (s < 2 * L - 1) return (s & 1) ? PS_RES : PS_GATE;
        return PS_SKIP + (s - (2 * L - 1));
    };
    auto st_ntn = [&](int s) -> int {
        const int ty = st_type(s);
        return ty == PS_GATE ? 8 : ty == PS_ZERO ? a.npt : 4;
    };
    int total = 0;
    for (int s = 0; s < nst; ++s) total += RT * st_ntn(s);
    unsigned* const done = a.sync + FWN_PS_HDR;
    unsigned* const err = a.sync + 1;
    float* const parkw = (float*)(lds + FWN_PS_PARK);
#ifdef FWN_PS_STAMP
    unsigned long long* const stamp_base = (unsigned long long*)(a.sync + ((FWN_PS_HDR + (2 * L + 3) * RT + 3) & ~3));
#endif

    // ---- the first ticket (the next one is taken while this one is worked on) ----
    if (wave == 0) {
        unsigned t = 0;
        if (tid0 == 0) t = ps_add(a.sync, 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        if (tid0 == 0) misc[0] = t;
    }
    PS_BARRIER();
    unsigned t_cur = misc[0];
    int it = 0;
    while ((int)t_cur < total) {
        // (lane-constant address terms: hipcc hoists them out of this loop, i.e. in front of the first wait for producers -
        // deliberately left so: made opaque per ticket they are recomputed BEHIND the wait, on the critical path: +3 % per pass)
        const int tid = tid0;
        const int lane = tid & 63, lr = lane & 31, lh = lane >> 5;
        // decode (wave-uniform)
        int s = 0, rem = (int)t_cur;
        for (;; ++s) {
            const int n = RT * st_ntn(s);
            if (rem < n) break;
            rem -= n;
        }
        const int ty = st_type(s), ntn = st_ntn(s);
        const int ly = (s - FRONT) >> 1;                       // layer of a gate / res stage
        const int ti = rem / ntn, tj = rem % ntn;
        const int m0 = ti * 64, n0 = tj * 64;
        const int sp = s - 1;                                  // the stage whose output this one reads
        const int rl = wm * 32 + lr;                           // this lane's row of the tile (MFMA A operand)
        const int dil = ly == 0 ? 1 : 3;                       // kernel_size ** layer (modules.py:152), L <= 2
        // which h buffer holds what: layer l reads hc, res_l writes hn; the last gate's input buffer takes U, the other S
        bf16* const hc = (ly & 1) ? a.hB : a.hA;
        bf16* const hn = (ly & 1) ? a.hA : a.hB;
        bf16* const hlast = ((L - 1) & 1) ? a.hB : a.hA;
        bf16* const hfree = ((L - 1) & 1) ? a.hA : a.hB;
        unsigned pending = 0;
        if (wave == 0 && lane == 0) pending = ps_add(a.sync, 1u);       // the next ticket: its number arrives under this one
        PS_STAMP(0);
#ifdef FWN_PS_STAMP_CLK
        if (wave == 0 && lane == 0) stamp_base[(size_t)t_cur * 8 + 1] = __builtin_amdgcn_s_memtime();
#endif
#ifdef FWN_PS_STAMP
        if (wave == 0 && lane == 0)
            stamp_base[(size_t)t_cur * 8 + 7] = (unsigned long long)s | ((unsigned long long)ti << 8) | ((unsigned long long)tj << 24) |
                                                ((unsigned long long)blockIdx.x << 32) | ((unsigned long long)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15) << 48);
#endif

        // ---- 1. the stage's parameters ----
        // weights [wrows][ldb], nsub 64-wide K sub-chunks at LDS offset woff; nseg K segments of kps k-steps; ksp accumulation
        // groups (gemm_ring_body's KSP of the launch-per-stage path: k-step mod ksp; 1 = quadrants of one group - launch_ring's
        // choice for the gate (N = 512) from 192 workgroups of 64 x 128 on, i.e. 3072 rows)
        const bf16* Wt; const float* bias_p; int wrows, ldb, nsub, woff, ksp, nseg, kps;
        if (ty == PS_FRONT)      { Wt = a.W2;       bias_p = a.bfront;    wrows = 256; ldb = 6 * CHE; nsub = 6 * CHE / 64; woff = FWN_PS_IMG_MAX; ksp = 1; nseg = 3; kps = CHE / 8; }
        else if (ty == PS_GATE)  { Wt = a.Wd[ly];   bias_p = a.bgate[ly]; wrows = 512; ldb = 768; nsub = 12; woff = FWN_PS_IMG_MAX; ksp = RT * 4 >= 192 ? 1 : 4; nseg = 3; kps = 16; }
        else if (ty == PS_RES)   { Wt = a.Wres[ly]; bias_p = nullptr;     wrows = 256; ldb = 256; nsub = 4; woff = 64 * 512; ksp = 4; nseg = 1; kps = 16; }
        else if (ty == PS_SKIP)  { Wt = a.Wskip;    bias_p = a.bskip;     wrows = 256; ldb = L * 256; nsub = L * 4; woff = L * 64 * 512; ksp = 4; nseg = L; kps = 16; }
        else if (ty == PS_FINAL) { Wt = a.Wfinal;   bias_p = a.bfinal;    wrows = 256; ldb = 256; nsub = 4; woff = 64 * 512; ksp = 4; nseg = 1; kps = 16; }
        else                     { Wt = a.Wzero;    bias_p = a.bzero;     wrows = a.npt * 64; ldb = 256; nsub = 4; woff = 64 * 512; ksp = 2; nseg = 1; kps = 16; }
        unsigned char* const bw = lds + woff;
        // wave roles in the K loop: waves 0 - 3 own the quadrants (row half wm, column half nh) of the 64 x 64 tile; the
        // ZeroConv: waves 0, 1 own a row half each with both column halves (its epilogue pairs columns n and n + 32 in a lane)
        const bool two = ty == PS_ZERO;
        const int nh = two ? 0 : (wave >> 1) & 1;
        const bool active = two ? wave < 2 : wave < 4;

        // ---- 2. everything that does not depend on other workgroups: the weights (LDS-DMA), the epilogue's operands ----
        ps_issue_weights(bw, Wt, wrows, ldb, n0, nsub, wave, lane);
        // the epilogue's operands that no other workgroup of this launch writes, requested BEFORE the wait for the producers -
        // by LDS-DMA like the weights, so that they cost no registers across the K loop and no round trip behind it: gate: the
        // hoisted conditioning projection of the tile (64 rows x 64 packed-N columns fp32 = 16 one-KiB pieces of 4 rows; an
        // earlier launch wrote it; rows past M read as zeros and are never stored); res: the bias of the 64 columns
        unsigned char* const epi = lds + FWN_PS_EPI;
        if (ty == PS_GATE) {
            const srd_t sp = make_srd(a.P + (size_t)ly * M * 512, (uint32_t)((size_t)M * 2048));
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int pid = wave + q * 8;
                buf_load16_lds(sp, (uint32_t)((m0 + pid * 4 + (lane >> 4)) * 2048 + n0 * 4 + (lane & 15) * 16), epi + pid * 1024);
            }
        } else if (ty == PS_RES && wave == 0) {
            if (lane < 16) buf_load16_lds(make_srd(a.bres[ly], 1024u), (uint32_t)(n0 * 4 + lane * 16), epi);
        }
        // ZeroConv: the plane elements its coupling transforms (only this ticket touches them) and its tables
        TailZeroProb zp{hlast, a.Wzero, a.bzero, a.ezero, a.an, a.xa, a.xb, a.partial, M, a.Ch, a.npt, a.inverse, nullptr};
        static_assert(TailZeroProb::LDS_BYTES <= 18432, "epilogue operand region");
        if (ty == PS_ZERO) zp.stage_lds(epi, m0, n0, wave, lane);
#ifndef FWN_PS_STAMP_CLK
        PS_STAMP(1);
#endif

        // ---- 3. the tickets this one reads from; the barrier behind which every wave may load handed-off bytes ----
        if (wave == 0) {
            if (lane == 0) misc[2 + (it & 1)] = pending;
            unsigned gave = 0u;
            if (ty == PS_GATE) {
                if (sp >= 0) {
                    const unsigned need = (unsigned)st_ntn(sp);
                    for (int d = -1; d <= 1; ++d)
                        if ((unsigned)(ti + d) < (unsigned)RT) gave |= ps_wait_ge(done + sp * RT + ti + d, need, err, 1u + s, a.spin_ticks);
                }
            } else if (ty != PS_FRONT) {
                gave |= ps_wait_ge(done + sp * RT + ti, (unsigned)st_ntn(sp), err, 1u + s, a.spin_ticks);
                if (ty == PS_FINAL)          // U overwrites the h buffer the last gate reads: its neighbours' halo reads first
                    for (int d = -1; d <= 1; d += 2)
                        if ((unsigned)(ti + d) < (unsigned)RT) gave |= ps_wait_ge(done + (sp - 1) * RT + ti + d, 8u, err, 1u + s, a.spin_ticks);
            }
            if (lane == 0) misc[4] = gave;       // this ticket's own give-up, through LDS (ordered by the barrier below)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        PS_STAMP(2);

        // (ZeroConv: the give-up word, requested here so that its round trip is over by the epilogue)
        unsigned errw = 0u;
        if (ty == PS_ZERO) errw = ps_ld(err) | misc[4];
        // ---- 4. the activation rows -> LDS image(s) ----
        u32x4 hin = {0u, 0u, 0u, 0u};
        const uint32_t hbytes = (uint32_t)((size_t)M * FWN_HID * 2);
        int zrow = 0;                                          // image row that reads as zeros (gate / front: taps outside the clip)
        if (ty == PS_FRONT) {
            // rows m0 - 1 .. m0 + 64 of the a-plane (an earlier launch wrote it: plain loads) as (hi | lo) bf16, ActNorm applied;
            // image rows of 512 bytes whatever CHE; the padded channels of a 16-channel plane are zeros
            constexpr int NR = 66;
            zrow = NR;
            const int cq = CHP >> 2;                           // float4 groups per plane row
            const int ntask = (NR + 1) * cq;
            const int apply_an = a.inverse ? 0 : 1;
            const int tau_t = (tid % cq) * 4;                  // 512 % cq == 0: the same channel group for all tasks of a thread
            const float4 an_sh = apply_an ? *(const float4*)(a.an + tau_t) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            const float4 an_sc = apply_an ? *(const float4*)(a.an + CHP + tau_t) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);
            if (CHP < CHE)
                for (int i = tid; i < (NR + 1) * 4; i += 512) {          // CHP = 16, CHE = 32: pieces 2, 3 (hi) and 6, 7 (lo)
                    const int j = i >> 2, q = i & 3;
                    *(uint4*)(lds + lds_off256(j, (q >> 1) * 4 + 2 + (q & 1))) = make_uint4(0u, 0u, 0u, 0u);
                }
            float4 vin[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int task = tid + q * 512;
                const int j = task / cq, g = m0 - 1 + j;
                const bool ok = task < ntask && j < NR && (unsigned)g < (unsigned)M;
                vin[q] = *(const float4*)(a.xa + (size_t)(ok ? g : 0) * CHP + (ok ? tau_t : 0));
            }
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int task = tid + q * 512;
                if (task < ntask) {
                    const int j = task / cq, g = m0 - 1 + j;
                    const bool ok = j < NR && (unsigned)g < (unsigned)M;
                    const float4 v = vin[q];
                    const float f[4] = {v.x, v.y, v.z, v.w}, sh[4] = {an_sh.x, an_sh.y, an_sh.z, an_sh.w}, sc[4] = {an_sc.x, an_sc.y, an_sc.z, an_sc.w};
                    union { bf16 e[4]; uint2 u; } hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float y = apply_an ? (f[e] + sh[e]) * sc[e] : f[e];
                        y = ok ? y : 0.0f;
                        hi.e[e] = (bf16)y;
                        lo.e[e] = (bf16)(y - (float)hi.e[e]);
                    }
                    *(uint2*)(lds + lds_off256(j, tau_t >> 3) + (tau_t & 7) * 2) = hi.u;
                    *(uint2*)(lds + lds_off256(j, (CHE + tau_t) >> 3) + (tau_t & 7) * 2) = lo.u;
                }
            }
        } else if (ty == PS_GATE) {
            const int R = 64 + 2 * dil;
            zrow = R;
            u32x4 v[5];
            ps_rows_issue<5, 0>(v, make_srd(hc, hbytes), m0 - dil, tid);
            ps_rows_write<5, 0>(v, lds, R, tid);
            if (tid < 32) { u32x4 z = {0u, 0u, 0u, 0u}; *(u32x4*)(lds + lds_off256(R, tid)) = z; }
        } else if (ty == PS_SKIP && L == 2) {
            u32x4 v[8];
            ps_rows_issue<4, 0>(v, make_srd(a.o, hbytes), m0, tid);
            ps_rows_issue<4, 4>(v, make_srd(a.o + (size_t)M * FWN_HID, hbytes), m0, tid);
            ps_rows_write<4, 0>(v, lds, 64, tid);
            ps_rows_write<4, 4>(v, lds + 64 * 512, 64, tid);
        } else {
            // skip (L = 1) / res / final / zero: one image of the tile's rows; res: + the residual rows of this lane's
            // epilogue item (row tid >> 3, 8 columns) - handed-off bytes as well
            const bf16* src = ty == PS_RES ? a.o + (size_t)ly * M * FWN_HID : ty == PS_SKIP ? a.o : ty == PS_FINAL ? hfree : hlast;
            u32x4 v[4];
            ps_rows_issue<4, 0>(v, make_srd(src, hbytes), m0, tid);
            hin = __builtin_amdgcn_raw_buffer_load_b128(make_srd(ty == PS_RES ? (const bf16*)hc : src, hbytes),
                                                        (uint32_t)((m0 + (tid >> 3)) * 512 + n0 * 2 + (tid & 7) * 16), 0, 16);
            ps_rows_write<4, 0>(v, lds, 64, tid);
        }
        PS_BARRIER();
        PS_STAMP(3);

        // ---- 5. the K loop out of LDS: no wait, no barrier ----
        f32x16 acc[1][2];
        {
            // segment s: this lane's row at byte address segp[s] of a [rows][512 B] image, 16-byte piece c of a row at
            // ((c & 15) ^ (row & 15)) * 16 + (c >> 4) * 256 (lds_off256)
            int so0 = rl, so1 = 64 + rl, so2 = rl;             // image ROW of this lane per segment (skip: the second layer's image)
            if (ty == PS_GATE || ty == PS_FRONT) {
                const int dd = ty == PS_GATE ? dil : 1;
                const int r = m0 + rl, t = r % Ti;
                const bool in = r < M;
                so0 = (in && (unsigned)(t - dd) < (unsigned)Ti) ? rl : zrow;
                so1 = in ? rl + dd : zrow;
                so2 = (in && (unsigned)(t + dd) < (unsigned)Ti) ? rl + 2 * dd : zrow;
            }
            const float c0 = bias_p ? bias_p[n0 + nh * 32 + lr] : 0.0f;
            const float c1 = (bias_p && two) ? bias_p[n0 + 32 + lr] : 0.0f;
            const unsigned char* const bq = bw + (nh * 32 + lr) * 128;
            if (active) {
                if (ty == PS_ZERO) ps_kloop<2, true, 1, 16>(acc[0], c0, c1, bq, lds, so0, so1, so2, lr, lh);
                else if (ty == PS_GATE) {
                    if (ksp == 4) ps_kloop<4, false, 3, 16>(acc[0], c0, c1, bq, lds, so0, so1, so2, lr, lh);
                    else ps_kloop<1, false, 3, 16>(acc[0], c0, c1, bq, lds, so0, so1, so2, lr, lh);
                } else if (ty == PS_FRONT) {
                    if (kps == 16) ps_kloop<1, false, 3, 16>(acc[0], c0, c1, bq, lds, so0, so1, so2, lr, lh);
                    else if (kps == 8) ps_kloop<1, false, 3, 8>(acc[0], c0, c1, bq, lds, so0, so1, so2, lr, lh);
                    else ps_kloop<1, false, 3, 4>(acc[0], c0, c1, bq, lds, so0, so1, so2, lr, lh);
                } else if (nseg == 2) ps_kloop<4, false, 2, 16>(acc[0], c0, c1, bq, lds, so0, so1, so2, lr, lh);
                else ps_kloop<4, false, 1, 16>(acc[0], c0, c1, bq, lds, so0, so1, so2, lr, lh);
            }
        }
        PS_STAMP(4);
        FWN_RING_BARRIER();                                    // the park overlays image / weights: every fragment read has returned
        // ---- 7. epilogue ----
        if (ty == PS_ZERO) {
            // ZeroConv + coupling + both ActNorms + log-det partial: TailZeroProb's epilogue expressions on operands staged in LDS
            // (plain stores: only this ticket touches these plane elements, and the NEXT launch reads them)
            // A spin that gave up anywhere up the chain of this row tile (sync[1] != 0: the stores of the ticket that set it were
            // drained before it published) must not pass silently: this ticket's plane elements and log-det partial become NaN,
            // so log_p / log-det / the waveform of the call are NaN (and fwn_flow_persist_status reports the word).
            if (wave < 2) zp.epilogue_lds(acc, m0 + wm * 32, n0, lane, epi, errw != 0u);
        } else {
            // the 64 x 64 fp32 tile -> LDS (lds_epi_park's layout, two 32-row wave tiles), then rows of 8 columns per lane:
            // ONE 16-byte write-through store per lane
            if (wave < 4) {
                float* wt = parkw + wm * 2048;
#pragma unroll
                for (int r2 = 0; r2 < 16; ++r2) {
                    const int row = acc_row_c(r2) + 4 * lh;
                    wt[row * 64 + ((nh * 32 + lr) ^ (((r2 >> 1) & 1) << 2))] = acc[0][0][r2];
                }
            }
            FWN_RING_BARRIER();
#ifdef FWN_PS_STAMP_EPI
            PS_STAMP(2);
#endif
            const srd_t so_h = make_srd(ty == PS_FRONT ? a.hA : ty == PS_RES ? hn : ty == PS_SKIP ? hfree : ty == PS_FINAL ? hlast : a.o + (size_t)ly * M * FWN_HID,
                                        (uint32_t)((size_t)M * FWN_HID * 2));
            u32x4 outw = {0u, 0u, 0u, 0u};
            uint32_t voff = FWN_OOB;
            if (ty == PS_GATE) {
                if (tid < 256) {
                    const int row = tid >> 2, cg = tid & 3;
                    float fv[8], gv[8];
                    ps_take8(parkw, row, 2 * cg, 2 * cg + 1, fv);
                    ps_take8(parkw, row, 8 + 2 * cg, 8 + 2 * cg + 1, gv);
                    const float* pl = (const float*)epi + row * 64 + cg * 8;
                    const float4 pf0 = *(const float4*)pl, pf1 = *(const float4*)(pl + 4);
                    const float4 pg0 = *(const float4*)(pl + 32), pg1 = *(const float4*)(pl + 36);
                    const float pf[8] = {pf0.x, pf0.y, pf0.z, pf0.w, pf1.x, pf1.y, pf1.z, pf1.w};
                    const float pg[8] = {pg0.x, pg0.y, pg0.z, pg0.w, pg1.x, pg1.y, pg1.z, pg1.w};
                    Pack16 out;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f32x2 y = gated_unit2(f32x2{fv[e], fv[e + 1]} + f32x2{pf[e], pf[e + 1]},
                                                    f32x2{gv[e], gv[e + 1]} + f32x2{pg[e], pg[e + 1]});
                        out.e[e] = (bf16)y.x; out.e[e + 1] = (bf16)y.y;
                    }
                    outw = out.w;
                    const int ch = (n0 >> 7) * 64 + ((n0 >> 6) & 1) * 32 + cg * 8;
                    if (m0 + row < M) voff = (uint32_t)((m0 + row) * FWN_HID + ch) * 2u;
                }
            } else {
                const int row = tid >> 3, c8 = tid & 7;
                float v[8];
                if (ty == PS_SKIP || ty == PS_FINAL) {
                    // output column o holds GEMM column swap_bits23(o) (packing.acc_k_perm: S and U are natural-order matrices)
                    const int a4 = (c8 >> 1) * 4, be = c8 & 1;
                    ps_take8(parkw, row, a4 + be, a4 + 2 + be, v);
                } else {
                    ps_take8(parkw, row, 2 * c8, 2 * c8 + 1, v);
                }
                Pack16 out;
                if (ty == PS_RES) {
                    const float4 pf0 = *(const float4*)((const float*)epi + c8 * 8), pf1 = *(const float4*)((const float*)epi + c8 * 8 + 4);
                    const float bb[8] = {pf0.x, pf0.y, pf0.z, pf0.w, pf1.x, pf1.y, pf1.z, pf1.w};
                    Pack16 hv;
                    hv.w = hin;
#pragma unroll
                    for (int e = 0; e < 8; ++e) out.e[e] = (bf16)(((float)hv.e[e] + v[e] + bb[e]) * 0.70710678118654752f);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) out.e[e] = (bf16)fmaxf(v[e], 0.0f);
                }
                outw = out.w;
                if (m0 + row < M) voff = (uint32_t)((m0 + row) * FWN_HID + n0 + c8 * 8) * 2u;
            }
            __builtin_amdgcn_raw_buffer_store_b128(outw, so_h, voff, 0, 16);      // aux 16 = sc1: write-through
        }
        // ---- 8. publish: every storing wave drained, the workgroup's barrier, ONE agent-scope add ----
#ifndef FWN_PS_STAMP_CLK
        PS_STAMP(5);
#endif
        PS_BARRIER();
        PS_STAMP(6);
#ifdef FWN_PS_STAMP_CLK
        if (wave == 0 && lane == 0) stamp_base[(size_t)t_cur * 8 + 5] = __builtin_amdgcn_s_memtime();
#endif
        if (wave == 0 && lane == 0 && ty != PS_ZERO) ps_add(done + s * RT + ti, 1u);
        t_cur = misc[2 + (it & 1)];
        ++it;
    }
}
// (host side - option slot, sync-word count, the shape rule and the launchers - in flow_persist.hip: ADVICE r5, a header that
// defined external-linkage functions and a global could be included once only)
