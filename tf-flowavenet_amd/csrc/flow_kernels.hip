// One flow step of the FloWaveNet coupling stack on gfx950:
//   front conv -> [gated dilated layer (+cond) -> res 1x1] x (L-1) -> gated layer -> tail
// The tail kernel fuses skip-sum GEMM, final 1x1, ZeroConv1d, the affine coupling,
// ActNorm and the log-det reduction.  Reference arithmetic: modules.py:110-128 (ResBlock),
// :161-186 (WaveNet), model.py:86-102 (ActNorm), :121-161 (AffineCoupling).
//
// Data layout: flow state = two fp32 planes [M][Ch] (even / odd samples),
// hidden activations bf16 [M][256], conditioning = two bf16 planes [M][cin].
#include "common.h"
#include "gemm_ring.h"
#include "fwn_internal.h"

// ---------------------------------------------------------------------------
// Problem descriptors for gemm128_body
// ---------------------------------------------------------------------------
struct RowCtxT {
    int row;   // flattened row index b*Ti + t (may be >= M in the last tile)
    int t;     // position inside the clip (row % Ti), for zero padding at clip edges
};

// ---- front conv: h0 = ReLU(conv_k3(actnorm(x_a)) + bias), modules.py:164-165 -------------
// The fp32 flow state enters the bf16 MFMA as a hi/lo pair (x = hi + lo, both bf16) so the
// network input carries ~16 mantissa bits: K is traversed twice, once per half, against the
// same weights.  K index inside a half: k = tap*Ch + tau.
struct FrontProb {
    static constexpr bool A_DMA = false;
    const float* xa;      // plane [M][Ch] fp32
    const float* an;      // [4][Ch] (shift, scale, iscale, logs3) of the a-plane, natural order
    const bf16* W;        // [256][kpad]
    const float* bias;    // [256]
    bf16* hout;           // [M][256]
    int M, Ti, Ch, chlog, kpad, apply_an;
    unsigned char* h8out = nullptr;   // [M][256] e4m3 copy for the fp8 gate (front_valu_kernel only)
    typedef RowCtxT RowCtx;
    struct ChunkCtx { int k0; int lo; };
    __device__ int nchunks() const { return 2 * (kpad / FWN_BK); }
    __device__ RowCtx row_ctx(int row) const { return RowCtx{row, row % Ti}; }
    __device__ ChunkCtx chunk_ctx(int q) const {
        const int nk = kpad / FWN_BK;
        const int lo = q >= nk;
        return ChunkCtx{(q - lo * nk) * FWN_BK, lo};
    }
    __device__ float fetch(const RowCtx& rc, int k) const {
        // branch-free: clamp the address, select the value
        const int tap = k >> chlog, tau = k & (Ch - 1);
        const int tt = rc.t + tap - 1;
        const bool ok = rc.row < M && k < 3 * Ch && (unsigned)tt < (unsigned)Ti;
        const int srow = ok ? rc.row + tap - 1 : 0;
        float v = xa[(size_t)srow * Ch + tau];
        if (apply_an) v = (v + an[tau]) * an[Ch + tau];
        return ok ? v : 0.0f;
    }
    __device__ uint4 load_a(const ChunkCtx& cc, const RowCtx& rc, int c8) const {
        const int k8 = cc.k0 + c8 * 8;
        float f[8];
        if (Ch >= 8) {
            const int tap = k8 >> chlog, tau0 = k8 & (Ch - 1);
            const int tt = rc.t + tap - 1;
            const bool ok = rc.row < M && k8 < 3 * Ch && (unsigned)tt < (unsigned)Ti;
            const float* src = xa + (size_t)(ok ? rc.row + tap - 1 : 0) * Ch + (ok ? tau0 : 0);
            const float4 v0 = *(const float4*)src, v1 = *(const float4*)(src + 4);
            const float g[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int tau = ok ? tau0 + e : e;
                const float v = apply_an ? (g[e] + an[tau]) * an[Ch + tau] : g[e];
                f[e] = ok ? v : 0.0f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fetch(rc, k8 + e);
        }
        Pack16 out;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bf16 hi = (bf16)f[e];
            out.e[e] = cc.lo ? (bf16)(f[e] - (float)hi) : hi;
        }
        return out.u;
    }
    __device__ srd_t b_srd(const ChunkCtx&) const { return make_srd(W, (uint32_t)(256u * kpad * 2u)); }
    __device__ uint32_t b_voff(const ChunkCtx& cc, int n, int c8) const {
        return (uint32_t)(n * kpad + cc.k0 + c8 * 8) * 2u;
    }
    template <int MI>
    __device__ void epilogue(const f32x16 (&acc)[MI][2], int mrow0, int ncol0, int lane) const {
        const int lr = lane & 31;
        const srd_t so = make_srd(hout, (uint32_t)((size_t)M * FWN_HID * 2));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = ncol0 + ni * 32 + lr;
            const float b = bias[col];
            const uint32_t voff = (uint32_t)((mrow0 + 4 * (lane >> 5)) * FWN_HID + col) * 2u;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    buf_store_bf16(so, voff, (uint32_t)((mi * 32 + acc_row_c(r)) * FWN_HID * 2),
                                   fmaxf(acc[mi][ni][r] + b, 0.0f));
        }
    }
};

// ---- front conv for the early blocks (Ch <= 16, K = 3*Ch <= 48): fp32 VALU -------------------
// With so few input channels the conv is HBM-bound on writing h0 ([M][256] bf16), and an MFMA
// tile would be > 90 % zero padding.  One workgroup = 4*R rows x 256 channels; a wave owns R rows
// and each lane 4 channels; x (ActNorm applied, zero padded per clip) and the weights sit in LDS
// as fp32, so the flow state enters the network at full precision.
// KMAX >= K = 3 Ch sizes the LDS images: with the 48-deep images of the widest case every block held 60 KB and two
// workgroups (8 waves) per CU - too few stores in flight for an HBM-bound kernel; block 0 (K = 3) needs 4 KB.
template <int R, int KMAX>
__global__ __launch_bounds__(256) void front_valu_kernel(FrontProb p) {
    constexpr int ROWS = 4 * R;
    __shared__ __attribute__((aligned(16))) float wt[KMAX * 256];     // [k][256]
    __shared__ __attribute__((aligned(16))) float yt[KMAX * ROWS];    // [k][row]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = 3 * p.Ch;
    const int m0 = blockIdx.x * ROWS;
    {
        // Thread n stages row n of W ([256][kpad] bf16, kpad >= 64 > KMAX): its K values are one to six 16-byte pieces,
        // all in flight at once.  (Element by element - wt[k][n] = W[n][k] in a loop over k - every iteration was a
        // dependent 2-byte load: K serial L2 round trips per workgroup, ~10 us per launch whatever the row count.)
        constexpr int NP = (KMAX + 7) / 8;
        const uint4* wsrc = (const uint4*)(p.W + (size_t)tid * p.kpad);
        Pack16 wv[NP];
#pragma unroll
        for (int c = 0; c < NP; ++c) wv[c].u = wsrc[c];
#pragma unroll
        for (int c = 0; c < NP; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (c * 8 + e < KMAX) wt[(c * 8 + e) * 256 + tid] = (float)wv[c].e[e];
    }
    for (int i = tid; i < K * ROWS; i += 256) {
        const int r = i % ROWS, k = i / ROWS;
        const int row = m0 + r;
        yt[k * ROWS + r] = p.fetch(RowCtxT{row, row % p.Ti}, k);
    }
    __syncthreads();
    float acc[R][4];
    const float4 b4 = *(const float4*)(p.bias + 4 * lane);
#pragma unroll
    for (int r = 0; r < R; ++r) { acc[r][0] = b4.x; acc[r][1] = b4.y; acc[r][2] = b4.z; acc[r][3] = b4.w; }
    for (int k = 0; k < K; ++k) {
        const float4 w = *(const float4*)(wt + k * 256 + 4 * lane);
#pragma unroll
        for (int r4 = 0; r4 < R / 4; ++r4) {
            const float4 y = *(const float4*)(yt + k * ROWS + wave * R + 4 * r4);    // wave-uniform: broadcast
            const float yy[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[4 * r4 + j][0] += yy[j] * w.x; acc[4 * r4 + j][1] += yy[j] * w.y;
                acc[4 * r4 + j][2] += yy[j] * w.z; acc[4 * r4 + j][3] += yy[j] * w.w;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = m0 + wave * R + r;
        if (row < p.M) {
            union { bf16 e[4]; uint2 u; } o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o.e[c] = (bf16)fmaxf(acc[r][c], 0.0f);
            *(uint2*)(p.hout + (size_t)row * FWN_HID + 4 * lane) = o.u;
            if (p.h8out)         // quantise the bf16 value the bf16 path would read, not the fp32 one: one source of truth
                *(unsigned int*)(p.h8out + (size_t)row * FWN_HID + 4 * lane) =
                    pack_e4m3x2((float)o.e[0], (float)o.e[1]) | (pack_e4m3x2((float)o.e[2], (float)o.e[3]) << 16);
        }
    }
}

// ---- front conv for the late blocks (Ch >= 32) on the ring GEMM -----------------------------------
// The fp32 plane is first rewritten as a bf16 matrix [M][2*Ch] = (hi | lo) with ActNorm applied
// (xprep_kernel, HBM-bound, tiny), so the conv becomes three shifted DMA-able K segments like the
// dilated taps of the gate: K = tap*2Ch + half*Ch + tau against Wfront2.
__global__ __launch_bounds__(256) void xprep_kernel(const float* __restrict__ xa, const float* __restrict__ an,
                                                    int M, int Ch, int apply_an, bf16* __restrict__ xhl) {
    const long total = (long)M * Ch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int tau = (int)(i & (Ch - 1));
        const long row = i / Ch;
        float v = xa[i];
        if (apply_an) v = (v + an[tau]) * an[Ch + tau];
        const bf16 hi = (bf16)v;
        xhl[row * 2 * Ch + tau] = hi;
        xhl[row * 2 * Ch + Ch + tau] = (bf16)(v - (float)hi);
    }
}

struct FrontRingProb {
    static constexpr bool A_DMA = true;
    static constexpr bool ALLOW_256 = false;
    const bf16* xhl;      // [M][2*Ch]
    const bf16* W;        // [256][6*Ch]
    const float* bias;    // [256]
    bf16* hout;           // [M][256]
    int M, Ti, Ch;
    typedef RowCtxT RowCtx;
    struct ChunkCtx { int acol, bcol, shift; };
    template <int BK> __device__ int nchunks() const { return 6 * Ch / BK; }
    __device__ RowCtx row_ctx(int row) const { return RowCtx{row, row % Ti}; }
    template <int BK> __device__ ChunkCtx chunk_ctx(int q) const {
        const int cpt = 2 * Ch / BK;               // chunks per tap
        const int tap = q / cpt, kc = q % cpt;
        return ChunkCtx{kc * BK, tap * 2 * Ch + kc * BK, tap - 1};
    }
    __device__ srd_t a_srd(const ChunkCtx&) const { return make_srd(xhl, (uint32_t)((size_t)M * 2 * Ch * 2)); }
    __device__ uint32_t a_voff(const ChunkCtx& cc, const RowCtx& rc, int c8) const {
        const bool ok = rc.row < M && (unsigned)(rc.t + cc.shift) < (unsigned)Ti;
        return ok ? (uint32_t)((rc.row + cc.shift) * 2 * Ch + cc.acol + c8 * 8) * 2u : FWN_OOB;
    }
    __device__ srd_t b_srd(const ChunkCtx&) const { return make_srd(W, (uint32_t)(256u * 6u * Ch * 2u)); }
    __device__ uint32_t b_voff(const ChunkCtx& cc, int n, int c8) const {
        return (uint32_t)(n * 6 * Ch + cc.bcol + c8 * 8) * 2u;
    }
    __device__ float acc_init(int) const { return 0.0f; }
    template <int MI>
    __device__ void epilogue(const f32x16 (&acc)[MI][2], int mrow0, int ncol0, int lane) const {
        const int lr = lane & 31;
        const srd_t so = make_srd(hout, (uint32_t)((size_t)M * FWN_HID * 2));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = ncol0 + ni * 32 + lr;
            const float b = bias[col];
            const uint32_t voff = (uint32_t)((mrow0 + 4 * (lane >> 5)) * FWN_HID + col) * 2u;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    buf_store_bf16(so, voff, (uint32_t)((mi * 32 + acc_row_c(r)) * FWN_HID * 2),
                                   fmaxf(acc[mi][ni][r] + b, 0.0f));
        }
    }
};

// ---- front conv for the late blocks in ONE launch (round 3) ----------------------------------------
// xprep_kernel + the ring GEMM were two dependent launches of a few dozen workgroups (4.7 + 5.3 us per flow, 18 flows per
// pass).  Here a workgroup builds the (hi | lo) bf16 image of ITS rows (+ one halo row on either side) in LDS itself - ActNorm
// applied, zero rows outside the clip - and the three taps read that one image at row offsets 0, 1, 2 (the tap sharing of
// gate_halo.h); only the weights (Wfront2, K = tap*2Ch + half*Ch + tau) stream through a 4-slot LDS-DMA ring.
// Tile 64 rows x 64 channels, 4 waves as 2 x 2, one 32 x 32 accumulator tile each.
// LDS allocation: what it uses (64 - 98 KB).  Round 3 padded it to 156 KB because with its real size ~1 overlapped step in
// 100 came back with one clip off by ~1e-2 whenever other workgroups shared the CU.  Root cause (round 4, DESIGN.md section
// 3.5): that version refilled its weight ring inside the K loop, and hipcc had sunk the s_waitcnt lgkmcnt of the previous
// chunk's fragment reads below the raw s_barrier that licensed the refill - the LDS-DMA could overtake reads still in
// flight once a neighbour slowed the LDS.  This version issues every chunk in the prologue (no slot is ever reused), the
// ring kernels retire their LDS reads before their barriers (FWN_RING_BARRIER), and the real allocation soaks clean
// (tests/test_gpu_parity.py::test_overlapped_streams_soak).  FWN_FRONT_LDS_MIN remains as a developer switch.
#ifndef FWN_FRONT_LDS_MIN
#define FWN_FRONT_LDS_MIN 0
#endif
// CHP = channels of the plane (CHP < CH: the image and the packed weights are zero padded to CH = 32 channels per half - Ch = 16,
// whose K = 96 does not divide into 64-wide chunks per tap; the VALU kernel it replaces took 8 - 10 us per launch)
template <int CH, int CHP = CH>
__global__ __launch_bounds__(256) void front_mfma_kernel(const float* __restrict__ xa, const float* __restrict__ an, int apply_an,
                                                         const bf16* __restrict__ W2, const float* __restrict__ bias,
                                                         bf16* __restrict__ hout, int M, int Ti) {
    constexpr int RB = 4 * CH;                      // image row bytes: (hi | lo) x CH bf16
    constexpr int NR = 66, ZROW = 66;               // rows m0 - 1 .. m0 + 64, then one row of zeros
    constexpr int CPT = 2 * CH / 64;                // 64-wide K chunks per tap
    // every weight chunk of the workgroup's 64 output rows in flight from the prologue on (8 KB each, 96 KB for CH = 128:
    // the LDS is this workgroup's alone anyway, FWN_FRONT_LDS_MIN): one DMA latency for the whole slice instead of one per
    // three chunks - at a handful of workgroups per launch the K loop is nothing but that latency chain
    constexpr int NQ = 3 * CPT, D = NQ + 1;
    constexpr int A_BYTES = (NR + 1) * RB, B_SLOT = 64 * 128;
    constexpr int LDS_USED = D * B_SLOT + A_BYTES;
    constexpr int LDS_BYTES = LDS_USED > FWN_FRONT_LDS_MIN ? LDS_USED : FWN_FRONT_LDS_MIN;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    unsigned char* const ldsA = lds + D * B_SLOT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, lh = lane >> 5;
    const int tile_m = blockIdx.x >> 2, tile_n = blockIdx.x & 3;
    const int m0 = tile_m * 64, n0 = tile_n * 64;
    // 16-byte piece p of image row `row`: conflict-free for 32 lanes reading 32 rows at one piece index
    auto a_off = [](int row, int p) -> int {
        if constexpr (CH == 32) return row * 128 + ((p ^ ((row >> 1) & 7)) << 4);
        else if constexpr (CH == 64) return row * 256 + ((p ^ (row & 15)) << 4);
        else return row * 512 + ((p >> 4) << 8) + (((p & 15) ^ (row & 15)) << 4);
    };
    // weight ring: chunk q = columns [64 q, +64) of rows n0 .. n0 + 63: 8 pieces, two per wave
    const srd_t sw = make_srd(W2, (uint32_t)(256u * 6u * CH * 2u));
    auto issue = [&](int q) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 8 * (wave + 4 * j) + (lane >> 3);
            buf_load16_lds_nt(sw, (uint32_t)((n0 + r) * 6 * CH + q * 64 + ((lane & 7) ^ ((r >> 1) & 7)) * 8) * 2u,
                              lds + (q % D) * B_SLOT + (wave + 4 * j) * 1024);
        }
    };
#pragma unroll
    for (int q = 0; q < D - 1; ++q)
        if (q < NQ) issue(q);
    // the image: 4 channels of one row per task - fp32 in, ActNorm, (hi | lo) out
    constexpr int NTASK = (NR + 1) * (CHP / 4), NIT = (NTASK + 255) / 256;
    static_assert(256 % (CHP / 4) == 0, "a thread's channel group is the same for all its tasks");
    if constexpr (CHP < CH) {                       // the padded channels of both halves: zeros (their weights are zeros too, but 0 x NaN is not)
        constexpr int PP = (CH - CHP) / 8;          // 16-byte pieces per half and row
        for (int i = tid; i < (NR + 1) * 2 * PP; i += 256) {
            const int j = i / (2 * PP), q = i % (2 * PP);
            *(uint4*)(ldsA + a_off(j, (q / PP) * (CH / 8) + CHP / 8 + q % PP)) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    // (CH / 4 divides 256: every task of a thread has the same tau - its ActNorm shift / scale are loaded ONCE, together with
    // the plane rows; looked up per task they were 9 dependent L2 round trips in front of the K loop at CH = 128)
    const int tau_t = (tid % (CHP / 4)) * 4;
    const float4 an_sh = apply_an ? *(const float4*)(an + tau_t) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4 an_sc = apply_an ? *(const float4*)(an + CHP + tau_t) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    float4 vin[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {              // every load of the thread in flight at once (clamped addresses)
        const int task = tid + it * 256;
        const int j = task / (CHP / 4), tau = (task % (CHP / 4)) * 4;
        const int g = m0 - 1 + j;
        const bool ok = task < NTASK && j < NR && (unsigned)g < (unsigned)M;
        vin[it] = *(const float4*)(xa + (size_t)(ok ? g : 0) * CHP + (ok ? tau : 0));
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int task = tid + it * 256;
        if (task >= NTASK) break;
        const int j = task / (CHP / 4), tau = (task % (CHP / 4)) * 4;
        const int g = m0 - 1 + j;
        const bool ok = j < NR && (unsigned)g < (unsigned)M;
        const float4 v = vin[it];
        const float f[4] = {v.x, v.y, v.z, v.w}, sh[4] = {an_sh.x, an_sh.y, an_sh.z, an_sh.w}, sc[4] = {an_sc.x, an_sc.y, an_sc.z, an_sc.w};
        union { bf16 e[4]; uint2 u; } hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float y = apply_an ? (f[e] + sh[e]) * sc[e] : f[e];
            y = ok ? y : 0.0f;
            hi.e[e] = (bf16)y;
            lo.e[e] = (bf16)(y - (float)hi.e[e]);
        }
        *(uint2*)(ldsA + a_off(j, tau >> 3) + (tau & 7) * 2) = hi.u;
        *(uint2*)(ldsA + a_off(j, (CH + tau) >> 3) + (tau & 7) * 2) = lo.u;
    }
    // this lane's fragment rows per tap: image row (local + tap), or the zero row outside the clip / the matrix
    int arow[3];
    {
        const int r = m0 + wm * 32 + lr;
        const int t = r % Ti;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
            arow[tap] = (r < M && (unsigned)(t + tap - 1) < (unsigned)Ti) ? wm * 32 + lr + tap : ZROW;
    }
    f32x16 acc;
    {
        const float b0 = bias[n0 + wn * 32 + lr];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = b0;
    }
    const int bfr = (wn * 32 + lr) * 128;           // + swizzled piece: the B rows of this wave
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        // chunks issued so far: min(NQ, q + D - 1); those after q may stay in flight (2 pieces per wave and chunk)
        // every chunk was issued in the prologue and the image was built behind the plane loads, which were issued after
        // them (vmcnt retires in order): ONE barrier makes all of it visible, the K loop runs without waits
        if (q == 0) {
            FWN_WAIT_VMCNT(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the image rows this wave wrote
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        const unsigned char* lb = lds + (q % D) * B_SLOT;
        const int tap = q / CPT, kc = q % CPT;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 a = *(const bf16x8*)(ldsA + a_off(arow[tap], kc * 8 + kk * 2 + lh));
            const bf16x8 b = *(const bf16x8*)(lb + bfr + ((((kk * 2 + lh) ^ (((wn * 32 + lr) >> 1) & 7))) << 4));
            acc = mfma32(a, b, acc);
        }
    }
    const srd_t so = make_srd(hout, (uint32_t)((size_t)M * FWN_HID * 2));
    const uint32_t voff = (uint32_t)((m0 + wm * 32 + 4 * lh) * FWN_HID + n0 + wn * 32 + lr) * 2u;
#pragma unroll
    for (int r = 0; r < 16; ++r) buf_store_bf16(so, voff, (uint32_t)(acc_row_c(r) * FWN_HID * 2), fmaxf(acc[r], 0.0f));
}

// ---- gated dilated layer: o = tanh(f) * sigmoid(g), modules.py:113-124 ---------------------
// K segments: 3 dilated taps over h (K = 3*256) then the 1x1 conditioning conv over c_a
// (K = cin), or a precomputed conditioning projection P added in the epilogue.
struct GateProb {
    static constexpr bool A_DMA = true;
    static constexpr bool ALLOW_256 = true;
    const bf16* h;        // [M][256]
    const bf16* ca;       // [M][cin] or nullptr
    const float* P;       // [M][512] packed-N order, or nullptr
    const bf16* Wd;       // [512][768]   packed-N rows, k = tap*256 + ch
    const bf16* Wc;       // [512][kcpad] packed-N rows
    const float* bias;    // [512] packed-N order (conv bias + cond bias)
    bf16* o;              // [M][256]
    int M, Ti, dil, cin, kcpad;
    bf16* aux = nullptr;  // training only: [M][512] = (tanh f | sigmoid g) kept for the backward pass
    // fp8 dilated taps (gate_halo_kernel<.., FP8>): e4m3 copies of h and of the packed conv weights (stored as W 2^e)
    const unsigned char* h8 = nullptr;    // [M][256]
    const unsigned char* Wd8 = nullptr;   // [512][768]
    int sb = 127;                         // E8M0 scale operand of the weights: 127 - e
#ifdef FWN_STAMP
    unsigned long long* stamps = nullptr; // diagnostic build only: [2 workgroups][16 waves][24 steps][4] s_memtime stamps
#endif
    typedef RowCtxT RowCtx;
    struct ChunkCtx { int cond, acol, bcol, shift, kvalid; };
    template <int BK> __device__ int nchunks() const { return (3 * FWN_HID + (ca ? kcpad : 0)) / BK; }
    __device__ RowCtx row_ctx(int row) const { return RowCtx{row, row % Ti}; }
    template <int BK> __device__ ChunkCtx chunk_ctx(int q) const {
        constexpr int CPT = FWN_HID / BK;          // chunks per tap
        if (q < 3 * CPT) {
            const int tap = q / CPT, kc = q % CPT;
            return ChunkCtx{0, kc * BK, tap * FWN_HID + kc * BK, (tap - 1) * dil, BK};
        }
        const int qc = q - 3 * CPT;
        return ChunkCtx{1, qc * BK, qc * BK, 0, cin - qc * BK};
    }
    __device__ srd_t a_srd(const ChunkCtx& cc) const {
        return cc.cond ? make_srd(ca, (uint32_t)((size_t)M * cin * 2)) : make_srd(h, (uint32_t)((size_t)M * FWN_HID * 2));
    }
    __device__ uint32_t a_voff(const ChunkCtx& cc, const RowCtx& rc, int c8) const {
        const bool ok = rc.row < M && (unsigned)(rc.t + cc.shift) < (unsigned)Ti && c8 * 8 < cc.kvalid;
        const int lda = cc.cond ? cin : FWN_HID;
        const uint32_t off = (uint32_t)((rc.row + cc.shift) * lda + cc.acol + c8 * 8) * 2u;
        return ok ? off : FWN_OOB;
    }
    __device__ srd_t b_srd(const ChunkCtx& cc) const {
        return cc.cond ? make_srd(Wc, (uint32_t)(512u * kcpad * 2u)) : make_srd(Wd, 512u * 768u * 2u);
    }
    __device__ uint32_t b_voff(const ChunkCtx& cc, int n, int c8) const {
        const int ldb = cc.cond ? kcpad : 3 * FWN_HID;
        return (uint32_t)(n * ldb + cc.bcol + c8 * 8) * 2u;
    }
    __device__ float acc_init(int col) const { return bias[col]; }   // packed-N order, like the columns
    // gemm_ring.h PREFETCH (small tiles): the hoisted conditioning projection of this tile, filter then gate columns
    static constexpr bool PREFETCH = true;
    template <int MI>
    __device__ void prefetch(float (&pre)[MI][32], int mrow0, int ncol0, int lane) const {
        if (!P) return;
        const srd_t sp = make_srd(P, (uint32_t)((size_t)M * 512 * 4));
        const uint32_t vp = (uint32_t)((mrow0 + 4 * (lane >> 5)) * 512 + ncol0 + (lane & 31)) * 4u;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t sro = (uint32_t)((mi * 32 + acc_row_c(r)) * 512 * 4);
                pre[mi][r] = buf_load_f32(sp, vp, sro);
                pre[mi][16 + r] = buf_load_f32(sp, vp, sro + 128);
            }
    }
    template <int MI>
    __device__ void epilogue_pre(const f32x16 (&acc)[MI][2], int mrow0, int ncol0, int lane, const float (&pre)[MI][32]) const {
        epilogue_impl<MI, true>(acc, mrow0, ncol0, lane, pre);
    }
    template <int MI>
    __device__ void epilogue(const f32x16 (&acc)[MI][2], int mrow0, int ncol0, int lane) const {
        const float none[MI][32] = {};
        epilogue_impl<MI, false>(acc, mrow0, ncol0, lane, none);
    }
    template <int MI, bool PRE>
    __device__ __forceinline__ void epilogue_impl(const f32x16 (&acc)[MI][2], int mrow0, int ncol0, int lane, const float (&pre)[MI][32]) const {
        // packed-N: ncol0 = nb*128 + wn*64; columns [ncol0, +32) = filter, [+32, +64) = gate of
        // channels nb*64 + wn*32 + lr.
        const int lr = lane & 31;
        const int ch = (ncol0 >> 7) * 64 + ((ncol0 >> 6) & 1) * 32 + lr;
        const srd_t so = make_srd(o, (uint32_t)((size_t)M * FWN_HID * 2));
        const int rbase = mrow0 + 4 * (lane >> 5);
        const uint32_t voff = (uint32_t)(rbase * FWN_HID + ch) * 2u;
        if (aux) {   // training forward: the two factors are stored for the gate's derivative (P: hoisted conditioning)
            // branch-free like the other paths: rows past M fall outside the buffer descriptors
            const srd_t sp = make_srd(P ? (const void*)P : (const void*)o, P ? (uint32_t)((size_t)M * 512 * 4) : 0u);
            const srd_t sa = make_srd(aux, (uint32_t)((size_t)M * 512 * 2));
            const uint32_t vp = (uint32_t)(rbase * 512 + ncol0 + lr) * 4u;
            const uint32_t va = (uint32_t)(rbase * 512 + ch) * 2u;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                float pf[16], pg[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t sro = (uint32_t)((mi * 32 + acc_row_c(r)) * 512 * 4);
                    pf[r] = !P ? 0.0f : PRE ? pre[mi][r] : buf_load_f32(sp, vp, sro);
                    pg[r] = !P ? 0.0f : PRE ? pre[mi][16 + r] : buf_load_f32(sp, vp, sro + 128);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float a = __builtin_amdgcn_exp2f(fminf(acc[mi][0][r] + pf[r], 40.0f));
                    const float b = __builtin_amdgcn_exp2f(fminf(acc[mi][1][r] + pg[r], 40.0f));
                    const float tf = (1.0f - a) * __builtin_amdgcn_rcpf(1.0f + a), sg = __builtin_amdgcn_rcpf(1.0f + b);
                    const uint32_t ro = (uint32_t)(mi * 32 + acc_row_c(r));
                    buf_store_bf16(so, voff, ro * FWN_HID * 2, tf * sg);
                    buf_store_bf16(sa, va, ro * 512 * 2, tf);
                    buf_store_bf16(sa, va, ro * 512 * 2 + 512, sg);
                }
            }
        } else if (P) {    // hoisted conditioning projection (fp32 [M][512], packed-N columns)
            const srd_t sp = make_srd(P, (uint32_t)((size_t)M * 512 * 4));
            const uint32_t vp = (uint32_t)(rbase * 512 + ncol0 + lr) * 4u;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                float pf[16], pg[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t sro = (uint32_t)((mi * 32 + acc_row_c(r)) * 512 * 4);
                    pf[r] = PRE ? pre[mi][r] : buf_load_f32(sp, vp, sro);
                    pg[r] = PRE ? pre[mi][16 + r] : buf_load_f32(sp, vp, sro + 128);
                }
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 y = gated_unit2(
                        f32x2{acc[mi][0][r], acc[mi][0][r + 1]} + f32x2{pf[r], pf[r + 1]},
                        f32x2{acc[mi][1][r], acc[mi][1][r + 1]} + f32x2{pg[r], pg[r + 1]});
                    buf_store_bf16(so, voff, (uint32_t)((mi * 32 + acc_row_c(r)) * FWN_HID * 2), y.x);
                    buf_store_bf16(so, voff, (uint32_t)((mi * 32 + acc_row_c(r + 1)) * FWN_HID * 2), y.y);
                }
            }
        } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 y = gated_unit2(f32x2{acc[mi][0][r], acc[mi][0][r + 1]},
                                                f32x2{acc[mi][1][r], acc[mi][1][r + 1]});
                    buf_store_bf16(so, voff, (uint32_t)((mi * 32 + acc_row_c(r)) * FWN_HID * 2), y.x);
                    buf_store_bf16(so, voff, (uint32_t)((mi * 32 + acc_row_c(r + 1)) * FWN_HID * 2), y.y);
                }
        }
    }
};

#include "gate_halo.h"

// ---- residual 1x1: h' = (h + res_conv(o)) * sqrt(0.5), modules.py:126-128 ------------------
struct ResProb {
    static constexpr bool A_DMA = true;
    static constexpr bool ALLOW_256 = false;   // 4 row tiles of residual reads would spill at 256 VGPRs
    const bf16* o;        // [M][256]
    const bf16* hin;      // [M][256]
    const bf16* W;        // [256][256]
    const float* bias;    // [256]
    bf16* hout;           // [M][256]
    int M;
    unsigned char* h8out = nullptr;   // [M][256] e4m3 copy of hout for the next layer's fp8 gate
    struct RowCtx { int row; };
    struct ChunkCtx { int k0; };
    template <int BK> __device__ int nchunks() const { return FWN_HID / BK; }
    __device__ RowCtx row_ctx(int row) const { return RowCtx{row}; }
    template <int BK> __device__ ChunkCtx chunk_ctx(int q) const { return ChunkCtx{q * BK}; }
    __device__ srd_t a_srd(const ChunkCtx&) const { return make_srd(o, (uint32_t)((size_t)M * FWN_HID * 2)); }
    __device__ uint32_t a_voff(const ChunkCtx& cc, const RowCtx& rc, int c8) const {
        return rc.row < M ? (uint32_t)(rc.row * FWN_HID + cc.k0 + c8 * 8) * 2u : FWN_OOB;
    }
    __device__ srd_t b_srd(const ChunkCtx&) const { return make_srd(W, 256u * 256u * 2u); }
    __device__ uint32_t b_voff(const ChunkCtx& cc, int n, int c8) const {
        return (uint32_t)(n * FWN_HID + cc.k0 + c8 * 8) * 2u;
    }
    __device__ float acc_init(int) const { return 0.0f; }
    // Row-major epilogue (gemm_ring.h, LDS_EPI): h in, h out (and the e4m3 copy) as 16- / 8-byte pieces of 8 columns.
    // This layer is HBM-bound (o read, h read, h write); with one column per lane it issued 64 two-byte VMEM
    // instructions per lane.  Same arithmetic, same order: (h + acc + b) * sqrt(1/2).
    static constexpr bool LDS_EPI = true;
    __device__ bool rows_launch() const { return true; }
    __device__ bool rows_tile(int) const { return true; }
    template <int MI>
    __device__ void epilogue_rows(const float* wt, int mrow0, int ncol0, int lane) const {
        const int col = ncol0 + (lane & 7) * 8;
        const float4 b0 = *(const float4*)(bias + col), b1 = *(const float4*)(bias + col + 4);
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        const uint32_t bytes = (uint32_t)((size_t)M * FWN_HID * 2);
        const srd_t so = make_srd(hout, bytes), s8 = make_srd(h8out ? (void*)h8out : (void*)hout, h8out ? (uint32_t)((size_t)M * FWN_HID) : 0u);
        Pack16 hv[4 * MI];
#pragma unroll
        for (int it = 0; it < 4 * MI; ++it) {        // plain (clamped) 16-byte loads: see DESIGN.md on raw_buffer_load_b128
            const int row = mrow0 + it * 8 + (lane >> 3);
            hv[it].u = *(const uint4*)(hin + (size_t)(row < M ? row : 0) * FWN_HID + col);
        }
#pragma unroll
        for (int it = 0; it < 4 * MI; ++it) {
            const int row = mrow0 + it * 8 + (lane >> 3);
            float a[8];
            lds_epi_take(wt, it, lane, a);
            Pack16 out;
#pragma unroll
            for (int e = 0; e < 8; ++e) out.e[e] = (bf16)(((float)hv[it].e[e] + a[e] + bb[e]) * 0.70710678118654752f);
            const uint32_t voff = row < M ? (uint32_t)(row * FWN_HID + col) * 2u : FWN_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(out.w, so, voff, 0, 0);
            if (h8out) {
                typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
                const u32x2 q = {pack_e4m3x2((float)out.e[0], (float)out.e[1]) | (pack_e4m3x2((float)out.e[2], (float)out.e[3]) << 16),
                                 pack_e4m3x2((float)out.e[4], (float)out.e[5]) | (pack_e4m3x2((float)out.e[6], (float)out.e[7]) << 16)};
                __builtin_amdgcn_raw_buffer_store_b64(q, s8, voff == FWN_OOB ? FWN_OOB : voff >> 1, 0, 0);
            }
        }
    }
    // gemm_ring.h PREFETCH (small tiles): the residual rows of this tile
    static constexpr bool PREFETCH = true;
    template <int MI>
    __device__ void prefetch(float (&pre)[MI][32], int mrow0, int ncol0, int lane) const {
        const srd_t si = make_srd(hin, (uint32_t)((size_t)M * FWN_HID * 2));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const uint32_t voff = (uint32_t)((mrow0 + 4 * (lane >> 5)) * FWN_HID + ncol0 + ni * 32 + (lane & 31)) * 2u;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    pre[mi][ni * 16 + r] = buf_load_bf16(si, voff, (uint32_t)((mi * 32 + acc_row_c(r)) * FWN_HID * 2));
        }
    }
    template <int MI>
    __device__ void epilogue_pre(const f32x16 (&acc)[MI][2], int mrow0, int ncol0, int lane, const float (&pre)[MI][32]) const {
        epilogue_impl<MI, true>(acc, mrow0, ncol0, lane, pre);
    }
    template <int MI>
    __device__ void epilogue(const f32x16 (&acc)[MI][2], int mrow0, int ncol0, int lane) const {
        const float none[MI][32] = {};
        epilogue_impl<MI, false>(acc, mrow0, ncol0, lane, none);
    }
    template <int MI, bool PRE>
    __device__ __forceinline__ void epilogue_impl(const f32x16 (&acc)[MI][2], int mrow0, int ncol0, int lane, const float (&pre)[MI][32]) const {
        const int lr = lane & 31;
        const uint32_t bytes = (uint32_t)((size_t)M * FWN_HID * 2);
        const srd_t si = make_srd(hin, bytes), so = make_srd(hout, bytes);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = ncol0 + ni * 32 + lr;
            const float b = bias[col];
            const uint32_t voff = (uint32_t)((mrow0 + 4 * (lane >> 5)) * FWN_HID + col) * 2u;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                float hv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    hv[r] = PRE ? pre[mi][ni * 16 + r] : buf_load_bf16(si, voff, (uint32_t)((mi * 32 + acc_row_c(r)) * FWN_HID * 2));
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    buf_store_bf16(so, voff, (uint32_t)((mi * 32 + acc_row_c(r)) * FWN_HID * 2),
                                   (hv[r] + acc[mi][ni][r] + b) * 0.70710678118654752f);
                if (h8out) {
                    const srd_t s8 = make_srd(h8out, (uint32_t)((size_t)M * FWN_HID));
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const unsigned int pk = pack_e4m3x2((float)(bf16)((hv[r] + acc[mi][ni][r] + b) * 0.70710678118654752f),
                                                            (float)(bf16)((hv[r + 1] + acc[mi][ni][r + 1] + b) * 0.70710678118654752f));
                        buf_store_u8(s8, voff >> 1, (uint32_t)((mi * 32 + acc_row_c(r)) * FWN_HID), pk);
                        buf_store_u8(s8, voff >> 1, (uint32_t)((mi * 32 + acc_row_c(r + 1)) * FWN_HID), pk >> 8);
                    }
                }
            }
        }
    }
};

// ---- conditioning projection hoisted out of the flow chain: P = c_a @ Wc -------------------
struct CondProb {
    static constexpr bool NT_B = true;    // K = cin up to 10240: the weight rows are streamed once by a few row tiles (FWN_NT_SMALL)
    static constexpr bool A_DMA = true;
    static constexpr bool ALLOW_256 = false;
    const bf16* ca;       // [M][cin]
    const bf16* Wc;       // [512][kcpad]
    float* P;             // [M][512] (this split's output)
    int M, cin, kcpad;
    int k_begin = 0, k_end = 0;   // K range of this workgroup's split (k_end == 0: all of kcpad); multiples of 64
    struct RowCtx { int row; };
    struct ChunkCtx { int k0; };
    template <int BK> __device__ int nchunks() const { return ((k_end ? k_end : kcpad) - k_begin) / BK; }
    __device__ RowCtx row_ctx(int row) const { return RowCtx{row}; }
    template <int BK> __device__ ChunkCtx chunk_ctx(int q) const { return ChunkCtx{k_begin + q * BK}; }
    __device__ srd_t a_srd(const ChunkCtx&) const { return make_srd(ca, (uint32_t)((size_t)M * cin * 2)); }
    __device__ uint32_t a_voff(const ChunkCtx& cc, const RowCtx& rc, int c8) const {
        const bool ok = rc.row < M && cc.k0 + c8 * 8 < cin;
        return ok ? (uint32_t)(rc.row * cin + cc.k0 + c8 * 8) * 2u : FWN_OOB;
    }
    __device__ srd_t b_srd(const ChunkCtx&) const { return make_srd(Wc, (uint32_t)(512u * kcpad * 2u)); }
    __device__ uint32_t b_voff(const ChunkCtx& cc, int n, int c8) const {
        return (uint32_t)(n * kcpad + cc.k0 + c8 * 8) * 2u;
    }
    __device__ float acc_init(int) const { return 0.0f; }
    template <int MI>
    __device__ void epilogue(const f32x16 (&acc)[MI][2], int mrow0, int ncol0, int lane) const {
        const int lr = lane & 31;
        const srd_t sp = make_srd(P, (uint32_t)((size_t)M * 512 * 4));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const uint32_t voff = (uint32_t)((mrow0 + 4 * (lane >> 5)) * 512 + ncol0 + ni * 32 + lr) * 4u;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    buf_store_f32(sp, voff, (uint32_t)((mi * 32 + acc_row_c(r)) * 512 * 4), acc[mi][ni][r]);
        }
    }
};

template <int MI, class Prob>
__global__ __launch_bounds__(256) void gemm128_kernel(Prob p, int ntn) {
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    gemm128_body<MI, Prob>(p, wg / ntn, wg % ntn);
}

// Ring-pipelined GEMM: tile BM x BN, WM x WN waves, K-chunk BK, ring depth D.
template <int BM, int BN, int WM, int WN, int BK, int D, class Prob, int KSP = 1>
__global__ __launch_bounds__(64 * WM * WN * KSP) void gemm_ring_kernel(Prob p, int ntn) {
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    gemm_ring_body<BM, BN, WM, WN, BK, D, Prob, KSP>(p, wg / ntn, wg % ntn);
}

// Batched conditioning projections: blockIdx.y selects (flow, layer) of one parity group.
struct CondBatch {
    const bf16* ca;
    const bf16* ca_odd;   // != NULL: flows with an odd index read this plane (both parity groups of a block in one launch)
    const bf16* Wc_base;
    float* P_base;
    long w_stride;        // elements between consecutive (flow*L + layer) weight matrices
    long p_stride;        // elements between consecutive P matrices
    int flow0, flow_step, L;
    int M, cin, kcpad;
    // split-K over blockIdx.z (few rows, long K: a workgroup would stream K (BM + BN) 2 bytes through one CU): split 0
    // writes P, split z > 0 the same matrix in part_base + (z - 1) part_stride; fwn_launch_cond_reduce adds them in order
    float* part_base;
    long part_stride;
    int nsplit;
};
template <int BM, int BN, int WM, int WN, int BK, int D>
__global__ __launch_bounds__(64 * WM * WN) void cond_batch_kernel(CondBatch cb, int ntn) {
    const int z = blockIdx.y;
    const int flow = cb.flow0 + (z / cb.L) * cb.flow_step;
    const int zi = flow * cb.L + (z % cb.L);
    const int sp = blockIdx.z;
    float* out = (sp == 0 ? cb.P_base : cb.part_base + (size_t)(sp - 1) * cb.part_stride) + (size_t)zi * cb.p_stride;
    CondProb p{(cb.ca_odd && (flow & 1)) ? cb.ca_odd : cb.ca, cb.Wc_base + (size_t)zi * cb.w_stride, out, cb.M, cb.cin, cb.kcpad};
    if (cb.nsplit > 1) {       // 64-wide chunks dealt out evenly; every split gets at least one (fwn_cond_splits)
        const int nch = cb.kcpad / 64, per = (nch + cb.nsplit - 1) / cb.nsplit;
        p.k_begin = min(sp * per, nch) * 64;
        p.k_end = min((sp + 1) * per, nch) * 64;
        if (p.k_end == 0) p.k_end = 64, p.k_begin = 64;          // (cannot happen: split 0 always has chunks) keep nchunks() >= 0
    }
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    gemm_ring_body<BM, BN, WM, WN, BK, D, CondProb>(p, wg / ntn, wg % ntn);
}


// ---------------------------------------------------------------------------
// N-split tail for small row counts (M <= FWN_TAIL_SPLIT_MAX).
// The register-chained tail_kernel below streams ALL tail weights of a flow (0.4 - 0.5 MB) through every workgroup:
// at ~57 GB/s per CU that alone is 9 - 13 us, and with few rows there is nothing to amortise it over (22 - 36 us per
// launch at 4 .. 126 workgroups).  Below the threshold the tail runs as three ring GEMMs whose weights are split over
// workgroups by output column instead:  S = ReLU(sum_l o_l Wskip_l + bs)  ->  U = ReLU(S Wfinal + bf)  ->
// (log_s | t) = U Wzero, coupling + ActNorm (+ log-det partials) in the epilogue.
// Wskip / Wfinal rows are packed in accumulator order (packing.acc_k_perm = bits 2 and 3 of the channel index
// swapped - an involution), applied to the COLUMN of the 2-byte epilogue stores.
// ---------------------------------------------------------------------------
#include "tail_zero_prob.h"

struct TailLinProb {      // Y' = ReLU(sum_l A_l[M][256] . W[:, l*256 ..]^T + bias), columns stored at swap_bits23(col)
    static constexpr bool A_DMA = true;
    static constexpr bool ALLOW_256 = false;
    const bf16* A;        // [L][a_stride] rows of 256
    const bf16* W;        // [256][L*256]
    const float* bias;    // [256]
    bf16* out;            // [M][256]
    long a_stride;        // elements between layers of A
    int L, M;
    struct RowCtx { int row; };
    struct ChunkCtx { uint32_t abase; int bcol; };
    template <int BK> __device__ int nchunks() const { return L * FWN_HID / BK; }
    __device__ RowCtx row_ctx(int row) const { return RowCtx{row}; }
    template <int BK> __device__ ChunkCtx chunk_ctx(int q) const {
        constexpr int CPL = FWN_HID / BK;
        const int l = q / CPL, k0 = (q % CPL) * BK;
        return ChunkCtx{(uint32_t)(l * a_stride + k0), l * FWN_HID + k0};
    }
    __device__ srd_t a_srd(const ChunkCtx&) const { return make_srd(A, (uint32_t)(((size_t)(L - 1) * a_stride + (size_t)M * FWN_HID) * 2)); }
    __device__ uint32_t a_voff(const ChunkCtx& cc, const RowCtx& rc, int c8) const {
        return rc.row < M ? (cc.abase + (uint32_t)(rc.row * FWN_HID + c8 * 8)) * 2u : FWN_OOB;
    }
    __device__ srd_t b_srd(const ChunkCtx&) const { return make_srd(W, (uint32_t)(256u * L * FWN_HID * 2u)); }
    __device__ uint32_t b_voff(const ChunkCtx& cc, int n, int c8) const {
        return (uint32_t)(n * L * FWN_HID + cc.bcol + c8 * 8) * 2u;
    }
    __device__ float acc_init(int col) const { return bias[col]; }
    template <int MI>
    __device__ void epilogue(const f32x16 (&acc)[MI][2], int mrow0, int ncol0, int lane) const {
        const int lr = lane & 31;
        const srd_t so = make_srd(out, (uint32_t)((size_t)M * FWN_HID * 2));
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = swap_bits23(ncol0 + ni * 32 + lr);
            const uint32_t voff = (uint32_t)((mrow0 + 4 * (lane >> 5)) * FWN_HID + col) * 2u;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    buf_store_bf16(so, voff, (uint32_t)((mi * 32 + acc_row_c(r)) * FWN_HID * 2), fmaxf(acc[mi][ni][r], 0.0f));
        }
    }
};

#include "tail_chain.h"

// ---------------------------------------------------------------------------
// Host-side launchers (called from the C-ABI in api.hip)
// ---------------------------------------------------------------------------
// FWN_TAIL128_TWO = 1: from FWN_TAIL256_MIN rows on (block 0 of the 8-clip pass: two workgroups of 128 rows per CU exist)
// the fused tail runs as 128-row workgroups with 32-wide phase-1 chunks in TWO 32 KB slots (72 KB of LDS: two per CU, one's
// DMA latency and epilogue under the other's MFMA chains) instead of one 256-row workgroup per CU.  Measured below.
#ifndef FWN_TAIL128_TWO
#define FWN_TAIL128_TWO 0
#endif
#ifndef FWN_TAIL256_MIN
#define FWN_TAIL256_MIN (192 * 256)
#endif
#ifndef FWN_TAIL_SPLIT_MAX
#define FWN_TAIL_SPLIT_MAX 12288      // rows up to which the N-split tail (three ring GEMMs) replaces the fused tail
#endif
int fwn_tail_rows(int M) { return (M >= FWN_TAIL256_MIN && !FWN_TUNE(FWN_TAIL128_TWO, FWN_TAIL128_TWO)) ? 256 : 128; }   // rows per fused-tail workgroup
int fwn_tail_is_split(int M) { return M <= FWN_TUNE(FWN_TAIL_SPLIT_MAX, FWN_TAIL_SPLIT_MAX); }
// M <= FWN_TAIL_SPLIT_MAX: the skip sum as a ring GEMM that splits its weights over workgroups, then either (FWN_TAIL_SPLIT_CHAIN,
// default) tail_kernel<.., HAS_P1 = false> = final conv + ZeroConv + coupling in one launch of 64-row workgroups, or the round-2
// form (two more ring GEMMs).
#ifndef FWN_TAIL_SPLIT_CHAIN
#define FWN_TAIL_SPLIT_CHAIN 1
#endif
#ifndef FWN_TAIL_SPLIT_CHAIN_MIN
#define FWN_TAIL_SPLIT_CHAIN_MIN 6144    // fewer rows: too few 64-row workgroups to stream the final / ZeroConv weights through (in situ:
#endif                                   // block 4, 4032 rows, +2 us per flow; block 7, 504 rows, +17 us against the two ring GEMMs)
static bool tail_split_chain(int M) {
    return FWN_TUNE(FWN_TAIL_SPLIT_CHAIN, FWN_TAIL_SPLIT_CHAIN) != 0 && M >= FWN_TUNE(FWN_TAIL_SPLIT_CHAIN_MIN, FWN_TAIL_SPLIT_CHAIN_MIN);
}
static int tail_chain_rows(int M) { return fwn_tail_is_split(M) ? 64 : fwn_tail_rows(M); }
// rs_mt != 0: the register-streamed tail runs the launch (tail_rs.h: it can always write out_b elsewhere)
int fwn_tail_chain_xb_out(int M, int npt, int rs_mt) { return rs_mt != 0 || !fwn_tail_is_split(M) || tail_split_chain(M); }
int fwn_tail_chain_front(int M, int Ch, int npt, int rs_mt) { return fwn_tail_chain_xb_out(M, npt, rs_mt) && Ch <= 8 && npt == 1; }
// rs_mt: 32-row tiles per workgroup of the register-streamed tail (tail_rs.h) when that kernel runs the launch (fwn_tail_rs_mt), else 0
int fwn_tail_npartials_chain(int M, int Ch, int front, int rs_mt) {       // log-det partial slots one tail launch writes
    if (rs_mt == 0 && fwn_tail_is_split(M) && !tail_split_chain(M)) return ((M + 63) / 64) * 8;
    const int rw = (rs_mt ? 32 * rs_mt : tail_chain_rows(M)) - (front ? 2 : 0);      // chained front conv: tiles overlap by one row on either side
    return (M + rw - 1) / rw;
}
int fwn_tail_npartials(int M, int rs_mt) { return fwn_tail_npartials_chain(M, 0, 0, rs_mt); }

// process-wide developer option (fwn_set_option in api.hip; round 4 read two environment variables on every gate launch)
int g_fwn_opt_rs_persist = -1;
// compute units of the CURRENT device, cached per device (a process may drive several)
int fwn_device_cus() {
    static int cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cache[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 2) n = 256;
        cache[dev] = n;
    }
    return cache[dev];
}

static inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

template <class Prob>
static void launch_gemm128(const Prob& p, int M, int ntn, hipStream_t st) {
    // 128-row tiles when they still give >= 2 workgroups per CU, else 64-row tiles.
    const int t128 = (M + 127) / 128 * ntn;
    if (t128 >= 512) {
        hipLaunchKernelGGL((gemm128_kernel<2, Prob>), dim3(t128), dim3(256), 0, st, p, ntn);
    } else {
        const int t64 = (M + 63) / 64 * ntn;
        hipLaunchKernelGGL((gemm128_kernel<1, Prob>), dim3(t64), dim3(256), 0, st, p, ntn);
    }
}

// Tile choice for the ring GEMMs: the largest tile that still yields about one workgroup per CU
// (the chain of K-chunks inside a workgroup is serial, so fewer, fatter workgroups only pay while
// every CU has one).  N = 64 * ncol64 output columns.
#define RING_LAUNCH(BM, BN, WM, WN, BK, D) RING_LAUNCHK(BM, BN, WM, WN, BK, D, 1)
#define RING_LAUNCHK(BM, BN, WM, WN, BK, D, KSP)                                                    \
    hipLaunchKernelGGL((gemm_ring_kernel<BM, BN, WM, WN, BK, D, Prob, KSP>),                        \
                       dim3(((M + BM - 1) / BM) * (N / BN)), dim3(64 * WM * WN * KSP), 0, st, p, N / BN)
template <class Prob>
static void launch_ring(const Prob& p, int M, int N, int ksteps, hipStream_t st) {
    // 16-wave workgroups (4 waves per SIMD) hide the barrier / LDS latency of the K loop best
    // (tools/bench_gemm.hip): 256x256 reaches ~0.8 PF on the block-1 gate, 8-wave tiles ~0.7.
    const int FILL = 192;    // workgroups needed before a fatter tile pays (256 CUs)
    if (Prob::ALLOW_256 && N % 256 == 0 && ((M + 255) / 256) * (N / 256) >= FILL) {
        RING_LAUNCH(256, 256, 4, 4, 64, 2);
    } else if (((M + 255) / 256) * (N / 128) >= FILL) {
        RING_LAUNCH(256, 128, 8, 2, 64, 3);
    } else if (((M + 127) / 128) * (N / 128) >= FILL) {
        RING_LAUNCH(128, 128, 4, 2, 64, 3);
    } else if (((M + 63) / 64) * (N / 128) >= FILL) {
        RING_LAUNCH(64, 128, 2, 2, 64, 4);
    } else {
        // A launch this small is a latency chain: per K chunk every wave pays its DMA issues (~100+ cycles a piece), a
        // barrier and an LDS round trip for a handful of MFMAs.  128-wide chunks halve the number of round trips and
        // intra-workgroup split-K (4 wave groups taking alternate k-steps) spreads the DMA issues over 8 waves
        // (tools/bench_gemm.hip, K = 768: 8.1 -> 6.7 us); K must be a multiple of 128 for the wide chunks.
        const int mode = FWN_TUNE(FWN_SMALL_TILE, 2);
        if (ksteps % 8 == 0 && mode == 2) RING_LAUNCHK(64, 64, 2, 1, 128, 4, 4);
        else if (ksteps % 8 == 0 && mode == 1) RING_LAUNCHK(64, 64, 2, 1, 128, 4, 2);
        else RING_LAUNCHK(64, 64, 2, 1, 64, 4, 2);
    }
}

void fwn_launch_front(const float* xa, const float* an_a, const void* W, const void* W2, const float* bias,
                      void* hout, void* scratch, int M, int Ti, int Ch, int kpad, int apply_an, void* h8out, hipStream_t st) {
    if (W2 && (Ch == 32 || Ch == 64 || Ch == 128 || (Ch == 16 && !h8out)) && ((uintptr_t)xa & 15) == 0 && FWN_TUNE(FWN_FRONT_FUSED, 1)) {
        const int grid = ((M + 63) / 64) * 4;
        if (Ch == 16) hipLaunchKernelGGL((front_mfma_kernel<32, 16>), dim3(grid), dim3(256), 0, st, xa, an_a, apply_an, (const bf16*)W2, bias, (bf16*)hout, M, Ti);
        else if (Ch == 32) hipLaunchKernelGGL((front_mfma_kernel<32>), dim3(grid), dim3(256), 0, st, xa, an_a, apply_an, (const bf16*)W2, bias, (bf16*)hout, M, Ti);
        else if (Ch == 64) hipLaunchKernelGGL((front_mfma_kernel<64>), dim3(grid), dim3(256), 0, st, xa, an_a, apply_an, (const bf16*)W2, bias, (bf16*)hout, M, Ti);
        else hipLaunchKernelGGL((front_mfma_kernel<128>), dim3(grid), dim3(256), 0, st, xa, an_a, apply_an, (const bf16*)W2, bias, (bf16*)hout, M, Ti);
        return;
    }
    if (Ch >= 32 && W2 && scratch) {
        const long total = (long)M * Ch;
        const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
        hipLaunchKernelGGL(xprep_kernel, dim3(grid), dim3(256), 0, st, xa, an_a, M, Ch, apply_an, (bf16*)scratch);
        FrontRingProb rp{(const bf16*)scratch, (const bf16*)W2, bias, (bf16*)hout, M, Ti, Ch};
        launch_ring(rp, M, 256, 6 * Ch / 16, st);
        return;
    }
    FrontProb p{xa, an_a, (const bf16*)W, bias, (bf16*)hout, M, Ti, Ch, ilog2(Ch), kpad, apply_an};
    if (Ch <= 16) {
        p.h8out = (unsigned char*)h8out;
#define FRONT_VALU(R, KMAX, ROWS) hipLaunchKernelGGL((front_valu_kernel<R, KMAX>), dim3((M + ROWS - 1) / ROWS), dim3(256), 0, st, p)
        const bool big = M >= 64 * 512;
        if (Ch <= 2) { if (big) FRONT_VALU(16, 6, 64); else FRONT_VALU(4, 6, 16); }
        else if (Ch <= 4) { if (big) FRONT_VALU(16, 12, 64); else FRONT_VALU(4, 12, 16); }
        else if (Ch <= 8) { if (big) FRONT_VALU(16, 24, 64); else FRONT_VALU(4, 24, 16); }
        else { if (big) FRONT_VALU(16, 48, 64); else FRONT_VALU(4, 48, 16); }
#undef FRONT_VALU
        return;
    }
    launch_gemm128(p, M, 2, st);
}

// (the register-streamed gate - gate_rs.h - and its dispatch rule live in gate_rs.hip: fwn_launch_gate_rs)
void fwn_launch_gate(const void* h, const void* ca, const float* P, const void* Wd, const void* Wc, const void* Wgs,
                     const float* bias, void* o, int M, int Ti, int dil, int cin, int kcpad, void* aux,
                     hipStream_t st) {
    if (Wgs && fwn_gate_stream_ok(M, Ti, dil, cin, ca != nullptr && P == nullptr, aux != nullptr)) {
        fwn_launch_gate_rs(h, ca, Wgs, bias, o, M, Ti, dil, cin, st);
        return;
    }
    GateProb p{(const bf16*)h, (const bf16*)ca, P, (const bf16*)Wd, (const bf16*)Wc, bias, (bf16*)o,
               M, Ti, dil, cin, kcpad};
    p.aux = (bf16*)aux;
    const int t256 = (M + 255) / 256;
    if (dil <= FWN_HALO_MAXDIL && t256 * 4 >= 192) {
        // Tap-sharing tiles (gate_halo.h) for the MFMA/L2-bound sizes.  On warm caches they also win at
        // the small blocks (tools/bench_gemm.hip), but inside a pass every flow's weights arrive cold
        // from HBM and the deeper plain ring hides that better there (rocprof, in situ).
        if (t256 * 2 >= 192)
            hipLaunchKernelGGL((gate_halo_kernel<256, 256, GateProb>), dim3(t256 * 2), dim3(1024), 0, st, p, 2);
        else
            hipLaunchKernelGGL((gate_halo_kernel<256, 128, GateProb>), dim3(t256 * 4), dim3(1024), 0, st, p, 4);
        return;
    }
    // 128 x 128 tap-sharing tiles (8 waves, 66 KB of LDS: two workgroups per CU) where 256-row tiles would leave CUs empty
    // but 128-row tiles still fill the chip: block 3 of the 8-clip pass (8064 rows) - the ring tile there stages the
    // dilated taps three times (720 KB per workgroup against 593 KB) at one workgroup per CU
    const int t128 = (M + 127) / 128;
    if (dil <= FWN_HALO_MAXDIL && t128 * 4 >= 192 && FWN_TUNE(FWN_HALO128, 1)) {
        hipLaunchKernelGGL((gate_halo_kernel<128, 128, GateProb>), dim3(t128 * 4), dim3(512), 0, st, p, 4);
        return;
    }
    launch_ring(p, M, 512, (768 + (ca ? kcpad : 0)) / 16, st);
}

int fwn_gate_fp8_ok(int M, int dil) { return dil <= FWN_HALO_MAXDIL && ((M + 255) / 256) * 4 >= 192; }

void fwn_launch_gate_fp8(const void* h8, const void* ca, const void* Wd8, int wexp, const void* Wc, const float* bias, void* o,
                         int M, int Ti, int dil, int cin, int kcpad, hipStream_t st) {
    GateProb p{nullptr, (const bf16*)ca, nullptr, nullptr, (const bf16*)Wc, bias, (bf16*)o, M, Ti, dil, cin, kcpad};
    p.h8 = (const unsigned char*)h8;
    p.Wd8 = (const unsigned char*)Wd8;
    p.sb = 127 - wexp;
    const int t256 = (M + 255) / 256;
    if (t256 * 2 >= 192)
        hipLaunchKernelGGL((gate_halo_kernel<256, 256, GateProb, true>), dim3(t256 * 2), dim3(1024), 0, st, p, 2);
    else
        hipLaunchKernelGGL((gate_halo_kernel<256, 128, GateProb, true>), dim3(t256 * 4), dim3(1024), 0, st, p, 4);
}

void fwn_launch_res(const void* o, const void* hin, const void* W, const float* bias, void* hout, int M, void* h8out,
                    hipStream_t st) {
    ResProb p{(const bf16*)o, (const bf16*)hin, (const bf16*)W, bias, (bf16*)hout, M};
    p.h8out = (unsigned char*)h8out;
    // The large launches are HBM-bound (o read, h read, h write: 99 MB at block 0 of the 8-clip pass) and a workgroup runs its
    // phases one after the other - K loop (o chunks), residual loads, stores.  Two 64 KB workgroups per CU (128 x 128 tiles,
    // ring depth 2) put one workgroup's epilogue beside the other's K loop where ONE 147 KB workgroup per CU (256 x 128 tiles,
    // depth 3) left the memory pipe idle between phases: block 0 19.4 -> 17.6 us per launch, block 1 9.4 -> 9.3
    // (tools/probe/res_tiles.py, bit-identical; 128 x 128 at depth 3 / 4 x 32-wide chunks, 64 x 128 at depth 2 / 3 and
    // 256 x 128 at depth 2 are all slower than the old tile).
    if (((M + 255) / 256) * 2 >= 192 && FWN_TUNE(FWN_RES_TWO_PER_CU, 1)) {
        const int N = 256;
        typedef ResProb Prob;
        RING_LAUNCH(128, 128, 4, 2, 64, 2);
        return;
    }
    launch_ring(p, M, 256, 16, st);
}

// ---- hoisted conditioning: how many K splits, and the fixed-order sum of their partial outputs ----
// nz (flow, layer) matrices per launch.  With a workgroup or less per two CUs and at least 8 chunks per split, the K range
// is dealt over up to 8 workgroups (B = 1: M = 63 rows against K = 10240 at the last block).
// Round 6: 257 .. 512 rows against K >= 8192 (the last block of the 8-clip pass: 504 rows x 10240) take 256 x 128 tiles with the K
// range halved - 192 workgroups either way, but half the operand bytes per MFMA of the 128 x 128 tile: 109 -> 90 us
// (tools/probe/cond_bench.py; profiles/r06_notes.md section 7).
static bool cond_wide_split(int M, int kcpad) { return M > 256 && M <= 512 && kcpad >= 8192; }
int fwn_cond_nsplit(int M, int nz, int kcpad) {
    const int base = ((M + 63) / 64) * 4 * nz;
    if (FWN_TUNE(FWN_COND_WIDE, 1) && cond_wide_split(M, kcpad)) return 2;
    if (base >= FWN_TUNE(FWN_COND_BASE, 128)) return 1;
    int ns = 1;
    while (ns < 8 && base * ns * 2 <= FWN_TUNE(FWN_COND_FILL, 256) && kcpad / 64 / (ns * 2) >= 8) ns *= 2;
    return ns;
}
__global__ __launch_bounds__(256) void cond_reduce_kernel(float* __restrict__ P, const float* __restrict__ part, long part_stride,
                                                          int nsplit, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 a = ((const float4*)P)[i];
        for (int s = 0; s < nsplit - 1; ++s) {          // ascending split order: bit-reproducible
            const float4 b = ((const float4*)(part + (size_t)s * part_stride))[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        ((float4*)P)[i] = a;
    }
}
void fwn_launch_cond_reduce(float* P, const float* part, long part_stride, int nsplit, long n, hipStream_t st) {
    if (nsplit <= 1) return;
    const long n4 = n / 4;
    const long g = (n4 + 255) / 256;
    hipLaunchKernelGGL(cond_reduce_kernel, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(256), 0, st, P, part, part_stride, nsplit, n4);
}
// ---- tile choice for the hoisted conditioning ----
// These launches are operand streams (K = cin up to 10240): a workgroup pulls (BM + BN) K 2 bytes through its CU, one
// workgroup per CU, so a launch costs about rounds x (BM + BN) with rounds = ceil(workgroups / 256).  fwn_cond_tile picks
// the tile with the least of that for nz matrices (ties: the larger tile); tiles: 0 = 256 x 256, 1 = 256 x 128,
// 2 = 128 x 128, 3 = 64 x 128.
static const int kCondBM[4] = {256, 256, 128, 64}, kCondBN[4] = {256, 128, 128, 128};
static int cond_tile_wgs(int t, int M, int nz) { return ((M + kCondBM[t] - 1) / kCondBM[t]) * (512 / kCondBN[t]) * nz; }
static int cond_tile_cost(int t, int M, int nz) { return ((cond_tile_wgs(t, M, nz) + 255) / 256) * (kCondBM[t] + kCondBN[t]); }
static int fwn_cond_tile(int M, int nz, int* cost) {
    int best = 0, bc = cond_tile_cost(0, M, nz);
    for (int t = 1; t < 4; ++t) {
        const int c = cond_tile_cost(t, M, nz);
        if (c < bc) { bc = c; best = t; }
    }
    if (cost) *cost = bc;
    return best;
}
// Both parity groups of a block in ONE launch (ca_odd) when that is cheaper than one launch per group by the same
// measure, or when the K range is split (few rows: a launch latency less).
bool fwn_cond_merge(int M, int nz_group, int nsplit) {
    if (nsplit > 1) return true;
    int c1 = 0, c2 = 0;
    fwn_cond_tile(M, nz_group, &c1);
    fwn_cond_tile(M, 2 * nz_group, &c2);
    return c2 < 2 * c1;
}
// ca_odd != NULL: flows with an odd index read that plane (flow0 = 0, flow_step = 1, nflow = all flows of the block)
void fwn_launch_cond2(const void* ca, const void* ca_odd, const void* Wc_base, float* P_base, long w_stride, long p_stride,
                      int flow0, int flow_step, int nflow, int L, int M, int cin, int kcpad, float* part_base, long part_stride,
                      int nsplit, hipStream_t st) {
    CondBatch cb{(const bf16*)ca, (const bf16*)ca_odd, (const bf16*)Wc_base, P_base, w_stride, p_stride, flow0, flow_step, L,
                 M, cin, kcpad, part_base, part_stride, nsplit > 1 ? nsplit : 1};
    const int nz = nflow * L;                                    // matrices of this launch
    // a split K range: the smallest tile (few rows), 256 x 128 for the wide split above
    const int t = cb.nsplit > 1 ? (FWN_TUNE(FWN_COND_SPLIT_TILE, -1) >= 0 ? FWN_TUNE(FWN_COND_SPLIT_TILE, -1) : cond_wide_split(M, kcpad) ? 1 : 3)
                                : FWN_TUNE(FWN_COND_TILE, -1) >= 0 ? FWN_TUNE(FWN_COND_TILE, -1) : fwn_cond_tile(M, nz, nullptr);
    const dim3 grid(((M + kCondBM[t] - 1) / kCondBM[t]) * (512 / kCondBN[t]), nz, cb.nsplit);
    if (t == 0) hipLaunchKernelGGL((cond_batch_kernel<256, 256, 4, 4, 64, 2>), grid, dim3(1024), 0, st, cb, 2);
    else if (t == 1) hipLaunchKernelGGL((cond_batch_kernel<256, 128, 8, 2, 64, 3>), grid, dim3(1024), 0, st, cb, 4);
    else if (t == 2) hipLaunchKernelGGL((cond_batch_kernel<128, 128, 2, 2, 64, 3>), grid, dim3(256), 0, st, cb, 4);
    else hipLaunchKernelGGL((cond_batch_kernel<64, 128, 2, 2, 64, 4>), grid, dim3(256), 0, st, cb, 4);
}
void fwn_launch_cond(const void* ca, const void* Wc_base, float* P_base, long w_stride, long p_stride,
                     int flow0, int flow_step, int nflow, int L, int M, int cin, int kcpad, float* part_base, long part_stride,
                     int nsplit, hipStream_t st) {
    fwn_launch_cond2(ca, nullptr, Wc_base, P_base, w_stride, p_stride, flow0, flow_step, nflow, L, M, cin, kcpad, part_base, part_stride,
                     nsplit, st);
}

// ---- tail dispatch ----
// M > FWN_TAIL_SPLIT_MAX: the fused register-chained tail (tail_chain.h, HAS_P1), 256-row workgroups while those fill the
// chip, else 128-row ones; below: see FWN_TAIL_SPLIT_CHAIN above.

void fwn_launch_tail(const void* o, long o_stride, int L, const void* Ws, const float* bs, const void* Wf,
                     const float* bfin, const void* Wz, const float* bz, const float* ez, const float* an,
                     float* xa, float* xb, float* partial, int M, int Ch, int npt, int inverse, void* scratch_s,
                     void* scratch_u, const fwn_tail_chain* chain, const void* Wts, hipStream_t st) {
    TailArgs a{(const bf16*)o, (const bf16*)Ws, bs, (const bf16*)Wf, bfin, (const bf16*)Wz, bz, ez, an,
               xa, xb, partial, o_stride, L, M, Ch, npt, inverse};
    const bool front = chain && chain->h0_next;
    a.xb_out = chain ? chain->xb_out : nullptr;
    a.S = nullptr;
    a.h0_next = front ? (bf16*)chain->h0_next : nullptr;
    a.Wfn = front ? (const bf16*)chain->Wfn : nullptr;
    a.bfn = front ? chain->bfn : nullptr;
    a.an_next = front ? chain->an_next : nullptr;
    a.kfn = front ? chain->kfn : 0;
    a.Ti = front ? chain->Ti : 0;
    a.overlap = front ? 1 : 0;
    a.save_s = chain ? (bf16*)chain->save_s : nullptr;
    a.save_u = chain ? (bf16*)chain->save_u : nullptr;
    a.save_z = chain ? chain->save_z : nullptr;
    // the register-streamed tail (tail_rs.h) wherever the flow's fragment stream is packed and the shape is one of its
    if (const int mt = fwn_tail_rs_mt(M, L, Ch, npt, Wts != nullptr)) {
        fwn_launch_tail_rs(a, Wts, mt, st);
        return;
    }
#define TAIL_LAUNCH(NW, D, BK1, WDB, NPT, P1, FRONT)                                                                  \
    hipLaunchKernelGGL((tail_kernel<NW, D, BK1, WDB, NPT, P1, FRONT>),                                                 \
                       dim3((M + 32 * NW - (FRONT ? 2 : 0) - 1) / (32 * NW - (FRONT ? 2 : 0))), dim3(64 * NW), 0, st, a)
#define TAIL_BY_NPT(NW, D, BK1, WDB, P1)                                                                              \
    do {                                                                                                              \
        if (front) TAIL_LAUNCH(NW, D, BK1, WDB, 1, P1, true);                                                         \
        else if (npt == 1) TAIL_LAUNCH(NW, D, BK1, WDB, 1, P1, false);                                                \
        else if (npt == 2) TAIL_LAUNCH(NW, D, BK1, WDB, 2, P1, false);                                                \
        else TAIL_LAUNCH(NW, D, BK1, WDB, 4, P1, false);                                                              \
    } while (0)
    if (fwn_tail_is_split(M)) {          // scratch_s / scratch_u: [M][256] bf16 each (api.hip checks they are there)
        bf16* S = a.save_s ? a.save_s : (bf16*)scratch_s;      // training keeps S and U: they are written where it wants them
        bf16* U = a.save_u ? a.save_u : (bf16*)scratch_u;
        a.save_s = nullptr;                                    // (S comes from the ring GEMM, not from the chained kernel)
        TailLinProb p1{(const bf16*)o, (const bf16*)Ws, bs, S, o_stride, L, M};
        launch_ring(p1, M, 256, L * 16, st);
        if (tail_split_chain(M)) {
            a.S = S;
            TAIL_BY_NPT(2, 4, 64, true, false);
            return;
        }
        TailLinProb p2{S, (const bf16*)Wf, bfin, U, 0, 1, M};
        launch_ring(p2, M, 256, 16, st);
        TailZeroProb p3{U, (const bf16*)Wz, bz, ez, an, xa, xb, partial, M, Ch, npt, inverse, a.save_z};
        hipLaunchKernelGGL((gemm_ring_kernel<64, 64, 2, 1, 128, 3, TailZeroProb, 2>), dim3(((M + 63) / 64) * npt), dim3(256), 0,
                           st, p3, npt);
        return;
    }
    if (fwn_tail_rows(M) == 256) TAIL_BY_NPT(8, 4, 32, false, true);
    else if (M >= FWN_TAIL256_MIN) TAIL_BY_NPT(4, 2, 32, true, true);
    else TAIL_BY_NPT(4, 3, 64, true, true);
#undef TAIL_BY_NPT
#undef TAIL_LAUNCH
}
