/*
 * fwn.h - C ABI of libfwn.so: the MI355X (gfx950) FloWaveNet flow forward / inverse path.
 *
 * The reference (ryhorv/tf-flowavenet) has no FFI seam: its hot path sits behind the Python
 * class FloWaveNet (model.py:282-404) whose arithmetic is delegated to TensorFlow 1.12 ops.
 * This header is the boundary a maintainer binds instead of those TF ops (ctypes stub in
 * INTEGRATION.md).  Conventions:
 *   - every pointer is a DEVICE pointer into caller-owned memory unless marked "host";
 *   - sizes are explicit, the HIP stream is the last argument (void*, may be NULL = default);
 *   - no allocation, no synchronisation, no hidden global state: calls are asynchronous on
 *     `stream` and safe to issue concurrently on distinct streams with distinct workspaces;
 *   - return 0 on success, <0 on error; fwn_last_error() returns a thread-local message.
 *
 * Device data layouts (DESIGN.md "Data layout in HBM"):
 *   flow state   : fp32 planes[2][B][T/2]; plane q holds samples 2s+q.  At block i a plane is the
 *                  row-major matrix [M = B*T_i][Ch = 2^i], T_i = T / 2^(i+1).  The reference's
 *                  squeeze / unsqueeze (model.py:226-239,260-273) and change_order (:166-174)
 *                  are index bookkeeping on these planes: no data moves.
 *   conditioning : bf16 cplanes[2][B][T][num_mels/2]; plane q holds mel bins [q*half, (q+1)*half).
 *   hidden       : bf16 [M][256].
 *   weights      : bf16, weight-norm folded, [N][K] with K contiguous, permuted to the plane
 *                  orders above (tf-flowavenet_amd/packing.py builds the index tables).
 */
#ifndef FWN_H
#define FWN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FWN_VERSION 321            /* 0.3.21 (round 6): + fwn_model_desc.cond_stream (appended) and fwn_cond_stream* / fwn_pack_cond_stream (the register-streamed
                                    * conditioning projection, csrc/cond_rs.h; additive); fwn_pack_tail_stream_jobs.  0.3.20 (round 6): + fwn_flow_desc.Wts and fwn_tail_stream_bytes / fwn_pack_tail_stream / fwn_tail_stream_rows (the
                                    * register-streamed tail, csrc/tail_rs.h; additive: a 0.3.10 host that zero-fills its descriptors keeps working);
                                    * fwn_tail_partials / fwn_tail_partials_chained are upper bounds now.  0.3.10 (round 5): + fwn_flow_run_persist / fwn_flow_persist_* (one launch per small-M flow), fwn_model_desc.persist_mode
                                    * (was `reserved`: 0 keeps working), fwn_set_option.  0.3.1: + fwn_gate_clock (additive: a 0.3.0 host keeps working).  0.3.0: fwn_flow_desc gained Wfront3 / kf3 (round 3) and Wgs (round 4), fwn_model_desc
                                    * chain_mode, fwn_block_done_fn returns int; Wskip / Wfinal rows and biases are in
                                    * acc_k_perm order, Wzero's K axis is natural.  A host built against 0.2.0 must be
                                    * rebuilt and repack its weights. */
#define FWN_MAX_LAYERS 8
#define FWN_MAX_UPSAMPLE 4

#define FWN_OK 0
#define FWN_ERR_ARG (-1)           /* bad argument (null pointer, size, alignment, config) */
#define FWN_ERR_HIP (-2)           /* a HIP launch or runtime call failed */
#define FWN_ERR_WORKSPACE (-3)     /* workspace too small */
#define FWN_ERR_CALLBACK (-4)      /* a host callback (fwn_block_done_fn) asked to stop */

int fwn_version(void);
const char* fwn_last_error(void);

/* ---- K10: weight-norm + bf16 packing (replaces convolutional.py:73-80 + utils.py:19-29) ----
 * v is the TF-layout kernel flattened to [k_src][n_src] (k_src = kernel_size*C_in, n fastest).
 * scale[n] = g[n] / sqrt(max(sum_k v[k][n]^2, 1e-12)). */
int fwn_wn_scale(const float* v, const float* g, int k_src, int n_src, float* scale, void* stream);
/* out[n'*ld_dst + k'] = bf16(v[src_k[k']][src_n[n']] * scale[src_n[n']]) for n' < n_dst, k' < k_dst.
 * src_k[k'] < 0 writes 0 (K padding); rows with src_n[n'] < 0 are left untouched; scale may be NULL. */
int fwn_pack_bf16(const float* v, const float* scale, const int32_t* src_k, const int32_t* src_n,
                  int n_src, int k_dst, int n_dst, int64_t ld_dst, void* out_bf16, void* stream);

/* Grouped form: a whole model's scales and packed copies in two launches.  Both job tables live in
 * device memory; scale job s writes scales[s][0..n_src) (g / ||V||_col), pack job p multiplies by
 * scales[p.scale_slot] (slot < 0: none) and by p.mul; transposed: out[k][n] (ld_dst >= n_dst)
 * instead of out[n][k].  Index tables as in fwn_pack_bf16. */
typedef struct fwn_scale_job { const float* v; const float* g; int32_t k_src, n_src; } fwn_scale_job;
typedef struct fwn_pack_job {
    const float* v; const int32_t* src_k; const int32_t* src_n; void* out; int64_t ld_dst;
    int32_t n_src, k_dst, n_dst, scale_slot, transposed; float mul;
} fwn_pack_job;
int fwn_pack_jobs(const fwn_scale_job* scale_jobs, int n_scale_jobs, const fwn_pack_job* pack_jobs, int n_pack_jobs,
                  float* scales, int scale_ld, void* stream);

/* ---- small fp32 parameter tables straight from a flat vector of fp32 masters (a training step refreshes biases,
 * ActNorm / ZeroConv tables and the up-sampling kernels on the device, inside its hipGraph):
 * fwn_gather_tables: out[i] = F(post[i] * sum_t flat[idx[t*total + i]]) for i < total (idx < 0: term absent);
 *                    mode[i] 0: identity, 1: exp, both in fp64 rounded once; 2 / 3: the same in fp32, terms in order
 * fwn_sum_f32      : out[0] = sum_i in[i] (fixed order, fp64 accumulation)
 * fwn_upsample_wn  : the weight-normed up-sampling kernel out[2s][3] = v / ||v||_(k per kw column) * g (convolutional.py:179-186) */
int fwn_gather_tables(const float* flat, const int64_t* idx, int nterm, int64_t total, const double* post,
                      const unsigned char* mode, float* out, void* stream);
int fwn_sum_f32(const float* in, int64_t n, float* out, void* stream);
int fwn_upsample_wn(const float* v, const float* g, int s, float* out, void* stream);

/* ---- K1: one upsampling stage (replaces Conv2DTranspose.call + leaky_relu, model.py:301-311,
 * 398-404).  in [B][H][W] fp32, wk = weight-normed kernel [2s][3] fp32 (device), out rows H*s.
 * Exactly one of out_f32 ([B][H*s][W]) / out_cplanes (bf16 [2][B][H*s][W/2]) may be NULL. */
int fwn_upsample_stage(const float* in, int B, int H, int W, const float* wk, float bias, int s,
                       float* out_f32, void* out_cplanes, void* stream);
/* The same stage with the bias read from device memory (one float, e.g. the fp32 master itself): the launch
 * carries no parameter value and can be replayed from a hipGraph after the parameter changed. */
int fwn_upsample_stage_dev(const float* in, int B, int H, int W, const float* wk, const float* bias, int s,
                           float* out_f32, void* out_cplanes, void* stream);

/* ---- K2: x[B][T] <-> planes[2][B][T/2] (squeeze / unsqueeze entry and exit) ---- */
int fwn_split_planes(const float* x, int64_t B, int64_t T, float* planes, void* stream);
int fwn_merge_planes(const float* planes, int64_t B, int64_t T, float* x, void* stream);

/* ---- K3 (init only): ActNorm data-dependent init of one flow (model.py:30-83).
 * an[2][4][Ch] <- per (a|b) plane: shift b, scale exp(3 logs), inverse scale, 3*logs. */
int fwn_actnorm_ddi(const float* xa, const float* xb, int M, int Ch, float* an, void* stream);
/* The same init for a batch sharded over ranks (the reference's towers race on this assign, model.py:39 under
 * train.py:43-57): fwn_actnorm_moments writes this rank's mom[4 Ch + 1] doubles = per plane (sum_m x | sum_m x^2)
 * [2][2][Ch] followed by the row count M; the caller all-reduces (sums) them over the ranks;
 * fwn_actnorm_from_moments turns the global moments into the table `an` - identical on every rank. */
int fwn_actnorm_moments(const float* xa, const float* xb, int M, int Ch, double* mom, void* stream);
int fwn_actnorm_from_moments(const double* mom, int Ch, float* an, void* stream);

/* Static description of one flow (model.py:176-205 Flow = ActNorm + AffineCoupling(WaveNet)).
 * All pointers are device pointers to packed weights; layouts in packing.py. */
typedef struct fwn_flow_desc {
    int32_t Ch;          /* channels per plane at this block = 2^i */
    int32_t cin;         /* conditioning channels of c_a = (num_mels/2) * 2^(i+1) */
    int32_t kcpad;       /* cin rounded up to 64 */
    int32_t kfpad;       /* 3*Ch rounded up to 64 */
    int32_t npt;         /* ZeroConv pair tiles = max(1, ceil(Ch/32)) */
    int32_t L;           /* n_layer */
    const void* Wfront;  const float* bfront;               /* [256][kfpad], [256]            */
    const void* Wfront2;                                    /* Ch >= 32: [256][6*Ch], K = tap*2Ch + (hi|lo)*Ch + tau; Ch = 16: the same
                                                               with 32 channels per half, tau >= 16 zero ([256][192]); else NULL */
    /* Gate operands are stored pre-multiplied by the exponent scale of their nonlinearity: filter
     * rows by -2*log2(e), gate rows by -log2(e) (tanh(f)*sigmoid(g) from two exp2, packing.py GATE_MUL). */
    const void* Wd[FWN_MAX_LAYERS];                         /* [512][768] gate-packed rows     */
    const void* Wc[FWN_MAX_LAYERS];                         /* [512][kcpad]                    */
    const float* bgate[FWN_MAX_LAYERS];                     /* [512] conv bias + cond bias     */
    const void* Wres[FWN_MAX_LAYERS];                       /* [256][256] (layers 0..L-2)      */
    const float* bres[FWN_MAX_LAYERS];
    /* Wskip / Wfinal: row n' holds output channel n' with bits 2 and 3 swapped (packing.acc_k_perm: the order the tail
     * kernel's accumulator registers feed the next MFMA chain in), biases alike; every K axis in natural channel order */
    const void* Wskip;   const float* bskip;                /* [256][L*256], [256] (sum)       */
    const void* Wfinal;  const float* bfinal;               /* [256][256], [256]               */
    const void* Wzero;   const float* bzero; const float* ezero;   /* [npt*64][256], [npt*64]x2 */
    float* an;                                              /* [2][4][Ch] ActNorm (DDI writes)  */
    /* fp8 dilated-conv path (BASELINE configs[4]); NULL = not packed.  Wd8[l]: the gate-packed rows of Wd[l] as OCP
     * e4m3 bytes [512][768], stored as W * 2^wd8_exp[l] (one power-of-two scale per matrix, fwn_pack_e4m3). */
    const void* Wd8[FWN_MAX_LAYERS];
    int32_t wd8_exp[FWN_MAX_LAYERS];
    /* Chained front conv (Ch <= 8, else NULL / 0): the front weights once more as [256][kf3], K = (tap*Ch + tau)*2 + (hi|lo),
     * kf3 = 6*Ch rounded up to 16.  With it the whole-model calls let the tail of the PREVIOUS flow of the block compute
     * this flow's h0 (csrc/tail_chain.h); the stage entry points ignore it. */
    const void* Wfront3;
    int32_t kf3, reserved;
    /* Register-streamed gate (csrc/gate_rs.h; NULL = not packed): Wd[l] and Wc[l] once more in MFMA-fragment order
     * (fwn_pack_gate_stream), read by the gate kernel of the largest row counts (fwn_gate_stream_rows) instead of them. */
    const void* Wgs[FWN_MAX_LAYERS];
    /* Register-streamed tail (csrc/tail_rs.h; NULL = not packed): Wskip | Wfinal once more in MFMA-fragment order
     * (fwn_pack_tail_stream), read by the tail kernel of the larger row counts (fwn_tail_stream_rows) instead of them. */
    const void* Wts;
} fwn_flow_desc;

/* ---- fragment-order tail weights (round 6, csrc/tail_rs.h; replaces nothing in the reference: a second packing of the
 * operands of modules.py:175-179 for the kernel that streams them to registers).
 * fwn_tail_stream_bytes: size of the stream of one flow with L layers, 0 if no kernel is built for that L (L = 2 only);
 * fwn_pack_tail_stream : Wskip [256][L*256], Wfinal [256][256] (rows in accumulator order, as in fwn_flow_desc) -> out;
 * fwn_tail_stream_rows : smallest M from which fwn_tail / fwn_tail_chained / fwn_tail_train and the flow and model calls
 *                        use the stream (one ZeroConv pair tile: Ch <= 32). */
int64_t fwn_tail_stream_bytes(int L);
int fwn_pack_tail_stream(const void* Wskip, const void* Wfinal, int L, void* out, void* stream);
/* The same for njobs flows in one launch: `jobs` is a DEVICE array of {Wskip, Wfinal, out} device pointers (a training step
 * re-packs every flow's stream behind its grouped weight packing: packing.PackPlan). */
typedef struct fwn_tail_stream_job { const void* Wskip; const void* Wfinal; void* out; } fwn_tail_stream_job;
int fwn_pack_tail_stream_jobs(const fwn_tail_stream_job* jobs, int njobs, int L, void* stream);
int fwn_tail_stream_rows(void);

/* ---- fragment-order gate weights (round 4, csrc/gate_rs.h; replaces nothing in the reference: a second packing of the
 * operands of modules.py:113-124 for the kernel that streams them to registers).
 * fwn_gate_stream_bytes: size of the stream for `cin` conditioning channels, 0 if no kernel is built for that cin;
 * fwn_pack_gate_stream : Wd [512][768], Wc [512][kcpad] (gate-packed rows, as in fwn_flow_desc) -> out;
 * fwn_gate_stream_rows : smallest M from which fwn_gate / the flow and model calls use the stream (dilation <= 3,
 *                        Ti >= 256, conditioning fused: ca given, no training aux). */
int64_t fwn_gate_stream_bytes(int cin);
int fwn_pack_gate_stream(const void* Wd, const void* Wc, int cin, int kcpad, void* out, void* stream);
int fwn_gate_stream_rows(void);
/* Diagnostic (bench.py's roofline.clock_ghz): ONE launch of the register-streamed gate of `layer` exactly as fwn_gate would
 * run it at this shape (256-row tiles: M >= 24 576 rows, Wgs set), from an instantiation that also stamps the shader clock
 * (s_memtime) and the 100 MHz reference (s_memrealtime) at the start and end of every wave:
 * stamps[(workgroup * 8 + wave) * 4 + {0, 1, 2, 3}] = {cycles at start, at end, reference at start, at end}; returns the
 * number of workgroups (stamps holds 32 * that many uint64_t) or a negative error code if the shape has no such kernel.
 * In-kernel clock of a wave = (s[1] - s[0]) / (s[3] - s[2]) x 100 MHz; take it after seconds of back-to-back launches. */
int fwn_gate_clock(const fwn_flow_desc* d, int layer, const void* h, const void* ca, void* o, int M, int Ti, uint64_t* stamps,
                   void* stream);

/* ---- stage entry points (K4..K8), exposed so each kernel can be parity-tested alone ---- */
/* K4 front conv k=3 + ReLU over in_a (modules.py:144,164-165); apply_an: ActNorm on load.
 * scratch: NULL or [M][256] bf16 (for Ch >= 32 the fp32 state is first laid out as a hi|lo bf16
 * matrix there so the conv runs on the LDS-DMA ring GEMM; fwn_flow_run lends its second h buffer). */
int fwn_front(const fwn_flow_desc* d, const float* xa, void* h_out, void* scratch, int M, int Ti, int apply_an,
              void* stream);
/* K5 gated dilated layer `layer` (modules.py:113-124).  ca==NULL uses P (precomputed c_a@Wc). */
int fwn_gate(const fwn_flow_desc* d, int layer, const void* h, const void* ca, const float* P, void* o,
             int M, int Ti, void* stream);
/* K5 with the dilated taps in fp8 (v_mfma_scale_f32_32x32x64_f8f6f4): h8 = e4m3 copy of h [M][256] bytes
 * (fwn_cast_e4m3, or written by the front / res kernels inside fwn_flow_run_fp8), weights d->Wd8[layer]; the 1x1
 * conditioning conv stays bf16 in the same accumulators.  Exists for the MFMA-bound shapes only:
 * fwn_gate_fp8_supported(M, layer) != 0 (M >= 12288 rows, dilation <= 3); other shapes return FWN_ERR_ARG. */
int fwn_gate_fp8_supported(int M, int layer);
int fwn_gate_fp8(const fwn_flow_desc* d, int layer, const void* h8, const void* ca, void* o, int M, int Ti, void* stream);
/* bf16 [n] -> OCP e4m3 [n] (round to nearest even, saturating at +-448). */
int fwn_cast_e4m3(const void* src_bf16, void* dst_u8, int64_t n, void* stream);
/* e4m3 weight packing: fwn_wn_absmax folds max |v[k][n] scale[n] mul| of one source kernel into *amax (a device
 * float, zero it first; scale may be NULL); fwn_pack_e4m3 then writes out[n'*ld_dst + k'] = e4m3(v[src_k[k']][src_n[n']]
 * scale mul 2^e) with e = floor(log2(448 / *amax)) and stores e in *exp_out (device int32).  Index tables as in
 * fwn_pack_bf16 (several source kernels may share one output matrix and one amax: filter and gate rows). */
int fwn_wn_absmax(const float* v, const float* scale, int k_src, int n_src, float mul, float* amax, void* stream);
int fwn_pack_e4m3(const float* v, const float* scale, const int32_t* src_k, const int32_t* src_n, int n_src, int k_dst,
                  int n_dst, int64_t ld_dst, float mul, const float* amax, void* out_u8, int32_t* exp_out, void* stream);
/* K6 residual 1x1 (modules.py:126-128): h_out = (h_in + res_conv(o)) * sqrt(0.5). */
int fwn_res(const fwn_flow_desc* d, int layer, const void* o, const void* h_in, void* h_out, int M,
            void* stream);
/* K5' conditioning projections of a parity group of flows in one launch: for j = flow0, flow0 +
 * flow_step, ... (nflow flows) and every layer: P[j*L+l] = ca @ Wc[j][l].  Wc_base/w_stride address
 * the block-contiguous weights, P_base/p_stride the outputs ([M][512] fp32 each). */
int fwn_cond(const void* ca, const void* Wc_base, float* P_base, int64_t w_stride, int64_t p_stride,
             int flow0, int flow_step, int nflow, int L, int M, int cin, int kcpad, void* stream);
/* The same with the K range dealt over nsplit workgroups per tile (few rows against a long K: M = 63 rows, cin = 10240
 * at the last block of a single clip): split 0 writes P, split z > 0 the same matrices into part + (z - 1) part_stride
 * (floats; laid out like P); fwn_cond_reduce then adds the partials to P[0..n) in ascending order (bit-reproducible).
 * fwn_cond_splits: the split count the model-level calls use for nz = nflow * L matrices per launch (1: none). */
int fwn_cond_splits(int M, int nz, int kcpad);
int fwn_cond_split(const void* ca, const void* Wc_base, float* P_base, int64_t w_stride, int64_t p_stride,
                   int flow0, int flow_step, int nflow, int L, int M, int cin, int kcpad, float* part, int64_t part_stride,
                   int nsplit, void* stream);
int fwn_cond_reduce(float* P, const float* part, int64_t part_stride, int nsplit, int64_t n, void* stream);
/* ---- fragment-order conditioning weights (round 6, csrc/cond_rs.h; replaces nothing in the reference: a second packing of the
 * same [512][kcpad] matrices) for the hoisted projection from fwn_cond_stream_rows() rows on: a workgroup = 128 rows x one whole
 * matrix, weights streamed to registers, activations staged once in LDS (63 GFLOP of block 4 - 7 of the 8-clip pass in 55 - 65 us
 * against 76 - 93 for the ring tiles).
 * fwn_cond_stream_bytes : size of ONE matrix's stream (= the matrix: 512 kcpad 2 bytes), 0 if kcpad is not a multiple of 64
 * fwn_pack_cond_stream  : nz matrices Wc_base + z w_stride ([512][kcpad] bf16 each) -> out (nz streams back to back)
 * fwn_cond_stream       : P[j L + l] = ca @ Wc[j][l] for the nflow flows and L layers of a block in one launch (flows with an odd
 *                         index read ca_odd if given); nsplit > 1: like fwn_cond_split, then fwn_cond_reduce
 * fwn_cond_stream_splits: the split count the model-level calls use */
int64_t fwn_cond_stream_bytes(int kcpad);
int fwn_cond_stream_rows(void);
int fwn_cond_stream_splits(int M, int nz, int kcpad);
int fwn_pack_cond_stream(const void* Wc_base, int64_t w_stride, int kcpad, int nz, void* out, void* stream);
int fwn_cond_stream(const void* ca, const void* ca_odd, const void* Ws, float* P, int nflow, int L, int M, int cin, int kcpad,
                    float* part, int64_t part_stride, int nsplit, void* stream);
/* K6'+K7+K8 tail: skip sum, final 1x1, ZeroConv1d, affine coupling, ActNorm, log-det partials
 * (modules.py:175-180,51-56; model.py:86-102,124-141,146-161).  o = [L][M][256].
 * partial (forward only, may be NULL) receives AT MOST fwn_tail_partials(M) partial sums (how many depends on the kernel
 * that serves the flow's packed operands at this M: zero the buffer first, its sum is the log-det).
 * Large M: one register-chained kernel.  M <= 12288 rows: three GEMM launches whose weights are split over the
 * workgroups by output column (a workgroup of the fused kernel streams all 0.4 - 0.5 MB of tail weights itself:
 * 22 - 36 us whatever the row count); they keep S and U in `scratch` = [2][M][256] bf16 (may be NULL for larger M). */
int fwn_tail_partials(int M);
/* The exact count for flow d at M rows (which kernel serves it depends on its packed operands: d->Wts): mode -1 = fwn_tail /
 * fwn_tail_train, 0 = fwn_tail_chained without a next flow, 1 = with one.  Always <= the bounds above. */
int fwn_tail_partials_desc(const fwn_flow_desc* d, int M, int mode);
int fwn_tail(const fwn_flow_desc* d, const void* o, float* xa, float* xb, float* partial, int M,
             int inverse, void* scratch, void* stream);
/* The tail as the whole-model calls chain it (csrc/tail_chain.h), exposed for stage-level parity tests: out_b goes to
 * xb_out (a third plane buffer: xb is left untouched) and, when `next` is given (the flow that runs after d: Ch <= 8,
 * next->Wfront3 packed), next's front conv + ReLU (modules.py:144,164-165; forward: behind next's ActNorm) is computed in
 * the same launch into h0_next [M][256] bf16.  partial (forward) receives fwn_tail_partials_chained(M, d->Ch, next != NULL)
 * sums.  scratch as in fwn_tail.  Returns FWN_ERR_ARG where the tail at this shape cannot chain (fwn_tail_can_chain). */
int fwn_tail_can_chain(const fwn_flow_desc* d, int M, int with_front);
int fwn_tail_partials_chained(int M, int Ch, int with_front);
int fwn_tail_chained(const fwn_flow_desc* d, const fwn_flow_desc* next, const void* o, float* xa, const float* xb, float* xb_out,
                     void* h0_next, float* partial, int M, int Ti, int inverse, void* scratch, void* stream);
/* The same tail as the training step runs it (forward direction): it also keeps what the backward needs -
 * save_s = S = ReLU(skip sum) and save_u = U = ReLU(final conv), bf16 [M][256] in natural channel order, and
 * save_z = Z = U Wz + bz, fp32 [M][2 Ch] (log_s channels then t channels, plane order; before the exp(3 scale) factor).
 * o: layer l at o + l * o_stride elements (the training step keeps every layer's output in a buffer of its own). */
int fwn_tail_train(const fwn_flow_desc* d, const void* o, int64_t o_stride, float* xa, float* xb, float* partial, int M,
                   void* save_s, void* save_u, float* save_z, void* stream);

/* ---- one whole flow (replaces Flow.forward / Flow.reverse, model.py:185-202) ----
 * xa / xb: the planes holding in_a / in_b for this flow's swap parity, ca the matching
 * conditioning plane.  h0/h1: [M][256] bf16 scratch, o: [L][M][256] bf16 scratch,
 * P: NULL or [L][M][512] fp32 precomputed conditioning projections for this flow,
 * partial: NULL or [fwn_tail_partials(M)] log-det partial sums (forward).  ddi: run fwn_actnorm_ddi first. */
int fwn_flow_run(const fwn_flow_desc* d, int64_t B, int64_t T, float* xa, float* xb, const void* ca,
                 void* h0, void* h1, void* o, const float* P, float* partial, int inverse, int ddi,
                 void* stream);

/* ---- one whole flow as ONE launch (round 5, csrc/flow_persist.h): the same arithmetic as fwn_flow_run, bit for bit, for the
 * shapes where every stage of the flow is a launch of a handful of workgroups (the dependent-launch floor of the small
 * blocks: model.py:394-396 run one utterance at a time, synthesize.py:40-49).  Tickets (stage, 64-row tile, 64-column
 * tile) are taken from an atomic counter by 8-wave workgroups, dependencies are per row tile, a ticket's weights are
 * requested before it waits for its producers; hand-offs are write-through stores + agent-scope counters + sc1 loads.
 * Preconditions (fwn_flow_persist_supported != 0): conditioning hoisted (P given, ca == NULL), n_layer <= 2,
 * M = B * T / (2 Ch) <= 4096 rows, forward or inverse without data-dependent init, bf16 gates.
 * sync: fwn_flow_persist_sync_bytes(M, L) bytes of device memory that the CALLER ZEROES (stream-ordered) before every call;
 * sync[1] != 0 afterwards = a bounded spin gave up (fwn_flow_persist_status reads it back: that one synchronises); the flow's
 * outputs are then NaN (plane elements and log-det partials of the row tiles down the chain), also inside the whole-model
 * calls: log_p / logdet / the waveform come back NaN, never silently wrong. */
int fwn_flow_persist_supported(const fwn_flow_desc* d, int64_t B, int64_t T);
int64_t fwn_flow_persist_sync_bytes(int M, int L);
int fwn_flow_run_persist(const fwn_flow_desc* d, int64_t B, int64_t T, float* xa, float* xb, void* h0, void* h1, void* o,
                         const float* P, float* partial, int inverse, void* sync, void* stream);
int fwn_flow_persist_status(const void* sync, void* stream);
/* ---- process-wide developer options (replaces the environment variables the launch path read in round 4) ----
 * name: "rs_persist" (-1 auto, 0 / 1: the register-streamed gate's persistent form off / on); "persist_spin_us" (bound of the
 * one-launch flow's dependency spins in microseconds, 0 = the default 2 s: tests shorten it to provoke a give-up).  Returns the previous value,
 * or FWN_ERR_ARG for an unknown name.  (Round 4's second switch, the experimental co-resident gate, left the library:
 * tools/gate_co.h + tools/bench_gate_co.hip.) */
int fwn_set_option(const char* name, int value);

/* The same flow with the gated layers' dilated taps in fp8 wherever fwn_gate_fp8_supported says so (other layers /
 * shapes run the bf16 kernels): h8a / h8b are [M][256]-byte scratch buffers for the e4m3 copies of h. */
int fwn_flow_run_fp8(const fwn_flow_desc* d, int64_t B, int64_t T, float* xa, float* xb, const void* ca,
                     void* h0, void* h1, void* o, const float* P, float* partial, int inverse, int ddi,
                     void* h8a, void* h8b, void* stream);

/* ---- mel front-end (preprocessing.py:58-69; librosa.feature.melspectrogram semantics) ----
 * wav [B][T] fp32 -> mel [B][1 + T/hop][n_mels] in [0, 1]: centred STFT (reflect padding, `window`
 * [n_fft], power 2), `fb` [n_mels][n_fft/2 + 1] filterbank, 20 log10(max(1e-4, .)) - ref_level_db,
 * then clip((. - min_level_db) / -min_level_db, 0, 1).  n_fft a power of two, T > n_fft / 2. */
int fwn_mel_spectrogram(const float* wav, int64_t B, int64_t T, const float* window, const float* fb, int n_fft,
                        int hop, int n_mels, float ref_level_db, float min_level_db, float* mel, void* stream);

/* ---- K9: prior + log-det finalisation (model.py:342-347) ----
 * out2[0] = mean(0.5(-log 2pi - z^2)) over n = B*T plane elements, out2[1] = sum(partial)/(B*T). */
int fwn_prior_logp(const float* planes, int64_t n, const float* partial, int n_partial, float* out2,
                   void* stream);

/* ---- data-parallel optimiser step (replaces average_gradients + un-scale + clip_by_global_norm
 * + AdamOptimizer.apply_gradients, utils.py:34-60, train.py:27-32,75-81).  g is the flat fp32
 * gradient buffer AFTER the RCCL all-reduce (a sum over ranks); gscale = 1/(world * loss_scale)
 * turns it into the reference's averaged, un-scaled gradient.
 * fwn_grad_norm: gnorm_out[0] = ||g * gscale||_2 (deterministic; partial holds
 * fwn_grad_norm_partials(n) doubles).  fwn_clip_adam: g' = g*gscale * clip / max(gnorm, clip),
 * then TF-form Adam (lr_t = lr*sqrt(1-b2^t)/(1-b1^t), eps outside the sqrt) on fp32 masters. */
int fwn_grad_norm_partials(int64_t n);
int fwn_grad_norm(const float* g, int64_t n, float gscale, double* partial, float* gnorm_out, void* stream);
int fwn_clip_adam(float* w, const float* g, float* m, float* v, int64_t n, const float* gnorm, float gscale,
                  float clip, float lr, int64_t step, float beta1, float beta2, float eps, void* stream);
/* The same update with the bias-corrected rate lr_t = fwn_adam_rate(lr, step, b1, b2) read from DEVICE memory
 * (one float), so that the launch can be recorded in a hipGraph and replayed while the rate changes. */
double fwn_adam_rate(float lr, int64_t step, float beta1, float beta2);
int fwn_clip_adam_dev(float* w, const float* g, float* m, float* v, int64_t n, const float* gnorm, float gscale,
                      float clip, const float* lr_t, float beta1, float beta2, float eps, void* stream);

/* ---- training-side primitives (work in progress: the backward pass) ----
 * Generic multi-segment GEMM on the LDS-DMA ring core:
 *   Y[M][N] = oscale * relu?( mask?( sum_s shift_s(X_s)[M][k_s] . W[N][koff_s ..]^T + bias + rscale * R ) )
 * X_s: bf16 [rows_s][ld_s], row r of the product reads row r + shift_s (a tap); with Ti > 0 rows are
 * clips of Ti rows and taps that leave their clip read zero, with Ti == 0 only the matrix bounds apply.
 * W: bf16 [N][ldw], segment s in columns [koff_s, koff_s + k_s) (k_s a multiple of 8; padding finite).
 * mask: keep where mask[row][col] > 0 (the ReLU derivative from a stored activation).
 * Output bf16 [M][ldy], or fp32 (out_f32; accumulate: +=).  nsplit > 1 (fp32 only, no bias / R / mask):
 * the K chunks are split over nsplit partial outputs Y + z*split_stride, to be summed by
 * fwn_reduce_splits (fixed order: deterministic).
 * gate_aux != NULL (bf16 output only): the 256 output columns from gate_col0 (a multiple of 256) are a gradient d with
 * respect to a gated layer's output o = tanh(f) sigmoid(g); they are not stored in Y but, rounded to bf16 like Y, go
 * through the gate's derivative with gate_aux = [tanh f | sigmoid g] (bf16 [M][512], kept by fwn_gate_train):
 * gate_out[M][512] (bf16) = [d sg (1 - tf^2) | d tf sg (1 - sg)] - what fwn_gate_bwd computes from a stored d. */
#define FWN_GEMM_MAXSEG 8
typedef struct fwn_gemm_seg { const void* x; int32_t rows, ld, k, shift, koff, pad_; } fwn_gemm_seg;
typedef struct fwn_gemm_desc {
    fwn_gemm_seg seg[FWN_GEMM_MAXSEG];
    int32_t nseg, M, N, Ti;
    const void* W;      int32_t ldw, pad0_;
    const float* bias;
    const void* R;      int32_t ldr;    float rscale;
    const void* mask;   int32_t ldmask; int32_t relu;
    void* Y;            int32_t ldy;    int32_t out_f32;
    int32_t accumulate, nsplit;
    int64_t split_stride;
    float oscale;       int32_t gate_col0;
    const void* gate_aux; void* gate_out;
} fwn_gemm_desc;
int fwn_gemm(const fwn_gemm_desc* g, void* stream);
/* For tap z < ntap (shift = shift0 + z*dshift): dst[z*C + c][m] = src[m + shift][c] (zero where the tap
 * leaves its clip / the matrix; columns m >= M zero), dst bf16 [ntap*C (+1)][ld_dst]; ones_row: the row after
 * the last tap = 1 for m < M (bias gradients ride the weight-gradient GEMM). */
int fwn_transpose_shift(const void* src, int M, int C, int ld_src, int shift0, int dshift, int ntap, int Ti, void* dst,
                        int ld_dst, int ones_row, void* stream);
int fwn_reduce_splits(const float* partial, int nsplit, int64_t stride, int64_t n, float scale, float* out,
                      void* stream);

/* Training forward of a gated layer: as fwn_gate (conditioning fused from ca, or hoisted: P = c_a Wc of this layer
 * from fwn_cond; exactly one of the two), also storing aux [M][512] bf16 = (tanh f | sigmoid g) in natural channel
 * order for fwn_gate_bwd. */
int fwn_gate_train(const fwn_flow_desc* d, int layer, const void* h, const void* ca, const float* P, void* o, void* aux,
                   int M, int Ti, void* stream);
/* Element-wise pieces.  Planes are fp32 [M][Ch]; `an` points at one plane's table [4][Ch] = (shift,
 * scale, 1/scale, 3 logs) (fwn_flow_desc.an + role*4*Ch); Z is the ZeroConv output before its exp(3 scale)
 * factor ez [2Ch] (modules.py:51-56), fp32 [M][2Ch]: (log_s | t) = Z * ez.
 * fwn_actnorm_apply: x <- (x + shift) scale.                                  (model.py:86-94)
 * fwn_coupling_fwd : y_b <- (y_b - t) exp(-log_s); partial[nblocks] <- sums of -log_s   (:124-141)
 * fwn_coupling_bwd : g = dL/d out_b <- dL/d y_b; out_b <- y_b; dZ (bf16, ld ldz) <- (dL/dlog_s | dL/dt)
 *                    with the log-det term cls = 1/(2 M Ch) added; dzz <- dZ * Z (ZeroConv scale gradient)
 * fwn_gate_bwd     : dpre [M][512] <- (do sg (1 - tf^2) | do tf sg (1 - sg)), do bf16 [M][256] with row stride
 *                    ld_do (a column block of a wider matrix is fine)                  (modules.py:124)
 * fwn_colsum_prod  : out[c] <- scale * sum_m A[m][c] * (B ? B[m][c] : 1), fp32 [M][C], fixed order;
 *                    partial: scratch of fwn_colsum_partials(M, C) floats
 * fwn_actnorm_bwd  : dy <- dy * scale; y <- y / scale - shift (the plane before ActNorm)
 * fwn_wn_backward_group (below, with the grouped weight-gradient GEMM): weight-norm backward straight from
 *                    the split-K partials of the weight-gradient GEMM: dW[k][n] = scale * sum_s part[s][row_src ?
 *                    row_src[k] : k][col0 + n] (part rows have ldp columns), db[n] = the same sum over row
 *                    bias_row (< 0 / NULL: none); V fp32 [K][N], g [N] -> dV, dg; g == NULL: dV = dW (no
 *                    weight norm)   (convolutional.py:73-80) */
int fwn_actnorm_apply(float* x, const float* an, int64_t n, int Ch, void* stream);
/* Both planes of a flow in one launch: an2 = the flow's table [2][4][Ch] (fwn_flow_desc.an), n elements per plane. */
int fwn_actnorm_apply2(float* xa, float* xb, const float* an2, int64_t n, int Ch, void* stream);
int fwn_coupling_fwd(float* yb, const float* Z, const float* ez, int64_t M, int Ch, float* partial, int nblocks,
                     void* stream);
int fwn_coupling_bwd(float* g, float* out_b, const float* Z, const float* ez, int64_t M, int Ch, float cls, void* dZ,
                     int ldz, float* dzz, void* stream);
int fwn_gate_bwd(const void* d_o, int ld_do, const void* aux, int64_t M, void* dpre, void* stream);
int fwn_colsum_partials(int64_t M, int C);
int fwn_colsum_prod(const float* A, const float* B, int64_t M, int C, float scale, float* partial, float* out,
                    void* stream);
int fwn_actnorm_bwd(float* dy, float* y, const float* an, int64_t n, int Ch, void* stream);
/* The parameter-sized gradients around a flow's ActNorm, fused (two launches): for both planes (g, y = ActNorm
 * output; an [2][4][Ch]) s1 = sum_m g, s2 = sum_m g y, then fwn_actnorm_bwd in place; z = column sums of dzz
 * [M][2 Ch].  Outputs in the parameters' order through the index tables br [Ch] / zc [2 Ch] (int64, device):
 * db[role Ch + br[c]] = s1 scale, dlogs[role Ch + br[c]] = 3 s2 - 3/(2 Ch), dzscale[zc[j]] = 3 z[j].
 * partial: fwn_flow_small_grads_partials(M, Ch) doubles.  Ch: power of two <= 128. */
int64_t fwn_flow_small_grads_partials(int64_t M, int Ch);
int fwn_flow_small_grads(float* ga, float* ya, float* gb, float* yb, const float* dzz, const float* an, int64_t M, int Ch,
                         const int64_t* br, const int64_t* zc, double* partial, float* db, float* dlogs, float* dzscale,
                         void* stream);

/* Backward of one up-sampling stage (fwn_upsample_stage with fp32 output): y, dy [B][H*s][W], x [B][H][W].
 * dy <- dy * LeakyReLU'(y) in place; dx (may be NULL) <- gradient wrt x; dwk_bias [6s + 1] <- gradients of
 * the (weight-normed) kernel [2s][3] followed by the bias, two fixed-order passes through `partial`
 * (fwn_upsample_bwd_partials(B, H, s) floats).                                      (model.py:301-311) */
int fwn_upsample_bwd_partials(int B, int H, int s);
int fwn_upsample_bwd(float* dy, const float* y, const float* x, int B, int H, int W, int s, const float* wk,
                     float* dx, float* dwk_bias, float* partial, void* stream);

/* Weight-gradient GEMM without transposed copies: part[z][tap*Kx + i][j] = sum over the z-th share of the
 * rows m of X[m + shift0 + tap*dshift][i] * dY[m][j]  (bf16 X [M][ldx], dY [M][ldy]; taps that leave their clip of
 * Ti rows contribute zero; Ti == 0: matrix bounds only), fp32 partials [nsplit][ntap*Kx (+1)][N] for fwn_wn_backward /
 * fwn_reduce_splits; bias_row != 0 appends the column sums of dY (the bias gradient) as row ntap*Kx.  The operands are read transposed out of LDS (ds_read_b64_tr_b16).
 * fwn_colsum_bf16: out[c] = scale * sum_m dY[m][c] (bias gradients), scratch fwn_colsum_partials(M, C) floats. */
int fwn_tn_gemm(const void* x, int ldx, int Kx, int ntap, int shift0, int dshift, const void* dy, int ldy, int N, int M,
                int Ti, int nsplit, float* part, int64_t split_stride, int bias_row, void* stream);

/* Grouped forms: the weight gradients of one flow (same M, Ti) in ONE launch each - the job table travels in the
 * kernel arguments (at most FWN_MAX_GROUP jobs), every job needs only a few splits for the group to fill the chip. */
#define FWN_MAX_GROUP 16
typedef struct fwn_tn_job {
    const void* x; const void* dy; float* part; int64_t split_stride;
    int32_t ldx, Kx, ntap, shift0, dshift, ldy, N, nsplit, bias_row, reserved;
} fwn_tn_job;
int fwn_tn_gemm_group(const fwn_tn_job* jobs, int njobs, int M, int Ti, void* stream);
/* Edge of the square output tile the group launch uses at this M (128 or 256): what a caller sizes nsplit with. */
int fwn_tn_gemm_tile(int M);
typedef struct fwn_wn_job {
    const float* part; const int32_t* row_src; const float* V; const float* g; float* dV; float* dg; float* db;
    int64_t split_stride;
    int32_t nsplit, ldp, col0, bias_row, K, N; float scale; int32_t reserved;
    const int32_t* col_src;      /* NULL, or output column n reads partial column col0 + col_src[n] (a column permutation) */
} fwn_wn_job;
/* fwn_wn_backward for every job of the group (two launches); scratch: fwn_wn_group_scratch(jobs, njobs) doubles. */
int64_t fwn_wn_group_scratch(const fwn_wn_job* jobs, int njobs);
int fwn_wn_backward_group(const fwn_wn_job* jobs, int njobs, double* scratch, void* stream);
int fwn_colsum_bf16(const void* dy, int64_t M, int C, int ld, float scale, float* partial, float* out, void* stream);

/* ---- whole model (replaces FloWaveNet.forward / .reverse, model.py:317-396) ---- */
typedef struct fwn_model_desc {
    int32_t n_block, n_flow, n_layer, num_mels;
    int32_t n_up;
    int32_t up_scale[FWN_MAX_UPSAMPLE];
    const float* up_w[FWN_MAX_UPSAMPLE];      /* device [2s][3] weight-normed kernels */
    float up_bias[FWN_MAX_UPSAMPLE];
    const fwn_flow_desc* flows;               /* HOST array [n_block*n_flow] */
    int32_t cond_mode;                        /* 0 auto, 1 always fused in gate, 2 always hoisted */
    int32_t gate_fp8;                         /* != 0: fp8 dilated taps where supported (needs flows[].Wd8) */
    int32_t chain_mode;                       /* 0: chain the flows of a block (out_b to a third plane buffer, the next flow's
                                               * front conv in the previous flow's tail: csrc/tail_chain.h); 1: every flow on its own */
    int32_t persist_mode;                     /* flows of small-M blocks (hoisted conditioning) as ONE launch each (csrc/flow_persist.h;
                                               * same results bit for bit as a launch per stage): 0 = where that measured faster
                                               * (<= 512 rows: DESIGN.md section 3.7), 1 = never, 2 = wherever the form exists
                                               * (<= 4096 rows, n_layer <= 2) */
    /* Diagnostic (bench.py's per-block table), normally NULL: HOST array of n_block + 1 hipEvent_t handles.  The whole-model
     * calls record [k] on `stream` in front of the first launch of the k-th block they run (forward: block k, reverse: block
     * n_block - 1 - k) and [n_block] behind the last launch of the last one. */
    void* const* block_events;
    /* Per block: NULL, or the fragment streams (fwn_pack_cond_stream) of the block's n_flow * n_layer conditioning matrices in
     * (flow, layer) order: the hoisted projection of the block then runs the register-streamed kernel from fwn_cond_stream_rows()
     * rows on (0.3.21; a 0.3.20 host that zero-fills the descriptor keeps the ring tiles). */
    const void* cond_stream[16];
} fwn_model_desc;

size_t fwn_workspace_bytes(const fwn_model_desc* m, int64_t B, int64_t T);
/* x [B][T] fp32, mel [B][T/hop][num_mels] fp32 -> out2 = (log_p, logdet) fp32 on device.
 * z_planes (optional) receives the final flow state planes[2][B][T/2].  init != 0 performs the
 * ActNorm data-dependent init flow by flow (train.py:221,229 with init=True). */
int fwn_model_forward(const fwn_model_desc* m, int64_t B, int64_t T, const float* x, const float* mel,
                      void* workspace, size_t workspace_bytes, float* out2, float* z_planes, int init,
                      void* stream);
/* fwn_model_forward with init != 0 for one rank of a data-parallel job: before each flow's ActNorm tables are
 * derived, `reduce(user, buf, n, stream)` is called on the host with the DEVICE buffer of n doubles holding this
 * rank's moments (inside `workspace`); it must enqueue, in `stream` order, an in-place sum over all ranks (RCCL
 * all-reduce) and return 0.  reduce == NULL: single rank (moments of the local batch). */
typedef int (*fwn_reduce_fn)(void* user, double* buf, int n, void* stream);
int fwn_model_forward_init(const fwn_model_desc* m, int64_t B, int64_t T, const float* x, const float* mel,
                           void* workspace, size_t workspace_bytes, float* out2, float* z_planes,
                           fwn_reduce_fn reduce, void* user, void* stream);
/* z [B][T] fp32, mel -> x_out [B][T] fp32. */
int fwn_model_reverse(const fwn_model_desc* m, int64_t B, int64_t T, const float* z, const float* mel,
                      void* workspace, size_t workspace_bytes, float* x_out, void* stream);
/* The same question for a whole-model call (0.3.20): the per-flow sync blocks of fwn_model_forward / fwn_model_reverse live inside
 * their workspace.  After a pass with the same (m, B, T, workspace): 0 = no one-launch flow gave up, > 0 = the give-up code of the
 * first that did (log_p / logdet / the waveform are NaN then: retry, e.g. with fwn_model_desc.persist_mode = 1).  Synchronises. */
int fwn_model_persist_status(const fwn_model_desc* m, int64_t B, int64_t T, const void* workspace, void* stream);

/* ---- training: loss = -(log_p + logdet) (train.py:56-60) and its gradient with respect to every trainable tensor
 * (the one tf.gradients call of train.py:63-66) for one batch, in ONE call: training forward with what the backward
 * needs kept per flow, then the flows in reverse (coupling, ZeroConv / final / skip / res, the gated layers with their
 * dilated transposed convs, the conditioning gradient, the front conv, ActNorm), then the up-sampling convs.  The
 * descriptors hold device pointers to (a) the inference packing of the parameters (fwn_model_desc, conditioning fused:
 * cond_mode 1), (b) the natural-order / transposed bf16 copies the backward GEMMs read, (c) the fp32 masters in the
 * reference's layouts and (d) where each gradient goes (e.g. views of one flat buffer that an RCCL all-reduce sums).
 * Everything else lives in `workspace`.  on_block_done(user, i) is called on the host once every launch that writes
 * block i's gradients has been enqueued (blocks finish last to first; -1 = the up-sampling convs): the hook a
 * data-parallel step uses to start that block's all-reduce under the rest of the backward pass, or to cut a graph.
 * The hook returns 0 to go on; any other value stops the sequencing at once (nothing further is enqueued) and
 * fwn_train_loss_and_grads returns FWN_ERR_CALLBACK - a host that cannot let an exception cross the C frame reports
 * it this way and re-raises after the call. */
typedef struct fwn_conv_grad {          /* one trainable convolution */
    const float* V; const float* g;     /* kernel [K][N] fp32 (reference layout, K = kernel_size * C_in), weight-norm g [N] or NULL */
    float* dV; float* dg; float* db;    /* gradients of kernel, g (NULL iff g NULL) and bias */
} fwn_conv_grad;
typedef struct fwn_flow_train_desc {
    const void* WfT;                            /* [Ch][768]   front conv, transposed: K = tap*256 + n          */
    const void* WdT[FWN_MAX_LAYERS];            /* [256][1536] dilated filter|gate, K = tap*512 + (f|g)*256 + n */
    const void* WcT[FWN_MAX_LAYERS];            /* [cin][512]  conditioning filter|gate, row stride wct_ld      */
    const void* WresT[FWN_MAX_LAYERS];          /* [256][256]  layers 0..L-2                                    */
    /* Wskip / Wfin / Wz / bskip / bfin / bz: unused since the forward half runs the inference tail (fwn_tail_train) - may
     * be NULL; the backward reads the transposed copies and ez */
    const void* Wskip;  const void* WskipT_all; /* (unused), [L*256][256]                                       */
    const void* Wfin;   const void* WfinT;      /* (unused), [256][256] transposed                              */
    const void* Wz;     const void* WzT;        /* (unused), [256][ldz] columns in plane order                  */
    const float* bskip; const float* bfin; const float* bz; const float* ez;    /* ez [2Ch] = exp(3 scale), plane order */
    int32_t ldz;                                /* max(8, 2Ch)                                                  */
    int32_t wct_ld;                             /* row stride of WcT in elements (0: 512).  L*512 with WcT[l] =
                                                   WcT[0] + l*512: the layers side by side in one [cin][L*512] matrix -
                                                   the conditioning gradient of a flow is then ONE GEMM over K = L*512 */
    fwn_conv_grad front, final_, zero;          /* Conv_front, Conv_final, ZeroConv1d (g = NULL)                */
    fwn_conv_grad filt[FWN_MAX_LAYERS], gate[FWN_MAX_LAYERS], res[FWN_MAX_LAYERS], skip[FWN_MAX_LAYERS],
        filt_c[FWN_MAX_LAYERS], gate_c[FWN_MAX_LAYERS];
    float* d_an_b; float* d_an_logs; float* d_zscale;    /* [2Ch] each, the parameters' order                  */
} fwn_flow_train_desc;
typedef struct fwn_train_desc {
    const fwn_model_desc* model;                /* inference packing of the same parameters                     */
    const fwn_flow_train_desc* flows;           /* HOST array [n_block * n_flow]                                */
    /* per block (device): logical row of a weight gradient -> row of the GEMM that computed it; channel maps   */
    const int32_t* cond_rows[16]; const int32_t* front_rows[16]; const int32_t* zinv32[16];
    const int64_t* br[16]; const int64_t* zcol[16];
    const float* up_bias_dev[FWN_MAX_UPSAMPLE]; /* the bias masters (device scalars)                            */
    fwn_conv_grad up[FWN_MAX_UPSAMPLE];         /* V [2s][3], scalar g; dV, dg, db (bias)                       */
    const float* an_logdet;                     /* unused (the tail's log-det partials carry the ActNorm terms); any
                                                   non-NULL device pointer */
    int32_t zero_dead_res;                      /* != 0: also zero the gradients of the dead last-layer res_conv */
    /* != 0 (with side_stream): nothing consumes a block's gradients before the end of the call (one rank: no all-reduce),
     * so the side stream is NOT joined into `stream` block by block - the data-gradient chain never waits for the weight
     * gradients - but once, behind block 0; on_block_done(n_block - 1 .. 0) are then all called at that point, in order
     * (each still fires only after its block's gradients are complete in `stream` order - but none of them early: a hook that
     * starts collectives to overlap them with the backward pass wants 0 here).  On an error return every path first joins the
     * side stream into `stream`: nothing of the call is left running on it. */
    int32_t defer_block_done;
    /* Optional second hipStream_t (NULL: one stream).  The weight gradients of block i (grouped TN GEMMs + weight-norm
     * backward) and its conditioning-gradient GEMMs then run on it under the data-gradient chain of block i - 1 and are
     * joined into `stream` before on_block_done(i); the workspace grows by per-flow copies of the temporaries they
     * read.  Same results. */
    void* side_stream;
} fwn_train_desc;
typedef int (*fwn_block_done_fn)(void* user, int block);
size_t fwn_train_workspace_bytes(const fwn_train_desc* t, int64_t B, int64_t T);
/* x [B][T] fp32, mel [B][T/hop][num_mels] fp32 -> out3 = (loss, log_p, logdet) fp32 on device + every gradient. */
int fwn_train_loss_and_grads(const fwn_train_desc* t, int64_t B, int64_t T, const float* x, const float* mel,
                             void* workspace, size_t workspace_bytes, float* out3, fwn_block_done_fn on_block_done,
                             void* user, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FWN_H */
