// HBM-bound helper kernels around the flow chain (gfx950):
//   weight-norm + bf16 packing (convolutional.py:73-80), mel upsampling by two transposed
//   convs (model.py:398-404), even/odd plane split/merge (the squeeze of model.py:226-239 is
//   pure index math on these planes), ActNorm data-dependent init (model.py:30-83) and the
//   prior / log-det finalisation (model.py:342-343).
#include "common.h"
#include "fwn_internal.h"
#include "../../include/fwn.h"

// ---- weight-norm scale: scale[n] = g[n] / sqrt(max(sum_k V[k][n]^2, 1e-12)) ----------------
__global__ __launch_bounds__(1024) void wn_scale_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                        int k_src, int n_src, float* __restrict__ scale) {
    // 32 output channels x 32 K-partitions per workgroup (K is up to 10240 for the conditioning convs
    // and a training step recomputes every scale): coalesced over n, fixed-order tree over the partitions.
    __shared__ double red[32][33];
    const int nl = threadIdx.x & 31, kg = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + nl;
    double s = 0.0;
    if (n < n_src)
        for (int k = kg; k < k_src; k += 32) {
            const double x = v[(size_t)k * n_src + n];
            s += x * x;
        }
    red[kg][nl] = s;
    __syncthreads();
    for (int st = 16; st > 0; st >>= 1) {
        if (kg < st) red[kg][nl] += red[kg + st][nl];
        __syncthreads();
    }
    if (kg == 0 && n < n_src) scale[n] = (float)((double)g[n] / sqrt(fmax(red[0][nl], 1e-12)));
}

// ---- gather + scale + cast: out[n'][k'] = bf16(V[src_k[k']][src_n[n']] * scale[src_n[n']]) --
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ v, const float* __restrict__ scale,
                                                   const int* __restrict__ src_k, const int* __restrict__ src_n,
                                                   int n_src, int k_dst, long ld_dst, long total,
                                                   bf16* __restrict__ out) {
    // rows with src_n < 0 are skipped (left as the caller initialised them); columns with
    // src_k < 0 are written as zero (K padding).
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int kd = (int)(i % k_dst), nd = (int)(i / k_dst);
        const int sk = src_k[kd], sn = src_n[nd];
        if (sn < 0) continue;
        float val = 0.0f;
        if (sk >= 0) {
            val = v[(size_t)sk * n_src + sn];
            if (scale) val *= scale[sn];
        }
        out[(size_t)nd * ld_dst + kd] = (bf16)val;
    }
}

static inline int grid_for(long total);
// ---- e4m3 packing of the dilated conv weights (fp8 gate path, BASELINE configs[4]) ---------------------------------
// One power-of-two scale per packed matrix: e = floor(log2(448 / max|W|)), bytes = e4m3(W 2^e); the gate kernel undoes
// it with the MFMA's E8M0 scale operand.  absmax: max over all source elements of |v[k][n] scale[n] mul| (atomicMax on
// the bit pattern of a non-negative float: exact and order-independent); amax must be zeroed before the first call.
__global__ __launch_bounds__(256) void wn_absmax_kernel(const float* __restrict__ v, const float* __restrict__ scale,
                                                        int n_src, long total, float mul, unsigned int* __restrict__ amax) {
    float m = 0.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        float val = v[i] * mul;
        if (scale) val *= scale[i % n_src];
        m = fmaxf(m, fabsf(val));
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
    if ((threadIdx.x & 63) == 0) atomicMax(amax, __float_as_uint(m));
}
__device__ __forceinline__ int e4m3_exponent(float amax) {      // e with amax 2^e <= 448 < amax 2^(e+1)
    if (!(amax > 0.0f)) return 0;
    int e = (int)floorf(log2f(448.0f / amax));
    while (ldexpf(amax, e) > 448.0f) --e;
    while (ldexpf(amax, e + 1) <= 448.0f) ++e;
    return e < -100 ? -100 : (e > 100 ? 100 : e);
}
__global__ __launch_bounds__(256) void pack_e4m3_kernel(const float* __restrict__ v, const float* __restrict__ scale,
                                                        const int* __restrict__ src_k, const int* __restrict__ src_n,
                                                        int n_src, int k_dst, long ld_dst, long total, float mul,
                                                        const float* __restrict__ amax, unsigned char* __restrict__ out,
                                                        int* __restrict__ exp_out) {
    const int e = e4m3_exponent(*amax);
    if (blockIdx.x == 0 && threadIdx.x == 0) *exp_out = e;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int kd = (int)(i % k_dst), nd = (int)(i / k_dst);
        const int sk = src_k[kd], sn = src_n[nd];
        if (sn < 0) continue;
        float val = 0.0f;
        if (sk >= 0) {
            val = v[(size_t)sk * n_src + sn] * mul;
            if (scale) val *= scale[sn];
        }
        out[(size_t)nd * ld_dst + kd] = (unsigned char)pack_e4m3x2(ldexpf(val, e), 0.0f);
    }
}
// bf16 [n] -> e4m3 [n] (tests, and callers that hold only the bf16 h)
__global__ __launch_bounds__(256) void cast_e4m3_kernel(const bf16* __restrict__ src, unsigned char* __restrict__ dst, long n) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 2; i < n; i += (long)gridDim.x * 512) {
        const float a = (float)src[i], b = i + 1 < n ? (float)src[i + 1] : 0.0f;
        const unsigned int pk = pack_e4m3x2(a, b);
        dst[i] = (unsigned char)pk;
        if (i + 1 < n) dst[i + 1] = (unsigned char)(pk >> 8);
    }
}
void fwn_launch_wn_absmax(const float* v, const float* scale, int k_src, int n_src, float mul, float* amax, hipStream_t st) {
    const long total = (long)k_src * n_src;
    hipLaunchKernelGGL(wn_absmax_kernel, dim3(grid_for(total)), dim3(256), 0, st, v, scale, n_src, total, mul, (unsigned int*)amax);
}
void fwn_launch_pack_e4m3(const float* v, const float* scale, const int* src_k, const int* src_n, int n_src, int k_dst,
                          int n_dst, long ld_dst, float mul, const float* amax, void* out, int* exp_out, hipStream_t st) {
    const long total = (long)k_dst * n_dst;
    hipLaunchKernelGGL(pack_e4m3_kernel, dim3(grid_for(total)), dim3(256), 0, st, v, scale, src_k, src_n, n_src, k_dst, ld_dst,
                       total, mul, amax, (unsigned char*)out, exp_out);
}
void fwn_launch_cast_e4m3(const void* src, void* dst, long n, hipStream_t st) {
    hipLaunchKernelGGL(cast_e4m3_kernel, dim3(grid_for((n + 1) / 2)), dim3(256), 0, st, (const bf16*)src, (unsigned char*)dst, n);
}

// ---- small parameter tables straight from the flat fp32 masters (training: refreshed every step, on the device) ------
// out[i] = F(post[i] * sum_t flat[idx[t][i]])  (idx < 0: term absent).  mode[i]: 0 = fp64 sum, identity; 1 = fp64 sum,
// exp (both rounded once, the host path's arithmetic); 2 / 3 = the same two in fp32, terms added in order - the
// arithmetic of the framework expressions these tables used to be computed with (bit-compatible with them).
__global__ __launch_bounds__(256) void gather_tables_kernel(const float* __restrict__ flat, const long long* __restrict__ idx,
                                                            int nterm, long total, const double* __restrict__ post,
                                                            const unsigned char* __restrict__ mode, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int md = mode[i];
        if (md < 2) {
            double acc = 0.0;
            for (int t = 0; t < nterm; ++t) {
                const long long ix = idx[(size_t)t * total + i];
                acc += ix >= 0 ? (double)flat[ix] : 0.0;
            }
            acc *= post[i];
            out[i] = (float)(md ? exp(acc) : acc);
        } else {
            float acc = 0.0f;
            bool first = true;
            for (int t = 0; t < nterm; ++t) {
                const long long ix = idx[(size_t)t * total + i];
                if (ix < 0) continue;
                acc = first ? flat[ix] : acc + flat[ix];
                first = false;
            }
            const float pf = (float)post[i];
            acc = md == 3 ? expf(pf * acc) : acc * pf;
            out[i] = acc;
        }
    }
}
// out[0] = sum_i in[i] (one workgroup, fixed order, fp64 accumulation)
__global__ __launch_bounds__(256) void sum_f32_kernel(const float* __restrict__ in, long n, float* __restrict__ out) {
    __shared__ double red[256];
    double a = 0.0;
    for (long i = threadIdx.x; i < n; i += 256) a += (double)in[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)red[0];
}
// weight-normed up-sampling kernel (convolutional.py:179-186): out[k][kw] = v[k][kw] / sqrt(max(sum_k v[k][kw]^2, 1e-12)) * g
__global__ void upsample_wn_kernel(const float* __restrict__ v, const float* __restrict__ g, int s, float* __restrict__ out) {
    const int kw = threadIdx.x;
    if (kw >= 3) return;
    double ss = 0.0;
    for (int k = 0; k < 2 * s; ++k) ss += (double)v[k * 3 + kw] * (double)v[k * 3 + kw];
    const double nrm = sqrt(fmax(ss, 1e-12));
    for (int k = 0; k < 2 * s; ++k) out[k * 3 + kw] = (float)((double)v[k * 3 + kw] / nrm * (double)g[0]);
}
void fwn_launch_gather_tables(const float* flat, const long long* idx, int nterm, long total, const double* post,
                              const unsigned char* expflag, float* out, hipStream_t st) {
    hipLaunchKernelGGL(gather_tables_kernel, dim3(grid_for(total)), dim3(256), 0, st, flat, idx, nterm, total, post, expflag, out);
}
void fwn_launch_sum_f32(const float* in, long n, float* out, hipStream_t st) {
    hipLaunchKernelGGL(sum_f32_kernel, dim3(1), dim3(256), 0, st, in, n, out);
}
void fwn_launch_upsample_wn(const float* v, const float* g, int s, float* out, hipStream_t st) {
    hipLaunchKernelGGL(upsample_wn_kernel, dim3(1), dim3(64), 0, st, v, g, s, out);
}

// ---- grouped form of the two kernels above: a whole model's weight-norm scales and packed copies
// in two launches driven by device-resident job tables (a training step re-packs every weight from
// the fp32 masters; the tables are built once because master and output pointers are stable).
__global__ __launch_bounds__(1024) void wn_scale_jobs_kernel(const fwn_scale_job* __restrict__ jobs,
                                                             float* __restrict__ scales, int scale_ld) {
    __shared__ double red[32][33];
    const fwn_scale_job j = jobs[blockIdx.x];
    const int nl = threadIdx.x & 31, kg = threadIdx.x >> 5;
    const int n = blockIdx.y * 32 + nl;
    if (blockIdx.y * 32 >= j.n_src) return;
    double s = 0.0;
    if (n < j.n_src) {
        // 16 loads in flight per thread (a conditioning conv of the last block has 320 rows per thread: one dependent
        // load at a time was a 300 us latency chain); the squares are still added in ascending k
        typedef const __attribute__((address_space(1))) float* gf32;
        const gf32 v = (gf32)j.v + n;
        const size_t ld = (size_t)j.n_src;
        int k = kg;
        for (; k + 15 * 32 < j.k_src; k += 16 * 32) {
            float x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) x[u] = v[(size_t)(k + 32 * u) * ld];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += (double)x[u] * (double)x[u];
        }
        for (; k < j.k_src; k += 32) {
            const double x = v[(size_t)k * ld];
            s += x * x;
        }
    }
    red[kg][nl] = s;
    __syncthreads();
    for (int st = 16; st > 0; st >>= 1) {
        if (kg < st) red[kg][nl] += red[kg + st][nl];
        __syncthreads();
    }
    if (kg == 0 && n < j.n_src)
        scales[(size_t)blockIdx.x * scale_ld + n] = (float)((double)j.g[n] / sqrt(fmax(red[0][nl], 1e-12)));
}
// Every load is unconditional (indices clamped into range, the result selected afterwards): written as
// `valid ? v[..] : 0` each load sat in its own branch behind an s_waitcnt vmcnt(0) - one memory round trip per element.
template <bool HAS_SC>
__device__ __forceinline__ void pack_job_body(const fwn_pack_job& j, const float* __restrict__ sc, float (*tile)[65], int by, int ny) {
    bf16* out = (bf16*)j.out;
    const int kmax = j.k_dst - 1, nmax = j.n_dst - 1;
    // (scale * mul) first, like fwn_pack_bf16 fed a pre-multiplied scale: same bf16 bits
    // the job's pointers come from a table in memory: tell the compiler they are global (global_load, not flat_load)
    typedef const __attribute__((address_space(1))) float* gf32;
    typedef const __attribute__((address_space(1))) int* gi32;
    const gf32 jv = (gf32)j.v, scg = (gf32)sc;
    const gi32 jsk = (gi32)j.src_k, jsn = (gi32)j.src_n;
    auto value = [&](int kd, int nd, bool& skip) {
        const int sk = jsk[min(kd, kmax)], sn = jsn[min(nd, nmax)];
        skip = sn < 0 || nd > nmax;
        const int snc = max(sn, 0);
        const float v = jv[(size_t)max(sk, 0) * j.n_src + snc];
        const float m = HAS_SC ? scg[snc] * j.mul : j.mul;
        // a factor, not a select: a select on a loaded value is turned back into a branch around the load
        const float okf = (sk >= 0 && sn >= 0 && kd <= kmax && nd <= nmax) ? 1.0f : 0.0f;
        return v * m * okf;
    };
    if (j.transposed) {          // out[k][n], n contiguous like the source: straight through, 4 columns per thread
        const int n4 = (j.n_dst + 3) / 4;
        const long total = (long)j.k_dst * n4;
        const bool vec = (j.ld_dst & 3) == 0 && (((uintptr_t)out) & 7) == 0;
        // Fast form: the 4 columns map to 4 consecutive, 16-byte aligned source columns (identity / offset maps - every
        // training copy): one 16-byte load of the master and of the scales instead of 4 x (2 index loads + 2 loads).  The
        // per-element gathers of the general form made these jobs instruction-bound (2.1 TB/s against 3.3 for the
        // transposing jobs).  Same products in the same order: identical bits.
        const bool fast_ok = vec && (j.n_dst & 3) == 0 && (j.n_src & 3) == 0 && (((uintptr_t)j.v) & 15) == 0 && (((uintptr_t)j.src_n) & 15) == 0 &&
                             (!HAS_SC || (((uintptr_t)sc) & 15) == 0);
        typedef int vi4 __attribute__((ext_vector_type(4)));
        typedef float vf4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) vi4* gi32x4;
        typedef const __attribute__((address_space(1))) vf4* gf32x4;
        for (long i = (long)by * 256 + threadIdx.x; i < total; i += (long)ny * 256) {
            const int kd = (int)(i / n4), nd0 = (int)(i % n4) * 4;
            if (fast_ok) {
                const vi4 sn = *(gi32x4)(jsn + nd0);
                const int sk = jsk[kd];
                if (sk >= 0 && sn.x >= 0 && (sn.x & 3) == 0 && sn.y == sn.x + 1 && sn.z == sn.x + 2 && sn.w == sn.x + 3) {
                    const vf4 v4 = *(gf32x4)(jv + (size_t)sk * j.n_src + sn.x);
                    vf4 m4 = {j.mul, j.mul, j.mul, j.mul};
                    if (HAS_SC) {
                        const vf4 s4 = *(gf32x4)(scg + sn.x);
                        m4 = vf4{s4.x * j.mul, s4.y * j.mul, s4.z * j.mul, s4.w * j.mul};
                    }
                    union { bf16 e[4]; uint2 u; } pk;
                    pk.e[0] = (bf16)(v4.x * m4.x * 1.0f); pk.e[1] = (bf16)(v4.y * m4.y * 1.0f);
                    pk.e[2] = (bf16)(v4.z * m4.z * 1.0f); pk.e[3] = (bf16)(v4.w * m4.w * 1.0f);
                    *(uint2*)(out + (size_t)kd * j.ld_dst + nd0) = pk.u;
                    continue;
                }
            }
            float v[4];
            bool skip[4], any_skip = false;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = value(kd, nd0 + e, skip[e]);
                any_skip = any_skip || skip[e];
            }
            bf16* o = out + (size_t)kd * j.ld_dst + nd0;
            if (vec && !any_skip) {
                union { bf16 e[4]; uint2 u; } pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk.e[e] = (bf16)v[e];
                *(uint2*)o = pk.u;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (!skip[e]) o[e] = (bf16)v[e];
            }
        }
        return;
    }
    // out[n][k], k contiguous: a transpose of the source - 64 x 64 tiles through LDS so that both the fp32 reads
    // (n fastest) and the bf16 writes (k fastest) are coalesced
    const int tk = (j.k_dst + 63) / 64, tn = (j.n_dst + 63) / 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    for (int t = by; t < tk * tn; t += ny) {
        const int k0 = (t % tk) * 64, n0 = (t / tk) * 64;
        __syncthreads();
        float val[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {      // 16 independent gathers in flight
            bool skip;
            val[r] = value(k0 + ly + 4 * r, n0 + lx, skip);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[ly + 4 * r][lx] = val[r];
        __syncthreads();
        // output rows whose source column is -1 belong to another job of the same matrix: left alone
        int sno[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sno[r] = jsn[min(n0 + ly + 4 * r, nmax)];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nd = n0 + ly + 4 * r, kd = k0 + lx;
            if (kd < j.k_dst && nd < j.n_dst && sno[r] >= 0) out[(size_t)nd * j.ld_dst + kd] = (bf16)tile[lx][ly + 4 * r];
        }
    }
}
__global__ __launch_bounds__(256) void pack_jobs_kernel(const fwn_pack_job* __restrict__ jobs,
                                                        const float* __restrict__ scales, int scale_ld) {
    __shared__ float tile[64][65];
    // blockIdx.y = job: the workgroups of a job are dispatched together and the jobs in table order - jobs that read the same
    // master (its inference packing and its transposed training copy: the host sorts the table by source) run back to back
    // and the second one finds the master in the Infinity Cache (with the job index in x every job was in flight at once)
    const fwn_pack_job j = jobs[blockIdx.y];
    if (j.scale_slot >= 0) pack_job_body<true>(j, scales + (size_t)j.scale_slot * scale_ld, tile, blockIdx.x, gridDim.x);
    else pack_job_body<false>(j, nullptr, tile, blockIdx.x, gridDim.x);
}
void fwn_launch_pack_jobs(const fwn_scale_job* sjobs, int nsjobs, const fwn_pack_job* jobs, int njobs, float* scales,
                          int scale_ld, hipStream_t st) {
    if (nsjobs > 0)
        hipLaunchKernelGGL(wn_scale_jobs_kernel, dim3(nsjobs, (scale_ld + 31) / 32), dim3(1024), 0, st, sjobs, scales, scale_ld);
    if (njobs > 0) hipLaunchKernelGGL(pack_jobs_kernel, dim3(FWN_TUNE(FWN_PACK_Y, 160), njobs), dim3(256), 0, st, jobs, scales, scale_ld);
}

// ---- one Conv2DTranspose(filters=1, kernel (2s,3), strides (s,1), 'same') + LeakyReLU(0.4) --
// Gather form of SURVEY Appendix A: y[tau, w] = bias + sum over (i,k) with i*s + k - s/2 = tau
// and kw of x[i, w - kw + 1] * wk[k][kw].
__global__ __launch_bounds__(256) void upsample_kernel(const float* __restrict__ in, int B, int H, int W,
                                                       const float* __restrict__ wk, float bias_host,
                                                       const float* __restrict__ bias_dev, int s,
                                                       float* __restrict__ out_f32, bf16* __restrict__ out_planes) {
    const long total = (long)B * H * s * W;
    const int half = W >> 1;
    const float bias = bias_dev ? bias_dev[0] : bias_host;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int w = (int)(idx % W);
        const long rt = idx / W;
        const int tau = (int)(rt % ((long)H * s));
        const int b = (int)(rt / ((long)H * s));
        const int i1 = (tau + s / 2) / s;
        const int k1 = tau + s / 2 - i1 * s;
        // six taps, all loads unconditional (clamped addresses, validity as a factor): a branch per load costs a memory
        // round trip each.  The products are added in the same order as before.
        float xv[2][3], wv[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = i1 - j, k = k1 + j * s;
            const bool iok = i >= 0 && i < H;
            const float* xr = in + ((size_t)b * H + min(max(i, 0), H - 1)) * W;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ws = w - kw + 1;
                const bool ok = iok && ws >= 0 && ws < W;
                xv[j][kw] = xr[min(max(ws, 0), W - 1)] * (ok ? 1.0f : 0.0f);
                wv[j][kw] = wk[min(k, 2 * s - 1) * 3 + kw];
            }
        }
        float acc = bias;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) acc += xv[j][kw] * wv[j][kw];
        acc = fmaxf(acc, 0.4f * acc);
        if (out_f32) out_f32[idx] = acc;
        if (out_planes) {
            // planes[q][b][tau][w - q*half], q = mel half (conditioning half of change_order)
            const int q = w >= half;
            out_planes[(((size_t)q * B + b) * ((size_t)H * s) + tau) * half + (w - q * half)] = (bf16)acc;
        }
    }
}

// The same stage, 8 consecutive mel bins per thread (W and W / 2 multiples of 8: the real 80-bin front-end): the two input
// rows arrive as 16-byte pieces, the output leaves as one 16-byte piece of bf16 (or two of fp32).  One element per thread
// was three integer divisions, twelve 4-byte loads and a 2-byte store per output: 50 us for the 10 M outputs of the bench
// batch's last stage.  The products are added in the order of upsample_kernel: bit-identical results.
__global__ __launch_bounds__(256) void upsample8_kernel(const float* __restrict__ in, int B, int H, int W,
                                                        const float* __restrict__ wk, float bias_host,
                                                        const float* __restrict__ bias_dev, int s,
                                                        float* __restrict__ out_f32, bf16* __restrict__ out_planes) {
    const int W8 = W >> 3, half = W >> 1;
    const long total = (long)B * H * s * W8;
    const float bias = bias_dev ? bias_dev[0] : bias_host;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int w = (int)(idx % W8) * 8;
        const long rt = idx / W8;
        const int tau = (int)(rt % ((long)H * s));
        const int b = (int)(rt / ((long)H * s));
        const int i1 = (tau + s / 2) / s;
        const int k1 = tau + s / 2 - i1 * s;
        float x[2][10], wv[2][3];        // x[j][e] = in[i_j][w - 1 + e], zero outside the row / the image
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = i1 - j, k = k1 + j * s;
            const float ok = (i >= 0 && i < H) ? 1.0f : 0.0f;
            const float* xr = in + ((size_t)b * H + min(max(i, 0), H - 1)) * W;
            const float4 a = *(const float4*)(xr + w), c = *(const float4*)(xr + w + 4);
            const float lo = xr[max(w - 1, 0)], hi = xr[min(w + 8, W - 1)];
            x[j][0] = lo * (w > 0 ? ok : 0.0f);
            x[j][1] = a.x * ok; x[j][2] = a.y * ok; x[j][3] = a.z * ok; x[j][4] = a.w * ok;
            x[j][5] = c.x * ok; x[j][6] = c.y * ok; x[j][7] = c.z * ok; x[j][8] = c.w * ok;
            x[j][9] = hi * (w + 8 < W ? ok : 0.0f);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) wv[j][kw] = wk[min(k, 2 * s - 1) * 3 + kw];
        }
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float acc = bias;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc += x[j][e + 2 - kw] * wv[j][kw];      // in[i][w + e - kw + 1]
            y[e] = fmaxf(acc, 0.4f * acc);
        }
        if (out_f32) {
            float* o = out_f32 + (rt * W + w);
            *(float4*)o = make_float4(y[0], y[1], y[2], y[3]);
            *(float4*)(o + 4) = make_float4(y[4], y[5], y[6], y[7]);
        }
        if (out_planes) {
            const int q = w >= half;
            Pack16 pk;
#pragma unroll
            for (int e = 0; e < 8; ++e) pk.e[e] = (bf16)y[e];
            *(uint4*)(out_planes + (((size_t)q * B + b) * ((size_t)H * s) + tau) * half + (w - q * half)) = pk.u;
        }
    }
}

// ---- x[B][T] <-> planes[2][B][T/2] (even / odd samples) ------------------------------------
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ x, long B, long T,
                                                    float* __restrict__ planes) {
    const long n = B * T, hT = T >> 1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long b = i / T, t = i % T;
        planes[((t & 1) * B + b) * hT + (t >> 1)] = x[i];
    }
}
__global__ __launch_bounds__(256) void merge_kernel(const float* __restrict__ planes, long B, long T,
                                                    float* __restrict__ x) {
    const long n = B * T, hT = T >> 1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long b = i / T, t = i % T;
        x[i] = planes[((t & 1) * B + b) * hT + (t >> 1)];
    }
}

// ---- ActNorm data-dependent init for one flow -----------------------------------------------
// One workgroup per (plane role, channel).  an[role][0..3][tau] = shift b, scale exp(3 logs),
// inverse scale, 3*logs.  model.py:55-56 (b = -mean), :65-71 (logs from mean((x+b)^2)).
__device__ double block_sum(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}
__global__ __launch_bounds__(256) void ddi_kernel(const float* __restrict__ xa, const float* __restrict__ xb,
                                                  int M, int Ch, float* __restrict__ an) {
    __shared__ double red[256];
    const int role = blockIdx.x / Ch, tau = blockIdx.x % Ch;
    const float* src = role ? xb : xa;
    double s = 0.0;
    for (int r = threadIdx.x; r < M; r += 256) s += src[(size_t)r * Ch + tau];
    const double mean = block_sum(s, red) / M;
    const float bshift = (float)(-mean);
    double s2 = 0.0;
    for (int r = threadIdx.x; r < M; r += 256) {
        const double d = (double)(src[(size_t)r * Ch + tau] + bshift);
        s2 += d * d;
    }
    const double var = block_sum(s2, red) / M;
    if (threadIdx.x == 0) {
        const double den = sqrt(var) + 1e-7;
        float* o = an + (size_t)role * 4 * Ch;
        o[tau] = bshift;
        o[Ch + tau] = (float)(1.0 / den);
        o[2 * Ch + tau] = (float)den;
        o[3 * Ch + tau] = (float)(-log(den));
    }
}

// ---- the same init from moments summed over ranks (data-parallel DDI) -------------------------------
// The reference lets its towers race on the assign (model.py:39 under train.py:43-57); here every rank adds its
// per-channel sum and sum of squares, the caller all-reduces the 4 Ch + 1 doubles (the last one counts rows), and
// every rank derives the same table: b = -mean, var((x + b)) = E[x^2] + 2 b E[x] + b^2 with the fp32-rounded b the
// single-process kernel above also centres with.
__global__ __launch_bounds__(256) void ddi_moments_kernel(const float* __restrict__ xa, const float* __restrict__ xb,
                                                          int M, int Ch, double* __restrict__ mom) {
    __shared__ double red[256];
    const int role = blockIdx.x / Ch, tau = blockIdx.x % Ch;
    const float* src = role ? xb : xa;
    double s = 0.0, s2 = 0.0;
    for (int r = threadIdx.x; r < M; r += 256) {
        const double v = src[(size_t)r * Ch + tau];
        s += v;
        s2 += v * v;
    }
    const double t1 = block_sum(s, red), t2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        mom[(size_t)role * 2 * Ch + tau] = t1;
        mom[(size_t)role * 2 * Ch + Ch + tau] = t2;
        if (blockIdx.x == 0) mom[4 * (size_t)Ch] = (double)M;
    }
}
__global__ void ddi_from_moments_kernel(const double* __restrict__ mom, int Ch, float* __restrict__ an) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * Ch) return;
    const int role = i / Ch, tau = i % Ch;
    const double n = mom[4 * (size_t)Ch];
    const double mean = mom[(size_t)role * 2 * Ch + tau] / n, ex2 = mom[(size_t)role * 2 * Ch + Ch + tau] / n;
    const float bshift = (float)(-mean);
    const double b = (double)bshift;
    const double var = fmax(ex2 + 2.0 * b * mean + b * b, 0.0);
    const double den = sqrt(var) + 1e-7;
    float* o = an + (size_t)role * 4 * Ch;
    o[tau] = bshift;
    o[Ch + tau] = (float)(1.0 / den);
    o[2 * Ch + tau] = (float)den;
    o[3 * Ch + tau] = (float)(-log(den));
}

// ---- prior + log-det finalisation: out2 = (log_p, logdet), model.py:342-347 -----------------
__global__ __launch_bounds__(1024) void prior_kernel(const float* __restrict__ z, long n,
                                                     const float* __restrict__ partial, int n_partial,
                                                     double inv_bt, float* __restrict__ out2) {
    __shared__ double r1[1024], r2[1024];
    const int tid = threadIdx.x;
    double s = 0.0, p = 0.0;
    // eight loads in flight per thread, added in the same ascending order as one at a time (one dependent load per
    // iteration was a chain of n / 1024 memory round trips: 40 us for the bench batch, on the critical path of the pass)
    long i = tid;
    for (; i + 7 * 1024 < n; i += 8 * 1024) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = z[i + 1024L * u];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (double)v[u] * (double)v[u];
    }
    for (; i < n; i += 1024) {
        const double v = z[i];
        s += v * v;
    }
    int j = tid;
    for (; j + 7 * 1024 < n_partial; j += 8 * 1024) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[j + 1024 * u];
#pragma unroll
        for (int u = 0; u < 8; ++u) p += v[u];
    }
    for (; j < n_partial; j += 1024) p += partial[j];
    r1[tid] = s;
    r2[tid] = p;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if (tid < st) { r1[tid] += r1[tid + st]; r2[tid] += r2[tid + st]; }
        __syncthreads();
    }
    if (tid == 0) {
        out2[0] = (float)(0.5 * (-1.8378770664093453 - r1[0] / (double)n));
        out2[1] = (float)(r2[0] * inv_bt);
    }
}

// ---- data-parallel optimiser step (train.py:15-32,75-81, utils.py:34-60) --------------------
// Gradients live in ONE flat fp32 buffer (what the RCCL all-reduce sums); the reference's
// average_gradients / un-scale / clip_by_global_norm / Adam chain becomes two HBM-bound passes:
//   (1) per-workgroup partial sums of g^2 (fixed order -> deterministic global norm),
//   (2) fused  g' = g * gscale / max(||g * gscale||, clip);  TF-form Adam on fp32 master weights.
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, long n,
                                                             double* __restrict__ partial) {
    __shared__ double red[256];
    double s = 0.0;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 3 < n) {
            const float4 v = *(const float4*)(g + i);
            s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        } else {
            for (long j = i; j < n; ++j) s += (double)g[j] * g[j];
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// out[0] = sqrt(sum partial) * gscale  (= global norm of the averaged, un-scaled gradient)
__global__ __launch_bounds__(256) void sqnorm_final_kernel(const double* __restrict__ partial, int np,
                                                           float gscale, float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(sqrt(red[0]) * (double)gscale);
}
// tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v updates; w -= lr_t*m/(sqrt(v)+eps)
__global__ __launch_bounds__(256) void adam_clip_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long n,
                                                        const float* __restrict__ gnorm, float gscale, float clip,
                                                        float lr_host, const float* __restrict__ lr_dev, float b1,
                                                        float b2, float eps) {
    const float lr_t = lr_dev ? lr_dev[0] : lr_host;     // device-resident rate: the launch can live in a hipGraph
    const float sc = gscale / fmaxf(gnorm[0], clip);     // tf.clip_by_global_norm: g / max(gn, clip)*clip, clip=1
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 3 < n) {
            const float4 gv = *(const float4*)(g + i);
            float4 mv = *(float4*)(m + i), vv = *(float4*)(v + i), wv = *(float4*)(w + i);
            const float gg[4] = {gv.x * sc * clip, gv.y * sc * clip, gv.z * sc * clip, gv.w * sc * clip};
            float* mm = &mv.x; float* v2 = &vv.x; float* ww = &wv.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mm[k] = b1 * mm[k] + (1.0f - b1) * gg[k];
                v2[k] = b2 * v2[k] + (1.0f - b2) * gg[k] * gg[k];
                ww[k] -= lr_t * mm[k] / (sqrtf(v2[k]) + eps);
            }
            *(float4*)(m + i) = mv; *(float4*)(v + i) = vv; *(float4*)(w + i) = wv;
        } else {
            for (long j = i; j < n; ++j) {
                const float gj = g[j] * sc * clip;
                m[j] = b1 * m[j] + (1.0f - b1) * gj;
                v[j] = b2 * v[j] + (1.0f - b2) * gj * gj;
                w[j] -= lr_t * m[j] / (sqrtf(v[j]) + eps);
            }
        }
    }
}

// ---- launchers -------------------------------------------------------------------------------
static inline int grid_for(long total) {
    long g = (total + 255) / 256;
    return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}
void fwn_launch_wn_scale(const float* v, const float* g, int k_src, int n_src, float* scale, hipStream_t st) {
    hipLaunchKernelGGL(wn_scale_kernel, dim3((n_src + 31) / 32), dim3(1024), 0, st, v, g, k_src, n_src, scale);
}
void fwn_launch_pack(const float* v, const float* scale, const int* src_k, const int* src_n, int n_src,
                     int k_dst, int n_dst, long ld_dst, void* out, hipStream_t st) {
    const long total = (long)k_dst * n_dst;
    hipLaunchKernelGGL(pack_kernel, dim3(grid_for(total)), dim3(256), 0, st, v, scale, src_k, src_n, n_src,
                       k_dst, ld_dst, total, (bf16*)out);
}
void fwn_launch_upsample(const float* in, int B, int H, int W, const float* wk, float bias, const float* bias_dev, int s,
                         float* out_f32, void* out_planes, hipStream_t st) {
    const long total = (long)B * H * s * W;
    const bool al = (((uintptr_t)in | (uintptr_t)out_f32 | (uintptr_t)out_planes) & 15) == 0;
    if (W % 16 == 0 && al) {       // 8 bins per thread: whole 16-byte pieces in both mel halves
        hipLaunchKernelGGL(upsample8_kernel, dim3(grid_for(total / 8)), dim3(256), 0, st, in, B, H, W, wk, bias, bias_dev, s, out_f32,
                           (bf16*)out_planes);
        return;
    }
    hipLaunchKernelGGL(upsample_kernel, dim3(grid_for(total)), dim3(256), 0, st, in, B, H, W, wk, bias, bias_dev, s,
                       out_f32, (bf16*)out_planes);
}
void fwn_launch_split(const float* x, long B, long T, float* planes, hipStream_t st) {
    hipLaunchKernelGGL(split_kernel, dim3(grid_for(B * T)), dim3(256), 0, st, x, B, T, planes);
}
void fwn_launch_merge(const float* planes, long B, long T, float* x, hipStream_t st) {
    hipLaunchKernelGGL(merge_kernel, dim3(grid_for(B * T)), dim3(256), 0, st, planes, B, T, x);
}
void fwn_launch_ddi_moments(const float* xa, const float* xb, int M, int Ch, double* mom, hipStream_t st) {
    hipLaunchKernelGGL(ddi_moments_kernel, dim3(2 * Ch), dim3(256), 0, st, xa, xb, M, Ch, mom);
}
void fwn_launch_ddi_from_moments(const double* mom, int Ch, float* an, hipStream_t st) {
    hipLaunchKernelGGL(ddi_from_moments_kernel, dim3((2 * Ch + 63) / 64), dim3(64), 0, st, mom, Ch, an);
}
void fwn_launch_ddi(const float* xa, const float* xb, int M, int Ch, float* an, hipStream_t st) {
    hipLaunchKernelGGL(ddi_kernel, dim3(2 * Ch), dim3(256), 0, st, xa, xb, M, Ch, an);
}
void fwn_launch_prior(const float* planes, long n, const float* partial, int n_partial, double inv_bt,
                      float* out2, hipStream_t st) {
    hipLaunchKernelGGL(prior_kernel, dim3(1), dim3(1024), 0, st, planes, n, partial, n_partial, inv_bt, out2);
}

// ---- mel front-end: wav -> normalised log-mel frames (preprocessing.py:58-69) --------------------
// One workgroup per (frame, clip): centred reflect-padded frame x periodic Hann window -> direct
// DFT (the window is 1024 points and a 10 s clip has 862 frames: an FFT would buy microseconds)
// -> |.|^2 -> mel filterbank -> 20 log10(max(1e-4, .)) - ref -> clip((. - min) / -min, 0, 1).
// The 20 log10 of a POWER spectrum is the reference's (preprocessing.py:67) and kept.
__global__ __launch_bounds__(256) void mel_kernel(const float* __restrict__ wav, long T, int frames,
                                                  const float* __restrict__ window, const float* __restrict__ fb,
                                                  int n_fft, int hop, int n_mels, float ref_db, float min_db,
                                                  float* __restrict__ mel) {
    extern __shared__ float sm[];
    float* xw = sm;                    // [n_fft] windowed frame
    float* cs = xw + n_fft;            // [n_fft] cos(2 pi j / n_fft)
    float* sn = cs + n_fft;            // [n_fft] sin(2 pi j / n_fft)
    float* pw = sn + n_fft;            // [n_fft/2 + 1] power spectrum
    const int fr = blockIdx.x, nb = n_fft / 2 + 1;
    const float* y = wav + (long)blockIdx.y * T;
    for (int n = threadIdx.x; n < n_fft; n += 256) {
        long j = (long)fr * hop + n - n_fft / 2;
        if (j < 0) j = -j;                         // numpy 'reflect': the edge sample is not repeated
        if (j >= T) j = 2 * (T - 1) - j;
        xw[n] = y[j] * window[n];
        float sv, cv;
        sincospif(2.0f * (float)n / (float)n_fft, &sv, &cv);
        cs[n] = cv;
        sn[n] = sv;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nb; k += 256) {
        float re = 0.0f, im = 0.0f;
        int idx = 0;
        for (int n = 0; n < n_fft; ++n) {
            re = fmaf(xw[n], cs[idx], re);
            im = fmaf(-xw[n], sn[idx], im);
            idx = (idx + k) & (n_fft - 1);
        }
        pw[k] = re * re + im * im;
    }
    __syncthreads();
    for (int m = threadIdx.x; m < n_mels; m += 256) {
        const float* w = fb + (long)m * nb;
        float acc = 0.0f;
        for (int k = 0; k < nb; ++k) acc = fmaf(w[k], pw[k], acc);
        const float db = 20.0f * log10f(fmaxf(1e-4f, acc)) - ref_db;
        mel[((long)blockIdx.y * frames + fr) * n_mels + m] = fminf(fmaxf((db - min_db) / (-min_db), 0.0f), 1.0f);
    }
}
void fwn_launch_mel(const float* wav, long B, long T, const float* window, const float* fb, int n_fft, int hop,
                    int n_mels, float ref_db, float min_db, float* mel, hipStream_t st) {
    const int frames = (int)(1 + T / hop);
    const size_t lds = (size_t)(3 * n_fft + n_fft / 2 + 1) * sizeof(float);
    hipLaunchKernelGGL(mel_kernel, dim3(frames, (unsigned)B), dim3(256), lds, st, wav, T, frames, window, fb,
                       n_fft, hop, n_mels, ref_db, min_db, mel);
}

int fwn_sqnorm_blocks(long n) { long b = (n + 1023) / 1024; return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b)); }
void fwn_launch_grad_norm(const float* g, long n, float gscale, double* partial, float* out, hipStream_t st) {
    const int nb = fwn_sqnorm_blocks(n);
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nb), dim3(256), 0, st, g, n, partial);
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, st, partial, nb, gscale, out);
}
void fwn_launch_adam(float* w, const float* g, float* m, float* v, long n, const float* gnorm, float gscale,
                     float clip, float lr_t, const float* lr_dev, float b1, float b2, float eps, hipStream_t st) {
    hipLaunchKernelGGL(adam_clip_kernel, dim3(fwn_sqnorm_blocks(n)), dim3(256), 0, st, w, g, m, v, n, gnorm, gscale,
                       clip, lr_t, lr_dev, b1, b2, eps);
}
