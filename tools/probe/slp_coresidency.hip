// Reproducer for DESIGN.md section 3.5's open item (VERDICT r5 task 6): packed fp32 math (what hipcc's SLP vectoriser makes of
// scalar `acc += y * w` code) returning wrong results only while ANOTHER kernel shares the CU.
// Kernel P<MODE>: every lane runs one packed fp32 instruction per iteration on register pairs and checks both halves of the result
// against scalar v_mul_f32 / v_fma_f32 / v_add_f32 on the same registers, in the kernel (the scalar pair is also computed twice and
// compared with itself: the control).  Mismatches are counted per half and per 16-lane group; the first few are recorded.
//   MODE 0  v_pk_mul_f32 op_sel:[0,1]   (low = a.lo * b.hi, high = a.hi * b.hi: the broadcast form of the vectorised front conv)
//   MODE 1  v_pk_mul_f32                (low = a.lo * b.lo, high = a.hi * b.hi)
//   MODE 2  v_pk_fma_f32 op_sel:[0,1,0]
//   MODE 3  v_pk_fma_f32
//   MODE 4  v_pk_add_f32
// Kernel M<MFMA, DMA>: a loop in 64 KB of LDS (two workgroups per CU: room for P's workgroups beside them) of, optionally, MFMAs fed
// from LDS and, optionally, LDS-DMA loads; with neither it only reads LDS.
// P runs alone, beside each M variant on a second stream, and beside a second P.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/slp_coresidency.hip -o /tmp/slp && /tmp/slp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned long long u64;

__device__ inline unsigned fbits(float f) { return __builtin_bit_cast(unsigned, f); }
__device__ inline u64 pair(float lo, float hi) { return (u64)fbits(lo) | ((u64)fbits(hi) << 32); }

// op: 0 v_pk_mul_f32, 1 v_pk_fma_f32, 2 v_pk_add_f32, 3 v_pk_mul_f16; the four selects as in the instruction's op_sel / op_sel_hi
template <int MODE> struct Mode;
#define MODE_DEF(m, o, a, b, c, d) template <> struct Mode<m> { static constexpr int op = o, sla = a, slb = b, sha = c, shb = d; }
MODE_DEF(0, 0, 0, 1, 1, 1);   // v_pk_mul_f32 op_sel:[0,1]            low takes b.hi   (the vectorised front conv's broadcast)
MODE_DEF(1, 0, 0, 0, 1, 1);   // v_pk_mul_f32                         no swizzle
MODE_DEF(2, 1, 0, 1, 1, 1);   // v_pk_fma_f32 op_sel:[0,1,0]
MODE_DEF(3, 1, 0, 0, 1, 1);   // v_pk_fma_f32
MODE_DEF(4, 2, 0, 0, 1, 1);   // v_pk_add_f32
MODE_DEF(5, 0, 1, 0, 1, 1);   // v_pk_mul_f32 op_sel:[1,0]            low takes a.hi
MODE_DEF(6, 0, 0, 0, 1, 0);   // v_pk_mul_f32 op_sel_hi:[1,0]         high takes b.lo  (what hipcc emits for `pair * scalar`)
MODE_DEF(7, 0, 0, 0, 0, 1);   // v_pk_mul_f32 op_sel_hi:[0,1]         high takes a.lo
MODE_DEF(8, 2, 0, 1, 1, 1);   // v_pk_add_f32 op_sel:[0,1]
MODE_DEF(9, 0, 1, 1, 0, 0);   // v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[0,0]   both halves swapped
MODE_DEF(10, 3, 0, 1, 1, 1);  // v_pk_mul_f16 op_sel:[0,1]
MODE_DEF(11, 4, 0, 1, 1, 1);  // v_pk_fma_f32 op_sel:[0,0,1]          low takes c.hi
MODE_DEF(12, 5, 0, 1, 1, 1);  // v_pk_mov_b32 op_sel:[0,1]            (a.lo, b.hi)
MODE_DEF(13, 5, 1, 0, 1, 1);  // v_pk_mov_b32 op_sel:[1,0]            (a.hi, b.lo)
struct Stats { unsigned lo, hi, ctl, grp[4], nsample; unsigned sample[16][12]; };

template <int MODE>
__global__ __launch_bounds__(256) void pk_kernel(const float* __restrict__ in, Stats* __restrict__ st, int iters) {
    __shared__ float sh[4096];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) sh[i] = in[(blockIdx.x * 131 + i) & 65535];
    __syncthreads();
    unsigned nlo = 0, nhi = 0, nctl = 0;
    float c0 = 0.f, c1 = 0.f;
    for (int it = 0; it < iters; ++it) {
        const float ax = sh[(it * 2) & 4095], ay = sh[(it * 2 + 1) & 4095];                        // wave-uniform (LDS broadcast)
        const float bx = sh[(lane * 2 + it * 64) & 4095], by = sh[(lane * 2 + 1 + it * 64) & 4095];  // per lane
        const u64 a = pair(ax, ay), b = pair(bx, by), c = pair(c0, c1);
        u64 p;
        float s0, s1, r0, r1;
        // SEL_LO_A / SEL_LO_B: which half of a / b feeds the LOW result; SEL_HI_A / SEL_HI_B: which feeds the HIGH result
        constexpr int SLA = Mode<MODE>::sla, SLB = Mode<MODE>::slb, SHA = Mode<MODE>::sha, SHB = Mode<MODE>::shb;
        const float la = SLA ? ay : ax, lb = SLB ? by : bx, ha = SHA ? ay : ax, hb = SHB ? by : bx;
        if constexpr (Mode<MODE>::op == 0) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[%3,%4] op_sel_hi:[%5,%6]" : "=&v"(p) : "v"(a), "v"(b), "n"(SLA), "n"(SLB), "n"(SHA), "n"(SHB));
            asm volatile("v_mul_f32 %0, %1, %2" : "=&v"(s0) : "v"(la), "v"(lb));
            asm volatile("v_mul_f32 %0, %1, %2" : "=&v"(s1) : "v"(ha), "v"(hb));
            asm volatile("v_mul_f32 %0, %1, %2" : "=&v"(r0) : "v"(la), "v"(lb));
            asm volatile("v_mul_f32 %0, %1, %2" : "=&v"(r1) : "v"(ha), "v"(hb));
        } else if constexpr (Mode<MODE>::op == 1) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[%4,%5,0] op_sel_hi:[%6,%7,1]" : "=&v"(p) : "v"(a), "v"(b), "v"(c), "n"(SLA), "n"(SLB), "n"(SHA), "n"(SHB));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(s0) : "v"(la), "v"(lb), "v"(c0));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(s1) : "v"(ha), "v"(hb), "v"(c1));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(r0) : "v"(la), "v"(lb), "v"(c0));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(r1) : "v"(ha), "v"(hb), "v"(c1));
        } else if constexpr (Mode<MODE>::op == 2) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[%3,%4] op_sel_hi:[%5,%6]" : "=&v"(p) : "v"(a), "v"(b), "n"(SLA), "n"(SLB), "n"(SHA), "n"(SHB));
            asm volatile("v_add_f32 %0, %1, %2" : "=&v"(s0) : "v"(la), "v"(lb));
            asm volatile("v_add_f32 %0, %1, %2" : "=&v"(s1) : "v"(ha), "v"(hb));
            asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r0) : "v"(la), "v"(lb));
            asm volatile("v_add_f32 %0, %1, %2" : "=&v"(r1) : "v"(ha), "v"(hb));
        } else if constexpr (Mode<MODE>::op == 4) {   // the addend swizzled instead: low takes c.hi when SLB
            const float lc = SLB ? c1 : c0;
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,%4] op_sel_hi:[1,1,1]" : "=&v"(p) : "v"(a), "v"(b), "v"(c), "n"(SLB));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(s0) : "v"(ax), "v"(bx), "v"(lc));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(s1) : "v"(ay), "v"(by), "v"(c1));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(r0) : "v"(ax), "v"(bx), "v"(lc));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=&v"(r1) : "v"(ay), "v"(by), "v"(c1));
        } else if constexpr (Mode<MODE>::op == 5) {   // v_pk_mov_b32: low = a.(SLA ? hi : lo), high = b.(SLB ? hi : lo)
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[%3,%4]" : "=&v"(p) : "v"(a), "v"(b), "n"(SLA), "n"(SLB));
            s0 = la; s1 = lb; r0 = s0; r1 = s1;
        } else {  // the 16-bit packed family for comparison: v_pk_mul_f16 on the same register pair's LOW dword only
            unsigned p16, q16;
            asm volatile("v_pk_mul_f16 %0, %1, %2 op_sel:[%3,%4] op_sel_hi:[%5,%6]" : "=&v"(p16) : "v"(fbits(ax)), "v"(fbits(bx)), "n"(SLA), "n"(SLB), "n"(SHA), "n"(SHB));
            asm volatile("v_pk_mul_f16 %0, %1, %2 op_sel:[%3,%4] op_sel_hi:[%5,%6]" : "=&v"(q16) : "v"(fbits(ax)), "v"(fbits(bx)), "n"(SLA), "n"(SLB), "n"(SHA), "n"(SHB));
            p = (u64)(p16 & 0xffffu) | ((u64)(p16 >> 16) << 32);
            s0 = __builtin_bit_cast(float, q16 & 0xffffu); s1 = __builtin_bit_cast(float, q16 >> 16); r0 = s0; r1 = s1;
        }
        const unsigned plo = (unsigned)p, phi = (unsigned)(p >> 32);
        const bool lo_bad = plo != fbits(s0), hi_bad = phi != fbits(s1), ctl_bad = fbits(s0) != fbits(r0) || fbits(s1) != fbits(r1);
        nlo += lo_bad; nhi += hi_bad; nctl += ctl_bad;
        if ((lo_bad || hi_bad) && nlo + nhi <= 1) {
            const unsigned k = atomicAdd(&st->nsample, 1u);
            if (k < 16) {
                unsigned* d = st->sample[k];
                d[0] = fbits(ax); d[1] = fbits(ay); d[2] = fbits(bx); d[3] = fbits(by); d[4] = fbits(c0); d[5] = fbits(c1);
                d[6] = plo; d[7] = phi; d[8] = fbits(s0); d[9] = fbits(s1); d[10] = lane; d[11] = it;
            }
        }
        c0 = s0 * 0.5f; c1 = s1 * 0.5f;
    }
    if (nlo) atomicAdd(&st->lo, nlo);
    if (nhi) atomicAdd(&st->hi, nhi);
    if (nctl) atomicAdd(&st->ctl, nctl);
    if (nlo + nhi) atomicAdd(&st->grp[lane >> 4], nlo + nhi);
}

template <int MFMA, bool DMA>
__global__ __launch_bounds__(256) void m_kernel(const __bf16* __restrict__ A, float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 1u << 24, 0x00020000);
    for (int it = 0; it < iters; ++it) {
        if constexpr (DMA)
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(lds + ((it & 3) * 4 + j) * 4096 + wave * 1024),
                                                         16, (unsigned)(((it * 4 + j) * 4096 + wave * 1024 + lane * 16) & 0xffffff), 0, 0, 0);
        __syncthreads();
        for (int k = 0; k < 8; ++k) {
            const bf16x8 a = *(const bf16x8*)(lds + ((k * 1024 + wave * 256 + lane * 16) & 65535));
            const bf16x8 b = *(const bf16x8*)(lds + ((k * 1024 + 32768 + lane * 16) & 65535));
            if constexpr (MFMA == 1) {
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            } else if constexpr (MFMA == 2) {   // the fp32 matrix instruction
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a[0], (float)b[0], acc[i], 0, 0, 0);
            } else if constexpr (MFMA == 3) {   // the small shape: 16 x 16 x 32, 4 accumulator registers
                for (int i = 0; i < 4; ++i) {
                    f32x4 t = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
                    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, t, 0, 0, 0);
                    acc[i][0] = t[0]; acc[i][1] = t[1]; acc[i][2] = t[2]; acc[i][3] = t[3];
                }
            } else {
                for (int i = 0; i < 4; ++i) acc[i][k] += (float)a[i] * (float)b[i];
            }
        }
        __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + tid] = s;
}

static float* in; static Stats* st; static __bf16* A; static float* out; static hipStream_t s1, s2;

static void report(const char* mode, const char* what) {
    Stats h; CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, st, sizeof h, hipMemcpyDeviceToHost));
    printf("%-30s %-36s low %9u  high %9u  scalar-vs-scalar %u   by 16-lane group %u %u %u %u\n", mode, what, h.lo, h.hi, h.ctl, h.grp[0], h.grp[1], h.grp[2], h.grp[3]);
    for (unsigned k = 0; k < h.nsample && k < 3; ++k) {
        const unsigned* d = h.sample[k];
        float f[10]; memcpy(f, d, 40);
        printf("      lane %2u it %4u  a (%.9g, %.9g) b (%.9g, %.9g) c (%.9g, %.9g)  packed (%.9g, %.9g) [%08x %08x]  scalar (%.9g, %.9g) [%08x %08x]\n",
               d[10], d[11], f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], d[6], d[7], f[8], f[9], d[8], d[9]);
    }
    CK(hipMemset(st, 0, sizeof(Stats)));
}

template <int MODE> static void run_mode(const char* mode, bool full) {
    const int reps = 10, iters = 2000;
    auto P = [&](hipStream_t s) { hipLaunchKernelGGL(pk_kernel<MODE>, dim3(1024), dim3(256), 0, s, in, st, iters); };
    for (int r = 0; r < reps; ++r) P(s1);
    report(mode, "alone");
    for (int r = 0; r < reps; ++r) { hipLaunchKernelGGL((m_kernel<1, true>), dim3(2048), dim3(256), 0, s2, A, out, 60); P(s1); }
    report(mode, "beside MFMA 32x32x16 + LDS-DMA");
    if (!full) return;
    for (int r = 0; r < reps; ++r) { hipLaunchKernelGGL((m_kernel<1, false>), dim3(2048), dim3(256), 0, s2, A, out, 60); P(s1); }
    report(mode, "beside MFMA 32x32x16, no DMA");
    for (int r = 0; r < reps; ++r) { hipLaunchKernelGGL((m_kernel<3, true>), dim3(2048), dim3(256), 0, s2, A, out, 60); P(s1); }
    report(mode, "beside MFMA 16x16x32 + LDS-DMA");
    for (int r = 0; r < reps; ++r) { hipLaunchKernelGGL((m_kernel<2, true>), dim3(2048), dim3(256), 0, s2, A, out, 60); P(s1); }
    report(mode, "beside MFMA 32x32x2 f32 + LDS-DMA");
    for (int r = 0; r < reps; ++r) { hipLaunchKernelGGL((m_kernel<0, true>), dim3(2048), dim3(256), 0, s2, A, out, 60); P(s1); }
    report(mode, "beside LDS-DMA, no MFMA");
    for (int r = 0; r < reps; ++r) { hipLaunchKernelGGL((m_kernel<0, false>), dim3(2048), dim3(256), 0, s2, A, out, 60); P(s1); }
    report(mode, "beside LDS reads + VALU only");
    for (int r = 0; r < reps; ++r) { P(s2); P(s1); }
    report(mode, "beside a second instance");
}

int main() {
    CK(hipMalloc(&in, 65536 * 4)); CK(hipMalloc(&st, sizeof(Stats))); CK(hipMalloc(&A, 1u << 24)); CK(hipMalloc(&out, 4096 * 256 * 4));
    std::vector<float> h(65536);
    srand(7);
    for (auto& v : h) v = (rand() & 0xffff) / 32768.0f - 1.0f;
    CK(hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice));
    CK(hipMemset(A, 0x3c, 1u << 24)); CK(hipMemset(st, 0, sizeof(Stats)));
    CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    run_mode<0>("v_pk_mul_f32 op_sel:[0,1]", true);
    run_mode<1>("v_pk_mul_f32", false);
    run_mode<2>("v_pk_fma_f32 op_sel:[0,1,0]", false);
    run_mode<3>("v_pk_fma_f32", false);
    run_mode<4>("v_pk_add_f32", false);
    run_mode<5>("v_pk_mul_f32 op_sel:[1,0]", false);
    run_mode<6>("v_pk_mul_f32 op_sel_hi:[1,0]", false);
    run_mode<7>("v_pk_mul_f32 op_sel_hi:[0,1]", false);
    run_mode<8>("v_pk_add_f32 op_sel:[0,1]", false);
    run_mode<9>("v_pk_mul_f32 [1,1] hi [0,0]", false);
    run_mode<10>("v_pk_mul_f16 op_sel:[0,1]", false);
    run_mode<11>("v_pk_fma_f32 op_sel:[0,0,1]", false);
    run_mode<12>("v_pk_mov_b32 op_sel:[0,1]", false);
    run_mode<13>("v_pk_mov_b32 op_sel:[1,0]", false);
    return 0;
}
