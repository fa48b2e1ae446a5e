// Probe (GPU box): can one workgroup's LDS-DMA (buffer_load ... lds) damage the LDS of ANOTHER workgroup on the same CU?
// DESIGN.md section 3.5 records two silent-corruption findings that only appeared when a second kernel shared the CU and
// were worked around by giving the victim the whole LDS.  This separates the candidates:
//   mode 0 (control)  : the neighbour's DMA pieces land inside its own allocation and are waited for before it exits;
//   mode 1 (exit)     : the neighbour issues its pieces and ends WITHOUT waiting (s_endpgm with vmcnt outstanding) - does the
//                       hardware hold the LDS allocation until they have landed, or can they land in the next workgroup's LDS?
//   mode 2 (overrun)  : the neighbour's pieces target LDS addresses past the end of its own allocation (a wrong M0) - are
//                       LDS-DMA writes range-checked against the allocation like ds_write is?
// The victim fills its LDS with a pattern by ds_write, keeps re-reading it for ~1 ms and counts words that changed.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/lds_dma_coresidency.hip -o tools/probe/lds_dma_coresidency_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int VICTIM_WORDS = 48 * 1024 / 4;
constexpr int NEIGH_BYTES = 32 * 1024;

__global__ __launch_bounds__(256) void victim(unsigned* report, int spins) {
    __shared__ unsigned lds[VICTIM_WORDS];
    for (int i = threadIdx.x; i < VICTIM_WORDS; i += 256) lds[i] = 0xA5000000u ^ (unsigned)(i * 2654435761u) ^ blockIdx.x;
    __syncthreads();
    unsigned bad = 0, first = 0xffffffffu, seen = 0;
    for (int s = 0; s < spins; ++s) {
        for (int i = threadIdx.x; i < VICTIM_WORDS; i += 256) {
            const unsigned v = ((volatile unsigned*)lds)[i];
            if (v != (0xA5000000u ^ (unsigned)(i * 2654435761u) ^ blockIdx.x)) {
                ++bad;
                if (first == 0xffffffffu) { first = (unsigned)i; seen = v; }
            }
        }
        __builtin_amdgcn_s_sleep(20);
    }
    if (bad) {
        atomicAdd(&report[0], bad);
        atomicAdd(&report[1], 1u);
        report[2] = first;
        report[3] = seen;
    }
}

// 16 pieces of 1 KiB per wave = 64 KiB per workgroup of 4 waves in flight, into a 32 KiB allocation (each wave 8 KiB twice)
template <int MODE>
__global__ __launch_bounds__(256) void neighbour(const unsigned* src, unsigned bytes, unsigned* sink) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NEIGH_BYTES];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        // mode 2: the destination starts 8 KiB past the end of the allocation
        unsigned char* dst = lds + (MODE == 2 ? NEIGH_BYTES + 8192 : 0) + ((wave * 8 + (j & 7)) * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, (unsigned)((blockIdx.x * 64 + j * 4 + wave) * 1024 + lane * 16) % (bytes - 1024), 0, 0, 0);
    }
    if (MODE == 1) return;                      // leave with the pieces in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE == 0 && sink && lds[threadIdx.x * 16] == 0x77 && blockIdx.x == 0x7fffffff) sink[0] = 1;   // keep the LDS live
}

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    const unsigned bytes = 64u << 20;
    unsigned* src; CK(hipMalloc(&src, bytes));
    std::vector<unsigned> h(bytes / 4, 0xDEADBEEFu);
    CK(hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice));
    unsigned* report; CK(hipMalloc(&report, 64));
    for (int mode = 0; mode < 3; ++mode) {
        if (only >= 0 && mode != only) continue;
        unsigned tot[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 20; ++rep) {
            CK(hipMemset(report, 0, 64));
            hipLaunchKernelGGL(victim, dim3(512), dim3(256), 0, s0, report, 400);
            for (int k = 0; k < 200; ++k) {
                if (mode == 0) hipLaunchKernelGGL(neighbour<0>, dim3(1024), dim3(256), 0, s1, src, bytes, report + 8);
                if (mode == 1) hipLaunchKernelGGL(neighbour<1>, dim3(1024), dim3(256), 0, s1, src, bytes, report + 8);
                if (mode == 2) hipLaunchKernelGGL(neighbour<2>, dim3(1024), dim3(256), 0, s1, src, bytes, report + 8);
            }
            CK(hipDeviceSynchronize());
            unsigned r[4]; CK(hipMemcpy(r, report, 16, hipMemcpyDeviceToHost));
            tot[0] += r[0]; tot[1] += r[1];
            if (r[0]) { tot[2] = r[2]; tot[3] = r[3]; }
        }
        printf("mode %d (%s): %u damaged word reads in %u victim workgroups of %d", mode,
               mode == 0 ? "control: in range, waited" : mode == 1 ? "exit with pieces in flight" : "destination past the allocation",
               tot[0], tot[1], 20 * 512);
        if (tot[0]) printf("; e.g. word %u read %08x", tot[2], tot[3]);
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
