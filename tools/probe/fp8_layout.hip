// Probe (GPU box): operand / scale layout of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands, exact small-integer data.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/fp8_layout.hip -o /tmp/fp8_layout && /tmp/fp8_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// e4m3fn encode of small non-negative integers / simple values (exact): sign 0, bias 7
static uint8_t e4m3(float v) {
    if (v == 0.f) return 0;
    uint8_t s = v < 0 ? 0x80 : 0; v = fabsf(v);
    int e; float m = frexpf(v, &e);          // v = m * 2^e, m in [0.5,1)
    int E = e - 1 + 7; float frac = m * 2 - 1; // 1.frac
    int mant = (int)lrintf(frac * 8);
    if (mant == 8) { mant = 0; ++E; }
    if (E <= 0) { int sub = (int)lrintf(v / ldexpf(1.f, -9)); return s | (uint8_t)sub; }
    return s | (uint8_t)(E << 3) | (uint8_t)mant;
}
static float e4m3_to_f(uint8_t b) {
    int s = b >> 7, E = (b >> 3) & 15, m = b & 7;
    float v = E == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, E - 7);
    return s ? -v : v;
}

// A [32][64] bytes row-major, B [32 cols][64 k] row-major (B^T input), out [32][32]
// layout hypothesis: lane l: row/col = l & 31, k = 32 (l >> 5) + byte index 0..31 of the 8-dword fragment
__global__ void k(const uint8_t* A, const uint8_t* B, float* out, int sa, int sb) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = *(const int*)(A + r * 64 + 32 * h + 4 * i);
        b[i] = *(const int*)(B + r * 64 + 32 * h + 4 * i);
    }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int i = 0; i < 16; ++i) out[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
}

int main() {
    std::vector<uint8_t> A(32 * 64), B(32 * 64);
    std::vector<float> Af(32 * 64), Bf(32 * 64);
    srand(1);
    for (int i = 0; i < 32 * 64; ++i) {
        float va = (float)((rand() % 9) - 4) * 0.5f, vb = (float)((rand() % 7) - 3);
        A[i] = e4m3(va); B[i] = e4m3(vb); Af[i] = e4m3_to_f(A[i]); Bf[i] = e4m3_to_f(B[i]);
        if (Af[i] != va || Bf[i] != vb) { printf("encode bug %f %f\n", va, Af[i]); return 1; }
    }
    uint8_t *dA, *dB; float* dO;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dO, 32 * 32 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    for (int sa : {127, 128, 126}) for (int sb : {127, 125}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dO, sa, sb);
        std::vector<float> o(32 * 32);
        hipMemcpy(o.data(), dO, o.size() * 4, hipMemcpyDeviceToHost);
        double maxerr = 0, ratio = 0; int cnt = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double ref = 0; for (int kk = 0; kk < 64; ++kk) ref += (double)Af[i * 64 + kk] * Bf[j * 64 + kk];
            double want = ref * ldexp(1.0, sa - 127) * ldexp(1.0, sb - 127);
            maxerr = fmax(maxerr, fabs(o[i * 32 + j] - want));
            if (fabs(ref) > 1) { ratio += o[i * 32 + j] / ref; ++cnt; }
        }
        printf("scale_a %d scale_b %d: max |err| vs (A B^T) 2^(sa-127) 2^(sb-127) = %g   mean out/ref = %g\n", sa, sb, maxerr, ratio / cnt);
    }
    return 0;
}
