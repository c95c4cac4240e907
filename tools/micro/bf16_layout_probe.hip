// Developer probe (groundwork for DESIGN.md section 10): operand / result fragment layout of v_mfma_f32_16x16x32_bf16 on
// gfx950, checked against a host matmul, and the error of the 6-product BF16 split of an FP32 GEMM on the device next to
// the FP32 MFMA chain.  Assumed layout (confirmed if "layout check" prints max |diff| = 0):
//   A[16 x 32]: lane l holds row i = l % 16, k = 8 (l / 16) .. 8 (l / 16) + 7   (8 bf16 = 4 VGPRs, k ascending)
//   B[32 x 16]: lane l holds column j = l % 16, k = 8 (l / 16) .. + 7
//   D[16 x 16]: lane l holds column j = l % 16, rows 4 (l / 16) + r, r = 0..3   (as the FP32 16x16x4 instruction)
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/bf16_layout_probe.hip -o tools/micro/bf16_layout_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned short f2bf(float x) {          // round to nearest even
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ inline float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

union Frag { bf16x8 v; unsigned short h[8]; };

// D = A B for A [16 x K], B [K x 16] (row-major floats), K a multiple of 32: terms = 1 (plain bf16), 3 or 6 split products
__global__ void k_probe(const float* A, const float* B, float* D, float* D32, int K, int terms) {
    const int l = threadIdx.x, i16 = l & 15, kk = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc32 = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 32) {
        Frag a[3], b[3];
        for (int t = 0; t < 8; ++t) {
            float x = A[i16 * K + k0 + 8 * kk + t], y = B[(k0 + 8 * kk + t) * 16 + i16];
            for (int s = 0; s < 3; ++s) {
                a[s].h[t] = f2bf(x); x -= bf2f(a[s].h[t]);
                b[s].h[t] = f2bf(y); y -= bf2f(b[s].h[t]);
            }
        }
        // smallest products first; (s, t) = split index of A, of B
        const int order[6][2] = {{1, 1}, {2, 0}, {0, 2}, {1, 0}, {0, 1}, {0, 0}};
        for (int p = 6 - terms; p < 6; ++p)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[order[p][0]].v, b[order[p][1]].v, acc, 0, 0, 0);
    }
    for (int k0 = 0; k0 < K; k0 += 4)      // the FP32 chain the kernels use today
        acc32 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i16 * K + k0 + kk], B[(k0 + kk) * 16 + i16], acc32, 0, 0, 0);
    for (int r = 0; r < 4; ++r) {
        D[(4 * kk + r) * 16 + i16] = acc[r];
        D32[(4 * kk + r) * 16 + i16] = acc32[r];
    }
}

int main() {
    const int K = 64;
    std::vector<float> A(16 * K), B(K * 16), D(256), D32(256);
    srand(1);
    auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    float *dA, *dB, *dD, *dD32;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024); hipMalloc(&dD32, 1024);
    for (int pass = 0; pass < 2; ++pass) {
        // pass 0: values exactly representable in bf16 (small integers): any layout error shows up as a non-zero difference
        for (auto& x : A) x = pass == 0 ? (float)(rand() % 17 - 8) : tanhf(2.f * rnd());
        for (auto& x : B) x = pass == 0 ? (float)(rand() % 9 - 4) : rnd() / 8.f;
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        std::vector<double> ref(256, 0.0);
        double scale = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                for (int k = 0; k < K; ++k) ref[i * 16 + j] += (double)A[i * K + k] * B[k * 16 + j];
                scale = fmax(scale, fabs(ref[i * 16 + j]));
            }
        const int nterms[3] = {1, 3, 6};
        for (int v = 0; v < 3; ++v) {
            k_probe<<<1, 64>>>(dA, dB, dD, dD32, K, nterms[v]);
            hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
            hipMemcpy(D32.data(), dD32, 1024, hipMemcpyDeviceToHost);
            double e = 0, e32 = 0;
            for (int i = 0; i < 256; ++i) { e = fmax(e, fabs(D[i] - ref[i])); e32 = fmax(e32, fabs(D32[i] - ref[i])); }
            if (pass == 0) printf("layout check (%d product%s, exact inputs): max |diff| = %g   (fp32 mfma: %g)\n", nterms[v], nterms[v] > 1 ? "s" : "", e, e32);
            else printf("tanh x weights, K = %d, %d bf16 product%s: max |err| / max |ref| = %.2e   (fp32 mfma chain: %.2e)\n", K, nterms[v], nterms[v] > 1 ? "s" : "", e / scale, e32 / scale);
        }
    }
    return 0;
}
