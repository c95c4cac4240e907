// Developer probe (round 6): can the float32-equivalent GEMMs of the pass kernels move from a three-term BF16 split (six matrix
// instructions per product, 5.5 vector instructions per value) to a TWO-term FP16 split (three instructions, 2 per value)?
//   1. v_mfma_f32_16x16x32_f16 / 32x32x16_f16 with SUBNORMAL f16 inputs: kept or flushed?  (the low term of a value below 2^-3 is
//      subnormal; flushed, the split is worth 2^-14 instead of 2^-25 absolute: tools/micro/f16_split_accuracy.py)
//   2. v_cvt_pk_f16_f32 on subnormal results (MODE.fp_denorm), v_fma_mix_f32 as the residual x - hi in one instruction
//   3. accuracy of hi.hi + hi.lo + lo.hi on the device against float64, on the policy passes' operands
//   4. issue rates: the matrix instruction alone, with the 2-term split of two values (4 VALU) in its shadow, the split alone, next
//      to the BF16 instruction with its 11-instruction split
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/f16_split_probe.hip -o /tmp/f16_split_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// hi = f16(x0, x1) packed; lo = f16(x - hi) packed: v_cvt_pk_f16_f32, 2 x v_fma_mix_f32, v_cvt_pk_f16_f32
__device__ __forceinline__ void f16_split2_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
    float r0, r1;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(x1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(r0), "v"(r1));
}

// ---- 1 / 2: subnormals ----------------------------------------------------------------------------------------------------
__global__ void k_subnormal(float* out) {
    const int lane = threadIdx.x;
    // A[i][k]: every entry 2^-20 (an f16 subnormal: 16 ulps of 2^-24); B[k][j] = 1 -> D = 32 * 2^-20 = 2^-15 if kept, 0 if flushed
    const _Float16 sub = (_Float16)9.5367431640625e-07f, one = (_Float16)1.0f;
    h16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = sub; b[e] = one; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    f32x16 c2;
    for (int r = 0; r < 16; ++r) c2[r] = 0.f;
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    // subnormal x subnormal-scale partner: A = 2^-20, B = 2^-20: product 2^-40 (a normal float32) -> 32 * 2^-40
    h16x8 bs;
    for (int e = 0; e < 8; ++e) bs[e] = sub;
    f32x4 c3 = {0.f, 0.f, 0.f, 0.f};
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bs, c3, 0, 0, 0);
    // conversions: f16 of 3e-6 (subnormal result), residual through v_fma_mix_f32
    unsigned hi, lo;
    f16_split2_pair(3.0e-6f, 0.3f, hi, lo);
    if (lane == 0) {
        out[0] = c[0];
        out[1] = c2[0];
        out[2] = c3[0];
        out[3] = __builtin_bit_cast(float, hi);
        out[4] = __builtin_bit_cast(float, lo);
    }
}

// ---- 3: accuracy of a [16 m] x [K] x [16 n] product on the matrix pipe, every wave its own 16 x 16 block ---------------------
// A[m][k] row-major (M x K), B[k][n] row-major (K x N); lane (i16, kk) feeds A[i16][32 s + 8 kk + e], B[32 s + 8 kk + e][j = i16]
template <int MODE>     // 0: f16 two-term, three products   1: f16 hi only   2: bf16 three-term, six products
__global__ void k_gemm(const float* A, const float* B, float* D, int M, int N, int K, float sa, float sb) {
    const int lane = threadIdx.x & 63, i16 = lane & 15, kk = lane >> 4;
    const int bm = blockIdx.x, bn = blockIdx.y;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < K / 32; ++s) {
        float av[8], bv[8];
        for (int e = 0; e < 8; ++e) {
            av[e] = A[(16 * bm + i16) * K + 32 * s + 8 * kk + e] * sa;
            bv[e] = B[(32 * s + 8 * kk + e) * N + 16 * bn + i16] * sb;
        }
        if (MODE < 2) {
            u32x4 ah, al, bh, bl;
            for (int p = 0; p < 4; ++p) {
                unsigned h, l;
                f16_split2_pair(av[2 * p], av[2 * p + 1], h, l);
                ah[p] = h; al[p] = l;
                f16_split2_pair(bv[2 * p], bv[2 * p + 1], h, l);
                bh[p] = h; bl[p] = l;
            }
            if (MODE == 0) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, al), __builtin_bit_cast(h16x8, bh), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, ah), __builtin_bit_cast(h16x8, bl), acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, ah), __builtin_bit_cast(h16x8, bh), acc, 0, 0, 0);
        } else {
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            bf16x8 at[3], bt[3];
            for (int e = 0; e < 8; ++e) {
                float r = av[e];
                for (int t = 0; t < 3; ++t) { const __bf16 h = (__bf16)r; at[t][e] = h; r -= (float)h; }
                r = bv[e];
                for (int t = 0; t < 3; ++t) { const __bf16 h = (__bf16)r; bt[t][e] = h; r -= (float)h; }
            }
            for (int ta = 2; ta >= 0; --ta)
                for (int tb = 2 - ta; tb >= 0; --tb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at[ta], bt[tb], acc, 0, 0, 0);
        }
    }
    const float inv = 1.0f / (sa * sb);
    for (int r = 0; r < 4; ++r) D[(16 * bm + 4 * kk + r) * N + 16 * bn + i16] = acc[r] * inv;      // D: col = l & 15, row = 4 (l >> 4) + r
}

static double randn() {
    double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
    return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v);
}

static void accuracy(const char* name, int M, int N, int K, const std::vector<float>& A, const std::vector<float>& B, float sa, float sb) {
    float *dA, *dB, *dD;
    hipMalloc(&dA, 4 * M * K); hipMalloc(&dB, 4 * K * N); hipMalloc(&dD, 4 * M * N);
    hipMemcpy(dA, A.data(), 4 * M * K, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), 4 * K * N, hipMemcpyHostToDevice);
    std::vector<double> ref(M * N);
    double scale = 0;
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)B[k * N + j];
            ref[i * N + j] = s;
            scale = fmax(scale, fabs(s));
        }
    std::vector<float> D(M * N);
    printf("%s  (prescale %g, %g)\n", name, sa, sb);
    for (int mode = 0; mode < 3; ++mode) {
        dim3 g(M / 16, N / 16);
        if (mode == 0) k_gemm<0><<<g, 64>>>(dA, dB, dD, M, N, K, sa, sb);
        if (mode == 1) k_gemm<1><<<g, 64>>>(dA, dB, dD, M, N, K, sa, sb);
        if (mode == 2) k_gemm<2><<<g, 64>>>(dA, dB, dD, M, N, K, sa, sb);
        hipMemcpy(D.data(), dD, 4 * M * N, hipMemcpyDeviceToHost);
        double mx = 0, ss = 0;
        for (int i = 0; i < M * N; ++i) {
            const double e = fabs((double)D[i] - ref[i]);
            mx = fmax(mx, e);
            ss += e * e;
        }
        printf("  %-34s max |err| / max |ref| = %.2e   rms = %.2e\n",
               mode == 0 ? "f16 2-term, 3 products" : mode == 1 ? "f16 hi only" : "bf16 3-term, 6 products", mx / scale, sqrt(ss / (M * N)) / scale);
    }
    hipFree(dA); hipFree(dB); hipFree(dD);
}

// ---- 4: issue rates -------------------------------------------------------------------------------------------------------
enum { MF_F16, MF_BF16, MF_NONE };
template <int KIND, int SPLIT>       // SPLIT: 0 none, 1 the f16 two-term split of two values, 2 the bf16 three-term split of two values
__global__ void __launch_bounds__(512) k_rate(float* out, int iters, float seed) {
    const int tid = threadIdx.x;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
    s16x8 a8 = {(short)tid, 1, 2, 3, 4, 5, 6, 7}, b8 = {7, 6, 5, 4, 3, 2, 1, (short)tid};
    float x[4] = {seed * 0.37f, seed * 0.11f + tid, seed + 0.5f, seed - 0.25f};
    unsigned packed[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (KIND == MF_F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a8), "v"(b8));
            if (KIND == MF_BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a8), "v"(b8));
            if (SPLIT == 1) {
                float x0 = x[i], x1 = x[(i + 1) & 3], r0, r1;
                unsigned hi, lo;
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
                asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x0));
                asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(x1));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(r0), "v"(r1));
                packed[i] ^= hi ^ lo;
            }
            if (SPLIT == 2) {
                float x0 = x[i], x1 = x[(i + 1) & 3];
                unsigned hi, mid, lo;
                float h0, h1, r0, r1, m0, m1, s0, s1;
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
                asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(h0) : "v"(hi));
                asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(h1) : "v"(hi));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(x0), "v"(h0));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(x1), "v"(h1));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(mid) : "v"(r0), "v"(r1));
                asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(m0) : "v"(mid));
                asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(m1) : "v"(mid));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(s0) : "v"(r0), "v"(m0));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(s1) : "v"(r1), "v"(m1));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(s0), "v"(s1));
                packed[i] ^= hi ^ mid ^ lo;
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + (float)packed[i];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int KIND, int SPLIT>
static void rate(const char* label, int waves_per_simd) {
    const int cus = 256, iters = 20000, block = 256 * waves_per_simd;
    float* out;
    hipMalloc(&out, sizeof(float) * cus * block);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_rate<KIND, SPLIT><<<cus, block>>>(out, 100, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_rate<KIND, SPLIT><<<cus, block>>>(out, iters, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double groups = (double)iters * 4;
    printf("%-58s waves/SIMD %d  %8.3f ms  %7.2f ns per slot and SIMD  %8.1f TFLOP/s (matrix)\n", label, waves_per_simd, ms,
           ms * 1e6 / groups / waves_per_simd, KIND == MF_NONE ? 0.0 : 16384.0 * groups * waves_per_simd * 4 * cus / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    {
        float* d;
        hipMalloc(&d, 64);
        k_subnormal<<<1, 64>>>(d);
        float h[5];
        hipMemcpy(h, d, 20, hipMemcpyDeviceToHost);
        unsigned hi = *(unsigned*)&h[3], lo = *(unsigned*)&h[4];
        printf("subnormal f16 A (2^-20) x 1.0, K = 32: 16x16x32 -> %.6e, 32x32x16 (K = 16) -> %.6e   (kept: %.6e / %.6e; flushed: 0)\n", h[0], h[1],
               32 * 9.5367431640625e-07, 16 * 9.5367431640625e-07);
        printf("subnormal x subnormal (2^-40 each), K = 32 -> %.6e   (kept: %.6e)\n", h[2], 32 * 9.5367431640625e-07 * 9.5367431640625e-07);
        printf("split of (3.0e-6, 0.3): hi = %08x  lo = %08x   (hi low half: f16(3e-6) = 0x0032 if subnormal results are kept, 0 if flushed)\n", hi, lo);
    }
    srand(1);
    {
        const int M = 256, K = 64, N = 64;
        std::vector<float> X(M * 32), W1(32 * N), H(M * K), W2(K * N), W3(K * 16), HT(K * 16), DZ(16 * N), W2T(N * K), DZT(K * 16);
        for (int i = 0; i < M; ++i) for (int k = 0; k < 32; ++k) X[i * 32 + k] = k < 20 ? (float)randn() : 0.f;
        for (auto& v : W1) v = (float)(randn() / sqrt(20.0) * 2.885);
        for (auto& v : H) v = (float)tanh(randn());
        for (auto& v : W2) v = (float)(randn() / 8.0 * 2.885);
        for (int k = 0; k < K; ++k) for (int j = 0; j < 16; ++j) W3[k * 16 + j] = j < 6 ? (float)(randn() / 8.0) : 0.f;
        accuracy("layer 1: observations [256 x 32] x W1 [32 x 64]", M, N, 32, X, W1, 1.f, 1.f);
        accuracy("layer 2: tanh activations [256 x 64] x W2 [64 x 64]", M, N, K, H, W2, 1.f, 1.f);
        accuracy("output layer: [256 x 64] x W3 [64 x 16 (6 used)]", M, 16, K, H, W3, 1.f, 1.f);
        // contraction over samples: K = 32 samples here (two tiles' worth)
        std::vector<float> HT2(64 * 32), DZ2(32 * 64);
        for (auto& v : HT2) v = (float)tanh(randn());
        for (int s = 0; s < 32; ++s) { const double a = randn() / 4000.0; for (int j = 0; j < 64; ++j) DZ2[s * 64 + j] = (float)(randn() * a); }
        accuracy("weight gradient: H^T [64 x 32] x dZ [32 x 64], dZ ~ adv / N", 64, 64, 32, HT2, DZ2, 1.f, 1.f);
        accuracy("weight gradient: H^T [64 x 32] x dZ [32 x 64], dZ ~ adv / N", 64, 64, 32, HT2, DZ2, 1.f, 4096.f);
        std::vector<float> W2u(64 * 64), DZT2(64 * 16);
        for (auto& v : W2u) v = (float)(randn() / 8.0);
        for (int j = 0; j < 16; ++j) { const double a = randn() / 4000.0; for (int k = 0; k < 64; ++k) DZT2[k * 16 + j] = (float)(randn() * a); }
        accuracy("backward chain: W2 [64 x 64] x dZ2^T [64 x 16]", 64, 16, 64, W2u, DZT2, 1.f, 1.f);
        accuracy("backward chain: W2 [64 x 64] x dZ2^T [64 x 16]", 64, 16, 64, W2u, DZT2, 1.f, 4096.f);
        std::vector<float> Xb = X;
        for (auto& v : Xb) v *= 37.f;
        accuracy("layer 1, observations of size 37", M, N, 32, Xb, W1, 1.f, 1.f);
    }
    for (int w = 1; w <= 2; ++w) {
        rate<MF_F16, 0>("f16 16x16x32", w);
        rate<MF_BF16, 0>("bf16 16x16x32", w);
        rate<MF_NONE, 1>("f16 2-term split of 2 values alone (4 VALU)", w);
        rate<MF_NONE, 2>("bf16 3-term split of 2 values alone (11 VALU)", w);
        rate<MF_F16, 1>("f16 16x16x32 + f16 split of 2 values", w);
        rate<MF_BF16, 2>("bf16 16x16x32 + bf16 split of 2 values", w);
    }
    return 0;
}
