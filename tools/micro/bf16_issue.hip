// Developer microbenchmark (groundwork for DESIGN.md section 10): issue rates of the BF16 matrix instructions on gfx950
// and whether FP32 VALU work overlaps them -- it does not overlap v_mfma_f32_16x16x4_f32 (profiles/r01_mfma_valu_coissue_
// microbench.txt), which is what bounds the FP32 pass kernels.  Each variant issues 4 independent MFMAs per group, each
// followed by N v_fma_f32 (independent chains), at one and two waves per SIMD; and the cost of the error-compensated
// 3-way BF16 split of an FP32 value (cvt_pk / shift / sub chains).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/bf16_issue.hip -o tools/micro/bf16_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

enum Mfma { F32_16x16x4, BF16_16x16x16, BF16_16x16x32, NO_MFMA };

template <int KIND, int N, int SPLIT>
__global__ void __launch_bounds__(512) k(float* out, int iters, float seed) {
    const int tid = threadIdx.x;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
    float a = seed + tid, b = seed * 2 + tid;
    s16x4 a4 = {(short)tid, 1, 2, 3}, b4 = {3, 2, 1, (short)tid};
    s16x8 a8 = {(short)tid, 1, 2, 3, 4, 5, 6, 7}, b8 = {7, 6, 5, 4, 3, 2, 1, (short)tid};
    float v[4] = {seed, seed + 1, seed + 2, seed + 3};
    float x[4] = {seed * 0.37f, seed * 0.11f + tid, seed + 0.5f, seed - 0.25f};
    unsigned packed[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (KIND == F32_16x16x4) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (KIND == BF16_16x16x16) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a4), "v"(b4));
            if (KIND == BF16_16x16x32) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a8), "v"(b8));
#pragma unroll
            for (int u = 0; u < N; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[u & 3]) : "v"(1.0001f), "v"(0.5f));
#pragma unroll
            for (int u = 0; u < SPLIT; ++u) {
                // x -> (hi, mid, lo) as three bf16: two elements per v_cvt_pk_bf16_f32; residuals by shift + subtract
                float x0 = x[u & 3], x1 = x[(u + 1) & 3];
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
                packed[u & 3] ^= hi ^ mid ^ lo;
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i] + (float)packed[i];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int KIND, int N, int SPLIT>
void run(const char* label, int waves_per_simd, double flop_per_mfma) {
    const int cus = 256, iters = 20000;
    const int block = 256 * waves_per_simd;          // 4 SIMDs x waves_per_simd waves
    float* out;
    hipMalloc(&out, sizeof(float) * cus * block);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND, N, SPLIT><<<cus, block>>>(out, 100, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND, N, SPLIT><<<cus, block>>>(out, iters, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double groups = (double)iters * 4;                         // MFMA slots per wave
    const double ns_per_slot = ms * 1e6 / groups / waves_per_simd;   // per SIMD
    const double tf = KIND == NO_MFMA ? 0.0 : flop_per_mfma * groups * waves_per_simd * 4 * cus / (ms * 1e-3) / 1e12;
    printf("%-46s waves/SIMD %d  %8.3f ms  %7.2f ns per MFMA slot and SIMD  %8.1f TFLOP/s (mfma)\n", label, waves_per_simd, ms, ns_per_slot, tf);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<F32_16x16x4, 0, 0>("f32 16x16x4", w, 2048);
        run<F32_16x16x4, 4, 0>("f32 16x16x4 + 4 v_fma_f32", w, 2048);
        run<BF16_16x16x16, 0, 0>("bf16 16x16x16", w, 8192);
        run<BF16_16x16x16, 1, 0>("bf16 16x16x16 + 1 v_fma_f32", w, 8192);
        run<BF16_16x16x16, 2, 0>("bf16 16x16x16 + 2 v_fma_f32", w, 8192);
        run<BF16_16x16x16, 4, 0>("bf16 16x16x16 + 4 v_fma_f32", w, 8192);
        run<BF16_16x16x16, 8, 0>("bf16 16x16x16 + 8 v_fma_f32", w, 8192);
        run<BF16_16x16x32, 0, 0>("bf16 16x16x32", w, 16384);
        run<BF16_16x16x32, 4, 0>("bf16 16x16x32 + 4 v_fma_f32", w, 16384);
        run<BF16_16x16x32, 8, 0>("bf16 16x16x32 + 8 v_fma_f32", w, 16384);
        run<NO_MFMA, 8, 0>("8 v_fma_f32 alone", w, 0);
        run<NO_MFMA, 0, 1>("3-way bf16 split of 2 values alone (11 VALU)", w, 0);
        run<BF16_16x16x32, 0, 1>("bf16 16x16x32 + split of 2 values", w, 16384);
    }
    return 0;
}
