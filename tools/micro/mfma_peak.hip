// Developer microbenchmark: sustained v_mfma_f32_16x16x4_f32 rate on gfx950 with and without co-issued VALU / LDS work.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int VALU_PER, int LDS_PER>
__global__ void __launch_bounds__(1024) k(float* out, int iters, float seed) {
    __shared__ float lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += blockDim.x) lds[i] = seed + i;
    __syncthreads();
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
    float a = seed + tid, b = seed * 2 + tid;
    float v[4] = {seed, seed + 1, seed + 2, seed + 3};
    const float* lp = lds + (tid & 63);
    float l = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < VALU_PER; ++u) v[u & 3] = __builtin_fmaf(v[u & 3], 1.0001f, 0.5f);
#pragma unroll
            for (int u = 0; u < LDS_PER; ++u) l += lp[((it + i + u) & 31) * 64];
        }
    }
    float s = l + v[0] + v[1] + v[2] + v[3];
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int NACC, int VALU_PER, int LDS_PER>
void run(const char* name, int threads, int blocks_per_cu, float* d) {
    const int iters = 40000, ncu = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, VALU_PER, LDS_PER>), dim3(ncu * blocks_per_cu), dim3(threads), 0, 0, d, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) {
            const double waves = (double)ncu * blocks_per_cu * threads / 64;
            const double flop = waves * iters * NACC * 2048.0;
            printf("%-34s waves/SIMD %.1f  %7.3f ms  %7.1f TFLOP/s (mfma)\n", name, waves / (ncu * 4), ms, flop / ms * 1e-9);
        }
    }
}

int main() {
    float* d; hipMalloc(&d, 256 * 8 * 1024 * sizeof(float));
    run<4, 0, 0>("pure mfma, 4 acc", 256, 1, d);
    run<4, 0, 0>("pure mfma, 4 acc", 512, 1, d);
    run<4, 0, 0>("pure mfma, 4 acc", 1024, 1, d);
    run<8, 0, 0>("pure mfma, 8 acc", 256, 1, d);
    run<1, 0, 0>("pure mfma, 1 acc (dependent)", 256, 1, d);
    run<1, 0, 0>("pure mfma, 1 acc (dependent)", 512, 1, d);
    run<4, 1, 0>("mfma + 1 valu", 256, 1, d);
    run<4, 1, 0>("mfma + 1 valu", 512, 1, d);
    run<4, 2, 0>("mfma + 2 valu", 256, 1, d);
    run<4, 2, 0>("mfma + 2 valu", 512, 1, d);
    run<4, 4, 0>("mfma + 4 valu", 256, 1, d);
    run<4, 4, 0>("mfma + 4 valu", 512, 1, d);
    run<4, 8, 0>("mfma + 8 valu", 256, 1, d);
    run<4, 8, 0>("mfma + 8 valu", 512, 1, d);
    run<4, 0, 1>("mfma + 1 ds_read", 256, 1, d);
    run<4, 0, 1>("mfma + 1 ds_read", 512, 1, d);
    run<4, 0, 2>("mfma + 2 ds_read", 512, 1, d);
    run<4, 2, 1>("mfma + 2 valu + 1 ds_read", 512, 1, d);
    run<4, 4, 2>("mfma + 4 valu + 2 ds_read", 512, 1, d);
    return 0;
}
