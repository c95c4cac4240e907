// Developer microbenchmark: what a grid-wide rendezvous costs on gfx950 next to a kernel boundary, with one workgroup per CU
// (256 x 256 threads, 160 KB of LDS each) and the memory semantics a fused launch needs (every workgroup's global stores of
// phase n visible to every workgroup in phase n + 1: agent-scope release / acquire around a counter).
//   (a) cooperative_groups grid.sync() under hipLaunchCooperativeKernel
//   (b) a monotonic counter: one agent-scope fetch-add per workgroup, spin on a relaxed agent-scope load + s_sleep
//   (c) the same phases as back-to-back launches of a kernel that does one phase
// Each phase writes 24 KB per workgroup (a partial row) and reads 24 KB written by ANOTHER workgroup in the previous phase
// (checked: the run fails loudly on a stale read).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/grid_barrier.hip -o tools/micro/grid_barrier
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
namespace cg = cooperative_groups;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int ROW = 6144;   // floats per workgroup and phase

__device__ __forceinline__ void phase(float* buf, int ph, int G, int* bad) {
    extern __shared__ float sm[];
    const int b = blockIdx.x, t = threadIdx.x;
    // read what workgroup (b + 1) % G wrote in phase ph - 1, write this workgroup's row of phase ph (double buffered by phase parity)
    float* cur = buf + (size_t)(ph & 1) * G * ROW;
    const float* prev = buf + (size_t)((ph + 1) & 1) * G * ROW;
    float s = 0.f;
    if (ph > 0) {
        const float* p = prev + (size_t)((b + 1) % G) * ROW;
        for (int i = t; i < ROW; i += 256) {
            const float v = p[i];
            if (v != (float)(ph - 1)) atomicAdd(bad, 1);
            s += v;
        }
    }
    sm[t] = s;
    for (int i = t; i < ROW; i += 256) cur[(size_t)b * ROW + i] = (float)ph;
}

__global__ void __launch_bounds__(256) k_coop(float* buf, int phases, int* bad) {
    cg::grid_group g = cg::this_grid();
    for (int ph = 0; ph < phases; ++ph) {
        phase(buf, ph, gridDim.x, bad);
        g.sync();
    }
}

__device__ __forceinline__ void grid_rendezvous(unsigned* counter, unsigned target) {
    __syncthreads();                                   // all of this workgroup's stores are issued
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1L << 24)) break;           // never hang the box: give up (the check below then reports stale reads)
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
__global__ void __launch_bounds__(256) k_spin(float* buf, int phases, int* bad, unsigned* counter) {
    for (int ph = 0; ph < phases; ++ph) {
        phase(buf, ph, gridDim.x, bad);
        grid_rendezvous(counter, (unsigned)(ph + 1) * gridDim.x);
    }
}
__global__ void __launch_bounds__(256) k_one(float* buf, int ph, int* bad) { phase(buf, ph, gridDim.x, bad); }

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int G = prop.multiProcessorCount, phases = 200;
    const size_t smem = 160 * 1024 - 512;
    printf("%s: %d CUs, cooperativeLaunch %d\n", prop.gcnArchName, G, prop.cooperativeLaunch);
    float* buf; int* bad; unsigned* counter;
    CHECK(hipMalloc(&buf, sizeof(float) * 2 * G * ROW));
    CHECK(hipMalloc(&bad, 4)); CHECK(hipMalloc(&counter, 4));
    CHECK(hipFuncSetAttribute((const void*)k_coop, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CHECK(hipFuncSetAttribute((const void*)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CHECK(hipFuncSetAttribute((const void*)k_one, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms; int hbad;
    for (int rep = 0; rep < 2; ++rep) {
        // (c) launches
        CHECK(hipMemset(bad, 0, 4));
        CHECK(hipEventRecord(e0));
        for (int ph = 0; ph < phases; ++ph) k_one<<<G, 256, smem>>>(buf, ph, bad);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
        printf("back-to-back launches      : %7.2f us per phase  (stale reads %d)\n", ms * 1e3 / phases, hbad);
        // (b) spin
        CHECK(hipMemset(bad, 0, 4)); CHECK(hipMemset(counter, 0, 4));
        CHECK(hipEventRecord(e0));
        {
            int ph = phases;
            void* args[] = {&buf, &ph, &bad, &counter};
            CHECK(hipLaunchCooperativeKernel((const void*)k_spin, dim3(G), dim3(256), args, smem, 0));
        }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
        printf("one launch, counter + spin : %7.2f us per phase  (stale reads %d)\n", ms * 1e3 / phases, hbad);
        // (a) cooperative groups
        CHECK(hipMemset(bad, 0, 4));
        CHECK(hipEventRecord(e0));
        {
            int ph = phases;
            void* args[] = {&buf, &ph, &bad};
            CHECK(hipLaunchCooperativeKernel((const void*)k_coop, dim3(G), dim3(256), args, smem, 0));
        }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
        printf("one launch, grid.sync()    : %7.2f us per phase  (stale reads %d)\n", ms * 1e3 / phases, hbad);
    }
    return 0;
}
