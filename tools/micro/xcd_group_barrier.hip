// Developer microbenchmark (round 6; the round-5 review's item 2, step 1): what a rendezvous among the ~6 workgroups of ONE task costs
// when they sit on one XCD -- an Adam epoch's three passes have no cross-task dependency until the task mean, and 40 tasks / 8 XCDs =
// 5 tasks x ~6 CUs sharing one L2.  One workgroup per CU (256 x 256 threads, 160 KB of LDS: the pass kernels' footprint); workgroup
// b is assumed to land on XCD b % 8 (observed placement, not a contract), its index on the XCD is b / 8, groups of GS consecutive
// indices.  Every phase a workgroup writes a 24 KB row (a partial row of the pass kernels) and, after the rendezvous of its group,
// reads the row ANOTHER member of its group wrote in that phase (stale reads are counted: the run says so loudly).
//   (a) agent-scope release / acquire around a per-group counter (what the memory model asks for between any two workgroups)
//   (b) "same L2": the writer waits for its stores (s_waitcnt vmcnt(0): acknowledged by the L2 both sides share), bumps the counter with a
//       relaxed atomic (executed in that L2), the reader spins on it and invalidates its vector L1 (buffer_inv sc1) -- no L2 write-back
//   (c) the same phases with no rendezvous at all (what the data movement alone costs) and as back-to-back launches
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/xcd_group_barrier.hip -o tools/micro/xcd_group_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int ROW = 6144;   // floats per workgroup and phase
constexpr int GS = 6;       // workgroups per group

struct Who {
    int xcd, idx, group, member, size, peer;   // peer: global block index of the member whose row this workgroup reads
    bool active;
};
__device__ __forceinline__ Who who_am_i(int b, int G) {
    Who w;
    w.xcd = b & 7;
    w.idx = b >> 3;
    const int per = G >> 3, ngroups = per / GS;        // 32 workgroups per XCD -> 5 groups of 6, two idle
    w.group = w.idx / GS;
    w.member = w.idx - w.group * GS;
    w.active = w.group < ngroups;
    w.size = GS;
    w.peer = ((w.group * GS + (w.member + 1) % GS) << 3) | w.xcd;
    return w;
}

// a returning add executed where a non-sc1 atomic is: the L2 of the issuing XCD
__device__ __forceinline__ unsigned l2_add(unsigned* p, unsigned v) {
    unsigned old;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(old) : "v"(p), "v"(v) : "memory");
    return old;
}

template <int MODE>     // 0: agent fences, 1: same-L2, 2: no rendezvous, 3: same-L2 with workgroup-scope read-modify-writes (no sc1: served by the XCD's L2)
__global__ void __launch_bounds__(256) k_groups(float* buf, int phases, int* bad, unsigned* counters) {
    extern __shared__ float sm[];
    const int b = blockIdx.x, t = threadIdx.x, G = gridDim.x;
    const Who w = who_am_i(b, G);
    if (!w.active) return;
    unsigned* counter = counters + 32 * (w.xcd * 8 + w.group);      // one 128-byte line per group
    float s = 0.f;
    bool hung = false;
    if (t == 0) sm[0] = 0.f;
    for (int ph = 0; ph < phases; ++ph) {
        float* cur = buf + (size_t)(ph & 1) * G * ROW;
        for (int i = t; i < ROW; i += 256) cur[(size_t)b * ROW + i] = (float)ph;
        if (MODE != 2) {
            if (MODE != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // every wave: its own stores acknowledged by the L2
            __syncthreads();
            if (t == 0) {
                const unsigned target = (unsigned)(ph + 1) * w.size;
                long spins = 0;
                if (MODE == 3) {      // (written as instructions: the compiler turns an idempotent workgroup-scope RMW into a load the L1 may serve)
                    unsigned seen = l2_add(counter, 1u) + 1u;
                    while (seen < target) {
                        seen = l2_add(counter, 0u);
                        if (++spins > (1L << 14)) { hung = true; break; }
                    }
                } else {
                    if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    unsigned seen = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
                    while (seen < target) {
                        seen = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (++spins > (1L << 14)) { hung = true; break; }       // never hang the box
                    }
                    if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
            }
            if (t == 0 && hung) { atomicAdd(bad, 1 << 20); sm[0] = -1.f; }
            __syncthreads();
            if (sm[0] == -1.f) return;
            if (MODE == 1 || MODE == 3) asm volatile("buffer_inv sc1" ::: "memory");      // every wave: its CU's vector L1 holds last phase's lines
        }
        const float* p = cur + (size_t)w.peer * ROW;
        for (int i = t; i < ROW; i += 256) {
            const float v = p[i];
            if (MODE != 2 && v != (float)ph) atomicAdd(bad, 1);       // (mode 2 reads whatever is there: it only prices the traffic)
            s += v;
        }
        // (the double buffer gives a full phase of slack before a row is overwritten: one rendezvous per phase is enough here)
    }
    sm[1 + t] = s;
}

__global__ void __launch_bounds__(256) k_one(float* buf, int ph, int* bad) {
    extern __shared__ float sm[];
    const int b = blockIdx.x, t = threadIdx.x, G = gridDim.x;
    const Who w = who_am_i(b, G);
    if (!w.active) return;
    float* cur = buf + (size_t)(ph & 1) * G * ROW;
    const float* prev = buf + (size_t)((ph + 1) & 1) * G * ROW;
    float s = 0.f;
    if (ph > 0) {
        const float* p = prev + (size_t)w.peer * ROW;
        for (int i = t; i < ROW; i += 256) {
            const float v = p[i];
            if (v != (float)(ph - 1)) atomicAdd(bad, 1);
            s += v;
        }
    }
    sm[t] = s;
    for (int i = t; i < ROW; i += 256) cur[(size_t)b * ROW + i] = (float)ph;
}

// where the workgroups really are: XCC_ID of every block (s_getreg_b32 HW_REG_XCC_ID)
__global__ void k_where(int* out) {
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[blockIdx.x] = (int)(x & 0xF);
    }
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int G = prop.multiProcessorCount, phases = 400;
    const size_t smem = 160 * 1024 - 512;
    float* buf; int* bad; unsigned* counters; int* where;
    CHECK(hipMalloc(&buf, sizeof(float) * 2 * G * ROW));
    CHECK(hipMalloc(&bad, 4)); CHECK(hipMalloc(&counters, 4 * 32 * 64)); CHECK(hipMalloc(&where, 4 * G));
    CHECK(hipFuncSetAttribute((const void*)k_groups<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CHECK(hipFuncSetAttribute((const void*)k_groups<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CHECK(hipFuncSetAttribute((const void*)k_groups<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CHECK(hipFuncSetAttribute((const void*)k_groups<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    CHECK(hipFuncSetAttribute((const void*)k_one, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    {
        k_where<<<G, 64>>>(where);
        int* h = (int*)malloc(4 * G);
        CHECK(hipMemcpy(h, where, 4 * G, hipMemcpyDeviceToHost));
        int off = 0;
        for (int b = 0; b < G; ++b) off += (h[b] != (b & 7));
        printf("%s: %d CUs; workgroups NOT on XCD b %% 8: %d of %d\n", prop.gcnArchName, G, off, G);
        free(h);
    }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms; int hbad;
    const char* names[4] = {"one launch, agent release / acquire per group     ", "one launch, same-L2 (vmcnt + buffer_inv), agent RMW", "one launch, NO rendezvous (data movement only)     ",
                            "one launch, same-L2, workgroup-scope RMW polling   "};
    float per_phase[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(bad, 0, 4));
        CHECK(hipEventRecord(e0));
        for (int ph = 0; ph < phases; ++ph) k_one<<<G, 256, smem>>>(buf, ph, bad);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
        printf("back-to-back launches                              : %7.2f us per phase  (stale reads %d)\n", ms * 1e3 / phases, hbad);
        for (int mode = 0; mode < 4; ++mode) {
            CHECK(hipMemset(bad, 0, 4)); CHECK(hipMemset(counters, 0, 4 * 32 * 64));
            CHECK(hipEventRecord(e0));
            if (mode == 0) k_groups<0><<<G, 256, smem>>>(buf, phases, bad, counters);
            if (mode == 1) k_groups<1><<<G, 256, smem>>>(buf, phases, bad, counters);
            if (mode == 2) k_groups<2><<<G, 256, smem>>>(buf, phases, bad, counters);
            if (mode == 3) k_groups<3><<<G, 256, smem>>>(buf, phases, bad, counters);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
            per_phase[mode] = ms * 1e3 / phases;
            printf("%s : %7.2f us per phase  (stale reads %d)\n", names[mode], per_phase[mode], hbad);
            fflush(stdout);
        }
        printf("  -> rendezvous alone: agent fences %.2f us, same-L2 %.2f us, same-L2 with workgroup-scope RMWs %.2f us\n", per_phase[0] - per_phase[2],
               per_phase[1] - per_phase[2], per_phase[3] - per_phase[2]);
    }
    return 0;
}
