// promp_device.h -- device-side vocabulary shared by every kernel of libpromp_hip.
//
// gfx950 / CDNA4 only: 64-lane wavefronts, exact-FP32 MFMA (v_mfma_f32_32x32x2_f32,
// v_mfma_f32_16x16x4_f32) and FP64 MFMA (v_mfma_f64_16x16x4_f64).  There is no other backend.
//
// When PROMP_EMU is defined the same kernel sources are compiled by g++ against tests/emu/hip_emu.h,
// a SIMT interpreter (one host thread per lane, MFMA fragment layouts restated from
// cdna_hip_programming.md section 3).  That build exists only so that tests can check kernel indexing
// in a container without a GPU; it is never built into or loaded by the product.
#pragma once

#ifndef PROMP_SPLIT_TERMS
#define PROMP_SPLIT_TERMS 2
#endif
constexpr int PROMP_NT = PROMP_SPLIT_TERMS;                      // terms per value
constexpr int PROMP_NPROD = PROMP_NT * (PROMP_NT + 1) / 2;       // matrix instructions per product

#ifdef PROMP_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#define PROMP_DEV __device__ __forceinline__
#define PROMP_DEV_NOINLINE __device__ __attribute__((noinline))
#define PROMP_HD __host__ __device__ inline
#define PROMP_CX __host__ __device__ constexpr
#define PROMP_SMEM_DECL extern __shared__ __attribute__((aligned(16))) unsigned char promp_smem_raw[]
#define PROMP_SMEM_PTR promp_smem_raw
#define PROMP_LAUNCH(kern, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kern, grid, dim3(block), smem, stream, __VA_ARGS__)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// D[32x32] += A[32x2] * B[2x32].  lane l holds A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5) for register r in [0,16).
PROMP_DEV f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
// D[16x16] += A[16x4] * B[4x16].  lane l holds A[i=l&15][k=l>>4], B[k=l>>4][j=l&15];
// D: col = l&15, row = 4*(l>>4) + r for r in [0,4).
PROMP_DEV f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// FP64: same A/B lane map as mfma16; D: col = l&15, row = (l>>4) + 4*r.
PROMP_DEV f64x4 mfma16d(double a, double b, f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// D[16x16] += A[16x32] * B[32x16] on the BF16 matrix pipe (16 384 FLOP in half the issue time of the FP32 instruction's
// 2 048).  lane l holds A[i = l & 15][k = 8 (l >> 4) .. + 7] and B[k = 8 (l >> 4) .. + 7][j = l & 15], 8 bf16 each;
// D as mfma16.  Exact products, float32 accumulation (layout and error: tools/micro/bf16_layout_probe.hip).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
PROMP_DEV f32x4 mfma16_bf16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
// Error-compensated split of eight float32 values into three BF16 terms each: x = t0 + t1 + t2 up to 2^-24 |x|
// (v_cvt_pk_bf16_f32 rounds to nearest even; the residuals are exact in float32).
PROMP_DEV void bf16_split3(const float (&x)[8], bf16x8 (&t)[3]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h0 = (__bf16)x[e];
        const float r1 = x[e] - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        const float r2 = r1 - (float)h1;
        t[0][e] = h0;
        t[1][e] = h1;
        t[2][e] = (__bf16)r2;
    }
}
// ---- BF16 fragments as raw 32-bit words ---------------------------------------------------------------------------------
// The pass kernel moves its BF16 operands around as words of two bf16 (low half = the element with the lower index) and turns
// them into bf16 vectors only at the MFMA call, by whole-vector bit casts (building a bf16x8 element by element from 16-bit
// lanes is miscompiled by hipcc 7.2: tools/micro/tr16_wgrad_probe.hip).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
PROMP_DEV f32x4 mfma16_bf16w(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// D[32x32] += A[32x16] * B[16x32] on the BF16 pipe.  lane l holds A[i = l & 31][k = 8 (l >> 5) .. + 7] and
// B[k = 8 (l >> 5) .. + 7][j = l & 31]; D as mfma32: col = l & 31, row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)
// (layout confirmed on the device: profiles/r03_tr16_wgrad_probe.txt).
PROMP_DEV f32x16 mfma32_bf16w(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// ds_read_b64_tr_b16: every lane of a 16-lane group names an 8-byte chunk (4 bf16); with the 16 chunks read as the rows-major
// 4 x 16 matrix M[p / 4][4 (p % 4) .. + 3] = chunk of lane p, lane i receives column i: elements M[0..3][i]
// (= element i % 4 of the chunks of lanes i / 4, 4 + i / 4, 8 + i / 4, 12 + i / 4).  8-byte aligned addresses only.
PROMP_DEV u32x2 lds_tr16(const void* p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p));
}
// Error-compensated 3-way BF16 split of two float32 values: word t holds term t of x0 (low half) and of x1 (high half);
// x = t0 + t1 + t2 up to 2^-24 |x| (round to nearest even, residuals exact in float32).
PROMP_DEV void bf16_split3_pair(float x0, float x1, unsigned (&w)[3]) {
    f32x2 r;
    r[0] = x0;
    r[1] = x1;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const bf16x2 h = __builtin_convertvector(r, bf16x2);
        w[t] = __builtin_bit_cast(unsigned, h);
        if (t < 2) r -= __builtin_convertvector(h, f32x2);
    }
}
// ---- float32-equivalent products on the 16-bit matrix pipe: the split (round 6) ---------------------------------------------
// PROMP_SPLIT_TERMS = 2 (default): x = hi + lo, two FP16 terms (11 significant bits each: 2^-22 |x| as long as lo is not below the
// subnormal floor 2^-25 -- the matrix instruction keeps subnormal FP16 inputs, tools/micro/f16_split_probe.hip); a product is the
// THREE instructions hi.hi + hi.lo + lo.hi (lo.lo ~ 2^-22 dropped) on v_mfma_f32_*_f16, the split is four vector instructions per
// pair of values (v_cvt_pk_f16_f32, two v_fma_mix_f32 for the residuals, v_cvt_pk_f16_f32).  FP16 has a RANGE, so every operand a
// kernel splits is kept near 1 by an exact power of two that the kernel undoes on its accumulators (observations: per task;
// cotangents and the direction of the R-operator pass: per wave / per segment) -- see the kernels.
// PROMP_SPLIT_TERMS = 3 (-DPROMP_SPLIT_TERMS=3; rounds 3-5): three BF16 terms, SIX instructions (terms ta + tb <= 2), no range.
// Either way: word t of a split holds term t of x0 (low half) and of x1 (high half); products are walked ta = NT-1 .. 0,
// tb = NT-1-ta .. 0 (smallest first).
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
// (the term count of an operand array names its format: 2 = FP16, 3 = BF16)
template <int NT>
PROMP_DEV void split_pair(float x0, float x1, unsigned (&w)[NT]) {
    static_assert(NT == 2 || NT == 3, "two FP16 terms or three BF16 terms");
    if constexpr (NT == 2) {
        float r0, r1;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w[0]) : "v"(x0), "v"(x1));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(w[0]), "v"(x0));                   // x0 - hi (low half)
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(w[0]), "v"(x1));    // x1 - hi (high half)
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w[1]) : "v"(r0), "v"(r1));
    } else {
        bf16_split3_pair(x0, x1, w);
    }
}
// both FP16 halves of a word times a power of two (exact unless a half leaves the format: v_pk_mul_f16)
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
PROMP_DEV unsigned pk_scale_f16(unsigned w, float p2) {
    h16x2 f;
    f[0] = f[1] = (_Float16)p2;
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(h16x2, w) * f);
}
template <int NT>
PROMP_DEV f32x4 mfma16_sw(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (NT == 2) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
    else return mfma16_bf16w(a, b, c);
}
template <int NT>
PROMP_DEV f32x16 mfma32_sw(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (NT == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
    else return mfma32_bf16w(a, b, c);
}
PROMP_DEV float shfl_xor_f32(float v, int m) { return __shfl_xor(v, m, 64); }
// Cross-lane sums on the vector ALU (no LDS round trip, unlike ds_bpermute):
//   fold_groups16: v + the values of the same lane index in the other three 16-lane groups (v_permlane32_swap / v_permlane16_swap:
//                  the swap of a register with itself leaves the two halves / row pairs side by side)
//   row16_sum    : the sum over the 16 lanes of a group, in every lane of it (DPP quad permutes and row mirrors)
PROMP_DEV float fold_groups16(float v) {
    // (inline assembly: hipcc 7.2 drops the second result of __builtin_amdgcn_permlane{16,32}_swap -- it adds the first one to
    //  itself; checked on the device.  The s_nop covers the VALU-write -> permlane-read hazard the compiler cannot see here.)
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    const float w = a + b;
    float c = w, d = w;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(c), "+v"(d));
    return c + d;
}
PROMP_DEV float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));    // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));    // row_mirror
    return v;
}
PROMP_DEV double shfl_xor_f64(double v, int m) { return __shfl_xor(v, m, 64); }
PROMP_DEV double shfl_down_f64(double v, int d) { return __shfl_down(v, d, 64); }
PROMP_DEV double shfl_idx_f64(double v, int l) { return __shfl(v, l, 64); }
// value of lane `l` (a compile-time or wave-uniform index) as a wave-uniform double: two v_readlane_b32 into scalar
// registers, no LDS round trip
// v_rsq_f64: the hardware's reciprocal-square-root seed (~2^-26 relative); rsq_refine is the one correction step the math library's
// rsqrt() applies to it (e = 1 - d r0^2; r1 = r0 + r0 e (1/2 + 3/8 e): relative error ~2^-52).  Apart so that a caller can place
// the five dependent operations between independent work (k_fit_wave's column step).
PROMP_DEV double rsq_seed(double d) { return __builtin_amdgcn_rsq(d); }
PROMP_DEV double rsq_e(double d, double r0) { return fma(-d * r0, r0, 1.0); }
PROMP_DEV double rsq_finish(double r0, double e) { return fma(r0 * e, fma(e, 0.375, 0.5), r0); }
PROMP_DEV double readlane_f64(double v, int l) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, l), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
PROMP_DEV bool wave_any(bool p) { return __any(p); }
// Orders this wave's LDS traffic between producer and consumer lanes of the SAME wave (no s_barrier):
// LDS executes one wave's instructions in order, so it is enough to (a) stop the compiler from moving memory
// accesses across this point and (b) have the data written.  Deliberately NOT a fence: a fence would also wait
// for the global prefetch loads in flight (vmcnt) and serialise them with the tile pipeline.
PROMP_DEV void wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// Compiler-only ordering point for wave-private LDS traffic: LDS executes one wave's instructions in issue order, so a
// read issued after a write of the same wave sees it whichever lane wrote; nothing has to be waited for here.
PROMP_DEV void wave_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// Workgroup barrier that orders LDS traffic only: global stores in flight (partial rows on their way to L2) are not
// waited for, unlike __syncthreads().
PROMP_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Two-wave rendezvous through a pair of LDS words (gfx950 has no named barriers, and s_barrier is the whole workgroup's).
// LDS executes one wave's instructions in issue order, so a flag stored after data is seen after the data; the waiting side
// polls the partner's word (one broadcast read) and sleeps in between.  pair_post also waits for this wave's own LDS reads: the
// partner may overwrite what they read once it has seen the flag.
PROMP_DEV void pair_post(float* flags, int mine, int seq, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) ((volatile int*)flags)[mine] = seq;
}
PROMP_DEV void pair_wait(float* flags, int other, int seq) {
    while (__builtin_amdgcn_readfirstlane(((volatile int*)flags)[other]) < seq) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
// Agent-scope hand-off between workgroups (MI355X_MICROARCH.md, inter-workgroup visibility): the producer writes its
// data, releases (L2 write-back + drain), then bumps a counter with a relaxed agent-scope atomic; the workgroup that
// reads the final count acquires (L1 invalidate) before it loads the others' data.
PROMP_DEV void fence_release_agent() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the compiler may drop the wait after buffer_wbl2; the asm stays)
}
PROMP_DEV void fence_acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
PROMP_DEV int atomic_add_agent(int* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PROMP_DEV void atomic_store_agent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PROMP_DEV void atomic_max_agent(unsigned* p, unsigned v) { __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// hand-off to the HOST through page-locked memory: everything this thread stored before is visible to a host thread that
// reads the flag with acquire semantics and finds the new value
PROMP_DEV void release_store_system(unsigned* p, unsigned v) {
    __threadfence_system();
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// keeps the compiler from interleaving two independent GEMM groups (which would add their live ranges)
PROMP_DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// A zero the optimiser cannot see through.  Added to the base pointers inside a long unrolled loop body it keeps the
// hundreds of constant-offset address computations from being hoisted out of the loop (and then spilled) by LICM.
PROMP_DEV int opaque_zero() {
    int z = 0;
    asm volatile("" : "+v"(z));
    return z;
}
// An LDS address the optimiser cannot see through: accesses at small constant offsets from it keep those offsets as instruction
// immediates off ONE address register (folded into an absolute address they may not fit the offset field of ds_read2 / ds_write2).
PROMP_DEV double* opaque_lds(double* p) {
    unsigned a = (unsigned)(size_t)p;
    asm volatile("" : "+v"(a));
    return (double*)(__attribute__((address_space(3))) double*)(size_t)a;
}
// Tells the compiler a value is the same in every lane of the wave (e.g. the wave index threadIdx.x >> 6), so that
// everything derived from it lives in scalar registers.
// Pins a value into the accumulator half of the register file (AGPRs: usable as MFMA operands and load / store data
// only).  At one wave per SIMD a lane has 256 + 256 registers; weights that only ever feed MFMAs belong in the second
// half, where they do not compete with the activations and epilogue temporaries for the 256 VALU-visible registers.
PROMP_DEV f32x4 pin_agpr(f32x4 v) {
    asm volatile("" : "+a"(v));
    return v;
}
// Pins: zero-instruction pass-throughs the optimiser cannot see through.  A value that went through one exists, in a register of
// the named file, at that point of the instruction stream; together with sched_fence() they decide which stage of a software
// pipeline a piece of pure arithmetic belongs to (instruction selection otherwise sinks it to its first use, across any
// scheduling barrier).  pin_v: vector register; pin_a: accumulator register (MFMA accumulators that live across iterations).
PROMP_DEV void pin_v(float& x) { asm volatile("" : "+v"(x)); }
PROMP_DEV void pin_v(unsigned& x) { asm volatile("" : "+v"(x)); }
PROMP_DEV void pin_v(int& x) { asm volatile("" : "+v"(x)); }
PROMP_DEV void pin_s(unsigned& x) { asm volatile("" : "+s"(x)); }
PROMP_DEV void pin_v(f32x4& x) { asm volatile("" : "+v"(x)); }
PROMP_DEV void pin_v(u32x4& x) { asm volatile("" : "+v"(x)); }
PROMP_DEV void pin_a(f32x4& x) { asm volatile("" : "+a"(x)); }
PROMP_DEV void pin_a(f32x16& x) { asm volatile("" : "+a"(x)); }
// Instruction-group pipelines (scheduling hints, no code): the region that holds them is laid out as the sequence of groups the
// calls name, in call order.  One matrix instruction followed by `n_valu` vector instructions (and a DS read when `ds`): the
// in-order issue of a single wave then runs the vector work of one instruction stream in the shadow of the other's MFMAs.
#define PROMP_SCHED_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
#define PROMP_SCHED_VALU(n) __builtin_amdgcn_sched_group_barrier(0x002, n, 0)
#define PROMP_SCHED_DSREAD(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)
#define PROMP_SCHED_DSWRITE(n) __builtin_amdgcn_sched_group_barrier(0x200, n, 0)
// XCD-aware work-item index.  The dispatcher places block b on XCD b % 8 (observed, not a contract: a performance tool only), and
// every XCD has its own 4 MB L2.  Work tables are ordered by task, so with the identity mapping the workgroups of one task are
// spread over all eight L2s and each L2 sees every task's parameters.  This bijection hands XCD x the x-th contiguous eighth of
// the table: the workgroups that share a task's parameters share an L2 (k_wide_hvp re-reads theta and the direction once per
// 64-row round: 755 MB per launch through the fabric with the identity mapping, measured).
PROMP_DEV int xcd_item(int b, int G) {
    const int x = b & 7;
    int off = 0;
#pragma unroll
    for (int y = 0; y < 7; ++y) off += (y < x) ? ((G - y + 7) >> 3) : 0;
    return off + (b >> 3);
}
PROMP_DEV int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// issue priority of this wave among the waves of its SIMD (s_setprio takes an immediate: 0..3)
PROMP_DEV void wave_priority(int p) {
    if (p >= 3) __builtin_amdgcn_s_setprio(3);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}
PROMP_DEV unsigned long long promp_clock() { return (unsigned long long)clock64(); }
PROMP_DEV unsigned long long promp_wall_clock() { return (unsigned long long)wall_clock64(); }   // constant 100 MHz
PROMP_DEV float fast_exp(float x) { return __expf(x); }
PROMP_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }   // v_rcp_f32, 1 ulp
PROMP_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32
// two fused multiply-adds in one v_pk_fma_f32 (an FMA costs two issue slots next to FP32 MFMAs, packed or not)
PROMP_DEV f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
#endif

// Exact scales of the FP16 split's operands: 2^e as a float (e in [-126, 127]), and the exponent k that brings m = 2^e (1 + f) to
// [2^t, 2^(t+1)): k = t - e.  (m = 0 / subnormal reads as e = -127, infinity / NaN as e = 128: callers test m and clamp k.)
PROMP_DEV float pow2f(int e) { return __builtin_bit_cast(float, (unsigned)(e + 127) << 23); }
PROMP_DEV int scale_exp(float m, int t) { return t - ((int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xFF) - 127); }
PROMP_DEV f32x4 zero4() {
    f32x4 z;
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = 0.f;
    return z;
}
PROMP_DEV f64x4 zero4d() {
    f64x4 z;
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = 0.0;
    return z;
}

PROMP_DEV float wave_sum_f32(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f32(v, m);
    return v;
}
PROMP_DEV double wave_sum_f64(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
    return v;
}
PROMP_DEV double wave_min_f64(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        double o = shfl_xor_f64(v, m);
        v = o < v ? o : v;
    }
    return v;
}

// The primal / tangent pair of one layer in a single pass over k (R-operator kernels):
//     P[i][j] += A0_i B0_j                       T[i][j] += A1_i B0_j + s1 * A0_i B1_j
// One software pipeline instead of three: the A0 / B0 operands are loaded once, a k-step issues 3 NA NB independent
// MFMAs (enough to cover the LDS latency of the next step's operands), and two pipeline ramp-ups disappear.
// a1 == nullptr drops the A1 term (first layer: the input has no tangent).
template <int NA, int NB, bool HAS_A1>
PROMP_DEV void outer16_pt(f32x4 (&P)[NA][NB], f32x4 (&T)[NA][NB], const float* a0, const float* a1, int a_ss, int a_bs,
                          const float* b0, const float* b1, int b_ss, int b_bs, int NS, float s1) {
    float x0[NA], x1[NA], y0[NB], y1[NB];
    f32x4 T2[NA][NB];          // the A1 B0 term accumulates apart from the A0 B1 term and joins T at the end
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) T2[i][j] = zero4();
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        x0[i] = a0[i * a_bs];
        x1[i] = HAS_A1 ? a1[i * a_bs] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        y0[j] = b0[j * b_bs];
        y1[j] = s1 * b1[j * b_bs];
    }
#pragma unroll 2
    for (int s = 1; s <= NS; ++s) {
        float nx0[NA], nx1[NA], ny0[NB], ny1[NB];
        const int sn = s < NS ? s : 0;          // the last step re-reads step 0 (harmless) instead of branching
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            nx0[i] = a0[sn * a_ss + i * a_bs];
            nx1[i] = HAS_A1 ? a1[sn * a_ss + i * a_bs] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            ny0[j] = b0[sn * b_ss + j * b_bs];
            ny1[j] = s1 * b1[sn * b_ss + j * b_bs];
        }
        // three independent accumulator sets, each walked completely before the next: no MFMA waits for its predecessor
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) P[i][j] = mfma16(x0[i], y0[j], P[i][j]);
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) T[i][j] = mfma16(x0[i], y1[j], T[i][j]);
        if (HAS_A1) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) T2[i][j] = mfma16(x1[i], y0[j], T2[i][j]);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            x0[i] = nx0[i];
            x1[i] = nx1[i];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            y0[j] = ny0[j];
            y1[j] = ny1[j];
        }
    }
    if (HAS_A1) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) T[i][j] += T2[i][j];
    }
}

// acc[i][j] += s0 * A0_i B0_j + A1_i B1_j   (the two gradient terms of one kernel in the R-operator pass)
template <int NA, int NB>
PROMP_DEV void outer16_two(f32x4 (&acc)[NA][NB], const float* a0, const float* a1, int a_ss, int a_bs, const float* b0,
                           const float* b1, int b_ss, int b_bs, int NS, float s0) {
    float x0[NA], x1[NA], y0[NB], y1[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        x0[i] = a0[i * a_bs];
        x1[i] = a1[i * a_bs];
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        y0[j] = s0 * b0[j * b_bs];
        y1[j] = b1[j * b_bs];
    }
#pragma unroll 2
    for (int s = 1; s <= NS; ++s) {
        float nx0[NA], nx1[NA], ny0[NB], ny1[NB];
        const int sn = s < NS ? s : 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            nx0[i] = a0[sn * a_ss + i * a_bs];
            nx1[i] = a1[sn * a_ss + i * a_bs];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            ny0[j] = s0 * b0[sn * b_ss + j * b_bs];
            ny1[j] = b1[sn * b_ss + j * b_bs];
        }
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = mfma16(x0[i], y0[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = mfma16(x1[i], y1[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            x0[i] = nx0[i];
            x1[i] = nx1[i];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            y0[j] = ny0[j];
            y1[j] = ny1[j];
        }
    }
}

// tanh(x) = 1 - 2/(exp(2x)+1): absolute error ~1e-7, saturates correctly at +-1.
PROMP_DEV float fast_tanh(float x) { return 1.f - 2.f * fast_rcp(fast_exp(2.f * x) + 1.f); }

// The same for a pre-activation that arrives already scaled by PROMP_TANH_PRESCALE = 2 log2(e) (the kernels fold the
// factor into the staged kernel / bias of the layer): tanh(x) = 1 - 2 / (2^y + 1), y = 2 log2(e) x.  Two elements at a time:
// v_exp_f32, v_add_f32, v_rcp_f32 each and ONE v_pk_fma_f32 for the pair.
#define PROMP_TANH_PRESCALE 2.8853900817779268f
PROMP_DEV f32x2 tanh2_prescaled(float y0, float y1) {
    f32x2 r;
    r[0] = fast_rcp(fast_exp2(y0) + 1.f);
    r[1] = fast_rcp(fast_exp2(y1) + 1.f);
    f32x2 m2, one;
    m2[0] = m2[1] = -2.f;
    one[0] = one[1] = 1.f;
    return pk_fma(r, m2, one);
}
// h^2 - 1 (the NEGATED tanh derivative) for two elements in one v_pk_fma_f32; the caller's multiply takes the sign back
// as a source modifier (negating h first would cost a v_xor per element)
PROMP_DEV f32x2 neg_dtanh2(float h0, float h1) {
    f32x2 h, m1;
    h[0] = h0; h[1] = h1;
    m1[0] = m1[1] = -1.f;
    return pk_fma(h, h, m1);
}

// acc[ia][ib] (16x16 tiles) += sgn * A_ia * B_ib over NS k-steps of 16x16x4 MFMAs.
// Operand streams: at step s, A block ia feeds a[s*a_ss + ia*a_bs], B block ib feeds b[s*b_ss + ib*b_bs] (pointers
// already offset for this lane).  Which four k values a step contracts is the caller's choice (any bijection of K
// onto (step, lane>>4) works as long as A and B agree) - the kernels pick it per GEMM for conflict-free LDS banks.
// NA*NB independent accumulators hide the 40-cycle dependent latency; the operands of step s+1 are requested before
// the MFMAs of step s issue.
template <int NA, int NB>
PROMP_DEV void outer16(f32x4 (&acc)[NA][NB], const float* a, int a_ss, int a_bs, const float* b, int b_ss, int b_bs, int NS,
                       float sgn) {
    float a0[NA], b0[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) a0[i] = a[i * a_bs];
#pragma unroll
    for (int j = 0; j < NB; ++j) b0[j] = b[j * b_bs];
#pragma unroll 2
    for (int s = 1; s < NS; ++s) {
        float a1[NA], b1[NB];
#pragma unroll
        for (int i = 0; i < NA; ++i) a1[i] = a[s * a_ss + i * a_bs];
#pragma unroll
        for (int j = 0; j < NB; ++j) b1[j] = b[s * b_ss + j * b_bs];
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = mfma16(sgn * a0[i], b0[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < NA; ++i) a0[i] = a1[i];
#pragma unroll
        for (int j = 0; j < NB; ++j) b0[j] = b1[j];
    }
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = mfma16(sgn * a0[i], b0[j], acc[i][j]);
}
