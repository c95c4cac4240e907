// Developer microbenchmark: which instruction classes take matrix-issue time away from v_mfma_f32_16x16x4_f32 on gfx950.
// Every variant issues 4 independent MFMAs per iteration group, each followed by N instructions of one class (inline asm,
// independent register chains), at one and two waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/issue_mix.hip -o tools/micro/issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Kind { NONE, FMA, PKFMA, PKMUL, ADDU, MOV, EXP, RCP, CNDMASK, CVT, DSR32, DSR128, DSW32, SALU, MUL, LSHLADD, PERM, READLANE, DPP, FMAC, MULE64, ADDF, MULLIT, MULSGPR, WAITCNT, SNOP, EXPE64, FMAMK, FMASGPR };

template <int KIND, int N>
__device__ __forceinline__ void extra(float (&v)[4], f32x2 (&p)[4], int (&iv)[4], f32x4 (&ld)[2], unsigned ldsaddr, int& sacc) {
    const f32x2 pc0 = {1.0001f, 0.9999f}, pc1 = {0.5f, 0.25f};
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const int c = u & 3;
        if (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(1.0001f), "v"(0.5f));
        if (KIND == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[c]) : "v"(1.0001f));
        if (KIND == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[c]) : "v"(pc0), "v"(pc1));
        if (KIND == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[c]) : "v"(pc0));
        if (KIND == ADDU) asm volatile("v_add_u32 %0, %0, %1" : "+v"(iv[c]) : "v"(3));
        if (KIND == LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(iv[c]) : "v"(3));
        if (KIND == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(iv[c]) : "v"(iv[(c + 1) & 3]));
        if (KIND == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[c]));
        if (KIND == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[c]));
        if (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(iv[c]) : "v"(iv[(c + 1) & 3]));
        if (KIND == CVT) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(v[c]) : "v"(iv[c]));
        if (KIND == DSR32) asm volatile("ds_read_b32 %0, %1" : "=v"(v[c]) : "v"(ldsaddr));
        if (KIND == DSR128) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[c & 1]) : "v"(ldsaddr));
        if (KIND == DSW32) asm volatile("ds_write_b32 %0, %1" : : "v"(ldsaddr), "v"(v[c]));
        if (KIND == SALU) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc));
        if (KIND == FMAC) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(v[c]) : "v"(pc0[0]), "v"(pc1[0]));
        if (KIND == MULE64) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(v[c]) : "v"(pc0[0]));
        if (KIND == ADDF) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(v[c]) : "v"(pc1[0]));
        if (KIND == MULLIT) asm volatile("v_mul_f32_e32 %0, 0x4038aa3b, %0" : "+v"(v[c]));
        if (KIND == MULSGPR) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(v[c]) : "s"(1.0001f));
        if (KIND == WAITCNT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (KIND == SNOP) asm volatile("s_nop 0" ::: "memory");
        if (KIND == EXPE64) asm volatile("v_exp_f32_e64 %0, %0" : "+v"(v[c]));
        if (KIND == FMAMK) asm volatile("v_fmamk_f32 %0, %0, 0x3f8003a8, %1" : "+v"(v[c]) : "v"(pc1[0]));
        if (KIND == FMASGPR) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "s"(1.0001f), "v"(pc1[0]));
        if (KIND == PERM) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(iv[c]) : "v"(ldsaddr));
        if (KIND == READLANE) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sacc) : "v"(iv[c]));
        if (KIND == DPP) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(iv[c]) : "v"(iv[(c + 1) & 3]));
    }
}

template <int KIND, int N, int M32>
__global__ void __launch_bounds__(512) k(float* out, int iters, float seed) {
    __shared__ float lds[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += blockDim.x) lds[i] = seed + i;
    __syncthreads();
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
    float a = seed + tid, b = seed * 2 + tid;
    float v[4] = {seed, seed + 1, seed + 2, seed + 3};
    f32x2 p[4];
    int iv[4];
    f32x4 ld[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int i = 0; i < 4; ++i) { p[i] = {seed + i, seed - i}; iv[i] = tid + i; }
    const unsigned ldsaddr = (unsigned)(size_t)(lds) + (tid & 63) * (KIND == DSR128 ? 16 : 4);
    int sacc = 0;
    for (int it = 0; it < iters; ++it) {
        if (M32) {      // clustered: 4 MFMAs back to back, then the 4 N extras
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < 4; ++i) extra<KIND, N>(v, p, iv, ld, ldsaddr, sacc);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                extra<KIND, N>(v, p, iv, ld, ldsaddr, sacc);
            }
        }
        if (KIND == DSR32 || KIND == DSR128 || KIND == DSW32 || KIND == PERM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 7\n s_nop 7" ::: "memory");
    float s = v[0] + v[1] + v[2] + v[3] + (float)sacc;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + p[i][0] + p[i][1] + (float)iv[i];
    s += ld[0][0] + ld[1][1];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (sacc == 12345) out[0] = 1.f;
}

template <int KIND, int N, int M32 = 0>
void run(const char* name, float* d) {
    const int iters = 20000, ncu = 256;
    for (int threads = 256; threads <= 512; threads *= 2) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<KIND, N, M32>), dim3(ncu), dim3(threads), 0, 0, d, iters, 1.0f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double waves = (double)ncu * threads / 64;
        const double flop = waves * iters * 4 * 2048.0;
        // cycles per (mfma + N extras) on one SIMD at 2.4 GHz, summed over the waves of the SIMD
        const double cyc = ms * 1e-3 * 2.4e9 / (iters * 4.0 * (threads / 256));
        printf("%-22s N=%d waves/SIMD %d  %7.3f ms  %6.1f TF  %5.1f cyc/group  extra/instr %5.2f\n", name, N, threads / 256, ms,
               flop / ms * 1e-9, cyc, N ? (cyc - 32.0) / N : 0.0);
    }
}

int main() {
    float* d; hipMalloc(&d, 256 * 8 * 1024 * sizeof(float));
    run<NONE, 0>("pure mfma", d);
    run<FMA, 4>("v_fma_f32", d);
    run<FMA, 8>("v_fma_f32", d);
    run<MUL, 4>("v_mul_f32", d);
    run<PKFMA, 4>("v_pk_fma_f32", d);
    run<PKMUL, 4>("v_pk_mul_f32", d);
    run<ADDU, 4>("v_add_u32", d);
    run<ADDU, 8>("v_add_u32", d);
    run<LSHLADD, 4>("v_lshl_add_u32", d);
    run<MOV, 4>("v_mov_b32", d);
    run<CNDMASK, 4>("v_cndmask_b32", d);
    run<CVT, 4>("v_cvt_f32_i32", d);
    run<EXP, 2>("v_exp_f32", d);
    run<EXP, 4>("v_exp_f32", d);
    run<RCP, 4>("v_rcp_f32", d);
    run<DSR32, 1>("ds_read_b32", d);
    run<DSR32, 2>("ds_read_b32", d);
    run<DSR32, 4>("ds_read_b32", d);
    run<DSR128, 1>("ds_read_b128", d);
    run<DSR128, 2>("ds_read_b128", d);
    run<DSW32, 2>("ds_write_b32", d);
    run<PERM, 2>("ds_bpermute_b32", d);
    run<DPP, 4>("v_mov_b32_dpp", d);
    run<READLANE, 4>("v_readlane_b32", d);
    run<SALU, 4>("s_add_u32", d);
    run<WAITCNT, 4>("s_waitcnt", d);
    run<SNOP, 4>("s_nop 0", d);
    run<FMAC, 4>("v_fmac_f32_e32", d);
    run<FMAMK, 4>("v_fmamk_f32 (lit)", d);
    run<FMASGPR, 4>("v_fma_f32 sgpr,sgpr", d);
    run<MULE64, 4>("v_mul_f32_e64", d);
    run<ADDF, 4>("v_add_f32_e32", d);
    run<MULLIT, 4>("v_mul_f32 literal", d);
    run<MULSGPR, 4>("v_mul_f32 sgpr", d);
    run<EXPE64, 4>("v_exp_f32_e64", d);
    run<FMA, 4, 1>("clustered v_fma_f32", d);
    run<MUL, 4, 1>("clustered v_mul_f32", d);
    run<MUL, 8, 1>("clustered v_mul_f32", d);
    run<EXP, 4, 1>("clustered v_exp_f32", d);
    run<DSR32, 4, 1>("clustered ds_read_b32", d);
    run<DSR128, 2, 1>("clustered ds_read_b128", d);
    return 0;
}
