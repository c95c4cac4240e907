// tr16_wgrad_probe.hip -- developer tooling: the three hardware idioms the BF16-split pass kernel rests on, checked on the device.
//   (1) ds_read_b64_tr_b16: which (lane, element) of the 16-lane group's 4x16 chunk matrix lands where
//   (2) v_mfma_f32_32x32x16_bf16 operand / result layout
//   (3) the whole weight-gradient path: activations in the register-chain layout (lane = sample i16, units 16c + 4kk + r),
//       3-way BF16 split, swizzled 8-byte chunks in LDS, transpose reads, 6 of 9 products on 32x32x16, against float64
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/tr16_wgrad_probe.hip -o tools/micro/tr16_wgrad_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ s16x4 lds_tr16(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

// ---- (1) ----
__global__ void k_tr(unsigned short* out, const int* chunk_of_lane) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    unsigned short* s = (unsigned short*)sm;
    for (int i = threadIdx.x; i < 4096; i += 64) s[i] = (unsigned short)i;
    __syncthreads();
    const s16x4 v = lds_tr16(sm + 8 * chunk_of_lane[threadIdx.x]);
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = (unsigned short)v[e];
}

// ---- (2) ----
__global__ void k_mfma32(float* D, const float* A, const float* B) {   // A [32][16], B [16][32] row-major, bf16-exact values
    const int l = threadIdx.x, i = l & 31, kh = l >> 5;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)A[i * 16 + 8 * kh + e];
        b[e] = (__bf16)B[(8 * kh + e) * 32 + i];
    }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + i] = c[r];
}

// ---- (3) ----
// chunk (sample s, unit chunk q = unit / 4) of a [16][64] tile -> 8-byte slot (conflict-free for the ds_write_b64 of the
// chain layout and for the transpose reads of a 32-unit block)
__host__ __device__ inline int slot_of(int s, int q) {
    return 32 * (4 * (q >> 3) + (q & 3)) + 16 * ((q >> 2) & 1) + 4 * (s & 3) + ((s >> 2) ^ (q & 3));
}
__device__ __forceinline__ void split3(const float (&x)[4], unsigned (&lo)[3], unsigned (&hi)[3]) {
    float r[4] = {x[0], x[1], x[2], x[3]};
    for (int t = 0; t < 3; ++t) {
        unsigned short h[4];
        for (int e = 0; e < 4; ++e) {
            const __bf16 b = (__bf16)r[e];
            h[e] = __builtin_bit_cast(unsigned short, b);
            r[e] -= (float)b;
        }
        lo[t] = h[0] | ((unsigned)h[1] << 16);
        hi[t] = h[2] | ((unsigned)h[3] << 16);
    }
}
__global__ void k_wgrad(float* G, const float* H, const float* Dz, unsigned long long* cyc, float* dbgA) {   // H, Dz: [16][64]; G [64][64] = H^T Dz
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int l = threadIdx.x, i16 = l & 15, kk = l >> 4;
    unsigned char* TA = sm;               // 3 planes x 2 KB
    unsigned char* TB = sm + 3 * 2048;
    const unsigned long long t0 = clock64();
    // writer: chain layout
    for (int c = 0; c < 4; ++c) {
        float xa[4], xb[4];
        for (int r = 0; r < 4; ++r) {
            xa[r] = H[i16 * 64 + 16 * c + 4 * kk + r];
            xb[r] = Dz[i16 * 64 + 16 * c + 4 * kk + r];
        }
        unsigned lo[3], hi[3];
        split3(xa, lo, hi);
        for (int t = 0; t < 3; ++t) *(uint2*)(TA + 2048 * t + 8 * slot_of(i16, 4 * c + kk)) = make_uint2(lo[t], hi[t]);
        split3(xb, lo, hi);
        for (int t = 0; t < 3; ++t) *(uint2*)(TB + 2048 * t + 8 * slot_of(i16, 4 * c + kk)) = make_uint2(lo[t], hi[t]);
    }
    __syncthreads();
    // reader: lane (unit i = l & 31 of block b, k half kh): samples 8 kh + 4 t + (p >> 2), chunk 8 b + 4 g + (p & 3)
    const int g = (l >> 4) & 1, kh = l >> 5, p = l & 15;
    bf16x8 fa[2][3], fb[2][3];
    for (int b = 0; b < 2; ++b)
        for (int pl = 0; pl < 3; ++pl) {
            s16x4 v[2], w[2];
            for (int t = 0; t < 2; ++t) {
                const int off = 2048 * pl + 8 * slot_of(8 * kh + 4 * t + (p >> 2), 8 * b + 4 * g + (p & 3));
                v[t] = lds_tr16(TA + off);
                w[t] = lds_tr16(TB + off);
            }
            // (whole-vector casts: building a bf16x8 element by element from the i16 lanes was miscompiled -- elements 1..3 of
            //  each half came out equal)
            fa[b][pl] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7));
            fb[b][pl] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(w[0], w[1], 0, 1, 2, 3, 4, 5, 6, 7));
        }
    if (dbgA != nullptr)      // the A operand as the lanes see it: [b][lane][k-slot] = sum of the three planes
        for (int b = 0; b < 2; ++b)
            for (int e = 0; e < 8; ++e) dbgA[(b * 64 + l) * 8 + e] = (float)fa[b][0][e] + (float)fa[b][1][e] + (float)fa[b][2][e];
    const int TA_[6] = {1, 2, 0, 1, 0, 0}, TB_[6] = {1, 0, 2, 0, 1, 0};
    for (int bi = 0; bi < 2; ++bi)
        for (int bj = 0; bj < 2; ++bj) {
            f32x16 c = {};
            for (int pr = 0; pr < 6; ++pr) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[bi][TA_[pr]], fb[bj][TB_[pr]], c, 0, 0, 0);
            for (int r = 0; r < 16; ++r) G[(32 * bi + (r & 3) + 8 * (r >> 2) + 4 * kh) * 64 + 32 * bj + (l & 31)] = c[r];
        }
    if (l == 0) cyc[0] = clock64() - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    // (1)
    {
        std::vector<int> chunk(64);
        for (int l = 0; l < 64; ++l) chunk[l] = (l * 7 + 3) % 64 + 64 * (l % 3);       // scattered, distinct 8-byte chunks
        int* dch; unsigned short* dout;
        CK(hipMalloc(&dch, 64 * 4)); CK(hipMalloc(&dout, 256 * 2));
        CK(hipMemcpy(dch, chunk.data(), 64 * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 8192, 0, dout, dch);
        std::vector<unsigned short> out(256);
        CK(hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int grp = l >> 4, i = l & 15, src_lane = 16 * grp + 4 * j + (i >> 2), src_e = i & 3;
                const int expect = 4 * chunk[src_lane] + src_e;
                if (out[l * 4 + j] != expect) ++bad;
            }
        printf("(1) ds_read_b64_tr_b16: lane (g, i) elem j <- lane (g, 4 j + i / 4) elem i %% 4 : %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
        if (bad) {
            for (int l = 0; l < 64; ++l) {
                printf("  lane %2d:", l);
                for (int j = 0; j < 4; ++j) {
                    int sl = -1, se = -1;
                    for (int q = 0; q < 64; ++q)
                        if (out[l * 4 + j] / 4 == chunk[q]) { sl = q; se = out[l * 4 + j] % 4; }
                    printf(" (%d,%d)", sl, se);
                }
                printf("\n");
            }
        }
    }
    // (2)
    {
        std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32);
        srand(1);
        for (auto& x : A) x = (float)(rand() % 17 - 8);
        for (auto& x : B) x = (float)(rand() % 17 - 8);
        float *dA, *dB, *dD;
        CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_mfma32, dim3(1), dim3(64), 0, 0, dD, dA, dB);
        CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                float s = 0;
                for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 32 + j];
                if (s != D[i * 32 + j]) ++bad;
            }
        printf("(2) v_mfma_f32_32x32x16_bf16 layout (A[i = l&31][8 (l>>5) + e], D row (r&3) + 8 (r>>2) + 4 (l>>5), col l&31): %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
    }
    // (3)
    {
        std::vector<float> H(16 * 64), Dz(16 * 64), G(64 * 64);
        srand(7);
        for (auto& x : H) x = tanhf(3.f * ((float)rand() / RAND_MAX - 0.5f));
        for (auto& x : Dz) x = 1e-3f * ((float)rand() / RAND_MAX - 0.5f) * (rand() % 5 == 0 ? 30.f : 1.f);
        float *dH, *dZ, *dG, *dA; unsigned long long* dc;
        CK(hipMalloc(&dA, 2 * 64 * 8 * 4));
        CK(hipMalloc(&dH, H.size() * 4)); CK(hipMalloc(&dZ, Dz.size() * 4)); CK(hipMalloc(&dG, G.size() * 4)); CK(hipMalloc(&dc, 8));
        CK(hipMemcpy(dH, H.data(), H.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dZ, Dz.data(), Dz.size() * 4, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_wgrad, dim3(1), dim3(64), 12288, 0, dG, dH, dZ, dc, dA);
        CK(hipMemcpy(G.data(), dG, G.size() * 4, hipMemcpyDeviceToHost));
        unsigned long long cyc = 0;
        CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost));
        {
            std::vector<float> Aop(2 * 64 * 8);
            CK(hipMemcpy(Aop.data(), dA, Aop.size() * 4, hipMemcpyDeviceToHost));
            int badA = 0;
            for (int b = 0; b < 2; ++b)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e) {
                        const float want = H[(8 * (l >> 5) + e) * 64 + 32 * b + (l & 31)], got = Aop[(b * 64 + l) * 8 + e];
                        if (fabsf(want - got) > 1e-6f * fabsf(want) + 1e-12f) {
                            if (badA < 8) printf("    A operand b %d lane %d slot %d: want %g got %g\n", b, l, e, want, got);
                            ++badA;
                        }
                    }
            printf("    A operand (planes summed) vs H^T: %d mismatches\n", badA);
        }
        double maxref = 0, maxerr = 0, maxerr32 = 0;
        for (int m = 0; m < 64; ++m)
            for (int n = 0; n < 64; ++n) {
                double s = 0;
                float s32 = 0.f;
                for (int k = 0; k < 16; ++k) {
                    s += (double)H[k * 64 + m] * (double)Dz[k * 64 + n];
                    s32 = fmaf(H[k * 64 + m], Dz[k * 64 + n], s32);
                }
                maxref = fmax(maxref, fabs(s));
                maxerr = fmax(maxerr, fabs(s - (double)G[m * 64 + n]));
                maxerr32 = fmax(maxerr32, fabs(s - (double)s32));
            }
        printf("(3) weight gradient through swizzled LDS + transpose reads + 6 products: max err / max |ref| = %.3e (float32 fma chain: %.3e) : %s   [%llu cycles, one wave]\n",
               maxerr / maxref, maxerr32 / maxref, maxerr / maxref < 3e-7 ? "PASS" : "FAIL", cyc);
    }
    return 0;
}
