// hip_emu.h -- TEST INFRASTRUCTURE ONLY.  A SIMT interpreter just big enough to execute the kernels of
// promp_amd/csrc on the host: one OS thread per lane, 64-lane wavefronts, workgroup barriers, wave
// shuffles and the MFMA fragment layouts of gfx950 (restated from cdna_hip_programming.md section 3:
//   32x32x2 f32 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5)
//   16x16x4 f32 : A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D col=l&15, row=4*(l>>4)+r
//   16x16x4 f64 : same A/B,                               D col=l&15, row=(l>>4)+4*r
// each product accumulated as a k-ordered fma chain).
// It exists so that kernel indexing / host sequencing can be checked in a container without a GPU.
// It is built only by tests/emu/build_emu.py into tests/emu/libpromp_emu.so and is never loaded by
// promp_amd (the product binds libpromp_hip.so only and has no CPU path).
#pragma once
#include <algorithm>
#include <barrier>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#define PROMP_DEV inline
#define PROMP_DEV_NOINLINE inline
#define PROMP_HD inline
#define PROMP_CX constexpr
#define __global__
#define __device__
#define __host__
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef float f32x16 __attribute__((vector_size(64)));
typedef float f32x4 __attribute__((vector_size(16)));
typedef float f32x2 __attribute__((vector_size(8)));
typedef double f64x4 __attribute__((vector_size(32)));

namespace emu {

// Exchange slots of the wave-level instructions (MFMA operands, shuffles): two sets, used alternately.  Every lane counts its
// wave-level operations (TL::xpar); operation k writes set k & 1, meets the other lanes at ONE barrier and reads.  A second barrier
// behind the reads is not needed: a lane can only overwrite set k & 1 in operation k + 2, and it gets there only through the
// barrier of operation k + 1, which every lane reaches after it has finished reading in operation k.
struct Wave {
    std::barrier<> bar;
    float fa[2][64], fb[2][64];
    double da[2][64], db[2][64];
    unsigned short ha[2][64][8], hb[2][64][8];
    explicit Wave(int n) : bar(n) {}
};

struct Block {
    std::barrier<> bar;
    std::vector<std::unique_ptr<Wave>> waves;
    unsigned char* smem;
    int or_flag = 0;
    Block(int nthreads, size_t smem_bytes) : bar(nthreads) {
        for (int w = 0; w * 64 < nthreads; ++w) waves.emplace_back(new Wave(std::min(64, nthreads - w * 64)));
        smem = (unsigned char*)aligned_alloc(64, (smem_bytes + 1024 + 63) / 64 * 64);
        memset(smem, 0xFF, (smem_bytes + 1024 + 63) / 64 * 64);   // poison: uninitialised LDS reads show up as NaN
    }
    ~Block() { free(smem); }
};

struct TL {
    dim3 tidx, bidx, bdim, gdim;
    Block* blk = nullptr;
    int xpar = 0;          // parity of this lane's wave-level operation count (see Wave)
};
inline thread_local TL tl;

inline Wave& wave() { return *tl.blk->waves[tl.tidx.x >> 6]; }
inline int xslot() { const int p = tl.xpar; tl.xpar ^= 1; return p; }     // this operation's slot set; advances the count
inline int lane() { return tl.tidx.x & 63; }

template <class F>
void launch(dim3 grid, int block, size_t smem, F body) {
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned bx = 0; bx < grid.x; ++bx) {
            Block blk(block, smem);
            std::vector<std::thread> th;
            th.reserve(block);
            for (int t = 0; t < block; ++t)
                th.emplace_back([&, t]() {
                    tl.tidx = dim3(t);
                    tl.bidx = dim3(bx, by, bz);
                    tl.bdim = dim3(block);
                    tl.gdim = grid;
                    tl.blk = &blk;
                    tl.xpar = 0;
                    body();
                });
            for (auto& x : th) x.join();
        }
}

}  // namespace emu

#define threadIdx (emu::tl.tidx)
#define blockIdx (emu::tl.bidx)
#define blockDim (emu::tl.bdim)
#define gridDim (emu::tl.gdim)
#define PROMP_SMEM_DECL (void)0
#define PROMP_SMEM_PTR (emu::tl.blk->smem)
#define PROMP_LAUNCH(kern, grid, block, smem, stream, ...) emu::launch(grid, block, smem, [=]() { kern(__VA_ARGS__); })

inline void __syncthreads() { emu::tl.blk->bar.arrive_and_wait(); }
inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline int __syncthreads_or(int pred) {
    emu::Block* b = emu::tl.blk;
    b->bar.arrive_and_wait();
    if (pred) __atomic_store_n(&b->or_flag, 1, __ATOMIC_SEQ_CST);
    b->bar.arrive_and_wait();
    const int r = __atomic_load_n(&b->or_flag, __ATOMIC_SEQ_CST);
    b->bar.arrive_and_wait();
    if (emu::tl.tidx.x == 0) b->or_flag = 0;
    b->bar.arrive_and_wait();
    return r;
}

inline f32x16 mfma32(float a, float b, f32x16 c) {
    emu::Wave& W = emu::wave();
    const int l = emu::lane(), j = l & 31, h = l >> 5, s = emu::xslot();
    W.fa[s][l] = a;
    W.fb[s][l] = b;
    W.bar.arrive_and_wait();
    f32x16 d;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        float acc = c[r];
        acc = fmaf(W.fa[s][i], W.fb[s][j], acc);
        acc = fmaf(W.fa[s][i + 32], W.fb[s][j + 32], acc);
        d[r] = acc;
    }
    return d;
}
inline f32x4 mfma16(float a, float b, f32x4 c) {
    emu::Wave& W = emu::wave();
    const int l = emu::lane(), j = l & 15, g = l >> 4, s = emu::xslot();
    W.fa[s][l] = a;
    W.fb[s][l] = b;
    W.bar.arrive_and_wait();
    f32x4 d;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(W.fa[s][i + 16 * k], W.fb[s][j + 16 * k], acc);
        d[r] = acc;
    }
    return d;
}
// BF16 operands of v_mfma_f32_16x16x32_bf16: 8 values per lane and operand (k = 8 (l / 16) .. + 7 of row / column l % 16;
// layout confirmed on the device by tools/micro/bf16_layout_probe.hip); exact products, float32 accumulation
struct bf16x8 {
    unsigned short h[8];
};
inline unsigned short emu_f2bf(float x) {              // round to nearest even, as v_cvt_pk_bf16_f32
    unsigned u;
    memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
inline float emu_bf2f(unsigned short h) {
    const unsigned u = (unsigned)h << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}
inline void bf16_split3(const float (&x)[8], bf16x8 (&t)[3]) {
    for (int e = 0; e < 8; ++e) {
        float r = x[e];
        for (int s = 0; s < 3; ++s) {
            t[s].h[e] = emu_f2bf(r);
            r -= emu_bf2f(t[s].h[e]);
        }
    }
}
inline f32x4 mfma16_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    emu::Wave& W = emu::wave();
    const int l = emu::lane(), j = l & 15, g = l >> 4, s = emu::xslot();
    for (int e = 0; e < 8; ++e) {
        W.ha[s][l][e] = a.h[e];
        W.hb[s][l][e] = b.h[e];
    }
    W.bar.arrive_and_wait();
    f32x4 d;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        float acc = c[r];
        for (int kb = 0; kb < 4; ++kb)
            for (int e = 0; e < 8; ++e) acc += emu_bf2f(W.ha[s][i + 16 * kb][e]) * emu_bf2f(W.hb[s][j + 16 * kb][e]);   // (products exact in float32)
        d[r] = acc;
    }
    return d;
}
// ---- word-packed BF16 fragments (two bf16 per 32-bit word, low half first), see promp_device.h ----
typedef unsigned u32x4 __attribute__((vector_size(16)));
typedef unsigned u32x2 __attribute__((vector_size(8)));
inline unsigned short emu_word_half(const u32x4& v, int e) { return (unsigned short)(v[e >> 1] >> (16 * (e & 1))); }
inline f32x4 mfma16_bf16w(u32x4 a, u32x4 b, f32x4 c) {
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) {
        x.h[e] = emu_word_half(a, e);
        y.h[e] = emu_word_half(b, e);
    }
    return mfma16_bf16(x, y, c);
}
// v_mfma_f32_32x32x16_bf16: A[i = l & 31][k = 8 (l >> 5) + e], B[k][j = l & 31]; D col = l & 31, row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)
inline f32x16 mfma32_bf16w(u32x4 a, u32x4 b, f32x16 c) {
    emu::Wave& W = emu::wave();
    const int l = emu::lane(), j = l & 31, h = l >> 5, s = emu::xslot();
    for (int e = 0; e < 8; ++e) {
        W.ha[s][l][e] = emu_word_half(a, e);
        W.hb[s][l][e] = emu_word_half(b, e);
    }
    W.bar.arrive_and_wait();
    f32x16 d;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        float acc = c[r];
        for (int kb = 0; kb < 2; ++kb)
            for (int e = 0; e < 8; ++e) acc += emu_bf2f(W.ha[s][i + 32 * kb][e]) * emu_bf2f(W.hb[s][j + 32 * kb][e]);
        d[r] = acc;
    }
    return d;
}
// ds_read_b64_tr_b16 (semantics measured on the device, tools/micro/tr16_wgrad_probe.hip): lane i of a 16-lane group receives
// element i % 4 of the 8-byte chunks named by the lanes i / 4, 4 + i / 4, 8 + i / 4, 12 + i / 4 of its group
inline u32x2 lds_tr16(const void* p) {
    emu::Wave& W = emu::wave();
    const int l = emu::lane(), g = l & ~15, i = l & 15;
    if (((uintptr_t)p & 7) != 0) { fprintf(stderr, "lds_tr16: address not 8-byte aligned\n"); abort(); }
    // the hardware executes a wave's LDS instructions in order, so the chunks other lanes stored before this read are there;
    // here the lanes are free-running threads: wait until every lane has reached the read (= has done its stores)
    W.bar.arrive_and_wait();
    const int s = emu::xslot();
    memcpy(W.ha[s][l], p, 8);
    W.bar.arrive_and_wait();
    unsigned short v[4];
    for (int jj = 0; jj < 4; ++jj) v[jj] = W.ha[s][g + 4 * jj + (i >> 2)][i & 3];
    u32x2 r;
    r[0] = v[0] | ((unsigned)v[1] << 16);
    r[1] = v[2] | ((unsigned)v[3] << 16);
    return r;
}
inline void bf16_split3_pair(float x0, float x1, unsigned (&w)[3]) {
    float r0 = x0, r1 = x1;
    for (int t = 0; t < 3; ++t) {
        const unsigned short h0 = emu_f2bf(r0), h1 = emu_f2bf(r1);
        w[t] = h0 | ((unsigned)h1 << 16);
        r0 -= emu_bf2f(h0);
        r1 -= emu_bf2f(h1);
    }
}
// ---- the split of round 6 (promp_device.h: split_pair / mfma16_sw / mfma32_sw) -------------------------------------------
// FP16: round to nearest even with subnormal results kept (v_cvt_pk_f16_f32 under the kernels' default mode), subnormal inputs kept
// by the matrix instruction (tools/micro/f16_split_probe.hip); overflow -> inf.
inline unsigned short emu_f2h(float x) {
    unsigned u;
    memcpy(&u, &x, 4);
    const unsigned sign = (u >> 16) & 0x8000u, au = u & 0x7FFFFFFFu;
    if (au >= 0x7F800000u) return (unsigned short)(sign | 0x7C00u | ((au > 0x7F800000u) ? 0x200u : 0u));
    const int e = (int)(au >> 23) - 127;
    if (e > 15) return (unsigned short)(sign | 0x7C00u);
    unsigned mant = (au & 0x7FFFFFu) | 0x800000u;           // 24 significant bits
    int shift;                                              // bits dropped from the 24-bit significand
    unsigned hexp;
    if (e >= -14) { shift = 13; hexp = (unsigned)(e + 15); }
    else { shift = 13 + (-14 - e); hexp = 0; }
    if (au < 0x00800000u || shift > 25) return (unsigned short)sign;      // (float32 subnormals and anything below 2^-26: zero)
    const unsigned q = mant >> shift, rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
    unsigned r = q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u);
    // normal: q carries the implicit bit (0x400); a carry out of the significand bumps the exponent; subnormal: r is the field
    unsigned h = hexp ? ((hexp - 1u) << 10) + r : r;
    if (h >= 0x7C00u) h = 0x7C00u;
    return (unsigned short)(sign | h);
}
inline float emu_h2f(unsigned short h) {
    const unsigned sign = ((unsigned)h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    float x;
    if (e == 0) x = ldexpf((float)m, -24);
    else if (e == 31) x = m ? NAN : INFINITY;
    else x = ldexpf((float)(m | 0x400u), (int)e - 25);
    return sign ? -x : x;
}
inline unsigned pk_scale_f16(unsigned w, float p2) {
    const unsigned short lo = emu_f2h(emu_h2f((unsigned short)(w & 0xFFFFu)) * p2), hi = emu_f2h(emu_h2f((unsigned short)(w >> 16)) * p2);
    return lo | ((unsigned)hi << 16);
}
template <int NT>
inline float emu_half2f(unsigned short h) { return NT == 2 ? emu_h2f(h) : emu_bf2f(h); }
template <int NT>
inline void split_pair(float x0, float x1, unsigned (&w)[NT]) {
    float r0 = x0, r1 = x1;
    for (int t = 0; t < NT; ++t) {
        const unsigned short h0 = NT == 2 ? emu_f2h(r0) : emu_f2bf(r0), h1 = NT == 2 ? emu_f2h(r1) : emu_f2bf(r1);
        w[t] = h0 | ((unsigned)h1 << 16);
        r0 -= emu_half2f<NT>(h0);
        r1 -= emu_half2f<NT>(h1);
    }
}
template <int NT>
inline f32x4 mfma16_sw(u32x4 a, u32x4 b, f32x4 c) {
    emu::Wave& W = emu::wave();
    const int l = emu::lane(), j = l & 15, g = l >> 4, s = emu::xslot();
    for (int e = 0; e < 8; ++e) {
        W.ha[s][l][e] = emu_word_half(a, e);
        W.hb[s][l][e] = emu_word_half(b, e);
    }
    W.bar.arrive_and_wait();
    f32x4 d;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        float acc = c[r];
        for (int kb = 0; kb < 4; ++kb)
            for (int e = 0; e < 8; ++e) acc += emu_half2f<NT>(W.ha[s][i + 16 * kb][e]) * emu_half2f<NT>(W.hb[s][j + 16 * kb][e]);   // (products exact in float32)
        d[r] = acc;
    }
    return d;
}
template <int NT>
inline f32x16 mfma32_sw(u32x4 a, u32x4 b, f32x16 c) {
    emu::Wave& W = emu::wave();
    const int l = emu::lane(), j = l & 31, h = l >> 5, s = emu::xslot();
    for (int e = 0; e < 8; ++e) {
        W.ha[s][l][e] = emu_word_half(a, e);
        W.hb[s][l][e] = emu_word_half(b, e);
    }
    W.bar.arrive_and_wait();
    f32x16 d;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        float acc = c[r];
        for (int kb = 0; kb < 2; ++kb)
            for (int e = 0; e < 8; ++e) acc += emu_half2f<NT>(W.ha[s][i + 32 * kb][e]) * emu_half2f<NT>(W.hb[s][j + 32 * kb][e]);
        d[r] = acc;
    }
    return d;
}
inline f64x4 mfma16d(double a, double b, f64x4 c) {
    emu::Wave& W = emu::wave();
    const int l = emu::lane(), j = l & 15, g = l >> 4, s = emu::xslot();
    W.da[s][l] = a;
    W.db[s][l] = b;
    W.bar.arrive_and_wait();
    f64x4 d;
    for (int r = 0; r < 4; ++r) {
        const int i = g + 4 * r;
        double acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fma(W.da[s][i + 16 * k], W.db[s][j + 16 * k], acc);
        d[r] = acc;
    }
    return d;
}

inline double emu_shfl_f64(double v, int src) {
    emu::Wave& W = emu::wave();
    const int l = emu::lane(), s = emu::xslot();
    W.da[s][l] = v;
    W.bar.arrive_and_wait();
    return (src >= 0 && src < 64) ? W.da[s][src] : v;
}
inline float shfl_xor_f32(float v, int m) { return (float)emu_shfl_f64((double)v, emu::lane() ^ m); }
inline double shfl_xor_f64(double v, int m) { return emu_shfl_f64(v, emu::lane() ^ m); }
inline float fold_groups16(float v) {
    v += shfl_xor_f32(v, 32);
    v += shfl_xor_f32(v, 16);
    return v;
}
inline float row16_sum(float v) {
    for (int m = 1; m <= 8; m <<= 1) v += shfl_xor_f32(v, m);
    return v;
}
inline double shfl_down_f64(double v, int d) { return emu_shfl_f64(v, emu::lane() + d); }
inline double shfl_idx_f64(double v, int l) { return emu_shfl_f64(v, l); }
inline double readlane_f64(double v, int l) { return emu_shfl_f64(v, l); }
inline bool wave_any(bool p) {
    double m = p ? 1.0 : 0.0;
    for (int s = 32; s >= 1; s >>= 1) m = std::max(m, emu_shfl_f64(m, emu::lane() ^ s));
    return m > 0.0;
}
inline void wave_sync() { emu::wave().bar.arrive_and_wait(); }
inline void lds_barrier() { emu::tl.blk->bar.arrive_and_wait(); }
inline void wave_fence() { emu::wave().bar.arrive_and_wait(); }
// two-wave rendezvous through LDS words (promp_device.h): every lane of the posting wave has done its LDS traffic before lane 0
// raises the flag; the waiting lanes spin on it
inline void pair_post(float* flags, int mine, int seq, int lane) {
    emu::wave().bar.arrive_and_wait();
    if (lane == 0) __atomic_store_n((int*)flags + mine, seq, __ATOMIC_SEQ_CST);
}
inline void pair_wait(float* flags, int other, int seq) {
    while (__atomic_load_n((int*)flags + other, __ATOMIC_SEQ_CST) < seq) std::this_thread::yield();
    emu::wave().bar.arrive_and_wait();
}
inline void fence_release_agent() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void fence_acquire_agent() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline int atomic_add_agent(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline void atomic_store_agent(int* p, int v) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
inline void atomic_max_agent(unsigned* p, unsigned v) {
    unsigned cur = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (cur < v && !__atomic_compare_exchange_n(p, &cur, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
}
inline void sched_fence() {}
inline int opaque_zero() { return 0; }
inline double* opaque_lds(double* p) { return p; }
inline f32x4 pin_agpr(f32x4 v) { return v; }
inline int wave_uniform(int v) { return v; }
// the product's XCD-aware work-item bijection (promp_device.h), so that the emulated kernels take the same items
inline int xcd_item(int b, int G) {
    const int x = b & 7;
    int off = 0;
    for (int y = 0; y < 7; ++y) off += (y < x) ? ((G - y + 7) >> 3) : 0;
    return off + (b >> 3);
}
#define PROMP_SCHED_MFMA(n) ((void)0)
#define PROMP_SCHED_VALU(n) ((void)0)
#define PROMP_SCHED_DSREAD(n) ((void)0)
#define PROMP_SCHED_DSWRITE(n) ((void)0)
template <class T> inline void pin_v(T&) {}
template <class T> inline void pin_a(T&) {}
template <class T> inline void pin_s(T&) {}
inline unsigned long long promp_clock() { return 0; }
inline unsigned long long promp_wall_clock() { return 0; }
inline double rsqrt(double x) { return 1.0 / sqrt(x); }
inline double rsq_seed(double d) { return 1.0 / sqrt(d); }
inline double rsq_e(double d, double r0) { return fma(-d * r0, r0, 1.0); }
inline double rsq_finish(double r0, double e) { return fma(r0 * e, fma(e, 0.375, 0.5), r0); }
inline float fast_exp(float x) { return expf(x); }
inline float fast_rcp(float x) { return 1.0f / x; }
inline float fast_exp2(float x) { return exp2f(x); }
inline void wave_priority(int) {}
inline void release_store_system(unsigned* p, unsigned v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
inline f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { f32x2 r; r[0] = fmaf(a[0], b[0], c[0]); r[1] = fmaf(a[1], b[1], c[1]); return r; }

// ---- host runtime shims -------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorNotReady = 600 };
inline hipError_t hipStreamQuery(void*) { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emulated-hip-error"; }
typedef void* hipStream_t;
struct EmuEvent { std::chrono::steady_clock::time_point t; };
typedef EmuEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { char name[256]; char gcnArchName[64]; int multiProcessorCount; int clockRate; };
inline hipError_t hipGetDeviceCount(int* n) {      // PROMP_EMU_DEVICES: the ranks of a several-process test each take "their" device
    const char* e = getenv("PROMP_EMU_DEVICES");
    *n = e ? atoi(e) : 1;
    return 0;
}
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    snprintf(p->name, sizeof p->name, "kernel-emulator (tests only)");
    snprintf(p->gcnArchName, sizeof p->gcnArchName, "emulator");
    const char* e = getenv("PROMP_EMU_CUS");
    p->multiProcessorCount = e ? atoi(e) : 4;
    p->clockRate = 1000;
    return 0;
}
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = (void*)1; return 0; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return 0; }
constexpr unsigned hipStreamDefault = 0;
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (void*)1; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(64, (n + 63) / 64 * 64); return *p ? 0 : 1; }
inline hipError_t hipFree(void* p) { free(p); return 0; }
inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return 0; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return 0; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new EmuEvent(); return 0; }
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new EmuEvent(); return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
enum { hipHostMallocDefault = 0 };
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n); return *p ? 0 : 2; }
inline hipError_t hipHostFree(void* p) { free(p); return 0; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return 0; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return 0;
}

// ---- RCCL shim ------------------------------------------------------------------------------------
// The host code calls RCCL unconditionally (same sequence as the product).  One rank: all-reduce = identity.  Several ranks
// (processes of this host, e.g. tests that run bench.py --gpus 2 on this library): the communicator is a POSIX shared-memory
// segment named after the unique id -- one slot per rank, a sense-reversing barrier in the segment -- and a collective is
// "write my slot, meet, combine the slots in rank order, meet".  Synchronous like everything else in this interpreter.
#include <atomic>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>
typedef struct EmuComm* ncclComm_t;
struct EmuShm {
    std::atomic<int> arrived, generation;
    char pad[56];
};
struct EmuComm {
    int nranks, rank;
    EmuShm* sh;
    unsigned char* slots;
    size_t slot_bytes, map_bytes;
    char name[64];
};
struct ncclUniqueId { char internal[128]; };
typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclInvalidUsage = 5, ncclSystemError = 2 };
enum ncclDataType_t { ncclFloat, ncclDouble };
enum ncclRedOp_t { ncclSum, ncclMax };
inline const char* ncclGetErrorString(ncclResult_t) { return "kernel-emulation RCCL shim failed (shared-memory communicator)"; }
inline ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/promp_emu_%d_%lld", (int)getpid(),
             (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}
inline void emu_comm_barrier(EmuComm* c) {
    const int gen = c->sh->generation.load();
    if (c->sh->arrived.fetch_add(1) + 1 == c->nranks) {
        c->sh->arrived.store(0);
        c->sh->generation.fetch_add(1);
    } else {
        while (c->sh->generation.load() == gen) sched_yield();
    }
}
inline ncclResult_t ncclCommInitRank(ncclComm_t* c, int nranks, ncclUniqueId id, int rank) {
    EmuComm* m = new EmuComm{nranks, rank, nullptr, nullptr, 0, 0, {0}};
    if (nranks > 1) {
        id.internal[sizeof(m->name) - 1] = 0;
        snprintf(m->name, sizeof m->name, "%s", id.internal);
        m->slot_bytes = 1 << 20;
        m->map_bytes = sizeof(EmuShm) + (size_t)nranks * m->slot_bytes;
        const int fd = shm_open(m->name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)m->map_bytes) != 0) { delete m; return ncclSystemError; }
        void* p = mmap(nullptr, m->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) { delete m; return ncclSystemError; }
        m->sh = (EmuShm*)p;                    // (a fresh segment is zero-filled: arrived = generation = 0)
        m->slots = (unsigned char*)p + sizeof(EmuShm);
        emu_comm_barrier(m);
    }
    *c = m;
    return ncclSuccess;
}
inline ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (c->nranks > 1) {
        emu_comm_barrier(c);
        if (c->rank == 0) shm_unlink(c->name);
        munmap((void*)c->sh, c->map_bytes);
    }
    delete c;
    return ncclSuccess;
}
inline ncclResult_t ncclCommCount(ncclComm_t c, int* n) { *n = c->nranks; return ncclSuccess; }
inline ncclResult_t ncclCommUserRank(ncclComm_t c, int* r) { *r = c->rank; return ncclSuccess; }
inline ncclResult_t ncclAllGather(const void* in, void* out, size_t n, ncclDataType_t t, ncclComm_t c, hipStream_t) {
    const size_t bytes = n * (t == ncclDouble ? 8 : 4);
    if (c->nranks == 1) {
        if (in != out) memcpy(out, in, bytes);
        return ncclSuccess;
    }
    if (bytes > c->slot_bytes) return ncclInvalidUsage;
    memcpy(c->slots + (size_t)c->rank * c->slot_bytes, in, bytes);
    emu_comm_barrier(c);
    for (int r = 0; r < c->nranks; ++r) memcpy((unsigned char*)out + (size_t)r * bytes, c->slots + (size_t)r * c->slot_bytes, bytes);
    emu_comm_barrier(c);
    return ncclSuccess;
}
inline ncclResult_t ncclAllReduce(const void* in, void* out, size_t n, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t) {
    const size_t bytes = n * (t == ncclDouble ? 8 : 4);
    if (c->nranks == 1) {
        if (in != out) memcpy(out, in, bytes);
        return ncclSuccess;
    }
    if (bytes > c->slot_bytes) return ncclInvalidUsage;
    memcpy(c->slots + (size_t)c->rank * c->slot_bytes, in, bytes);
    emu_comm_barrier(c);
    for (size_t i = 0; i < n; ++i) {           // rank order 0, 1, ...: every rank computes the same bits
        if (t == ncclDouble) {
            double acc = ((const double*)c->slots)[i];
            for (int r = 1; r < c->nranks; ++r) {
                const double x = ((const double*)(c->slots + (size_t)r * c->slot_bytes))[i];
                acc = op == ncclMax ? (x > acc ? x : acc) : acc + x;
            }
            ((double*)out)[i] = acc;
        } else {
            float acc = ((const float*)c->slots)[i];
            for (int r = 1; r < c->nranks; ++r) {
                const float x = ((const float*)(c->slots + (size_t)r * c->slot_bytes))[i];
                acc = op == ncclMax ? (x > acc ? x : acc) : acc + x;
            }
            ((float*)out)[i] = acc;
        }
    }
    emu_comm_barrier(c);
    return ncclSuccess;
}
inline hipError_t hipDeviceGetPCIBusId(char* out, int n, int dev) { snprintf(out, n, "emu:%02d:00.0", dev); return 0; }
