// tests/emu/msmc_rt.hpp -- CPU interpreter vocabulary for the kernel sources in msmc-tts_amd/csrc.
//
// TEST INFRASTRUCTURE ONLY.  The -m "not gpu" tests build the *same* kernel sources against this
// header (-I tests/emu instead of -I csrc/gfx950) into tests/emu/libmsmc_emu.so and
// run them on host memory, one workgroup at a time, with every work-item a cooperative fiber.  It
// exists because there is no GPU in the build container: it checks indexing, tiling, LDS hand-offs,
// wave reductions and the documented MFMA fragment layouts (MI355X guide, section 3) before a kernel
// costs GPU minutes.  It proves nothing about performance, real-hardware races or alignment.
// The product never loads this library (msmctts_amd/hip/lib.py refuses host tensors).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
typedef void* hipStream_t;
typedef hipStream_t msmc_stream_t;

namespace emu {
extern dim3 tid, bid, bdim, gdim;
extern char* dyn_lds;
void block_barrier();
void wave_barrier();
int lane();
int wave();
uint32_t* slot(int lane_index);            // 64 x 32-byte exchange slots of the calling wave (double-buffered)
bool first_after_barrier();                // true for the first lane of the wave to resume in this collective
void collective_done();                    // the calling lane has finished reading this collective's slots
void run_grid(dim3 grid, dim3 block, size_t lds_bytes, void (*body)(void*), void* closure);
}  // namespace emu

#define threadIdx emu::tid
#define blockIdx emu::bid
#define blockDim emu::bdim
#define gridDim emu::gdim

#define MSMC_WAVE 64
#define MSMC_DEV static inline
#define MSMC_DEV_INLINE inline
#define MSMC_DYN_LDS(name) char* name = emu::dyn_lds

template <class F>
static void emu_trampoline(void* p) { (*(F*)p)(); }

// launch log of the interpreter build (msmc_prof_*): names only -- every record reads back as 1 ms -- so that the host-side
// attribution of bench.py (which launches belong to which call) can be tested without a GPU
struct MsmcProfRec { char name[120]; };
#define MSMC_PROF_MAX 16384
struct MsmcProfLog { int on = 0, n = 0; MsmcProfRec* rec = nullptr; };
inline MsmcProfLog msmc_prof_log;
inline int msmc_prof_mine = -1;
static inline void msmc_prof_note(const char* text) {
    MsmcProfLog& L = msmc_prof_log;
    msmc_prof_mine = L.n < MSMC_PROF_MAX ? L.n++ : -1;
    if (msmc_prof_mine < 0) return;
    MsmcProfRec& r = L.rec[msmc_prof_mine];
    int j = 0;
    for (const char* c = text; *c && *c != '<' && j < (int)sizeof(r.name) - 1; ++c)
        if (*c != '(' && *c != ' ') r.name[j++] = *c;
    r.name[j] = 0;
}
#define MSMC_LAUNCH(kernel, grid, block, lds, stream, ...)                                   \
    do {                                                                                       \
        if (msmc_prof_log.on) msmc_prof_note(#kernel);                                         \
        auto emu_body = [&]() { kernel(__VA_ARGS__); };                                        \
        emu::run_grid(grid, block, lds, &emu_trampoline<decltype(emu_body)>, (void*)&emu_body); \
    } while (0)

static inline void __syncthreads() { emu::block_barrier(); }

template <class T>
static inline T emu_exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 128, "slot too small");
    memcpy(emu::slot(emu::lane()), &v, sizeof(T));
    emu::wave_barrier();
    T r;
    memcpy(&r, emu::slot(src_lane & 63), sizeof(T));
    emu::collective_done();
    return r;
}
static inline int wave_uniform(int v) { return v; }
MSMC_DEV float wave_xor(float v, int mask) { return emu_exchange(v, emu::lane() ^ mask); }
MSMC_DEV int wave_xor(int v, int mask) { return emu_exchange(v, emu::lane() ^ mask); }
static inline unsigned int __umulhi(unsigned int a, unsigned int b) { return (unsigned int)(((unsigned long long)a * b) >> 32); }
MSMC_DEV float wave_sum(float v) {
    for (int m = 1; m < 64; m <<= 1) v = v + emu_exchange(v, emu::lane() ^ m);
    return v;
}
MSMC_DEV int wave_read_lane(int v, int src_lane) { return emu_exchange(v, src_lane); }
MSMC_DEV unsigned int wave_read_lane(unsigned int v, int src_lane) { return emu_exchange(v, src_lane); }
MSMC_DEV void wave_swap32(unsigned int& a, unsigned int& b) {
    const unsigned int pa = emu_exchange(a, emu::lane() ^ 32), pb = emu_exchange(b, emu::lane() ^ 32);
    if (emu::lane() < 32) b = pa;
    else a = pb;
}
MSMC_DEV float wave_xor16(float v) { return emu_exchange(v, emu::lane() ^ 16); }
MSMC_DEV float wave_xor32(float v) { return emu_exchange(v, emu::lane() ^ 32); }
MSMC_DEV int wave_xor16(int v) { return emu_exchange(v, emu::lane() ^ 16); }
MSMC_DEV int wave_xor32(int v) { return emu_exchange(v, emu::lane() ^ 32); }
MSMC_DEV float wave_down(float v, int delta) {
    int s = emu::lane() + delta;
    return emu_exchange(v, s < 64 ? s : emu::lane());
}
MSMC_DEV float wave_bcast(float v, int lane) { return emu_exchange(v, lane); }
MSMC_DEV int wave_bcast(int v, int lane) { return emu_exchange(v, lane); }
MSMC_DEV float wave_bcast_var(float v, int src_lane) { return emu_exchange(v, src_lane); }

MSMC_DEV bool wave_any(bool p) {
    int v = p ? 1 : 0;
    for (int m = 1; m < 64; m <<= 1) v |= emu_exchange(v, emu::lane() ^ m);      // butterfly OR over the 64 lanes
    return v != 0;
}
MSMC_DEV float fmax_raw(float a, float b) { return a > b ? a : b; }
MSMC_DEV float bits_and_or(float a, unsigned int keep, unsigned int ins) {
    unsigned int u;
    memcpy(&u, &a, 4);
    u = (u & keep) | ins;
    float r;
    memcpy(&r, &u, 4);
    return r;
}
MSMC_DEV float fmed3(float a, float b, float c) {
    const float lo = fminf(a, b), hi = fmaxf(a, b);
    return fmaxf(lo, fminf(hi, c));
}

MSMC_DEV void wave_sync() { emu::wave_barrier(); }   // fibers are not lock-step: make it a real barrier

// ---- MFMA, by the fragment layouts documented in csrc/gfx950/msmc_rt.hpp --------------------------
// Every lane deposits (a, b, c) in its slot; after the wave barrier the FIRST lane to resume computes
// the whole tile D = A.B + C for all 64 lanes (plain loops), the others just read their registers.
static inline float emu_bf16_f32(__bf16 h) {
    unsigned short s;
    memcpy(&s, &h, 2);
    unsigned int u = ((unsigned int)s) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
struct emu_f32_slot { float a, b, c[16]; };
struct emu_b16_slot { bf16x8 a, b; float c[16]; };

template <int M, int KG, int NREG>      // M x M tile, KG lane groups along k, NREG accumulators per lane
static inline void emu_mfma_f32_all() {
    float A[32][4], B[4][32];
    for (int l = 0; l < 64; ++l) {
        const emu_f32_slot* s = (const emu_f32_slot*)emu::slot(l);
        A[l % M][l / M] = s->a;
        B[l / M][l % M] = s->b;
    }
    for (int l = 0; l < 64; ++l) {
        emu_f32_slot* s = (emu_f32_slot*)emu::slot(l);
        const int col = l % M;
        for (int r = 0; r < NREG; ++r) {
            const int row = (M == 32) ? (r & 3) + 8 * (r >> 2) + 4 * (l >> 5) : 4 * (l >> 4) + r;
            float acc = s->c[r];
            for (int k = 0; k < KG; ++k) acc = fmaf(A[row][k], B[k][col], acc);
            s->c[r] = acc;
        }
    }
}
template <int M, int KG, int NREG>
static inline void emu_mfma_b16_all() {
    static float A[32][32], B[32][32];
    for (int l = 0; l < 64; ++l) {
        const emu_b16_slot* s = (const emu_b16_slot*)emu::slot(l);
        for (int e = 0; e < 8; ++e) {
            A[l % M][8 * (l / M) + e] = emu_bf16_f32(s->a[e]);
            B[8 * (l / M) + e][l % M] = emu_bf16_f32(s->b[e]);
        }
    }
    for (int l = 0; l < 64; ++l) {
        emu_b16_slot* s = (emu_b16_slot*)emu::slot(l);
        const int col = l % M;
        for (int r = 0; r < NREG; ++r) {
            const int row = (M == 32) ? (r & 3) + 8 * (r >> 2) + 4 * (l >> 5) : 4 * (l >> 4) + r;
            float acc = s->c[r];
            for (int k = 0; k < 8 * KG; ++k) acc += A[row][k] * B[k][col];
            s->c[r] = acc;
        }
    }
}
MSMC_DEV f32x4 mfma_f32_16x16x4(float a, float b, f32x4 c) {
    emu_f32_slot* me = (emu_f32_slot*)emu::slot(emu::lane());
    me->a = a; me->b = b;
    for (int r = 0; r < 4; ++r) me->c[r] = c[r];
    emu::wave_barrier();
    if (emu::first_after_barrier()) emu_mfma_f32_all<16, 4, 4>();
    me = (emu_f32_slot*)emu::slot(emu::lane());
    for (int r = 0; r < 4; ++r) c[r] = me->c[r];
    emu::collective_done();
    return c;
}
MSMC_DEV f32x16 mfma_f32_32x32x2(float a, float b, f32x16 c) {
    emu_f32_slot* me = (emu_f32_slot*)emu::slot(emu::lane());
    me->a = a; me->b = b;
    for (int r = 0; r < 16; ++r) me->c[r] = c[r];
    emu::wave_barrier();
    if (emu::first_after_barrier()) emu_mfma_f32_all<32, 2, 16>();
    me = (emu_f32_slot*)emu::slot(emu::lane());
    for (int r = 0; r < 16; ++r) c[r] = me->c[r];
    emu::collective_done();
    return c;
}
MSMC_DEV f32x16 mfma_bf16_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
    emu_b16_slot* me = (emu_b16_slot*)emu::slot(emu::lane());
    me->a = a; me->b = b;
    for (int r = 0; r < 16; ++r) me->c[r] = c[r];
    emu::wave_barrier();
    if (emu::first_after_barrier()) emu_mfma_b16_all<32, 2, 16>();
    me = (emu_b16_slot*)emu::slot(emu::lane());
    for (int r = 0; r < 16; ++r) c[r] = me->c[r];
    emu::collective_done();
    return c;
}
MSMC_DEV f32x4 mfma_bf16_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
    emu_b16_slot* me = (emu_b16_slot*)emu::slot(emu::lane());
    me->a = a; me->b = b;
    for (int r = 0; r < 4; ++r) me->c[r] = c[r];
    emu::wave_barrier();
    if (emu::first_after_barrier()) emu_mfma_b16_all<16, 4, 4>();
    me = (emu_b16_slot*)emu::slot(emu::lane());
    for (int r = 0; r < 4; ++r) c[r] = me->c[r];
    emu::collective_done();
    return c;
}

// ds_read_b64_tr_b16 by the semantics documented in csrc/gfx950/msmc_rt.hpp (hardware-verified)
MSMC_DEV u16x4 lds_read_tr16(const unsigned short* p) {
    u16x4 mine;
    memcpy(&mine, p, 8);
    memcpy(emu::slot(emu::lane()), &mine, 8);
    emu::wave_barrier();
    const int l = emu::lane(), L = l & 15, grp = l >> 4;
    u16x4 r;
    for (int j = 0; j < 4; ++j) {
        u16x4 src;
        memcpy(&src, emu::slot(16 * grp + 4 * j + (L >> 2)), 8);
        r[j] = src[L & 3];
    }
    emu::collective_done();
    return r;
}

// LDS-DMA: destination = wave-uniform base + 16 * lane (fibers copy at once; the wait is a no-op)
MSMC_DEV void lds_dma16(const void* gsrc, void* lds_wave_base) { memcpy((char*)lds_wave_base + 16 * emu::lane(), gsrc, 16); }
MSMC_DEV void lds_dma_wait() {}
template <int N> MSMC_DEV void lds_dma_wait_n() {}
MSMC_DEV void block_sync_lds() { emu::block_barrier(); }
MSMC_DEV u32x2 global_load8_async(const void* p) { return *(const u32x2*)p; }
MSMC_DEV void vm_pin(u32x2&) {}
MSMC_DEV void sched_fence() {}
MSMC_DEV u16x8 lds_read128_async(const void* p) { u16x8 v; memcpy(&v, p, 16); return v; }
MSMC_DEV u32x4 lds_read128_async4(const void* p) { u32x4 v; memcpy(&v, p, 16); return v; }
template <int OFF> MSMC_DEV u32x4 lds_read128_async4_off(const void* p) { u32x4 v; memcpy(&v, (const char*)p + OFF, 16); return v; }
MSMC_DEV int lds_read32_async(const void* p) { int v; memcpy(&v, p, 4); return v; }
template <int OFF> MSMC_DEV u32x2 lds_read_tr16_async(const void* p) {
    return __builtin_bit_cast(u32x2, lds_read_tr16((const unsigned short*)((const char*)p + OFF)));
}
template <int N> MSMC_DEV void lds_wait() {}

static inline unsigned int __float_as_uint(float f) { unsigned int u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned int u) { float f; memcpy(&f, &u, 4); return f; }
MSMC_DEV unsigned short f32_to_bf16_bits(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
MSMC_DEV float bf16_bits_to_f32(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }
MSMC_DEV unsigned int pack_bf16x2(float a, float b) {
    return (unsigned int)f32_to_bf16_bits(a) | ((unsigned int)f32_to_bf16_bits(b) << 16);
}
MSMC_DEV unsigned int bf16x2_leaky(unsigned int w, float slope) {
    const float f0 = __uint_as_float(w << 16), f1 = __uint_as_float(w & 0xffff0000u);
    const float r0 = f0 > 0.f ? f0 : f0 * slope, r1 = f1 > 0.f ? f1 : f1 * slope;
    return (unsigned int)f32_to_bf16_bits(r0) | ((unsigned int)f32_to_bf16_bits(r1) << 16);
}

static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
#define MSMC_BACKEND_NAME "emu"
// tiny by default (exercises the persistent grid-stride loops); MSMC_EMU_CUS=256 reproduces the launch
// heuristics of the real chip
static inline int msmc_emu_num_cu() { const char* e = getenv("MSMC_EMU_CUS"); return e ? atoi(e) : 3; }
#define MSMC_NUM_CU (msmc_emu_num_cu())
MSMC_DEV float fast_exp(float x) { return expf(x); }
MSMC_DEV long long msmc_clock() { return 0; }
MSMC_DEV float fast_sqrt(float x) { return sqrtf(x); }
static inline int msmc_check_launch() { return 0; }
static inline int msmc_rt_stream_create(void** out) { *out = nullptr; return 0; }      // (the interpreter has one stream)
static inline int msmc_rt_stream_destroy(void*) { return 0; }
static inline int msmc_prof_used() { return msmc_prof_log.n < MSMC_PROF_MAX ? msmc_prof_log.n : MSMC_PROF_MAX; }
static inline const char* msmc_prof_name(const char* name) {
    if (msmc_prof_log.on && msmc_prof_mine >= 0) {
        MsmcProfRec& r = msmc_prof_log.rec[msmc_prof_mine];
        int j = 0;
        for (; name[j] && j < (int)sizeof(r.name) - 1; ++j) r.name[j] = name[j];
        r.name[j] = 0;
    }
    return name;
}
static inline void msmc_prof_reset_impl() { msmc_prof_log.n = 0; msmc_prof_mine = -1; }
static inline int msmc_prof_read_impl(int i, char* name, int cap, float* ms) {
    if (i < 0 || i >= msmc_prof_used() || !name || cap <= 0 || !ms) return -1;
    const MsmcProfRec& r = msmc_prof_log.rec[i];
    int j = 0;
    for (; r.name[j] && j < cap - 1; ++j) name[j] = r.name[j];
    name[j] = 0;
    *ms = 1.0f;
    return 0;
}
static inline int msmc_allow_lds(const void*, int) { return 0; }
