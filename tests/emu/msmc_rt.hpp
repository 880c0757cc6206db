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
uint32_t* slot(int lane_index);            // 64 x 32-byte exchange slots of the calling wave
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

#define MSMC_LAUNCH(kernel, grid, block, lds, stream, ...)                                   \
    do {                                                                                       \
        auto emu_body = [&]() { kernel(__VA_ARGS__); };                                        \
        emu::run_grid(grid, block, lds, &emu_trampoline<decltype(emu_body)>, (void*)&emu_body); \
    } while (0)

static inline void __syncthreads() { emu::block_barrier(); }

template <class T>
static inline T emu_exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 32, "slot too small");
    memcpy(emu::slot(emu::lane()), &v, sizeof(T));
    emu::wave_barrier();
    T r;
    memcpy(&r, emu::slot(src_lane & 63), sizeof(T));
    emu::wave_barrier();
    return r;
}
MSMC_DEV float wave_xor(float v, int mask) { return emu_exchange(v, emu::lane() ^ mask); }
MSMC_DEV int wave_xor(int v, int mask) { return emu_exchange(v, emu::lane() ^ mask); }
MSMC_DEV float wave_down(float v, int delta) {
    int s = emu::lane() + delta;
    return emu_exchange(v, s < 64 ? s : emu::lane());
}
MSMC_DEV float wave_bcast(float v, int lane) { return emu_exchange(v, lane); }
MSMC_DEV int wave_bcast(int v, int lane) { return emu_exchange(v, lane); }

MSMC_DEV void wave_sync() { emu::wave_barrier(); }   // fibers are not lock-step: make it a real barrier

// ---- MFMA, by the fragment layouts documented in csrc/gfx950/msmc_rt.hpp ---------------------------------
struct emu_ab32 { float a, b; };
MSMC_DEV f32x4 mfma_f32_16x16x4(float a, float b, f32x4 c) {
    emu_ab32 me = {a, b};
    memcpy(emu::slot(emu::lane()), &me, sizeof(me));
    emu::wave_barrier();
    int l = emu::lane(), col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            emu_ab32 pa, pb;
            memcpy(&pa, emu::slot(row + 16 * k), sizeof(pa));
            memcpy(&pb, emu::slot(col + 16 * k), sizeof(pb));
            acc = fmaf(pa.a, pb.b, acc);
        }
        c[r] = acc;
    }
    emu::wave_barrier();
    return c;
}
MSMC_DEV f32x16 mfma_f32_32x32x2(float a, float b, f32x16 c) {
    emu_ab32 me = {a, b};
    memcpy(emu::slot(emu::lane()), &me, sizeof(me));
    emu::wave_barrier();
    int l = emu::lane(), col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            emu_ab32 pa, pb;
            memcpy(&pa, emu::slot(row + 32 * k), sizeof(pa));
            memcpy(&pb, emu::slot(col + 32 * k), sizeof(pb));
            acc = fmaf(pa.a, pb.b, acc);
        }
        c[r] = acc;
    }
    emu::wave_barrier();
    return c;
}
static inline float emu_bf16_f32(__bf16 h) {
    unsigned short s;
    memcpy(&s, &h, 2);
    unsigned int u = ((unsigned int)s) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
struct emu_ab16 { bf16x8 a, b; };
MSMC_DEV f32x16 mfma_bf16_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
    emu_ab16 me = {a, b};
    memcpy(emu::slot(emu::lane()), &me, sizeof(me));
    emu::wave_barrier();
    int l = emu::lane(), col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int g = 0; g < 2; ++g) {
            emu_ab16 pa, pb;
            memcpy(&pa, emu::slot(row + 32 * g), sizeof(pa));
            memcpy(&pb, emu::slot(col + 32 * g), sizeof(pb));
            for (int e = 0; e < 8; ++e) acc += emu_bf16_f32(pa.a[e]) * emu_bf16_f32(pb.b[e]);
        }
        c[r] = acc;
    }
    emu::wave_barrier();
    return c;
}
MSMC_DEV f32x4 mfma_bf16_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
    emu_ab16 me = {a, b};
    memcpy(emu::slot(emu::lane()), &me, sizeof(me));
    emu::wave_barrier();
    int l = emu::lane(), col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int g = 0; g < 4; ++g) {
            emu_ab16 pa, pb;
            memcpy(&pa, emu::slot(row + 16 * g), sizeof(pa));
            memcpy(&pb, emu::slot(col + 16 * g), sizeof(pb));
            for (int e = 0; e < 8; ++e) acc += emu_bf16_f32(pa.a[e]) * emu_bf16_f32(pb.b[e]);
        }
        c[r] = acc;
    }
    emu::wave_barrier();
    return c;
}

static inline unsigned int __float_as_uint(float f) { unsigned int u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned int u) { float f; memcpy(&f, &u, 4); return f; }
MSMC_DEV unsigned short f32_to_bf16_bits(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
MSMC_DEV float bf16_bits_to_f32(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }

static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
#define MSMC_BACKEND_NAME "emu"
#define MSMC_NUM_CU 3                // tiny on purpose: exercises the persistent grid-stride loops
static inline int msmc_check_launch() { return 0; }
static inline int msmc_allow_lds(const void*, int) { return 0; }
