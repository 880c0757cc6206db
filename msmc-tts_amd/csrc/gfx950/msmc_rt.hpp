// msmc_rt.hpp -- gfx950 device runtime vocabulary used by every kernel in csrc/.
//
// Thin, zero-cost names for the CDNA4 intrinsics the kernels rely on (64-wide wavefront
// cross-lane moves, f32 and bf16 MFMA) plus the launch macro.  Kernels include <msmc_rt.hpp> (found via -I csrc/gfx950) and never
// touch the builtins directly; tests/emu/msmc_rt.hpp provides the same vocabulary for the CPU
// interpreter that the -m "not gpu" tests use to check kernel logic (test infrastructure only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define MSMC_WAVE 64
#define MSMC_DEV static __device__ __forceinline__
#define MSMC_DEV_INLINE __device__ __forceinline__
#define MSMC_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) char name[]
typedef hipStream_t msmc_stream_t;

// ---- per-launch profiling log (msmc_prof_* of include/msmc_hip.h) ------------------------------------------
// Off by default: one thread-local flag test per launch.  On: every launch of the calling thread is bracketed by a HIP
// event pair ON THE STREAM IT IS LAUNCHED ON and logged under the kernel's symbol -- the launch macro supplies the
// template's name, launchers that know the instantiation (msmc_conv_gather / _wgrad ...) overwrite it through
// msmc_prof_name with the text rocprofv3 prints.  bench.py reads the log after a device synchronisation.
struct MsmcProfRec {
    char name[120];
    hipEvent_t e0, e1;
};
#define MSMC_PROF_MAX 16384
// ONE log per process: a training step launches from two threads (the forward pass from the caller's, the backward pass
// from the autograd engine's), one after the other; slots are claimed with an atomic counter.
struct MsmcProfLog {
    volatile int on = 0;
    int n = 0;                       // claimed slots (atomic increments)
    MsmcProfRec* rec = nullptr;
};
inline MsmcProfLog msmc_prof_log;
inline thread_local int msmc_prof_mine = -1;          // slot of the calling thread's most recent launch
static inline void msmc_prof_pre(hipStream_t st) {
    MsmcProfLog& L = msmc_prof_log;
    const int i = __atomic_fetch_add(&L.n, 1, __ATOMIC_RELAXED);
    msmc_prof_mine = i < MSMC_PROF_MAX ? i : -1;
    if (msmc_prof_mine < 0) return;
    MsmcProfRec& r = L.rec[i];
    r.name[0] = 0;
    hipEventCreate(&r.e0);
    hipEventCreate(&r.e1);
    hipEventRecord(r.e0, st);
}
static inline void msmc_prof_post(hipStream_t st, const char* text) {
    if (msmc_prof_mine < 0) return;
    MsmcProfRec& r = msmc_prof_log.rec[msmc_prof_mine];
    hipEventRecord(r.e1, st);
    int j = 0;                                        // "(kernel<T, 4>)" -> "kernel"
    for (const char* c = text; *c && *c != '<' && j < (int)sizeof(r.name) - 1; ++c)
        if (*c != '(' && *c != ' ') r.name[j++] = *c;
    r.name[j] = 0;
}
// the launcher knows the instantiation of the launch it just issued
static inline const char* msmc_prof_name(const char* name) {
    if (msmc_prof_log.on && msmc_prof_mine >= 0) {
        MsmcProfRec& r = msmc_prof_log.rec[msmc_prof_mine];
        int j = 0;
        for (; name[j] && j < (int)sizeof(r.name) - 1; ++j) r.name[j] = name[j];
        r.name[j] = 0;
    }
    return name;
}
static inline int msmc_prof_used() { return msmc_prof_log.n < MSMC_PROF_MAX ? msmc_prof_log.n : MSMC_PROF_MAX; }
static inline void msmc_prof_reset_impl() {
    MsmcProfLog& L = msmc_prof_log;
    for (int i = 0; i < msmc_prof_used(); ++i) {
        hipEventDestroy(L.rec[i].e0);
        hipEventDestroy(L.rec[i].e1);
    }
    L.n = 0;
    msmc_prof_mine = -1;
}
static inline int msmc_prof_read_impl(int i, char* name, int cap, float* ms) {
    MsmcProfLog& L = msmc_prof_log;
    if (i < 0 || i >= msmc_prof_used() || !name || cap <= 0 || !ms) return -1;
    MsmcProfRec& r = L.rec[i];
    if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(ms, r.e0, r.e1) != hipSuccess) return -2;
    int j = 0;
    for (; r.name[j] && j < cap - 1; ++j) name[j] = r.name[j];
    name[j] = 0;
    return 0;
}
#define MSMC_LAUNCH(kernel, grid, block, lds, stream, ...)                    \
    do {                                                                      \
        if (msmc_prof_log.on) msmc_prof_pre(stream);                          \
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);    \
        if (msmc_prof_log.on) msmc_prof_post(stream, #kernel);                \
    } while (0)

// ---- 64-lane cross-lane moves ------------------------------------------------------------
// value known to be identical in every lane of the wave -> scalar register
MSMC_DEV int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
MSMC_DEV float wave_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
MSMC_DEV int wave_xor(int v, int mask) { return __shfl_xor(v, mask, 64); }
MSMC_DEV float wave_down(float v, int delta) { return __shfl_down(v, delta, 64); }
MSMC_DEV float wave_bcast(float v, int lane) { return __shfl(v, lane, 64); }
MSMC_DEV int wave_bcast(int v, int lane) { return __shfl(v, lane, 64); }
// every lane reads the value of the lane IT names (ds_bpermute_b32)
MSMC_DEV float wave_bcast_var(float v, int src_lane) { return __shfl(v, src_lane, 64); }

// true in every lane when the predicate holds in at least one lane of the wave (wave-uniform branch conditions)
MSMC_DEV bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// median of three (v_med3_f32): with lo <= hi, med3(hi, lo, x) is the second largest of {hi, lo, x}
MSMC_DEV float fmed3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
// max(a, b) of two non-NaN values in ONE operation: fmaxf would be preceded by a canonicalising v_max_f32 v, v, v per
// operand whose origin the compiler cannot see (bit patterns assembled with integer operations)
MSMC_DEV float fmax_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (bits(a) & keep) | ins in ONE operation (v_and_or_b32): plants the wave-uniform value `ins` in the low mantissa bits
MSMC_DEV float bits_and_or(float a, unsigned int keep, unsigned int ins) {
    unsigned int r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(keep), "s"(ins));
    return __uint_as_float(r);
}
// two floats -> packed bf16 pair (round to nearest even, v_cvt_pk_bf16_f32): a in bits 0..15, b in bits 16..31
MSMC_DEV unsigned int pack_bf16x2(float a, float b) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ r = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(r, bf16x2_));
}

// value of lane l ^ 16 / l ^ 32 by the gfx950 row / half swaps (VALU: no LDS round trip, unlike ds_bpermute_b32).
// v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of its second,
// v_permlane32_swap the upper half of the first with the lower half of the second (verified on MI355X by
// tests/test_gpu_parity.py::test_wave_exchange_primitives).
MSMC_DEV unsigned int wave_xor16_u(unsigned int v) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) & 16u) ? r[0] : r[1];
}
MSMC_DEV unsigned int wave_xor32_u(unsigned int v) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) & 32u) ? r[0] : r[1];
}
// a of the upper half <-> b of the lower half (one v_permlane32_swap): lane l < 32 ends with (its a, lane l+32's a),
// lane l + 32 with (lane l's b, its b)
// value of lane `src_lane` (wave-uniform index) as a scalar: v_readlane_b32, no LDS round trip
MSMC_DEV int wave_read_lane(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }
MSMC_DEV unsigned int wave_read_lane(unsigned int v, int src_lane) { return (unsigned int)__builtin_amdgcn_readlane((int)v, src_lane); }
MSMC_DEV void wave_swap32(unsigned int& a, unsigned int& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
MSMC_DEV float wave_xor16(float v) { return __uint_as_float(wave_xor16_u(__float_as_uint(v))); }
MSMC_DEV float wave_xor32(float v) { return __uint_as_float(wave_xor32_u(__float_as_uint(v))); }
MSMC_DEV int wave_xor16(int v) { return (int)wave_xor16_u((unsigned int)v); }
MSMC_DEV int wave_xor32(int v) { return (int)wave_xor32_u((unsigned int)v); }

// sum over the 64 lanes, every lane ends with the total: four DPP steps inside each row of 16 (quad permutes, half-row and
// row mirrors -- one VALU instruction each, the permute rides on the add), then the row and half swaps
template <int CTRL>
MSMC_DEV float wave_dpp_t(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
MSMC_DEV float wave_sum(float v) {
    v = v + wave_dpp_t<0xB1>(v);                       // quad_perm [1,0,3,2]
    v = v + wave_dpp_t<0x4E>(v);                       // quad_perm [2,3,0,1]
    v = v + wave_dpp_t<0x141>(v);                      // row_half_mirror
    v = v + wave_dpp_t<0x140>(v);                      // row_mirror
    v = v + wave_xor16(v);
    v = v + wave_xor32(v);
    return v;
}

// Intra-wave LDS hand-off point: lanes of ONE wave exchange data through LDS (a wave executes its
// LDS instructions in order, so no hardware barrier is needed); this only stops the compiler from
// moving LDS accesses across the hand-off.
MSMC_DEV void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- matrix cores --------------------------------------------------------------------------
// f32-in / f32-accumulate: exact fp32, bit-identical to a k-ordered fmaf chain.
//   16x16x4 : A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D reg r -> row 4*(l>>4)+r, col l&15
//   32x32x2 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D reg r -> row (r&3)+8*(r>>2)+4*(l>>5), col l&31
MSMC_DEV f32x4 mfma_f32_16x16x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
MSMC_DEV f32x16 mfma_f32_32x32x2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// bf16 in / f32 accumulate:
//   32x32x16: A[i=l&31][k=8*(l>>5)+e], B[k=8*(l>>5)+e][j=l&31], e=0..7, D as 32x32x2
//   16x16x32: A[i=l&15][k=8*(l>>4)+e], B[k=8*(l>>4)+e][j=l&15],          D as 16x16x4
MSMC_DEV f32x16 mfma_bf16_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
MSMC_DEV f32x4 mfma_bf16_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ---- LDS transpose read (ds_read_b64_tr_b16) ------------------------------------------------------
// Every lane passes the address of 4 contiguous 16-bit elements (8-byte aligned); within each group of
// 16 lanes the 16 x 4 elements are transposed: lane l (L = l & 15) receives, for j = 0..3, element
// (L & 3) of the chunk addressed by lane 16*(l >> 4) + 4*j + (L >> 2).  With lane Ls addressing row
// (Ls >> 2), columns 4*(Ls & 3).. of a [4][16] block, lane L gets column L of that block: the
// K-contiguous MFMA fragment of a row-major [k][n] LDS tile.  (Semantics verified on MI355X with
// tests/probes/tr_probe.hip.)
MSMC_DEV u16x4 lds_read_tr16(const unsigned short* p) {
    typedef short s16x4_ __attribute__((ext_vector_type(4)));
    s16x4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)p);
    return __builtin_bit_cast(u16x4, v);
}

// ---- LDS-DMA (global_load_lds_dwordx4): 16 bytes per lane from a per-lane global address straight into LDS at
// wave-uniform base + 16 * lane (the image is lane-linear: swizzle the SOURCE, never the destination); no registers.
// lds_dma_wait() retires every outstanding piece of the calling wave; a barrier must follow before other waves read.
MSMC_DEV void lds_dma16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
MSMC_DEV void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// at most N younger pieces (or other vector-memory loads) of the calling wave still in flight: pieces land in issue order
template <int N>
MSMC_DEV void lds_dma_wait_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope release fence, which makes
// hipcc retire EVERY outstanding vector-memory operation (s_waitcnt vmcnt(0)) -- including LDS-DMA pieces of later ring
// stages that are meant to stay in flight across the barrier.  The caller retires the pieces it needs first
// (lds_dma_wait_n) and must not rely on this barrier for the visibility of global-memory writes.
MSMC_DEV void block_sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- hand-counted global loads beside an LDS-DMA ring: an ordinary load inside such a loop makes hipcc drain the whole
// vector-memory queue (s_waitcnt vmcnt(0)) BEFORE issuing it, i.e. right after the ring's next stage was requested.
// These 8-byte loads are invisible to its wait insertion: the caller retires them with lds_dma_wait() (vmcnt(0)) and then
// passes every destination through vm_pin() before the first use (volatile asm statements keep their order, so the uses
// cannot be scheduled above the wait).
MSMC_DEV u32x2 global_load8_async(const void* p) {
    u32x2 v;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p));
    return v;
}
MSMC_DEV void vm_pin(u32x2& v) { asm volatile("" : "+v"(v)); }

// The machine scheduler must not move instructions across this point (software pipelines written in source order:
// hipcc otherwise sinks prefetching LDS reads down to their first use and waits with lgkmcnt(0)).
MSMC_DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// ---- hand-counted LDS reads: hipcc waits with lgkmcnt(0) at the first use of an LDS read issued before a loop
// back edge, which exposes one LDS round trip per iteration of a software-pipelined fragment loop.  These reads are
// invisible to the compiler's wait insertion; the CALLER retires them with lds_wait<N>() (= at most N younger LDS
// operations of this wave still outstanding; LDS operations return in order) before the first use, and must drain
// (lds_wait<0>) before the destination registers die.  cdna_hip_programming.md 5.7, form (iii).
MSMC_DEV u16x8 lds_read128_async(const void* p) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((__attribute__((address_space(3))) const char*)p));
    return __builtin_bit_cast(u16x8, v);
}
// the same read as four dwords: a register quadruple that goes to an MFMA operand (or accumulator) as it is -- a 16-bit
// element vector would be unpacked and re-packed by the compiler right after the read, i.e. BEFORE the caller's wait
MSMC_DEV u32x4 lds_read128_async4(const void* p) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((__attribute__((address_space(3))) const char*)p));
    return v;
}
// ... at p + OFF bytes (immediate offset field of the instruction, OFF < 65536: no address arithmetic per read)
template <int OFF>
MSMC_DEV u32x4 lds_read128_async4_off(const void* p) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((__attribute__((address_space(3))) const char*)p), "n"(OFF));
    return v;
}
MSMC_DEV int lds_read32_async(const void* p) {
    int v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"((__attribute__((address_space(3))) const char*)p));
    return v;
}
// transposing read (see lds_read_tr16) at p + OFF bytes, hand-counted like the two above
template <int OFF>
MSMC_DEV u32x2 lds_read_tr16_async(const void* p) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2"
                 : "=v"(v)
                 : "v"((__attribute__((address_space(3))) const char*)p), "n"(OFF));
    return v;
}
template <int N>
MSMC_DEV void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// hardware exponential (v_exp_f32 after a multiply by log2 e): ~1e-6 relative, exp(-inf) = 0
MSMC_DEV float fast_exp(float x) { return __expf(x); }
// shader clock (s_memtime): diagnostics only
MSMC_DEV long long msmc_clock() { return (long long)__builtin_amdgcn_s_memtime(); }
// hardware square root (v_sqrt_f32, ~1 ulp): for bounds and estimates, never for a reference-defined value
MSMC_DEV float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// ---- bf16 <-> f32 (round to nearest even), bit-level so host and device agree ---------------
// (round 4: v_cvt_pk_bf16_f32 -- the same round-to-nearest-even bits for every finite value in ONE vector instruction; the
//  bit-level form it replaces -- NaN test, rounding add, shift: ~6 instructions per value -- made the epilogues of the thin-layer
//  kernels a visible share of their vector-ALU time.  Epilogues that convert value PAIRS call pack_bf16x2 directly.)
MSMC_DEV unsigned short f32_to_bf16_bits(float f) { return (unsigned short)pack_bf16x2(f, f); }
MSMC_DEV float bf16_bits_to_f32(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }

// leaky ReLU of two packed bf16 values, slope in [0, 1]: max(x, slope * x) in fp32 (x for x > 0, slope * x otherwise),
// rounded back to bf16 by v_cvt_pk_bf16_f32 (round to nearest even, as f32_to_bf16_bits)
MSMC_DEV unsigned int bf16x2_leaky(unsigned int w, float slope) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const float f0 = __uint_as_float(w << 16), f1 = __uint_as_float(w & 0xffff0000u);
    const f32x2_ r = {fmaxf(f0, f0 * slope), fmaxf(f1, f1 * slope)};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(r, bf16x2_));
}

// ---- host-side helpers used by the C-ABI launchers ------------------------------------------
#define MSMC_BACKEND_NAME "gfx950"
#define MSMC_NUM_CU 256              // MI355X: 8 XCDs x 32 CUs
static inline int msmc_check_launch() { return (int)hipGetLastError(); }
// a stream of the library's own (outside the host framework's pool): non-blocking with respect to the null stream
static inline int msmc_rt_stream_create(void** out) {
    hipStream_t st = nullptr;
    const hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    *out = (void*)st;
    return (int)e;
}
static inline int msmc_rt_stream_destroy(void* st) { return (int)hipStreamDestroy((hipStream_t)st); }
// Kernels that carve more than 64 KiB of dynamic LDS must opt in once per function.
static inline int msmc_allow_lds(const void* fn, int bytes) {
    if (bytes <= 64 * 1024) return 0;
    return (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
