// common.h -- shared device/host helpers for libsggan.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <atomic>
#include <type_traits>
#include "../../include/sggan.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define SGG_WAVE 64

// 16-byte chunk <-> floats.  VEC = elements per 16-byte chunk (4 for f32, 8 for bf16).
template <typename T> struct ET;
template <> struct ET<float> {
    static constexpr int VEC = 4;
    __device__ static inline void unpack(const u32x4& c, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(c[i]);
    }
    __device__ static inline u32x4 pack(const float* f) {
        u32x4 c;
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __float_as_uint(f[i]);
        return c;
    }
};
template <> struct ET<bf16> {
    static constexpr int VEC = 8;
    __device__ static inline void unpack(const u32x4& c, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(c[i] << 16);
            f[2 * i + 1] = __uint_as_float(c[i] & 0xffff0000u);
        }
    }
    __device__ static inline u32x4 pack(const float* f) {
        u32x4 c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16 lo = (bf16)f[2 * i], hi = (bf16)f[2 * i + 1];   // v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
            c[i] = (uint32_t)__builtin_bit_cast(uint16_t, lo) | ((uint32_t)__builtin_bit_cast(uint16_t, hi) << 16);
        }
        return c;
    }
};

// Exchange between the 16-lane rows of a wave (v_permlane16_swap_b32, new on gfx950): the ODD rows of `a` change places with the
// EVEN rows of `b` -- afterwards a = {a row 0, b row 0, a row 2, b row 2}, b = {a row 1, b row 1, a row 3, b row 3}
// (tools/probes/permlane16_swap.hip).  With a 16x16 MFMA accumulator D[channel 4 * row + e][pixel column] of two pixel
// fragments in a / b this hands rows 0, 2 EIGHT consecutive channels of fragment a's pixel (a: 8 row' + 0..3, b: + 4..7) and rows
// 1, 3 those of fragment b's: 16-byte stores instead of 8-byte ones.
__device__ __forceinline__ void row_swap16(uint32_t& a, uint32_t& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
__device__ __forceinline__ void row_swap16(float& a, float& b) {
    uint32_t x = __float_as_uint(a), y = __float_as_uint(b);
    row_swap16(x, y);
    a = __uint_as_float(x); b = __uint_as_float(y);
}

__device__ inline u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ inline void st16(void* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }
// streaming forms (the `nt` cache policy): for bytes this kernel is the LAST reader of / that nobody reads soon
__device__ inline u32x4 ld16_nt(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
__device__ inline void st16_nt(void* p, const u32x4& v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); }
__device__ inline u32x4 zero16() { u32x4 z = {0u, 0u, 0u, 0u}; return z; }

__device__ inline float act_apply(float v, int act, float leak) {
    switch (act) {
        case SGG_ACT_RELU: return v > 0.f ? v : 0.f;
        case SGG_ACT_LRELU: return v > 0.f ? v : leak * v;
        case SGG_ACT_TANH: return tanhf(v);
        default: return v;
    }
}
// Epilogues: the activation code is a launch-wide constant, so it is tested ONCE and the per-element code is compiled per
// activation (act_apply's switch inside an unrolled store loop put branches -- tanhf is a call-sized body -- between every
// pair of stores and made hipcc drain vmcnt(0) after each addend load: 12 us of a 141 us GEMM).
template <int ACT> __device__ inline float act_apply_c(float v, float leak) {
    if constexpr (ACT == SGG_ACT_RELU) return v > 0.f ? v : 0.f;
    else if constexpr (ACT == SGG_ACT_LRELU) return v > 0.f ? v : leak * v;
    else if constexpr (ACT == SGG_ACT_TANH) return tanhf(v);
    else return v;
}
template <typename F> __device__ inline void act_dispatch(int act, F&& f) {   // f(std::integral_constant<int, ACT>{})
    switch (act) {
        case SGG_ACT_RELU: f(std::integral_constant<int, SGG_ACT_RELU>{}); break;
        case SGG_ACT_LRELU: f(std::integral_constant<int, SGG_ACT_LRELU>{}); break;
        case SGG_ACT_TANH: f(std::integral_constant<int, SGG_ACT_TANH>{}); break;
        default: f(std::integral_constant<int, SGG_ACT_NONE>{}); break;
    }
}
template <int ACT> __device__ inline float act_grad_c(float pre, float leak) {
    if constexpr (ACT == SGG_ACT_RELU) return pre > 0.f ? 1.f : 0.f;
    else if constexpr (ACT == SGG_ACT_LRELU) return pre > 0.f ? 1.f : leak;
    else return 1.f;
}
// derivative evaluated from the pre-activation (relu/lrelu) -- sign only
__device__ inline float act_grad_from_pre(float pre, int act, float leak) {
    switch (act) {
        case SGG_ACT_RELU: return pre > 0.f ? 1.f : 0.f;
        case SGG_ACT_LRELU: return pre > 0.f ? 1.f : leak;
        default: return 1.f;
    }
}

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// exact n/d for 0 <= n < 2^31 via (n*m) >> (31+s)   (Hacker's Delight round-up magic)
struct FastDiv {
    uint32_t m, s, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;
    f.s = s;
    f.m = (uint32_t)(((1ull << (31 + s)) / d) + 1ull);
    return f;
}
__device__ inline uint32_t fdiv(uint32_t n, const FastDiv& f) {
    return (uint32_t)(((uint64_t)n * f.m) >> (31 + f.s));
}

// LDS-DMA (global_load_lds_dwordx4: each lane's 16 bytes land at m0 + 16 * lane) issued as INLINE ASM instead of
// __builtin_amdgcn_global_load_lds.  hipcc's waitcnt pass books the builtin as a FLAT operation that touches both memory and
// LDS, and while one of those is "pending" -- in a software-pipelined loop: always -- it turns every s_waitcnt it inserts into a
// full drain: lgkmcnt(0) in front of the first MFMA that consumes an LDS fragment (so fragment reads issued ahead never
// overlap the MFMAs) and vmcnt(0) in front of every use of a register load.  The asm form is invisible to that pass; ordering
// against the LDS reads is the kernels' own s_waitcnt vmcnt(0) + s_barrier, which they had anyway.
__device__ inline void dma16_to_lds(const void* gsrc, __attribute__((address_space(3))) void* lds_wave_base) {
    const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(m) : "memory", "m0");
}

// ... with the streaming (nt) cache policy: operand bytes nobody reads again soon
__device__ inline void dma16_to_lds_nt(const void* gsrc, __attribute__((address_space(3))) void* lds_wave_base) {
    const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(gsrc), "s"(m) : "memory", "m0");
}

// the same with the address as a UNIFORM 64-bit base (an SGPR pair) + a 32-bit per-lane byte offset: no 64-bit VGPR address
// per instruction (the bases of the pieces of one tile differ by scalars -- SALU adds instead of per-lane 64-bit VALU adds,
// and nothing for hipcc to hoist into registers across the main loop)
__device__ inline void dma16_to_lds_s(const void* sbase, uint32_t voff, __attribute__((address_space(3))) void* lds_wave_base) {
    const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_wave_base);
    // (readfirstlane: a no-op for a value hipcc already holds in SGPRs; where its divergence analysis is unsure the "s"
    // constraint alone is handed a VGPR pair, which does not assemble)
    const uint64_t b = (uint64_t)(uintptr_t)sbase;
    // (the builtin returns a SIGNED int: without the uint32_t casts a low half >= 0x80000000 sign-extends over the high half)
    const uint64_t sb = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b) |
                        ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sb), "s"(m) : "memory", "m0");
}

// ... and through a buffer descriptor: `buffer_load_dwordx4 voff, rsrc, soff offen lds`.  The address is base + soff + voff
// with base / size in four SGPRs for the whole kernel, so a piece costs NO vector instruction -- the scalar part of the
// address goes in soff -- and lanes whose voff is outside the buffer write ZEROS to LDS (checked on gfx950 with
// tools/probes/bufload_lds.hip): zero padding needs no select against a zero page and no per-piece lane mask.  The range
// check is documented for voff (+ the immediate); callers keep voff alone either inside the buffer or >= SGG_BUF_OOB with
// the buffer smaller than that, so the result is the same whether or not the hardware adds soff before checking.
typedef int sgg_rsrc_t __attribute__((ext_vector_type(4)));
#define SGG_BUF_OOB 0x40000000u
__device__ inline sgg_rsrc_t sgg_make_rsrc(const void* base, uint32_t bytes) {
    const uint64_t b = (uint64_t)(uintptr_t)base;
    sgg_rsrc_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(b >> 32) & 0xffffu));     // stride 0, no swizzle
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;                                                              // raw 32-bit data format
    return r;
}
__device__ inline void dma16_buf_to_lds(uint32_t voff, sgg_rsrc_t rsrc, uint32_t soff, __attribute__((address_space(3))) void* lds_wave_base) {
    const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_wave_base);
    const uint32_t so = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(so), "s"(m) : "memory", "m0");
}

__device__ inline void dma4_to_lds(const void* gsrc, __attribute__((address_space(3))) void* lds_wave_base) {   // 4 bytes per lane at m0 + 4 * lane
    const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gsrc), "s"(m) : "memory", "m0");
}
// ... and the waits as BUILTINS (s_waitcnt simm16, gfx9 encoding: vmcnt[3:0] | expcnt << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14): unlike an
// asm string they are seen by the waitcnt pass, which then knows what is outstanding and counts its own waits from there.
template <int N> __device__ inline void sgg_wait_vm() {                 // until at most N vector-memory operations are outstanding
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70);
    asm volatile("" ::: "memory");
}
#define SGG_WAIT_VM0() sgg_wait_vm<0>()
#define SGG_WAIT_LGKM0() do { __builtin_amdgcn_s_waitcnt(0xC07F); asm volatile("" ::: "memory"); } while (0)

static inline int sgg_check_launch() { return hipGetLastError() == hipSuccess ? SGG_OK : SGG_ELAUNCH; }

// Kernels that need more than 64 KB of dynamic LDS must have the attribute raised once PER DEVICE (a process may drive
// several GPUs, from several host threads).  `done` holds one bit per device ordinal; setting the attribute twice is
// harmless, so a relaxed race between two threads costs one redundant call, never a missing one.
static inline int sgg_lds_attr(const void* kern, int bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return SGG_ELAUNCH;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return SGG_OK;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) { (void)hipGetLastError(); return SGG_ELAUNCH; }
    done.fetch_or(bit, std::memory_order_release);
    return SGG_OK;
}
#define SGG_LDS_ATTR(kern, bytes)                                                        \
    do {                                                                                 \
        static std::atomic<uint64_t> sgg_attr_done_{0};                                  \
        int sgg_attr_rc_ = sgg_lds_attr((const void*)(kern), (int)(bytes), sgg_attr_done_); \
        if (sgg_attr_rc_) return sgg_attr_rc_;                                           \
    } while (0)

// Measurement hook (sgg_time_next_launch, include/sggan.h): a host thread may arm ONE pair of HIP events; the next launch that
// goes through sgg_launch_timed -- the main kernel of a timed family: the halo GEMMs, the all-taps weight gradient, the
// instance-norm apply pass -- carries them ON ITS OWN DISPATCH PACKET (hipExtLaunchKernel), so their timestamps are the
// kernel's begin and end as the command processor records them, the same clock rocprofv3's kernel trace reads.
struct SggTimedLaunch { hipEvent_t start = nullptr, stop = nullptr; int consumed = 0; };
SggTimedLaunch& sgg_timed_launch();                     // thread-local (misc.hip)
template <typename F, typename... Args>
static inline void sgg_launch_timed(F kern, dim3 grid, dim3 block, unsigned lds, hipStream_t s, Args... args) {
    SggTimedLaunch& t = sgg_timed_launch();
    if (t.start && !t.consumed) {
        hipExtLaunchKernelGGL(kern, grid, block, lds, s, t.start, t.stop, 0, args...);
        t.consumed = 1;
    } else {
        hipLaunchKernelGGL(kern, grid, block, lds, s, args...);
    }
}

// compute units of the CURRENT device (persistent kernels size their grids by it); cached per device ordinal, 256 if the
// query fails (MI355X)
static inline int sgg_num_cus() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v <= 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cache[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

// kernel-selection switches (defaults = the shipped configuration); see sgg_config() in conv.hip
struct SggConfig {
    int halo3 = 1, s2halo = 1, w9 = 1, w9s2 = 1, stem_dgrad_halo = 1, wgrad_rowfast = 1, glds = 1, n7 = 1;
    int in_fused_maxhw = -1;      // < 0: IN_FUSED_MAXHW of norm.hip
    int ablate = 0;
};
const SggConfig& sgg_config();
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
