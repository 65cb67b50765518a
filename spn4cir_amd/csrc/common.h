// Shared device helpers for the spn4cir_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SPN_OK 0
#define SPN_ERR_ARG (-1)
#define SPN_ERR_SHAPE (-2)
#define SPN_ERR_WORKSPACE (-3)

#define SPN_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return (int)e__;              \
    } while (0)

#include <stdlib.h>
// value of an SPN_* environment variable AS IT WAS WHEN THE LIBRARY WAS LOADED (config.hip; nullptr if unset): the only way
// the kernels' A/B switches read the environment, so that spn_config_dump() shows everything that can have changed them
const char* spn_env(const char* name);
// integer tuning knob from the environment: values < 1 or unparsable text fall back to the default (a zero-block grid
// would fail the launch or leave outputs unwritten)
static inline int env_int_min1(const char* name, int dflt) {
    const char* e = spn_env(name);
    if (!e || !*e) return dflt;
    char* end = nullptr;
    const long v = strtol(e, &end, 10);
    return (end == e || v < 1 || v > (1 << 24)) ? dflt : (int)v;
}

typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

static constexpr int WAVE = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// reduce over the 16 lanes that share (lane >> 4): xor 1,2,4,8
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float bf2f(bf16_t x) { return (float)x; }
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)x; }

// Buffer resource over [base, base+bytes): raw buffer, out-of-range reads return 0.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// 16-byte async global->LDS copy: LDS destination = lds_base (wave-uniform) + lane*16.
#ifndef SPN_GLDS_AUX
#define SPN_GLDS_AUX 0   // cache-policy bits of the DMA (1 = sc0, 2 = nt, 16 = sc1)
#endif
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, void* lds_base, uint32_t voffset_bytes) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_base), 16, voffset_bytes, 0, 0, SPN_GLDS_AUX);
}

// Write-through (sc1) stores of streaming outputs through a buffer descriptor over [base, base + 4 GB): the line is not kept in
// the XCD's L2 (MI355X_MICROARCH.md, "stores of each flavour"), so a kernel that only streams its result out neither displaces
// what the next kernel will re-read from L2 nor leaves up to 32 MB of dirty lines to be written back at the kernel boundary.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wt_rsrc(const void* base) { return make_rsrc(base, 0xfffffff0u); }
__device__ __forceinline__ void store_wt16(const __amdgpu_buffer_rsrc_t& rs, size_t byte_off, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (uint32_t)byte_off, 0, 16);
}
__device__ __forceinline__ void store_wt8(const __amdgpu_buffer_rsrc_t& rs, size_t byte_off, u32x2 v) {
    __builtin_amdgcn_raw_buffer_store_b64(v, rs, (uint32_t)byte_off, 0, 16);
}
// process-wide default of the streaming kernels' store policy (config.hip): SPN_STREAM_WT=0 keeps plain stores
int spn_stream_wt();

__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef int v2i __attribute__((ext_vector_type(2)));
// Transposed LDS reads as inline asm, for loops that keep LDS-DMA in flight: the builtin forms carry no memory operand, so
// hipcc assumes they may read what a pending buffer_load ... lds writes and puts `s_waitcnt vmcnt(0)` in front of them - once
// per tile, draining the prefetch ring (ISA of the bank kernels' dq phase).  The asm forms are invisible to the compiler's
// counters: the caller waits with wait_lgkm<N>() (N = DS instructions issued after the ones it needs; 4-bit counter) and ties
// the result registers behind the wait with lds_tie() before using them.
__device__ __forceinline__ s16x4 lds_tr16_b64_asm(const void* lds_addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)lds_addr) : "memory");
    return v;
}
__device__ __forceinline__ v2i lds_tr8_b64_asm(const void* lds_addr) {
    v2i v;
    asm volatile("ds_read_b64_tr_b8 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)lds_addr) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N < 15 ? N : 15) : "memory");
}
template <typename T>
__device__ __forceinline__ void lds_tie(T& x) {
    asm volatile("" : "+v"(x));
}

// Workgroup barrier for loops that keep LDS-DMA (buffer_load ... lds) in flight across it: __syncthreads() is a
// workgroup-scope fence, and hipcc counts a pending LDS-DMA as an LDS write of the wave - it emits `s_waitcnt vmcnt(0)` in
// front of the barrier and drains every prefetched tile (seen in the ISA of the bank kernels: each 32-row tile waited for the
// tile requested a few hundred cycles earlier).  This form waits for the wave's own ds_* traffic only; completion of the
// DMA'd tile a phase is about to read is the caller's counted `s_waitcnt vmcnt(N)` BEFORE the barrier, as in the GEMM loops.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}


// LDS transpose read: within each 16-lane group the 16 lanes address a 4x16 block of 16-bit
// elements (lane i -> row i>>2, columns (i&3)*4..+3, 8 bytes each); lane i receives column i
// (rows 0..3).  See cdna_hip_programming.md T10.
__device__ __forceinline__ s16x4 lds_tr16_b64(const void* lds_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_addr));
}

// Bijective XCD-aware remap of a 1-D block id (cdna_hip_programming.md T1): blocks that the
// dispatcher places on one XCD (id % 8) get a contiguous range of logical ids.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float quick_gelu_f(float x) { return x * fast_sigmoid(1.702f * x); }
__device__ __forceinline__ float quick_gelu_grad_f(float x) {
    const float s = fast_sigmoid(1.702f * x);
    return s * (1.0f + 1.702f * x * (1.0f - s));
}
// Standard normal cdf and density for the exact GELU (med.py BertIntermediate, timm Mlp): Phi(x) = 0.5 (1 + erf(x / sqrt 2)) with erf
// from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 - three decimal orders below a bf16 ulp of the activation), which needs
// e^{-z^2} at z = |x| / sqrt 2 - the SAME exponential as the density.  One v_exp, one v_rcp and a degree-5 Horner form instead of
// ocml's erff (~35 VALU instructions) plus the exponential: the GELU epilogue of the 256 x 256 NT tile is VALU-bound (128 outputs per
// lane; 55 us per one-round launch at 4 096 x 3 072 x 768 against 16 us of k loop).  The fp32-exact evaluation towers (exact.hip)
// keep erff.
__device__ __forceinline__ f32x2 gelu_cdf_pdf(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    const float ex = __expf(-0.5f * x * x);                                   // e^{-z^2}
    float poly = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
    poly = __builtin_fmaf(t, poly, 1.421413741f);
    poly = __builtin_fmaf(t, poly, -0.284496736f);
    poly = __builtin_fmaf(t, poly, 0.254829592f);
    const float half_erf = 0.5f - 0.5f * (poly * t) * ex;                     // erf(z) / 2, z >= 0
    return f32x2{0.5f + copysignf(half_erf, x), 0.3989422804014327f * ex};
}
// activation and its derivative from ONE sigmoid / erf evaluation (ACT: 1 = QuickGELU, 2 = exact GELU, kernels.h)
template <int ACT>
__device__ __forceinline__ f32x2 act_and_grad_pair(float x) {
    if constexpr (ACT == 1) {
        const float s = fast_sigmoid(1.702f * x);
        return f32x2{x * s, s * (1.0f + 1.702f * x * (1.0f - s))};
    } else {
        const f32x2 cp = gelu_cdf_pdf(x);
        return f32x2{x * cp[0], cp[0] + x * cp[1]};
    }
}
// y = act(x), g = act'(x); a macro because vector elements cannot bind to references
#define act_and_grad_into(ACT_, X, Y, G) do { const f32x2 ag__ = act_and_grad_pair<ACT_>(X); (Y) = ag__[0]; (G) = ag__[1]; } while (0)
__device__ __forceinline__ float gelu_erf_f(float x) { return x * gelu_cdf_pdf(x)[0]; }
__device__ __forceinline__ float gelu_erf_grad_f(float x) {
    const f32x2 cp = gelu_cdf_pdf(x);
    return cp[0] + x * cp[1];
}
