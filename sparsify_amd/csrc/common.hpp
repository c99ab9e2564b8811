// Shared host/device helpers for libsparsify_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/sparsify_hip.h"

namespace sphip {

// ---------------------------------------------------------------- errors (host)
void set_error(const char* fmt, ...);

#define SPHIP_REQUIRE(cond, code, ...)            \
    do {                                          \
        if (!(cond)) {                            \
            ::sphip::set_error(__VA_ARGS__);      \
            return (code);                        \
        }                                         \
    } while (0)

#define SPHIP_CHECK_LAUNCH(name)                                                        \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            ::sphip::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return SPHIP_ELAUNCH;                                                       \
        }                                                                               \
    } while (0)

static inline bool valid_dtype(int dt) { return dt == SPHIP_F32 || dt == SPHIP_BF16; }
static inline size_t dtype_size(int dt) { return dt == SPHIP_BF16 ? 2 : 4; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Division of a 32-bit unsigned by a launch-invariant divisor (Granlund-Montgomery):
// q = (t + ((n - t) >> sh1)) >> sh2 with t = umulhi(n, magic).
struct FastDiv {
    unsigned magic, sh1, sh2;
};
static inline FastDiv make_fastdiv(unsigned d) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    FastDiv f;
    f.magic = (unsigned)((((1ull << l) - d) << 32) / d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 0 ? l - 1 : 0;
    return f;
}

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ unsigned fastdiv(unsigned n, const FastDiv f) {
    const unsigned t = __umulhi(n, f.magic);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

typedef unsigned short bf16_t;   // raw bf16 storage

__device__ __forceinline__ float bf16_to_f32(bf16_t u) { return __uint_as_float(((unsigned)u) << 16); }

// round-to-nearest-even; NaN stays NaN (plain cast -> v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float load(const float* p, int64_t i) { return p[i]; }
    static __device__ __forceinline__ void store(float* p, int64_t i, float v) { p[i] = v; }
};
template <> struct Elem<bf16_t> {
    static __device__ __forceinline__ float load(const bf16_t* p, int64_t i) { return bf16_to_f32(p[i]); }
    static __device__ __forceinline__ void store(bf16_t* p, int64_t i, float v) { p[i] = f32_to_bf16(v); }
};

// load 4 consecutive elements as fp32 (pointer must be 4-element aligned)
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const bf16_t* p) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    float4 r;
    r.x = __uint_as_float(u.x << 16);
    r.y = __uint_as_float(u.x & 0xffff0000u);
    r.z = __uint_as_float(u.y << 16);
    r.w = __uint_as_float(u.y & 0xffff0000u);
    return r;
}

// Order-preserving unsigned image of a value's bits (16 bit for bf16, 32 bit for f32): the
// selection kernels compare keys; key 0 is reserved for "masked out" (-inf on the way back).
__device__ __forceinline__ unsigned to_key(bf16_t u) {
    const unsigned v = u;
    return (v & 0x8000u) ? ((~v) & 0xffffu) : (v | 0x8000u);
}
__device__ __forceinline__ unsigned to_key(float f) {
    const unsigned v = __float_as_uint(f);
    return (v & 0x80000000u) ? ~v : (v | 0x80000000u);
}
template <typename T> struct KeyOf;
template <> struct KeyOf<bf16_t> {
    static constexpr int BITS = 16;
    static constexpr int VEC = 8;
    static __device__ __forceinline__ bf16_t value(unsigned key) {
        if (key == 0) return (bf16_t)0xff80u;   // masked-out latent -> -inf
        return (bf16_t)((key & 0x8000u) ? (key & 0x7fffu) : ((~key) & 0xffffu));
    }
};
template <> struct KeyOf<float> {
    static constexpr int BITS = 32;
    static constexpr int VEC = 4;
    static __device__ __forceinline__ float value(unsigned key) {
        if (key == 0) return __uint_as_float(0xff800000u);
        return __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
    }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64). `red` = 16 floats of LDS.
// Result valid in every thread. Fixed order => deterministic.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// ---- schedule-free SignSGD element update (dense.hip; fused into the weight-gradient kernels)
// lerp(a, b, w) = |w| < 0.5 ? a + w (b - a) : b - (b - a)(1 - w), as torch.lerp.
__device__ __forceinline__ float lerpf(float a, float b, float w) {
    const float d = b - a;
    return fabsf(w) < 0.5f ? fmaf(w, d, a) : fmaf(-d, 1.f - w, b);
}
// x = lerp(y, z, w_undo); z -= lr sign; x = lerp(x, z, c); y = lerp(x, z, w_y)
__device__ __forceinline__ void sf_update(float& y, float& z, float sgn, float lr, float w_undo, float c, float w_y) {
    const float x = lerpf(y, z, w_undo);
    z = z - lr * sgn;
    const float x2 = lerpf(x, z, c);
    y = lerpf(x2, z, w_y);
}
// Arguments of the schedule-free form of a fused step: z == nullptr means plain SignSGD.
struct SfArgs {
    float* z;
    float w_undo, c, w_y;
};

}  // namespace sphip
