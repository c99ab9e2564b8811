// Dense, HBM-bound parameter-side kernels: encoder-input centring / casts, decoder row
// normalisation, projected gradient, SignSGD, and the dead-latent counters.
// Reference: sparsify/sparse_coder.py:191-192 (x - b_dec), :205-209 (autocast casts),
// :273-279 (set_decoder_norm_to_unit_norm), :281-295
// (remove_gradient_parallel_to_decoder_directions); sparsify/sign_sgd.py:15-23;
// sparsify/trainer.py:415, :422, :531-538 (num_tokens_since_fired bookkeeping).
//
// gfx950 design: one 256-thread workgroup per decoder row for the row-wise ops (the 16 KB
// row is read once from HBM; the second sweep hits L1/L2), flat grid-stride float4 kernels
// for the element-wise ones. The fused decoder update reads (W, G) once and writes W once
// instead of the reference's einsum / sub / sign / add passes.
#include "common.hpp"
#include <math.h>

namespace sphip {

template <typename T>
__global__ __launch_bounds__(256) void center_cast_kernel(const T* __restrict__ x, const float* __restrict__ b,
                                                          float* __restrict__ of, bf16_t* __restrict__ ob,
                                                          int64_t n, int D) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float v = Elem<T>::load(x, i);
        if (b != nullptr) v -= b[i % D];
        if (of != nullptr) of[i] = v;
        if (ob != nullptr) ob[i] = f32_to_bf16(v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void center_cast4_kernel(const T* __restrict__ x, const float* __restrict__ b,
                                                           float* __restrict__ of, bf16_t* __restrict__ ob,
                                                           int64_t n4, int D4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 v = load4(x + i * 4);
        if (b != nullptr) {
            const float4 bb = *reinterpret_cast<const float4*>(b + (i % D4) * 4);
            v.x -= bb.x; v.y -= bb.y; v.z -= bb.z; v.w -= bb.w;
        }
        if (of != nullptr) *reinterpret_cast<float4*>(of + i * 4) = v;
        if (ob != nullptr) {
            const unsigned lo = (unsigned)f32_to_bf16(v.x) | ((unsigned)f32_to_bf16(v.y) << 16);
            const unsigned hi = (unsigned)f32_to_bf16(v.z) | ((unsigned)f32_to_bf16(v.w) << 16);
            *reinterpret_cast<uint2*>(ob + i * 4) = make_uint2(lo, hi);
        }
    }
}

// W[j,:] /= ||W[j,:]|| + eps
template <int VEC>
__global__ __launch_bounds__(256) void unit_norm_kernel(float* __restrict__ W, int D, float eps) {
    __shared__ float red[16];
    float* row = W + (int64_t)blockIdx.x * D;
    float acc = 0.f;
    for (int c = threadIdx.x * VEC; c < D; c += 256 * VEC) {
        if (VEC == 4) {
            const float4 v = *reinterpret_cast<const float4*>(row + c);
            acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
        } else {
            const float v = row[c];
            acc = fmaf(v, v, acc);
        }
    }
    const float denom = sqrtf(block_sum(acc, red)) + eps;
    for (int c = threadIdx.x * VEC; c < D; c += 256 * VEC) {
        if (VEC == 4) {
            float4 v = *reinterpret_cast<const float4*>(row + c);
            v.x /= denom; v.y /= denom; v.z /= denom; v.w /= denom;
            *reinterpret_cast<float4*>(row + c) = v;
        } else {
            row[c] = row[c] / denom;
        }
    }
}

// MODE 0: G -= <G,W> W (in place on G).   MODE 1: W -= lr * sign(G - <G,W> W) (G untouched).
template <int VEC, int MODE>
__global__ __launch_bounds__(256) void decoder_row_kernel(float* __restrict__ W, float* __restrict__ G, int D,
                                                          float lr) {
    __shared__ float red[16];
    float* w = W + (int64_t)blockIdx.x * D;
    float* g = G + (int64_t)blockIdx.x * D;
    float acc = 0.f;
    for (int c = threadIdx.x * VEC; c < D; c += 256 * VEC) {
        if (VEC == 4) {
            const float4 a = *reinterpret_cast<const float4*>(g + c);
            const float4 b = *reinterpret_cast<const float4*>(w + c);
            acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
        } else {
            acc = fmaf(g[c], w[c], acc);
        }
    }
    const float par = block_sum(acc, red);
    for (int c = threadIdx.x * VEC; c < D; c += 256 * VEC) {
        if (VEC == 4) {
            const float4 a = *reinterpret_cast<const float4*>(g + c);
            float4 b = *reinterpret_cast<const float4*>(w + c);
            float4 p = make_float4(a.x - par * b.x, a.y - par * b.y, a.z - par * b.z, a.w - par * b.w);
            if (MODE == 0) {
                *reinterpret_cast<float4*>(g + c) = p;
            } else {
                b.x -= lr * (float)((p.x > 0.f) - (p.x < 0.f));
                b.y -= lr * (float)((p.y > 0.f) - (p.y < 0.f));
                b.z -= lr * (float)((p.z > 0.f) - (p.z < 0.f));
                b.w -= lr * (float)((p.w > 0.f) - (p.w < 0.f));
                *reinterpret_cast<float4*>(w + c) = b;
            }
        } else {
            const float p = g[c] - par * w[c];
            if (MODE == 0) g[c] = p;
            else w[c] = w[c] - lr * (float)((p > 0.f) - (p < 0.f));
        }
    }
}


// Row-resident forms (D <= 1024 * R, D % 4 == 0): the row (and its gradient) is loaded once
// into registers, R float4 per thread, so every operand crosses HBM exactly once.
// OP 0: W /= ||W|| + eps.   OP 1: W -= lr * sign(G - <G,W> W).   OP 2: OP 1 then OP 0 on the
// updated row (optimizer step + the re-normalisation that opens the next train step,
// trainer.py:408). The reductions add in the same order as the two-sweep kernels above, so
// results are bitwise identical to them. OP 3: W -= lr * sign(G), the plain SignSGD step (a
// transcoder's decoder, which is neither projected nor normalised): signsgd_kernel's arithmetic,
// here for the sake of SM.
// SM (needs D % 32 == 0): the finished row also goes to the slice-major copy [D/32][M][32] the
// L2-resident decode gathers from (sphip_slice_major_f32), while it is in registers: the separate
// copy pass (2 GiB read + 2 GiB written at the headline shape) becomes 2 GiB of extra stores.
template <int R, int OP, bool SM = false>
__global__ __launch_bounds__(256) void decoder_row_resident_kernel(float* __restrict__ W, const float* __restrict__ G,
                                                                   int D, float lr, float eps,
                                                                   float* __restrict__ sm = nullptr, int64_t M = 0) {
    __shared__ float red[16];
    // SPHIP_DEC_REMAP: workgroups are dealt to the XCDs round-robin, so rows j .. j+7 would run on eight
    // different XCDs; with the remap one XCD's consecutive workgroups take eight adjacent rows, whose
    // 128-byte pieces of a slice-major slice are 1 KiB contiguous behind one L2
#ifndef SPHIP_DEC_REMAP
#define SPHIP_DEC_REMAP 1
#endif
    int64_t row = blockIdx.x;
    if (SPHIP_DEC_REMAP && SM && (gridDim.x & 63) == 0) {
        const unsigned xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        row = (int64_t)(local >> 3) * 64 + xcd * 8 + (local & 7);
    }
    float* w = W + row * D;
    const float* g = G + row * D;
    float4 b[R], a[R];
    // every stream crosses the kernel once (two 2 GiB matrices in, one or two out): non-temporal
    // accesses, bits 1 / 2 / 4 / 8 = W load, G load, W store, slice-major store. A/B on one box
    // (make EXTRA=-DSPHIP_DEC_NT=n): 0 -> 1.649 ms, 2 -> 1.667, 6 -> 1.668, 14 -> 1.615, 15 -> 1.576 ms
    // with the slice-major copy; 1.176 -> 1.050 ms without it
#ifndef SPHIP_DEC_NT
#define SPHIP_DEC_NT 15
#endif
    typedef float f4 __attribute__((ext_vector_type(4)));
    auto ld = [](const float* p, bool nt) {
        if (nt) {
            const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
            return make_float4(t[0], t[1], t[2], t[3]);
        }
        return *reinterpret_cast<const float4*>(p);
    };
    auto st = [](float* p, float4 v, bool nt) {
        if (nt) {
            const f4 t = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(t, reinterpret_cast<f4*>(p));
        } else {
            *reinterpret_cast<float4*>(p) = v;
        }
    };
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int c = (threadIdx.x + 256 * i) * 4;
        b[i] = c < D ? ld(w + c, (SPHIP_DEC_NT & 1) != 0) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (OP != 0) a[i] = c < D ? ld(g + c, (SPHIP_DEC_NT & 2) != 0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (OP == 3) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            b[i].x -= lr * (float)((a[i].x > 0.f) - (a[i].x < 0.f));
            b[i].y -= lr * (float)((a[i].y > 0.f) - (a[i].y < 0.f));
            b[i].z -= lr * (float)((a[i].z > 0.f) - (a[i].z < 0.f));
            b[i].w -= lr * (float)((a[i].w > 0.f) - (a[i].w < 0.f));
        }
    } else if (OP != 0) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            acc = fmaf(a[i].x, b[i].x, acc); acc = fmaf(a[i].y, b[i].y, acc);
            acc = fmaf(a[i].z, b[i].z, acc); acc = fmaf(a[i].w, b[i].w, acc);
        }
        const float par = block_sum(acc, red);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const float4 p = make_float4(a[i].x - par * b[i].x, a[i].y - par * b[i].y, a[i].z - par * b[i].z,
                                         a[i].w - par * b[i].w);
            b[i].x -= lr * (float)((p.x > 0.f) - (p.x < 0.f));
            b[i].y -= lr * (float)((p.y > 0.f) - (p.y < 0.f));
            b[i].z -= lr * (float)((p.z > 0.f) - (p.z < 0.f));
            b[i].w -= lr * (float)((p.w > 0.f) - (p.w < 0.f));
        }
    }
    if (OP == 0 || OP == 2) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int c = (threadIdx.x + 256 * i) * 4;
            if (c < D) {
                acc = fmaf(b[i].x, b[i].x, acc); acc = fmaf(b[i].y, b[i].y, acc);
                acc = fmaf(b[i].z, b[i].z, acc); acc = fmaf(b[i].w, b[i].w, acc);
            }
        }
        const float denom = sqrtf(block_sum(acc, red)) + eps;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            b[i].x /= denom; b[i].y /= denom; b[i].z /= denom; b[i].w /= denom;
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int c4 = threadIdx.x + 256 * i;               // 16-byte chunk of the row
        const int c = c4 * 4;
        if (c < D) {
            st(w + c, b[i], (SPHIP_DEC_NT & 4) != 0);
            // slice c4 / 8, piece of row blockIdx.x, chunk c4 % 8: eight threads write one 128-byte line
            if (SM) st(sm + (((int64_t)(c4 >> 3) * M + row) * 8 + (c4 & 7)) * 4, b[i], (SPHIP_DEC_NT & 8) != 0);
        }
    }
}

template <int OP>
static bool launch_row_resident(float* W, const float* G, int64_t M, int64_t D, float lr, float eps, hipStream_t s,
                                float* sm = nullptr) {
    if (D % 4 != 0 || D > 8192 || ((uintptr_t)W & 15) != 0 || ((uintptr_t)G & 15) != 0) return false;
    dim3 grid((unsigned)M), block(256);
    if (sm != nullptr) {
        if (D % 32 != 0 || ((uintptr_t)sm & 15) != 0) return false;
        if (D <= 1024) hipLaunchKernelGGL((decoder_row_resident_kernel<1, OP, true>), grid, block, 0, s, W, G, (int)D, lr, eps, sm, M);
        else if (D <= 2048) hipLaunchKernelGGL((decoder_row_resident_kernel<2, OP, true>), grid, block, 0, s, W, G, (int)D, lr, eps, sm, M);
        else if (D <= 4096) hipLaunchKernelGGL((decoder_row_resident_kernel<4, OP, true>), grid, block, 0, s, W, G, (int)D, lr, eps, sm, M);
        else hipLaunchKernelGGL((decoder_row_resident_kernel<8, OP, true>), grid, block, 0, s, W, G, (int)D, lr, eps, sm, M);
        return true;
    }
    if (D <= 1024) hipLaunchKernelGGL((decoder_row_resident_kernel<1, OP>), grid, block, 0, s, W, G, (int)D, lr, eps);
    else if (D <= 2048) hipLaunchKernelGGL((decoder_row_resident_kernel<2, OP>), grid, block, 0, s, W, G, (int)D, lr, eps);
    else if (D <= 4096) hipLaunchKernelGGL((decoder_row_resident_kernel<4, OP>), grid, block, 0, s, W, G, (int)D, lr, eps);
    else hipLaunchKernelGGL((decoder_row_resident_kernel<8, OP>), grid, block, 0, s, W, G, (int)D, lr, eps);
    return true;
}

__device__ __forceinline__ float4 sign_step4(float4 v, const float4 a, float lr) {
    v.x -= lr * (float)((a.x > 0.f) - (a.x < 0.f));
    v.y -= lr * (float)((a.y > 0.f) - (a.y < 0.f));
    v.z -= lr * (float)((a.z > 0.f) - (a.z < 0.f));
    v.w -= lr * (float)((a.w > 0.f) - (a.w < 0.f));
    return v;
}
__device__ __forceinline__ uint2 pack_bf16x4(float4 v) {
    return make_uint2((unsigned)f32_to_bf16(v.x) | ((unsigned)f32_to_bf16(v.y) << 16),
                      (unsigned)f32_to_bf16(v.z) | ((unsigned)f32_to_bf16(v.w) << 16));
}

// 8 elements per thread and iteration: two 16-B loads per operand in flight and one 16-B store
// for the bf16 shadow (8-B shadow stores ran the pass at 4.5 TB/s instead of ~5.5)
__global__ __launch_bounds__(256) void signsgd_kernel(float* __restrict__ p, const float* __restrict__ g, int64_t n,
                                                      float lr, bf16_t* __restrict__ shadow) {
    const int64_t n8 = n >> 3;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
        float4 v0 = *reinterpret_cast<const float4*>(p + i * 8);
        float4 v1 = *reinterpret_cast<const float4*>(p + i * 8 + 4);
        const float4 a0 = *reinterpret_cast<const float4*>(g + i * 8);
        const float4 a1 = *reinterpret_cast<const float4*>(g + i * 8 + 4);
        v0 = sign_step4(v0, a0, lr);
        v1 = sign_step4(v1, a1, lr);
        *reinterpret_cast<float4*>(p + i * 8) = v0;
        *reinterpret_cast<float4*>(p + i * 8 + 4) = v1;
        if (shadow != nullptr) {
            const uint2 lo = pack_bf16x4(v0), hi = pack_bf16x4(v1);
            *reinterpret_cast<uint4*>(shadow + i * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const int64_t i = (n8 << 3) + threadIdx.x;
        const float a = g[i];
        const float v = p[i] - lr * (float)((a > 0.f) - (a < 0.f));
        p[i] = v;
        if (shadow != nullptr) shadow[i] = f32_to_bf16(v);
    }
}

// ---------------------------------------------------------------------------------------
// Sign exchange for data-parallel SignSGD (parallel.py: reduce-scatter of the gradients, each
// rank takes the sign of its row block, all-gather of 1-byte signs, every rank applies them).
// sign codes: -1, 0, +1 as int8.
__device__ __forceinline__ int sgn(float v) { return (v > 0.f) - (v < 0.f); }
__device__ __forceinline__ unsigned pack_signs(float4 v) {
    return (unsigned)(sgn(v.x) & 0xff) | ((unsigned)(sgn(v.y) & 0xff) << 8) | ((unsigned)(sgn(v.z) & 0xff) << 16) |
           ((unsigned)(sgn(v.w) & 0xff) << 24);
}
__device__ __forceinline__ float4 unpack_signs(unsigned u) {
    return make_float4((float)(signed char)(u & 0xff), (float)(signed char)((u >> 8) & 0xff),
                       (float)(signed char)((u >> 16) & 0xff), (float)(signed char)(u >> 24));
}

__global__ __launch_bounds__(256) void sign_i8_kernel(const float* __restrict__ g, int64_t n, signed char* __restrict__ out) {
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
        reinterpret_cast<unsigned*>(out)[i] = pack_signs(*reinterpret_cast<const float4*>(g + i * 4));
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        out[i] = (signed char)sgn(g[i]);
    }
}

// out[j,:] = sign(G[j,:] - <G[j,:],W[j,:]> W[j,:]); row-resident (D <= 1024 R) or two sweeps (R = 0)
template <int R>
__global__ __launch_bounds__(256) void project_sign_i8_kernel(const float* __restrict__ W, const float* __restrict__ G,
                                                              int D, signed char* __restrict__ out) {
    __shared__ float red[16];
    const float* w = W + (int64_t)blockIdx.x * D;
    const float* g = G + (int64_t)blockIdx.x * D;
    signed char* o = out + (int64_t)blockIdx.x * D;
    if (R > 0) {
        float4 b[R > 0 ? R : 1], a[R > 0 ? R : 1];
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int c = (threadIdx.x + 256 * i) * 4;
            b[i] = c < D ? *reinterpret_cast<const float4*>(w + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            a[i] = c < D ? *reinterpret_cast<const float4*>(g + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            acc = fmaf(a[i].x, b[i].x, acc); acc = fmaf(a[i].y, b[i].y, acc);
            acc = fmaf(a[i].z, b[i].z, acc); acc = fmaf(a[i].w, b[i].w, acc);
        }
        const float par = block_sum(acc, red);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int c = (threadIdx.x + 256 * i) * 4;
            if (c < D)
                *reinterpret_cast<unsigned*>(o + c) = pack_signs(make_float4(
                    a[i].x - par * b[i].x, a[i].y - par * b[i].y, a[i].z - par * b[i].z, a[i].w - par * b[i].w));
        }
    } else {
        float acc = 0.f;
        for (int c = threadIdx.x; c < D; c += 256) acc = fmaf(g[c], w[c], acc);
        const float par = block_sum(acc, red);
        for (int c = threadIdx.x; c < D; c += 256) o[c] = (signed char)sgn(g[c] - par * w[c]);
    }
}

__global__ __launch_bounds__(256) void apply_signs_kernel(float* __restrict__ p, const signed char* __restrict__ sg,
                                                          int64_t n, float lr, bf16_t* __restrict__ shadow) {
    const int64_t n8 = n >> 3;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
        float4 v0 = *reinterpret_cast<const float4*>(p + i * 8);
        float4 v1 = *reinterpret_cast<const float4*>(p + i * 8 + 4);
        const uint2 u = reinterpret_cast<const uint2*>(sg)[i];
        const float4 t0 = unpack_signs(u.x), t1 = unpack_signs(u.y);
        v0.x -= lr * t0.x; v0.y -= lr * t0.y; v0.z -= lr * t0.z; v0.w -= lr * t0.w;
        v1.x -= lr * t1.x; v1.y -= lr * t1.y; v1.z -= lr * t1.z; v1.w -= lr * t1.w;
        *reinterpret_cast<float4*>(p + i * 8) = v0;
        *reinterpret_cast<float4*>(p + i * 8 + 4) = v1;
        if (shadow != nullptr) {
            const uint2 lo = pack_bf16x4(v0), hi = pack_bf16x4(v1);
            *reinterpret_cast<uint4*>(shadow + i * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const int64_t i = (n8 << 3) + threadIdx.x;
        const float v = p[i] - lr * (float)sg[i];
        p[i] = v;
        if (shadow != nullptr) shadow[i] = f32_to_bf16(v);
    }
}

// W[j,:] -= lr * signs[j,:], then W[j,:] /= ||W[j,:]|| + eps (same arithmetic order as
// decoder_row_resident_kernel<R, 2>, so the data-parallel step equals the single-GPU one)
template <int R>
__global__ __launch_bounds__(256) void apply_signs_norm_kernel(float* __restrict__ W, const signed char* __restrict__ sg,
                                                               int D, float lr, float eps, float* __restrict__ sm,
                                                               int64_t sm_rows, int64_t sm_row0) {
    __shared__ float red[16];
    float* w = W + (int64_t)blockIdx.x * D;
    const signed char* t = sg + (int64_t)blockIdx.x * D;
    float4 b[R];
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int c = (threadIdx.x + 256 * i) * 4;
        b[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < D) {
            b[i] = *reinterpret_cast<const float4*>(w + c);
            const float4 u = unpack_signs(*reinterpret_cast<const unsigned*>(t + c));
            b[i].x -= lr * u.x; b[i].y -= lr * u.y; b[i].z -= lr * u.z; b[i].w -= lr * u.w;
            acc = fmaf(b[i].x, b[i].x, acc); acc = fmaf(b[i].y, b[i].y, acc);
            acc = fmaf(b[i].z, b[i].z, acc); acc = fmaf(b[i].w, b[i].w, acc);
        }
    }
    const float denom = sqrtf(block_sum(acc, red)) + eps;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int c = (threadIdx.x + 256 * i) * 4;
        if (c < D) {
            b[i].x /= denom; b[i].y /= denom; b[i].z /= denom; b[i].w /= denom;
            *reinterpret_cast<float4*>(w + c) = b[i];
            // the slice-major copy [D/32][sm_rows][32] of the whole matrix, of which W holds the
            // rows from sm_row0 on (a row block of the data-parallel exchange)
            const int c4 = c >> 2;
            if (sm != nullptr)
                *reinterpret_cast<float4*>(sm + (((int64_t)(c4 >> 3) * sm_rows + sm_row0 + blockIdx.x) * 8 + (c4 & 7)) * 4) = b[i];
        }
    }
}

// ---------------------------------------------------------------------------------------
// Schedule-free SignSGD (the reference's default optimizer "signum":
// ScheduleFreeWrapper(SignSGD, momentum), trainer.py:146-162) in one pass. Per element, with
// y the parameter (where gradients are taken), z the base iterate, x the running average:
//     x  = lerp(y, z, 1 - 1/beta)          (undo y = beta x + (1 - beta) z)
//     z' = z - lr * sign(g)                (SignSGD on z, sign_sgd.py:23)
//     x' = lerp(x, z', c)                  (c = w_t / sum w)
//     y' = lerp(x', z', 1 - beta)
// lerp(a, b, w) = |w| < 0.5 ? a + w (b - a) : b - (b - a)(1 - w), as torch.lerp.
// (lerpf / sf_update: common.hpp, shared with the fused encoder-gradient kernels of wgrad.hip)

__global__ __launch_bounds__(256) void sf_signsgd_kernel(float* __restrict__ p, float* __restrict__ zb,
                                                         const float* __restrict__ g, int64_t n, float lr,
                                                         float w_undo, float c, float w_y, bf16_t* __restrict__ shadow) {
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 y = *reinterpret_cast<const float4*>(p + i * 4);
        float4 z = *reinterpret_cast<const float4*>(zb + i * 4);
        const float4 a = *reinterpret_cast<const float4*>(g + i * 4);
        sf_update(y.x, z.x, (float)sgn(a.x), lr, w_undo, c, w_y);
        sf_update(y.y, z.y, (float)sgn(a.y), lr, w_undo, c, w_y);
        sf_update(y.z, z.z, (float)sgn(a.z), lr, w_undo, c, w_y);
        sf_update(y.w, z.w, (float)sgn(a.w), lr, w_undo, c, w_y);
        *reinterpret_cast<float4*>(p + i * 4) = y;
        *reinterpret_cast<float4*>(zb + i * 4) = z;
        if (shadow != nullptr) *reinterpret_cast<uint2*>(shadow + i * 4) = pack_bf16x4(y);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        float y = p[i], z = zb[i];
        sf_update(y, z, (float)sgn(g[i]), lr, w_undo, c, w_y);
        p[i] = y;
        zb[i] = z;
        if (shadow != nullptr) shadow[i] = f32_to_bf16(y);
    }
}

// Decoder rows: the gradient is first projected on the row's orthogonal complement
// (sparse_coder.py:281-295, W = y), and with renorm the new y is normalised in the same pass
// (the set_decoder_norm_to_unit_norm that opens the next step). Row-resident, D <= 1024 R.
// SM: the finished row also goes to the slice-major copy, as in decoder_row_resident_kernel.
template <int R, bool SM = false>
__global__ __launch_bounds__(256) void sf_decoder_kernel(float* __restrict__ W, float* __restrict__ Z,
                                                         const float* __restrict__ G, int D, float lr, float w_undo,
                                                         float c, float w_y, int renorm, float eps,
                                                         float* __restrict__ sm = nullptr, int64_t M = 0) {
    __shared__ float red[16];
    float* w = W + (int64_t)blockIdx.x * D;
    float* zr = Z + (int64_t)blockIdx.x * D;
    const float* g = G + (int64_t)blockIdx.x * D;
    float4 y[R], z[R], a[R];
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int col = (threadIdx.x + 256 * i) * 4;
        const bool in = col < D;
        y[i] = in ? *reinterpret_cast<const float4*>(w + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        z[i] = in ? *reinterpret_cast<const float4*>(zr + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        a[i] = in ? *reinterpret_cast<const float4*>(g + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        acc = fmaf(a[i].x, y[i].x, acc); acc = fmaf(a[i].y, y[i].y, acc);
        acc = fmaf(a[i].z, y[i].z, acc); acc = fmaf(a[i].w, y[i].w, acc);
    }
    // renorm: bit 0 = normalise the new row, bit 1 = no projection (the plain step of a matrix that is neither
    // projected nor normalised, sphip_schedulefree_signsgd_step_rows: a - 0 * y is a, bit for bit)
    const float par = (renorm & 2) ? 0.f : block_sum(acc, red);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int col = (threadIdx.x + 256 * i) * 4;
        const float4 pr = make_float4(a[i].x - par * y[i].x, a[i].y - par * y[i].y, a[i].z - par * y[i].z,
                                      a[i].w - par * y[i].w);
        sf_update(y[i].x, z[i].x, (float)sgn(pr.x), lr, w_undo, c, w_y);
        sf_update(y[i].y, z[i].y, (float)sgn(pr.y), lr, w_undo, c, w_y);
        sf_update(y[i].z, z[i].z, (float)sgn(pr.z), lr, w_undo, c, w_y);
        sf_update(y[i].w, z[i].w, (float)sgn(pr.w), lr, w_undo, c, w_y);
        if (col < D) {
            ss = fmaf(y[i].x, y[i].x, ss); ss = fmaf(y[i].y, y[i].y, ss);
            ss = fmaf(y[i].z, y[i].z, ss); ss = fmaf(y[i].w, y[i].w, ss);
        }
    }
    if (renorm & 1) {
        const float denom = sqrtf(block_sum(ss, red)) + eps;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            y[i].x /= denom; y[i].y /= denom; y[i].z /= denom; y[i].w /= denom;
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int col = (threadIdx.x + 256 * i) * 4;
        if (col < D) {
            *reinterpret_cast<float4*>(w + col) = y[i];
            *reinterpret_cast<float4*>(zr + col) = z[i];
            const int c4 = col >> 2;
            if (SM) *reinterpret_cast<float4*>(sm + (((int64_t)(c4 >> 3) * M + blockIdx.x) * 8 + (c4 & 7)) * 4) = y[i];
        }
    }
}

__global__ __launch_bounds__(256) void mark_fired_kernel(const int64_t* __restrict__ idx, int64_t n, int64_t M,
                                                         uint8_t* __restrict__ fired) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int64_t j = idx[i];
        if (j >= 0 && j < M) fired[j] = 1;
    }
}

__global__ __launch_bounds__(256) void dead_counter_kernel(int64_t* __restrict__ counts, uint8_t* __restrict__ fired,
                                                           int64_t M, int64_t n_tokens) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < M) {
        counts[j] = fired[j] ? 0 : counts[j] + n_tokens;
        fired[j] = 0;
    }
}

__global__ __launch_bounds__(256) void dead_mask_kernel(const int64_t* __restrict__ counts, int64_t M,
                                                        int64_t threshold, uint8_t* __restrict__ mask,
                                                        int* __restrict__ n_dead) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int dead = (j < M && counts[j] > threshold) ? 1 : 0;
    if (j < M) mask[j] = (uint8_t)dead;
    const int c = wave_sum_i(dead);
    if ((threadIdx.x & 63) == 0 && c != 0) atomicAdd(n_dead, c);
}

static inline unsigned flat_grid(int64_t work_items) {
    int64_t b = cdiv(work_items, 256);
    if (b < 1) b = 1;
    if (b > 8192) b = 8192;
    return (unsigned)b;
}

// The same two steps with the signs given (data-parallel sign exchange, parallel.SignExchange):
// sign(mean gradient) -- for decoder rows of the projected mean gradient -- arrives as int8.
__global__ __launch_bounds__(256) void sf_apply_signs_kernel(float* __restrict__ p, float* __restrict__ zb,
                                                             const signed char* __restrict__ sg, int64_t n, float lr,
                                                             float w_undo, float c, float w_y, bf16_t* __restrict__ shadow) {
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 y = *reinterpret_cast<const float4*>(p + i * 4);
        float4 z = *reinterpret_cast<const float4*>(zb + i * 4);
        const float4 t = unpack_signs(reinterpret_cast<const unsigned*>(sg)[i]);
        sf_update(y.x, z.x, t.x, lr, w_undo, c, w_y);
        sf_update(y.y, z.y, t.y, lr, w_undo, c, w_y);
        sf_update(y.z, z.z, t.z, lr, w_undo, c, w_y);
        sf_update(y.w, z.w, t.w, lr, w_undo, c, w_y);
        *reinterpret_cast<float4*>(p + i * 4) = y;
        *reinterpret_cast<float4*>(zb + i * 4) = z;
        if (shadow != nullptr) *reinterpret_cast<uint2*>(shadow + i * 4) = pack_bf16x4(y);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        float y = p[i], z = zb[i];
        sf_update(y, z, (float)sg[i], lr, w_undo, c, w_y);
        p[i] = y;
        zb[i] = z;
        if (shadow != nullptr) shadow[i] = f32_to_bf16(y);
    }
}

template <int R>
__global__ __launch_bounds__(256) void sf_decoder_apply_signs_kernel(float* __restrict__ W, float* __restrict__ Z,
                                                                     const signed char* __restrict__ sg, int D, float lr,
                                                                     float w_undo, float c, float w_y, int renorm,
                                                                     float eps, float* __restrict__ sm, int64_t sm_rows,
                                                                     int64_t sm_row0) {
    __shared__ float red[16];
    float* w = W + (int64_t)blockIdx.x * D;
    float* zr = Z + (int64_t)blockIdx.x * D;
    const signed char* t = sg + (int64_t)blockIdx.x * D;
    float4 y[R], z[R];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int col = (threadIdx.x + 256 * i) * 4;
        y[i] = z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < D) {
            y[i] = *reinterpret_cast<const float4*>(w + col);
            z[i] = *reinterpret_cast<const float4*>(zr + col);
            const float4 u = unpack_signs(*reinterpret_cast<const unsigned*>(t + col));
            sf_update(y[i].x, z[i].x, u.x, lr, w_undo, c, w_y);
            sf_update(y[i].y, z[i].y, u.y, lr, w_undo, c, w_y);
            sf_update(y[i].z, z[i].z, u.z, lr, w_undo, c, w_y);
            sf_update(y[i].w, z[i].w, u.w, lr, w_undo, c, w_y);
            ss = fmaf(y[i].x, y[i].x, ss); ss = fmaf(y[i].y, y[i].y, ss);
            ss = fmaf(y[i].z, y[i].z, ss); ss = fmaf(y[i].w, y[i].w, ss);
        }
    }
    if (renorm) {
        const float denom = sqrtf(block_sum(ss, red)) + eps;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            y[i].x /= denom; y[i].y /= denom; y[i].z /= denom; y[i].w /= denom;
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int col = (threadIdx.x + 256 * i) * 4;
        if (col < D) {
            *reinterpret_cast<float4*>(w + col) = y[i];
            *reinterpret_cast<float4*>(zr + col) = z[i];
            const int c4 = col >> 2;
            if (sm != nullptr)
                *reinterpret_cast<float4*>(sm + (((int64_t)(c4 >> 3) * sm_rows + sm_row0 + blockIdx.x) * 8 + (c4 & 7)) * 4) = y[i];
        }
    }
}

}  // namespace sphip

extern "C" int sphip_center_cast(const void* x, int x_dtype, const float* b_dec, float* xc_f32, void* xc_bf16,
                                 int64_t N, int64_t D, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(valid_dtype(x_dtype), SPHIP_EINVAL, "center_cast: bad dtype");
    SPHIP_REQUIRE(x && (xc_f32 || xc_bf16), SPHIP_EINVAL, "center_cast: null pointer");
    SPHIP_REQUIRE(N > 0 && D > 0 && D < (1ll << 31), SPHIP_EINVAL, "center_cast: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = N * D;
    const bool vec = (D % 4 == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)xc_f32 & 15) == 0) &&
                     (((uintptr_t)xc_bf16 & 7) == 0) && (((uintptr_t)b_dec & 15) == 0);
    if (vec) {
        if (x_dtype == SPHIP_F32) hipLaunchKernelGGL((center_cast4_kernel<float>), dim3(flat_grid(n / 4)), dim3(256), 0, s, (const float*)x, b_dec, xc_f32, (bf16_t*)xc_bf16, n / 4, (int)(D / 4));
        else hipLaunchKernelGGL((center_cast4_kernel<bf16_t>), dim3(flat_grid(n / 4)), dim3(256), 0, s, (const bf16_t*)x, b_dec, xc_f32, (bf16_t*)xc_bf16, n / 4, (int)(D / 4));
    } else {
        if (x_dtype == SPHIP_F32) hipLaunchKernelGGL((center_cast_kernel<float>), dim3(flat_grid(n)), dim3(256), 0, s, (const float*)x, b_dec, xc_f32, (bf16_t*)xc_bf16, n, (int)D);
        else hipLaunchKernelGGL((center_cast_kernel<bf16_t>), dim3(flat_grid(n)), dim3(256), 0, s, (const bf16_t*)x, b_dec, xc_f32, (bf16_t*)xc_bf16, n, (int)D);
    }
    SPHIP_CHECK_LAUNCH("center_cast");
    return SPHIP_OK;
}

extern "C" int sphip_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(src && dst, SPHIP_EINVAL, "cast_f32_to_bf16: null pointer");
    SPHIP_REQUIRE(n > 0, SPHIP_EINVAL, "cast_f32_to_bf16: empty");
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (n % 4 == 0) && (((uintptr_t)src & 15) == 0) && (((uintptr_t)dst & 7) == 0);
    if (vec) hipLaunchKernelGGL((center_cast4_kernel<float>), dim3(flat_grid(n / 4)), dim3(256), 0, s, src, (const float*)nullptr, (float*)nullptr, (bf16_t*)dst, n / 4, 1);
    else hipLaunchKernelGGL((center_cast_kernel<float>), dim3(flat_grid(n)), dim3(256), 0, s, src, (const float*)nullptr, (float*)nullptr, (bf16_t*)dst, n, 1);
    SPHIP_CHECK_LAUNCH("cast_f32_to_bf16");
    return SPHIP_OK;
}

static const float SPHIP_F32_EPS = 1.1920928955078125e-07f;   // torch.finfo(torch.float32).eps

extern "C" int sphip_unit_norm_rows(float* W, int64_t M, int64_t D, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(W, SPHIP_EINVAL, "unit_norm_rows: null pointer");
    SPHIP_REQUIRE(M > 0 && D > 0 && M < (1ll << 31) && D < (1ll << 31), SPHIP_EINVAL, "unit_norm_rows: bad shape");
    const float eps = SPHIP_F32_EPS;
    hipStream_t s = (hipStream_t)stream;
    if (launch_row_resident<0>(W, W, M, D, 0.f, eps, s)) {
        SPHIP_CHECK_LAUNCH("unit_norm_rows");
        return SPHIP_OK;
    }
    if (D % 4 == 0 && ((uintptr_t)W & 15) == 0) hipLaunchKernelGGL((unit_norm_kernel<4>), dim3((unsigned)M), dim3(256), 0, s, W, (int)D, eps);
    else hipLaunchKernelGGL((unit_norm_kernel<1>), dim3((unsigned)M), dim3(256), 0, s, W, (int)D, eps);
    SPHIP_CHECK_LAUNCH("unit_norm_rows");
    return SPHIP_OK;
}

static int decoder_row_op(float* W, float* G, int64_t M, int64_t D, float lr, int mode, void* stream, const char* name) {
    using namespace sphip;
    SPHIP_REQUIRE(W && G, SPHIP_EINVAL, "%s: null pointer", name);
    SPHIP_REQUIRE(M > 0 && D > 0 && M < (1ll << 31) && D < (1ll << 31), SPHIP_EINVAL, "%s: bad shape", name);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = D % 4 == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)G & 15) == 0;
    dim3 grid((unsigned)M), block(256);
    if (mode == 0) {
        if (vec) hipLaunchKernelGGL((decoder_row_kernel<4, 0>), grid, block, 0, s, W, G, (int)D, lr);
        else hipLaunchKernelGGL((decoder_row_kernel<1, 0>), grid, block, 0, s, W, G, (int)D, lr);
    } else {
        if (vec) hipLaunchKernelGGL((decoder_row_kernel<4, 1>), grid, block, 0, s, W, G, (int)D, lr);
        else hipLaunchKernelGGL((decoder_row_kernel<1, 1>), grid, block, 0, s, W, G, (int)D, lr);
    }
    SPHIP_CHECK_LAUNCH(name);
    return SPHIP_OK;
}

extern "C" int sphip_remove_parallel_grad(float* G, const float* W, int64_t M, int64_t D, void* stream) {
    return decoder_row_op(const_cast<float*>(W), G, M, D, 0.f, 0, stream, "remove_parallel_grad");
}

extern "C" int sphip_decoder_project_signsgd(float* W, const float* G, int64_t M, int64_t D, float lr, int renorm,
                                             float* slice_major, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(W && G, SPHIP_EINVAL, "decoder_project_signsgd: null pointer");
    SPHIP_REQUIRE(M > 0 && D > 0 && M < (1ll << 31) && D < (1ll << 31), SPHIP_EINVAL, "decoder_project_signsgd: bad shape");
    SPHIP_REQUIRE(slice_major == nullptr || (D % 32 == 0 && ((uintptr_t)slice_major & 15) == 0), SPHIP_EINVAL,
                  "decoder_project_signsgd: the slice-major copy needs D %% 32 == 0 and a 16-byte aligned buffer");
    hipStream_t s = (hipStream_t)stream;
    const bool done = renorm ? launch_row_resident<2>(W, G, M, D, lr, SPHIP_F32_EPS, s, slice_major)
                             : launch_row_resident<1>(W, G, M, D, lr, SPHIP_F32_EPS, s, slice_major);
    if (done) {
        SPHIP_CHECK_LAUNCH("decoder_project_signsgd");
        return SPHIP_OK;
    }
    int st = decoder_row_op(W, const_cast<float*>(G), M, D, lr, 1, stream, "decoder_project_signsgd");
    if (st == SPHIP_OK && renorm) st = sphip_unit_norm_rows(W, M, D, stream);
    if (st == SPHIP_OK && slice_major != nullptr) st = sphip_slice_major_f32(W, slice_major, M, D, stream);
    return st;
}

extern "C" int sphip_signsgd_step(float* p, const float* g, int64_t n, float lr, void* bf16_shadow, void* stream);

extern "C" int sphip_signsgd_step_rows(float* W, const float* G, int64_t M, int64_t D, float lr, float* slice_major,
                                       void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(W && G, SPHIP_EINVAL, "signsgd_step_rows: null pointer");
    SPHIP_REQUIRE(M > 0 && D > 0 && M < (1ll << 31) && D < (1ll << 31), SPHIP_EINVAL, "signsgd_step_rows: bad shape");
    SPHIP_REQUIRE(slice_major == nullptr || (D % 32 == 0 && ((uintptr_t)slice_major & 15) == 0), SPHIP_EINVAL,
                  "signsgd_step_rows: the slice-major copy needs D %% 32 == 0 and a 16-byte aligned buffer");
    if (launch_row_resident<3>(W, G, M, D, lr, SPHIP_F32_EPS, (hipStream_t)stream, slice_major)) {
        SPHIP_CHECK_LAUNCH("signsgd_step_rows");
        return SPHIP_OK;
    }
    SPHIP_REQUIRE((M * D) % 8 == 0, SPHIP_EUNSUPPORTED, "signsgd_step_rows: M * D must be a multiple of 8");
    int st = sphip_signsgd_step(W, G, M * D, lr, nullptr, stream);
    if (st == SPHIP_OK && slice_major != nullptr) st = sphip_slice_major_f32(W, slice_major, M, D, stream);
    return st;
}

extern "C" int sphip_signsgd_step(float* p, const float* g, int64_t n, float lr, void* bf16_shadow, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(p && g, SPHIP_EINVAL, "signsgd_step: null pointer");
    SPHIP_REQUIRE(n > 0, SPHIP_EINVAL, "signsgd_step: empty");
    SPHIP_REQUIRE(((uintptr_t)p & 15) == 0 && ((uintptr_t)g & 15) == 0 && ((uintptr_t)bf16_shadow & 15) == 0, SPHIP_EINVAL,
                  "signsgd_step: operands must be 16-byte aligned");
    hipLaunchKernelGGL(signsgd_kernel, dim3(flat_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, p, g, n, lr, (bf16_t*)bf16_shadow);
    SPHIP_CHECK_LAUNCH("signsgd_step");
    return SPHIP_OK;
}

#ifndef SPHIP_ADAM_NT
#define SPHIP_ADAM_NT 1
#endif
namespace sphip {
// torch.optim.Adam's update (betas, eps, no weight decay, no amsgrad) in one pass over
// (p, g, exp_avg, exp_avg_sq), optionally refreshing the bf16 shadow of p. The operations and
// their order are those of torch's single-tensor implementation: m.lerp_(g, 1 - b1);
// v.mul_(b2).addcmul_(g, g, 1 - b2); denom = sqrt(v) / sqrt(bc2) + eps; p.addcdiv_(m, denom, -lr / bc1).
__device__ __forceinline__ float adam_elem(float& p, float g, float& m, float& v, float w1, float b2, float w2,
                                           float step_size, float bc2_sqrt, float eps) {
    m = lerpf(m, g, w1);                                   // torch.lerp's two branches (common.hpp): beta1 <= 0.5 takes the other
    v = __fadd_rn(__fmul_rn(v, b2), __fmul_rn(__fmul_rn(w2, g), g));
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bc2_sqrt), eps);
    p = __fadd_rn(p, __fmul_rn(-step_size, __fdiv_rn(m, denom)));
    return p;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float w1,
                                                   float b2, float w2, float step_size, float bc2_sqrt, float eps,
                                                   bf16_t* __restrict__ shadow) {
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    // every stream is touched once per step and is far larger than the caches: non-temporal
    typedef float f4 __attribute__((ext_vector_type(4)));
    auto ldx = [](const float* q) {
#if SPHIP_ADAM_NT
        const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(q));
        return make_float4(t[0], t[1], t[2], t[3]);
#else
        return *reinterpret_cast<const float4*>(q);
#endif
    };
    auto stx = [](float* q, float4 x) {
#if SPHIP_ADAM_NT
        const f4 t = {x.x, x.y, x.z, x.w};
        __builtin_nontemporal_store(t, reinterpret_cast<f4*>(q));
#else
        *reinterpret_cast<float4*>(q) = x;
#endif
    };
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 P = ldx(p + i * 4), M = ldx(m + i * 4), V = ldx(v + i * 4);
        const float4 G = ldx(g + i * 4);
        adam_elem(P.x, G.x, M.x, V.x, w1, b2, w2, step_size, bc2_sqrt, eps);
        adam_elem(P.y, G.y, M.y, V.y, w1, b2, w2, step_size, bc2_sqrt, eps);
        adam_elem(P.z, G.z, M.z, V.z, w1, b2, w2, step_size, bc2_sqrt, eps);
        adam_elem(P.w, G.w, M.w, V.w, w1, b2, w2, step_size, bc2_sqrt, eps);
        stx(p + i * 4, P), stx(m + i * 4, M), stx(v + i * 4, V);
        if (shadow != nullptr) *reinterpret_cast<uint2*>(shadow + i * 4) = pack_bf16x4(P);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        float P = p[i], M = m[i], V = v[i];
        adam_elem(P, g[i], M, V, w1, b2, w2, step_size, bc2_sqrt, eps);
        p[i] = P, m[i] = M, v[i] = V;
        if (shadow != nullptr) shadow[i] = f32_to_bf16(P);
    }
}
}  // namespace sphip

extern "C" int sphip_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                               double beta1, double beta2, double eps, int64_t step, void* bf16_shadow, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(p && g && exp_avg && exp_avg_sq, SPHIP_EINVAL, "adam_step: null pointer");
    SPHIP_REQUIRE(n > 0 && step >= 1, SPHIP_EINVAL, "adam_step: n and step must be positive");
    SPHIP_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0, SPHIP_EINVAL, "adam_step: bad hyperparameters");
    SPHIP_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0 &&
                      ((uintptr_t)bf16_shadow & 7) == 0,
                  SPHIP_EINVAL, "adam_step: operands must be 16-byte aligned");
    // scalars in double as torch computes them on the host (1 - beta, 1 - beta ** step, lr / bc1),
    // rounded to fp32 once, where torch hands them to its kernels
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(adam_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, p, g, exp_avg, exp_avg_sq, n,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), step_size, bc2_sqrt, (float)eps,
                       (bf16_t*)bf16_shadow);
    SPHIP_CHECK_LAUNCH("adam_step");
    return SPHIP_OK;
}

extern "C" int sphip_sign_i8(const float* g, int64_t n, int8_t* out, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(g && out, SPHIP_EINVAL, "sign_i8: null pointer");
    SPHIP_REQUIRE(n > 0, SPHIP_EINVAL, "sign_i8: empty");
    SPHIP_REQUIRE(((uintptr_t)g & 15) == 0 && ((uintptr_t)out & 3) == 0, SPHIP_EINVAL, "sign_i8: operands must be aligned (16 B / 4 B)");
    hipLaunchKernelGGL(sign_i8_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, g, n, (signed char*)out);
    SPHIP_CHECK_LAUNCH("sign_i8");
    return SPHIP_OK;
}

extern "C" int sphip_decoder_project_sign_i8(const float* W, const float* G, int64_t M, int64_t D, int8_t* out, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(W && G && out, SPHIP_EINVAL, "decoder_project_sign_i8: null pointer");
    SPHIP_REQUIRE(M > 0 && D > 0 && M < (1ll << 31) && D < (1ll << 31), SPHIP_EINVAL, "decoder_project_sign_i8: bad shape");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)M), block(256);
    signed char* o = (signed char*)out;
    const bool vec = D % 4 == 0 && D <= 8192 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)G & 15) == 0 && ((uintptr_t)out & 3) == 0;
    if (!vec) hipLaunchKernelGGL((project_sign_i8_kernel<0>), grid, block, 0, s, W, G, (int)D, o);
    else if (D <= 1024) hipLaunchKernelGGL((project_sign_i8_kernel<1>), grid, block, 0, s, W, G, (int)D, o);
    else if (D <= 2048) hipLaunchKernelGGL((project_sign_i8_kernel<2>), grid, block, 0, s, W, G, (int)D, o);
    else if (D <= 4096) hipLaunchKernelGGL((project_sign_i8_kernel<4>), grid, block, 0, s, W, G, (int)D, o);
    else hipLaunchKernelGGL((project_sign_i8_kernel<8>), grid, block, 0, s, W, G, (int)D, o);
    SPHIP_CHECK_LAUNCH("decoder_project_sign_i8");
    return SPHIP_OK;
}

extern "C" int sphip_apply_signs(float* p, const int8_t* signs, int64_t n, float lr, void* bf16_shadow, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(p && signs, SPHIP_EINVAL, "apply_signs: null pointer");
    SPHIP_REQUIRE(n > 0, SPHIP_EINVAL, "apply_signs: empty");
    SPHIP_REQUIRE(((uintptr_t)p & 15) == 0 && ((uintptr_t)signs & 7) == 0 && ((uintptr_t)bf16_shadow & 15) == 0, SPHIP_EINVAL,
                  "apply_signs: operands must be aligned (16 B / 8 B / 16 B)");
    hipLaunchKernelGGL(apply_signs_kernel, dim3(flat_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, p, (const signed char*)signs, n, lr, (bf16_t*)bf16_shadow);
    SPHIP_CHECK_LAUNCH("apply_signs");
    return SPHIP_OK;
}

extern "C" int sphip_decoder_apply_signs(float* W, const int8_t* signs, int64_t M, int64_t D, float lr, int renorm,
                                         float* slice_major, int64_t sm_rows, int64_t sm_row0, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(W && signs, SPHIP_EINVAL, "decoder_apply_signs: null pointer");
    SPHIP_REQUIRE(M > 0 && D > 0 && M < (1ll << 31) && D < (1ll << 31), SPHIP_EINVAL, "decoder_apply_signs: bad shape");
    SPHIP_REQUIRE(slice_major == nullptr || (D % 32 == 0 && ((uintptr_t)slice_major & 15) == 0 && sm_row0 >= 0 &&
                                             sm_row0 + M <= sm_rows),
                  SPHIP_EINVAL, "decoder_apply_signs: the slice-major copy needs D %% 32 == 0, a 16-byte aligned buffer and "
                  "rows [%lld, %lld) inside its %lld rows", (long long)sm_row0, (long long)(sm_row0 + M), (long long)sm_rows);
    hipStream_t s = (hipStream_t)stream;
    const signed char* t = (const signed char*)signs;
    const bool vec = D % 4 == 0 && D <= 8192 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)signs & 3) == 0;
    if (renorm && vec) {
        dim3 grid((unsigned)M), block(256);
        if (D <= 1024) hipLaunchKernelGGL((apply_signs_norm_kernel<1>), grid, block, 0, s, W, t, (int)D, lr, SPHIP_F32_EPS, slice_major, sm_rows, sm_row0);
        else if (D <= 2048) hipLaunchKernelGGL((apply_signs_norm_kernel<2>), grid, block, 0, s, W, t, (int)D, lr, SPHIP_F32_EPS, slice_major, sm_rows, sm_row0);
        else if (D <= 4096) hipLaunchKernelGGL((apply_signs_norm_kernel<4>), grid, block, 0, s, W, t, (int)D, lr, SPHIP_F32_EPS, slice_major, sm_rows, sm_row0);
        else hipLaunchKernelGGL((apply_signs_norm_kernel<8>), grid, block, 0, s, W, t, (int)D, lr, SPHIP_F32_EPS, slice_major, sm_rows, sm_row0);
        SPHIP_CHECK_LAUNCH("decoder_apply_signs");
        return SPHIP_OK;
    }
    SPHIP_REQUIRE(slice_major == nullptr, SPHIP_EUNSUPPORTED,
                  "decoder_apply_signs: the slice-major copy is written by the fused form only (renorm, D %% 4 == 0, D <= 8192)");
    const int st = sphip_apply_signs(W, signs, M * D, lr, nullptr, stream);
    if (st != SPHIP_OK || !renorm) return st;
    return sphip_unit_norm_rows(W, M, D, stream);
}

extern "C" int sphip_schedulefree_signsgd_step(float* p, float* z, const float* g, int64_t n, float lr, float beta,
                                               float c, void* bf16_shadow, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(p && z && g, SPHIP_EINVAL, "schedulefree_signsgd_step: null pointer");
    SPHIP_REQUIRE(n > 0 && beta > 0.f && beta <= 1.f, SPHIP_EINVAL, "schedulefree_signsgd_step: bad n / beta");
    SPHIP_REQUIRE(((uintptr_t)p & 15) == 0 && ((uintptr_t)z & 15) == 0 && ((uintptr_t)g & 15) == 0 &&
                      ((uintptr_t)bf16_shadow & 7) == 0,
                  SPHIP_EINVAL, "schedulefree_signsgd_step: operands must be 16-byte aligned");
    hipLaunchKernelGGL(sf_signsgd_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, p, z, g, n, lr,
                       1.f - 1.f / beta, c, 1.f - beta, (bf16_t*)bf16_shadow);
    SPHIP_CHECK_LAUNCH("schedulefree_signsgd_step");
    return SPHIP_OK;
}

static int sf_row_step(float* W, float* Z, const float* G, int64_t M, int64_t D, float lr, float beta, float c, int renorm,
                       float* slice_major, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(W && Z && G, SPHIP_EINVAL, "decoder_schedulefree_signsgd_step: null pointer");
    SPHIP_REQUIRE(M > 0 && D > 0 && M < (1ll << 31) && beta > 0.f && beta <= 1.f, SPHIP_EINVAL,
                  "decoder_schedulefree_signsgd_step: bad shape / beta");
    SPHIP_REQUIRE(D % 4 == 0 && D <= 8192 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)Z & 15) == 0 && ((uintptr_t)G & 15) == 0,
                  SPHIP_EUNSUPPORTED, "decoder_schedulefree_signsgd_step: needs D %% 4 == 0, D <= 8192, 16-byte aligned rows");
    SPHIP_REQUIRE(slice_major == nullptr || (D % 32 == 0 && ((uintptr_t)slice_major & 15) == 0), SPHIP_EINVAL,
                  "decoder_schedulefree_signsgd_step: the slice-major copy needs D %% 32 == 0 and a 16-byte aligned buffer");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)M), block(256);
    const float wu = 1.f - 1.f / beta, wy = 1.f - beta;
    if (slice_major != nullptr) {
        float* sm = slice_major;
        if (D <= 1024) hipLaunchKernelGGL((sf_decoder_kernel<1, true>), grid, block, 0, s, W, Z, G, (int)D, lr, wu, c, wy, renorm, SPHIP_F32_EPS, sm, M);
        else if (D <= 2048) hipLaunchKernelGGL((sf_decoder_kernel<2, true>), grid, block, 0, s, W, Z, G, (int)D, lr, wu, c, wy, renorm, SPHIP_F32_EPS, sm, M);
        else if (D <= 4096) hipLaunchKernelGGL((sf_decoder_kernel<4, true>), grid, block, 0, s, W, Z, G, (int)D, lr, wu, c, wy, renorm, SPHIP_F32_EPS, sm, M);
        else hipLaunchKernelGGL((sf_decoder_kernel<8, true>), grid, block, 0, s, W, Z, G, (int)D, lr, wu, c, wy, renorm, SPHIP_F32_EPS, sm, M);
        SPHIP_CHECK_LAUNCH("decoder_schedulefree_signsgd_step");
        return SPHIP_OK;
    }
    if (D <= 1024) hipLaunchKernelGGL((sf_decoder_kernel<1>), grid, block, 0, s, W, Z, G, (int)D, lr, wu, c, wy, renorm, SPHIP_F32_EPS);
    else if (D <= 2048) hipLaunchKernelGGL((sf_decoder_kernel<2>), grid, block, 0, s, W, Z, G, (int)D, lr, wu, c, wy, renorm, SPHIP_F32_EPS);
    else if (D <= 4096) hipLaunchKernelGGL((sf_decoder_kernel<4>), grid, block, 0, s, W, Z, G, (int)D, lr, wu, c, wy, renorm, SPHIP_F32_EPS);
    else hipLaunchKernelGGL((sf_decoder_kernel<8>), grid, block, 0, s, W, Z, G, (int)D, lr, wu, c, wy, renorm, SPHIP_F32_EPS);
    SPHIP_CHECK_LAUNCH("decoder_schedulefree_signsgd_step");
    return SPHIP_OK;
}

extern "C" int sphip_decoder_schedulefree_signsgd_step(float* W, float* Z, const float* G, int64_t M, int64_t D, float lr,
                                                       float beta, float c, int renorm, float* slice_major, void* stream) {
    return sf_row_step(W, Z, G, M, D, lr, beta, c, renorm ? 1 : 0, slice_major, stream);
}

extern "C" int sphip_schedulefree_signsgd_step_rows(float* W, float* Z, const float* G, int64_t M, int64_t D, float lr,
                                                    float beta, float c, float* slice_major, void* stream) {
    return sf_row_step(W, Z, G, M, D, lr, beta, c, 2, slice_major, stream);
}

extern "C" int sphip_schedulefree_apply_signs(float* p, float* z, const int8_t* signs, int64_t n, float lr, float beta,
                                              float c, void* bf16_shadow, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(p && z && signs, SPHIP_EINVAL, "schedulefree_apply_signs: null pointer");
    SPHIP_REQUIRE(n > 0 && beta > 0.f && beta <= 1.f, SPHIP_EINVAL, "schedulefree_apply_signs: bad n / beta");
    SPHIP_REQUIRE(((uintptr_t)p & 15) == 0 && ((uintptr_t)z & 15) == 0 && ((uintptr_t)signs & 3) == 0 &&
                      ((uintptr_t)bf16_shadow & 7) == 0,
                  SPHIP_EINVAL, "schedulefree_apply_signs: operands must be 16-byte aligned (signs: 4)");
    hipLaunchKernelGGL(sf_apply_signs_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, p, z,
                       (const signed char*)signs, n, lr, 1.f - 1.f / beta, c, 1.f - beta, (bf16_t*)bf16_shadow);
    SPHIP_CHECK_LAUNCH("schedulefree_apply_signs");
    return SPHIP_OK;
}

extern "C" int sphip_decoder_schedulefree_apply_signs(float* W, float* Z, const int8_t* signs, int64_t M, int64_t D,
                                                      float lr, float beta, float c, int renorm, float* slice_major,
                                                      int64_t sm_rows, int64_t sm_row0, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(W && Z && signs, SPHIP_EINVAL, "decoder_schedulefree_apply_signs: null pointer");
    SPHIP_REQUIRE(slice_major == nullptr || (D % 32 == 0 && ((uintptr_t)slice_major & 15) == 0 && sm_row0 >= 0 &&
                                             sm_row0 + M <= sm_rows),
                  SPHIP_EINVAL, "decoder_schedulefree_apply_signs: the slice-major copy needs D %% 32 == 0, a 16-byte aligned "
                  "buffer and rows [%lld, %lld) inside its %lld rows", (long long)sm_row0, (long long)(sm_row0 + M),
                  (long long)sm_rows);
    SPHIP_REQUIRE(M > 0 && D > 0 && M < (1ll << 31) && beta > 0.f && beta <= 1.f, SPHIP_EINVAL,
                  "decoder_schedulefree_apply_signs: bad shape / beta");
    SPHIP_REQUIRE(D % 4 == 0 && D <= 8192 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)Z & 15) == 0 &&
                      ((uintptr_t)signs & 3) == 0,
                  SPHIP_EUNSUPPORTED, "decoder_schedulefree_apply_signs: needs D %% 4 == 0, D <= 8192, aligned rows");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)M), block(256);
    const float wu = 1.f - 1.f / beta, wy = 1.f - beta;
    const signed char* t = (const signed char*)signs;
    if (D <= 1024) hipLaunchKernelGGL((sf_decoder_apply_signs_kernel<1>), grid, block, 0, s, W, Z, t, (int)D, lr, wu, c, wy, renorm, SPHIP_F32_EPS, slice_major, sm_rows, sm_row0);
    else if (D <= 2048) hipLaunchKernelGGL((sf_decoder_apply_signs_kernel<2>), grid, block, 0, s, W, Z, t, (int)D, lr, wu, c, wy, renorm, SPHIP_F32_EPS, slice_major, sm_rows, sm_row0);
    else if (D <= 4096) hipLaunchKernelGGL((sf_decoder_apply_signs_kernel<4>), grid, block, 0, s, W, Z, t, (int)D, lr, wu, c, wy, renorm, SPHIP_F32_EPS, slice_major, sm_rows, sm_row0);
    else hipLaunchKernelGGL((sf_decoder_apply_signs_kernel<8>), grid, block, 0, s, W, Z, t, (int)D, lr, wu, c, wy, renorm, SPHIP_F32_EPS, slice_major, sm_rows, sm_row0);
    SPHIP_CHECK_LAUNCH("decoder_schedulefree_apply_signs");
    return SPHIP_OK;
}

extern "C" int sphip_mark_fired(const int64_t* idx, int64_t n_entries, int64_t M, uint8_t* did_fire, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(idx && did_fire, SPHIP_EINVAL, "mark_fired: null pointer");
    SPHIP_REQUIRE(n_entries > 0 && M > 0, SPHIP_EINVAL, "mark_fired: bad shape");
    hipLaunchKernelGGL(mark_fired_kernel, dim3(flat_grid(n_entries)), dim3(256), 0, (hipStream_t)stream, idx, n_entries, M, did_fire);
    SPHIP_CHECK_LAUNCH("mark_fired");
    return SPHIP_OK;
}

extern "C" int sphip_dead_counter_step(int64_t* counts, uint8_t* did_fire, int64_t M, int64_t n_tokens, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(counts && did_fire, SPHIP_EINVAL, "dead_counter_step: null pointer");
    SPHIP_REQUIRE(M > 0 && n_tokens >= 0, SPHIP_EINVAL, "dead_counter_step: bad shape");
    hipLaunchKernelGGL(dead_counter_kernel, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, counts, did_fire, M, n_tokens);
    SPHIP_CHECK_LAUNCH("dead_counter_step");
    return SPHIP_OK;
}

extern "C" int sphip_dead_mask(const int64_t* counts, int64_t M, int64_t threshold, uint8_t* mask, int32_t* n_dead,
                               void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(counts && mask && n_dead, SPHIP_EINVAL, "dead_mask: null pointer");
    SPHIP_REQUIRE(M > 0, SPHIP_EINVAL, "dead_mask: bad shape");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(n_dead, 0, sizeof(int32_t), s) != hipSuccess) {
        set_error("dead_mask: memset failed");
        return SPHIP_ELAUNCH;
    }
    hipLaunchKernelGGL(dead_mask_kernel, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, s, counts, M, threshold, mask, n_dead);
    SPHIP_CHECK_LAUNCH("dead_mask");
    return SPHIP_OK;
}
