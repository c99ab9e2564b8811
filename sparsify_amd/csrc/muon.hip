// Muon step pieces around the Newton-Schulz products (sparsify/muon.py:11-40, 115-181). The
// products themselves run on the bf16 MFMA GEMM kernels with a fused axpy epilogue
// (sphip_gemm_bf16_axpy, encoder_gemm.hip / encoder_gemm256.hip); this file holds the
// byte-bound passes: momentum + Nesterov + bf16 cast + Frobenius sum of squares in one pass over
// the gradient, the normalisation, a bf16 transposition (the iterate is kept in the tall
// orientation, its transpose is what X X^T contracts over), and the parameter update.
#include "common.hpp"

namespace sphip {

constexpr int MU_THREADS = 256;
constexpr int MU_VEC = 4;
constexpr int MU_MAX_BLOCKS = 2048;      // partial sums of the Frobenius norm (workspace floats)

// torch.lerp(a, b, w): a + w (b - a) for w < 0.5, else b - (b - a) (1 - w)  (ATen lerp kernel)
__device__ __forceinline__ float torch_lerp(float a, float b, float w) {
    const float d = b - a;
    return w < 0.5f ? a + w * d : b - d * (1.f - w);
}

// muon.py:157-160: buf.lerp_(g, 1 - beta); u = nesterov ? g.lerp(buf, beta) : buf.
// Writes buf, bf16(u) (muon.py:24: X = G.bfloat16()) and this block's sum of bf16(u)^2.
__global__ __launch_bounds__(MU_THREADS) void muon_momentum_kernel(const float* __restrict__ g, float* __restrict__ buf,
                                                                  int64_t n, float beta, int nesterov,
                                                                  bf16_t* __restrict__ u_out, float* __restrict__ partial) {
    __shared__ float red[16];
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * MU_THREADS * MU_VEC;
    for (int64_t i = ((int64_t)blockIdx.x * MU_THREADS + threadIdx.x) * MU_VEC; i < n; i += stride) {
        if (i + MU_VEC <= n) {
            const float4 g4 = *reinterpret_cast<const float4*>(g + i);
            const float4 b4 = beta > 0.f ? *reinterpret_cast<const float4*>(buf + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
            float bv[4] = {b4.x, b4.y, b4.z, b4.w};
            bf16_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float u = gv[j];
                if (beta > 0.f) {
                    bv[j] = torch_lerp(bv[j], gv[j], 1.f - beta);
                    u = nesterov ? torch_lerp(gv[j], bv[j], beta) : bv[j];
                }
                o[j] = f32_to_bf16(u);
                const float r = bf16_to_f32(o[j]);
                acc += r * r;
            }
            if (beta > 0.f) *reinterpret_cast<float4*>(buf + i) = make_float4(bv[0], bv[1], bv[2], bv[3]);
            *reinterpret_cast<uint2*>(u_out + i) =
                make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16));
        } else {
            for (int64_t e = i; e < n; ++e) {
                float u = g[e];
                if (beta > 0.f) {
                    const float b = torch_lerp(buf[e], g[e], 1.f - beta);
                    buf[e] = b;
                    u = nesterov ? torch_lerp(g[e], b, beta) : b;
                }
                const bf16_t o = f32_to_bf16(u);
                u_out[e] = o;
                const float r = bf16_to_f32(o);
                acc += r * r;
            }
        }
    }
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// norm[0] = sum of the partials (fixed order), norm[1] = bf16(bf16(sqrt(sum)) + 1e-7): X.norm() of
// a bf16 tensor is a bf16 value, and so is norm + 1e-7 (muon.py:31). With nparts = 0 the sum is
// taken from norm[0] as it stands (row-sharded Newton-Schulz: the ranks' sums were added there).
__global__ __launch_bounds__(MU_THREADS) void muon_norm_kernel(const float* __restrict__ partial, int nparts,
                                                              float* __restrict__ norm) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nparts; i += MU_THREADS) acc += partial[i];
    float tot = block_sum(acc, red);
    if (threadIdx.x == 0) {
        if (nparts == 0) tot = norm[0];
        const float nrm = bf16_to_f32(f32_to_bf16(sqrtf(tot)));
        norm[0] = tot;
        norm[1] = bf16_to_f32(f32_to_bf16(nrm + 1e-7f));
    }
}

// x = bf16(x / denom[0]) in place (muon.py:31)
__global__ __launch_bounds__(MU_THREADS) void scale_bf16_kernel(bf16_t* __restrict__ x, int64_t n,
                                                               const float* __restrict__ denom) {
    const float d = denom[1];          // the pair muon_norm_kernel writes: {sum of squares, denominator}
    const int64_t stride = (int64_t)gridDim.x * MU_THREADS * 8;
    for (int64_t i = ((int64_t)blockIdx.x * MU_THREADS + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 8 <= n) {
            uint4 raw = *reinterpret_cast<const uint4*>(x + i);
            unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16_t lo = f32_to_bf16(bf16_to_f32((bf16_t)(w[j] & 0xffffu)) / d);
                const bf16_t hi = f32_to_bf16(bf16_to_f32((bf16_t)(w[j] >> 16)) / d);
                w[j] = (unsigned)lo | ((unsigned)hi << 16);
            }
            *reinterpret_cast<uint4*>(x + i) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (int64_t e = i; e < n; ++e) x[e] = f32_to_bf16(bf16_to_f32(x[e]) / d);
        }
    }
}

// dst (C, ld) = src (R, C)^T, 64 x 64 tiles through LDS; both sides move contiguous row pieces.
// Columns R..ld-1 of dst (padding up to a multiple the GEMM's contraction needs) are zero-filled.
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t R,
                                                       int64_t C, int64_t ld) {
    __shared__ T tile[64][65 + (sizeof(T) == 2 ? 1 : 0)];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int64_t r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[r * C + c] : (T)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int64_t c = c0 + i, r = r0 + tx;
        if (c < C && r < ld) dst[c * ld + r] = tile[tx][i];        // rows past R were staged as zeros
    }
}

// out = a (fp32) + b (bf16 or fp32): the skip term's product added to the reconstruction
// (sparse_coder.py:221-222; under autocast the product is a bf16 tensor, the sum fp32)
template <typename T>
__global__ __launch_bounds__(MU_THREADS) void add_kernel(const float* __restrict__ a, const T* __restrict__ b,
                                                        float* __restrict__ out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * MU_THREADS * MU_VEC;
    for (int64_t i = ((int64_t)blockIdx.x * MU_THREADS + threadIdx.x) * MU_VEC; i < n; i += stride) {
        if (i + MU_VEC <= n) {
            float4 d = *reinterpret_cast<const float4*>(a + i);
            const float4 y = load4(b + i);
            d.x += y.x, d.y += y.y, d.z += y.z, d.w += y.w;
            *reinterpret_cast<float4*>(out + i) = d;
        } else {
            for (int64_t e = i; e < n; ++e) out[e] = a[e] + Elem<T>::load(b, e);
        }
    }
}

// muon.py:121-122 + :180: p = p * decay; p += -step * o (o bf16, the orthogonalised update).
__global__ __launch_bounds__(MU_THREADS) void muon_apply_kernel(float* __restrict__ p, const bf16_t* __restrict__ o,
                                                               int64_t n, float decay, float step,
                                                               bf16_t* __restrict__ shadow) {
    const int64_t stride = (int64_t)gridDim.x * MU_THREADS * MU_VEC;
    for (int64_t i = ((int64_t)blockIdx.x * MU_THREADS + threadIdx.x) * MU_VEC; i < n; i += stride) {
        if (i + MU_VEC <= n) {
            float4 p4 = *reinterpret_cast<const float4*>(p + i);
            const float4 o4 = load4(o + i);
            float pv[4] = {p4.x, p4.y, p4.z, p4.w};
            const float ov[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = pv[j] * decay;
                pv[j] = t - step * ov[j];
            }
            *reinterpret_cast<float4*>(p + i) = make_float4(pv[0], pv[1], pv[2], pv[3]);
            if (shadow != nullptr) {
                const bf16_t s0 = f32_to_bf16(pv[0]), s1 = f32_to_bf16(pv[1]), s2 = f32_to_bf16(pv[2]), s3 = f32_to_bf16(pv[3]);
                *reinterpret_cast<uint2*>(shadow + i) =
                    make_uint2((unsigned)s0 | ((unsigned)s1 << 16), (unsigned)s2 | ((unsigned)s3 << 16));
            }
        } else {
            for (int64_t e = i; e < n; ++e) {
                const float t = p[e] * decay;
                p[e] = t - step * bf16_to_f32(o[e]);
                if (shadow != nullptr) shadow[e] = f32_to_bf16(p[e]);
            }
        }
    }
}

static int stream_blocks(int64_t n, int per_thread) {
    const int64_t want = cdiv(n, (int64_t)MU_THREADS * per_thread);
    return (int)(want < 1 ? 1 : (want > MU_MAX_BLOCKS ? MU_MAX_BLOCKS : want));
}

}  // namespace sphip

extern "C" size_t sphip_muon_workspace_bytes(void) { return (size_t)(sphip::MU_MAX_BLOCKS + 64) * sizeof(float); }

extern "C" int sphip_muon_momentum(const float* g, float* buf, int64_t n, float beta, int nesterov, void* u_bf16,
                                   float* norm, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(g && u_bf16 && norm && workspace && (buf || beta <= 0.f), SPHIP_EINVAL, "muon_momentum: null pointer");
    SPHIP_REQUIRE(n > 0, SPHIP_EINVAL, "muon_momentum: empty tensor");
    SPHIP_REQUIRE(beta >= 0.f && beta < 1.f, SPHIP_EINVAL, "muon_momentum: momentum %g outside [0, 1)", (double)beta);
    SPHIP_REQUIRE(workspace_bytes >= sphip_muon_workspace_bytes(), SPHIP_EWORKSPACE,
                  "muon_momentum: workspace too small (%zu < %zu)", workspace_bytes, sphip_muon_workspace_bytes());
    SPHIP_REQUIRE(((uintptr_t)g & 15) == 0 && ((uintptr_t)buf & 15) == 0 && ((uintptr_t)u_bf16 & 7) == 0, SPHIP_EINVAL,
                  "muon_momentum: operands must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int blocks = stream_blocks(n, MU_VEC * 4);
    float* partial = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(muon_momentum_kernel, dim3(blocks), dim3(MU_THREADS), 0, s, g, buf, n, beta, nesterov,
                       (bf16_t*)u_bf16, partial);
    hipLaunchKernelGGL(muon_norm_kernel, dim3(1), dim3(MU_THREADS), 0, s, (const float*)partial, blocks, norm);
    SPHIP_CHECK_LAUNCH("muon_momentum");
    return SPHIP_OK;
}

extern "C" int sphip_muon_denominator(float* norm, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(norm, SPHIP_EINVAL, "muon_denominator: null pointer");
    hipLaunchKernelGGL(muon_norm_kernel, dim3(1), dim3(MU_THREADS), 0, (hipStream_t)stream, (const float*)nullptr, 0, norm);
    SPHIP_CHECK_LAUNCH("muon_denominator");
    return SPHIP_OK;
}

extern "C" int sphip_muon_normalise(void* x, int64_t n, const float* norm, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(x && norm, SPHIP_EINVAL, "muon_normalise: null pointer");
    SPHIP_REQUIRE(n > 0 && ((uintptr_t)x & 15) == 0, SPHIP_EINVAL, "muon_normalise: empty or misaligned tensor");
    hipLaunchKernelGGL(scale_bf16_kernel, dim3(stream_blocks(n, 32)), dim3(MU_THREADS), 0, (hipStream_t)stream, (bf16_t*)x, n,
                       norm);
    SPHIP_CHECK_LAUNCH("muon_normalise");
    return SPHIP_OK;
}

extern "C" int sphip_transpose(const void* src, void* dst, int64_t R, int64_t C, int64_t ld_dst, int dtype,
                               void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(valid_dtype(dtype), SPHIP_EINVAL, "transpose: bad dtype %d", dtype);
    SPHIP_REQUIRE(src && dst && src != dst, SPHIP_EINVAL, "transpose: null or aliased pointers");
    SPHIP_REQUIRE(R > 0 && C > 0 && ld_dst >= R && ld_dst < R + 64 && cdiv(ld_dst, 64) < 65536 && cdiv(C, 64) < (1ll << 31),
                  SPHIP_EINVAL, "transpose: bad shape R=%lld C=%lld ld_dst=%lld (R <= ld_dst < R + 64)", (long long)R,
                  (long long)C, (long long)ld_dst);
    dim3 grid((unsigned)cdiv(C, 64), (unsigned)cdiv(ld_dst, 64)), block(256);
    if (dtype == SPHIP_BF16)
        hipLaunchKernelGGL((transpose_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst,
                           R, C, ld_dst);
    else
        hipLaunchKernelGGL((transpose_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)src, (float*)dst, R,
                           C, ld_dst);
    SPHIP_CHECK_LAUNCH("transpose");
    return SPHIP_OK;
}

extern "C" int sphip_add(const float* a, const void* b, int b_dtype, float* out, int64_t n, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(valid_dtype(b_dtype), SPHIP_EINVAL, "add: bad dtype %d", b_dtype);
    SPHIP_REQUIRE(a && b && out && n > 0, SPHIP_EINVAL, "add: null pointer or empty tensor");
    SPHIP_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)b & (b_dtype == SPHIP_BF16 ? 7 : 15)) == 0,
                  SPHIP_EINVAL, "add: misaligned tensor");
    const dim3 grid(stream_blocks(n, MU_VEC * 4)), block(MU_THREADS);
    if (b_dtype == SPHIP_BF16)
        hipLaunchKernelGGL((add_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, a, (const bf16_t*)b, out, n);
    else
        hipLaunchKernelGGL((add_kernel<float>), grid, block, 0, (hipStream_t)stream, a, (const float*)b, out, n);
    SPHIP_CHECK_LAUNCH("add");
    return SPHIP_OK;
}

extern "C" int sphip_muon_apply(float* p, const void* o_bf16, int64_t n, float decay, float step, void* bf16_shadow,
                                void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(p && o_bf16, SPHIP_EINVAL, "muon_apply: null pointer");
    SPHIP_REQUIRE(n > 0 && ((uintptr_t)p & 15) == 0 && ((uintptr_t)o_bf16 & 7) == 0 && ((uintptr_t)bf16_shadow & 7) == 0,
                  SPHIP_EINVAL, "muon_apply: empty or misaligned tensor");
    hipLaunchKernelGGL(muon_apply_kernel, dim3(stream_blocks(n, MU_VEC * 4)), dim3(MU_THREADS), 0, (hipStream_t)stream, p,
                       (const bf16_t*)o_bf16, n, decay, step, (bf16_t*)bf16_shadow);
    SPHIP_CHECK_LAUNCH("muon_apply");
    return SPHIP_OK;
}
