// AuxK over the dead latents as dense products (sparse_coder.py:230-251).
//
// The AuxK decode sums k_aux = d_in/2 decoder rows per token, all of them rows of dead latents.
// When the dead latents are few (n_dead <= ~16 k_aux) the (tokens x dead latents) activation
// matrix is dense enough that the matrix cores beat the row gather by an order of magnitude:
//     e_hat  = A  W_dead  + b_dec        A  (N, Kd) bf16, zero where a dead latent was not selected
//     G_dead = A^T dL/de_hat             (Kd, D) rows of W_dec.grad
// The selected activations are bf16 under autocast, so A is exact. The fp32 factor (decoder
// rows, output gradient) is split into three bf16 terms (8 + 8 + 8 significand bits: exact),
// every bf16 x bf16 product is exact in the fp32 accumulator, and the three partial products
// are added in fp32 (sphip_gemm_bf16_f32acc): the result differs from the reference's fp32
// embedding_bag only by the order of the additions.
//
// This file holds the byte work around those products: compaction of the dead mask, the
// scatter of the selected activations into A and A^T, the split + transpose of an fp32 matrix
// into bf16 terms, and the scatter of the gradient rows back into the dense (M, D) gradient.
#include "common.hpp"

namespace sphip {

// ---------------------------------------------------------------- dead mask -> slots
// One workgroup: thread t owns a contiguous run of latents, counts its dead ones, an
// LDS scan gives the run's first slot. slot_of[m] = slot or -1; list[slot] = m.
__global__ __launch_bounds__(1024) void dead_compact_kernel(const uint8_t* __restrict__ mask, int M,
                                                            int* __restrict__ slot_of, int64_t* __restrict__ list,
                                                            int list_cap) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (M + 1023) / 1024;
    const int lo = min(t * per, M), hi = min(lo + per, M);
    int c = 0;
    for (int m = lo; m < hi; ++m) c += mask[m] != 0;
    part[t] = c;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int s = part[t] - c;
    for (int m = lo; m < hi; ++m) {
        if (mask[m] != 0) {
            slot_of[m] = s;
            if (s < list_cap) list[s] = m;
            ++s;
        } else {
            slot_of[m] = -1;
        }
    }
}

// ---------------------------------------------------------------- selected activations -> A, A^T
__global__ __launch_bounds__(256) void dead_scatter_kernel(const int64_t* __restrict__ idx,
                                                           const bf16_t* __restrict__ acts,
                                                           const int* __restrict__ slot_of, int64_t total, int k,
                                                           int M, int64_t ldA, int64_t ldT, bf16_t* __restrict__ A,
                                                           bf16_t* __restrict__ At) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t j = idx[i];
    if (j < 0 || j >= M) return;
    const int s = slot_of != nullptr ? slot_of[j] : (int)j;     // no map: the indices are slots already
    if (s < 0) return;
    const int64_t n = i / k;
    const bf16_t a = acts[i];
    A[n * ldA + s] = a;
    At[(int64_t)s * ldT + n] = a;
}

// ---------------------------------------------------------------- fp32 -> 3 bf16 terms, transposed
__device__ __forceinline__ bf16_t split_term(float& r) {
    const bf16_t h = f32_to_bf16(r);
    r -= bf16_to_f32(h);
    return h;
}

// src: fp32 rows of C columns, row r = list ? list[r] : r for r < R (rows R..ldr-1 of the output
// are zero padding). dst_p[c * ldr + r] = p-th bf16 term of src[row r][c]. 64 x 64 tiles
// through LDS: reads run along c, writes along r.
__global__ __launch_bounds__(256) void split_transpose_kernel(const float* __restrict__ src,
                                                              const int64_t* __restrict__ list, int64_t R, int C,
                                                              int64_t ldr, bf16_t* __restrict__ d0,
                                                              bf16_t* __restrict__ d1, bf16_t* __restrict__ d2) {
    __shared__ float tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int64_t r = r0 + i;
        float v = 0.f;
        if (r < R && c0 + tx < C) {
            const int64_t row = list != nullptr ? list[r] : r;
            v = src[row * C + c0 + tx];
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i;
        const int64_t r = r0 + tx;
        if (c >= C || r >= ldr) continue;
        float v = tile[tx][i];
        const int64_t o = (int64_t)c * ldr + r;
        d0[o] = split_term(v);
        d1[o] = split_term(v);
        d2[o] = split_term(v);
    }
}

// ---------------------------------------------------------------- gradient rows -> dense gradient
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src,
                                                           const int64_t* __restrict__ list, int64_t R, int D4,
                                                           float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * D4) return;
    const int64_t r = i / D4;
    const int c = (int)(i - r * D4);
    reinterpret_cast<float4*>(dst)[list[r] * D4 + c] = reinterpret_cast<const float4*>(src)[i];
}

// dst[list[r], :] += src[r, :] (the rows of `list` are distinct: plain read-modify-write)
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ src,
                                                               const int64_t* __restrict__ list, int64_t R, int D4,
                                                               float* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * D4) return;
    const int64_t r = i / D4;
    const int c = (int)(i - r * D4);
    float4* d = reinterpret_cast<float4*>(dst) + list[r] * D4 + c;
    const float4 a = *d, b = reinterpret_cast<const float4*>(src)[i];
    *d = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// dst row r = src row list[r] for r < R, zero for R <= r < rows_out; rows of `units` 2-byte
// elements (the bf16 shadow of the encoder rows / bias of the dead latents)
__global__ __launch_bounds__(256) void gather_rows_u16_kernel(const bf16_t* __restrict__ src,
                                                              const int64_t* __restrict__ list, int64_t R,
                                                              int64_t rows_out, int units, bf16_t* __restrict__ dst) {
    const int64_t r = blockIdx.x;
    if (r >= rows_out) return;
    bf16_t* d = dst + r * units;
    if (r < R) {
        const bf16_t* s = src + list[r] * units;
        if ((units & 7) == 0) {
            for (int c = threadIdx.x * 8; c < units; c += 256 * 8)
                *reinterpret_cast<uint4*>(d + c) = *reinterpret_cast<const uint4*>(s + c);
        } else {
            for (int c = threadIdx.x; c < units; c += 256) d[c] = s[c];
        }
    } else {
        for (int c = threadIdx.x; c < units; c += 256) d[c] = 0;
    }
}

// out[n][:] = bias (the fp32 products are then accumulated on top)
__global__ __launch_bounds__(256) void broadcast_rows_kernel(const float* __restrict__ bias, int64_t total4, int D4,
                                                             float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(bias)[i % D4];
}

int launch_gemm256_f32out(const void* x, const void* W, float* out, int64_t N, int64_t K, int64_t M, int64_t ld_out,
                          int accumulate, hipStream_t s);   // encoder_gemm256.hip

}  // namespace sphip

using namespace sphip;

extern "C" int sphip_dead_compact(const uint8_t* mask, int64_t M, int32_t* slot_of, int64_t* dead_list,
                                  int64_t list_capacity, void* stream) {
    SPHIP_REQUIRE(mask && slot_of && dead_list, SPHIP_EINVAL, "dead_compact: null pointer");
    SPHIP_REQUIRE(M > 0 && M < (1ll << 31) && list_capacity >= 0 && list_capacity < (1ll << 31), SPHIP_EINVAL,
                  "dead_compact: bad sizes M=%lld capacity=%lld", (long long)M, (long long)list_capacity);
    hipLaunchKernelGGL(dead_compact_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mask, (int)M, slot_of,
                       dead_list, (int)list_capacity);
    SPHIP_CHECK_LAUNCH("dead_compact");
    return SPHIP_OK;
}

extern "C" int sphip_dead_scatter_acts(const int64_t* idx, const void* acts_bf16, const int32_t* slot_of, int64_t N,
                                       int64_t k, int64_t M, int64_t Kd, int64_t ldA, int64_t ldT, void* A, void* At,
                                       void* stream) {
    SPHIP_REQUIRE(idx && acts_bf16 && A && At, SPHIP_EINVAL, "dead_scatter_acts: null pointer");
    SPHIP_REQUIRE(N > 0 && k > 0 && M > 0 && Kd > 0 && ldA >= Kd && ldT >= N && k < (1ll << 31) && M < (1ll << 31),
                  SPHIP_EINVAL, "dead_scatter_acts: bad sizes N=%lld k=%lld Kd=%lld ldA=%lld ldT=%lld", (long long)N,
                  (long long)k, (long long)Kd, (long long)ldA, (long long)ldT);
    hipStream_t s = (hipStream_t)stream;
    SPHIP_REQUIRE(hipMemsetAsync(A, 0, (size_t)(N * ldA * 2), s) == hipSuccess &&
                      hipMemsetAsync(At, 0, (size_t)(Kd * ldT * 2), s) == hipSuccess,
                  SPHIP_ELAUNCH, "dead_scatter_acts: memset failed");
    const int64_t total = N * k;
    SPHIP_REQUIRE(cdiv(total, 256) < (1ll << 31), SPHIP_EUNSUPPORTED, "dead_scatter_acts: too many entries");
    hipLaunchKernelGGL(dead_scatter_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, idx,
                       (const bf16_t*)acts_bf16, slot_of, total, (int)k, (int)M, ldA, ldT, (bf16_t*)A, (bf16_t*)At);
    SPHIP_CHECK_LAUNCH("dead_scatter_acts");
    return SPHIP_OK;
}

extern "C" int sphip_split_bf16_transposed(const float* src, const int64_t* row_list, int64_t R, int64_t C, int64_t ldr,
                                           void* hi, void* mid, void* lo, void* stream) {
    SPHIP_REQUIRE(src && hi && mid && lo, SPHIP_EINVAL, "split_bf16_transposed: null pointer");
    SPHIP_REQUIRE(R > 0 && C > 0 && ldr >= R && C < (1ll << 31), SPHIP_EINVAL,
                  "split_bf16_transposed: bad sizes R=%lld C=%lld ldr=%lld", (long long)R, (long long)C, (long long)ldr);
    SPHIP_REQUIRE(cdiv(ldr, 64) < (1ll << 31) && cdiv(C, 64) < 65536, SPHIP_EUNSUPPORTED,
                  "split_bf16_transposed: grid too large");
    hipLaunchKernelGGL(split_transpose_kernel, dim3((unsigned)cdiv(ldr, 64), (unsigned)cdiv(C, 64)), dim3(256), 0,
                       (hipStream_t)stream, src, row_list, R, (int)C, ldr, (bf16_t*)hi, (bf16_t*)mid, (bf16_t*)lo);
    SPHIP_CHECK_LAUNCH("split_bf16_transposed");
    return SPHIP_OK;
}

extern "C" int sphip_gemm_bf16_f32acc(const void* x, const void* W, float* out, int64_t N, int64_t K, int64_t M,
                                      int64_t ld_out, int accumulate, void* stream) {
    SPHIP_REQUIRE(x && W && out, SPHIP_EINVAL, "gemm_bf16_f32acc: null pointer");
    SPHIP_REQUIRE(N > 0 && K > 0 && M > 0 && N < (1ll << 31) && M < (1ll << 31) && ld_out >= M, SPHIP_EINVAL,
                  "gemm_bf16_f32acc: bad sizes N=%lld K=%lld M=%lld ld_out=%lld", (long long)N, (long long)K,
                  (long long)M, (long long)ld_out);
    SPHIP_REQUIRE(K % 64 == 0 && M % 4 == 0 && ld_out % 4 == 0, SPHIP_EUNSUPPORTED,
                  "gemm_bf16_f32acc: needs K %% 64 == 0, M %% 4 == 0, ld_out %% 4 == 0 (K=%lld M=%lld ld_out=%lld)",
                  (long long)K, (long long)M, (long long)ld_out);
    // the kernel addresses a 256-row operand tile with 32-bit byte offsets
    SPHIP_REQUIRE(K * 2 * 256 < (1ll << 31), SPHIP_EUNSUPPORTED, "gemm_bf16_f32acc: K=%lld too long", (long long)K);
    SPHIP_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)out & 15) == 0, SPHIP_EINVAL,
                  "gemm_bf16_f32acc: operands must be 16-byte aligned");
    return launch_gemm256_f32out(x, W, out, N, K, M, ld_out, accumulate, (hipStream_t)stream);
}

extern "C" int sphip_scatter_rows(const float* src, const int64_t* row_list, int64_t R, int64_t D, int64_t M, float* dst,
                                  void* stream) {
    SPHIP_REQUIRE(src && row_list && dst, SPHIP_EINVAL, "scatter_rows: null pointer");
    SPHIP_REQUIRE(R > 0 && D > 0 && M >= R && D % 4 == 0, SPHIP_EINVAL, "scatter_rows: bad sizes R=%lld D=%lld M=%lld",
                  (long long)R, (long long)D, (long long)M);
    hipStream_t s = (hipStream_t)stream;
    SPHIP_REQUIRE(hipMemsetAsync(dst, 0, (size_t)(M * D * 4), s) == hipSuccess, SPHIP_ELAUNCH, "scatter_rows: memset failed");
    const int64_t total = R * (D / 4);
    SPHIP_REQUIRE(cdiv(total, 256) < (1ll << 31), SPHIP_EUNSUPPORTED, "scatter_rows: too many elements");
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, src, row_list, R,
                       (int)(D / 4), dst);
    SPHIP_CHECK_LAUNCH("scatter_rows");
    return SPHIP_OK;
}

extern "C" int sphip_scatter_add_rows(const float* src, const int64_t* row_list, int64_t R, int64_t D, int64_t M,
                                      float* dst, void* stream) {
    SPHIP_REQUIRE(src && row_list && dst, SPHIP_EINVAL, "scatter_add_rows: null pointer");
    SPHIP_REQUIRE(R > 0 && D > 0 && M >= R && D % 4 == 0, SPHIP_EINVAL, "scatter_add_rows: bad sizes R=%lld D=%lld M=%lld",
                  (long long)R, (long long)D, (long long)M);
    SPHIP_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, SPHIP_EINVAL,
                  "scatter_add_rows: src and dst must be 16-byte aligned");
    const int64_t total = R * (D / 4);
    SPHIP_REQUIRE(cdiv(total, 256) < (1ll << 31), SPHIP_EUNSUPPORTED, "scatter_add_rows: too many elements");
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, src,
                       row_list, R, (int)(D / 4), dst);
    SPHIP_CHECK_LAUNCH("scatter_add_rows");
    return SPHIP_OK;
}

extern "C" int sphip_broadcast_rows(const float* bias, int64_t N, int64_t D, float* out, void* stream) {
    SPHIP_REQUIRE(bias && out, SPHIP_EINVAL, "broadcast_rows: null pointer");
    SPHIP_REQUIRE(N > 0 && D > 0 && D % 4 == 0, SPHIP_EINVAL, "broadcast_rows: bad sizes N=%lld D=%lld", (long long)N,
                  (long long)D);
    const int64_t total4 = N * (D / 4);
    SPHIP_REQUIRE(cdiv(total4, 256) < (1ll << 31), SPHIP_EUNSUPPORTED, "broadcast_rows: too many elements");
    hipLaunchKernelGGL(broadcast_rows_kernel, dim3((unsigned)cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, bias,
                       total4, (int)(D / 4), out);
    SPHIP_CHECK_LAUNCH("broadcast_rows");
    return SPHIP_OK;
}

extern "C" int sphip_gather_rows_bf16(const void* src, const int64_t* row_list, int64_t R, int64_t rows_out, int64_t D,
                                      void* dst, void* stream) {
    SPHIP_REQUIRE(src && row_list && dst, SPHIP_EINVAL, "gather_rows_bf16: null pointer");
    SPHIP_REQUIRE(R > 0 && rows_out >= R && rows_out < (1ll << 31) && D > 0 && D < (1ll << 31), SPHIP_EINVAL,
                  "gather_rows_bf16: bad sizes R=%lld rows_out=%lld D=%lld", (long long)R, (long long)rows_out, (long long)D);
    SPHIP_REQUIRE((D & 7) != 0 || ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0, SPHIP_EINVAL,
                  "gather_rows_bf16: rows of a multiple of 8 elements must be 16-byte aligned");
    hipLaunchKernelGGL(gather_rows_u16_kernel, dim3((unsigned)rows_out), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, row_list, R, rows_out, (int)D, (bf16_t*)dst);
    SPHIP_CHECK_LAUNCH("gather_rows_bf16");
    return SPHIP_OK;
}
