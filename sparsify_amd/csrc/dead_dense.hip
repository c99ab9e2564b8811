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
#include <stdlib.h>

namespace sphip {

// ---------------------------------------------------------------- dead mask -> slots
// One workgroup: thread t owns a contiguous run of 16-latent chunks (16 mask bytes per load), counts
// its dead ones, a wave scan plus the 16 wave totals give the run's first slot, and a second pass
// over the (cache-resident) mask writes slot_of[m] = slot or -1, 16 at a time, and list[slot] = m.
__device__ __forceinline__ uint4 mask_chunk(const uint8_t* __restrict__ mask, int m0, int M, bool aligned) {
    if (aligned && m0 + 16 <= M) return *reinterpret_cast<const uint4*>(mask + m0);
    unsigned w[4] = {0u, 0u, 0u, 0u};
    for (int b = 0; b < 16 && m0 + b < M; ++b) w[b >> 2] |= (mask[m0 + b] != 0 ? 1u : 0u) << (8 * (b & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ int nonzero_bytes(unsigned w) {
    w |= w >> 4;
    w |= w >> 2;
    w |= w >> 1;
    return __popc(w & 0x01010101u);
}

__global__ __launch_bounds__(1024) void dead_compact_kernel(const uint8_t* __restrict__ mask, int M,
                                                            int* __restrict__ slot_of, int64_t* __restrict__ list,
                                                            int list_cap) {
    __shared__ int wave_tot[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool aligned = ((uintptr_t)mask & 15) == 0;
    const bool aligned_out = ((uintptr_t)slot_of & 15) == 0;
    const int nchunks = (M + 15) / 16;
    const int per = (nchunks + 1023) / 1024;
    const int c_lo = min(t * per, nchunks), c_hi = min(c_lo + per, nchunks);
    int c = 0;
    for (int ch = c_lo; ch < c_hi; ++ch) {
        const uint4 v = mask_chunk(mask, ch * 16, M, aligned);
        c += nonzero_bytes(v.x) + nonzero_bytes(v.y) + nonzero_bytes(v.z) + nonzero_bytes(v.w);
    }
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int n = __shfl_up(incl, o, 64);
        if (lane >= o) incl += n;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int s = incl - c;
    for (int w = 0; w < wave; ++w) s += wave_tot[w];
    for (int ch = c_lo; ch < c_hi; ++ch) {
        const int m0 = ch * 16;
        const uint4 v = mask_chunk(mask, m0, M, aligned);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        int out[16];
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const bool dead = ((w[b >> 2] >> (8 * (b & 3))) & 0xffu) != 0u;
            out[b] = dead ? s : -1;
            if (dead) {
                if (s < list_cap) list[s] = m0 + b;
                ++s;
            }
        }
        if (aligned_out && m0 + 16 <= M) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<int4*>(slot_of + m0 + 4 * q) = make_int4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
        } else {
#pragma unroll
            for (int b = 0; b < 16; ++b)
                if (m0 + b < M) slot_of[m0 + b] = out[b];
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

// The same matrices without 2-byte scatters into global memory: a workgroup composes one token's row
// of A in LDS (zero, scatter, one coalesced copy out; columns Kd .. ldA - 1 are zeroed too) ...
__global__ __launch_bounds__(256) void dead_rows_kernel(const int64_t* __restrict__ idx, const bf16_t* __restrict__ acts,
                                                        const int* __restrict__ slot_of, int k, int M, int Kd,
                                                        int64_t ldA, bf16_t* __restrict__ A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
    bf16_t* rowbuf = reinterpret_cast<bf16_t*>(dyn_smem);
    const int64_t n = blockIdx.x;
    const int chunks = (int)(ldA / 8);                                  // 16-byte chunks of the row, padding included
    for (int c = threadIdx.x; c < chunks; c += 256) reinterpret_cast<uint4*>(rowbuf)[c] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    for (int e = threadIdx.x; e < k; e += 256) {
        const int64_t j = idx[n * k + e];
        if (j < 0 || j >= M) continue;
        const int s = slot_of != nullptr ? slot_of[j] : (int)j;
        if (s >= 0 && s < Kd) rowbuf[s] = acts[n * k + e];
    }
    __syncthreads();
    uint4* dst = reinterpret_cast<uint4*>(A + n * ldA);
    for (int c = threadIdx.x; c < chunks; c += 256) dst[c] = reinterpret_cast<const uint4*>(rowbuf)[c];
}

// ... and A^T is its transpose, 64 tokens x 128 slots per workgroup through LDS: reads run along the
// slots (4 bytes per lane), writes along the tokens (4 bytes per lane, 128-byte row pieces). Tokens
// N .. ldT - 1 (padding of A^T's rows) are written as zeros.
__global__ __launch_bounds__(256) void transpose_u16_kernel(const bf16_t* __restrict__ A, int64_t N, int Kd, int64_t ldA,
                                                            int64_t ldT, bf16_t* __restrict__ At) {
    __shared__ bf16_t tile[64][130];
    const int64_t n0 = (int64_t)blockIdx.x * 64;
    const int s0 = blockIdx.y * 128;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int64_t n = n0 + i;
        const int sc = s0 + 2 * tx;
        unsigned v = 0u;
        if (n < N && sc < Kd) v = *reinterpret_cast<const unsigned*>(A + n * ldA + sc);     // Kd, ldA even
        tile[i][2 * tx] = (bf16_t)(v & 0xffffu);
        tile[i][2 * tx + 1] = (bf16_t)(v >> 16);
    }
    __syncthreads();
    const int l = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    for (int r = r0; r < 128; r += 8) {
        const int sr = s0 + r;
        const int64_t n = n0 + 2 * l;
        if (sr >= Kd || n >= ldT) continue;
        const unsigned v = (unsigned)tile[2 * l][r] | ((unsigned)tile[2 * l + 1][r] << 16);
        *reinterpret_cast<unsigned*>(At + (int64_t)sr * ldT + n) = v;                          // ldT even
    }
}

// ---------------------------------------------------------------- fp32 -> 3 bf16 terms, transposed
__device__ __forceinline__ bf16_t split_term(float& r) {
    const bf16_t h = f32_to_bf16(r);
    r -= bf16_to_f32(h);
    return h;
}

// src: fp32 rows of C columns, row r = list ? list[r] : r for r < R (rows R..ldr-1 of the output
// are zero padding). dst_p[c * ldr + r] = p-th bf16 term of src[row r][c]. 64 x 64 tiles
// through LDS: reads run along c, writes along r. (A 128 x 64 tile form writing two rows per lane,
// 4-byte stores, measured slower: 0.56 against 0.39 ms for the two calls of the AuxK step.)
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
    SPHIP_REQUIRE(mask && slot_of && (dead_list || list_capacity == 0), SPHIP_EINVAL, "dead_compact: null pointer");
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
    // rows composed in LDS + a tiled transpose (0.81 -> see DESIGN) when a row fits 64 KiB of LDS
    const bool composed = ldA * 2 <= 65536 && ldA % 8 == 0 && Kd % 2 == 0 && ldT % 2 == 0 && N < (1ll << 31) &&
                          (((uintptr_t)A | (uintptr_t)At) & 15) == 0 && cdiv(Kd, 128) < 65536 &&
                          getenv("SPHIP_DEAD_SCATTER") == nullptr;
    if (composed) {
        hipLaunchKernelGGL(dead_rows_kernel, dim3((unsigned)N), dim3(256), (size_t)(ldA * 2), s, idx, (const bf16_t*)acts_bf16,
                           slot_of, (int)k, (int)M, (int)Kd, ldA, (bf16_t*)A);
        SPHIP_CHECK_LAUNCH("dead_scatter_acts(rows)");
        hipLaunchKernelGGL(transpose_u16_kernel, dim3((unsigned)cdiv(ldT, 64), (unsigned)cdiv(Kd, 128)), dim3(256), 0, s,
                           (const bf16_t*)A, N, (int)Kd, ldA, ldT, (bf16_t*)At);
        SPHIP_CHECK_LAUNCH("dead_scatter_acts(transpose)");
        return SPHIP_OK;
    }
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
