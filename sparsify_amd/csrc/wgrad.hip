// Latent-major segmented reduction: dense weight gradients without atomics.
//   gW[j,:]   = sum_{e in segment j} coeff[e] * rows[e / k, :]
//   gbias[j]  = sum_{e in segment j} coeff[e]
//   dcoeff[e] = < rows[e / k, :], W[j,:] >
// Reference: aggregate_gradient_for_embedding_k, sparsify/xformers.py:88-136 (W_dec.grad and
// the gradient of the top-k activations) and the index_add_ scatter of
// FusedEncoder.backward, sparsify/fused_encoder.py:67-84 (encoder.weight.grad / bias.grad),
// which materialises an (N*k, D) fp32 temporary and adds it with atomics.
//
// gfx950 design: HBM / Infinity-Cache bound row gather. One wave per latent: the wave keeps
// the latent's output row (and, for dcoeff, its W row) in registers -- up to 4096 columns,
// 16 float4 per lane -- and streams the segment's `rows` with 16 independent 16-B loads per
// lane in flight. Entry metadata (perm, coeff, token) is wave-uniform -> scalar loads.
// The dot product for dcoeff is a 6-step wave reduction, no LDS, no barriers. Every output
// row is written exactly once (zeros for empty segments), sums run in segment order.
#include "common.hpp"

namespace sphip {

template <typename RT, typename CT, int NV, bool DOT>
__global__ __launch_bounds__(256) void seg_wgrad_kernel(const unsigned* __restrict__ perm,
                                                        const int* __restrict__ seg_begin,
                                                        const CT* __restrict__ coeff,
                                                        const RT* __restrict__ rows,
                                                        const float* __restrict__ W, float* __restrict__ gW,
                                                        float* __restrict__ gbias, float* __restrict__ dcoeff,
                                                        int k, int M, int D) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= M) return;
    const int beg = seg_begin[j], end = seg_begin[j + 1];
    const int chunk_cols = 256 * NV;

    for (int c0 = 0; c0 < D; c0 += chunk_cols) {
        float4 acc[NV];
        float4 w[DOT ? NV : 1];
        bool live[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int col = c0 + (lane + 64 * v) * 4;
            live[v] = col < D;
            acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (DOT) w[v] = live[v] ? *reinterpret_cast<const float4*>(W + (int64_t)j * D + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float bsum = 0.f;
        for (int p = beg; p < end; ++p) {
            const unsigned e = perm[p];
            const float c = Elem<CT>::load(coeff, e);
            const RT* rp = rows + (int64_t)(e / (unsigned)k) * D + c0 + lane * 4;
            float4 r[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) r[v] = live[v] ? load4(rp + 256 * v) : make_float4(0.f, 0.f, 0.f, 0.f);
            float dot = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                acc[v].x = fmaf(c, r[v].x, acc[v].x);
                acc[v].y = fmaf(c, r[v].y, acc[v].y);
                acc[v].z = fmaf(c, r[v].z, acc[v].z);
                acc[v].w = fmaf(c, r[v].w, acc[v].w);
                if (DOT) {
                    dot = fmaf(r[v].x, w[v].x, dot);
                    dot = fmaf(r[v].y, w[v].y, dot);
                    dot = fmaf(r[v].z, w[v].z, dot);
                    dot = fmaf(r[v].w, w[v].w, dot);
                }
            }
            if (DOT) {
                dot = wave_sum(dot);
                if (lane == 0) {
                    if (c0 == 0) dcoeff[e] = dot;
                    else dcoeff[e] += dot;   // same lane wrote it in the previous column chunk
                }
            }
            bsum += c;
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int col = c0 + (lane + 64 * v) * 4;
            if (live[v]) *reinterpret_cast<float4*>(gW + (int64_t)j * D + col) = acc[v];
        }
        if (gbias != nullptr && c0 == 0 && lane == 0) gbias[j] = bsum;
    }
}

template <typename RT, typename CT, bool DOT>
static void launch_nv(const unsigned* perm, const int* seg, const void* coeff, const void* rows, const float* W,
                      float* gW, float* gbias, float* dcoeff, int k, int M, int D, hipStream_t s) {
    dim3 grid((unsigned)cdiv(M, 4)), block(256);
    if (D <= 256)
        hipLaunchKernelGGL((seg_wgrad_kernel<RT, CT, 1, DOT>), grid, block, 0, s, perm, seg, (const CT*)coeff, (const RT*)rows, W, gW, gbias, dcoeff, k, M, D);
    else if (D <= 1024)
        hipLaunchKernelGGL((seg_wgrad_kernel<RT, CT, 4, DOT>), grid, block, 0, s, perm, seg, (const CT*)coeff, (const RT*)rows, W, gW, gbias, dcoeff, k, M, D);
    else if (D <= 2048)
        hipLaunchKernelGGL((seg_wgrad_kernel<RT, CT, 8, DOT>), grid, block, 0, s, perm, seg, (const CT*)coeff, (const RT*)rows, W, gW, gbias, dcoeff, k, M, D);
    else
        hipLaunchKernelGGL((seg_wgrad_kernel<RT, CT, 16, DOT>), grid, block, 0, s, perm, seg, (const CT*)coeff, (const RT*)rows, W, gW, gbias, dcoeff, k, M, D);
}

template <typename RT, typename CT>
static void launch_dot(bool dot, const unsigned* perm, const int* seg, const void* coeff, const void* rows,
                       const float* W, float* gW, float* gbias, float* dcoeff, int k, int M, int D, hipStream_t s) {
    if (dot) launch_nv<RT, CT, true>(perm, seg, coeff, rows, W, gW, gbias, dcoeff, k, M, D, s);
    else launch_nv<RT, CT, false>(perm, seg, coeff, rows, W, gW, gbias, dcoeff, k, M, D, s);
}

}  // namespace sphip

extern "C" int sphip_segmented_weight_grad(const uint32_t* perm, const int32_t* seg_begin, const void* coeff,
                                           int coeff_dtype, const void* rows, int rows_dtype, const float* W,
                                           float* gW, float* gbias, float* dcoeff, int64_t N, int64_t k, int64_t M,
                                           int64_t D, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(valid_dtype(coeff_dtype) && valid_dtype(rows_dtype), SPHIP_EINVAL, "segmented_weight_grad: bad dtype");
    SPHIP_REQUIRE(perm && seg_begin && coeff && rows && gW, SPHIP_EINVAL, "segmented_weight_grad: null pointer");
    SPHIP_REQUIRE(dcoeff == nullptr || W != nullptr, SPHIP_EINVAL, "segmented_weight_grad: dcoeff needs W");
    SPHIP_REQUIRE(N > 0 && k > 0 && M > 0 && D > 0, SPHIP_EINVAL, "segmented_weight_grad: empty shape");
    SPHIP_REQUIRE(N * k < (1ll << 31) && M < (1ll << 31) - 4 && D < (1ll << 30), SPHIP_EUNSUPPORTED,
                  "segmented_weight_grad: shape too large");
    SPHIP_REQUIRE(D % 4 == 0, SPHIP_EUNSUPPORTED, "segmented_weight_grad: D=%lld must be a multiple of 4", (long long)D);
    SPHIP_REQUIRE(((uintptr_t)rows & 15) == 0 && ((uintptr_t)gW & 15) == 0 && ((uintptr_t)W & 15) == 0, SPHIP_EINVAL,
                  "segmented_weight_grad: rows/W/gW must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const bool dot = dcoeff != nullptr;
    const unsigned* pp = perm;
    const int* sp = seg_begin;
    if (rows_dtype == SPHIP_F32) {
        if (coeff_dtype == SPHIP_F32) launch_dot<float, float>(dot, pp, sp, coeff, rows, W, gW, gbias, dcoeff, (int)k, (int)M, (int)D, s);
        else launch_dot<float, bf16_t>(dot, pp, sp, coeff, rows, W, gW, gbias, dcoeff, (int)k, (int)M, (int)D, s);
    } else {
        if (coeff_dtype == SPHIP_F32) launch_dot<bf16_t, float>(dot, pp, sp, coeff, rows, W, gW, gbias, dcoeff, (int)k, (int)M, (int)D, s);
        else launch_dot<bf16_t, bf16_t>(dot, pp, sp, coeff, rows, W, gW, gbias, dcoeff, (int)k, (int)M, (int)D, s);
    }
    SPHIP_CHECK_LAUNCH("segmented_weight_grad");
    return SPHIP_OK;
}
