// Token-major sparse gather-accumulate:
//   out[n,:] = sum_s coeff[n,s] * table[idx[n,s],:]  (+ bias)
// Reference: decoder_impl / eager_decode (F.embedding_bag, mode="sum", per_sample_weights),
// sparsify/utils.py:81-90; Triton embedding_bag_k, sparsify/xformers.py:9-53; and the
// grad-input gather of FusedEncoder.backward, sparsify/fused_encoder.py:58-64.
//
// gfx950 design: HBM/Infinity-Cache bound. Two kernels: the column-sliced one below for
// tables beyond the Infinity Cache (the training shapes), and this one for everything
// else. One 256-thread workgroup per (token, 1024-column
// chunk); each lane owns one float4 column group and streams the k table rows with 8
// independent 16-B loads in flight (the slot indices/weights are wave-uniform and come
// through scalar loads), so a CU keeps several hundred KB of row reads outstanding.
// fp32 accumulation in slot order -> bitwise reproducible.
#include "common.hpp"
#include <stdlib.h>

namespace sphip {

// an index outside [0, M) marks an inactive slot (sparsify_hip.h)
__device__ __forceinline__ bool off(int64_t j, int64_t M) { return (unsigned long long)j >= (unsigned long long)M; }

template <typename CT, int VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(const int64_t* __restrict__ idx,
                                                          const CT* __restrict__ coeff,
                                                          const float* __restrict__ table,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ out, int k, int D, int64_t M) {
    const int64_t n = blockIdx.y;
    const int col = (blockIdx.x * 256 + threadIdx.x) * VEC;
    if (col >= D) return;
    const int64_t* ip = idx + n * k;
    const CT* cp = coeff + n * k;
    const float* tcol = table + col;

    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;

    constexpr int U = 8;
    int s = 0;
    for (; s + U <= k; s += U) {
        float w[U];
        const float* rp[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // a negative index marks an inactive slot (latent owned by another shard): weight
            // 0 and row 0 as a stand-in address (always cache-resident), so the unrolled load
            // batch stays branch-free
            const int64_t j = ip[s + u];
            rp[u] = tcol + (off(j, M) ? 0 : j) * (int64_t)D;
            w[u] = off(j, M) ? 0.f : Elem<CT>::load(cp, s + u);
        }
        if (VEC == 4) {
            float4 r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = *reinterpret_cast<const float4*>(rp[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc[0] = fmaf(w[u], r[u].x, acc[0]);
                acc[1 % VEC] = fmaf(w[u], r[u].y, acc[1 % VEC]);
                acc[2 % VEC] = fmaf(w[u], r[u].z, acc[2 % VEC]);
                acc[3 % VEC] = fmaf(w[u], r[u].w, acc[3 % VEC]);
            }
        } else {
            float r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = *rp[u];
#pragma unroll
            for (int u = 0; u < U; ++u) acc[0] = fmaf(w[u], r[u], acc[0]);
        }
    }
    for (; s < k; ++s) {
        if (off(ip[s], M)) continue;
        const float* rp = tcol + ip[s] * (int64_t)D;
        const float w = Elem<CT>::load(cp, s);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = fmaf(w, rp[v], acc[v]);
    }
    if (bias != nullptr) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += bias[col + v];
    }
    float* o = out + n * (int64_t)D + col;
    if (VEC == 4) *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1 % VEC], acc[2 % VEC], acc[3 % VEC]);
    else o[0] = acc[0];
}

// Column-sliced variant for tables larger than the 256 MiB Infinity Cache: one wave per
// (token, SLICE-column slice), slices are the slow grid dimension, so at any time the whole
// chip gathers from the same M x SLICE*4 B column slice of the table (128 MiB at M = 131072,
// SLICE = 256), which stays resident in the Infinity Cache while all N tokens use it: the
// table then comes from HBM once per launch instead of N*k/M times.
template <typename CT>
__global__ __launch_bounds__(256) void gather_rows_sliced_kernel(const int64_t* __restrict__ idx,
                                                                 const CT* __restrict__ coeff,
                                                                 const float* __restrict__ table,
                                                                 const float* __restrict__ bias,
                                                                 float* __restrict__ out, int N, int k, int D, int64_t M) {
    // one wave per token, so the index / weight loads stay scalar (half a wave per token with
    // 128-column slices measured 5 % slower)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n = (int64_t)blockIdx.x * 4 + wave;
    if (n >= N) return;
    const int col = blockIdx.y * 256 + (threadIdx.x & 63) * 4;
    const int64_t* ip = idx + n * k;
    const CT* cp = coeff + n * k;
    const float* tcol = table + col;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int U = 8;
    int s = 0;
    for (; s + U <= k; s += U) {
        float w[U];
        const float* rp[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = ip[s + u];
            rp[u] = tcol + (off(j, M) ? 0 : j) * (int64_t)D;
            w[u] = off(j, M) ? 0.f : Elem<CT>::load(cp, s + u);
        }
        float4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = *reinterpret_cast<const float4*>(rp[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc.x = fmaf(w[u], r[u].x, acc.x);
            acc.y = fmaf(w[u], r[u].y, acc.y);
            acc.z = fmaf(w[u], r[u].z, acc.z);
            acc.w = fmaf(w[u], r[u].w, acc.w);
        }
    }
    for (; s < k; ++s) {
        const int64_t j = ip[s];
        if (off(j, M)) continue;
        const float4 r = *reinterpret_cast<const float4*>(tcol + j * (int64_t)D);
        const float w = Elem<CT>::load(cp, s);
        acc.x = fmaf(w, r.x, acc.x);
        acc.y = fmaf(w, r.y, acc.y);
        acc.z = fmaf(w, r.z, acc.z);
        acc.w = fmaf(w, r.w, acc.w);
    }
    if (bias != nullptr) {
        const float4 b = *reinterpret_cast<const float4*>(bias + col);
        acc.x += b.x, acc.y += b.y, acc.z += b.z, acc.w += b.w;
    }
    // written once, read by a later kernel: non-temporal, the table slice keeps the caches
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 t = {acc.x, acc.y, acc.z, acc.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f4*>(out + n * (int64_t)D + col));
}

// Indices outside [0, M) are inactive slots and read nothing (sparsify_hip.h).


}  // namespace sphip

extern "C" int sphip_gather_rows_weighted(const int64_t* idx, const void* coeff, int coeff_dtype, const float* table,
                                          const float* bias, float* out, int64_t N, int64_t k, int64_t M, int64_t D,
                                          void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(valid_dtype(coeff_dtype), SPHIP_EINVAL, "gather_rows_weighted: bad coeff dtype %d", coeff_dtype);
    SPHIP_REQUIRE(idx && coeff && table && out, SPHIP_EINVAL, "gather_rows_weighted: null pointer");
    SPHIP_REQUIRE(N > 0 && k > 0 && M > 0 && D > 0, SPHIP_EINVAL, "gather_rows_weighted: empty shape");
    SPHIP_REQUIRE(N < 65536ll * 32768 && D < (1ll << 31) && k < (1ll << 31), SPHIP_EUNSUPPORTED,
                  "gather_rows_weighted: shape too large");
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (D % 4 == 0) && (((uintptr_t)table & 15) == 0) && (((uintptr_t)out & 15) == 0);
    // SPHIP_GATHER=full|sliced forces a kernel (tests); the sliced one pays when the table
    // does not fit the Infinity Cache but one 1-KiB column slice of it does
    const char* force = getenv("SPHIP_GATHER");
    bool sliced = vec && D % 256 == 0 && (bias == nullptr || ((uintptr_t)bias & 15) == 0) && cdiv(N, 4) < (1ll << 31) &&
                  D / 256 <= 65535 && M * D * 4 > (200ll << 20) && M * 1024 <= (160ll << 20) && N * k >= 8 * M;
    if (force && force[0] == 'f') sliced = false;
    if (force && force[0] == 's' && vec && D % 256 == 0 && D / 256 <= 65535) sliced = true;
    if (sliced) {
        dim3 grid((unsigned)cdiv(N, 4), (unsigned)(D / 256)), block(256);
        if (coeff_dtype == SPHIP_F32)
            hipLaunchKernelGGL((gather_rows_sliced_kernel<float>), grid, block, 0, s, idx, (const float*)coeff, table, bias, out, (int)N, (int)k, (int)D, M);
        else
            hipLaunchKernelGGL((gather_rows_sliced_kernel<bf16_t>), grid, block, 0, s, idx, (const bf16_t*)coeff, table, bias, out, (int)N, (int)k, (int)D, M);
        SPHIP_CHECK_LAUNCH("gather_rows_weighted(sliced)");
        return SPHIP_OK;
    }
    // blockIdx.y carries the token; split very large N over several launches (grid.y <= 65535)
    const int64_t maxy = 65535;
    for (int64_t n0 = 0; n0 < N; n0 += maxy) {
        const int64_t nn = (N - n0 < maxy) ? (N - n0) : maxy;
        const int64_t* ip = idx + n0 * k;
        float* op = out + n0 * D;
        if (vec) {
            dim3 grid((unsigned)cdiv(D, 1024), (unsigned)nn), block(256);
            if (coeff_dtype == SPHIP_F32)
                hipLaunchKernelGGL((gather_rows_kernel<float, 4>), grid, block, 0, s, ip, (const float*)coeff + n0 * k, table, bias, op, (int)k, (int)D, M);
            else
                hipLaunchKernelGGL((gather_rows_kernel<bf16_t, 4>), grid, block, 0, s, ip, (const bf16_t*)coeff + n0 * k, table, bias, op, (int)k, (int)D, M);
        } else {
            dim3 grid((unsigned)cdiv(D, 256), (unsigned)nn), block(256);
            if (coeff_dtype == SPHIP_F32)
                hipLaunchKernelGGL((gather_rows_kernel<float, 1>), grid, block, 0, s, ip, (const float*)coeff + n0 * k, table, bias, op, (int)k, (int)D, M);
            else
                hipLaunchKernelGGL((gather_rows_kernel<bf16_t, 1>), grid, block, 0, s, ip, (const bf16_t*)coeff + n0 * k, table, bias, op, (int)k, (int)D, M);
        }
        SPHIP_CHECK_LAUNCH("gather_rows_weighted");
    }
    return SPHIP_OK;
}
