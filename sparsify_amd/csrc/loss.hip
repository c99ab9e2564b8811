// FVU / AuxK loss pieces and the column reductions around them.
// Reference: SparseCoder.forward, sparsify/sparse_coder.py:225-228 (e = y - sae_out,
// total_variance = (y - y.mean(0)).pow(2).sum()), :248-254 (auxk / l2 / fvu), :260
// (multi-topk fvu); `+ self.b_dec` backward (column sum), :202; trainer.py:399-404
// (step-0 means).
//
// gfx950 design: all HBM-bound streaming kernels, float4 (or 4 x bf16) per lane. Every
// reduction is two-stage with a fixed partition (row chunks -> partials -> one final
// workgroup), no atomics, so losses and bias gradients are bitwise reproducible and never
// touch the host.
#include "common.hpp"

namespace sphip {

constexpr int RED_ROWS = 256;     // row chunks for column reductions
constexpr int RED_BLOCKS = 1024;  // workgroups for flat reductions

struct RedPlan {
    size_t off_partial, off_mean, off_scalar, total;
};
static RedPlan red_plan(int64_t N, int64_t D) {
    RedPlan p;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    p.off_partial = take((size_t)RED_ROWS * (size_t)D * 4);
    p.off_mean = take((size_t)D * 4);
    p.off_scalar = take((size_t)(RED_ROWS * cdiv(D, 256) > RED_BLOCKS ? RED_ROWS * cdiv(D, 256) : RED_BLOCKS) * 4);
    p.total = o;
    (void)N;
    return p;
}

// partial[r, col] = sum over the rows of chunk r of (w[row] *) a[row, col]
template <typename T, bool WEIGHTED, int VEC>
__global__ __launch_bounds__(256) void col_partial_kernel(const T* __restrict__ a, const float* __restrict__ w,
                                                          int64_t N, int D, int rows_per, float* __restrict__ partial) {
    const int col = (blockIdx.x * 256 + threadIdx.x) * VEC;
    if (col >= D) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per;
    const int64_t r1 = (r0 + rows_per < N) ? r0 + rows_per : N;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    const T* p = a + col;
    constexpr int U = 8;
    int64_t r = r0;
    for (; r + U <= r1; r += U) {
        if (VEC == 4) {
            float4 x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = load4(p + (r + u) * D);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float s = WEIGHTED ? w[r + u] : 1.f;
                acc[0] = fmaf(s, x[u].x, acc[0]);
                acc[1 % VEC] = fmaf(s, x[u].y, acc[1 % VEC]);
                acc[2 % VEC] = fmaf(s, x[u].z, acc[2 % VEC]);
                acc[3 % VEC] = fmaf(s, x[u].w, acc[3 % VEC]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) acc[0] = fmaf(WEIGHTED ? w[r + u] : 1.f, Elem<T>::load(p, (r + u) * D), acc[0]);
        }
    }
    for (; r < r1; ++r) {
        const float s = WEIGHTED ? w[r] : 1.f;
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = fmaf(s, Elem<T>::load(p, r * D + v), acc[v]);
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) partial[(int64_t)blockIdx.y * D + col + v] = acc[v];
}

__global__ __launch_bounds__(256) void col_final_kernel(const float* __restrict__ partial, int nchunks, int D,
                                                        float scale, float* __restrict__ out) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= D) return;
    // 16 loads in flight, added in chunk order (same sum as the plain loop)
    float s = 0.f;
    int r = 0;
    for (; r + 16 <= nchunks; r += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = partial[(int64_t)(r + u) * D + col];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; r < nchunks; ++r) s += partial[(int64_t)r * D + col];
    out[col] = s * scale;
}

// scalar partial per workgroup: sum over the chunk of (y - mean[col])^2
template <typename T, int VEC>
__global__ __launch_bounds__(256) void tv_partial_kernel(const T* __restrict__ y, const float* __restrict__ mean,
                                                         int64_t N, int D, int rows_per,
                                                         float* __restrict__ partials) {
    __shared__ float red[16];
    const int col = (blockIdx.x * 256 + threadIdx.x) * VEC;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per;
    const int64_t r1 = (r0 + rows_per < N) ? r0 + rows_per : N;
    float acc = 0.f;
    if (col < D) {
        // bf16 activations: the reference computes y.mean(0) and y - mean in bf16 and only the
        // square and the sum in fp32 (autocast fp32 policy for pow / sum,
        // sparse_coder.py:205-209,228); reproduce those two roundings.
        constexpr bool LOW = sizeof(T) == 2;
        float m[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) m[v] = LOW ? bf16_to_f32(f32_to_bf16(mean[col + v])) : mean[col + v];
        for (int64_t r = r0; r < r1; ++r) {
            if (VEC == 4) {
                const float4 x = load4(y + r * D + col);
                float d0 = x.x - m[0], d1 = x.y - m[1 % VEC], d2 = x.z - m[2 % VEC], d3 = x.w - m[3 % VEC];
                if (LOW) {
                    d0 = bf16_to_f32(f32_to_bf16(d0)); d1 = bf16_to_f32(f32_to_bf16(d1));
                    d2 = bf16_to_f32(f32_to_bf16(d2)); d3 = bf16_to_f32(f32_to_bf16(d3));
                }
                acc = fmaf(d0, d0, acc); acc = fmaf(d1, d1, acc); acc = fmaf(d2, d2, acc); acc = fmaf(d3, d3, acc);
            } else {
                float d = Elem<T>::load(y, r * D + col) - m[0];
                if (LOW) d = bf16_to_f32(f32_to_bf16(d));
                acc = fmaf(d, d, acc);
            }
        }
    }
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void scalar_final_kernel(const float* __restrict__ partials, int n,
                                                           float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
    const float tot = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = tot;
}

template <typename T>
__global__ __launch_bounds__(256) void residual_kernel(const T* __restrict__ a, const float* __restrict__ b,
                                                       float* __restrict__ e, int64_t n,
                                                       float* __restrict__ partials) {
    __shared__ float red[16];
    float acc = 0.f;
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 x = load4(a + i * 4);
        const float4 y = *reinterpret_cast<const float4*>(b + i * 4);
        const float4 d = make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w);
        if (e != nullptr) *reinterpret_cast<float4*>(e + i * 4) = d;
        acc = fmaf(d.x, d.x, acc); acc = fmaf(d.y, d.y, acc); acc = fmaf(d.z, d.z, acc); acc = fmaf(d.w, d.w, acc);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        const float d = Elem<T>::load(a, i) - b[i];
        if (e != nullptr) e[i] = d;
        acc = fmaf(d, d, acc);
    }
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void scale_ratio_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          int64_t n, float alpha, const float* __restrict__ num,
                                                          const float* __restrict__ den) {
    const float f = alpha * (num ? num[0] : 1.f) / (den ? den[0] : 1.f);
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 x = *reinterpret_cast<const float4*>(in + i * 4);
        x.x *= f; x.y *= f; x.z *= f; x.w *= f;
        *reinterpret_cast<float4*>(out + i * 4) = x;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        out[i] = in[i] * f;
    }
}

static inline int rows_per_chunk(int64_t N) { return (int)cdiv(N, RED_ROWS); }
static inline int num_chunks(int64_t N) { return (int)cdiv(N, rows_per_chunk(N)); }

template <bool WEIGHTED>
static void launch_col_partial(const void* a, int dt, const float* w, int64_t N, int64_t D, float* partial,
                               hipStream_t s) {
    const int rp = rows_per_chunk(N), nc = num_chunks(N);
    const bool vec = (D % 4 == 0) && (((uintptr_t)a & 15) == 0);
    if (vec) {
        dim3 grid((unsigned)cdiv(D, 1024), (unsigned)nc), block(256);
        if (dt == SPHIP_F32) hipLaunchKernelGGL((col_partial_kernel<float, WEIGHTED, 4>), grid, block, 0, s, (const float*)a, w, N, (int)D, rp, partial);
        else hipLaunchKernelGGL((col_partial_kernel<bf16_t, WEIGHTED, 4>), grid, block, 0, s, (const bf16_t*)a, w, N, (int)D, rp, partial);
    } else {
        dim3 grid((unsigned)cdiv(D, 256), (unsigned)nc), block(256);
        if (dt == SPHIP_F32) hipLaunchKernelGGL((col_partial_kernel<float, WEIGHTED, 1>), grid, block, 0, s, (const float*)a, w, N, (int)D, rp, partial);
        else hipLaunchKernelGGL((col_partial_kernel<bf16_t, WEIGHTED, 1>), grid, block, 0, s, (const bf16_t*)a, w, N, (int)D, rp, partial);
    }
}

}  // namespace sphip

extern "C" size_t sphip_reduce_workspace_bytes(int64_t N, int64_t D) {
    if (N <= 0 || D <= 0) return 0;
    return sphip::red_plan(N, D).total;
}

#define SPHIP_WS_CHECK(name)                                                                              \
    const RedPlan plan = red_plan(N, D);                                                                  \
    SPHIP_REQUIRE(workspace != nullptr && workspace_bytes >= plan.total, SPHIP_EWORKSPACE,                \
                  name ": workspace %zu < %zu bytes", workspace_bytes, plan.total);                       \
    SPHIP_REQUIRE(((uintptr_t)workspace & 255) == 0, SPHIP_EINVAL, name ": workspace must be 256-byte aligned"); \
    char* ws = (char*)workspace;                                                                          \
    float* partial = (float*)(ws + plan.off_partial);                                                     \
    float* meanbuf = (float*)(ws + plan.off_mean);                                                        \
    float* scal = (float*)(ws + plan.off_scalar);                                                         \
    (void)partial; (void)meanbuf; (void)scal;

extern "C" int sphip_colsum(const void* a, int a_dtype, int64_t N, int64_t D, float scale, float* out,
                            void* workspace, size_t workspace_bytes, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(valid_dtype(a_dtype), SPHIP_EINVAL, "colsum: bad dtype");
    SPHIP_REQUIRE(a && out, SPHIP_EINVAL, "colsum: null pointer");
    SPHIP_REQUIRE(N > 0 && D > 0 && D < (1ll << 31), SPHIP_EINVAL, "colsum: bad shape");
    SPHIP_WS_CHECK("colsum");
    hipStream_t s = (hipStream_t)stream;
    launch_col_partial<false>(a, a_dtype, nullptr, N, D, partial, s);
    hipLaunchKernelGGL(col_final_kernel, dim3((unsigned)cdiv(D, 256)), dim3(256), 0, s, partial, num_chunks(N), (int)D, scale, out);
    SPHIP_CHECK_LAUNCH("colsum");
    return SPHIP_OK;
}

extern "C" int sphip_vec_mat(const float* v, const float* W, int64_t M, int64_t D, float scale, float* out,
                             void* workspace, size_t workspace_bytes, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(v && W && out, SPHIP_EINVAL, "vec_mat: null pointer");
    SPHIP_REQUIRE(M > 0 && D > 0 && D < (1ll << 31), SPHIP_EINVAL, "vec_mat: bad shape");
    const int64_t N = M;
    SPHIP_WS_CHECK("vec_mat");
    hipStream_t s = (hipStream_t)stream;
    launch_col_partial<true>(W, SPHIP_F32, v, M, D, partial, s);
    hipLaunchKernelGGL(col_final_kernel, dim3((unsigned)cdiv(D, 256)), dim3(256), 0, s, partial, num_chunks(M), (int)D, scale, out);
    SPHIP_CHECK_LAUNCH("vec_mat");
    return SPHIP_OK;
}

static int total_variance_impl(const void* y, int y_dtype, int64_t N, int64_t D, float* tv, void* workspace,
                               size_t workspace_bytes, hipStream_t s) {
    using namespace sphip;
    SPHIP_REQUIRE(valid_dtype(y_dtype), SPHIP_EINVAL, "total_variance: bad dtype");
    SPHIP_REQUIRE(y && tv, SPHIP_EINVAL, "total_variance: null pointer");
    SPHIP_REQUIRE(N > 0 && D > 0 && D < (1ll << 31), SPHIP_EINVAL, "total_variance: bad shape");
    SPHIP_WS_CHECK("total_variance");
    launch_col_partial<false>(y, y_dtype, nullptr, N, D, partial, s);
    hipLaunchKernelGGL(col_final_kernel, dim3((unsigned)cdiv(D, 256)), dim3(256), 0, s, partial, num_chunks(N), (int)D, 1.0f / (float)N, meanbuf);
    const int rp = rows_per_chunk(N), nc = num_chunks(N);
    const bool vec = (D % 4 == 0) && (((uintptr_t)y & 15) == 0);
    const unsigned gx = (unsigned)cdiv(D, vec ? 1024 : 256);
    dim3 grid(gx, (unsigned)nc), block(256);
    if (vec) {
        if (y_dtype == SPHIP_F32) hipLaunchKernelGGL((tv_partial_kernel<float, 4>), grid, block, 0, s, (const float*)y, meanbuf, N, (int)D, rp, scal);
        else hipLaunchKernelGGL((tv_partial_kernel<bf16_t, 4>), grid, block, 0, s, (const bf16_t*)y, meanbuf, N, (int)D, rp, scal);
    } else {
        if (y_dtype == SPHIP_F32) hipLaunchKernelGGL((tv_partial_kernel<float, 1>), grid, block, 0, s, (const float*)y, meanbuf, N, (int)D, rp, scal);
        else hipLaunchKernelGGL((tv_partial_kernel<bf16_t, 1>), grid, block, 0, s, (const bf16_t*)y, meanbuf, N, (int)D, rp, scal);
    }
    hipLaunchKernelGGL(scalar_final_kernel, dim3(1), dim3(256), 0, s, scal, (int)(gx * nc), tv);
    SPHIP_CHECK_LAUNCH("total_variance");
    return SPHIP_OK;
}

extern "C" int sphip_total_variance(const void* y, int y_dtype, int64_t N, int64_t D, float* tv, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    return total_variance_impl(y, y_dtype, N, D, tv, workspace, workspace_bytes, (hipStream_t)stream);
}

// ---- the allocation-free form: the partial sums live in a file-scope device array instead of a
// caller-provided workspace. Every GPU the library runs on holds its own copy of the array (at its
// own address) until the process ends; the host-side name `tv_scratch` is only a handle for
// hipGetSymbolAddress, the kernels get the resolved per-device pointer like any other workspace.
namespace sphip {
constexpr int64_t TV_STATIC_MAX_D = 8192;
// red_plan(N, 8192): 256 x 8192 x 4 (column partials) + 8192 x 4 (mean) + 8192 x 4 (scalar partials)
constexpr size_t TV_STATIC_BYTES = (size_t)RED_ROWS * TV_STATIC_MAX_D * 4 + 2 * (size_t)TV_STATIC_MAX_D * 4;
}  // namespace sphip
static __device__ __align__(256) unsigned char tv_scratch[sphip::TV_STATIC_BYTES];

#include <mutex>
extern "C" int sphip_total_variance_static(const void* y, int y_dtype, int64_t N, int64_t D, float* tv, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(D > 0 && D <= TV_STATIC_MAX_D, SPHIP_EUNSUPPORTED,
                  "total_variance_static: D=%lld exceeds the built-in scratch (D <= %lld); use sphip_total_variance "
                  "with a caller workspace", (long long)D, (long long)TV_STATIC_MAX_D);
    SPHIP_REQUIRE(N > 0 && red_plan(N, D).total <= TV_STATIC_BYTES, SPHIP_EUNSUPPORTED,
                  "total_variance_static: shape N=%lld D=%lld does not fit the built-in scratch", (long long)N, (long long)D);
    constexpr int MAX_DEV = 64;
    static void* dev_ptr[MAX_DEV] = {nullptr};
    static std::mutex mu;
    int dev = -1;
    SPHIP_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_DEV, SPHIP_ELAUNCH,
                  "total_variance_static: no current device");
    // the scratch used is the CURRENT device's copy: the data and the stream must live there too (another
    // GPU's kernels would be handed this one's array: peer access at best, a fault otherwise)
    hipPointerAttribute_t attr;
    SPHIP_REQUIRE(y != nullptr && hipPointerGetAttributes(&attr, y) == hipSuccess && attr.device == dev, SPHIP_EINVAL,
                  "total_variance_static: y is not memory of the current device %d (hipSetDevice to the data's GPU first)", dev);
    if (stream != nullptr) {
        hipDevice_t sdev = -1;
        SPHIP_REQUIRE(hipStreamGetDevice((hipStream_t)stream, &sdev) == hipSuccess && (int)sdev == dev, SPHIP_EINVAL,
                      "total_variance_static: the stream belongs to device %d, the current device is %d", (int)sdev, dev);
    }
    void* ptr;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (dev_ptr[dev] == nullptr) {
            // once per device, with that device current: the address of THIS device's copy
            void* p = nullptr;
            const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(tv_scratch));
            SPHIP_REQUIRE(e == hipSuccess && p != nullptr, SPHIP_ELAUNCH, "total_variance_static: hipGetSymbolAddress: %s",
                          hipGetErrorString(e));
            dev_ptr[dev] = p;
        }
        ptr = dev_ptr[dev];
    }
    return total_variance_impl(y, y_dtype, N, D, tv, ptr, TV_STATIC_BYTES, (hipStream_t)stream);
}

extern "C" int sphip_residual_sumsq(const void* a, int a_dtype, const float* b, float* e, float* sq, int64_t N,
                                    int64_t D, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(valid_dtype(a_dtype), SPHIP_EINVAL, "residual_sumsq: bad dtype");
    SPHIP_REQUIRE(a && b && sq, SPHIP_EINVAL, "residual_sumsq: null pointer");
    SPHIP_REQUIRE(N > 0 && D > 0, SPHIP_EINVAL, "residual_sumsq: bad shape");
    SPHIP_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0 && ((uintptr_t)e & 15) == 0, SPHIP_EINVAL,
                  "residual_sumsq: operands must be 16-byte aligned");
    SPHIP_WS_CHECK("residual_sumsq");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = N * D;
    int64_t blocks = cdiv(n >> 2, 256);
    if (blocks < 1) blocks = 1;
    if (blocks > RED_BLOCKS) blocks = RED_BLOCKS;
    if (a_dtype == SPHIP_F32) hipLaunchKernelGGL((residual_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, s, (const float*)a, b, e, n, scal);
    else hipLaunchKernelGGL((residual_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)a, b, e, n, scal);
    hipLaunchKernelGGL(scalar_final_kernel, dim3(1), dim3(256), 0, s, scal, (int)blocks, sq);
    SPHIP_CHECK_LAUNCH("residual_sumsq");
    return SPHIP_OK;
}

extern "C" int sphip_scale_by_ratio(const float* in, float* out, int64_t n, float alpha, const float* num,
                                    const float* den, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(in && out, SPHIP_EINVAL, "scale_by_ratio: null pointer");
    SPHIP_REQUIRE(n > 0, SPHIP_EINVAL, "scale_by_ratio: empty");
    SPHIP_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0, SPHIP_EINVAL, "scale_by_ratio: operands must be 16-byte aligned");
    int64_t blocks = cdiv(n >> 2, 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scale_ratio_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, n, alpha, num, den);
    SPHIP_CHECK_LAUNCH("scale_by_ratio");
    return SPHIP_OK;
}
