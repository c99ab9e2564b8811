// Top-k without the (N, M) pre-activation matrix.
// Reference: FusedEncoder.forward, sparsify/fused_encoder.py:29-33 (relu(linear) then
// torch.topk) -- the reference materialises pre_acts and returns it (fused_encoder.py:50); only
// AuxK and multi-top-k ever read it (sparse_coder.py:230-262). When nobody will, the 4 GiB
// matrix need not be written and swept:
//   1. the first latents (about M/4) are computed the ordinary way and their top-k taken: the
//      smallest selected value t0 of a token is a lower bound of its final k-th value;
//   2. the GEMM over the remaining latents keeps nothing but the 16-byte chunks (8 latents)
//      that hold a value above t0, appended to a per-token list (encoder_gemm256.hip, OUTF 4);
//   3. lazy_select_kernel below picks the k largest of {step 1's selection} + {listed values
//      above t0}: the same set as top-k over the whole row, ties to the lowest index (ties at
//      t0 are won by step 1's entries, whose indices are lower than any listed one).
// A token whose list overflows (or whose candidates do not fit the sort buffer) raises a flag in
// device memory; the caller enqueues sphip_encoder_preacts_if / sphip_topk_if behind, which
// redo the ordinary path when the flag is up and cost two empty launches when it is not. No
// host round trip either way.
#include "common.hpp"

namespace sphip {

constexpr int LZ_THREADS = 256;
constexpr int LZ_CAND = 1024;                      // sort capacity (8 KiB of LDS)
constexpr int LZ_MAXCAP = 4096;                    // chunks per token the select kernel can hold
constexpr int LZ_CH = LZ_MAXCAP / LZ_THREADS;      // chunks per thread
constexpr int64_t LZ_HEAD = 256;                   // bytes reserved for the flag at the workspace's start

int launch_gemm256_emit(const void* x, const void* W, const void* bias, int64_t N, int64_t D, int64_t M,
                        const void* tau, unsigned* cnt, void* list, unsigned* ids, int* flag, int64_t cap,
                        int64_t id_base, hipStream_t s);   // encoder_gemm256.hip

static inline int64_t r256(int64_t b) { return (b + 255) / 256 * 256; }

struct LazyWs {
    int* flag;
    unsigned* cnt;
    unsigned* ids;
    uint4* list;
    int64_t bytes;
};

static LazyWs lazy_layout(void* ws, int64_t N, int64_t cap) {
    char* p = (char*)ws;
    LazyWs w;
    w.flag = (int*)p;
    w.cnt = (unsigned*)(p + LZ_HEAD);
    w.ids = (unsigned*)(p + LZ_HEAD + r256(N * 4));
    w.list = (uint4*)(p + LZ_HEAD + r256(N * 4) + r256(N * cap * 4));
    w.bytes = LZ_HEAD + r256(N * 4) + r256(N * cap * 4) + N * cap * 16;
    return w;
}

// Smallest value of every row of non-negative bf16 values (bits; a negative value counts as 0).
__global__ __launch_bounds__(256) void row_min_kernel(const bf16_t* __restrict__ vals, int64_t N, int k,
                                                      unsigned short* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    unsigned m = 0xffffu;
    for (int i = lane; i < k; i += 64) {
        const unsigned v = vals[row * k + i];
        m = min(m, (v & 0x8000u) ? 0u : v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = min(m, (unsigned)__shfl_xor((int)m, o, 64));
    if (lane == 0) out[row] = (unsigned short)m;
}

__device__ __forceinline__ unsigned lz_block_count(unsigned c, unsigned* part, int slot) {
    c = (unsigned)wave_sum_i((int)c);
    unsigned* p = part + slot * (LZ_THREADS / 64);
    if ((threadIdx.x & 63) == 0) p[threadIdx.x >> 6] = c;
    __syncthreads();
    unsigned t = 0;
#pragma unroll
    for (int w = 0; w < LZ_THREADS / 64; ++w) t += p[w];
    return t;
}

// One workgroup per token. Values are compared as raw bf16 bits (all non-negative). The packed
// comparison: with Tq = ((q | 0x8000) in both halves), bits 15 / 31 of Tq - w are set exactly
// where a half of w is <= q.
__global__ __launch_bounds__(LZ_THREADS) void lazy_select_kernel(const bf16_t* __restrict__ vals_a,
                                                                 const int64_t* __restrict__ idx_a, int k,
                                                                 const unsigned short* __restrict__ tau,
                                                                 const unsigned* __restrict__ cnt,
                                                                 const uint4* __restrict__ list,
                                                                 const unsigned* __restrict__ ids, int cap,
                                                                 int* __restrict__ flag, bf16_t* __restrict__ vals,
                                                                 int64_t* __restrict__ idx_out) {
    __shared__ unsigned long long cand[LZ_CAND];
    __shared__ unsigned part[2 * (LZ_THREADS / 64)];
    __shared__ unsigned s_count;
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t row = blockIdx.x;
    const unsigned c = cnt[row];
    if (c > (unsigned)cap) return;          // the GEMM raised the flag: every row is redone
    const uint4* my_list = list + row * cap;
    const unsigned* my_ids = ids + row * cap;

    const bool a_has = tid < k;
    unsigned a_raw = 0;
    int64_t a_idx = 0;
    if (a_has) {
        const unsigned v = vals_a[row * k + tid];
        a_raw = (v & 0x8000u) ? 0u : v;
        a_idx = idx_a[row * k + tid];
    }
    // this thread's chunks; halves that do not exceed the token's threshold become 0 = "no candidate"
    const unsigned T0 = ((unsigned)tau[row] | 0x8000u) * 0x10001u;
    const int umax = (int)((c + LZ_THREADS - 1) / LZ_THREADS);          // block-uniform
    unsigned w[LZ_CH][4];
#pragma unroll
    for (int u = 0; u < LZ_CH; ++u) {
        w[u][0] = w[u][1] = w[u][2] = w[u][3] = 0u;
        if (u < umax) {
            const unsigned ci = (unsigned)tid + (unsigned)u * LZ_THREADS;
            if (ci < c) {
                const uint4 v = my_list[ci];
                const unsigned r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned le = ((T0 - r[j]) & 0x80008000u) >> 15;      // 1 per half that is <= t0
                    w[u][j] = r[j] & ~(le * 0xffffu);
                }
            }
        }
    }
    // elements >= q (q >= 1) among this thread's chunks
    auto count_ge = [&](unsigned q) {
        const unsigned Tq = ((q - 1u) | 0x8000u) * 0x10001u;
        unsigned below = 0;
#pragma unroll
        for (int u = 0; u < LZ_CH; ++u)
            if (u < umax) {
#pragma unroll
                for (int j = 0; j < 4; ++j) below += (unsigned)__popc((Tq - w[u][j]) & 0x80008000u);
            }
        return 8u * (unsigned)umax - below;
    };

    // ---- k-th largest value of the candidates (15 value bits)
    int slot = 0;
    unsigned kth = 0;
    for (int bit = 14; bit >= 0; --bit) {
        const unsigned q = kth | (1u << bit);
        const unsigned mine = count_ge(q) + ((a_has && a_raw >= q) ? 1u : 0u);
        const unsigned tot = lz_block_count(mine, part, slot);
        slot ^= 1;
        if (tot >= (unsigned)k) kth = q;
    }
    // ---- everything >= kth goes to the sort buffer (listed values: kth or 1, whichever is larger)
    const unsigned qlist = kth > 0u ? kth : 1u;
    {
        const unsigned mine = count_ge(qlist) + ((a_has && a_raw >= kth) ? 1u : 0u);
        const unsigned tot = lz_block_count(mine, part, slot);
        slot ^= 1;
        if (tot > (unsigned)LZ_CAND) {      // rows full of ties at the k-th value
            if (tid == 0) *flag = 1;
            return;
        }
    }
    if (tid == 0) s_count = 0;
    __syncthreads();
    auto push = [&](bool hit, unsigned raw, unsigned index) {
        const unsigned long long m = __ballot(hit);
        if (m == 0ull) return;
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&s_count, (unsigned)__popcll(m));
        base = __shfl(base, 0, 64);
        if (hit) {
            const unsigned pos = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            cand[pos] = ((unsigned long long)(raw | 0x8000u) << 32) | (unsigned)(~index);
        }
    };
    push(a_has && a_raw >= kth, a_raw, (unsigned)a_idx);
    {
        const unsigned Tq = ((qlist - 1u) | 0x8000u) * 0x10001u;
#pragma unroll
        for (int u = 0; u < LZ_CH; ++u) {
            if (u < umax) {                                  // block-uniform
                const unsigned all_below =
                    (Tq - w[u][0]) & (Tq - w[u][1]) & (Tq - w[u][2]) & (Tq - w[u][3]) & 0x80008000u;
                const bool any = all_below != 0x80008000u;
                if (__ballot(any) != 0ull) {
                    const unsigned ci = (unsigned)tid + (unsigned)u * LZ_THREADS;
                    const unsigned first = any ? my_ids[ci] * 8u : 0u;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const unsigned raw = (w[u][j >> 1] >> ((j & 1) * 16)) & 0xffffu;
                        push(any && raw >= qlist, raw, first + (unsigned)j);
                    }
                }
            }
        }
    }
    __syncthreads();
    const unsigned count = s_count;
    // ---- bitonic sort, descending on (value, ~index): the order sphip_topk's list path leaves
    unsigned P = 2;
    while (P < count) P <<= 1;
    for (unsigned i = count + tid; i < P; i += LZ_THREADS) cand[i] = 0ull;
    for (unsigned size = 2; size <= P; size <<= 1) {
        for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (unsigned t = tid; t < (P >> 1); t += LZ_THREADS) {
                const unsigned lo = 2 * t - (t & (stride - 1));
                const unsigned hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const unsigned long long a = cand[lo], b = cand[hi];
                if ((a < b) == desc) { cand[lo] = b; cand[hi] = a; }
            }
        }
    }
    __syncthreads();
    for (int t = tid; t < k; t += LZ_THREADS) {
        const unsigned long long e = cand[t];
        vals[row * k + t] = KeyOf<bf16_t>::value((unsigned)(e >> 32));
        idx_out[row * k + t] = (int64_t)(~(unsigned)e);
    }
}

}  // namespace sphip

extern "C" size_t sphip_lazy_workspace_bytes(int64_t N, int64_t cap) {
    using namespace sphip;
    if (N <= 0 || cap <= 0 || cap > LZ_MAXCAP) return 0;
    return (size_t)lazy_layout(nullptr, N, cap).bytes;
}

extern "C" int sphip_row_min(const void* vals, int64_t N, int64_t k, void* out, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(vals && out, SPHIP_EINVAL, "row_min: null pointer");
    SPHIP_REQUIRE(N > 0 && k > 0 && k < (1ll << 31) && sphip::cdiv(N, 4) < (1ll << 31), SPHIP_EINVAL,
                  "row_min: bad shape N=%lld k=%lld", (long long)N, (long long)k);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(row_min_kernel, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, s, (const bf16_t*)vals, N, (int)k,
                       (unsigned short*)out);
    SPHIP_CHECK_LAUNCH("row_min");
    return SPHIP_OK;
}

extern "C" int sphip_encoder_preacts_emit(const void* x, const void* W, const void* bias, int64_t N, int64_t D,
                                          int64_t M_block, int64_t first_latent, const void* tau, int64_t cap,
                                          int reset, void* ws, int64_t ws_bytes, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(x && W && tau && ws, SPHIP_EINVAL, "encoder_preacts_emit: null pointer");
    SPHIP_REQUIRE(N > 0 && D > 0 && M_block > 0 && N < (1ll << 31) && M_block < (1ll << 31) && D < (1ll << 31),
                  SPHIP_EINVAL, "encoder_preacts_emit: bad shape N=%lld D=%lld M_block=%lld", (long long)N, (long long)D,
                  (long long)M_block);
    SPHIP_REQUIRE(D % 64 == 0, SPHIP_EUNSUPPORTED, "encoder_preacts_emit: D=%lld is not a multiple of 64", (long long)D);
    SPHIP_REQUIRE(first_latent >= 0 && first_latent % 8 == 0 && first_latent + M_block < (1ll << 31), SPHIP_EINVAL,
                  "encoder_preacts_emit: first_latent=%lld must be a non-negative multiple of 8", (long long)first_latent);
    SPHIP_REQUIRE(cap > 0 && cap <= LZ_MAXCAP, SPHIP_EINVAL, "encoder_preacts_emit: cap=%lld out of [1, %d]",
                  (long long)cap, LZ_MAXCAP);
    SPHIP_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)ws & 255) == 0 &&
                      ((uintptr_t)tau & 1) == 0, SPHIP_EINVAL,
                  "encoder_preacts_emit: operands must be 16-byte aligned, the workspace 256-byte aligned");
    const LazyWs w = lazy_layout(ws, N, cap);
    SPHIP_REQUIRE(ws_bytes >= w.bytes, SPHIP_EINVAL, "encoder_preacts_emit: workspace %lld < %lld bytes",
                  (long long)ws_bytes, (long long)w.bytes);
    hipStream_t s = (hipStream_t)stream;
    if (reset) {
        SPHIP_REQUIRE(hipMemsetAsync(ws, 0, (size_t)(LZ_HEAD + r256(N * 4)), s) == hipSuccess, SPHIP_ELAUNCH,
                      "encoder_preacts_emit: memset failed");
    }
    return launch_gemm256_emit(x, W, bias, N, D, M_block, tau, w.cnt, w.list, w.ids, w.flag, cap, first_latent / 8, s);
}

extern "C" int sphip_topk_candidates(const void* vals_a, const int64_t* idx_a, const void* tau, int64_t N, int64_t k,
                                     int64_t cap, void* ws, int64_t ws_bytes, void* vals, int64_t* idx, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(vals_a && idx_a && tau && ws && vals && idx, SPHIP_EINVAL, "topk_candidates: null pointer");
    SPHIP_REQUIRE(N > 0 && N < (1ll << 31) && k > 0 && k <= LZ_THREADS, SPHIP_EUNSUPPORTED,
                  "topk_candidates: N=%lld k=%lld (k <= %d)", (long long)N, (long long)k, LZ_THREADS);
    SPHIP_REQUIRE(cap > 0 && cap <= LZ_MAXCAP, SPHIP_EINVAL, "topk_candidates: cap=%lld out of [1, %d]", (long long)cap,
                  LZ_MAXCAP);
    SPHIP_REQUIRE(((uintptr_t)ws & 255) == 0, SPHIP_EINVAL, "topk_candidates: the workspace must be 256-byte aligned");
    const LazyWs w = lazy_layout(ws, N, cap);
    SPHIP_REQUIRE(ws_bytes >= w.bytes, SPHIP_EINVAL, "topk_candidates: workspace %lld < %lld bytes", (long long)ws_bytes,
                  (long long)w.bytes);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(lazy_select_kernel, dim3((unsigned)N), dim3(LZ_THREADS), 0, s, (const bf16_t*)vals_a, idx_a, (int)k,
                       (const unsigned short*)tau, (const unsigned*)w.cnt, (const uint4*)w.list, (const unsigned*)w.ids,
                       (int)cap, w.flag, (bf16_t*)vals, idx);
    SPHIP_CHECK_LAUNCH("topk_candidates");
    return SPHIP_OK;
}
