// Latent segments: a stable counting sort of the N*k (token, slot) entries by latent id.
// Reference: count_per_embedding_k + cumsum + map_embeddings_and_outputs_k,
// sparsify/xformers.py:56-85 and :139-163. The reference ranks entries inside a segment with
// relaxed global atomics (arrival order, not reproducible); here the order inside a segment
// is ascending entry id, so every downstream fp32 sum has a fixed order.
//
// gfx950 design: LSD radix sort with at most 10-bit digits (2 passes up to 2^20 latents).
// A 256-thread workgroup owns a tile of 4096 consecutive entries; each of its 4 waves owns
// 1024 consecutive entries and walks them 64 at a time. The rank of an entry among equal
// digits of its batch comes from wave ballots (one per digit bit), the rank among earlier
// batches from a per-wave LDS counter row -- no atomics anywhere, so ranks are stable.
// Per pass: digit histogram per tile -> exclusive scan over (digit, tile) -> scatter.
// seg_begin is read off the sorted keys (boundary detection), also without atomics.
#include "common.hpp"

namespace sphip {

constexpr int SEG_THREADS = 256;
constexpr int SEG_WAVES = 4;
constexpr int SEG_ITEMS = 16;                                  // batches of 64 per wave
constexpr int SEG_TILE = SEG_THREADS * SEG_ITEMS;              // 4096 entries per workgroup
constexpr int SEG_MAX_DIGIT_BITS = 10;

struct SegPlan {
    int passes, digit_bits;
    int64_t tiles, E;
    size_t off_key[2], off_val[2], off_hist, off_spine, total;
};

static SegPlan make_plan(int64_t N, int64_t k, int64_t M) {
    SegPlan p;
    p.E = N * k;
    int bits = 1;
    while ((1ll << bits) < M + 1) ++bits;     // keys 0..M (M = sentinel bin)
    p.passes = (bits + SEG_MAX_DIGIT_BITS - 1) / SEG_MAX_DIGIT_BITS;
    p.digit_bits = (bits + p.passes - 1) / p.passes;
    p.tiles = cdiv(p.E, SEG_TILE);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    p.off_key[0] = take((size_t)p.E * 4);
    p.off_key[1] = take((size_t)p.E * 4);
    p.off_val[0] = take((size_t)p.E * 4);
    p.off_val[1] = take((size_t)p.E * 4);
    p.off_hist = take(((size_t)p.tiles << p.digit_bits) * 4 + 4);
    p.off_spine = take((size_t)cdiv(p.tiles << p.digit_bits, 4096) * 4 + 4);
    p.total = o;
    return p;
}

// Negative ids mark inactive slots (latent owned by another shard); they -- and any
// out-of-range id -- sort into a sentinel bin M after all real latents, so no kernel that
// walks segments 0..M-1 ever sees them. In-range input is untouched.
__device__ __forceinline__ unsigned clamp_key(int64_t v, int64_t M) {
    return (unsigned)((v < 0 || v >= M) ? M : v);
}

template <bool FIRST>
__device__ __forceinline__ void load_entries(const int64_t* idx64, const unsigned* keys, const unsigned* vals,
                                             int64_t base, int64_t E, int64_t M, int lane,
                                             unsigned (&key)[SEG_ITEMS], unsigned (&val)[SEG_ITEMS]) {
#pragma unroll
    for (int it = 0; it < SEG_ITEMS; ++it) {
        const int64_t e = base + it * 64 + lane;
        if (e < E) {
            if (FIRST) { key[it] = clamp_key(idx64[e], M); val[it] = (unsigned)e; }
            else { key[it] = keys[e]; val[it] = vals[e]; }
        } else { key[it] = 0xffffffffu; val[it] = 0; }
    }
}

// Stable rank of one 64-entry batch among equal digits. cnt = this wave's LDS counter row
// [R]: on entry the next free slot per digit, advanced by the batch's group sizes.
__device__ __forceinline__ void rank_batch(unsigned digit, bool valid, int digit_bits, int lane, unsigned* cnt,
                                           unsigned& pos) {
    unsigned long long m = __ballot(valid);
    for (int b = 0; b < digit_bits; ++b) {
        const unsigned long long bm = __ballot((digit >> b) & 1u);
        m &= ((digit >> b) & 1u) ? bm : ~bm;
    }
    // m = valid lanes with the same digit (garbage for invalid lanes, which do nothing)
    const unsigned before = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    const unsigned total = (unsigned)__popcll(m);
    unsigned start = 0;
    if (valid) start = cnt[digit];
    pos = start + before;
    // all reads of this batch are done before any write: same wave, program order, LDS is
    // in-order per wave; the barrier-free handshake is the wave itself.
    __builtin_amdgcn_wave_barrier();
    if (valid && before + 1 == total) cnt[digit] = start + total;   // last lane of the group
    __builtin_amdgcn_wave_barrier();
}

template <bool FIRST>
__global__ __launch_bounds__(SEG_THREADS) void seg_hist_kernel(const int64_t* __restrict__ idx64,
                                                               const unsigned* __restrict__ keys, int64_t E,
                                                               int64_t M, int shift, int digit_bits,
                                                               int64_t tiles, unsigned* __restrict__ hist) {
    __shared__ unsigned cnt[SEG_WAVES][1 << SEG_MAX_DIGIT_BITS];
    const int R = 1 << digit_bits;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int d = lane; d < R; d += 64) cnt[wave][d] = 0;
    __builtin_amdgcn_wave_barrier();
    const int64_t base = (int64_t)blockIdx.x * SEG_TILE + (int64_t)wave * (64 * SEG_ITEMS);
    for (int it = 0; it < SEG_ITEMS; ++it) {
        const int64_t e = base + it * 64 + lane;
        const bool valid = e < E;
        unsigned key = 0;
        if (valid) key = FIRST ? clamp_key(idx64[e], M) : keys[e];
        unsigned pos;
        rank_batch((key >> shift) & (unsigned)(R - 1), valid, digit_bits, lane, cnt[wave], pos);
    }
    __syncthreads();
    for (int d = tid; d < R; d += SEG_THREADS)
        hist[(int64_t)d * tiles + blockIdx.x] = cnt[0][d] + cnt[1][d] + cnt[2][d] + cnt[3][d];
}

// Exclusive scan of n unsigned values in three small kernels (chunk totals -> spine ->
// apply); a chunk is 4096 values per 1024-thread workgroup.
constexpr int SCAN_CHUNK = 4096;

__device__ __forceinline__ unsigned scan_block_1024(unsigned mine, unsigned* wave_tot, unsigned& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    __syncthreads();
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const unsigned t = wave_tot[w];
        if (w < wave) off += t;
        tot += t;
    }
    total = tot;
    return off + incl - mine;   // exclusive prefix of this thread
}

__global__ __launch_bounds__(1024) void scan_totals_kernel(const unsigned* __restrict__ a, int64_t n,
                                                           unsigned* __restrict__ chunk_tot) {
    __shared__ unsigned wave_tot[16];
    const int64_t i = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * 4;
    unsigned mine = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) mine += (i + j < n) ? a[i + j] : 0u;
    unsigned total;
    scan_block_1024(mine, wave_tot, total);
    if (threadIdx.x == 0) chunk_tot[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void scan_spine_kernel(unsigned* __restrict__ chunk_tot, int64_t nchunks) {
    __shared__ unsigned wave_tot[16];
    unsigned carry = 0;
    for (int64_t base = 0; base < nchunks; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const unsigned mine = (i < nchunks) ? chunk_tot[i] : 0u;
        unsigned total;
        const unsigned excl = scan_block_1024(mine, wave_tot, total);
        if (i < nchunks) chunk_tot[i] = carry + excl;
        carry += total;
    }
}

__global__ __launch_bounds__(1024) void scan_apply_kernel(unsigned* __restrict__ a, int64_t n,
                                                          const unsigned* __restrict__ chunk_off) {
    __shared__ unsigned wave_tot[16];
    const int64_t i = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * 4;
    unsigned v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (i + j < n) ? a[i + j] : 0u;
    unsigned total;
    unsigned run = chunk_off[blockIdx.x] + scan_block_1024(v[0] + v[1] + v[2] + v[3], wave_tot, total);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (i + j < n) a[i + j] = run;
        run += v[j];
    }
}

// also used by gather_l2.hip
void launch_exclusive_scan(unsigned* a, int64_t n, unsigned* chunk_tot, hipStream_t s) {
    const int64_t nchunks = cdiv(n, SCAN_CHUNK);
    hipLaunchKernelGGL(scan_totals_kernel, dim3((unsigned)nchunks), dim3(1024), 0, s, a, n, chunk_tot);
    hipLaunchKernelGGL(scan_spine_kernel, dim3(1), dim3(1024), 0, s, chunk_tot, nchunks);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nchunks), dim3(1024), 0, s, a, n, chunk_tot);
}

template <bool FIRST>
__global__ __launch_bounds__(SEG_THREADS) void seg_scatter_kernel(const int64_t* __restrict__ idx64,
                                                                  const unsigned* __restrict__ keys,
                                                                  const unsigned* __restrict__ vals, int64_t E,
                                                                  int64_t M, int shift, int digit_bits,
                                                                  int64_t tiles,
                                                                  const unsigned* __restrict__ hist_scanned,
                                                                  unsigned* __restrict__ out_keys,
                                                                  unsigned* __restrict__ out_vals) {
    __shared__ unsigned cnt[SEG_WAVES][1 << SEG_MAX_DIGIT_BITS];
    const int R = 1 << digit_bits;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int d = lane; d < R; d += 64) cnt[wave][d] = 0;
    __builtin_amdgcn_wave_barrier();
    const int64_t base = (int64_t)blockIdx.x * SEG_TILE + (int64_t)wave * (64 * SEG_ITEMS);
    unsigned key[SEG_ITEMS], val[SEG_ITEMS];
    load_entries<FIRST>(idx64, keys, vals, base, E, M, lane, key, val);
    // counting sub-pass: per-wave digit counts
#pragma unroll
    for (int it = 0; it < SEG_ITEMS; ++it) {
        const bool valid = base + it * 64 + lane < E;
        unsigned pos;
        rank_batch((key[it] >> shift) & (unsigned)(R - 1), valid, digit_bits, lane, cnt[wave], pos);
    }
    __syncthreads();
    // turn counts into starting offsets: global offset of (digit, tile) + earlier waves
    for (int d = tid; d < R; d += SEG_THREADS) {
        unsigned run = hist_scanned[(int64_t)d * tiles + blockIdx.x];
#pragma unroll
        for (int w = 0; w < SEG_WAVES; ++w) {
            const unsigned c = cnt[w][d];
            cnt[w][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < SEG_ITEMS; ++it) {
        const bool valid = base + it * 64 + lane < E;
        unsigned pos;
        rank_batch((key[it] >> shift) & (unsigned)(R - 1), valid, digit_bits, lane, cnt[wave], pos);
        if (valid) {
            out_keys[pos] = key[it];
            out_vals[pos] = val[it];
        }
    }
}

// seg_begin[j] = first position whose key >= j ; seg_begin[M] = E.
__global__ __launch_bounds__(256) void seg_bounds_kernel(const unsigned* __restrict__ sorted_keys, int64_t E,
                                                         int64_t M, int* __restrict__ seg_begin) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p > E) return;
    const int64_t prev = (p == 0) ? -1 : (int64_t)sorted_keys[p - 1];
    const int64_t cur = (p == E) ? M : (int64_t)sorted_keys[p];
    for (int64_t j = prev + 1; j <= cur; ++j) seg_begin[j] = (int)p;
}

}  // namespace sphip

extern "C" size_t sphip_segments_workspace_bytes(int64_t N, int64_t k, int64_t M) {
    if (N <= 0 || k <= 0 || M <= 0) return 0;
    return sphip::make_plan(N, k, M).total;
}

extern "C" int sphip_build_latent_segments(const int64_t* idx, int64_t N, int64_t k, int64_t M, uint32_t* perm,
                                           int32_t* seg_begin, void* workspace, size_t workspace_bytes,
                                           void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(idx && perm && seg_begin && workspace, SPHIP_EINVAL, "build_latent_segments: null pointer");
    SPHIP_REQUIRE(N > 0 && k > 0 && M > 0, SPHIP_EINVAL, "build_latent_segments: empty shape");
    SPHIP_REQUIRE(N * k < (1ll << 31), SPHIP_EUNSUPPORTED, "build_latent_segments: N*k=%lld must be < 2^31",
                  (long long)(N * k));
    SPHIP_REQUIRE(M < (1ll << 31), SPHIP_EUNSUPPORTED, "build_latent_segments: M too large");
    const SegPlan p = make_plan(N, k, M);
    SPHIP_REQUIRE(workspace_bytes >= p.total, SPHIP_EWORKSPACE, "build_latent_segments: workspace %zu < %zu bytes",
                  workspace_bytes, p.total);
    SPHIP_REQUIRE(((uintptr_t)workspace & 255) == 0, SPHIP_EINVAL, "build_latent_segments: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    unsigned* key[2] = {(unsigned*)(ws + p.off_key[0]), (unsigned*)(ws + p.off_key[1])};
    unsigned* val[2] = {(unsigned*)(ws + p.off_val[0]), (unsigned*)(ws + p.off_val[1])};
    unsigned* hist = (unsigned*)(ws + p.off_hist);
    unsigned* spine = (unsigned*)(ws + p.off_spine);
    const int64_t nh = p.tiles << p.digit_bits;
    dim3 grid((unsigned)p.tiles), block(SEG_THREADS);
    const unsigned* in_key = nullptr;
    const unsigned* in_val = nullptr;
    for (int pass = 0; pass < p.passes; ++pass) {
        const int shift = pass * p.digit_bits;
        const bool last = pass == p.passes - 1;
        unsigned* ok = key[pass & 1];
        unsigned* ov = last ? perm : val[pass & 1];
        if (pass == 0) {
            hipLaunchKernelGGL((seg_hist_kernel<true>), grid, block, 0, s, idx, in_key, p.E, M, shift, p.digit_bits, p.tiles, hist);
            launch_exclusive_scan(hist, nh, spine, s);
            hipLaunchKernelGGL((seg_scatter_kernel<true>), grid, block, 0, s, idx, in_key, in_val, p.E, M, shift, p.digit_bits, p.tiles, hist, ok, ov);
        } else {
            hipLaunchKernelGGL((seg_hist_kernel<false>), grid, block, 0, s, idx, in_key, p.E, M, shift, p.digit_bits, p.tiles, hist);
            launch_exclusive_scan(hist, nh, spine, s);
            hipLaunchKernelGGL((seg_scatter_kernel<false>), grid, block, 0, s, idx, in_key, in_val, p.E, M, shift, p.digit_bits, p.tiles, hist, ok, ov);
        }
        in_key = ok;
        in_val = ov;
    }
    hipLaunchKernelGGL(seg_bounds_kernel, dim3((unsigned)cdiv(p.E + 1, 256)), dim3(256), 0, s, in_key, p.E, M, seg_begin);
    SPHIP_CHECK_LAUNCH("build_latent_segments");
    return SPHIP_OK;
}
