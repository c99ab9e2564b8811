// Row-wise top-k / group-max selection over the pre-activations.
// Reference: torch.topk(preacts, k, dim=-1, sorted=False), sparsify/fused_encoder.py:33;
// the AuxK variant torch.where(dead_mask[None], pre_acts, -inf).topk(k_aux),
// sparsify/sparse_coder.py:240-243; groupmax, fused_encoder.py:34-43.
//
// gfx950 design: one 1024-thread workgroup (16 waves) per token row; keys are the
// order-preserving unsigned image of the value bits (16 bit for bf16, 32 bit for f32).
//   fast path  (k <= 3072): sweep 1 takes each thread's maximum key (for k > 1024: the maxima
//              of up to 4 interleaved groups of its chunks); the k-th largest of
//              these maxima (bisection on the key bits, one barrier per bit) is a
//              proven lower bound tau of the k-th largest element. Sweep 2 (row now in
//              L2 / Infinity Cache) appends every element >= tau to an LDS candidate list
//              (wave-aggregated LDS atomic), typically ~1.1 k entries. A bitonic sort of the
//              candidates on (key desc, index asc) yields the exact, deterministic top-k.
//   tile path  (block maxima from the GEMM epilogue, no mask): the thread maxima are taken
//              over the row's block maxima (N x M/128 values, 1/128 of the row) instead of
//              over the row, and sweep 2 reads only the blocks whose maximum reaches tau --
//              about k of the M/128 blocks -- 16 lanes per 256-B block. Same tau rule, same
//              candidate list, same result.
//   exact path (k > 1024, or more than CAP candidates, e.g. rows full of ties): bisection
//              of the k-th largest key over the whole row, then an index-ordered emission
//              (block scans) that takes ties in ascending index order.
// Both paths choose the same SET: all keys above the k-th key plus the lowest-index ties.
// The kernel is instantiated for 1024 threads per row (any shape) and, for the tile path with
// k <= 256, for 256 threads per row: a row's selection is a chain of ~80 workgroup barriers
// (16 bisection steps, ~45 sort stages), so it is latency-bound, and four-wave workgroups pass a
// barrier faster and put four times as many rows in flight per CU (8 KiB candidate list
// instead of 64 KiB).
#include "common.hpp"
#include <stdlib.h>

#ifndef SPHIP_TOPK_U
#define SPHIP_TOPK_U 4         // block loads in flight per lane on the four-wave path (6: no gain, 8 spills)
#endif
#ifndef SPHIP_TOPK_REGS
#define SPHIP_TOPK_REGS 1
#endif

namespace sphip {

constexpr int TK_THREADS = 1024;   // threads per row, general instantiation
constexpr int TK_CAP = 8192;       // its candidate capacity (8 B each -> 64 KB LDS)
constexpr int TK_THREADS_S = 256;  // threads per row, tile path with k <= 256
constexpr int TK_CAP_S = 1024;     // its candidate capacity (8 KB LDS); more candidates -> exact path
constexpr int TK_GROUPS = 4;       // maxima kept per thread: the fast path covers k <= 3 * threads
constexpr int TK_TILE = SPHIP_TOPK_TILE;   // latents per block maximum (sphip_encoder_preacts)
constexpr int TK_TCAP = 2048;  // selected-block list capacity

// Loads VEC consecutive keys starting at element i (i % VEC == 0 when ALIGNED).
template <typename T, bool ALIGNED>
__device__ __forceinline__ void load_keys(const T* row, const uint8_t* keep, int i, int M,
                                          unsigned (&key)[KeyOf<T>::VEC]) {
    constexpr int V = KeyOf<T>::VEC;
    if (ALIGNED) {
        uint4 raw = *reinterpret_cast<const uint4*>(row + i);
        const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
        for (int j = 0; j < V; ++j) key[j] = to_key(e[j]);
        if (keep != nullptr) {
            // V mask bytes in one load (the launcher checks the mask's alignment for ALIGNED)
            unsigned long long kb;
            if (V == 8) kb = *reinterpret_cast<const unsigned long long*>(keep + i);
            else kb = *reinterpret_cast<const unsigned*>(keep + i);
#pragma unroll
            for (int j = 0; j < V; ++j)
                if (((kb >> (8 * j)) & 0xffull) == 0ull) key[j] = 0;
        }
    } else {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int c = i + j;
            key[j] = (c < M && (keep == nullptr || keep[c])) ? to_key(row[c]) : 0u;
        }
    }
}

// block-wide sum of a per-thread count; `slot` alternates between calls so one barrier
// per call is enough. part = [2][TK_WAVES] in LDS.
template <int TK_WAVES>
__device__ __forceinline__ unsigned block_count(unsigned c, unsigned* part, int slot) {
    c = (unsigned)wave_sum_i((int)c);
    unsigned* p = part + slot * TK_WAVES;
    if ((threadIdx.x & 63) == 0) p[threadIdx.x >> 6] = c;
    __syncthreads();
    unsigned t = 0;
#pragma unroll
    for (int w = 0; w < TK_WAVES; ++w) t += p[w];
    return t;
}

template <typename T, bool ALIGNED, int TK_THREADS = sphip::TK_THREADS, int TK_CAP = sphip::TK_CAP>
__global__ __launch_bounds__(TK_THREADS, 8) void topk_kernel(const T* __restrict__ pre,
                                                          const uint8_t* __restrict__ keep,
                                                          const T* __restrict__ tile_max, int M,
                                                          int k, T* __restrict__ vals,
                                                          int64_t* __restrict__ idx_out,
                                                          const int* __restrict__ run_if = nullptr) {
    if (run_if != nullptr && *run_if == 0) return;     // predicated launch (sphip_topk_if)
    constexpr int TK_WAVES = TK_THREADS / 64;
    constexpr int V = KeyOf<T>::VEC;
    constexpr int BITS = KeyOf<T>::BITS;
    __shared__ unsigned long long cand[TK_CAP];
    __shared__ unsigned part[2 * TK_WAVES];
    __shared__ unsigned s_count, s_ntiles;
    __shared__ unsigned scan_tot[TK_WAVES];
    __shared__ unsigned short tiles[TK_TCAP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* row = pre + (int64_t)blockIdx.x * M;
    T* out_v = vals + (int64_t)blockIdx.x * k;
    int64_t* out_i = idx_out + (int64_t)blockIdx.x * k;
    const int span = TK_THREADS * V;
    int slot = 0;

    // group maxima per thread in use: about twice as many maxima as k, so that tau (the k-th
    // largest of them) sits near their median and not at their minimum
    // short rows with a large k (the compact AuxK selection: k = 2048 of ~13 k): ~18 sweeps of a
    // cache-resident row are cheaper than sorting thousands of candidates
    bool exact = k > 3 * TK_THREADS || (k >= 256 && (int64_t)M <= 16ll * k);
    const int R = k <= TK_THREADS / 2 ? 1 : (k <= TK_THREADS ? 2 : TK_GROUPS);
    if (!exact) {
        const int T_ = (M + TK_TILE - 1) / TK_TILE;
        // (fewer blocks than k: their maxima bound nothing, take the plain sweep)
        const bool by_tile = ALIGNED && tile_max != nullptr && keep == nullptr && T_ <= 65536 && T_ >= k && k <= TK_THREADS;
        // block maxima of this token: element t at tm_row[t * 64] (layout: sparsify_hip.h)
        const T* tm_row = tile_max + (int64_t)(blockIdx.x >> 6) * T_ * 64 + (blockIdx.x & 63);
        // ---- sweep 1: thread maxima (over the block maxima when they are given)
        // every maximum is a distinct element of the row, so the k-th largest of them bounds
        // the k-th largest element from below
        unsigned tmax[TK_GROUPS] = {0u, 0u, 0u, 0u};
        if (by_tile) {
            // chunk c of the block maxima goes to group c % 4: up to 4 * threads blocks keep a
            // maximum of their own, which puts tau closer to the k-th largest element
            int c = 0;
            for (int t = tid; t < T_; t += TK_THREADS, ++c) {
                const unsigned m = to_key(tm_row[t * 64]);
                const int grp = c & (TK_GROUPS - 1);
#pragma unroll
                for (int r = 0; r < TK_GROUPS; ++r) tmax[r] = (r == grp) ? max(tmax[r], m) : tmax[r];
            }
        } else {
            int c = 0;
            for (int i = tid * V; i < M; i += span, ++c) {
                unsigned key[V];
                load_keys<T, ALIGNED>(row, keep, i, M, key);
                unsigned m = 0;
#pragma unroll
                for (int j = 0; j < V; ++j) m = max(m, key[j]);
                const int grp = R == 1 ? 0 : c % R;
#pragma unroll
                for (int r = 0; r < TK_GROUPS; ++r) tmax[r] = (r == grp) ? max(tmax[r], m) : tmax[r];
            }
        }
        // ---- k-th largest thread maximum (lower bound of the k-th largest element)
        unsigned tau = 0;
        for (int bit = BITS - 1; bit >= 0; --bit) {
            const unsigned t = tau | (1u << bit);
            unsigned mine = 0;
#pragma unroll
            for (int r = 0; r < TK_GROUPS; ++r) mine += (tmax[r] >= t) ? 1u : 0u;
            const unsigned c = block_count<TK_WAVES>(mine, part, slot);
            slot ^= 1;
            if (c >= (unsigned)k) tau = t;
        }
        // ---- sweep 2: collect candidates >= tau
        if (tid == 0) { s_count = 0; s_ntiles = 0; }
        __syncthreads();
        // appends this lane's hit to the candidate list (wave-aggregated LDS atomic)
        auto push = [&](bool hit, unsigned key, int index) {
            const unsigned long long m = __ballot(hit);
            if (m == 0ull) return;
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&s_count, (unsigned)__popcll(m));
            base = __shfl(base, 0, 64);
            if (hit) {
                const unsigned pos = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                if (pos < (unsigned)TK_CAP) cand[pos] = ((unsigned long long)key << 32) | (unsigned)(~(unsigned)index);
            }
        };
        if (by_tile) {
            // list of the blocks that can hold a candidate
            // (up to TK_GROUPS chunks: a thread's group maxima are the block maxima it loaded, no reload)
            const bool in_regs = SPHIP_TOPK_REGS && T_ <= TK_GROUPS * TK_THREADS;
            int c = 0;
            for (int t0 = 0; t0 < T_; t0 += TK_THREADS, ++c) {
                const int t = t0 + tid;
                unsigned mkey = 0;
                if (in_regs) {
#pragma unroll
                    for (int r = 0; r < TK_GROUPS; ++r) mkey = (r == c) ? tmax[r] : mkey;
                } else if (t < T_) {
                    mkey = to_key(tm_row[t * 64]);
                }
                const bool hit = t < T_ && mkey >= tau;
                const unsigned long long m = __ballot(hit);
                if (m == 0ull) continue;
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&s_ntiles, (unsigned)__popcll(m));
                base = __shfl(base, 0, 64);
                if (hit) {
                    const unsigned pos = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                    if (pos < (unsigned)TK_TCAP) tiles[pos] = (unsigned short)t;
                }
            }
            __syncthreads();
            const unsigned nt = s_ntiles;
            if (nt > (unsigned)TK_TCAP) {
                exact = true;   // block-uniform
            } else {
                constexpr int LPT = TK_TILE / V;          // lanes per block: 16 (bf16) / 32 (f32)
                constexpr int GPW = 64 / LPT;             // blocks per wave instruction
                // U block loads per lane are issued before the first candidate is pushed (the
                // list's ballots and LDS atomics would otherwise serialise the loads' latencies)
                // (kept as the 16 bytes they arrive in and widened to keys when they are pushed: with
                // more of them in flight per lane, SPHIP_TOPK_U = 6, measured no faster)
                constexpr int U = TK_THREADS >= 1024 ? 1 : SPHIP_TOPK_U;
                for (unsigned g0 = (unsigned)wave * GPW; g0 < nt; g0 += U * TK_WAVES * GPW) {
                    uint4 raw[U];
                    auto where = [&](int u) {              // first element of this lane's 16 bytes, M if none
                        const unsigned g = g0 + (unsigned)(u * TK_WAVES * GPW) + (unsigned)(lane / LPT);
                        return g < nt ? (int)tiles[g] * TK_TILE + (lane % LPT) * V : M;
                    };
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int at = where(u);
                        raw[u] = make_uint4(0u, 0u, 0u, 0u);
                        if (at < M) raw[u] = *reinterpret_cast<const uint4*>(row + at);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (g0 + (unsigned)(u * TK_WAVES * GPW) >= nt) break;      // wave-uniform
                        const int at = where(u);
                        const T* e = reinterpret_cast<const T*>(&raw[u]);
#pragma unroll
                        for (int j = 0; j < V; ++j) {
                            const unsigned key = at < M ? to_key(e[j]) : 0u;
                            push(at < M && key >= tau, key, at + j);
                        }
                    }
                }
            }
        } else {
            for (int i0 = 0; i0 < M; i0 += span) {
                const int i = i0 + tid * V;
                unsigned key[V];
                unsigned vmax = 0;
                if (i < M) {
                    load_keys<T, ALIGNED>(row, keep, i, M, key);
#pragma unroll
                    for (int j = 0; j < V; ++j) vmax = max(vmax, (ALIGNED || i + j < M) ? key[j] : 0u);
                }
                const bool any = (i < M) && (vmax >= tau);
                if (__ballot(any) == 0ull) continue;
#pragma unroll
                for (int j = 0; j < V; ++j) push(any && (i + j < M) && key[j] >= tau, any ? key[j] : 0u, i + j);
            }
        }
        __syncthreads();
        const unsigned count = s_count;
        if (exact || count > (unsigned)TK_CAP) {
            exact = true;   // block-uniform
        } else {
            // ---- bitonic sort, descending on (key, ~index)
            unsigned P = 2;
            while (P < count) P <<= 1;
            for (unsigned i = count + tid; i < P; i += TK_THREADS) cand[i] = 0ull;
            for (unsigned size = 2; size <= P; size <<= 1) {
                for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
                    __syncthreads();
                    for (unsigned t = tid; t < (P >> 1); t += TK_THREADS) {
                        const unsigned lo = 2 * t - (t & (stride - 1));
                        const unsigned hi = lo + stride;
                        const bool desc = (lo & size) == 0;
                        const unsigned long long a = cand[lo], b = cand[hi];
                        if ((a < b) == desc) { cand[lo] = b; cand[hi] = a; }
                    }
                }
            }
            __syncthreads();
            for (int t = tid; t < k; t += TK_THREADS) {
                const unsigned long long e = cand[t];
                out_v[t] = KeyOf<T>::value((unsigned)(e >> 32));
                out_i[t] = (int64_t)(~(unsigned)e);
            }
            return;
        }
    }

    // ---- exact path: bisection of the k-th largest key over the whole row
    unsigned kth = 0;
    for (int bit = BITS - 1; bit >= 0; --bit) {
        const unsigned t = kth | (1u << bit);
        unsigned c = 0;
        for (int i = tid * V; i < M; i += span) {
            unsigned key[V];
            load_keys<T, ALIGNED>(row, keep, i, M, key);
#pragma unroll
            for (int j = 0; j < V; ++j) c += ((ALIGNED || i + j < M) && key[j] >= t) ? 1u : 0u;
        }
        const unsigned tot = block_count<TK_WAVES>(c, part, slot);
        slot ^= 1;
        if (tot >= (unsigned)k) kth = t;
    }
    unsigned cgt = 0;
    for (int i = tid * V; i < M; i += span) {
        unsigned key[V];
        load_keys<T, ALIGNED>(row, keep, i, M, key);
#pragma unroll
        for (int j = 0; j < V; ++j) cgt += ((ALIGNED || i + j < M) && key[j] > kth) ? 1u : 0u;
    }
    const unsigned count_gt = block_count<TK_WAVES>(cgt, part, slot);
    slot ^= 1;
    const unsigned need_eq = (unsigned)k - count_gt;

    // index-ordered emission: [0, count_gt) strictly greater, [count_gt, k) lowest-index ties
    unsigned run_gt = 0, run_eq = 0;
    for (int i0 = 0; i0 < M; i0 += span) {
        const int i = i0 + tid * V;
        unsigned key[V];
        unsigned g = 0, q = 0;
        if (i < M) {
            load_keys<T, ALIGNED>(row, keep, i, M, key);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const bool in = ALIGNED || i + j < M;
                g += (in && key[j] > kth) ? 1u : 0u;
                q += (in && key[j] == kth) ? 1u : 0u;
            }
        }
        // block exclusive scan of (g | q << 16); per-thread counts <= 8, totals <= 8192
        unsigned packed = g | (q << 16);
        unsigned incl = packed;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned n = __shfl_up(incl, o, 64);
            if (lane >= o) incl += n;
        }
        __syncthreads();   // scan_tot free for reuse
        if (lane == 63) scan_tot[wave] = incl;
        __syncthreads();
        unsigned wave_off = 0, total = 0;
#pragma unroll
        for (int w = 0; w < TK_WAVES; ++w) {
            const unsigned t = scan_tot[w];
            if (w < wave) wave_off += t;
            total += t;
        }
        const unsigned excl = wave_off + incl - packed;
        unsigned pg = run_gt + (excl & 0xffffu);
        unsigned pq = run_eq + (excl >> 16);
        if (i < M) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                if (!(ALIGNED || i + j < M)) continue;
                if (key[j] > kth) {
                    out_v[pg] = KeyOf<T>::value(key[j]);
                    out_i[pg] = i + j;
                    ++pg;
                } else if (key[j] == kth) {
                    if (pq < need_eq) {
                        out_v[count_gt + pq] = KeyOf<T>::value(key[j]);
                        out_i[count_gt + pq] = i + j;
                    }
                    ++pq;
                }
            }
        }
        run_gt += total & 0xffffu;
        run_eq += total >> 16;
    }
}

// The exact path for rows that fit the workgroup's registers (M <= 2 * VEC * 1024: the compact AuxK
// selection, k_aux = 2048 of ~13 k dead latents): the row is read once, 2 * VEC keys per thread (16-bit
// keys two to a register: the kernel has to fit 64 registers for two rows per CU), and the
// bisection of the k-th largest key, the count above it and the index-ordered emission all run on
// registers instead of sweeping the (cache-resident) row 18 times. Same steps, same element-to-thread
// map and same scans as the exact path of topk_kernel, hence the same output, position for position.
// A/B on one box, N = 16384 rows of 13120, k = 2048: 1.21 ms with sweeps, 0.90 ms from registers
// (1.48 ms as four-wave workgroups owning 64 consecutive elements per thread, eight rows per CU).
template <typename T>
__global__ __launch_bounds__(1024, 8) void topk_resident_kernel(const T* __restrict__ pre, int M, int k,
                                                                T* __restrict__ vals, int64_t* __restrict__ idx_out,
                                                                const int* __restrict__ run_if) {
    if (run_if != nullptr && *run_if == 0) return;
    constexpr int TKT = 1024, TKW = TKT / 64;
    constexpr int V = KeyOf<T>::VEC, BITS = KeyOf<T>::BITS, Q = 2;
    constexpr int PER = 32 / BITS, W = V / PER;
    __shared__ unsigned part[2 * TKW];
    __shared__ unsigned scan_tot[TKW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* row = pre + (int64_t)blockIdx.x * M;
    T* out_v = vals + (int64_t)blockIdx.x * k;
    int64_t* out_i = idx_out + (int64_t)blockIdx.x * k;
    unsigned kreg[Q][W];
    bool have[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int i = (q * TKT + tid) * V;
        have[q] = i < M;
        unsigned key[V];
#pragma unroll
        for (int j = 0; j < V; ++j) key[j] = 0u;
        if (have[q]) load_keys<T, true>(row, nullptr, i, M, key);
#pragma unroll
        for (int w = 0; w < W; ++w) {
            if (PER == 1) kreg[q][w] = key[w];
            else kreg[q][w] = key[2 * w] | (key[2 * w + 1] << 16);
        }
    }
    auto K = [&](int q, int j) -> unsigned {
        if (PER == 1) return kreg[q][j];
        return (j & 1) ? (kreg[q][j >> 1] >> 16) : (kreg[q][j >> 1] & 0xffffu);
    };
    int slot = 0;
    unsigned kth = 0;
    for (int bit = BITS - 1; bit >= 0; --bit) {
        const unsigned t = kth | (1u << bit);
        unsigned c = 0;
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int j = 0; j < V; ++j) c += (have[q] && K(q, j) >= t) ? 1u : 0u;
        const unsigned tot = block_count<TKW>(c, part, slot);
        slot ^= 1;
        if (tot >= (unsigned)k) kth = t;
    }
    unsigned cgt = 0;
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int j = 0; j < V; ++j) cgt += (have[q] && K(q, j) > kth) ? 1u : 0u;
    const unsigned count_gt = block_count<TKW>(cgt, part, slot);
    const unsigned need_eq = (unsigned)k - count_gt;
    unsigned run_gt = 0, run_eq = 0;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if (q * TKT * V >= M) break;                      // block-uniform
        unsigned g = 0, e = 0;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            g += (have[q] && K(q, j) > kth) ? 1u : 0u;
            e += (have[q] && K(q, j) == kth) ? 1u : 0u;
        }
        const unsigned packed = g | (e << 16);
        unsigned incl = packed;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned n = __shfl_up(incl, o, 64);
            if (lane >= o) incl += n;
        }
        __syncthreads();   // scan_tot free for reuse
        if (lane == 63) scan_tot[wave] = incl;
        __syncthreads();
        unsigned wave_off = 0, total = 0;
#pragma unroll
        for (int w = 0; w < TKW; ++w) {
            const unsigned t = scan_tot[w];
            if (w < wave) wave_off += t;
            total += t;
        }
        const unsigned excl = wave_off + incl - packed;
        unsigned pg = run_gt + (excl & 0xffffu);
        unsigned pq = run_eq + (excl >> 16);
        if (have[q]) {
            const int i = (q * TKT + tid) * V;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const unsigned key = K(q, j);
                if (key > kth) {
                    out_v[pg] = KeyOf<T>::value(key);
                    out_i[pg] = i + j;
                    ++pg;
                } else if (key == kth) {
                    if (pq < need_eq) {
                        out_v[count_gt + pq] = KeyOf<T>::value(key);
                        out_i[count_gt + pq] = i + j;
                    }
                    ++pq;
                }
            }
        }
        run_gt += total & 0xffffu;
        run_eq += total >> 16;
    }
}

// groupmax: one wave per (token, group); first maximal index wins (torch.max semantics).
template <typename T>
__global__ __launch_bounds__(256) void groupmax_kernel(const T* __restrict__ pre, int M, int k, int width,
                                                       int64_t total, T* __restrict__ vals,
                                                       int64_t* __restrict__ idx_out) {
    const int lane = threadIdx.x & 63;
    const int64_t gid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gid >= total) return;
    const int64_t n = gid / k;
    const int g = (int)(gid % k);
    const T* p = pre + n * M + (int64_t)g * width;
    unsigned long long best = 0ull;
    for (int i = lane; i < width; i += 64) {
        const unsigned long long e = ((unsigned long long)to_key(p[i]) << 32) | (unsigned)(~(unsigned)i);
        best = e > best ? e : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o, 64);
        best = other > best ? other : best;
    }
    if (lane == 0) {
        vals[gid] = KeyOf<T>::value((unsigned)(best >> 32));
        idx_out[gid] = (int64_t)g * width + (int64_t)(~(unsigned)best);
    }
}

// Multi-top-k (sparse_coder.py:256-262): the wider selection contains the narrower one of the same
// row, whose rows the main decode has already summed. A wave per token: the narrower selection's
// last member in the order both selections are cut from (value descending, lowest index first) is
// the minimum of (key << 32 | ~index) over its k entries; a wider entry at or above it is a member.
template <typename T>
__global__ __launch_bounds__(256) void exclude_selected_kernel(const T* __restrict__ vals_wide,
                                                               const int64_t* __restrict__ idx_wide, int kw,
                                                               const T* __restrict__ vals_sel,
                                                               const int64_t* __restrict__ idx_sel, int k, int64_t N,
                                                               int64_t* __restrict__ idx_out) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    unsigned long long last = ~0ull;
    for (int s = lane; s < k; s += 64) {
        const int64_t i = idx_sel[n * k + s];
        if (i < 0) continue;                       // an inactive slot selects nothing
        const unsigned long long r = ((unsigned long long)to_key(vals_sel[n * k + s]) << 32) | (unsigned)(~(unsigned)i);
        last = r < last ? r : last;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(last, o, 64);
        last = other < last ? other : last;
    }
    for (int s = lane; s < kw; s += 64) {
        const int64_t i = idx_wide[n * kw + s];
        const unsigned long long r = ((unsigned long long)to_key(vals_wide[n * kw + s]) << 32) | (unsigned)(~(unsigned)i);
        idx_out[n * kw + s] = (i >= 0 && r >= last) ? -1 : i;
    }
}

}  // namespace sphip

namespace sphip {
static int topk_impl(const int* run_if, const void* pre, int dtype, const uint8_t* keep_mask, const void* tile_max,
                     int64_t N, int64_t M, int64_t k, void* vals, int64_t* idx, void* stream);
}

extern "C" int sphip_topk(const void* pre, int dtype, const uint8_t* keep_mask, const void* tile_max, int64_t N,
                          int64_t M, int64_t k, void* vals, int64_t* idx, void* stream) {
    return sphip::topk_impl(nullptr, pre, dtype, keep_mask, tile_max, N, M, k, vals, idx, stream);
}

extern "C" int sphip_topk_if(const int* run_if, const void* pre, int dtype, const uint8_t* keep_mask,
                             const void* tile_max, int64_t N, int64_t M, int64_t k, void* vals, int64_t* idx,
                             void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(run_if != nullptr && ((uintptr_t)run_if & 3) == 0, SPHIP_EINVAL, "topk_if: null or unaligned predicate");
    return topk_impl(run_if, pre, dtype, keep_mask, tile_max, N, M, k, vals, idx, stream);
}

static int sphip::topk_impl(const int* run_if, const void* pre, int dtype, const uint8_t* keep_mask,
                            const void* tile_max, int64_t N, int64_t M, int64_t k, void* vals, int64_t* idx,
                            void* stream) {
    SPHIP_REQUIRE(valid_dtype(dtype), SPHIP_EINVAL, "topk: bad dtype %d", dtype);
    SPHIP_REQUIRE(pre && vals && idx, SPHIP_EINVAL, "topk: null pointer");
    SPHIP_REQUIRE(N > 0 && M > 0, SPHIP_EINVAL, "topk: empty shape N=%lld M=%lld", (long long)N, (long long)M);
    SPHIP_REQUIRE(k > 0 && k <= M, SPHIP_EINVAL, "topk: k=%lld out of range for M=%lld (selected index out of range)",
                  (long long)k, (long long)M);
    SPHIP_REQUIRE(M < (1ll << 31) - 65536 && N < (1ll << 31), SPHIP_EUNSUPPORTED, "topk: shape exceeds int32");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)N), block(TK_THREADS);
    const int vec = dtype == SPHIP_BF16 ? 8 : 4;
    const bool aligned = (M % vec == 0) && (((uintptr_t)pre & 15) == 0) && (((uintptr_t)keep_mask & 7) == 0);
    // tile path with few selected latents: four waves per row (the kernel's own by_tile rule
    // holds for these shapes; SPHIP_TOPK_WIDE=1 keeps the 1024-thread form, A/B switch)
    const int64_t T_ = cdiv(M, (int64_t)TK_TILE);
    const bool wide_only = getenv("SPHIP_TOPK_WIDE") != nullptr;
    const bool small = aligned && tile_max != nullptr && keep_mask == nullptr && k <= TK_THREADS_S && T_ >= k &&
                       T_ <= 16 * TK_THREADS_S && !(k >= 256 && M <= 16 * k) && !wide_only;
    // the kernel's exact path, taken for a short row with a large k: from registers when the row fits
    const bool resident = aligned && keep_mask == nullptr && (k > 3 * TK_THREADS || (k >= 256 && M <= 16 * k)) &&
                          M <= 2 * vec * TK_THREADS && getenv("SPHIP_TOPK_NO_RESIDENT") == nullptr;
    if (resident) {
        dim3 rblock(TK_THREADS);
        if (dtype == SPHIP_BF16)
            hipLaunchKernelGGL((topk_resident_kernel<bf16_t>), grid, rblock, 0, s, (const bf16_t*)pre, (int)M, (int)k, (bf16_t*)vals, idx, run_if);
        else
            hipLaunchKernelGGL((topk_resident_kernel<float>), grid, rblock, 0, s, (const float*)pre, (int)M, (int)k, (float*)vals, idx, run_if);
    } else if (small) {
        dim3 sblock(TK_THREADS_S);
        if (dtype == SPHIP_BF16)
            hipLaunchKernelGGL((topk_kernel<bf16_t, true, TK_THREADS_S, TK_CAP_S>), grid, sblock, 0, s, (const bf16_t*)pre, keep_mask, (const bf16_t*)tile_max, (int)M, (int)k, (bf16_t*)vals, idx, run_if);
        else
            hipLaunchKernelGGL((topk_kernel<float, true, TK_THREADS_S, TK_CAP_S>), grid, sblock, 0, s, (const float*)pre, keep_mask, (const float*)tile_max, (int)M, (int)k, (float*)vals, idx, run_if);
    } else if (dtype == SPHIP_BF16) {
        if (aligned)
            hipLaunchKernelGGL((topk_kernel<bf16_t, true>), grid, block, 0, s, (const bf16_t*)pre, keep_mask, (const bf16_t*)tile_max, (int)M, (int)k, (bf16_t*)vals, idx, run_if);
        else
            hipLaunchKernelGGL((topk_kernel<bf16_t, false>), grid, block, 0, s, (const bf16_t*)pre, keep_mask, (const bf16_t*)tile_max, (int)M, (int)k, (bf16_t*)vals, idx, run_if);
    } else {
        if (aligned)
            hipLaunchKernelGGL((topk_kernel<float, true>), grid, block, 0, s, (const float*)pre, keep_mask, (const float*)tile_max, (int)M, (int)k, (float*)vals, idx, run_if);
        else
            hipLaunchKernelGGL((topk_kernel<float, false>), grid, block, 0, s, (const float*)pre, keep_mask, (const float*)tile_max, (int)M, (int)k, (float*)vals, idx, run_if);
    }
    SPHIP_CHECK_LAUNCH("topk");
    return SPHIP_OK;
}

extern "C" int sphip_groupmax(const void* pre, int dtype, int64_t N, int64_t M, int64_t k, void* vals, int64_t* idx,
                              void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(valid_dtype(dtype), SPHIP_EINVAL, "groupmax: bad dtype %d", dtype);
    SPHIP_REQUIRE(pre && vals && idx, SPHIP_EINVAL, "groupmax: null pointer");
    SPHIP_REQUIRE(N > 0 && M > 0 && k > 0 && k <= M, SPHIP_EINVAL, "groupmax: bad shape N=%lld M=%lld k=%lld",
                  (long long)N, (long long)M, (long long)k);
    SPHIP_REQUIRE(M % k == 0, SPHIP_EINVAL, "groupmax: M=%lld is not divisible by k=%lld", (long long)M, (long long)k);
    SPHIP_REQUIRE(M < (1ll << 31), SPHIP_EUNSUPPORTED, "groupmax: M exceeds int32");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = N * k;
    SPHIP_REQUIRE(cdiv(total, 4) < (1ll << 31), SPHIP_EUNSUPPORTED, "groupmax: grid too large");
    dim3 grid((unsigned)cdiv(total, 4)), block(256);
    if (dtype == SPHIP_BF16)
        hipLaunchKernelGGL((groupmax_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)pre, (int)M, (int)k, (int)(M / k), total, (bf16_t*)vals, idx);
    else
        hipLaunchKernelGGL((groupmax_kernel<float>), grid, block, 0, s, (const float*)pre, (int)M, (int)k, (int)(M / k), total, (float*)vals, idx);
    SPHIP_CHECK_LAUNCH("groupmax");
    return SPHIP_OK;
}

extern "C" int sphip_exclude_selected(const void* vals_wide, const int64_t* idx_wide, int64_t kw, const void* vals_sel,
                                      const int64_t* idx_sel, int64_t k, int dtype, int64_t N, int64_t* idx_out,
                                      void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(valid_dtype(dtype), SPHIP_EINVAL, "exclude_selected: bad dtype %d", dtype);
    SPHIP_REQUIRE(vals_wide && idx_wide && vals_sel && idx_sel && idx_out, SPHIP_EINVAL, "exclude_selected: null pointer");
    SPHIP_REQUIRE(N > 0 && k > 0 && kw >= k && kw < (1ll << 31), SPHIP_EINVAL, "exclude_selected: bad shape N=%lld k=%lld kw=%lld",
                  (long long)N, (long long)k, (long long)kw);
    SPHIP_REQUIRE(cdiv(N, 4) < (1ll << 31), SPHIP_EUNSUPPORTED, "exclude_selected: grid too large");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)cdiv(N, 4)), block(256);
    if (dtype == SPHIP_BF16)
        hipLaunchKernelGGL((exclude_selected_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)vals_wide, idx_wide, (int)kw,
                           (const bf16_t*)vals_sel, idx_sel, (int)k, N, idx_out);
    else
        hipLaunchKernelGGL((exclude_selected_kernel<float>), grid, block, 0, s, (const float*)vals_wide, idx_wide, (int)kw,
                           (const float*)vals_sel, idx_sel, (int)k, N, idx_out);
    SPHIP_CHECK_LAUNCH("exclude_selected");
    return SPHIP_OK;
}
