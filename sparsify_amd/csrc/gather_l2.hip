// L2-resident sparse decode for tables far larger than the caches:
//   out[n,:] = sum_s coeff[n,s] * table[idx[n,s],:]  (+ bias)
// Reference: decoder_impl / eager_decode, sparsify/utils.py:81-90; embedding_bag_k,
// sparsify/xformers.py:9-53 (same contract as sphip_gather_rows_weighted, gather.hip).
//
// The token-major kernels of gather.hip read k random 1-KiB row slices per (token, slice) out
// of a 128 MiB column slice that only the Infinity Cache holds: 7.5 TB/s, that cache's ceiling
// for random rows. Here the working set of the moment is made to fit the 4 MiB L2 of an XCD:
// the latents are cut into B blocks of BLOCK_ROWS rows, and a launch handles ONE latent block
// for all tokens and all 128-byte column slices, the slices dealt to the XCDs, so that an XCD
// gathers from BLOCK_ROWS x 128 B (4 MiB) of the table at a time. B launches follow each other
// on the stream; the first writes out = bias + partial sum, the others add theirs, so the
// sum has a fixed order (block by block, slots in order inside a block) and needs no atomics.
//
//   bucket_kernel<false>: one wave per token counts its slots per latent block
//   exclusive scan over the (block, token) counts -> start of every (block, token) list
//   bucket_kernel<true> : writes {latent, coeff} of every slot to its list (slot order kept)
//   gather_l2_kernel    : 8 lanes per (token, 128-B slice); the 32 tokens of a workgroup own one
//                         contiguous range of the block's list, staged in LDS by coalesced
//                         loads; four row pieces in flight per lane group, full batches
//                         without per-entry guards (as seg_wgrad_sliced_kernel, wgrad.hip)
#include "common.hpp"
#include <stdlib.h>
#include <type_traits>

namespace sphip {

void launch_exclusive_scan(unsigned* a, int64_t n, unsigned* chunk_tot, hipStream_t s);   // segments.hip

// Latents per block. A block's 128-byte row pieces of one slice are what an XCD gathers from at a
// time: 32768 rows = 4 MiB, the whole L2 -- more than fits beside the lists and the output, yet
// measured fastest at the headline shape (make EXTRA=-DSPHIP_L2_BLOCK_ROWS=n, one box, M = 131072):
// 12288 rows 3.68 ms, 16384 3.28, 20480 3.21, 24576 3.09, 28672 3.04, 32768 3.04, 43776 3.87.
// Larger blocks mean fewer launches, each of which re-reads and re-writes the whole output
// (four passes instead of eight), and longer lists per (token, block), i.e. fuller batches.
#ifndef SPHIP_L2_BLOCK_ROWS
#define SPHIP_L2_BLOCK_ROWS 32768
#endif
constexpr int L2_BLOCK_ROWS = SPHIP_L2_BLOCK_ROWS;
constexpr int L2_MAX_BLOCKS = 32;
#ifndef SPHIP_L2_STAGE
#define SPHIP_L2_STAGE 2048
#endif
constexpr int L2_STAGE = SPHIP_L2_STAGE;         // staged list entries per workgroup (16 KiB with 8-byte entries)

struct L2Plan {
    int blocks;
    size_t off_start, off_spine, off_list, total;
};

static L2Plan l2_plan(int64_t N, int64_t k, int64_t M) {
    L2Plan p;
    p.blocks = (int)cdiv(M, L2_BLOCK_ROWS);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    const int64_t cells = (int64_t)p.blocks * N + 1;
    p.off_start = take((size_t)cells * 4);
    p.off_spine = take((size_t)cdiv(cells, 4096) * 4 + 4);
    p.off_list = take((size_t)(N * k) * 8);
    p.total = o;
    return p;
}

#ifndef SPHIP_L2_MIN_USES
// uses of a table row per launch below which the token-major form is taken. Headline widths, one box, L2 form
// (+ 0.5 ms for the slice-major copy the decoder's step then writes) against the token-major form:
// 6144 tokens 1.35 (+0.5) / 2.53 ms, 4096 tokens 1.01 (+0.5) / 1.70, 3072 tokens 0.83 (+0.5) / 1.28
#define SPHIP_L2_MIN_USES 6
#endif
static bool l2_eligible(int64_t N, int64_t k, int64_t M, int64_t D) {
    if (D % 32 != 0 || N * k >= (1ll << 31) || M >= (1ll << 31) || N >= (1ll << 26)) return false;
    const int64_t B = cdiv(M, L2_BLOCK_ROWS);
    // worth it when the table is beyond the Infinity Cache, every (token, block) list still
    // holds a few entries, and every table row is used several times per launch
    return B >= 2 && B <= L2_MAX_BLOCKS && M * D * 4 > (512ll << 20) && k >= 8 * B && N * k >= SPHIP_L2_MIN_USES * M;
}

// One wave per token. SCATTER = false: start[b * N + n] = number of the token's slots whose
// latent lies in block b. SCATTER = true (start now holds the scanned offsets): the slots are
// written to their lists, in slot order. Slots with a negative or out-of-range index
// (inactive: latent owned by another shard) belong to no block.
// PK4 (bf16 coefficients, blocks of at most 65536 rows): an entry is 4 bytes, the row inside its block
// in the upper half and the coefficient's bf16 bits in the lower -- the lists are re-read once per
// column slice (128 times at D = 4096), so their size is L2 fill traffic of the gather kernel.
template <typename CT, bool SCATTER, bool PK4 = false>
__global__ __launch_bounds__(256) void bucket_kernel(const int64_t* __restrict__ idx, const CT* __restrict__ coeff,
                                                     int64_t N, int k, int64_t M, int B, unsigned* __restrict__ start,
                                                     uint2* __restrict__ list) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    unsigned mine = 0;                                   // lane b: count / running offset of block b
    if (SCATTER && lane < B) mine = start[(int64_t)lane * N + n];
    for (int s0 = 0; s0 < k; s0 += 64) {
        const int s = s0 + lane;
        int64_t j = -1;
        if (s < k) j = idx[n * k + s];
        const int bl = (j < 0 || j >= M) ? -1 : (int)(j / L2_BLOCK_ROWS);
        unsigned pos = 0;
        for (int b = 0; b < B; ++b) {
            const unsigned long long m = __ballot(bl == b);
            if (SCATTER) {
                const unsigned base = (unsigned)__shfl((int)mine, b, 64);
                if (bl == b) pos = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            }
            if (lane == b) mine += (unsigned)__popcll(m);
        }
        if (SCATTER && bl >= 0) {
            if constexpr (PK4) {
                reinterpret_cast<unsigned*>(list)[pos] = ((unsigned)(j - (int64_t)bl * L2_BLOCK_ROWS) << 16) |
                                                         (unsigned)reinterpret_cast<const unsigned short*>(coeff)[n * k + s];
            } else {
                list[pos] = make_uint2((unsigned)j, __float_as_uint(Elem<CT>::load(coeff, n * k + s)));
            }
        }
    }
    if (!SCATTER && lane < B) start[(int64_t)lane * N + n] = mine;
}

// SM: `table` is the slice-major copy [D/32][M][32] (sphip_slice_major_f32): the row pieces of a
// slice are then contiguous. In the row-major table they lie D*4 bytes apart, a power-of-two
// stride that maps a latent block's pieces onto a few L2 sets (measured: L2 hit rate 27 %, the
// gather then runs at the Infinity Cache's rate like the token-major kernels).
// BIG: four times the staging area, for lists that would not fit otherwise (32 k / B entries per
// workgroup: the 4k-row decode of multi-top-k); it costs occupancy, so short lists keep the small one
// (headline decode 2.92 ms small / 3.17 ms big; multi-top-k's 4k decode 14.7 ms small / 9.5 ms big).
template <bool FIRST, bool SM, bool PK4 = false, bool BIG = false>
__global__ __launch_bounds__(256) void gather_l2_kernel(const uint2* __restrict__ list,
                                                        const unsigned* __restrict__ start,   // of this block
                                                        const float* __restrict__ table, const float* __restrict__ bias,
                                                        float* __restrict__ out, int N, int D, int64_t M,
                                                        int groups_per_slice, int num_slices, unsigned row0 = 0) {
    // XCD-aware order: the workgroups of one XCD (same blockIdx % 8) walk slice after slice
    const unsigned xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const unsigned slice = xcd + 8 * (local / (unsigned)groups_per_slice);
    if (slice >= (unsigned)num_slices) return;
    const int n0 = (int)(local % (unsigned)groups_per_slice) * 32;
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 3, sub = lane & 7;
    // The workgroup's 32 tokens are handed to the lane groups in order of list length (wave 0
    // takes the 8 longest lists of this latent block): a wave lives as long as its longest list,
    // and the lists of a (token, block) pair vary like a binomial around k / B. The sum of a
    // token is unchanged (one lane group, list order).
    __shared__ int sh_sb[33];
    __shared__ int sh_ord[32];
    if (threadIdx.x < 33) sh_sb[threadIdx.x] = (int)start[min(n0 + (int)threadIdx.x, N)];
    __syncthreads();
    {
        const int i = threadIdx.x >> 3;                                // 8 lanes rank token i against 4 others each
        const int mylen = sh_sb[i + 1] - sh_sb[i];
        int rank = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = sub * 4 + q, l = sh_sb[o + 1] - sh_sb[o];
            rank += (l > mylen || (l == mylen && o < i)) ? 1 : 0;
        }
        rank += __builtin_amdgcn_update_dpp(0, rank, 0xB1, 0xF, 0xF, true);
        rank += __builtin_amdgcn_update_dpp(0, rank, 0x4E, 0xF, 0xF, true);
        rank += __builtin_amdgcn_update_dpp(0, rank, 0x141, 0xF, 0xF, true);
        if (sub == 0) sh_ord[rank] = i;
    }
    __syncthreads();
    const int nl = sh_ord[(threadIdx.x >> 6) * 8 + grp];
    const int n = n0 + nl;
    const int col = (int)slice * 32 + sub * 4;
    const bool live = n < N;
    int beg = 0, end = 0;
    if (live) { beg = sh_sb[nl]; end = sh_sb[nl + 1]; }

    typedef typename std::conditional<PK4, unsigned, uint2>::type Entry;
    const Entry* elist = reinterpret_cast<const Entry*>(list);
    constexpr int STAGE = BIG ? 4 * L2_STAGE : L2_STAGE;
    __shared__ Entry sh[STAGE];
    const int rb = sh_sb[0], re = sh_sb[32];                             // block-uniform
    const bool staged = re - rb <= STAGE;
    if (staged) {
        for (int i = threadIdx.x; i < re - rb; i += 256) sh[i] = elist[rb + i];
        __syncthreads();
    }
    auto row_of = [&](Entry e) -> unsigned {
        if constexpr (PK4) return row0 + (e >> 16);
        else return e.x;
    };
    auto coeff_of = [&](Entry e) -> float {
        if constexpr (PK4) return __uint_as_float(e << 16);
        else return __uint_as_float(e.y);
    };
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 prev = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        if (FIRST) {
            if (bias != nullptr) prev = *reinterpret_cast<const f4*>(bias + col);
        } else {
            prev = __builtin_nontemporal_load(reinterpret_cast<const f4*>(out + (int64_t)n * D + col));
        }
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // piece of latent j: tcol + j * stride
    const float* tcol = SM ? table + ((int64_t)slice * M) * 32 + sub * 4 : table + col;
    const int64_t stride = SM ? 32 : D;
    // pieces in flight per lane group (make EXTRA=-DSPHIP_L2_U=n, 32768-row blocks): 3 -> 3.04 ms, 4 -> 3.03, 6 -> 2.99, 8 -> 3.10
#ifndef SPHIP_L2_U
#define SPHIP_L2_U 4
#endif
    constexpr int U = SPHIP_L2_U;
    auto run = [&](auto entry) {
        const int last = max(end - 1, 0);
        Entry nxt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) nxt[u] = entry(min(beg + u, last));
        int p = beg;
        for (; p + U <= end; p += U) {
            Entry pk[U];
            float4 r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) pk[u] = nxt[u];
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = load4(tcol + (int64_t)row_of(pk[u]) * stride);
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[u] = entry(min(p + U + u, last));
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float cf = coeff_of(pk[u]);
                acc[0] = fmaf(cf, r[u].x, acc[0]);
                acc[1] = fmaf(cf, r[u].y, acc[1]);
                acc[2] = fmaf(cf, r[u].z, acc[2]);
                acc[3] = fmaf(cf, r[u].w, acc[3]);
            }
        }
        if (p < end) {
            float4 r[U];
#pragma unroll
            for (int u = 0; u < U - 1; ++u) {
                if (p + u < end) r[u] = load4(tcol + (int64_t)row_of(nxt[u]) * stride);
            }
#pragma unroll
            for (int u = 0; u < U - 1; ++u) {
                if (p + u < end) {
                    const float cf = coeff_of(nxt[u]);
                    acc[0] = fmaf(cf, r[u].x, acc[0]);
                    acc[1] = fmaf(cf, r[u].y, acc[1]);
                    acc[2] = fmaf(cf, r[u].z, acc[2]);
                    acc[3] = fmaf(cf, r[u].w, acc[3]);
                }
            }
        }
    };
    if (end > beg) {
        if (staged) run([&](int pos) { return sh[pos - rb]; });
        else run([&](int pos) { return elist[pos]; });
    }
    if (live) {
        const f4 o = {prev[0] + acc[0], prev[1] + acc[1], prev[2] + acc[2], prev[3] + acc[3]};
        __builtin_nontemporal_store(o, reinterpret_cast<f4*>(out + (int64_t)n * D + col));
    }
}

}  // namespace sphip

extern "C" size_t sphip_gather_l2_workspace_bytes(int64_t N, int64_t k, int64_t M, int64_t D) {
    using namespace sphip;
    if (N <= 0 || k <= 0 || M <= 0 || D <= 0 || !l2_eligible(N, k, M, D)) return 0;
    return l2_plan(N, k, M).total;
}

extern "C" int sphip_gather_rows_weighted_l2(const int64_t* idx, const void* coeff, int coeff_dtype, const float* table,
                                             const float* table_slice_major, const float* bias, float* out, int64_t N,
                                             int64_t k, int64_t M, int64_t D,
                                             void* workspace, size_t workspace_bytes, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(valid_dtype(coeff_dtype), SPHIP_EINVAL, "gather_rows_weighted_l2: bad coeff dtype %d", coeff_dtype);
    SPHIP_REQUIRE(idx && coeff && table && out && workspace, SPHIP_EINVAL, "gather_rows_weighted_l2: null pointer");
    SPHIP_REQUIRE(N > 0 && k > 0 && M > 0 && D > 0, SPHIP_EINVAL, "gather_rows_weighted_l2: empty shape");
    SPHIP_REQUIRE(l2_eligible(N, k, M, D), SPHIP_EUNSUPPORTED,
                  "gather_rows_weighted_l2: shape N=%lld k=%lld M=%lld D=%lld is not eligible "
                  "(sphip_gather_l2_workspace_bytes returns 0 for it): use sphip_gather_rows_weighted",
                  (long long)N, (long long)k, (long long)M, (long long)D);
    SPHIP_REQUIRE((((uintptr_t)table | (uintptr_t)table_slice_major | (uintptr_t)out | (uintptr_t)bias) & 15) == 0,
                  SPHIP_EINVAL, "gather_rows_weighted_l2: table, bias and out must be 16-byte aligned");
    const L2Plan p = l2_plan(N, k, M);
    SPHIP_REQUIRE(workspace_bytes >= p.total, SPHIP_EWORKSPACE, "gather_rows_weighted_l2: workspace %zu < %zu bytes",
                  workspace_bytes, p.total);
    SPHIP_REQUIRE(((uintptr_t)workspace & 255) == 0, SPHIP_EINVAL, "gather_rows_weighted_l2: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    unsigned* start = (unsigned*)(ws + p.off_start);
    unsigned* spine = (unsigned*)(ws + p.off_spine);
    uint2* list = (uint2*)(ws + p.off_list);
    const int64_t cells = (int64_t)p.blocks * N + 1;
    // the cell after the last (block, token) pair counts nothing: after the scan it holds the total
    SPHIP_REQUIRE(hipMemsetAsync(start + cells - 1, 0, 4, s) == hipSuccess, SPHIP_ELAUNCH,
                  "gather_rows_weighted_l2: memset failed");
    dim3 tgrid((unsigned)cdiv(N, 4)), block(256);
    if (coeff_dtype == SPHIP_F32)
        hipLaunchKernelGGL((bucket_kernel<float, false>), tgrid, block, 0, s, idx, (const float*)coeff, N, (int)k, M, p.blocks, start, list);
    else
        hipLaunchKernelGGL((bucket_kernel<bf16_t, false>), tgrid, block, 0, s, idx, (const bf16_t*)coeff, N, (int)k, M, p.blocks, start, list);
    launch_exclusive_scan(start, cells, spine, s);
    const bool pk4 = coeff_dtype == SPHIP_BF16 && L2_BLOCK_ROWS <= 65536 && getenv("SPHIP_L2_PK8") == nullptr;
    if (coeff_dtype == SPHIP_F32)
        hipLaunchKernelGGL((bucket_kernel<float, true>), tgrid, block, 0, s, idx, (const float*)coeff, N, (int)k, M, p.blocks, start, list);
    else if (pk4)
        hipLaunchKernelGGL((bucket_kernel<bf16_t, true, true>), tgrid, block, 0, s, idx, (const bf16_t*)coeff, N, (int)k, M, p.blocks, start, list);
    else
        hipLaunchKernelGGL((bucket_kernel<bf16_t, true>), tgrid, block, 0, s, idx, (const bf16_t*)coeff, N, (int)k, M, p.blocks, start, list);
    const int num_slices = (int)(D / 32);
    const int gps = (int)cdiv(N, 32);
    const int64_t nblocks = cdiv(num_slices, 8) * gps * 8;
    SPHIP_REQUIRE(nblocks < (1ll << 31), SPHIP_EUNSUPPORTED, "gather_rows_weighted_l2: grid too large");
    const bool sm = table_slice_major != nullptr;
    const float* tb = sm ? table_slice_major : table;
    // a workgroup's 32 tokens bring about 32 k / B entries: beyond the small staging area, take the big one
    const bool big_stage = (32 * k / p.blocks) * 5 / 4 > L2_STAGE;
    for (int b = 0; b < p.blocks; ++b) {
        const unsigned* st = start + (int64_t)b * N;
        dim3 grid((unsigned)nblocks);
        const unsigned row0 = (unsigned)((int64_t)b * L2_BLOCK_ROWS);
#define SPHIP_L2_LAUNCH(F, S) \
    do { \
        if (pk4 && big_stage) hipLaunchKernelGGL((gather_l2_kernel<F, S, true, true>), grid, block, 0, s, list, st, tb, bias, out, (int)N, (int)D, M, gps, num_slices, row0); \
        else if (pk4) hipLaunchKernelGGL((gather_l2_kernel<F, S, true>), grid, block, 0, s, list, st, tb, bias, out, (int)N, (int)D, M, gps, num_slices, row0); \
        else hipLaunchKernelGGL((gather_l2_kernel<F, S, false>), grid, block, 0, s, list, st, tb, bias, out, (int)N, (int)D, M, gps, num_slices, row0); \
    } while (0)
        if (b == 0 && sm) SPHIP_L2_LAUNCH(true, true);
        else if (b == 0) SPHIP_L2_LAUNCH(true, false);
        else if (sm) SPHIP_L2_LAUNCH(false, true);
        else SPHIP_L2_LAUNCH(false, false);
#undef SPHIP_L2_LAUNCH
    }
    SPHIP_CHECK_LAUNCH("gather_rows_weighted_l2");
    return SPHIP_OK;
}
