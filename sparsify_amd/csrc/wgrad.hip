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
#include <type_traits>
#include <stdlib.h>

namespace sphip {

// timing experiments on the fused update (make EXTRA=-DSPHIP_FUSE_X=n): bit 1 no parameter store, 2 no shadow store
#ifndef SPHIP_FUSE_X
#define SPHIP_FUSE_X 0
#endif
// The fused (optimizer-in-kernel) form of the L2-sliced kernel holds the parameter piece across
// the segment loop and needs ~82 registers with four pieces in flight: five waves per SIMD. It is
// compiled for SPHIP_FUSE_WAVES waves (register budget 512 / waves) with SPHIP_FUSE_U pieces in
// flight per lane group; the other forms keep the compiler's own allocation (56-66 registers;
// forcing the dot form from 66 to 60 registers for an eighth wave measured 4.33 against 4.27 ms).
// A/B on one box (make EXTRA="-DSPHIP_FUSE_WAVES=w -DSPHIP_FUSE_U=u"): 5 waves x 4 pieces 2.65 ms,
// 6 x 4 2.42, 7 x 3 2.38, 8 x 2 2.37 (7 x 4 and 8 x 4 spill: 2.46 / 2.81). The schedule-free form
// (the z piece on top) is compiled for one wave less.
// (with 4-byte list entries, PK4, the kernel needs four registers fewer and is compiled for one wave more)
#ifndef SPHIP_FUSE_WAVES
#define SPHIP_FUSE_WAVES 7
#endif
#ifndef SPHIP_FUSE_U
#define SPHIP_FUSE_U 3
#endif

// Segment-ordered entry list: packed[p] = {entry id e = perm[p], fp32 bits of coeff[e]}.
// Turns the per-entry dependent gather (perm -> coeff) of the reduction kernels into one
// sequential 8-byte stream; built once per call (N*k entries, a few tens of microseconds).
// packed4 (bf16 coefficients, N <= 65536; may be null): the same entries as 4 bytes, token in the
// upper half and the coefficient's bf16 bits in the lower, for the L2-sliced kernel with dots, which
// re-reads the list once per column slice.
template <typename CT>
__global__ __launch_bounds__(256) void seg_pack_kernel(const unsigned* __restrict__ perm,
                                                       const CT* __restrict__ coeff, int64_t E,
                                                       const int* __restrict__ seg_begin, int M,
                                                       uint2* __restrict__ packed, float* __restrict__ dcoeff,
                                                       unsigned* __restrict__ packed4 = nullptr,
                                                       FastDiv kdiv = FastDiv{}) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p < E) {
        const unsigned e = perm[p];
        packed[p] = make_uint2(e, __float_as_uint(Elem<CT>::load(coeff, e)));
        if (sizeof(CT) == 2 && packed4 != nullptr)
            packed4[p] = (fastdiv(e, kdiv) << 16) | (unsigned)reinterpret_cast<const unsigned short*>(coeff)[e];
        // entries past seg_begin[M] are inactive slots: no segment covers them
        if (dcoeff != nullptr && p >= seg_begin[M]) dcoeff[e] = 0.f;
    }
}

template <typename RT, int NV, bool DOT>
__global__ __launch_bounds__(256) void seg_wgrad_kernel(const uint2* __restrict__ packed,
                                                        const int* __restrict__ seg_begin,
                                                        const RT* __restrict__ rows,
                                                        const float* __restrict__ W, float* __restrict__ gW,
                                                        float* __restrict__ gbias, float* __restrict__ dcoeff,
                                                        const float* __restrict__ row_shift, FastDiv kdiv, int M,
                                                        int D) {
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
            const uint2 pk = packed[p];
            const unsigned e = pk.x;
            const float c = __uint_as_float(pk.y);
            const RT* rp = rows + (int64_t)fastdiv(e, kdiv) * D + c0 + lane * 4;
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
            if (!live[v]) continue;
            if (row_shift != nullptr) {
                // rows were given un-shifted: sum c*(row - shift) = sum c*row - (sum c)*shift
                const float4 sh = *reinterpret_cast<const float4*>(row_shift + col);
                acc[v].x = fmaf(-bsum, sh.x, acc[v].x);
                acc[v].y = fmaf(-bsum, sh.y, acc[v].y);
                acc[v].z = fmaf(-bsum, sh.z, acc[v].z);
                acc[v].w = fmaf(-bsum, sh.w, acc[v].w);
            }
            *reinterpret_cast<float4*>(gW + (int64_t)j * D + col) = acc[v];
        }
        if (gbias != nullptr && c0 == 0 && lane == 0) gbias[j] = bsum;
    }
}

// ---------------------------------------------------------------------------------------
// Column-sliced variant (no dcoeff): the gradient is computed one 128-byte column slice of
// `rows` at a time. A slice of the whole token batch (N x 128 B, 2 MiB at N = 16384) fits
// one XCD's 4 MiB L2, and all workgroups that land on one XCD (blockIdx % 8) work through
// the same slice before moving on, so the k-fold re-reads of every row piece are L2 hits
// instead of Infinity-Cache / HBM traffic. Eight lanes cover one 128-B row piece; each
// 8-lane group owns one latent, so a wave reduces 8 latents at once, sequentially in
// segment order (bitwise the same sums as the full-row kernel), with 4 entries per group in
// flight. Outputs are written with non-temporal stores to keep the slice resident.
template <typename T> struct Piece;
template <> struct Piece<float> {
    static constexpr int COLS = 4;    // columns per lane (16 B)
    typedef float4 Raw;
    static __device__ __forceinline__ Raw load(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ Raw zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ float get(const Raw& r, int c) { return c == 0 ? r.x : c == 1 ? r.y : c == 2 ? r.z : r.w; }
};
// bf16 pieces stay as loaded (4 registers instead of 8 floats; four of them are in flight per lane)
// and are widened where they are consumed: one shift or mask per element
template <> struct Piece<bf16_t> {
    static constexpr int COLS = 8;
    typedef uint4 Raw;
    static __device__ __forceinline__ Raw load(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
    static __device__ __forceinline__ Raw zero() { return make_uint4(0u, 0u, 0u, 0u); }
    static __device__ __forceinline__ float get(const Raw& r, int c) {
        const unsigned w = (c >> 1) == 0 ? r.x : (c >> 1) == 1 ? r.y : (c >> 1) == 2 ? r.z : r.w;
        return __uint_as_float((c & 1) ? (w & 0xffff0000u) : (w << 16));
    }
};

// rows (N, D) -> slice-major copy [slice][n][128 B]. In the row-major layout the pieces of one
// slice are D*sizeof(RT) bytes apart (8 KiB at D = 4096 bf16): a power-of-two stride that maps
// the whole slice onto a handful of L2 sets. The slice-major copy makes every slice one
// contiguous N x 128 B block. One 16-B chunk per thread; columns past D are zero-filled.
template <typename RT>
__global__ __launch_bounds__(256) void slice_major_kernel(const RT* __restrict__ rows, uint4* __restrict__ out,
                                                          int64_t N, int D, int num_slices) {
    constexpr int C = Piece<RT>::COLS;
    const int64_t chunks_per_row = (int64_t)num_slices * 8;
    const int64_t total = N * chunks_per_row;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
        const int64_t n = i / chunks_per_row;
        const int ch = (int)(i % chunks_per_row);          // chunk inside the row: slice*8 + sub
        const int col = ch * C;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (col < D) v = *reinterpret_cast<const uint4*>(rows + n * D + col);
        out[((int64_t)(ch >> 3) * N + n) * 8 + (ch & 7)] = v;
    }
}

// FUSE: gW is the PARAMETER; instead of storing the gradient row piece the kernel applies
// SignSGD to it (sign_sgd.py:23: p -= lr * sign(g)) and refreshes the bf16 shadow, so the
// (M, D) gradient is neither written nor read back by an optimizer pass
// (sphip_segmented_weight_grad_signsgd).
// FUSE with bpart: the kernel also leaves, per workgroup, sum_j gb_in[j] * W_old[j, slice] over its 32
// latents (gb_in = the latents' coefficient sums, sphip_segment_coeff_sums): the column sums of
// these (M / 32, D) partials are gb^T W_enc, b_dec's encoder-side gradient (sparse_coder.py:192
// through fused_encoder.py:56-65), taken from the parameter pieces the update reads anyway instead
// of a pass of its own over the matrix. Order of the sum: the 8 latents of a wave by a butterfly
// over its lane groups, then the four waves in order -- fixed by the segment lengths, not by timing.
// DOT (fp32 rows): W is the table the rows were decoded with; the kernel also leaves, per slice,
// the partial dot product <rows[n, slice], W[j, slice]> of every entry in slab[slice][position]
// (8-lane butterfly per entry, lanes 0..3 of a group store the four entries of a batch as 16
// contiguous bytes); seg_dcoeff_bounded_kernel sums the slices in order.
#ifndef SPHIP_DOT_WAVES
#define SPHIP_DOT_WAVES 8
#endif
template <typename RT, bool FUSE = false, bool DOT = false, bool SFREE = false, bool PK4 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FUSE ? SPHIP_FUSE_WAVES + (PK4 ? 1 : 0) - (SFREE ? 1 : 0) : (DOT ? SPHIP_DOT_WAVES : 1), FUSE ? SPHIP_FUSE_WAVES + (PK4 ? 1 : 0) - (SFREE ? 1 : 0) : 8))) void seg_wgrad_sliced_kernel(const uint2* __restrict__ packed,
                                                               const int* __restrict__ seg_begin,
                                                               const RT* __restrict__ rows, float* __restrict__ gW,
                                                               float* __restrict__ gbias,
                                                               const float* __restrict__ row_shift, FastDiv kdiv,
                                                               int64_t N, int M, int D, int blocks_per_slice,
                                                               int num_slices, float lr = 0.f,
                                                               bf16_t* __restrict__ shadow = nullptr,
                                                               const int* __restrict__ skew = nullptr, int skew_T = 0,
                                                               const float* __restrict__ Wdot = nullptr,
                                                               float* __restrict__ slab = nullptr, int64_t E = 0,
                                                               SfArgs sf = SfArgs{nullptr, 0.f, 0.f, 0.f},
                                                               const float* __restrict__ gb_in = nullptr,
                                                               float* __restrict__ bpart = nullptr, int accum = 0) {
    constexpr int C = Piece<RT>::COLS;
    constexpr int SLICE_COLS = 8 * C;
    // a workgroup holds 32 latents for as long as its longest segment lasts: with `skew` set,
    // segments longer than skew_T are left to the wave-per-latent wide kernel launched next
    // to this one (seg_long_list_kernel lists them)
    // XCD-aware order: the workgroups of one XCD (same blockIdx % 8) walk slice after slice
    const unsigned xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const unsigned slice = xcd + 8 * (local / (unsigned)blocks_per_slice);
    if (slice >= (unsigned)num_slices) return;
    const unsigned lblock = local % (unsigned)blocks_per_slice;
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 3, sub = lane & 7;
    // Which of the workgroup's 32 consecutive latents a lane group reduces: a wave lives as long
    // as the longest of its 8 segments (about 31 entries when the mean is 24), so the latents are
    // handed out in order of segment length -- wave 0 takes the 8 longest, wave 3 the 8 shortest --
    // and a wave's lane groups finish together. Every latent is still summed by one lane group in
    // segment order: the sums do not change, only where they are computed.
    __shared__ int sh_sb[33];
    __shared__ int sh_ord[32];
    __shared__ float sh_wp[FUSE ? 4 : 1][FUSE ? SLICE_COLS : 1];     // per-wave partials of gb^T W (bpart)
    __shared__ int sh_cnt;                                           // waves that have left theirs
    if (threadIdx.x < 33) sh_sb[threadIdx.x] = seg_begin[min((int)lblock * 32 + (int)threadIdx.x, M)];
    if ((FUSE || DOT) && threadIdx.x == 0) sh_cnt = 0;
    __syncthreads();
#ifndef SPHIP_SLICED_SORT
#define SPHIP_SLICED_SORT 1
#endif
    if (SPHIP_SLICED_SORT) {
        auto eff_len = [&](int i) {
            const int l = sh_sb[i + 1] - sh_sb[i];
            return (skew != nullptr && l > skew_T) ? 0 : l;           // left to the wide kernel
        };
        const int i = threadIdx.x >> 3;                                // 8 lanes rank latent i against 4 others each
        const int mylen = eff_len(i);
        int rank = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = sub * 4 + q, l = eff_len(o);
            rank += (l > mylen || (l == mylen && o < i)) ? 1 : 0;
        }
        rank += __builtin_amdgcn_update_dpp(0, rank, 0xB1, 0xF, 0xF, true);
        rank += __builtin_amdgcn_update_dpp(0, rank, 0x4E, 0xF, 0xF, true);
        rank += __builtin_amdgcn_update_dpp(0, rank, 0x141, 0xF, 0xF, true);
        if (sub == 0) sh_ord[rank] = i;
    } else if (threadIdx.x < 32) {
        sh_ord[threadIdx.x] = threadIdx.x;
    }
    __syncthreads();
    const int jl = sh_ord[(threadIdx.x >> 6) * 8 + grp];
    const int j = (int)lblock * 32 + jl;
    const int col = (int)slice * SLICE_COLS + sub * C;
    bool live = (j < M) && (col < D);

    int beg = 0, end = 0;
    if (j < M) { beg = sh_sb[jl]; end = sh_sb[jl + 1]; }
    const bool mine = skew == nullptr || end - beg <= skew_T;
    if (!mine) { beg = end = 0; live = false; }
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
    float bsum = 0.f;
    // FUSE: the parameter piece is requested now and used after the segment loop
    float wold[FUSE ? C : 1];
    float zold[(FUSE && SFREE) ? C : 1];               // schedule-free form (sf.z != nullptr): the base iterate
    // bpart: the pieces of latents left to the wide kernel (not `mine`) are read too -- that kernel
    // updates them in a later launch on the same stream, so this one still sees the old values
    const bool inrange = (j < M) && (col < D);
    float gbj = 0.f;
    if (FUSE && bpart != nullptr && inrange) gbj = gb_in[j];
    if (FUSE && !live && bpart != nullptr && inrange) {
#pragma unroll
        for (int c = 0; c < C; c += 4) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(gW + (int64_t)j * D + col + c));
            wold[c] = t[0], wold[c + 1] = t[1], wold[c + 2] = t[2], wold[c + 3] = t[3];
        }
    }
    if (FUSE && live) {
#pragma unroll
        for (int c = 0; c < C; c += 4) {
            // streamed once: non-temporal, like the stores below, so that the parameter traffic
            // does not push the resident slice of `rows` out of L2
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(gW + (int64_t)j * D + col + c));
            wold[c] = t[0], wold[c + 1] = t[1], wold[c + 2] = t[2], wold[c + 3] = t[3];
            if (SFREE) {
                const f4 u = __builtin_nontemporal_load(reinterpret_cast<const f4*>(sf.z + (int64_t)j * D + col + c));
                zold[c] = u[0], zold[c + 1] = u[1], zold[c + 2] = u[2], zold[c + 3] = u[3];
            }
        }
    }
    float wdot[DOT ? C : 1];
    if (DOT) {
#pragma unroll
        for (int c = 0; c < C; ++c) wdot[c] = 0.f;
        if (live && end > beg) {                        // (a latent nobody selected has no dots to take)
#pragma unroll
            for (int c = 0; c < C; c += 4) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(Wdot + (int64_t)j * D + col + c));
                wdot[c] = t[0], wdot[c + 1] = t[1], wdot[c + 2] = t[2], wdot[c + 3] = t[3];
            }
        }
    }
    float* sl = DOT ? slab + (int64_t)slice * E : nullptr;
    // The 32 latents of a workgroup are consecutive, so their segments are one contiguous range
    // of `packed` (~32 N k / M entries): it is staged in LDS with coalesced loads, one round
    // trip for the whole workgroup, instead of every lane group fetching its entries 8 bytes at
    // a time from the Infinity Cache once per batch (the entry list is re-read for every
    // slice and does not stay in L2; that fetch, not the L2-resident row pieces, was what a
    // batch waited for). Ranges beyond the staging area (skewed latents) read global memory.
#ifndef SPHIP_SLICED_STAGE
#define SPHIP_SLICED_STAGE 2048
#endif
    constexpr int STAGE = PK4 ? 2 * SPHIP_SLICED_STAGE : SPHIP_SLICED_STAGE;   // entries (16 KiB either way)
    // PK4: `packed` points at 4-byte entries (token << 16 | bf16 coefficient, seg_pack_kernel)
    typedef typename std::conditional<PK4, unsigned, uint2>::type Entry;
    const Entry* plist = reinterpret_cast<const Entry*>(packed);
    __shared__ Entry sh_pk[STAGE];
    auto token_of = [&](Entry e) -> unsigned {
        if constexpr (PK4) return e >> 16;
        else return fastdiv(e.x, kdiv);
    };
    auto coeff_of = [&](Entry e) -> float {
        if constexpr (PK4) return __uint_as_float(e << 16);
        else return __uint_as_float(e.y);
    };
    const int rb = sh_sb[0], re = sh_sb[32];            // block-uniform
    const bool staged = re - rb <= STAGE;
#ifndef SPHIP_DOT_LDS
#define SPHIP_DOT_LDS 2
#endif
    // DOT with a staged list: an entry's partial dot is parked in the LDS slot of the entry itself
    // (read into registers a batch earlier, never read again) and the last wave to finish writes the
    // workgroup's whole range of the slab as contiguous lines, instead of every lane group storing
    // 16 bytes per batch (a sixth of the kernel's L2 requests)
    const bool park = DOT && SPHIP_DOT_LDS != 0 && staged;
    // partial dots of the batch at positions p .. p + U - 1 (entries past `stop` are not stored)
    typedef typename Piece<RT>::Raw Raw;
    auto dots = [&](const Raw* v, int p, int stop, int count) {
        float mine_d = 0.f;
#pragma unroll
        for (int u = 0; u < count; ++u) {
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) d = fmaf(Piece<RT>::get(v[u], c), wdot[c], d);
            // sum over the 8 lanes of the group with DPP moves (quad xor 1, quad xor 2, half-row
            // mirror) instead of three ds_bpermute round trips through the LDS pipe per entry; the
            // quad sums are the same bits in all four lanes (commutative adds), so lane 7 - i
            // delivers what lane i ^ 4 would: bit for bit the xor butterfly's result. A group's 8
            // lanes enter and leave the loops together, so no lane reads a disabled one.
            d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0xB1, 0xF, 0xF, true));
            d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x4E, 0xF, 0xF, true));
            d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x141, 0xF, 0xF, true));
            mine_d = (sub == u) ? d : mine_d;
        }
        if (sub < count && p + sub < stop) {
            if (park) {
                if constexpr (PK4) sh_pk[p + sub - rb] = __float_as_uint(mine_d);
                else sh_pk[p + sub - rb].x = __float_as_uint(mine_d);
            }
            else sl[p + sub] = mine_d;
        }
    };
    // `rows` is the slice-major copy: piece of token n = base + n * 128 B
    const RT* base = rows + ((int64_t)slice * N * 8 + sub) * C;

    if (staged) {
        for (int i = threadIdx.x; i < re - rb; i += 256) sh_pk[i] = plist[rb + i];
        __syncthreads();
    }
    // Software pipeline: the entries of batch b+1 are requested before the row pieces
    // of batch b are consumed, so a batch costs one L2 round trip instead of two. Full batches
    // run without per-entry guards (the lane groups of a wave differ in segment length, and
    // every guarded load is a branch on the exec mask): entry positions past the segment are
    // clamped to its last entry when they are prefetched and only the final partial batch checks.
    if (!live) beg = end = 0;
#ifndef SPHIP_SLICED_U
#define SPHIP_SLICED_U 4
#endif
    constexpr int U = FUSE ? SPHIP_FUSE_U : SPHIP_SLICED_U;
    auto run = [&](auto entry) {
        const int last = max(end - 1, 0);
        Entry nxt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) nxt[u] = entry(min(beg + u, last));
        int p = beg;
        for (; p + U <= end; p += U) {
            Entry pk[U];
            Raw v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) pk[u] = nxt[u];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = Piece<RT>::load(base + (int64_t)token_of(pk[u]) * (8 * C));
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[u] = entry(min(p + U + u, last));
            if (DOT) dots(v, p, end, U);
#pragma unroll
            for (int u = 0; u < U; ++u) {      // the exact sequential fma chain of the full-row kernel
                const float cf = coeff_of(pk[u]);
#pragma unroll
                for (int c = 0; c < C; ++c) acc[c] = fmaf(cf, Piece<RT>::get(v[u], c), acc[c]);
                bsum += cf;
            }
        }
        if (p < end) {
            Raw v[U];
#pragma unroll
            for (int u = 0; u < U - 1; ++u) {
                v[u] = Piece<RT>::zero();
                if (p + u < end) v[u] = Piece<RT>::load(base + (int64_t)token_of(nxt[u]) * (8 * C));
            }
            if (DOT) dots(v, p, end, U - 1);
#pragma unroll
            for (int u = 0; u < U - 1; ++u) {
                if (p + u < end) {
                    const float cf = coeff_of(nxt[u]);
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[c] = fmaf(cf, Piece<RT>::get(v[u], c), acc[c]);
                    bsum += cf;
                }
            }
        }
    };
    if (end > beg) {
        if (staged) run([&](int pos) { return sh_pk[pos - rb]; });
        else run([&](int pos) { return plist[pos]; });
    }
    if (live && row_shift != nullptr) {
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = fmaf(-bsum, row_shift[col + c], acc[c]);
    }
    if (DOT && park) {                                   // block-uniform
        // no workgroup barrier (the waves end at different times by design): every wave counts
        // itself when its dots are parked, the one that counts last stores the range. Slots of
        // latents left to the wide kernel still hold their entries: those positions of this slab
        // are written again, with their partial dots, by seg_long_chunk_kernel, which runs after this kernel.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        int prev = 0;
        if (lane == 0) prev = __hip_atomic_fetch_add(&sh_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        prev = __builtin_amdgcn_readfirstlane(prev);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (prev == 3) {
            for (int i = lane; i < re - rb; i += 64) {
                float d;
                if constexpr (PK4) d = __uint_as_float(sh_pk[i]);
                else d = __uint_as_float(sh_pk[i].x);
                if (SPHIP_DOT_LDS == 2) __builtin_nontemporal_store(d, sl + rb + i);
                else sl[rb + i] = d;
            }
        }
    }
    typedef float f4 __attribute__((ext_vector_type(4)));
    if constexpr (FUSE && C == 8) {
        // bf16 rows: a lane owns 8 consecutive columns, i.e. 32 bytes of the fp32 parameter row, so
        // two 16-byte stores per lane would each fill every other 16-byte chunk of two lines
        // (measured: the 2 GiB parameter store cost 0.40 ms of the kernel's 2.57, the 1 GiB shadow
        // store, whole lines, 0.14). The updated values change lanes inside the group first
        // (lane d takes columns 4d .. 4d+3 and 32+4d .. 32+4d+3) and leave as whole 128-byte lines.
        // Every lane of the wave takes part in the exchange; lanes that are not live store nothing.
        float wn[8], zn[SFREE ? 8 : 1];
        unsigned short shv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float g = acc[e];
            float y1 = wold[e], z1 = SFREE ? zold[e] : 0.f;
            if (SFREE) sf_update(y1, z1, (float)((g > 0.f) - (g < 0.f)), lr, sf.w_undo, sf.c, sf.w_y);
            else y1 -= lr * (float)((g > 0.f) - (g < 0.f));
            wn[e] = y1;
            if (SFREE) zn[e] = z1;
            shv[e] = f32_to_bf16(y1);
        }
        if (live && shadow != nullptr && !(SPHIP_FUSE_X & 2)) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            const u4 t = {(unsigned)shv[0] | ((unsigned)shv[1] << 16), (unsigned)shv[2] | ((unsigned)shv[3] << 16),
                          (unsigned)shv[4] | ((unsigned)shv[5] << 16), (unsigned)shv[6] | ((unsigned)shv[7] << 16)};
            __builtin_nontemporal_store(t, reinterpret_cast<u4*>(shadow + (int64_t)j * D + col));
        }
        const int srcA = (lane & ~7) + (sub >> 1), srcB = srcA + 4;
        const bool odd = (sub & 1) != 0;
        auto to_lines = [&](const float (&v)[8], f4& A, f4& B) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float alo = __shfl(v[q], srcA, 64), ahi = __shfl(v[4 + q], srcA, 64);
                const float blo = __shfl(v[q], srcB, 64), bhi = __shfl(v[4 + q], srcB, 64);
                A[q] = odd ? ahi : alo;
                B[q] = odd ? bhi : blo;
            }
        };
        const int colA = (int)slice * SLICE_COLS + sub * 4, colB = colA + 32;
        const bool rowlive = (j < M) && mine;                  // uniform over the lane group
        f4 A, B;
        to_lines(wn, A, B);
        if (!(SPHIP_FUSE_X & 1)) {
            // (other cache policies on these stores, one box: nt 2.31 ms, sc1 nt 2.37, sc1 2.40, sc0 sc1 2.41, sc0 2.42)
            if (rowlive && colA < D) __builtin_nontemporal_store(A, reinterpret_cast<f4*>(gW + (int64_t)j * D + colA));
            if (rowlive && colB < D) __builtin_nontemporal_store(B, reinterpret_cast<f4*>(gW + (int64_t)j * D + colB));
        }
        if constexpr (SFREE) {
            to_lines(zn, A, B);
            if (rowlive && colA < D) __builtin_nontemporal_store(A, reinterpret_cast<f4*>(sf.z + (int64_t)j * D + colA));
            if (rowlive && colB < D) __builtin_nontemporal_store(B, reinterpret_cast<f4*>(sf.z + (int64_t)j * D + colB));
        }
    } else if (live) {
        float* o = gW + (int64_t)j * D + col;
        if (FUSE) {
            unsigned short sh[C];
#pragma unroll
            for (int c = 0; c < C; c += 4) {
                f4 w = {wold[c], wold[c + 1], wold[c + 2], wold[c + 3]};
                if constexpr (SFREE) {
                    f4 z = {zold[c], zold[c + 1], zold[c + 2], zold[c + 3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float g = acc[c + e];
                        float y1 = w[e], z1 = z[e];
                        sf_update(y1, z1, (float)((g > 0.f) - (g < 0.f)), lr, sf.w_undo, sf.c, sf.w_y);
                        w[e] = y1, z[e] = z1;
                        sh[c + e] = f32_to_bf16(y1);
                    }
                    __builtin_nontemporal_store(z, reinterpret_cast<f4*>(sf.z + (int64_t)j * D + col + c));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float g = acc[c + e];
                        w[e] -= lr * (float)((g > 0.f) - (g < 0.f));
                        sh[c + e] = f32_to_bf16(w[e]);
                    }
                }
                __builtin_nontemporal_store(w, reinterpret_cast<f4*>(o + c));
            }
            if (shadow != nullptr) {
                bf16_t* so = shadow + (int64_t)j * D + col;
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int c = 0; c < C; c += 4) {
                    const u2 t = {(unsigned)sh[c] | ((unsigned)sh[c + 1] << 16), (unsigned)sh[c + 2] | ((unsigned)sh[c + 3] << 16)};
                    __builtin_nontemporal_store(t, reinterpret_cast<u2*>(so + c));
                }
            }
        } else {
            // accum (kernel argument: uniform): the rows of an earlier token block are already there
            // (sphip_segmented_weight_grad_acc) -- block sums added in block order
            f4 prev[C / 4];
            if (accum) {
#pragma unroll
                for (int c = 0; c < C; c += 4) prev[c / 4] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(o + c));
            }
#pragma unroll
            for (int c = 0; c < C; c += 4) {
                f4 t = {acc[c], acc[c + 1], acc[c + 2], acc[c + 3]};
                if (accum) t += prev[c / 4];
                __builtin_nontemporal_store(t, reinterpret_cast<f4*>(o + c));
            }
        }
    }
    if (gbias != nullptr && slice == 0 && sub == 0 && j < M && mine) gbias[j] = bsum;
    if constexpr (FUSE) {
        if (bpart != nullptr) {                                   // kernel argument: uniform
            float cb[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float t = inrange ? gbj * wold[c] : 0.f;
                t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x128, 0xF, 0xF, true));  // lane ^ 8
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                cb[c] = t;
            }
            const int wv = threadIdx.x >> 6;
            if (grp == 0) {
#pragma unroll
                for (int c = 0; c < C; ++c) sh_wp[wv][sub * C + c] = cb[c];
            }
            // no workgroup barrier (the waves end at different times by design): every wave leaves
            // its partial and counts itself; the one that counts last adds the four up
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            int prev = 0;
            if (lane == 0) prev = __hip_atomic_fetch_add(&sh_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            prev = __builtin_amdgcn_readfirstlane(prev);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (prev == 3 && lane < SLICE_COLS) {
                const float t = ((sh_wp[0][lane] + sh_wp[1][lane]) + sh_wp[2][lane]) + sh_wp[3][lane];
                const int oc = (int)slice * SLICE_COLS + lane;
                if (oc < D) bpart[(int64_t)lblock * D + oc] = t;
            }
        }
    }
}

// gbias[j] = sum of the coefficients of latent j's segment: the bias gradient, from this one kernel
// whatever form computes the weight gradient (the forms' own coefficient sums only serve their row
// shift). A segment of up to 64 entries is one sequential chain in segment order; a longer one is
// summed by a whole wave -- lane l adds entries l, l + 64, ... in order, the 64 lane sums meet in the
// xor butterfly -- so that a latent firing on thousands of tokens is not one thread's chain of
// thousands of dependent gathers (1.4 ms at 7000 entries). Fixed by the segment alone either way.
template <typename CT>
__global__ __launch_bounds__(256) void seg_coeff_sum_kernel(const uint32_t* __restrict__ perm,
                                                            const int* __restrict__ seg_begin,
                                                            const CT* __restrict__ coeff, int M,
                                                            float* __restrict__ out, int accum = 0) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int beg = 0, end = 0;
    if (j < M) { beg = seg_begin[j]; end = seg_begin[j + 1]; }
    const bool is_long = end - beg > 64;
    if (j < M && !is_long) {
        float t = 0.f;
        int p = beg;
        for (; p + 4 <= end; p += 4) {                     // four gathers in flight, added in order
            const uint32_t e0 = perm[p], e1 = perm[p + 1], e2 = perm[p + 2], e3 = perm[p + 3];
            const float c0 = Elem<CT>::load(coeff, e0), c1 = Elem<CT>::load(coeff, e1);
            const float c2 = Elem<CT>::load(coeff, e2), c3 = Elem<CT>::load(coeff, e3);
            t += c0, t += c1, t += c2, t += c3;
        }
        for (; p < end; ++p) t += Elem<CT>::load(coeff, perm[p]);
        out[j] = accum ? out[j] + t : t;
    }
    unsigned long long todo = __ballot(is_long);
    while (todo != 0) {                                     // wave-uniform
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int b = __shfl(beg, src, 64), e = __shfl(end, src, 64);
        float t = 0.f;
        for (int p = b + lane; p < e; p += 64) t += Elem<CT>::load(coeff, perm[p]);
        t = wave_sum(t);
        if (lane == 0) out[j + src] = accum ? out[j + src] + t : t;    // lane 0's latent is the wave's first
    }
}


// ---------------------------------------------------------------------------------------
// Wide-sliced variant WITH dcoeff (the W_dec gradient of the training shapes): one wave per
// (latent, 256-column slice), slices are the slow grid dimension, so the whole chip reduces
// the same N x 1 KiB column slice of `rows` (16 MiB at N = 16384) at a time and its k-fold
// re-reads are served by the Infinity Cache / L2 instead of HBM. A lane owns 4 columns.
// The dot product is split by slice: the 8 entries of a batch leave 8 partial products per
// lane, a 10-step transposing butterfly leaves the slice's dot of entry u in lanes 8u..8u+7,
// and lanes 0, 8, .., 56 store 8 consecutive floats of the slab[slice][position] array;
// seg_dcoeff_kernel then sums the slices in a fixed order. Entry metadata is wave-uniform
// (scalar loads). gW is summed in segment order: bitwise the same as the full-row kernel.
template <typename RT, bool DOT, bool FUSE>
__device__ __forceinline__ void wide_latent(const uint2* __restrict__ packed, const int* __restrict__ seg_begin,
                                            const RT* __restrict__ rows, const float* __restrict__ W,
                                            float* __restrict__ gW, float* __restrict__ gbias, float* __restrict__ slab,
                                            const float* __restrict__ row_shift, FastDiv kdiv, int64_t E, int D, int j,
                                            int slice, int lane, float lr, bf16_t* __restrict__ shadow, SfArgs sf);

template <typename RT, bool DOT, bool FUSE = false>
__global__ __launch_bounds__(256) void seg_wgrad_wide_kernel(const uint2* __restrict__ packed,
                                                             const int* __restrict__ seg_begin,
                                                             const RT* __restrict__ rows,
                                                             const float* __restrict__ W, float* __restrict__ gW,
                                                             float* __restrict__ gbias, float* __restrict__ slab,
                                                             const float* __restrict__ row_shift, FastDiv kdiv,
                                                             int64_t E, int M, int D, float lr = 0.f,
                                                             bf16_t* __restrict__ shadow = nullptr, int slice0 = 0,
                                                             SfArgs sf = SfArgs{nullptr, 0.f, 0.f, 0.f}) {
    const int lane = threadIdx.x & 63;
    const int wv = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slice = blockIdx.y + slice0;
    if (wv >= M) return;
    wide_latent<RT, DOT, FUSE>(packed, seg_begin, rows, W, gW, gbias, slab, row_shift, kdiv, E, D, wv, slice, lane, lr,
                               shadow, sf);
}

// The accumulation of entries [beg, end) of latent j over one 256-column slice: acc += c * row piece in
// entry order, bsum += c, and (DOT) the slice's partial dot of every entry into slab[slice][position].
template <typename RT, bool DOT>
__device__ __forceinline__ void wide_accumulate(const uint2* __restrict__ packed, const RT* __restrict__ rows,
                                                const float* __restrict__ W, float* __restrict__ slab, FastDiv kdiv,
                                                int64_t E, int D, int j, int beg, int end, int slice, int lane,
                                                float4& acc, float& bsum) {
    const int col = slice * 256 + lane * 4;
    // W and gW are streamed once: non-temporal, so that they do not displace the resident slice of `rows`
    typedef float f4 __attribute__((ext_vector_type(4)));
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    if (DOT) {
        const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(W + (int64_t)j * D + col));
        w = make_float4(t[0], t[1], t[2], t[3]);
    }
    const RT* rcol = rows + col;
    float* sl = slab + (int64_t)slice * E;

    constexpr int U = 8;
    // the segment's entries come in with one coalesced load per 64 (lane l holds entry p0 + l)
    // and reach the batches through v_readlane: a batch then waits for its row loads only, not
    // for a scalar fetch of its entries from the Infinity Cache first
    uint2 ent = make_uint2(0u, 0u);
    for (int p = beg; p < end; p += U) {
        const int q = (p - beg) & 63;                // wave-uniform; U divides 64
        if (q == 0) ent = (p + lane < end) ? packed[p + lane] : make_uint2(0u, 0u);
        float c[U];
        float4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p + u < end) {                       // wave-uniform
                const unsigned ex = (unsigned)__builtin_amdgcn_readlane((int)ent.x, q + u);
                c[u] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)ent.y, q + u));
                r[u] = load4(rcol + (int64_t)fastdiv(ex, kdiv) * D);
            } else {
                c[u] = 0.f;
                r[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        float d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p + u < end) {                       // keeps the sequential fma chain exact
                acc.x = fmaf(c[u], r[u].x, acc.x);
                acc.y = fmaf(c[u], r[u].y, acc.y);
                acc.z = fmaf(c[u], r[u].z, acc.z);
                acc.w = fmaf(c[u], r[u].w, acc.w);
                bsum += c[u];
            }
            if (DOT) d[u] = fmaf(r[u].w, w.w, fmaf(r[u].z, w.z, fmaf(r[u].y, w.y, r[u].x * w.x)));
        }
        if (!DOT) continue;
        // transposing butterfly: 8 values x 64 lanes -> lane l holds the wave sum of d[l >> 3]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool up = lane & 32;
            const float keep = up ? d[t + 4] : d[t], send = up ? d[t] : d[t + 4];
            d[t] = keep + __shfl_xor(send, 32, 64);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bool up = lane & 16;
            const float keep = up ? d[t + 2] : d[t], send = up ? d[t] : d[t + 2];
            d[t] = keep + __shfl_xor(send, 16, 64);
        }
        {
            const bool up = lane & 8;
            const float keep = up ? d[1] : d[0], send = up ? d[0] : d[1];
            d[0] = keep + __shfl_xor(send, 8, 64);
        }
        d[0] += __shfl_xor(d[0], 4, 64);
        d[0] += __shfl_xor(d[0], 2, 64);
        d[0] += __shfl_xor(d[0], 1, 64);
        const int u = lane >> 3;
        if ((lane & 7) == 0 && p + u < end) sl[p + u] = d[0];
    }
}

template <typename RT, bool DOT, bool FUSE>
__device__ __forceinline__ void wide_latent(const uint2* __restrict__ packed, const int* __restrict__ seg_begin,
                                            const RT* __restrict__ rows, const float* __restrict__ W,
                                            float* __restrict__ gW, float* __restrict__ gbias, float* __restrict__ slab,
                                            const float* __restrict__ row_shift, FastDiv kdiv, int64_t E, int D, int j,
                                            int slice, int lane, float lr, bf16_t* __restrict__ shadow, SfArgs sf) {
    const int col = slice * 256 + lane * 4;
    typedef float f4 __attribute__((ext_vector_type(4)));
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float bsum = 0.f;
    wide_accumulate<RT, DOT>(packed, rows, W, slab, kdiv, E, D, j, seg_begin[j], seg_begin[j + 1], slice, lane, acc, bsum);
    if (row_shift != nullptr) {
        const float4 sh = *reinterpret_cast<const float4*>(row_shift + col);
        acc.x = fmaf(-bsum, sh.x, acc.x);
        acc.y = fmaf(-bsum, sh.y, acc.y);
        acc.z = fmaf(-bsum, sh.z, acc.z);
        acc.w = fmaf(-bsum, sh.w, acc.w);
    }
    if (FUSE) {       // gW is the parameter: SignSGD step in place (see seg_wgrad_sliced_kernel)
        float4 p4 = *reinterpret_cast<const float4*>(gW + (int64_t)j * D + col);
        if (sf.z != nullptr) {
            float4 z4 = *reinterpret_cast<const float4*>(sf.z + (int64_t)j * D + col);
            sf_update(p4.x, z4.x, (float)((acc.x > 0.f) - (acc.x < 0.f)), lr, sf.w_undo, sf.c, sf.w_y);
            sf_update(p4.y, z4.y, (float)((acc.y > 0.f) - (acc.y < 0.f)), lr, sf.w_undo, sf.c, sf.w_y);
            sf_update(p4.z, z4.z, (float)((acc.z > 0.f) - (acc.z < 0.f)), lr, sf.w_undo, sf.c, sf.w_y);
            sf_update(p4.w, z4.w, (float)((acc.w > 0.f) - (acc.w < 0.f)), lr, sf.w_undo, sf.c, sf.w_y);
            *reinterpret_cast<float4*>(sf.z + (int64_t)j * D + col) = z4;
        } else {
            p4.x -= lr * (float)((acc.x > 0.f) - (acc.x < 0.f));
            p4.y -= lr * (float)((acc.y > 0.f) - (acc.y < 0.f));
            p4.z -= lr * (float)((acc.z > 0.f) - (acc.z < 0.f));
            p4.w -= lr * (float)((acc.w > 0.f) - (acc.w < 0.f));
        }
        *reinterpret_cast<float4*>(gW + (int64_t)j * D + col) = p4;
        if (shadow != nullptr) {
            const unsigned lo = (unsigned)f32_to_bf16(p4.x) | ((unsigned)f32_to_bf16(p4.y) << 16);
            const unsigned hi = (unsigned)f32_to_bf16(p4.z) | ((unsigned)f32_to_bf16(p4.w) << 16);
            *reinterpret_cast<uint2*>(shadow + (int64_t)j * D + col) = make_uint2(lo, hi);
        }
    } else {
        const f4 t = {acc.x, acc.y, acc.z, acc.w};
        __builtin_nontemporal_store(t, reinterpret_cast<f4*>(gW + (int64_t)j * D + col));
    }
    if (gbias != nullptr && slice == 0 && lane == 0) gbias[j] = bsum;
}

// ---------------------------------------------------------------------------------------
// Long segments (dense latents). The L2-sliced kernel gives a latent to one 8-lane group, so a
// workgroup lives as long as its longest segment; a wave-per-latent kernel is no better: one wave walks
// the segment 8 entries at a time, and with a latent that fires on 40 % of 16384 tokens that single
// chain of ~800 round trips (1.8 - 2.3 ms) was the whole launch. Latents with more than SKEW_T entries
// are therefore listed on the device (no host round trip; the list is usually empty), their segments
// cut into chunks of LONG_CH entries, every chunk reduced by a lane group of its own into a partial row
// (seg_long_chunk_kernel, entry order inside the chunk), and the partial rows of a latent
// added up in chunk order by seg_long_combine_kernel, which writes the gradient row or applies the fused
// optimizer step. The order of the sum is fixed by the segment alone (chunks of LONG_CH in entry order,
// then the chunks in order): the same bits on every run, but not the bits of one sequential chain over
// the whole segment, which is what the other kernel forms compute.
// hdr[0] = number of long latents, hdr[1] = number of chunks (both zeroed before the launch);
// list[i] = {latent, first chunk}, ctab[chunk] = {latent, chunk inside the latent}. Which slot a latent
// gets is arrival order: it decides where its partial rows lie, not what they hold.
#ifndef SPHIP_LONG_CH
#define SPHIP_LONG_CH 128
#endif
constexpr int LONG_CH = SPHIP_LONG_CH;
__global__ __launch_bounds__(256) void seg_long_list_kernel(const int* __restrict__ seg_begin, int M, int T, int cap,
                                                            int chunk_cap, int* __restrict__ hdr,
                                                            int2* __restrict__ list, int2* __restrict__ ctab) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= M) return;
    const int len = seg_begin[j + 1] - seg_begin[j];
    if (len <= T) return;
    const int nch = (len + LONG_CH - 1) / LONG_CH;
    const int pos = atomicAdd(hdr, 1);
    const int cb = atomicAdd(hdr + 1, nch);
    if (pos >= cap || cb + nch > chunk_cap) return;          // cannot happen: cap >= E / T, chunk_cap >= E / CH + E / T
    list[pos] = make_int2(j, cb);
    for (int c = 0; c < nch; ++c) ctab[cb + c] = make_int2(j, c);
}

// One workgroup reduces 32 chunks over one 128-byte column slice, eight lanes per chunk, from the
// slice-major copy of the rows like seg_wgrad_sliced_kernel (the N x 128 B slice stays in the XCD's L2);
// the chunks are equally long, so the lane groups finish together. Fixed grid: the workgroups of an XCD
// walk (slice, block of chunks) items slice after slice. DOT: the slice's partial dot of every entry
// goes to slab[slice][position], the layout of the L2-sliced kernel (whose parked ranges may have left
// garbage at these positions: this kernel runs after it on the same stream).
template <typename RT, bool DOT, bool PK4>
__global__ __launch_bounds__(256) void seg_long_chunk_kernel(const void* __restrict__ plist_,
                                                             const int* __restrict__ seg_begin,
                                                             const RT* __restrict__ rows_sm,
                                                             const float* __restrict__ Wdot, float* __restrict__ slab,
                                                             FastDiv kdiv, int64_t N, int64_t E, int D, int num_slices,
                                                             const int* __restrict__ hdr, const int2* __restrict__ ctab,
                                                             int chunk_cap, float* __restrict__ prow,
                                                             float* __restrict__ pbsum) {
    constexpr int C = Piece<RT>::COLS;
    constexpr int SLICE_COLS = 8 * C;
    constexpr int U = 8;
    typedef typename std::conditional<PK4, unsigned, uint2>::type Entry;
    typedef typename Piece<RT>::Raw Raw;
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ Entry sh_pk[32 * LONG_CH];
    const Entry* plist = reinterpret_cast<const Entry*>(plist_);
    const int n_chunks = min(hdr[1], chunk_cap);
    const int n_blocks = (n_chunks + 31) >> 5;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, G = gridDim.x >> 3;
    const int rounds = (num_slices + 7) >> 3;
    const int g = threadIdx.x >> 3, sub = threadIdx.x & 7;
    Entry* my = sh_pk + g * LONG_CH;
    for (int item = local; item < rounds * n_blocks; item += G) {      // block-uniform
        const int slice = xcd + 8 * (item / n_blocks);
        if (slice >= num_slices) break;
        const int w = (item % n_blocks) * 32 + g;
        int j = 0, beg = 0, len = 0;
        if (w < n_chunks) {
            const int2 t = ctab[w];
            j = t.x;
            beg = seg_begin[j] + t.y * LONG_CH;
            len = min(seg_begin[j + 1] - beg, LONG_CH);
        }
        const int col = slice * SLICE_COLS + sub * C;
        const bool live = w < n_chunks && col < D;
        __syncthreads();                                                 // the previous item's slots are done with
        for (int i = sub; i < len; i += 8) my[i] = plist[beg + i];
        float wdot[DOT ? C : 1];
        if (DOT) {
#pragma unroll
            for (int c = 0; c < C; ++c) wdot[c] = 0.f;
            if (live) {
#pragma unroll
                for (int c = 0; c < C; c += 4) {
                    const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(Wdot + (int64_t)j * D + col + c));
                    wdot[c] = t[0], wdot[c + 1] = t[1], wdot[c + 2] = t[2], wdot[c + 3] = t[3];
                }
            }
        }
        __syncthreads();
        float acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.f;
        float bsum = 0.f;
        const RT* base = rows_sm + ((int64_t)slice * N * 8 + sub) * C;
        const int n = live ? len : 0;
        for (int p = 0; p < n; p += U) {
            Entry pk[U];
            Raw v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) pk[u] = my[min(p + u, n - 1)];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                unsigned tok;
                if constexpr (PK4) tok = pk[u] >> 16;
                else tok = fastdiv(pk[u].x, kdiv);
                v[u] = Piece<RT>::load(base + (int64_t)tok * (8 * C));
            }
            if (DOT) {
                float mine_d = 0.f;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float d = 0.f;
#pragma unroll
                    for (int c = 0; c < C; ++c) d = fmaf(Piece<RT>::get(v[u], c), wdot[c], d);
                    d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0xB1, 0xF, 0xF, true));
                    d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x4E, 0xF, 0xF, true));
                    d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x141, 0xF, 0xF, true));
                    mine_d = (sub == u) ? d : mine_d;
                }
                // parked in the slot of the entry itself (it is in registers by now)
                if (p + sub < n) {
                    if constexpr (PK4) my[p + sub] = __float_as_uint(mine_d);
                    else my[p + sub].x = __float_as_uint(mine_d);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (p + u < n) {                        // group-uniform; keeps the sequential fma chain exact
                    float cf;
                    if constexpr (PK4) cf = __uint_as_float(pk[u] << 16);
                    else cf = __uint_as_float(pk[u].y);
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[c] = fmaf(cf, Piece<RT>::get(v[u], c), acc[c]);
                    bsum += cf;
                }
            }
        }
        if (DOT) {
            __syncthreads();
            float* sl = slab + (int64_t)slice * E + beg;
            for (int i = sub; i < n; i += 8) {
                float d;
                if constexpr (PK4) d = __uint_as_float(my[i]);
                else d = __uint_as_float(my[i].x);
                sl[i] = d;
            }
        }
        if (live) {
            float* o = prow + (int64_t)w * D + col;
#pragma unroll
            for (int c = 0; c < C; c += 4) {
                const f4 t = {acc[c], acc[c + 1], acc[c + 2], acc[c + 3]};
                __builtin_nontemporal_store(t, reinterpret_cast<f4*>(o + c));
            }
            if (slice == 0 && sub == 0) pbsum[w] = bsum;
        }
    }
}

// gW[j, :] (or, FUSE, the optimizer step of parameter row j) from the partial rows of latent j's chunks
template <bool FUSE>
__global__ __launch_bounds__(256) void seg_long_combine_kernel(const int* __restrict__ hdr, const int2* __restrict__ list,
                                                               int cap, const int* __restrict__ seg_begin,
                                                               const float* __restrict__ prow,
                                                               const float* __restrict__ pbsum,
                                                               const float* __restrict__ row_shift,
                                                               float* __restrict__ gW, int D, float lr,
                                                               bf16_t* __restrict__ shadow, SfArgs sf, int accum = 0) {
    const int n_long = min(hdr[0], cap);
    for (int i = blockIdx.x; i < n_long; i += gridDim.x) {
        const int2 t = list[i];
        const int j = t.x, cb = t.y;
        const int nch = (seg_begin[j + 1] - seg_begin[j] + LONG_CH - 1) / LONG_CH;
        float bsum = pbsum[cb];
        for (int c = 1; c < nch; ++c) bsum += pbsum[cb + c];
        for (int col = threadIdx.x * 4; col < D; col += 1024) {
            float4 acc = *reinterpret_cast<const float4*>(prow + (int64_t)cb * D + col);
            for (int c = 1; c < nch; ++c) {
                const float4 v = *reinterpret_cast<const float4*>(prow + (int64_t)(cb + c) * D + col);
                acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
            }
            if (row_shift != nullptr) {
                const float4 sh = *reinterpret_cast<const float4*>(row_shift + col);
                acc.x = fmaf(-bsum, sh.x, acc.x);
                acc.y = fmaf(-bsum, sh.y, acc.y);
                acc.z = fmaf(-bsum, sh.z, acc.z);
                acc.w = fmaf(-bsum, sh.w, acc.w);
            }
            if (FUSE) {       // gW is the parameter: SignSGD / schedule-free step in place (see seg_wgrad_sliced_kernel)
                float4 p4 = *reinterpret_cast<const float4*>(gW + (int64_t)j * D + col);
                if (sf.z != nullptr) {
                    float4 z4 = *reinterpret_cast<const float4*>(sf.z + (int64_t)j * D + col);
                    sf_update(p4.x, z4.x, (float)((acc.x > 0.f) - (acc.x < 0.f)), lr, sf.w_undo, sf.c, sf.w_y);
                    sf_update(p4.y, z4.y, (float)((acc.y > 0.f) - (acc.y < 0.f)), lr, sf.w_undo, sf.c, sf.w_y);
                    sf_update(p4.z, z4.z, (float)((acc.z > 0.f) - (acc.z < 0.f)), lr, sf.w_undo, sf.c, sf.w_y);
                    sf_update(p4.w, z4.w, (float)((acc.w > 0.f) - (acc.w < 0.f)), lr, sf.w_undo, sf.c, sf.w_y);
                    *reinterpret_cast<float4*>(sf.z + (int64_t)j * D + col) = z4;
                } else {
                    p4.x -= lr * (float)((acc.x > 0.f) - (acc.x < 0.f));
                    p4.y -= lr * (float)((acc.y > 0.f) - (acc.y < 0.f));
                    p4.z -= lr * (float)((acc.z > 0.f) - (acc.z < 0.f));
                    p4.w -= lr * (float)((acc.w > 0.f) - (acc.w < 0.f));
                }
                *reinterpret_cast<float4*>(gW + (int64_t)j * D + col) = p4;
                if (shadow != nullptr) {
                    const unsigned lo = (unsigned)f32_to_bf16(p4.x) | ((unsigned)f32_to_bf16(p4.y) << 16);
                    const unsigned hi = (unsigned)f32_to_bf16(p4.z) | ((unsigned)f32_to_bf16(p4.w) << 16);
                    *reinterpret_cast<uint2*>(shadow + (int64_t)j * D + col) = make_uint2(lo, hi);
                }
            } else {
                if (accum) {                                 // an earlier token block's row is already there
                    const float4 o4 = *reinterpret_cast<const float4*>(gW + (int64_t)j * D + col);
                    acc.x += o4.x, acc.y += o4.y, acc.z += o4.z, acc.w += o4.w;
                }
                *reinterpret_cast<float4*>(gW + (int64_t)j * D + col) = acc;
            }
        }
    }
}

// dcoeff[entry at position p] = sum over slices of slab[slice][p], slices in ascending order
__global__ __launch_bounds__(256) void seg_dcoeff_bounded_kernel(const uint2* __restrict__ packed,
                                                                 const float* __restrict__ slab, int64_t E,
                                                                 const int* __restrict__ seg_begin, int M,
                                                                 int num_slices, float* __restrict__ dcoeff) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= seg_begin[M]) return;
    float t = 0.f;
    int sidx = 0;
    for (; sidx + 16 <= num_slices; sidx += 16) {       // 16 loads in flight, summed in slice order
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = __builtin_nontemporal_load(slab + (int64_t)(sidx + u) * E + p);
#pragma unroll
        for (int u = 0; u < 16; ++u) t += v[u];
    }
    for (; sidx < num_slices; ++sidx) t += slab[(int64_t)sidx * E + p];
    dcoeff[packed[p].x] = t;
}

// segments longer than this go from the L2-sliced kernel to the chunked long-segment kernels, whose
// workgroups (a fixed grid of LONG_GRID) walk the chunks of such latents
#ifndef SPHIP_SKEW_T
#define SPHIP_SKEW_T 128
#endif
constexpr int SKEW_T = SPHIP_SKEW_T;
// without dots the wide (wave-per-latent) kernel takes over from the L2-sliced one when the mean segment
// is this long (AuxK as row gathers: ~2500 entries on each dead latent); the multi-top-k decode
// (mean 4 N k / M = 96 at the headline shape) stays with the L2-sliced kernel: 20.7 -> see DESIGN
#ifndef SPHIP_WIDE_MEAN
#define SPHIP_WIDE_MEAN 256
#endif
#ifndef SPHIP_LONG_GRID
#define SPHIP_LONG_GRID 1024
#endif
constexpr unsigned LONG_GRID = SPHIP_LONG_GRID;      // a multiple of 8: the workgroups of an XCD share slices

template <typename RT, bool FUSE = false, bool DOT = false>
static bool launch_sliced(const uint2* packed, const int* seg, const void* rows, void* rows_sm, float* gW,
                          float* gbias, const float* row_shift, int64_t N, int k, int M, int D, hipStream_t s,
                          float lr = 0.f, bf16_t* shadow = nullptr, const int* skew = nullptr,
                          const float* Wdot = nullptr, float* slab = nullptr, SfArgs sf = SfArgs{nullptr, 0.f, 0.f, 0.f},
                          const float* gb_in = nullptr, float* bpart = nullptr, const unsigned* packed4 = nullptr,
                          int accum = 0) {
    constexpr int C = Piece<RT>::COLS;
    if (D % C != 0) return false;
    const int num_slices = (int)cdiv(D, 8 * C);
    {
        int64_t blocks = cdiv(N * num_slices * 8, 256);
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL((slice_major_kernel<RT>), dim3((unsigned)blocks), dim3(256), 0, s, (const RT*)rows,
                           (uint4*)rows_sm, N, D, num_slices);
    }
    const int bps = (int)cdiv(M, 32);
    const int64_t rounds = cdiv(num_slices, 8);
    const int64_t nblocks = rounds * bps * 8;
    if (nblocks >= (1ll << 31)) return false;
    if (packed4 != nullptr && FUSE && sf.z != nullptr)
        hipLaunchKernelGGL((seg_wgrad_sliced_kernel<RT, FUSE, DOT, FUSE, true>), dim3((unsigned)nblocks), dim3(256), 0, s,
                           reinterpret_cast<const uint2*>(packed4), seg,
                           (const RT*)rows_sm, gW, gbias, row_shift, make_fastdiv((unsigned)k), N, M, D, bps, num_slices, lr,
                           shadow, skew, SKEW_T, Wdot, slab, (int64_t)N * k, sf, gb_in, bpart, accum);
    else if (packed4 != nullptr)
        hipLaunchKernelGGL((seg_wgrad_sliced_kernel<RT, FUSE, DOT, false, true>), dim3((unsigned)nblocks), dim3(256), 0, s,
                           reinterpret_cast<const uint2*>(packed4), seg,
                           (const RT*)rows_sm, gW, gbias, row_shift, make_fastdiv((unsigned)k), N, M, D, bps, num_slices, lr,
                           shadow, skew, SKEW_T, Wdot, slab, (int64_t)N * k, sf, gb_in, bpart, accum);
    else if (FUSE && sf.z != nullptr)
        hipLaunchKernelGGL((seg_wgrad_sliced_kernel<RT, FUSE, DOT, FUSE>), dim3((unsigned)nblocks), dim3(256), 0, s, packed, seg,
                           (const RT*)rows_sm, gW, gbias, row_shift, make_fastdiv((unsigned)k), N, M, D, bps, num_slices, lr,
                           shadow, skew, SKEW_T, Wdot, slab, (int64_t)N * k, sf, gb_in, bpart, accum);
    else
        hipLaunchKernelGGL((seg_wgrad_sliced_kernel<RT, FUSE, DOT, false>), dim3((unsigned)nblocks), dim3(256), 0, s, packed, seg,
                           (const RT*)rows_sm, gW, gbias, row_shift, make_fastdiv((unsigned)k), N, M, D, bps, num_slices, lr,
                           shadow, skew, SKEW_T, Wdot, slab, (int64_t)N * k, sf, gb_in, bpart, accum);
    return true;
}

template <typename RT, bool DOT>
static void launch_nv(const uint2* packed, const int* seg, const void* rows, const float* W, float* gW, float* gbias,
                      float* dcoeff, const float* row_shift, int k, int M, int D, hipStream_t s) {
    dim3 grid((unsigned)cdiv(M, 4)), block(256);
    const FastDiv kd = make_fastdiv((unsigned)k);
    if (D <= 256)
        hipLaunchKernelGGL((seg_wgrad_kernel<RT, 1, DOT>), grid, block, 0, s, packed, seg, (const RT*)rows, W, gW, gbias, dcoeff, row_shift, kd, M, D);
    else if (D <= 1024)
        hipLaunchKernelGGL((seg_wgrad_kernel<RT, 4, DOT>), grid, block, 0, s, packed, seg, (const RT*)rows, W, gW, gbias, dcoeff, row_shift, kd, M, D);
    else if (D <= 2048)
        hipLaunchKernelGGL((seg_wgrad_kernel<RT, 8, DOT>), grid, block, 0, s, packed, seg, (const RT*)rows, W, gW, gbias, dcoeff, row_shift, kd, M, D);
    else
        hipLaunchKernelGGL((seg_wgrad_kernel<RT, 16, DOT>), grid, block, 0, s, packed, seg, (const RT*)rows, W, gW, gbias, dcoeff, row_shift, kd, M, D);
}

}  // namespace sphip

static size_t packed8_bytes(int64_t N, int64_t k) { return (((size_t)N * (size_t)k * sizeof(uint2)) + 255) & ~(size_t)255; }
// the 8-byte entries, then the same list as 4-byte entries (seg_pack_kernel: bf16 coefficients only)
static size_t packed_bytes(int64_t N, int64_t k) { return packed8_bytes(N, k) + ((((size_t)N * (size_t)k * 4) + 255) & ~(size_t)255); }
static size_t rows_sm_bytes(int64_t N, int64_t D) { return (size_t)N * (size_t)(((D * 4 + 127) / 128) * 128); }
// the wide-sliced dcoeff path: D a multiple of 256 with at least two slices
static bool wide_eligible(int64_t D) { return D % 256 == 0 && D >= 512 && D / 256 <= 65535; }
static size_t slab_bytes(int64_t N, int64_t k, int64_t D) {
    return wide_eligible(D) ? (size_t)(D / 256) * (size_t)N * (size_t)k * sizeof(float) : 0;
}
// the L2-sliced dcoeff path (fp32 rows): one partial per entry and 128-byte slice, next to the
// slice-major copy of the rows
static bool l2dot_eligible(int64_t N, int64_t k, int64_t D) {
    return D % 32 == 0 && D >= 512 && (size_t)(D / 32) * (size_t)N * (size_t)k * sizeof(float) <= ((size_t)8 << 30);
}
static size_t l2dot_bytes(int64_t N, int64_t k, int64_t D) {
    // slice-major rows + 128-byte-slice partials
    return l2dot_eligible(N, k, D) ? ((rows_sm_bytes(N, D) + 255) & ~(size_t)255) +
                                         (size_t)(D / 32) * (size_t)N * (size_t)k * sizeof(float)
                                   : 0;
}

extern "C" size_t sphip_weight_grad_workspace_bytes(int64_t N, int64_t k, int64_t D);
// The long-segment path's tail of the workspace (seg_long_* kernels): header, list of long latents
// (at most E / SKEW_T), chunk table and per-chunk coefficient sums (at most E / LONG_CH + E / SKEW_T
// chunks), then one partial row of D floats per chunk.
struct LongTail {
    int* hdr;
    int2* list;
    int2* ctab;
    float* pbsum;
    float* prow;
    int cap, chunk_cap;
};
static size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }
static int long_cap(int64_t N, int64_t k) { return (int)(N * k / sphip::SKEW_T + 2); }
static int long_chunk_cap(int64_t N, int64_t k) { return (int)(N * k / sphip::LONG_CH + N * k / sphip::SKEW_T + 2); }
static size_t long_tail_bytes(int64_t N, int64_t k, int64_t D) {
    if (!wide_eligible(D)) return 0;
    const size_t cap = (size_t)long_cap(N, k), cc = (size_t)long_chunk_cap(N, k);
    return 256 + up256(cap * sizeof(int2)) + up256(cc * sizeof(int2)) + up256(cc * sizeof(float)) +
           up256(cc * (size_t)D * sizeof(float));
}
// the tail sits at the end of the workspace
static LongTail long_tail(void* workspace, int64_t N, int64_t k, int64_t D) {
    char* p = (char*)workspace + sphip_weight_grad_workspace_bytes(N, k, D) - long_tail_bytes(N, k, D);
    LongTail t;
    t.cap = long_cap(N, k), t.chunk_cap = long_chunk_cap(N, k);
    t.hdr = (int*)p, p += 256;
    t.list = (int2*)p, p += up256((size_t)t.cap * sizeof(int2));
    t.ctab = (int2*)p, p += up256((size_t)t.chunk_cap * sizeof(int2));
    t.pbsum = (float*)p, p += up256((size_t)t.chunk_cap * sizeof(float));
    t.prow = (float*)p;
    return t;
}
// lists the long latents: returns false if the launch could not be enqueued
static bool list_long_latents(const LongTail& t, const int* seg_begin, int64_t M, hipStream_t s) {
    if (hipMemsetAsync(t.hdr, 0, 2 * sizeof(int), s) != hipSuccess) return false;
    hipLaunchKernelGGL(sphip::seg_long_list_kernel, dim3((unsigned)sphip::cdiv(M, 256)), dim3(256), 0, s, seg_begin, (int)M,
                       sphip::SKEW_T, t.cap, t.chunk_cap, t.hdr, t.list, t.ctab);
    return true;
}
// the chunk partials of the listed latents (rows_sm: the slice-major copy launch_sliced has made), then
// their combination into gW rows (FUSE: the optimizer step)
template <typename RT, bool DOT, bool FUSE>
static void reduce_long_latents(const LongTail& t, const uint2* packed, const unsigned* packed4, const int* seg_begin,
                                const void* rows_sm, const float* W, float* gW, float* slab, const float* row_shift,
                                int64_t N, int64_t k, int64_t D, hipStream_t s, float lr = 0.f,
                                sphip::bf16_t* shadow = nullptr, sphip::SfArgs sf = sphip::SfArgs{nullptr, 0.f, 0.f, 0.f},
                                int accum = 0) {
    using namespace sphip;
    const int num_slices = (int)cdiv(D, 8 * Piece<RT>::COLS);
    const FastDiv kd = make_fastdiv((unsigned)k);
    dim3 grid(LONG_GRID), block(256);
    if (packed4 != nullptr)
        hipLaunchKernelGGL((seg_long_chunk_kernel<RT, DOT, true>), grid, block, 0, s, (const void*)packed4, seg_begin,
                           (const RT*)rows_sm, W, slab, kd, N, N * k, (int)D, num_slices, t.hdr, t.ctab, t.chunk_cap, t.prow, t.pbsum);
    else
        hipLaunchKernelGGL((seg_long_chunk_kernel<RT, DOT, false>), grid, block, 0, s, (const void*)packed, seg_begin,
                           (const RT*)rows_sm, W, slab, kd, N, N * k, (int)D, num_slices, t.hdr, t.ctab, t.chunk_cap, t.prow, t.pbsum);
    hipLaunchKernelGGL((seg_long_combine_kernel<FUSE>), dim3(256), block, 0, s, t.hdr, t.list, t.cap, seg_begin, t.prow,
                       t.pbsum, row_shift, gW, (int)D, lr, shadow, sf, accum);
}

extern "C" size_t sphip_weight_grad_workspace_bytes(int64_t N, int64_t k, int64_t D) {
    if (N <= 0 || k <= 0 || D <= 0) return 0;
    // packed entry list + max(slice-major copy of `rows` (no dcoeff), per-slice dot slab (dcoeff))
    size_t a = rows_sm_bytes(N, D);
    const size_t b = slab_bytes(N, k, D), c = l2dot_bytes(N, k, D);
    a = a > b ? a : b;
    a = a > c ? a : c;
    // + the long-segment path's lists and partial rows
    return packed_bytes(N, k) + ((a + 255) & ~(size_t)255) + long_tail_bytes(N, k, D);
}

static int launch_coeff_sums(const uint32_t* perm, const int32_t* seg_begin, const void* coeff, int coeff_dtype, int64_t M,
                             float* out, hipStream_t s, int accum = 0) {
    using namespace sphip;
    dim3 grid((unsigned)cdiv(M, 256)), block(256);
    if (coeff_dtype == SPHIP_F32)
        hipLaunchKernelGGL((seg_coeff_sum_kernel<float>), grid, block, 0, s, perm, seg_begin, (const float*)coeff, (int)M, out, accum);
    else
        hipLaunchKernelGGL((seg_coeff_sum_kernel<bf16_t>), grid, block, 0, s, perm, seg_begin, (const bf16_t*)coeff, (int)M, out, accum);
    SPHIP_CHECK_LAUNCH("segment_coeff_sums");
    return SPHIP_OK;
}

static int weight_grad_impl(const uint32_t* perm, const int32_t* seg_begin, const void* coeff,
                            int coeff_dtype, const void* rows, int rows_dtype, const float* W,
                            float* gW, float* gbias, float* dcoeff, const float* row_shift, int64_t N,
                            int64_t k, int64_t M, int64_t D, void* workspace, size_t workspace_bytes,
                            void* stream, int accum = 0);

// Whether the kernel form this shape takes can add into rows that are already there (the L2-sliced forms);
// mirrors the dispatch of weight_grad_impl, which checks it before it launches anything.
extern "C" int sphip_weight_grad_accumulates(int64_t N, int64_t k, int64_t M, int64_t D, int rows_dtype, int want_dcoeff) {
    using namespace sphip;
    if (N <= 0 || k <= 0 || M <= 0 || D <= 0 || !valid_dtype(rows_dtype) || N * k >= (1ll << 31) || D % 4 != 0) return 0;
    const char* force = getenv("SPHIP_WGRAD");
    const bool big = N * D * (int64_t)dtype_size(rows_dtype) >= (8ll << 20);
    const int64_t C = rows_dtype == SPHIP_F32 ? Piece<float>::COLS : Piece<bf16_t>::COLS;
    const bool sliced_ok = D % C == 0 && cdiv(cdiv(D, 8 * C), (int64_t)8) * cdiv(M, (int64_t)32) * 8 < (1ll << 31);
    if (want_dcoeff)
        return rows_dtype == SPHIP_F32 && l2dot_eligible(N, k, D) && D / 32 <= 65535 && sliced_ok &&
               ((force && force[0] == 'l') || (!force && big && N * k >= 4 * M && N * 128 <= (3ll << 20)));
    const bool want_wide = (force && force[0] == 'w') || (big && N * k >= SPHIP_WIDE_MEAN * M);
    return !(force && (force[0] == 'f' || force[0] == 'w')) && !(wide_eligible(D) && !(force && force[0] == 'f') && want_wide) &&
           sliced_ok;
}

extern "C" int sphip_segmented_weight_grad(const uint32_t* perm, const int32_t* seg_begin, const void* coeff,
                                           int coeff_dtype, const void* rows, int rows_dtype, const float* W,
                                           float* gW, float* gbias, float* dcoeff, const float* row_shift, int64_t N,
                                           int64_t k, int64_t M, int64_t D, void* workspace, size_t workspace_bytes,
                                           void* stream) {
    // the bias gradient comes from seg_coeff_sum_kernel whatever form computes the weight gradient
    const int rc = weight_grad_impl(perm, seg_begin, coeff, coeff_dtype, rows, rows_dtype, W, gW, nullptr, dcoeff, row_shift,
                                    N, k, M, D, workspace, workspace_bytes, stream);
    if (rc != SPHIP_OK || gbias == nullptr) return rc;
    return launch_coeff_sums(perm, seg_begin, coeff, coeff_dtype, M, gbias, (hipStream_t)stream);
}

extern "C" int sphip_segmented_weight_grad_acc(const uint32_t* perm, const int32_t* seg_begin, const void* coeff,
                                               int coeff_dtype, const void* rows, int rows_dtype, const float* W,
                                               float* gW, float* gbias, float* dcoeff, const float* row_shift,
                                               int64_t N, int64_t k, int64_t M, int64_t D, void* workspace,
                                               size_t workspace_bytes, int accumulate, void* stream) {
    const int acc = accumulate != 0;
    const int rc = weight_grad_impl(perm, seg_begin, coeff, coeff_dtype, rows, rows_dtype, W, gW, nullptr, dcoeff, row_shift,
                                    N, k, M, D, workspace, workspace_bytes, stream, acc);
    if (rc != SPHIP_OK || gbias == nullptr) return rc;
    return launch_coeff_sums(perm, seg_begin, coeff, coeff_dtype, M, gbias, (hipStream_t)stream, acc);
}

static int weight_grad_impl(const uint32_t* perm, const int32_t* seg_begin, const void* coeff,
                            int coeff_dtype, const void* rows, int rows_dtype, const float* W,
                            float* gW, float* gbias, float* dcoeff, const float* row_shift, int64_t N,
                            int64_t k, int64_t M, int64_t D, void* workspace, size_t workspace_bytes,
                            void* stream, int accum) {
    using namespace sphip;
    SPHIP_REQUIRE(valid_dtype(coeff_dtype) && valid_dtype(rows_dtype), SPHIP_EINVAL, "segmented_weight_grad: bad dtype");
    SPHIP_REQUIRE(perm && seg_begin && coeff && rows && gW, SPHIP_EINVAL, "segmented_weight_grad: null pointer");
    SPHIP_REQUIRE(dcoeff == nullptr || W != nullptr, SPHIP_EINVAL, "segmented_weight_grad: dcoeff needs W");
    SPHIP_REQUIRE(N > 0 && k > 0 && M > 0 && D > 0, SPHIP_EINVAL, "segmented_weight_grad: empty shape");
    SPHIP_REQUIRE(N * k < (1ll << 31) && M < (1ll << 31) - 4 && D < (1ll << 30), SPHIP_EUNSUPPORTED,
                  "segmented_weight_grad: shape too large");
    SPHIP_REQUIRE(D % 4 == 0, SPHIP_EUNSUPPORTED, "segmented_weight_grad: D=%lld must be a multiple of 4", (long long)D);
    SPHIP_REQUIRE(((uintptr_t)rows & 15) == 0 && ((uintptr_t)gW & 15) == 0 && ((uintptr_t)W & 15) == 0 &&
                      ((uintptr_t)row_shift & 15) == 0,
                  SPHIP_EINVAL, "segmented_weight_grad: rows/W/gW/row_shift must be 16-byte aligned");
    const size_t need = sphip_weight_grad_workspace_bytes(N, k, D);
    SPHIP_REQUIRE(workspace != nullptr && workspace_bytes >= need, SPHIP_EWORKSPACE,
                  "segmented_weight_grad: workspace %zu < %zu bytes", workspace_bytes, need);
    SPHIP_REQUIRE(((uintptr_t)workspace & 255) == 0, SPHIP_EINVAL, "segmented_weight_grad: workspace must be 256-byte aligned");
    SPHIP_REQUIRE(!accum || sphip_weight_grad_accumulates(N, k, M, D, rows_dtype, dcoeff != nullptr), SPHIP_EUNSUPPORTED,
                  "segmented_weight_grad_acc: this shape does not take an L2-sliced kernel (N=%lld k=%lld M=%lld D=%lld)",
                  (long long)N, (long long)k, (long long)M, (long long)D);
    hipStream_t s = (hipStream_t)stream;
    const bool dot = dcoeff != nullptr;
    const int64_t E = N * k;
    uint2* packed = (uint2*)workspace;
    void* rows_sm = (char*)workspace + packed_bytes(N, k);
    // 4-byte entries for the L2-sliced kernels (bf16 coefficients: the activations of a bf16-autocast step)
    unsigned* packed4 = (coeff_dtype == SPHIP_BF16 && N <= 65536 && getenv("SPHIP_WGRAD_PK8") == nullptr)
                            ? (unsigned*)((char*)workspace + packed8_bytes(N, k)) : nullptr;
    if (coeff_dtype == SPHIP_F32)
        hipLaunchKernelGGL((seg_pack_kernel<float>), dim3((unsigned)cdiv(E, 256)), dim3(256), 0, s, perm, (const float*)coeff, E, seg_begin, (int)M, packed, dcoeff);
    else
        hipLaunchKernelGGL((seg_pack_kernel<bf16_t>), dim3((unsigned)cdiv(E, 256)), dim3(256), 0, s, perm, (const bf16_t*)coeff, E, seg_begin, (int)M, packed, dcoeff, packed4, make_fastdiv((unsigned)k));
    const char* force = getenv("SPHIP_WGRAD");     // "full" | "sliced" | "wide" (tests)
    // wide-sliced kernel: always for dcoeff at training sizes; without dcoeff only for long
    // segments (AuxK: N*k_aux entries over the few dead latents), where the 8-latents-per-wave
    // L2-sliced kernel would run one 8-lane group per wave
    const bool big = N * D * (int64_t)dtype_size(rows_dtype) >= (8ll << 20);
    const bool want_wide = dot ? ((force && force[0] == 's') || (big && N * k >= 4 * M))
                               : ((force && force[0] == 'w') || (big && N * k >= SPHIP_WIDE_MEAN * M));
    // dcoeff at training sizes with fp32 rows: the L2-sliced kernel with per-slice dot partials
    // (the N x 128 B slice of the rows stays in L2: twice the Infinity Cache's rate);
    // SPHIP_WGRAD=wide keeps the wide kernel (A/B), =l2 forces this one
    if (dot && rows_dtype == SPHIP_F32 && l2dot_eligible(N, k, D) && D / 32 <= 65535 &&
        ((force && force[0] == 'l') || (!force && big && N * k >= 4 * M && N * 128 <= (3ll << 20)))) {
        float* slab = (float*)((char*)rows_sm + ((rows_sm_bytes(N, D) + 255) & ~(size_t)255));
        // latents with more than SKEW_T entries (dense latents: a workgroup of the sliced kernel
        // lives as long as its longest segment) are listed on the device and left to the chunked
        // long-segment kernels, launched next to the sliced one with small fixed grids
        const bool use_long = wide_eligible(D);
        const LongTail lt = use_long ? long_tail(workspace, N, k, D) : LongTail{};
        const int* skew = use_long ? lt.hdr : nullptr;
        if (use_long)
            SPHIP_REQUIRE(list_long_latents(lt, seg_begin, M, s), SPHIP_ELAUNCH, "segmented_weight_grad: memset failed");
        if (launch_sliced<float, false, true>(packed, seg_begin, rows, rows_sm, gW, gbias, row_shift, N, (int)k, (int)M,
                                              (int)D, s, 0.f, nullptr, skew, W, slab, SfArgs{nullptr, 0.f, 0.f, 0.f}, nullptr,
                                              nullptr, packed4, accum)) {
            SPHIP_CHECK_LAUNCH("segmented_weight_grad(l2 dot)");
            if (use_long)
                reduce_long_latents<float, true, false>(lt, packed, packed4, seg_begin, rows_sm, W, gW, slab, row_shift, N, k, D, s,
                                                        0.f, nullptr, SfArgs{nullptr, 0.f, 0.f, 0.f}, accum);
            hipLaunchKernelGGL(seg_dcoeff_bounded_kernel, dim3((unsigned)cdiv(E, 256)), dim3(256), 0, s, packed, slab, E, seg_begin, (int)M, (int)(D / 32), dcoeff);
            SPHIP_CHECK_LAUNCH("segmented_weight_grad(dcoeff)");
            return SPHIP_OK;
        }
    }
    // accumulation into rows that are already there is built into the L2-sliced forms only (the caller adds otherwise)
    const bool sliced_nodot = !dot && !(force && (force[0] == 'f' || force[0] == 'w')) &&
                              !(wide_eligible(D) && !(force && force[0] == 'f') && want_wide);
    SPHIP_REQUIRE(!accum || sliced_nodot, SPHIP_EUNSUPPORTED,
                  "segmented_weight_grad_acc: this shape does not take an L2-sliced kernel (N=%lld k=%lld M=%lld D=%lld)",
                  (long long)N, (long long)k, (long long)M, (long long)D);
    if (wide_eligible(D) && !(force && force[0] == 'f') && want_wide) {
        float* slab = (float*)rows_sm;
        const FastDiv kd = make_fastdiv((unsigned)k);
        dim3 grid((unsigned)cdiv(M, 4), (unsigned)(D / 256)), block(256);
        if (rows_dtype == SPHIP_F32) {
            if (dot) hipLaunchKernelGGL((seg_wgrad_wide_kernel<float, true>), grid, block, 0, s, packed, seg_begin, (const float*)rows, W, gW, gbias, slab, row_shift, kd, E, (int)M, (int)D);
            else hipLaunchKernelGGL((seg_wgrad_wide_kernel<float, false>), grid, block, 0, s, packed, seg_begin, (const float*)rows, W, gW, gbias, slab, row_shift, kd, E, (int)M, (int)D);
        } else {
            if (dot) hipLaunchKernelGGL((seg_wgrad_wide_kernel<bf16_t, true>), grid, block, 0, s, packed, seg_begin, (const bf16_t*)rows, W, gW, gbias, slab, row_shift, kd, E, (int)M, (int)D);
            else hipLaunchKernelGGL((seg_wgrad_wide_kernel<bf16_t, false>), grid, block, 0, s, packed, seg_begin, (const bf16_t*)rows, W, gW, gbias, slab, row_shift, kd, E, (int)M, (int)D);
        }
        SPHIP_CHECK_LAUNCH("segmented_weight_grad(wide)");
        if (!dot) return SPHIP_OK;
        // positions >= seg_begin[M] belong to inactive slots (zeroed by seg_pack_kernel); the
        // device-side bound is read by the kernel, the grid covers all E positions
        hipLaunchKernelGGL(seg_dcoeff_bounded_kernel, dim3((unsigned)cdiv(E, 256)), dim3(256), 0, s, packed, slab, E, seg_begin, (int)M, (int)(D / 256), dcoeff);
        SPHIP_CHECK_LAUNCH("segmented_weight_grad(dcoeff)");
        return SPHIP_OK;
    }
    if (!dot && !(force && (force[0] == 'f' || force[0] == 'w'))) {
        // the L2-sliced kernel, and next to it the wide kernel: a device-side word (the longest
        // segment) decides which of the two launches does the work, without a host round trip
        const bool use_long = wide_eligible(D) && big && !force;
        const LongTail lt = use_long ? long_tail(workspace, N, k, D) : LongTail{};
        const int* skew = use_long ? lt.hdr : nullptr;
        if (use_long)
            SPHIP_REQUIRE(list_long_latents(lt, seg_begin, M, s), SPHIP_ELAUNCH, "segmented_weight_grad: memset failed");
        const bool done = rows_dtype == SPHIP_F32
                              ? launch_sliced<float>(packed, seg_begin, rows, rows_sm, gW, gbias, row_shift, N, (int)k, (int)M, (int)D, s, 0.f, nullptr, skew, nullptr, nullptr, SfArgs{nullptr, 0.f, 0.f, 0.f}, nullptr, nullptr, packed4, accum)
                              : launch_sliced<bf16_t>(packed, seg_begin, rows, rows_sm, gW, gbias, row_shift, N, (int)k, (int)M, (int)D, s, 0.f, nullptr, skew, nullptr, nullptr, SfArgs{nullptr, 0.f, 0.f, 0.f}, nullptr, nullptr, packed4, accum);
        if (done) {
            if (use_long) {
                if (rows_dtype == SPHIP_F32)
                    reduce_long_latents<float, false, false>(lt, packed, packed4, seg_begin, rows_sm, nullptr, gW, nullptr, row_shift, N, k, D, s,
                                                             0.f, nullptr, SfArgs{nullptr, 0.f, 0.f, 0.f}, accum);
                else
                    reduce_long_latents<bf16_t, false, false>(lt, packed, packed4, seg_begin, rows_sm, nullptr, gW, nullptr, row_shift, N, k, D, s,
                                                              0.f, nullptr, SfArgs{nullptr, 0.f, 0.f, 0.f}, accum);
            }
            SPHIP_CHECK_LAUNCH("segmented_weight_grad(sliced)");
            return SPHIP_OK;
        }
    }
    SPHIP_REQUIRE(!accum, SPHIP_EUNSUPPORTED, "segmented_weight_grad_acc: the L2-sliced kernel does not take D=%lld", (long long)D);
    if (rows_dtype == SPHIP_F32) {
        if (dot) launch_nv<float, true>(packed, seg_begin, rows, W, gW, gbias, dcoeff, row_shift, (int)k, (int)M, (int)D, s);
        else launch_nv<float, false>(packed, seg_begin, rows, W, gW, gbias, dcoeff, row_shift, (int)k, (int)M, (int)D, s);
    } else {
        if (dot) launch_nv<bf16_t, true>(packed, seg_begin, rows, W, gW, gbias, dcoeff, row_shift, (int)k, (int)M, (int)D, s);
        else launch_nv<bf16_t, false>(packed, seg_begin, rows, W, gW, gbias, dcoeff, row_shift, (int)k, (int)M, (int)D, s);
    }
    SPHIP_CHECK_LAUNCH("segmented_weight_grad");
    return SPHIP_OK;
}

extern "C" int sphip_segment_coeff_sums(const uint32_t* perm, const int32_t* seg_begin, const void* coeff, int coeff_dtype,
                                        int64_t M, float* out, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(valid_dtype(coeff_dtype), SPHIP_EINVAL, "segment_coeff_sums: bad dtype");
    SPHIP_REQUIRE(perm && seg_begin && coeff && out, SPHIP_EINVAL, "segment_coeff_sums: null pointer");
    SPHIP_REQUIRE(M > 0 && M < (1ll << 31) - 256, SPHIP_EINVAL, "segment_coeff_sums: bad M");
    return launch_coeff_sums(perm, seg_begin, coeff, coeff_dtype, M, out, (hipStream_t)stream);
}

static int fused_weight_grad_step(const uint32_t* perm, const int32_t* seg_begin, const void* coeff, int coeff_dtype,
                                  const void* rows, int rows_dtype, const float* row_shift, int64_t N, int64_t k,
                                  int64_t M, int64_t D, float* param, float lr, void* bf16_shadow, void* workspace,
                                  size_t workspace_bytes, void* stream, sphip::SfArgs sf, const float* coeff_sums,
                                  float* bdec_partials) {
    using namespace sphip;
    SPHIP_REQUIRE(valid_dtype(coeff_dtype) && valid_dtype(rows_dtype), SPHIP_EINVAL, "segmented_weight_grad_signsgd: bad dtype");
    SPHIP_REQUIRE((coeff_sums == nullptr) == (bdec_partials == nullptr), SPHIP_EINVAL,
                  "segmented_weight_grad_signsgd: coeff_sums and bdec_partials go together");
    SPHIP_REQUIRE(perm && seg_begin && coeff && rows && param, SPHIP_EINVAL, "segmented_weight_grad_signsgd: null pointer");
    SPHIP_REQUIRE(N > 0 && k > 0 && M > 0 && D > 0, SPHIP_EINVAL, "segmented_weight_grad_signsgd: empty shape");
    SPHIP_REQUIRE(N * k < (1ll << 31) && M < (1ll << 31) - 4 && D < (1ll << 30), SPHIP_EUNSUPPORTED,
                  "segmented_weight_grad_signsgd: shape too large");
    SPHIP_REQUIRE(D % (rows_dtype == SPHIP_BF16 ? 8 : 4) == 0, SPHIP_EUNSUPPORTED,
                  "segmented_weight_grad_signsgd: D=%lld must be a multiple of %d", (long long)D,
                  rows_dtype == SPHIP_BF16 ? 8 : 4);
    SPHIP_REQUIRE(((uintptr_t)rows & 15) == 0 && ((uintptr_t)param & 15) == 0 && ((uintptr_t)row_shift & 15) == 0 &&
                      ((uintptr_t)bf16_shadow & 7) == 0,
                  SPHIP_EINVAL, "segmented_weight_grad_signsgd: rows/param/row_shift must be 16-byte aligned");
    const size_t need = sphip_weight_grad_workspace_bytes(N, k, D);
    SPHIP_REQUIRE(workspace != nullptr && workspace_bytes >= need, SPHIP_EWORKSPACE,
                  "segmented_weight_grad_signsgd: workspace %zu < %zu bytes", workspace_bytes, need);
    SPHIP_REQUIRE(((uintptr_t)workspace & 255) == 0, SPHIP_EINVAL, "segmented_weight_grad_signsgd: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int64_t E = N * k;
    uint2* packed = (uint2*)workspace;
    void* rows_sm = (char*)workspace + packed_bytes(N, k);
    // (bf16 coefficients -- the gradient of bf16 activations is bf16 -- also go out as 4-byte entries)
    unsigned* packed4 = (coeff_dtype == SPHIP_BF16 && N <= 65536 && getenv("SPHIP_WGRAD_PK8") == nullptr)
                            ? (unsigned*)((char*)workspace + packed8_bytes(N, k)) : nullptr;
    if (coeff_dtype == SPHIP_F32)
        hipLaunchKernelGGL((seg_pack_kernel<float>), dim3((unsigned)cdiv(E, 256)), dim3(256), 0, s, perm, (const float*)coeff, E, seg_begin, (int)M, packed, (float*)nullptr);
    else
        hipLaunchKernelGGL((seg_pack_kernel<bf16_t>), dim3((unsigned)cdiv(E, 256)), dim3(256), 0, s, perm, (const bf16_t*)coeff, E, seg_begin, (int)M, packed, (float*)nullptr, packed4, make_fastdiv((unsigned)k));
    const bool big = N * D * (int64_t)dtype_size(rows_dtype) >= (8ll << 20);
    const bool use_long = wide_eligible(D) && big;
    const LongTail lt = use_long ? long_tail(workspace, N, k, D) : LongTail{};
    const int* skew = use_long ? lt.hdr : nullptr;
    if (use_long)
        SPHIP_REQUIRE(list_long_latents(lt, seg_begin, M, s), SPHIP_ELAUNCH, "segmented_weight_grad_signsgd: memset failed");
    const bool done = rows_dtype == SPHIP_F32
                          ? launch_sliced<float, true>(packed, seg_begin, rows, rows_sm, param, nullptr, row_shift, N, (int)k, (int)M, (int)D, s, lr, (bf16_t*)bf16_shadow, skew, nullptr, nullptr, sf, coeff_sums, bdec_partials, packed4)
                          : launch_sliced<bf16_t, true>(packed, seg_begin, rows, rows_sm, param, nullptr, row_shift, N, (int)k, (int)M, (int)D, s, lr, (bf16_t*)bf16_shadow, skew, nullptr, nullptr, sf, coeff_sums, bdec_partials, packed4);
    SPHIP_REQUIRE(done, SPHIP_EUNSUPPORTED, "segmented_weight_grad_signsgd: shape not supported by the sliced kernel");
    if (use_long) {
        // (after the sliced kernel: its gb^T W partials read the old rows of the long latents too)
        if (rows_dtype == SPHIP_F32)
            reduce_long_latents<float, false, true>(lt, packed, packed4, seg_begin, rows_sm, nullptr, param, nullptr, row_shift, N, k, D, s, lr, (bf16_t*)bf16_shadow, sf);
        else
            reduce_long_latents<bf16_t, false, true>(lt, packed, packed4, seg_begin, rows_sm, nullptr, param, nullptr, row_shift, N, k, D, s, lr, (bf16_t*)bf16_shadow, sf);
    }
    SPHIP_CHECK_LAUNCH("segmented_weight_grad_signsgd");
    return SPHIP_OK;
}

extern "C" int sphip_segmented_weight_grad_signsgd(const uint32_t* perm, const int32_t* seg_begin, const void* coeff,
                                                   int coeff_dtype, const void* rows, int rows_dtype,
                                                   const float* row_shift, int64_t N, int64_t k, int64_t M, int64_t D,
                                                   float* param, float lr, void* bf16_shadow, const float* coeff_sums,
                                                   float* bdec_partials, void* workspace, size_t workspace_bytes,
                                                   void* stream) {
    return fused_weight_grad_step(perm, seg_begin, coeff, coeff_dtype, rows, rows_dtype, row_shift, N, k, M, D, param, lr,
                                  bf16_shadow, workspace, workspace_bytes, stream, sphip::SfArgs{nullptr, 0.f, 0.f, 0.f},
                                  coeff_sums, bdec_partials);
}

extern "C" int sphip_segmented_weight_grad_schedulefree(const uint32_t* perm, const int32_t* seg_begin, const void* coeff,
                                                        int coeff_dtype, const void* rows, int rows_dtype,
                                                        const float* row_shift, int64_t N, int64_t k, int64_t M,
                                                        int64_t D, float* param, float* z, float lr, float beta, float c,
                                                        void* bf16_shadow, const float* coeff_sums,
                                                        float* bdec_partials, void* workspace, size_t workspace_bytes,
                                                        void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(z != nullptr && ((uintptr_t)z & 15) == 0, SPHIP_EINVAL,
                  "segmented_weight_grad_schedulefree: z must be a 16-byte aligned buffer");
    SPHIP_REQUIRE(beta > 0.f && beta <= 1.f, SPHIP_EINVAL, "segmented_weight_grad_schedulefree: bad beta");
    return fused_weight_grad_step(perm, seg_begin, coeff, coeff_dtype, rows, rows_dtype, row_shift, N, k, M, D, param, lr,
                                  bf16_shadow, workspace, workspace_bytes, stream, SfArgs{z, 1.f - 1.f / beta, c, 1.f - beta},
                                  coeff_sums, bdec_partials);
}

// table (M, D) fp32 row-major -> [D/32][M][32 floats]: one contiguous M x 128 B block per column
// slice, the layout sphip_gather_rows_weighted_l2 gathers from (gather_l2.hip).
// A workgroup moves 32 rows x 8 slices: thread (row, 16-B piece) walks the 8 slices, so the reads
// cover 1 KiB of each row and every store instruction writes 4 KiB contiguous bytes of one slice
// (the element-order kernel above scatters a wave's store over 8 slices 128 * M bytes apart:
// 0.97 TB/s on a 2 GiB table).
namespace sphip {
__global__ __launch_bounds__(256) void slice_major_tiled_kernel(const float* __restrict__ src, uint4* __restrict__ dst,
                                                                int64_t R, int D, int num_slices) {
    // 32 rows x 8 slices (1 KiB of each row) per step through LDS: a wave reads 1 KiB of one row
    // and writes 1 KiB of one slice (8 rows x 128 B), both contiguous
    constexpr int ROWCH = 65;                               // 16-byte chunks per LDS row (64 + 1 pad)
    __shared__ uint4 tile[32 * ROWCH];
    const int64_t row0 = (int64_t)blockIdx.x * 32;
    const int t = threadIdx.x;
    for (int s0 = 0; s0 < num_slices; s0 += 8) {
        const int chunks = min(8, num_slices - s0) * 8;    // valid chunks of the 1 KiB window
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = r * 4 + (t >> 6), c = t & 63;
            if (row0 + row < R && c < chunks)
                tile[row * ROWCH + c] = *reinterpret_cast<const uint4*>(src + (row0 + row) * D + s0 * 32 + c * 4);
        }
        __syncthreads();
        const int row = t >> 3, sub = t & 7;
        if (row0 + row < R) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < num_slices) dst[((int64_t)(s0 + u) * R + row0 + row) * 8 + sub] = tile[row * ROWCH + u * 8 + sub];
        }
        __syncthreads();
    }
}
}  // namespace sphip

extern "C" int sphip_slice_major_f32(const float* src, float* dst, int64_t R, int64_t D, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(src && dst, SPHIP_EINVAL, "slice_major_f32: null pointer");
    SPHIP_REQUIRE(R > 0 && D > 0 && D % 32 == 0 && D < (1ll << 31), SPHIP_EINVAL,
                  "slice_major_f32: needs D %% 32 == 0, got R=%lld D=%lld", (long long)R, (long long)D);
    SPHIP_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, SPHIP_EINVAL, "slice_major_f32: operands must be 16-byte aligned");
    const int num_slices = (int)(D / 32);
    SPHIP_REQUIRE(cdiv(R, 32) < (1ll << 31) && cdiv(num_slices, 8) <= 65535, SPHIP_EUNSUPPORTED, "slice_major_f32: shape too large");
    hipLaunchKernelGGL(slice_major_tiled_kernel, dim3((unsigned)cdiv(R, 32)), dim3(256), 0,
                       (hipStream_t)stream, src, (uint4*)dst, R, (int)D, num_slices);
    SPHIP_CHECK_LAUNCH("slice_major_f32");
    return SPHIP_OK;
}
