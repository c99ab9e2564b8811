// Encoder pre-activations, large-shape bf16 kernel: 256(latent) x 256(token) x 64(K) tile,
// 512 threads = 8 waves (2 latent groups x 4 token groups, 128 x 64 outputs per wave),
// v_mfma_f32_16x16x32_bf16, operands staged by LDS-DMA (buffer_load_dwordx4 ... lds) into a
// 128 KiB double-buffered image, counted vmcnt waits and raw s_barriers so up to a K tile of
// staging stays in flight across barriers, and the two wave groups that share a SIMD run
// staggered by one barrier (one group's MFMA segment overlaps the other's LDS-read segment).
// Reference semantics: FusedEncoder.forward, sparsify/fused_encoder.py:29, under bf16
// autocast (sparse_coder.py:205-209); same contract as encoder_gemm.hip.
//
// LDS image of one K tile (64 KiB, two of them):
//     X    : 256 token rows x 128 B                                         -> read in L0
//     W-lo : rows [g*128 +  0, g*128 +  64) of both latent groups g (16 KiB) -> read in L0
//     W-hi : rows [g*128 + 64, g*128 + 128)                        (16 KiB) -> read in L1
// Every region is a linear [rows][8 chunks of 16 B] array whose chunk index is XOR-swizzled
// with (row>>1)&7 (conflict-free ds_read_b128 for the 16x16x32 fragment shape). LDS-DMA
// writes linearly (wave base + lane*16), so the swizzle is applied to the per-lane SOURCE
// address.
//
// Schedule per K tile t: four segments, each closed by a barrier (code segment s = 4t + i):
//     L0: ds_read X(t) (8) + W-lo(t) (8)        | vmcnt(6): W-hi(t) landed
//     M0: DMA W-hi(t+1)                         | 32 MFMA (w-lo x x)
//     L1: ds_read W-hi(t) (8)                   | vmcnt(2): X(t+1), W-lo(t+1) landed
//     M1: DMA X(t+2), W-lo(t+2)                 | 32 MFMA (w-hi x x)
// Wave group 1 runs one barrier behind group 0, so while one group is in an M segment its
// SIMD partner is in an L segment. With that skew:
// RAW: a region read in code segment r is awaited (own vmcnt) in a segment <= r-2, so every
//      wave has passed a barrier after its wait before the leading group starts r.
// WAR: a region read in code segment r is overwritten by DMA issued in a segment >= r+3,
//      after the lagging group has completed (lgkmcnt) its reads of r.
#include "common.hpp"
#include <stdlib.h>

namespace sphip {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int G_TILE = 256;
// MFMA groups of 4 issued after an M segment's closing barrier (raising the barrier hides its
// release latency under the last MFMAs). A/B on one box, interleaved, with the non-temporal
// epilogue stores: 0 -> 12.27 ms, 1 -> 12.02 ms (2 was slower: 12.6 against 12.1 ms earlier)
constexpr int EARLY = 1;
constexpr int G_SCHED_DEFAULT = 0;            // see SCHED at gemm256_tile
constexpr int G_BK = 64;                      // bf16 elements per K tile (128 B)
constexpr int R_X = 0;                        // region offsets inside a K-tile buffer
constexpr int R_WLO = 256 * 128;              // 32 KiB
constexpr int R_WHI = R_WLO + 128 * 128;      // 48 KiB
constexpr int BUF_BYTES = R_WHI + 128 * 128;  // 64 KiB
constexpr int EPI_ROW = 272;                  // epilogue staging: 128 bf16 + 16 B pad per token row
constexpr int EPI_WAVE = 64 * EPI_ROW;        // 17 KiB per wave
constexpr int G_LDS_BYTES = (2 * BUF_BYTES > 8 * EPI_WAVE) ? 2 * BUF_BYTES : 8 * EPI_WAVE;

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

#define SPHIP_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__device__ __forceinline__ void seg_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// Candidate emission (OUTF 4, sphip_encoder_preacts_emit) and the predicated launch
// (sphip_encoder_preacts_if): see the epilogue below and lazy.hip.
struct EmitArgs {
    const unsigned short* tau = nullptr;   // per token: bits of the bf16 threshold (a relu output, >= +0)
    unsigned* cnt = nullptr;               // per token: chunks appended so far
    u32x4_t* list = nullptr;               // [N][cap] chunks of 8 consecutive bf16 pre-activations
    unsigned* ids = nullptr;               // [N][cap] index of a chunk's first latent / 8
    int* flag = nullptr;                   // raised when a token's list overflows
    const int* run_if = nullptr;           // predicated launch: every workgroup leaves when *run_if == 0
    int cap = 0;
    unsigned id_base = 0;                  // (first latent of this launch's block) / 8
    int diag = 0;                          // SPHIP_EMIT_DIAG (timing experiments): 1 no atomics, 2 no stores, 4 no pass 1
};

// OUTF: 0 = the encoder's epilogue (bias, ReLU, bf16, block maxima); 1 = store the fp32
// accumulators as they are (P is float*); 2 = add them to the fp32 values already in P
// (sphip_gemm_bf16_f32acc: products whose fp32 operand was split into bf16 terms);
// 3 = bf16 P = rnd(rnd(alpha * rnd(acc)) + rnd(beta * C)) with C = `bias` (row stride tilesT, may
// be null), the Newton-Schulz products of Muon (sphip_gemm_bf16_axpy, encoder_gemm.hip);
// 4 = the encoder's values are not stored: every 16-byte chunk (8 latents) of a token that holds a
// value above the token's threshold is appended to the token's candidate list.
// SCHED: 0 = the staging DMA is issued between the MFMAs of the M segments (header above);
// 1 = it is issued in the L segments, behind the segment's fragment reads: the wave that issues
// it is then the one waiting at the barrier while its SIMD partner runs MFMAs, so the M segments
// are bare MFMA chains. W-hi(t+1) goes out in L0(t), X(t+2) / W-lo(t+2) in L1(t), each one
// segment earlier than under SCHED 0 (code segment r + 2 instead of r + 3 after the last read of
// the region in segment r): the L segments therefore complete their fragment reads (lgkmcnt(0))
// BEFORE their closing barrier, so that when the leading group enters segment r + 2 the lagging
// group has passed the barrier closing its segment r with its reads done.
template <bool RELU, int DIAG, int OUTF, int SCHED = 0>
__device__ __forceinline__ void gemm256_tile(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                             const bf16_t* __restrict__ bias, bf16_t* __restrict__ P, int N, int D,
                                             int M, int tm, int tn, bf16_t* __restrict__ tile_max, int64_t ldp,
                                             int tilesT, unsigned* dbg, char* smem, float alpha = 1.f,
                                             float beta = 0.f, const EmitArgs& ea = EmitArgs()) {
    const int m0 = (DIAG & 16) ? (tm & 1) * G_TILE : tm * G_TILE, n0 = (DIAG & 16) ? (tn & 1) * G_TILE : tn * G_TILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // OUTF 4: lane L holds the threshold of the wave's token L as T = (t | 0x8000) in both
    // halves of a dword: for two packed non-negative bf16 values w, bits 15 / 31 of T - w are
    // clear exactly where a value exceeds t. Tokens past N get a threshold nothing exceeds.
    unsigned tau_T = 0xFFFFFFFFu;
    if (OUTF == 4) {
        const int n = n0 + wn * 64 + lane;
        const unsigned t = n < N ? (unsigned)ea.tau[n] : 0x7FFFu;
        tau_T = (t | 0x8000u) * 0x10001u;
    }
    const int r16 = lane & 15, q4 = lane >> 4;
    const int kTiles = D / G_BK;

    // ---- staging: thread owns 16-B chunk c = tid + 512*i of a region (X: i < 4, W halves: i < 2).
    // hr = c >> 3 is the region row; (hr >> 1) & 7 does not depend on i (rows advance by 64).
    // Sources go through buffer descriptors of the two operand tiles: one per-lane byte offset
    // (row hr0, swizzled chunk) serves every piece, the piece's row block and the K tile are
    // a scalar offset, and rows past the end of X / W read as zeros (descriptor range check).
    const int hr0 = tid >> 3;
    const int logical = (tid & 7) ^ ((hr0 >> 1) & 7);            // source chunk for this LDS slot
    const int dst0 = 64 * wave * 16;                             // wave-uniform base of the 1 KiB piece
    const unsigned voff = (unsigned)hr0 * (unsigned)D * 2u + (unsigned)logical * 16u;
    const unsigned rowblk = 64u * (unsigned)D * 2u;              // bytes of a 64-row block
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(X + (int64_t)n0 * D), 0, (int)(min(G_TILE, N - n0) * D * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(W + (int64_t)m0 * D), 0, (int)(min(G_TILE, M - m0) * D * 2), 0x00020000);
    auto stage_x1 = [&](int t, int i) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lptr_t)(smem + (t & 1) * BUF_BYTES + R_X + dst0 + 512 * 16 * i),
                                                 16, voff, (unsigned)t * (G_BK * 2) + (unsigned)i * rowblk, 0, 0);
    };
    // W-lo piece i holds tile rows i*128 + [0,64), W-hi piece i rows i*128 + 64 + [0,64)
    auto stage_w1 = [&](int hi, int t, int i) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            wrsrc, (lptr_t)(smem + (t & 1) * BUF_BYTES + (hi ? R_WHI : R_WLO) + dst0 + 512 * 16 * i), 16, voff,
            (unsigned)t * (G_BK * 2) + (unsigned)(2 * i + hi) * rowblk, 0, 0);
    };
    auto stage_x = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) stage_x1(t, i);
    };
    auto stage_w = [&](int hi, int t) {
        stage_w1(hi, t, 0);
        stage_w1(hi, t, 1);
    };

    // ---- fragment reads. They are issued from inline asm: hipcc's waitcnt pass would put an
    // s_waitcnt vmcnt(0) in front of every compiler-visible ds_read that follows an LDS-DMA
    // (it cannot tell which DMA a read depends on) and drain the staging pipeline each
    // segment. The reads' completion is awaited by the explicit lgkmcnt(0) below, which names
    // every destination register so no consumer can be scheduled above it.
    // Per-lane byte address inside a region, k-step ks: chunk = 4*ks + q4. The row part of
    // the XOR ((row>>1)&7 = (r16>>1)&7) does not depend on the 16-row tile index, so the
    // tiles of a sub-block are reached with immediate offsets (tile stride 16*128 B).
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const int swz = (r16 >> 1) & 7;
    unsigned wbase[2], xbase[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        wbase[ks] = lds0 + (wm * 64 + r16) * 128 + (((4 * ks + q4) ^ swz) << 4);
        xbase[ks] = lds0 + (wn * 64 + r16) * 128 + (((4 * ks + q4) ^ swz) << 4);
    }

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    u32x4_t wf[4][2], xf[4][2];
    unsigned tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned tlast = 0;
#define SPHIP_STAMP(i)                                                        \
    if (DIAG & 32) {                                                          \
        __builtin_amdgcn_sched_barrier(0);                                    \
        const unsigned now_ = (unsigned)__builtin_readcyclecounter();         \
        tacc[i] += now_ - tlast;                                              \
        tlast = now_;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                    \
    }

#define SPHIP_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define SPHIP_READ4(F, REGION)                          \
    SPHIP_DSR(F[0][0], a0, (REGION) + 0 * 2048);        \
    SPHIP_DSR(F[0][1], a1, (REGION) + 0 * 2048);        \
    SPHIP_DSR(F[1][0], a0, (REGION) + 1 * 2048);        \
    SPHIP_DSR(F[1][1], a1, (REGION) + 1 * 2048);        \
    SPHIP_DSR(F[2][0], a0, (REGION) + 2 * 2048);        \
    SPHIP_DSR(F[2][1], a1, (REGION) + 2 * 2048);        \
    SPHIP_DSR(F[3][0], a0, (REGION) + 3 * 2048);        \
    SPHIP_DSR(F[3][1], a1, (REGION) + 3 * 2048)
#define SPHIP_LGKM0(F)                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                           \
                 : "+v"(F[0][0]), "+v"(F[0][1]), "+v"(F[1][0]), "+v"(F[1][1]), "+v"(F[2][0]),      \
                   "+v"(F[2][1]), "+v"(F[3][0]), "+v"(F[3][1]))

    // 32 MFMAs of one M segment; the segment's DMA instructions (NDMA of them, issued by
    // dma(j)) are spread between the MFMAs so that their issue cost (M0 write, address add,
    // VMEM issue: ~45 cycles each when issued back to back at the head) hides under the
    // matrix pipe instead of delaying the first MFMA.
    auto mma = [&](int mbase, int ndma, auto dma) {
        if (DIAG & 2) {
            for (int j = 0; j < ndma; ++j) dma(j);
            seg_barrier();
            return;
        }
        constexpr int VAR = DIAG >> 6;               // schedule variants (diagnostic builds)
        if (VAR != 1) __builtin_amdgcn_s_setprio(VAR == 2 ? 3 : 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mbase + mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8_t, wf[mt][ks]), __builtin_bit_cast(bf16x8_t, xf[nt][ks]),
                        acc[mbase + mt][nt], 0, 0, 0);
                const int j = ks * 4 + mt;
                if (VAR == 3) {                      // DMA issue one group later
                    if (j >= 1 && j - 1 < ndma) {
                        __builtin_amdgcn_sched_barrier(0);
                        dma(j - 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else if (VAR == 4) {               // two DMAs per group
                    __builtin_amdgcn_sched_barrier(0);
                    if (2 * j < ndma) dma(2 * j);
                    if (2 * j + 1 < ndma) dma(2 * j + 1);
                    __builtin_amdgcn_sched_barrier(0);
                } else if (j < ndma) {
                    __builtin_amdgcn_sched_barrier(0);
                    dma(j);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the segment's closing barrier is raised early: its release latency then
                // overlaps this wave's last MFMAs instead of idling the matrix pipe
                if (j == 7 - EARLY) seg_barrier();
            }
        if (VAR != 1) __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: K tile 0 completely plus X / W-lo of K tile 1; X(0), W-lo(0) retired
    stage_x(0);
    stage_w(0, 0);
    stage_w(1, 0);
    if (kTiles > 1) {
        stage_x(1);
        stage_w(0, 1);
        SPHIP_VMCNT(8);
    } else {
        SPHIP_VMCNT(2);
    }
    seg_barrier();
    if (wm == 1) seg_barrier();          // stagger the second wave group by one barrier
    if (DIAG & 32) tlast = (unsigned)__builtin_readcyclecounter();

    // One K tile. Outstanding DMA per wave, in issue order, at the two waits:
    //   L0: [W-hi(t) 2][X(t+1) 4][W-lo(t+1) 2]      L1: [X(t+1) 4][W-lo(t+1) 2][W-hi(t+1) 2]
#define SPHIP_KTILE(ISSUE_HI, ISSUE_X, WAIT_L0, WAIT_L1)                                         \
    {                                                                                           \
        const unsigned bo = (unsigned)(t & 1) * BUF_BYTES;                                      \
        /* L0 */                                                                                \
        if (!(DIAG & 4)) {                                                                      \
            { const unsigned a0 = xbase[0] + bo, a1 = xbase[1] + bo; SPHIP_READ4(xf, R_X); }    \
            { const unsigned a0 = wbase[0] + bo, a1 = wbase[1] + bo; SPHIP_READ4(wf, R_WLO); }  \
        }                                                                                       \
        if (SCHED == 1 && (ISSUE_HI) && !(DIAG & 1)) stage_w(1, t + 1);                         \
        WAIT_L0;                                                                                \
        if (SCHED == 1) { SPHIP_LGKM0(xf); SPHIP_LGKM0(wf); }                                   \
        SPHIP_STAMP(0)                                                                          \
        seg_barrier();                                                                          \
        SPHIP_STAMP(1)                                                                          \
        /* M0 */                                                                                \
        if (SCHED == 0) { SPHIP_LGKM0(xf); SPHIP_LGKM0(wf); }                                   \
        SPHIP_STAMP(2)                                                                          \
        mma(0, (SCHED == 0 && (ISSUE_HI) && !(DIAG & 1)) ? 2 : 0, [&](int j) { stage_w1(1, t + 1, j); }); \
        SPHIP_STAMP(3)                                                                          \
        SPHIP_STAMP(4)                                                                          \
        /* L1 */                                                                                \
        if (!(DIAG & 4)) {                                                                      \
            const unsigned a0 = wbase[0] + bo, a1 = wbase[1] + bo;                              \
            SPHIP_READ4(wf, R_WHI);                                                             \
        }                                                                                       \
        if (SCHED == 1 && (ISSUE_X) && !(DIAG & 1)) { stage_x(t + 2); stage_w(0, t + 2); }      \
        WAIT_L1;                                                                                \
        if (SCHED == 1) { SPHIP_LGKM0(wf); }                                                    \
        SPHIP_STAMP(5)                                                                          \
        seg_barrier();                                                                          \
        SPHIP_STAMP(6)                                                                          \
        /* M1 */                                                                                \
        if (SCHED == 0) { SPHIP_LGKM0(wf); }                                                    \
        SPHIP_STAMP(7)                                                                          \
        mma(4, (SCHED == 0 && (ISSUE_X) && !(DIAG & 1)) ? 6 : 0, [&](int j) {                   \
            if (j < 4) stage_x1(t + 2, j);                                                      \
            else stage_w1(0, t + 2, j - 4);                                                     \
        });                                                                                     \
        SPHIP_STAMP(8)                                                                          \
        SPHIP_STAMP(9)                                                                          \
    }

    // outstanding DMA per wave at the two waits under SCHED 1 (the segment's own issue included):
    //   L0: [W-hi(t) 2][X(t+1) 4][W-lo(t+1) 2][W-hi(t+1) 2]   L1: [X(t+1) 4][W-lo(t+1) 2][W-hi(t+1) 2][X(t+2) 4][W-lo(t+2) 2]
    int t = 0;
    if (SCHED == 1) {
        for (; t < kTiles - 2; ++t) SPHIP_KTILE(true, true, SPHIP_VMCNT(8), SPHIP_VMCNT(8))
        if (kTiles > 1) {
            SPHIP_KTILE(true, false, SPHIP_VMCNT(8), SPHIP_VMCNT(2))
            ++t;
        }
    } else {
    for (; t < kTiles - 2; ++t) SPHIP_KTILE(true, true, SPHIP_VMCNT(6), SPHIP_VMCNT(2))
    if (kTiles > 1) {
        SPHIP_KTILE(true, false, SPHIP_VMCNT(6), SPHIP_VMCNT(2))
        ++t;
    }
    }
    SPHIP_KTILE(false, false, SPHIP_VMCNT(0), (void)0)
#undef SPHIP_KTILE

    if (wm == 0) seg_barrier();          // matches the stagger barrier of the other group
    if ((DIAG & 32) && dbg != nullptr && (blockIdx.x % 4099) == 7 && lane == 0) {
        unsigned* o = dbg + ((blockIdx.x / 4099) * 8 + wave) * 10;
        for (int i = 0; i < 10; ++i) o[i] = tacc[i];
    }

    if (DIAG & 512) {                       // diagnostic: no epilogue at all (the accumulators stay live)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    if (OUTF == 3) {
        const int64_t ldc = tilesT;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int m = m0 + wm * 128 + (mi >> 2) * 64 + (mi & 3) * 16 + q4 * 4;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = n0 + wn * 64 + ni * 16 + r16;
                if (n >= N || m >= M) continue;                    // M % 4 == 0 (host check)
                float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bias != nullptr) c4 = load4(bias + (int64_t)n * ldc + m);
                const float cv[4] = {c4.x, c4.y, c4.z, c4.w};
                bf16_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = bf16_to_f32(f32_to_bf16(acc[mi][ni][j]));
                    if (alpha != 1.f) t = bf16_to_f32(f32_to_bf16(alpha * t));
                    if (bias != nullptr) t += bf16_to_f32(f32_to_bf16(beta * cv[j]));
                    o[j] = f32_to_bf16(t);
                }
                *reinterpret_cast<uint2*>(P + (int64_t)n * ldp + m) =
                    make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16));
            }
        }
        return;
    }
    if (OUTF == 1 || OUTF == 2) {
        // fp32 output: the lane's 4 consecutive columns of one row per accumulator tile, 16-B
        // pieces straight from the registers (these products write N x M / K-fold fewer bytes
        // than they read, the store shape does not matter)
        float* Pf = reinterpret_cast<float*>(P);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int m = m0 + wm * 128 + (mi >> 2) * 64 + (mi & 3) * 16 + q4 * 4;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = n0 + wn * 64 + ni * 16 + r16;
                if (n >= N || m >= M) continue;                    // M % 4 == 0 (host check)
                float4* dst = reinterpret_cast<float4*>(Pf + (int64_t)n * ldp + m);
                float4 v = make_float4(acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]);
                if (OUTF == 2) {
                    const float4 o = *dst;
                    v.x += o.x, v.y += o.y, v.z += o.z, v.w += o.w;
                }
                *dst = v;
            }
        }
        return;
    }

    // ---- epilogue: + bias, relu, one rounding to bf16. A lane owns 4 consecutive latents of
    // one token per accumulator tile (C/D map of 16x16: col = lane&15, row = 4*(lane>>4)+reg),
    // i.e. 8-B pieces; the wave's 64 x 128 sub-tile goes through a private LDS area (the
    // staging image is dead after the barrier above) and leaves as 256-B row segments.
    char* epi = smem + wave * EPI_WAVE;
    float bv[8][4];
    // The lane's bias values (4 consecutive latents for each of its 8 accumulator rows). With an
    // 8-byte aligned bias and M % 4 == 0 they arrive as eight 8-byte buffer loads issued back to
    // back (one round trip; the descriptor returns zeros for latents past M); the element-wise
    // form branches around every load and waits for each in turn (eight dependent round trips per tile).
    if (bias != nullptr && (((uintptr_t)bias & 7) == 0) && (M & 3) == 0) {
        const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc((void*)bias, 0, M * 2, 0x00020000);
        typedef unsigned bu2_t __attribute__((ext_vector_type(2)));
        bu2_t raw[8];
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int m = m0 + wm * 128 + (mi >> 2) * 64 + (mi & 3) * 16 + q4 * 4;
            raw[mi] = __builtin_amdgcn_raw_buffer_load_b64(brsrc, m * 2, 0, 0);
        }
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            bv[mi][0] = __uint_as_float(raw[mi][0] << 16), bv[mi][1] = __uint_as_float(raw[mi][0] & 0xffff0000u);
            bv[mi][2] = __uint_as_float(raw[mi][1] << 16), bv[mi][3] = __uint_as_float(raw[mi][1] & 0xffff0000u);
        }
    } else {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int m = m0 + wm * 128 + (mi >> 2) * 64 + (mi & 3) * 16 + q4 * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[mi][j] = 0.f;
            if (bias == nullptr) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (m + j < M) bv[mi][j] = bf16_to_f32(bias[m + j]);
        }
    }
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int ml = (mi >> 2) * 64 + (mi & 3) * 16 + q4 * 4;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            bf16_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = acc[mi][ni][j] + bv[mi][j];
                o[j] = f32_to_bf16(RELU ? (v > 0.f ? v : 0.f) : v);
            }
            *reinterpret_cast<uint2*>(epi + (ni * 16 + r16) * EPI_ROW + ml * 2) =
                make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (DIAG & 8) return;
    const int chunk = lane & 15;
    const int mcol = m0 + wm * 128 + chunk * 8;
    if (OUTF == 4) {
        // The 16-byte pad behind a token's 256 staged bytes holds {T, chunks hit, list position}.
        // Pass 1 marks the chunks holding a value above the threshold (one ballot per four
        // tokens, kept in scalar registers) and counts them per token; lane L then reserves
        // token L's share of its list with one returning atomic; pass 2 stores the marked chunks.
        *reinterpret_cast<unsigned*>(epi + lane * EPI_ROW + 256) = tau_T;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        unsigned long long hm[16];
        if (ea.diag & 4) return;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int tok = i * 4 + (lane >> 4);
            const uint4 v = *reinterpret_cast<const uint4*>(epi + tok * EPI_ROW + chunk * 16);
            const unsigned T = *reinterpret_cast<const unsigned*>(epi + tok * EPI_ROW + 256);
            const unsigned below = (T - v.x) & (T - v.y) & (T - v.z) & (T - v.w) & 0x80008000u;
            hm[i] = __ballot(below != 0x80008000u);
            const unsigned grp = (unsigned)(hm[i] >> ((lane >> 4) * 16)) & 0xffffu;
            if (chunk == 0) *reinterpret_cast<unsigned*>(epi + tok * EPI_ROW + 260) = (unsigned)__popc(grp);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            const unsigned mine = *reinterpret_cast<const unsigned*>(epi + lane * EPI_ROW + 260);
            const int n = n0 + wn * 64 + lane;
            unsigned base = 0;
            if (mine != 0 && n < N && !(ea.diag & 1)) {
                base = __hip_atomic_fetch_add(ea.cnt + n, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (base + mine > (unsigned)ea.cap) *ea.flag = 1;
            }
            *reinterpret_cast<unsigned*>(epi + lane * EPI_ROW + 264) = base;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (hm[i] == 0ull) continue;                             // wave-uniform
            const int tok = i * 4 + (lane >> 4);
            const int n = n0 + wn * 64 + tok;
            const unsigned grp = (unsigned)(hm[i] >> ((lane >> 4) * 16)) & 0xffffu;
            if (((grp >> chunk) & 1u) == 0u || n >= N) continue;
            const unsigned pos = *reinterpret_cast<const unsigned*>(epi + tok * EPI_ROW + 264) +
                                 (unsigned)__popc(grp & ((1u << chunk) - 1u));
            if (pos >= (unsigned)ea.cap || (ea.diag & 2)) continue;  // overflow: the flag is up
            const uint4 v = *reinterpret_cast<const uint4*>(epi + tok * EPI_ROW + chunk * 16);
            const int64_t at = (int64_t)n * ea.cap + pos;
            const u32x4_t nv = {v.x, v.y, v.z, v.w};
            ea.list[at] = nv;
            ea.ids[at] = ea.id_base + (unsigned)(mcol >> 3);
        }
        return;
    }
    // ldp: row stride of P; tilesT: row length of tile_max (both those of the whole matrix when
    // this launch computes a block of latents, sphip_encoder_preacts_block)
    const bool vec = (M & 7) == 0 && (ldp & 7) == 0;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int tok = i * 4 + (lane >> 4);
        const int n = n0 + wn * 64 + tok;
        const uint4 v = *reinterpret_cast<const uint4*>(epi + tok * EPI_ROW + chunk * 16);
        if (tile_max != nullptr) {
            // maximum of the token's 128 latents held by these 16 lanes, in the order-preserving
            // key domain of the top-k kernel (common.hpp)
            unsigned kmax;
            if (RELU) {
                // relu output is >= +0 (or NaN): the raw bf16 bits already order like the keys,
                // and the zeros staged for columns past M cannot raise a maximum
                typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
                const u16x2_t a = __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, v.x), __builtin_bit_cast(u16x2_t, v.y));
                const u16x2_t b2 = __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, v.z), __builtin_bit_cast(u16x2_t, v.w));
                const u16x2_t c2 = __builtin_elementwise_max(a, b2);
                kmax = max((unsigned)c2.x, (unsigned)c2.y) | 0x8000u;
            } else {
                const unsigned w4[4] = {v.x, v.y, v.z, v.w};
                kmax = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned key = to_key((bf16_t)(w4[j >> 1] >> ((j & 1) * 16)));
                    kmax = max(kmax, (mcol + j < M) ? key : 0u);
                }
            }
            // 16-lane maximum with DPP row operations (xor 1, xor 2, half mirror, mirror)
            kmax = max(kmax, (unsigned)__builtin_amdgcn_update_dpp(0, (int)kmax, 0xB1, 0xF, 0xF, true));
            kmax = max(kmax, (unsigned)__builtin_amdgcn_update_dpp(0, (int)kmax, 0x4E, 0xF, 0xF, true));
            kmax = max(kmax, (unsigned)__builtin_amdgcn_update_dpp(0, (int)kmax, 0x141, 0xF, 0xF, true));
            kmax = max(kmax, (unsigned)__builtin_amdgcn_update_dpp(0, (int)kmax, 0x140, 0xF, 0xF, true));
            // parked in the token's pad; the wave's 64 maxima leave as one 128-byte line below
            if (chunk == 0) *reinterpret_cast<unsigned short*>(epi + tok * EPI_ROW + 256) = KeyOf<bf16_t>::value(kmax);
        }
        if (n >= N || mcol >= M) continue;
        bf16_t* dst = P + (int64_t)n * ldp + mcol;
        if (vec) {
            // 4 GiB written once per launch and read back only in part (top-k): non-temporal, the
            // operand tiles keep L2
            const u32x4_t nv = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(nv, reinterpret_cast<u32x4_t*>(dst));
        } else {
            const unsigned w4[4] = {v.x, v.y, v.z, v.w};
            for (int j = 0; j < 8 && mcol + j < M; ++j) dst[j] = (bf16_t)(w4[j >> 1] >> ((j & 1) * 16));
        }
    }
    if (tile_max != nullptr) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int n = n0 + wn * 64 + lane;                       // n0 + wn * 64 is a multiple of 64: (n & 63) == lane
        if (n < N && m0 + wm * 128 < M)
            tile_max[((int64_t)(n >> 6) * tilesT + ((m0 + wm * 128) >> 7)) * 64 + lane] =
                *reinterpret_cast<const unsigned short*>(epi + lane * EPI_ROW + 256);
    }
}

// One workgroup per tile, or (gridDim.x < nTiles, a multiple of 8) a resident workgroup per CU
// walking its XCD's tiles: virtual block vb = blockIdx.x + i * gridDim.x keeps vb & 7 = the XCD
// the workgroup runs on, so the XCD-contiguous tile ranges below are the same in both forms.
template <bool RELU, int DIAG = 0, int OUTF = 0, int SCHED = 0>
__global__ __launch_bounds__(512, 2) void encoder_gemm256_kernel(const bf16_t* __restrict__ X,
                                                                 const bf16_t* __restrict__ W,
                                                                 const bf16_t* __restrict__ bias,
                                                                 bf16_t* __restrict__ P, int N, int D, int M,
                                                                 int tilesM, int tilesN, bf16_t* __restrict__ tile_max,
                                                                 int64_t ldp, int tilesT, int bandh,
                                                                 unsigned* dbg = nullptr, float alpha = 1.f,
                                                                 float beta = 0.f, EmitArgs ea = EmitArgs()) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (ea.run_if != nullptr && *ea.run_if == 0) return;
    const unsigned nb = (unsigned)tilesM * (unsigned)tilesN;
    for (unsigned vb = blockIdx.x; vb < nb; vb += gridDim.x) {
        // ---- tile coordinates (XCD-contiguous ranges, 8-tall bands; see encoder_gemm.hip)
        const unsigned xcd = vb & 7, slot = vb >> 3;
        const unsigned q = nb >> 3, r = nb & 7;
        const unsigned L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        const unsigned band = (unsigned)bandh * (unsigned)tilesN;
        const unsigned g = L / band, within = L % band;
        const unsigned first = g * (unsigned)bandh;
        const unsigned gsize = min((unsigned)tilesM - first, (unsigned)bandh);
        gemm256_tile<RELU, DIAG, OUTF, SCHED>(X, W, bias, P, N, D, M, (int)(first + within % gsize), (int)(within / gsize),
                                       tile_max, ldp, tilesT, dbg, smem, alpha, beta, ea);
        if (vb + gridDim.x < nb) __syncthreads();      // the next tile's staging overwrites the epilogue area
    }
}

// Diagnostic builds (make DIAG=1): SPHIP_GEMM_DIAG selects a kernel with parts of the loop
// removed (bit 0: no DMA, 1: no MFMA, 2: no ds_read, 3: no epilogue store, 4: every tile reads
// tile 0/1 of each operand, 5: per-segment cycle stamps written to the SPHIP_GEMM_DBG
// buffer; values 64 k select schedule variant k: 1 no s_setprio, 2 s_setprio 3, 3 DMA issue one
// MFMA group later, 4 two DMAs per group -- measured 12.32 / 12.37 / 12.40 / 12.94 ms against
// 12.40 ms for the shipped schedule); benchmarks/gemm_bench.py and gemm_stamps.py drive them.
// latent tiles per band of the tile walk (SPHIP_GEMM_BAND, A/B switch; 8: the 32 workgroups an
// XCD runs at a time cover 8 latent x 4 token tiles)
static int gemm_band() {
    static const int b = [] {
        const char* e = getenv("SPHIP_GEMM_BAND");
        const int v = e ? atoi(e) : 8;
        return v >= 1 && v <= 64 ? v : 8;
    }();
    return b;
}

// Host launcher, called from sphip_encoder_preacts (encoder_gemm.hip). Returns false when the
// shape is not eligible (caller falls back to the 128x128 kernel).
bool launch_gemm256_bf16(const void* x, const void* W, const void* bias, void* pre, void* tile_max, int64_t N,
                         int64_t D, int64_t M, int64_t ldp, int64_t tilesT, int relu, hipStream_t s, int* status,
                         const int* run_if) {
    if (D % G_BK != 0 || D < G_BK) return false;
    const int tilesM = (int)cdiv(M, G_TILE), tilesN = (int)cdiv(N, G_TILE);
    const int64_t nblocks = (int64_t)tilesM * tilesN;
    if (nblocks >= (1ll << 31)) return false;
    // SPHIP_GEMM_SCHED (A/B switch, read per call): where the staging DMA is issued, see SCHED above
    const char* sc = getenv("SPHIP_GEMM_SCHED");
    const int sched = sc ? (atoi(sc) == 1 ? 1 : 0) : G_SCHED_DEFAULT;
    typedef void (*kern_t)(const bf16_t*, const bf16_t*, const bf16_t*, bf16_t*, int, int, int, int, int, bf16_t*,
                           int64_t, int, int, unsigned*, float, float, EmitArgs);
    static const kern_t kerns[2][2] = {{encoder_gemm256_kernel<false, 0, 0, 0>, encoder_gemm256_kernel<false, 0, 0, 1>},
                                       {encoder_gemm256_kernel<true, 0, 0, 0>, encoder_gemm256_kernel<true, 0, 0, 1>}};
    static bool attr_set[2][2] = {{false, false}, {false, false}};
    const kern_t fn = kerns[relu ? 1 : 0][sched];
    if (!attr_set[relu ? 1 : 0][sched]) {
        if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_BYTES) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set[relu ? 1 : 0][sched] = true;
    }
    // SPHIP_GEMM_PERSIST=n: n resident workgroups per CU-count of 256 walk the tiles (A/B switch)
    const char* pers = getenv("SPHIP_GEMM_PERSIST");
    // a predicated launch is resident: when the predicate is down its 512 workgroups leave at once
    const int64_t resident = run_if ? 512 : (pers ? 256ll * atoi(pers) : 0);
    EmitArgs ea;
    ea.run_if = run_if;
    dim3 grid((unsigned)((resident > 0 && resident < nblocks) ? resident : nblocks)), block(512);
#ifdef SPHIP_DIAG
    const char* dg = getenv("SPHIP_GEMM_DIAG");
    const int diag = dg ? atoi(dg) : 0;
    const char* dbs = getenv("SPHIP_GEMM_DBG");
    unsigned* dbgp = dbs ? (unsigned*)strtoull(dbs, nullptr, 0) : nullptr;
#define SPHIP_DIAG_LAUNCH(DG)                                                                                     \
    {                                                                                                             \
        (void)hipFuncSetAttribute((const void*)encoder_gemm256_kernel<true, DG>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                            G_LDS_BYTES);                                                                         \
        hipLaunchKernelGGL((encoder_gemm256_kernel<true, DG>), grid, block, G_LDS_BYTES, s, (const bf16_t*)x,       \
                           (const bf16_t*)W, (const bf16_t*)bias, (bf16_t*)pre, (int)N, (int)D, (int)M, tilesM, tilesN, (bf16_t*)tile_max, ldp, (int)tilesT, gemm_band(), dbgp); \
    }
    if (diag == 32 && sched == 1) {
        (void)hipFuncSetAttribute((const void*)encoder_gemm256_kernel<true, 32, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_BYTES);
        hipLaunchKernelGGL((encoder_gemm256_kernel<true, 32, 0, 1>), grid, block, G_LDS_BYTES, s, (const bf16_t*)x,
                           (const bf16_t*)W, (const bf16_t*)bias, (bf16_t*)pre, (int)N, (int)D, (int)M, tilesM, tilesN, (bf16_t*)tile_max, ldp, (int)tilesT, gemm_band(), dbgp);
    } else if (diag == 1) SPHIP_DIAG_LAUNCH(1)
    else if (diag == 2) SPHIP_DIAG_LAUNCH(2)
    else if (diag == 3) SPHIP_DIAG_LAUNCH(3)
    else if (diag == 4) SPHIP_DIAG_LAUNCH(4)
    else if (diag == 5) SPHIP_DIAG_LAUNCH(5)
    else if (diag == 6) SPHIP_DIAG_LAUNCH(6)
    else if (diag == 7) SPHIP_DIAG_LAUNCH(7)
    else if (diag == 8) SPHIP_DIAG_LAUNCH(8)
    else if (diag == 13) SPHIP_DIAG_LAUNCH(13)
    else if (diag == 15) SPHIP_DIAG_LAUNCH(15)
    else if (diag == 16) SPHIP_DIAG_LAUNCH(16)
    else if (diag == 32) SPHIP_DIAG_LAUNCH(32)
    else if (diag == 33) SPHIP_DIAG_LAUNCH(33)
    else if (diag == 36) SPHIP_DIAG_LAUNCH(36)
    else if (diag == 45) SPHIP_DIAG_LAUNCH(45)
    else if (diag == 37) SPHIP_DIAG_LAUNCH(37)
    else if (diag == 22) SPHIP_DIAG_LAUNCH(22)
    else if (diag == 20) SPHIP_DIAG_LAUNCH(20)
    else if (diag == 64) SPHIP_DIAG_LAUNCH(64)
    else if (diag == 128) SPHIP_DIAG_LAUNCH(128)
    else if (diag == 192) SPHIP_DIAG_LAUNCH(192)
    else if (diag == 256) SPHIP_DIAG_LAUNCH(256)
    else if (diag == 512) SPHIP_DIAG_LAUNCH(512)
    else
#endif
    hipLaunchKernelGGL(fn, grid, block, G_LDS_BYTES, s, (const bf16_t*)x, (const bf16_t*)W, (const bf16_t*)bias,
                       (bf16_t*)pre, (int)N, (int)D, (int)M, tilesM, tilesN, (bf16_t*)tile_max, ldp, (int)tilesT,
                       gemm_band(), (unsigned*)nullptr, 1.f, 0.f, ea);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("encoder_preacts(256): launch failed: %s", hipGetErrorString(e));
        *status = SPHIP_ELAUNCH;
    } else {
        *status = SPHIP_OK;
    }
    return true;
}

// out (+)= x W^T in fp32 from bf16 operands: x (N,K), W (M,K), out (N, ld_out) float.
int launch_gemm256_f32out(const void* x, const void* W, float* out, int64_t N, int64_t K, int64_t M, int64_t ld_out,
                          int accumulate, hipStream_t s) {
    const int tilesM = (int)cdiv(M, G_TILE), tilesN = (int)cdiv(N, G_TILE);
    static bool attr_set[2] = {false, false};
    const void* fn = accumulate ? (const void*)encoder_gemm256_kernel<false, 0, 2>
                                : (const void*)encoder_gemm256_kernel<false, 0, 1>;
    if (!attr_set[accumulate ? 1 : 0]) {
        SPHIP_REQUIRE(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_BYTES) == hipSuccess,
                      SPHIP_ELAUNCH, "gemm_bf16_f32acc: cannot reserve %d bytes of LDS", G_LDS_BYTES);
        attr_set[accumulate ? 1 : 0] = true;
    }
    dim3 grid((unsigned)((int64_t)tilesM * tilesN)), block(512);
    if (accumulate)
        hipLaunchKernelGGL((encoder_gemm256_kernel<false, 0, 2>), grid, block, G_LDS_BYTES, s, (const bf16_t*)x,
                           (const bf16_t*)W, (const bf16_t*)nullptr, (bf16_t*)out, (int)N, (int)K, (int)M, tilesM, tilesN,
                           (bf16_t*)nullptr, ld_out, 0, gemm_band());
    else
        hipLaunchKernelGGL((encoder_gemm256_kernel<false, 0, 1>), grid, block, G_LDS_BYTES, s, (const bf16_t*)x,
                           (const bf16_t*)W, (const bf16_t*)nullptr, (bf16_t*)out, (int)N, (int)K, (int)M, tilesM, tilesN,
                           (bf16_t*)nullptr, ld_out, 0, gemm_band());
    SPHIP_CHECK_LAUNCH("gemm_bf16_f32acc");
    return SPHIP_OK;
}

// bf16 out = rnd(rnd(alpha * rnd(x W^T)) + rnd(beta * C)): see OUTF 3 above.
int launch_gemm256_axpy(const void* x, const void* W, const void* C, void* out, int64_t N, int64_t K, int64_t M,
                        int64_t ld_c, int64_t ld_out, float alpha, float beta, hipStream_t s) {
    const int tilesM = (int)cdiv(M, G_TILE), tilesN = (int)cdiv(N, G_TILE);
    SPHIP_REQUIRE(ld_c < (1ll << 31), SPHIP_EUNSUPPORTED, "gemm_bf16_axpy: ld_c exceeds int32");
    static bool attr_set = false;
    if (!attr_set) {
        SPHIP_REQUIRE(hipFuncSetAttribute((const void*)encoder_gemm256_kernel<false, 0, 3>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_BYTES) == hipSuccess,
                      SPHIP_ELAUNCH, "gemm_bf16_axpy: cannot reserve %d bytes of LDS", G_LDS_BYTES);
        attr_set = true;
    }
    dim3 grid((unsigned)((int64_t)tilesM * tilesN)), block(512);
    hipLaunchKernelGGL((encoder_gemm256_kernel<false, 0, 3>), grid, block, G_LDS_BYTES, s, (const bf16_t*)x,
                       (const bf16_t*)W, (const bf16_t*)C, (bf16_t*)out, (int)N, (int)K, (int)M, tilesM, tilesN,
                       (bf16_t*)nullptr, ld_out, (int)ld_c, gemm_band(), (unsigned*)nullptr, alpha, beta);
    SPHIP_CHECK_LAUNCH("gemm_bf16_axpy");
    return SPHIP_OK;
}

// relu(x W^T + b) in bf16, not stored: chunks holding a value above the token's threshold are
// appended to the token's candidate list (OUTF 4 above). W holds the M rows of a latent block
// whose first latent is 8 * id_base.
int launch_gemm256_emit(const void* x, const void* W, const void* bias, int64_t N, int64_t D, int64_t M,
                        const void* tau, unsigned* cnt, void* list, unsigned* ids, int* flag, int64_t cap,
                        int64_t id_base, hipStream_t s) {
    const int tilesM = (int)cdiv(M, G_TILE), tilesN = (int)cdiv(N, G_TILE);
    static bool attr_set = false;
    if (!attr_set) {
        SPHIP_REQUIRE(hipFuncSetAttribute((const void*)encoder_gemm256_kernel<true, 0, 4>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_BYTES) == hipSuccess,
                      SPHIP_ELAUNCH, "encoder_preacts_emit: cannot reserve %d bytes of LDS", G_LDS_BYTES);
        attr_set = true;
    }
    EmitArgs ea;
    ea.tau = (const unsigned short*)tau;
    ea.cnt = cnt;
    ea.list = (u32x4_t*)list;
    ea.ids = ids;
    ea.flag = flag;
    ea.cap = (int)cap;
    ea.id_base = (unsigned)id_base;
    {
        const char* dg = getenv("SPHIP_EMIT_DIAG");
        ea.diag = dg ? atoi(dg) : 0;
    }
    dim3 grid((unsigned)((int64_t)tilesM * tilesN)), block(512);
    hipLaunchKernelGGL((encoder_gemm256_kernel<true, 0, 4>), grid, block, G_LDS_BYTES, s, (const bf16_t*)x,
                       (const bf16_t*)W, (const bf16_t*)bias, (bf16_t*)nullptr, (int)N, (int)D, (int)M, tilesM, tilesN,
                       (bf16_t*)nullptr, (int64_t)0, 0, gemm_band(), (unsigned*)nullptr, 1.f, 0.f, ea);
    SPHIP_CHECK_LAUNCH("encoder_preacts_emit");
    return SPHIP_OK;
}

}  // namespace sphip
