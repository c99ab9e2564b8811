// Encoder pre-activations, large-shape bf16 kernel: 256(latent) x 256(token) x 64(K) tile,
// 512 threads = 8 waves (2 latent groups x 4 token groups, 128 x 64 outputs per wave),
// v_mfma_f32_16x16x32_bf16, operands staged by LDS-DMA (global_load_lds_dwordx4) into a
// 128 KiB double-buffered image, counted vmcnt waits and raw s_barriers so two tiles of
// staging stay in flight across barriers, and the two wave groups that share a SIMD run
// staggered by one barrier (one group's MFMA segment overlaps the other's LDS-read /
// DMA-issue segment).
// Reference semantics: FusedEncoder.forward, sparsify/fused_encoder.py:29, under bf16
// autocast (sparse_coder.py:205-209); same contract as encoder_gemm.hip.
//
// LDS image. A K tile of one operand (256 rows x 128 B of K) is kept as two "half tiles" of
// 128 rows, split NOT by wave group but by when they are consumed:
//     W-lo : rows  [g*128 +  0, g*128 +  64)  of both latent groups g   -> read in phase 0
//     W-hi : rows  [g*128 + 64, g*128 + 128)                            -> read in phase 2
//     X-lo : rows  [n*64  +  0, n*64  +  32)  of the four token groups n -> read in phase 0
//     X-hi : rows  [n*64  + 32, n*64  +  64)                            -> read in phase 1
// so every half tile is read in exactly one phase of the K tile. A half tile is a linear
// [128 rows][8 chunks of 16 B] array whose chunk index is XOR-swizzled with (row>>1)&7
// (conflict-free ds_read_b128 for the 16x16x32 fragment shape). LDS-DMA writes linearly
// (wave base + lane*16), so the swizzle is applied to the per-lane SOURCE address.
//
// Schedule per K tile t (4 phases, each = LOAD segment | barrier | MFMA segment | barrier):
//     phase 0: ds_read W-lo(8) + X-lo(4) | DMA X-lo(t+1) | 16 MFMA (w-lo x x-lo)
//     phase 1: ds_read X-hi(4)           | DMA W-lo(t+1) | 16 MFMA (w-lo x x-hi)
//     phase 2: ds_read W-hi(8)           | DMA X-hi(t+1) | 16 MFMA (w-hi x x-hi)
//     phase 3: -                         | DMA W-hi(t+1) | 16 MFMA (w-hi x x-lo)
// RAW: each wave retires the DMA of the half tile read in the NEXT phase with a counted
// s_waitcnt vmcnt(4) at the end of its LOAD segment, i.e. before a barrier that every wave
// passes before the earliest reader (the un-delayed wave group) starts that phase.
// WAR: a half tile is overwritten 4 phases after the phase that read it.
#include "common.hpp"

namespace sphip {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int G_TILE = 256;
constexpr int G_BK = 64;                    // bf16 elements per K tile (128 B)
constexpr int HALF_BYTES = 128 * 128;       // 16 KiB
constexpr int BUF_BYTES = 4 * HALF_BYTES;   // one K tile of both operands: 64 KiB
constexpr int G_LDS_BYTES = 2 * BUF_BYTES;  // 128 KiB
// half-tile slots inside a buffer
constexpr int H_XLO = 0, H_WLO = 1, H_XHI = 2, H_WHI = 3;

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

__device__ __forceinline__ void dma16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

#define SPHIP_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__device__ __forceinline__ void seg_barrier() {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <bool RELU>
__global__ __launch_bounds__(512, 2) void encoder_gemm256_kernel(const bf16_t* __restrict__ X,
                                                                 const bf16_t* __restrict__ W,
                                                                 const bf16_t* __restrict__ bias,
                                                                 bf16_t* __restrict__ P, int N, int D, int M,
                                                                 int tilesM, int tilesN) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- tile coordinates (XCD-contiguous ranges, 8-tall bands; see encoder_gemm.hip)
    int tm, tn;
    {
        const unsigned nb = gridDim.x, bid = blockIdx.x;
        const unsigned xcd = bid & 7, slot = bid >> 3;
        const unsigned q = nb >> 3, r = nb & 7;
        const unsigned L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        const unsigned band = 8u * (unsigned)tilesN;
        const unsigned g = L / band, within = L % band;
        const unsigned first = g * 8;
        const unsigned gsize = min((unsigned)tilesM - first, 8u);
        tm = (int)(first + within % gsize);
        tn = (int)(within / gsize);
    }
    const int m0 = tm * G_TILE, n0 = tn * G_TILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int r16 = lane & 15, q4 = lane >> 4;
    const int kTiles = D / G_BK;

    // ---- staging: thread owns chunks c = tid + 512*i (i = 0,1) of every half tile
    const bf16_t* src[4][2];
    int dst_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + 512 * i;
        const int hr = c >> 3;                                   // half-tile row 0..127
        const int logical = (c & 7) ^ ((hr >> 1) & 7);           // source chunk for this LDS slot
        dst_off[i] = (512 * i + 64 * wave) * 16;                 // wave-uniform LDS base of the 1 KiB piece
        const int wrow_lo = (hr >> 6) * 128 + (hr & 63);         // W tile row of W-lo[hr]
        const int xrow_lo = (hr >> 5) * 64 + (hr & 31);          // X tile row of X-lo[hr]
        const int gm_lo = min(m0 + wrow_lo, M - 1), gm_hi = min(m0 + wrow_lo + 64, M - 1);
        const int gn_lo = min(n0 + xrow_lo, N - 1), gn_hi = min(n0 + xrow_lo + 32, N - 1);
        src[H_XLO][i] = X + (int64_t)gn_lo * D + logical * 8;
        src[H_XHI][i] = X + (int64_t)gn_hi * D + logical * 8;
        src[H_WLO][i] = W + (int64_t)gm_lo * D + logical * 8;
        src[H_WHI][i] = W + (int64_t)gm_hi * D + logical * 8;
    }
    auto stage = [&](int half, int t) {
        char* base = smem + (t & 1) * BUF_BYTES + half * HALF_BYTES;
        dma16(src[half][0] + (int64_t)t * G_BK, base + dst_off[0]);
        dma16(src[half][1] + (int64_t)t * G_BK, base + dst_off[1]);
    };

    // ---- fragment reads. They are issued from inline asm: hipcc's waitcnt pass would put an
    // s_waitcnt vmcnt(0) in front of every compiler-visible ds_read that follows an LDS-DMA
    // (it cannot tell which DMA a read depends on) and drain the staging pipeline each
    // phase. The reads' completion is awaited by the explicit lgkmcnt(0) below, which names
    // every destination register so no consumer can be scheduled above it.
    // Per-lane byte address inside a half tile, k-step ks: chunk = 4*ks + q4. The row part
    // of the XOR ((hr>>1)&7 = (r16>>1)&7) does not depend on the 16-row tile index, so the
    // tiles of a sub-block are reached with immediate offsets (tile stride 16*128 B).
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const int swz = (r16 >> 1) & 7;
    unsigned wbase[2], xbase[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        wbase[ks] = lds0 + (wm * 64 + r16) * 128 + (((4 * ks + q4) ^ swz) << 4);
        xbase[ks] = lds0 + (wn * 32 + r16) * 128 + (((4 * ks + q4) ^ swz) << 4);
    }

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    u32x4_t wf[4][2], xlo[2][2], xhi[2][2];

#define SPHIP_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define SPHIP_READ_W(HALF)                                              \
    SPHIP_DSR(wf[0][0], wa0, (HALF) * HALF_BYTES + 0 * 2048);           \
    SPHIP_DSR(wf[0][1], wa1, (HALF) * HALF_BYTES + 0 * 2048);           \
    SPHIP_DSR(wf[1][0], wa0, (HALF) * HALF_BYTES + 1 * 2048);           \
    SPHIP_DSR(wf[1][1], wa1, (HALF) * HALF_BYTES + 1 * 2048);           \
    SPHIP_DSR(wf[2][0], wa0, (HALF) * HALF_BYTES + 2 * 2048);           \
    SPHIP_DSR(wf[2][1], wa1, (HALF) * HALF_BYTES + 2 * 2048);           \
    SPHIP_DSR(wf[3][0], wa0, (HALF) * HALF_BYTES + 3 * 2048);           \
    SPHIP_DSR(wf[3][1], wa1, (HALF) * HALF_BYTES + 3 * 2048)
#define SPHIP_READ_X(XF, HALF)                                          \
    SPHIP_DSR(XF[0][0], xa0, (HALF) * HALF_BYTES + 0 * 2048);           \
    SPHIP_DSR(XF[0][1], xa1, (HALF) * HALF_BYTES + 0 * 2048);           \
    SPHIP_DSR(XF[1][0], xa0, (HALF) * HALF_BYTES + 1 * 2048);           \
    SPHIP_DSR(XF[1][1], xa1, (HALF) * HALF_BYTES + 1 * 2048)
#define SPHIP_LGKM0_W()                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                   \
                 : "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(wf[1][0]), "+v"(wf[1][1]), "+v"(wf[2][0]),         \
                   "+v"(wf[2][1]), "+v"(wf[3][0]), "+v"(wf[3][1]))
#define SPHIP_LGKM0_X(XF) \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(XF[0][0]), "+v"(XF[0][1]), "+v"(XF[1][0]), "+v"(XF[1][1]))

    auto mma = [&](int mbase, int nbase, const u32x4_t (&xf)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mbase + mt][nbase + nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8_t, wf[mt][ks]), __builtin_bit_cast(bf16x8_t, xf[nt][ks]),
                        acc[mbase + mt][nbase + nt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: K tile 0 completely, first two half tiles retired
    stage(H_XLO, 0);
    stage(H_WLO, 0);
    stage(H_XHI, 0);
    stage(H_WHI, 0);
    SPHIP_VMCNT(4);
    seg_barrier();
    if (wm == 1) seg_barrier();          // stagger the second wave group by one barrier

    // One K tile. NEXT: stage tile t+1 while computing tile t.
    // Outstanding DMA (per wave, in issue order) at each wait is listed on the right.
#define SPHIP_KTILE(NEXT, WAIT_A, WAIT_B, WAIT_C)                                                        \
    {                                                                                                   \
        const unsigned bo = (unsigned)(t & 1) * BUF_BYTES;                                              \
        const unsigned wa0 = wbase[0] + bo, wa1 = wbase[1] + bo, xa0 = xbase[0] + bo, xa1 = xbase[1] + bo; \
        /* phase 0 */                                                                                   \
        SPHIP_READ_X(xlo, H_XLO);                                                                       \
        SPHIP_READ_W(H_WLO);                                                                            \
        if (NEXT) stage(H_XLO, t + 1);                                                                  \
        WAIT_A; /* retire X-hi(t)              [X-hi(t) W-hi(t) X-lo(t+1)] */                           \
        seg_barrier();                                                                                  \
        SPHIP_LGKM0_X(xlo);                                                                             \
        SPHIP_LGKM0_W();                                                                                \
        mma(0, 0, xlo);                                                                                 \
        seg_barrier();                                                                                  \
        /* phase 1 */                                                                                   \
        SPHIP_READ_X(xhi, H_XHI);                                                                       \
        if (NEXT) stage(H_WLO, t + 1);                                                                  \
        WAIT_B; /* retire W-hi(t)              [W-hi(t) X-lo(t+1) W-lo(t+1)] */                         \
        seg_barrier();                                                                                  \
        SPHIP_LGKM0_X(xhi);                                                                             \
        mma(0, 2, xhi);                                                                                 \
        seg_barrier();                                                                                  \
        /* phase 2 */                                                                                   \
        SPHIP_READ_W(H_WHI);                                                                            \
        if (NEXT) stage(H_XHI, t + 1);                                                                  \
        seg_barrier();                                                                                  \
        SPHIP_LGKM0_W();                                                                                \
        mma(4, 2, xhi);                                                                                 \
        seg_barrier();                                                                                  \
        /* phase 3 */                                                                                   \
        if (NEXT) stage(H_WHI, t + 1);                                                                  \
        WAIT_C; /* retire X-lo(t+1), W-lo(t+1) [X-lo W-lo X-hi W-hi (t+1)] */                           \
        seg_barrier();                                                                                  \
        mma(4, 0, xlo);                                                                                 \
        seg_barrier();                                                                                  \
    }

    int t = 0;
    for (; t < kTiles - 1; ++t) SPHIP_KTILE(true, SPHIP_VMCNT(4), SPHIP_VMCNT(4), SPHIP_VMCNT(4))
    // last K tile: nothing left to stage; outstanding = [X-hi(t) W-hi(t)]
    SPHIP_KTILE(false, SPHIP_VMCNT(2), SPHIP_VMCNT(0), (void)0)
#undef SPHIP_KTILE

    if (wm == 0) seg_barrier();          // matches the stagger barrier of the other group

    // ---- epilogue: + bias, relu, one rounding to bf16; lane owns 4 consecutive latents of
    //      one token per accumulator tile (C/D map of 16x16: col = lane&15, row = 4*(lane>>4)+reg)
    const bool vec = (M & 3) == 0;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int m = m0 + wm * 128 + (mi >> 2) * 64 + (mi & 3) * 16 + q4 * 4;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (m + j < M) bv[j] = bf16_to_f32(bias[m + j]);
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + (ni >> 1) * 32 + (ni & 1) * 16 + r16;
            if (n >= N || m >= M) continue;
            bf16_t o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = acc[mi][ni][j] + bv[j];
                o[j] = f32_to_bf16(RELU ? (v > 0.f ? v : 0.f) : v);
            }
            bf16_t* dst = P + (int64_t)n * M + m;
            if (vec && m + 3 < M) {
                *reinterpret_cast<uint2*>(dst) = make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16),
                                                            (unsigned)o[2] | ((unsigned)o[3] << 16));
            } else {
                for (int j = 0; j < 4 && m + j < M; ++j) dst[j] = o[j];
            }
        }
    }
}

// Host launcher, called from sphip_encoder_preacts (encoder_gemm.hip). Returns false when the
// shape is not eligible (caller falls back to the 128x128 kernel).
bool launch_gemm256_bf16(const void* x, const void* W, const void* bias, void* pre, int64_t N, int64_t D,
                         int64_t M, int relu, hipStream_t s, int* status) {
    if (D % G_BK != 0 || D < G_BK) return false;
    const int tilesM = (int)cdiv(M, G_TILE), tilesN = (int)cdiv(N, G_TILE);
    const int64_t nblocks = (int64_t)tilesM * tilesN;
    if (nblocks >= (1ll << 31)) return false;
    static bool attr_set[2] = {false, false};
    const void* fn = relu ? (const void*)encoder_gemm256_kernel<true> : (const void*)encoder_gemm256_kernel<false>;
    if (!attr_set[relu ? 1 : 0]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_BYTES) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        attr_set[relu ? 1 : 0] = true;
    }
    dim3 grid((unsigned)nblocks), block(512);
    if (relu)
        hipLaunchKernelGGL((encoder_gemm256_kernel<true>), grid, block, G_LDS_BYTES, s, (const bf16_t*)x, (const bf16_t*)W,
                           (const bf16_t*)bias, (bf16_t*)pre, (int)N, (int)D, (int)M, tilesM, tilesN);
    else
        hipLaunchKernelGGL((encoder_gemm256_kernel<false>), grid, block, G_LDS_BYTES, s, (const bf16_t*)x, (const bf16_t*)W,
                           (const bf16_t*)bias, (bf16_t*)pre, (int)N, (int)D, (int)M, tilesM, tilesN);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("encoder_preacts(256): launch failed: %s", hipGetErrorString(e));
        *status = SPHIP_ELAUNCH;
    } else {
        *status = SPHIP_OK;
    }
    return true;
}

}  // namespace sphip
