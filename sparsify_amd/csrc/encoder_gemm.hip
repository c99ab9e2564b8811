// Encoder pre-activations: pre[n, m] = relu( sum_d x[n,d] * W[m,d] + b[m] )
// Reference: FusedEncoder.forward, sparsify/fused_encoder.py:29 (F.relu(F.linear(x, W, b)));
// under the reference's bf16 autocast (sparse_coder.py:205-209) all three operands are bf16
// and the result is bf16 -- that is the SPHIP_BF16 instantiation. The SPHIP_F32 one is the
// path a direct fp32 call of fused_encoder takes (tests/test_encode.py).
//
// gfx950 design (v1): 128(latent) x 128(token) output tile per 256-thread workgroup, four
// waves as 2x2, each wave 2x2 MFMA tiles of 32x32 (v_mfma_f32_32x32x16_bf16 /
// v_mfma_f32_32x32x2_f32). Both operands are K-contiguous ("TN"), so an LDS tile row is
// 128 B of K (64 bf16 / 32 f32) and every lane's MFMA fragment is one 16-B ds_read_b128.
// The 16-B chunks of a row are XOR-swizzled with (row>>1)&7 so the 16 lanes of a
// ds_read_b128 lane group hit 16 different bank slots. W is the MFMA "A" operand (latents
// on the accumulator rows): each lane then owns 4 consecutive latents of one token per
// register quad, i.e. contiguous 8-B (bf16) / 16-B (f32) output stores.
// LDS is double buffered (64 KB, 2 workgroups per CU); the next K tile is fetched into
// registers while the current one is multiplied.
// Workgroup ids are remapped so that each XCD (private 4 MiB L2) walks a contiguous range
// of 8-latent-tile x 8-token-tile patches.
#include "common.hpp"
#include <stdlib.h>

namespace sphip {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TILE = 128;        // rows of W / rows of X per workgroup
constexpr int ROW_BYTES = 128;   // K bytes per LDS row
constexpr int CHUNKS = 8;        // 16-B chunks per LDS row

template <typename T> struct GemmT;
template <> struct GemmT<bf16_t> { static constexpr int EPC = 8; };   // elements per 16-B chunk
template <> struct GemmT<float> { static constexpr int EPC = 4; };

__device__ __forceinline__ void mma_chunk(const uint4& a, const uint4& b, f32x16& c, bf16_t) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void mma_chunk(const uint4& a, const uint4& b, f32x16& c, float) {
    // lane half h holds k = 4h + i of this 8-wide k group for MFMA i; A and B agree, so
    // the four MFMAs together sum all 8 k values.
    const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], fb[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1], fb[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2], fb[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[3], fb[3], c, 0, 0, 0);
}

__device__ __forceinline__ void store4(float* p, const float (&v)[4], bool vec) {
    if (vec) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else { p[0] = v[0]; p[1] = v[1]; p[2] = v[2]; p[3] = v[3]; }
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4], bool vec) {
    const bf16_t b0 = f32_to_bf16(v[0]), b1 = f32_to_bf16(v[1]), b2 = f32_to_bf16(v[2]), b3 = f32_to_bf16(v[3]);
    if (vec) *reinterpret_cast<uint2*>(p) = make_uint2((unsigned)b0 | ((unsigned)b1 << 16), (unsigned)b2 | ((unsigned)b3 << 16));
    else { p[0] = b0; p[1] = b1; p[2] = b2; p[3] = b3; }
}

// blockIdx -> (latent tile, token tile): XCD-contiguous ranges, then 8-tall bands swept
// token-tile by token-tile (any tile counts).
__device__ __forceinline__ void tile_coords(int tilesM, int tilesN, int& tm, int& tn) {
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

// EPI 0: the encoder's epilogue (+ bias, ReLU, one rounding to T; P row stride M).
// EPI 1 (bf16 only): P = rnd(rnd(alpha * rnd(acc)) + rnd(beta * C)), rnd = round to bf16 -- the
//        roundings torch applies to `beta * C + alpha * (x @ W.mT)` on bf16 tensors (the
//        Newton-Schulz products, sparsify/muon.py:36-38); `bias` carries C (row stride ldc, may
//        be null: no second term), P has row stride ldp.
template <typename T, bool RELU, int EPI = 0>
__global__ __launch_bounds__(256, 2) void encoder_gemm_kernel(const T* __restrict__ X,
                                                              const T* __restrict__ W,
                                                              const T* __restrict__ bias,
                                                              T* __restrict__ P, int N, int D, int M,
                                                              int tilesM, int tilesN, int64_t ldp = 0,
                                                              int64_t ldc = 0, float alpha = 1.f,
                                                              float beta = 0.f) {
    constexpr int EPC = GemmT<T>::EPC;
    constexpr int BKE = CHUNKS * EPC;   // K elements per tile
    __shared__ uint4 lds[2][2][TILE * CHUNKS];   // [buffer][W|X][row*8 + swizzled chunk]

    int tm, tn;
    tile_coords(tilesM, tilesN, tm, tn);
    const int m0 = tm * TILE, n0 = tn * TILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    // staging coordinates: chunk c = tid + 256*i  ->  row = (tid>>3) + 32*i, ch = tid&7
    const int ch = tid & 7, srow = tid >> 3;
    const int kTiles = (D + BKE - 1) / BKE;

    uint4 ra[4], rb[4];
    auto fetch = [&](int kt) {
        const int kel = kt * BKE + ch * EPC;
        const bool kin = kel < D;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 32 * i;
            const int gm = m0 + row, gn = n0 + row;
            ra[i] = (kin && gm < M) ? *reinterpret_cast<const uint4*>(W + (int64_t)gm * D + kel) : make_uint4(0, 0, 0, 0);
            rb[i] = (kin && gn < N) ? *reinterpret_cast<const uint4*>(X + (int64_t)gn * D + kel) : make_uint4(0, 0, 0, 0);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 32 * i;
            const int phys = ch ^ ((row >> 1) & 7);
            lds[buf][0][row * CHUNKS + phys] = ra[i];
            lds[buf][1][row * CHUNKS + phys] = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.f;

    fetch(0);
    stash(0);
    __syncthreads();

    for (int kt = 0; kt < kTiles; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < kTiles) fetch(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row = wm * 64 + mi * 32 + r;
                a[mi] = lds[buf][0][row * CHUNKS + ((2 * ks + h) ^ ((row >> 1) & 7))];
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int row = wn * 64 + ni * 32 + r;
                b[ni] = lds[buf][1][row * CHUNKS + ((2 * ks + h) ^ ((row >> 1) & 7))];
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) mma_chunk(a[mi], b[ni], acc[mi][ni], T{});
        }
        if (kt + 1 < kTiles) stash(buf ^ 1);
        __syncthreads();
    }

    if (EPI == 1) {
        const bool vec = (ldp & 3) == 0 && (ldc & 3) == 0;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + wm * 64 + mi * 32 + 8 * q + 4 * h;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int n = n0 + wn * 64 + ni * 32 + r;
                    if (n >= N || m >= M) continue;
                    const bool whole = vec && m + 3 < M;
                    float cv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (bias != nullptr) {
                        const T* cp = bias + (int64_t)n * ldc + m;
                        if (whole) {
                            const float4 c4 = load4(cp);
                            cv[0] = c4.x, cv[1] = c4.y, cv[2] = c4.z, cv[3] = c4.w;
                        } else {
                            for (int j = 0; j < 4 && m + j < M; ++j) cv[j] = Elem<T>::load(cp, j);
                        }
                    }
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float t = bf16_to_f32(f32_to_bf16(acc[mi][ni][4 * q + j]));
                        if (alpha != 1.f) t = bf16_to_f32(f32_to_bf16(alpha * t));
                        if (bias != nullptr) t += bf16_to_f32(f32_to_bf16(beta * cv[j]));
                        v[j] = t;                       // store4 rounds once more
                    }
                    T* dst = P + (int64_t)n * ldp + m;
                    if (whole) store4(dst, v, true);
                    else
                        for (int j = 0; j < 4 && m + j < M; ++j) Elem<T>::store(dst, j, v[j]);
                }
            }
        return;
    }

    // epilogue: + bias, relu, one rounding to T, 4 consecutive latents per store
    const bool vec = (M & 3) == 0;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = m0 + wm * 64 + mi * 32 + 8 * q + 4 * h;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias != nullptr) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (m + j < M) bv[j] = Elem<T>::load(bias, m + j);
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int n = n0 + wn * 64 + ni * 32 + r;
                if (n >= N || m >= M) continue;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = acc[mi][ni][4 * q + j] + bv[j];
                    v[j] = RELU ? (t > 0.f ? t : 0.f) : t;
                }
                T* dst = P + (int64_t)n * M + m;
                if (m + 3 < M) store4(dst, v, vec);
                else
                    for (int j = 0; j < 4 && m + j < M; ++j) Elem<T>::store(dst, j, v[j]);
            }
        }
    }
}

template <typename T>
static int launch_gemm(const void* x, const void* W, const void* bias, void* pre, int64_t N, int64_t D,
                       int64_t M, int relu, hipStream_t s) {
    const int tilesM = (int)cdiv(M, TILE), tilesN = (int)cdiv(N, TILE);
    const int64_t nblocks = (int64_t)tilesM * tilesN;
    SPHIP_REQUIRE(nblocks < (1ll << 31), SPHIP_EUNSUPPORTED, "encoder_preacts: grid too large");
    dim3 grid((unsigned)nblocks), block(256);
    if (relu)
        hipLaunchKernelGGL((encoder_gemm_kernel<T, true>), grid, block, 0, s, (const T*)x, (const T*)W,
                           (const T*)bias, (T*)pre, (int)N, (int)D, (int)M, tilesM, tilesN);
    else
        hipLaunchKernelGGL((encoder_gemm_kernel<T, false>), grid, block, 0, s, (const T*)x, (const T*)W,
                           (const T*)bias, (T*)pre, (int)N, (int)D, (int)M, tilesM, tilesN);
    SPHIP_CHECK_LAUNCH("encoder_preacts");
    return SPHIP_OK;
}

bool launch_gemm256_bf16(const void* x, const void* W, const void* bias, void* pre, void* tile_max, int64_t N,
                         int64_t D, int64_t M, int64_t ldp, int64_t tilesT, int relu, hipStream_t s,
                         int* status, const int* run_if = nullptr);   // encoder_gemm256.hip

int launch_gemm256_axpy(const void* x, const void* W, const void* C, void* out, int64_t N, int64_t K, int64_t M,
                        int64_t ld_c, int64_t ld_out, float alpha, float beta, hipStream_t s);   // encoder_gemm256.hip

// Block maxima of pre for the kernels that do not take them in their epilogue (small shapes,
// fp32): one wave per (token, 128-latent block). NaNs rank above +inf, as in topk.hip.
template <typename T>
__global__ __launch_bounds__(256) void tile_max_kernel(const T* __restrict__ pre, T* __restrict__ tile_max,
                                                       int64_t total, int M, int tilesT) {
    const int lane = threadIdx.x & 63;
    const int64_t gid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gid >= total) return;
    const int64_t n = gid / tilesT;
    const int t = (int)(gid % tilesT);
    const T* p = pre + n * M + (int64_t)t * 128;
    unsigned kmax = 0;
    for (int i = lane; i < 128 && t * 128 + i < M; i += 64) kmax = max(kmax, to_key(p[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, o, 64));
    if (lane == 0) tile_max[((n >> 6) * tilesT + t) * 64 + (n & 63)] = KeyOf<T>::value(kmax);
}

template <typename T>
static int launch_tile_max(const void* pre, void* tile_max, int64_t N, int64_t M, hipStream_t s) {
    const int tilesT = (int)cdiv(M, 128);
    const int64_t total = N * tilesT;
    SPHIP_REQUIRE(cdiv(total, 4) < (1ll << 31), SPHIP_EUNSUPPORTED, "encoder_preacts: tile_max grid too large");
    hipLaunchKernelGGL((tile_max_kernel<T>), dim3((unsigned)cdiv(total, 4)), dim3(256), 0, s, (const T*)pre,
                       (T*)tile_max, total, (int)M, tilesT);
    SPHIP_CHECK_LAUNCH("encoder_preacts(tile_max)");
    return SPHIP_OK;
}

}  // namespace sphip

extern "C" int sphip_encoder_preacts_uses_tile256(int dtype, int64_t N, int64_t D, int64_t M) {
    if (dtype != SPHIP_BF16 || D % 64 != 0 || D < 64) return 0;
    // large shapes: 256x256 LDS-DMA kernel (needs >= one workgroup per CU to pay off);
    // SPHIP_GEMM=128|256 forces a kernel (tests).
    const char* force = getenv("SPHIP_GEMM");
    return (force ? (force[0] == '2') : (sphip::cdiv(M, 256) * sphip::cdiv(N, 256) >= 256)) ? 1 : 0;
}

extern "C" int sphip_encoder_preacts(const void* x, const void* W, const void* bias, void* pre, void* tile_max,
                                     int dtype, int64_t N, int64_t D, int64_t M, int relu, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(valid_dtype(dtype), SPHIP_EINVAL, "encoder_preacts: bad dtype %d", dtype);
    SPHIP_REQUIRE(x && W && pre, SPHIP_EINVAL, "encoder_preacts: null pointer");
    SPHIP_REQUIRE(N > 0 && D > 0 && M > 0, SPHIP_EINVAL, "encoder_preacts: empty shape N=%lld D=%lld M=%lld",
                  (long long)N, (long long)D, (long long)M);
    SPHIP_REQUIRE(N < (1ll << 31) && M < (1ll << 31) && D < (1ll << 31), SPHIP_EUNSUPPORTED,
                  "encoder_preacts: dimension exceeds int32");
    const int epc = dtype == SPHIP_BF16 ? 8 : 4;
    SPHIP_REQUIRE(D % epc == 0, SPHIP_EUNSUPPORTED, "encoder_preacts: D=%lld must be a multiple of %d",
                  (long long)D, epc);
    SPHIP_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)pre & 15) == 0,
                  SPHIP_EINVAL, "encoder_preacts: operands must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SPHIP_BF16) {
        // large shapes: 256x256 LDS-DMA kernel (needs >= one workgroup per CU to pay off);
        // SPHIP_GEMM=128|256 forces a kernel (tests).
        const bool want256 = sphip_encoder_preacts_uses_tile256(dtype, N, D, M) != 0;
        int status = SPHIP_OK;
        if (want256 && launch_gemm256_bf16(x, W, bias, pre, tile_max, N, D, M, M, cdiv(M, 128), relu, s, &status)) return status;
        status = launch_gemm<bf16_t>(x, W, bias, pre, N, D, M, relu, s);
        if (status != SPHIP_OK || tile_max == nullptr) return status;
        return launch_tile_max<bf16_t>(pre, tile_max, N, M, s);
    }
    const int status = launch_gemm<float>(x, W, bias, pre, N, D, M, relu, s);
    if (status != SPHIP_OK || tile_max == nullptr) return status;
    return launch_tile_max<float>(pre, tile_max, N, M, s);
}

extern "C" int sphip_encoder_preacts_block(const void* x, const void* W, const void* bias, void* pre, void* tile_max,
                                           int dtype, int64_t N, int64_t D, int64_t M_block, int64_t ld_pre,
                                           int64_t tiles_total, int relu, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(dtype == SPHIP_BF16, SPHIP_EUNSUPPORTED, "encoder_preacts_block: bf16 only");
    SPHIP_REQUIRE(x && W && pre, SPHIP_EINVAL, "encoder_preacts_block: null pointer");
    SPHIP_REQUIRE(N > 0 && D > 0 && M_block > 0, SPHIP_EINVAL, "encoder_preacts_block: empty shape");
    SPHIP_REQUIRE(N < (1ll << 31) && ld_pre < (1ll << 31) && D < (1ll << 31), SPHIP_EUNSUPPORTED,
                  "encoder_preacts_block: dimension exceeds int32");
    SPHIP_REQUIRE(M_block % 128 == 0 && ld_pre >= M_block && ld_pre % 8 == 0, SPHIP_EINVAL,
                  "encoder_preacts_block: the block must be whole 128-latent tiles of a row of ld_pre >= M_block, "
                  "ld_pre a multiple of 8 (M_block=%lld ld_pre=%lld)", (long long)M_block, (long long)ld_pre);
    SPHIP_REQUIRE(tile_max == nullptr || tiles_total >= M_block / 128, SPHIP_EINVAL,
                  "encoder_preacts_block: tiles_total=%lld is shorter than the block", (long long)tiles_total);
    SPHIP_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)pre & 15) == 0, SPHIP_EINVAL,
                  "encoder_preacts_block: operands must be 16-byte aligned");
    int status = SPHIP_OK;
    if (launch_gemm256_bf16(x, W, bias, pre, tile_max, N, D, M_block, ld_pre, tiles_total, relu, (hipStream_t)stream,
                            &status))
        return status;
    set_error("encoder_preacts_block: D=%lld is not a multiple of 64", (long long)D);
    return SPHIP_EUNSUPPORTED;
}

extern "C" int sphip_encoder_preacts_if(const int* run_if, const void* x, const void* W, const void* bias, void* pre,
                                        void* tile_max, int64_t N, int64_t D, int64_t M, int relu, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(run_if && x && W && pre, SPHIP_EINVAL, "encoder_preacts_if: null pointer");
    SPHIP_REQUIRE(N > 0 && D > 0 && M > 0 && N < (1ll << 31) && M < (1ll << 31) && D < (1ll << 31), SPHIP_EINVAL,
                  "encoder_preacts_if: bad shape N=%lld D=%lld M=%lld", (long long)N, (long long)D, (long long)M);
    SPHIP_REQUIRE(M % 8 == 0, SPHIP_EUNSUPPORTED, "encoder_preacts_if: M=%lld must be a multiple of 8", (long long)M);
    SPHIP_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)pre & 15) == 0 &&
                      ((uintptr_t)run_if & 3) == 0, SPHIP_EINVAL, "encoder_preacts_if: operands must be 16-byte aligned");
    int status = SPHIP_OK;
    if (launch_gemm256_bf16(x, W, bias, pre, tile_max, N, D, M, M, cdiv(M, 128), relu, (hipStream_t)stream, &status,
                            run_if))
        return status;
    set_error("encoder_preacts_if: D=%lld is not a multiple of 64", (long long)D);
    return SPHIP_EUNSUPPORTED;
}

extern "C" int sphip_gemm_bf16_axpy(const void* x, const void* W, const void* C, void* out, int64_t N, int64_t K,
                                    int64_t M, int64_t ld_c, int64_t ld_out, float alpha, float beta, void* stream) {
    using namespace sphip;
    SPHIP_REQUIRE(x && W && out, SPHIP_EINVAL, "gemm_bf16_axpy: null pointer");
    SPHIP_REQUIRE(N > 0 && K > 0 && M > 0 && N < (1ll << 31) && M < (1ll << 31) && K < (1ll << 31), SPHIP_EINVAL,
                  "gemm_bf16_axpy: bad sizes N=%lld K=%lld M=%lld", (long long)N, (long long)K, (long long)M);
    SPHIP_REQUIRE(ld_out >= M && (C == nullptr || ld_c >= M), SPHIP_EINVAL,
                  "gemm_bf16_axpy: row strides shorter than M (ld_c=%lld ld_out=%lld M=%lld)", (long long)ld_c,
                  (long long)ld_out, (long long)M);
    SPHIP_REQUIRE(K % 8 == 0, SPHIP_EUNSUPPORTED, "gemm_bf16_axpy: K=%lld must be a multiple of 8", (long long)K);
    SPHIP_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)out & 7) == 0 &&
                      ((uintptr_t)C & 7) == 0, SPHIP_EINVAL, "gemm_bf16_axpy: operands must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    // large products: the 256x256 LDS-DMA kernel (same rule as the encoder; SPHIP_GEMM=128|256 forces one)
    const char* force = getenv("SPHIP_GEMM");
    if (K % 64 == 0 && M % 4 == 0 && ld_out % 4 == 0 && (C == nullptr || ld_c % 4 == 0) && K * 2 * 256 < (1ll << 31) &&
        (force ? force[0] == '2' : cdiv(M, 256) * cdiv(N, 256) >= 256))
        return launch_gemm256_axpy(x, W, C, out, N, K, M, ld_c, ld_out, alpha, beta, s);
    const int tilesM = (int)cdiv(M, TILE), tilesN = (int)cdiv(N, TILE);
    SPHIP_REQUIRE((int64_t)tilesM * tilesN < (1ll << 31), SPHIP_EUNSUPPORTED, "gemm_bf16_axpy: grid too large");
    hipLaunchKernelGGL((encoder_gemm_kernel<bf16_t, false, 1>), dim3((unsigned)(tilesM * tilesN)), dim3(256), 0, s,
                       (const bf16_t*)x, (const bf16_t*)W, (const bf16_t*)C, (bf16_t*)out, (int)N, (int)K, (int)M, tilesM,
                       tilesN, ld_out, ld_c, alpha, beta);
    SPHIP_CHECK_LAUNCH("gemm_bf16_axpy");
    return SPHIP_OK;
}
