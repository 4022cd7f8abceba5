// conv_x3.hip -- 3x3 convolution of the fp32-equivalent ("bf16x3") VAE decode: implicit GEMM over split-bf16 operands.
//
// Replaces F.conv2d of AutoencoderKL's decoder run in fp32 (reference: vae.to(torch.float32),
// scripts/train_sd3_fast_pickscore.py:481; call site sd3_pipeline_with_logprob_fast.py:667-670).  gfx950 has no fp32-rate
// matrix path (f32 MFMA = 1/16 of the bf16 rate), so every f32 operand travels as hi = bf16(v), lo = bf16(v - hi) and
// a product is formed as  x_hi w_hi + x_hi w_lo + x_lo w_hi  on the bf16 MFMA with f32 accumulation (the dropped lo*lo term is
// 2^-16 relative).
//
// Round 2 ran this through the plain convolution kernel by TRIPLING the contraction axis ([hi | hi | lo] x [hi | lo | hi]):
// three k-tiles, three stage loads and three barriers per 64 channels, each with the bf16 kernel's MFMA : byte ratio.  Here one
// k-tile stages the four 64-wide pieces (x_hi, x_lo, w_hi, w_lo) ONCE and issues the three MFMA products from them:
// 2/3 of the L2 -> LDS bytes and 1/3 of the barriers for the same MFMA work, i.e. 1.5x the arithmetic intensity of the bf16
// convolution -- which is what the two-stage structure needs to keep the matrix pipe busy.
//   * operands are read where the round-2 layout has them (activations [.., 3C] = [hi | hi | lo], weights per tap
//     [hi | lo | hi]): hi at +0, x_lo at +2C, w_lo at +C; nothing else in the decoder changes
//   * workgroup 512 threads = 8 waves (4 x 2), tile 192 pixels x 128 channels x 64, wave tile 48 x 64,
//     v_mfma_f32_16x16x32_bf16 with swapped operands (a lane owns 4 consecutive output channels of one pixel: float4 stores
//     in the f32 epilogue)
//   * LDS: 2 stages x (2 x 24 KiB pixel pieces + 2 x 16 KiB weight pieces) = 160 KiB, ALL of a CU's LDS (one workgroup per
//     CU, two waves per SIMD); HBM/L2 -> LDS by global_load_lds_dwordx4, lane-linear 1 KiB images with the
//     chunk ^ (row & 7) source swizzle of gemm.hip
//   * 72 MFMAs per wave and k-tile behind 28 ds_read_b128 and 10 DMA instructions.  The kernel is bound by the LDS port
//     (fragment reads + DMA writes), not by L2 or the matrix pipe: the 128-pixel tile of round 2 read 0.50 KiB of
//     fragments per MFMA and wrote 64 KiB per 48 MFMA-slots, this one 0.39 KiB and 80 KiB per 72 (measured: 8 x 512^2
//     decode 73.6 -> 69.9 ms, rollout step 422.9 -> 418.3 ms on the same box)
#include "gemm_device.hpp"

namespace advgrpo {

namespace {

typedef __attribute__((address_space(3))) void* x3_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* x3_gptr_t;

constexpr int X3_BN = 128, X3_WM = 4, X3_WN = 2, X3_BK = 64;
constexpr int X3_PIECE_W = X3_BN * X3_BK * 2;        // 16 KiB: one 128-row x 64-channel bf16 weight piece
constexpr int x3_lds_bytes(int bm) { return 2 * (2 * bm * X3_BK * 2 + 2 * X3_PIECE_W); }   // 2 stages x (x_hi, x_lo, w_hi, w_lo)

template <int X3_BM>
__global__ __launch_bounds__(512, 2) void conv3x3_x3_kernel(const GemmParams p) {
    constexpr int NW = X3_WM * X3_WN, TM = X3_BM / X3_WM, TN = X3_BN / X3_WN, FM = TM / 16, FN = TN / 16;
    constexpr int INST = X3_BM / 8 / NW;             // DMA instructions per wave and pixel piece (8 rows each): 2 or 3
    constexpr int INST_W = X3_BN / 8 / NW;           // ... and weight piece: 2
    constexpr int X3_PIECE = X3_BM * X3_BK * 2;      // one BM-row x 64-channel bf16 pixel piece
    constexpr int X3_STAGE = 2 * X3_PIECE + 2 * X3_PIECE_W;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / X3_WN, wn = wave % X3_WN;
    const int C = p.Cin / 3;                          // logical channels; pixel pitch of the activations = 3C elements

    const int tiles_n = (p.N + X3_BN - 1) / X3_BN, tiles_m = (p.M + X3_BM - 1) / X3_BM;
    const int swz = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    int tile_m, tile_n;
    tile_coords(swz, tiles_m, tiles_n, 4, tile_m, tile_n);
    const int m0 = tile_m * X3_BM, n0 = tile_n * X3_BN;

    // ---- per-lane DMA sources: output pixel of this lane's A rows, weight row pointers
    const int lrow = lane >> 3, schunk = (lane & 7) ^ lrow;
    int a_y[INST], a_x[INST];
    int64_t a_img[INST];
    const bf16_t* w_src[INST_W];
#pragma unroll
    for (int it = 0; it < INST; ++it) {
        int r = m0 + (wave + it * NW) * 8 + lrow;
        r = r < p.M ? r : p.M - 1;
        const int hw = p.Hout * p.Wout;
        const int bi = r / hw, rem = r - bi * hw;
        a_y[it] = rem / p.Wout;
        a_x[it] = rem - a_y[it] * p.Wout;
        a_img[it] = (int64_t)bi * (p.Hout >> p.ups) * (p.Wout >> p.ups) * p.Cin;
    }
#pragma unroll
    for (int it = 0; it < INST_W; ++it) {
        int n = n0 + (wave + it * NW) * 8 + lrow;
        n = n < p.N ? n : p.N - 1;
        w_src[it] = p.W + (int64_t)n * p.ldw + schunk * 8;
    }
    const int win = p.Wout >> p.ups;
    const int kt_per_tap = C / X3_BK;
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * X3_STAGE;
        const int tap = kt / kt_per_tap, c0 = (kt - tap * kt_per_tap) * X3_BK;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
        for (int it = 0; it < INST; ++it) {
            const int yy = a_y[it] + dy, xx = a_x[it] + dx;
            const bool ok = (unsigned)yy < (unsigned)p.Hout && (unsigned)xx < (unsigned)p.Wout;
            const bf16_t* hi = ok ? p.A + a_img[it] + ((int64_t)(yy >> p.ups) * win + (xx >> p.ups)) * p.Cin + c0 + schunk * 8
                                  : p.zero_page + schunk * 8;
            const bf16_t* lo = ok ? hi + 2 * C : hi;                          // [hi | hi | lo]
            char* dst = base + (wave + it * NW) * 1024;
            __builtin_amdgcn_global_load_lds((x3_gptr_t)hi, (x3_lds_ptr_t)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((x3_gptr_t)lo, (x3_lds_ptr_t)(dst + X3_PIECE), 16, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < INST_W; ++it) {
            char* dst = base + 2 * X3_PIECE + (wave + it * NW) * 1024;
            const bf16_t* wh = w_src[it] + (int64_t)tap * p.Cin + c0;         // per tap [hi | lo | hi]
            __builtin_amdgcn_global_load_lds((x3_gptr_t)wh, (x3_lds_ptr_t)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((x3_gptr_t)(wh + C), (x3_lds_ptr_t)(dst + X3_PIECE_W), 16, 0, 0);
        }
    };

    int frag_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) frag_off[ks] = (lane & 15) * 128 + (((ks * 4 + (lane >> 4)) ^ (lane & 7)) << 4);

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = 9 * kt_per_tap;
    stage(0, 0);
    __syncthreads();          // (hipcc drains the LDS-DMA before the barrier)
    auto load_frags = [&](const char* ta, const char* tb, int ks, bf16x8_t (&ah)[FM], bf16x8_t (&al)[FM], bf16x8_t (&bh)[FN],
                          bf16x8_t (&bl)[FN]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            ah[i] = *reinterpret_cast<const bf16x8_t*>(ta + i * 2048 + frag_off[ks]);
            al[i] = *reinterpret_cast<const bf16x8_t*>(ta + X3_PIECE + i * 2048 + frag_off[ks]);
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            bh[j] = *reinterpret_cast<const bf16x8_t*>(tb + j * 2048 + frag_off[ks]);
            bl[j] = *reinterpret_cast<const bf16x8_t*>(tb + X3_PIECE_W + j * 2048 + frag_off[ks]);
        }
    };
    // the three products of one 32-deep step; small terms first, the hi * hi product last
    auto products = [&](const bf16x8_t (&ah)[FM], const bf16x8_t (&al)[FM], const bf16x8_t (&bh)[FN], const bf16x8_t (&bl)[FN])
                        __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
    };
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const char* ta = smem + cur * X3_STAGE + wm * TM * 128;
        const char* tb = smem + cur * X3_STAGE + 2 * X3_PIECE + wn * TN * 128;
        // order (pinned): fragments of step 0 | request the next tile | fragments of step 1 | products 0 | products 1.  The
        // second step's LDS latency and the DMA issue sit under the first step's 24 MFMAs.
        bf16x8_t ah0[FM], al0[FM], bh0[FN], bl0[FN], ah1[FM], al1[FM], bh1[FN], bl1[FN];
        load_frags(ta, tb, 0, ah0, al0, bh0, bl0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(ta, tb, 1, ah1, al1, bh1, bl1);
        __builtin_amdgcn_sched_barrier(0);
        products(ah0, al0, bh0, bl0);
        __builtin_amdgcn_sched_barrier(0);
        products(ah1, al1, bh1, bl1);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    gemm_epilogue_f32io<FM, FN, TM, TN>(p, acc, m0, n0, wm, wn, lane);
}

}  // namespace

// p as filled by advgrpo_conv3x3_nhwc_x3 (gemm.hip): Cin = 3C, K = 9 * 3C, lda = 3C, ldw = 27C, f32_io
int conv3x3_x3_launch(const GemmParams& p, hipStream_t s) {
    ADVGRPO_CHECK(p.conv && p.f32_io && p.Cin % 192 == 0 && p.zero_page && p.batch == 1 && p.splitk == 1,
                  "conv3x3_x3: bad parameter block");
    constexpr int BM = 192, LDS = x3_lds_bytes(BM);
    static bool attr_set = false;
    if (!attr_set) {
        ADVGRPO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_x3_kernel<BM>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess,
                      "conv3x3_x3: %d bytes of LDS refused", LDS);
        attr_set = true;
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + X3_BN - 1) / X3_BN);
    hipLaunchKernelGGL(conv3x3_x3_kernel<BM>, dim3(tiles), dim3(512), LDS, s, p);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

}  // namespace advgrpo
