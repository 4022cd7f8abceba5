// gemm4w.hip -- EXPERIMENT (variant 31, ADVGRPO_GEMM_FORCE=31): the eight-phase GEMM's wave tile and epilogue in a 4-wave workgroup,
// TWO workgroups per CU, so that one workgroup's epilogue (VALU time + memory burst) runs under the other's k loop.
//
//   tile 128 (M) x 256 (N) x 64, waves 1 x 4, each wave 128 x 64 of C (the same 128 accumulator VGPRs, fragments and
//   fused epilogue as gemm8p_kernel.hpp); 72 KiB of LDS: three 8 KiB slots for the A items (A sub s = rows s*64..+63) and
//   three 16 KiB slots for the W items (W sub s = columns s*32..+31 of the four wave columns), items of a kind rotate
//   through their three slots.  One item is requested per phase, three to four phases before its first read:
//       ph0  wait W1(k)          barrier  read W1(k)            request W0(k+1)   compute (A0, W0)
//       ph1  wait A1(k)          barrier  read A1(k)            request W1(k+1)   compute (A0, W1)
//       ph2                                                     request A1(k+1)   compute (A1, W1)
//       ph3  wait A0,W0(k+1)     barrier  read A0(k+1), W0(k+1) request A0(k+2)   compute (A1, W0)
//   There is no partner wave group inside the workgroup: a wave reads the fragments of the NEXT phase before it issues the
//   MFMAs of the current one, and the second workgroup of the CU fills what is left.  Hazards: a request into a slot comes
//   at least one barrier after every wave consumed the item that was there (RAW: counted vmcnt + barrier before the read).
#include "../gemm8p_kernel.hpp"

namespace advgrpo {

namespace {

constexpr int W4_BM = 128, W4_BN = 256, W4_BK = 64;
constexpr int W4_AITEM = 8192, W4_WITEM = 16384;
constexpr int W4_W_REGION = 3 * W4_AITEM;
constexpr int W4_LDS = 3 * W4_AITEM + 3 * W4_WITEM;       // 72 KiB: two workgroups per CU

struct W4Tile {
    int m0, n0, nk;
    const char* a_bytes;
    const char* w_bytes;
    uint32_t a_off[2][2], b_off[2];        // W: the four instructions of an item are 64 columns apart (uniform stride w_step)
    int64_t w_step;
};

__device__ __forceinline__ void w4_setup(W4Tile& t, const GemmParams& pr, int id, int wave) {
    const GemmParams* p = &pr;
    const int lane = p8_lane();
    const int tiles_n = (p->N + W4_BN - 1) / W4_BN, tiles_m = (p->M + W4_BM - 1) / W4_BM;
    int tile_m, tile_n;
    tile_coords(id, tiles_m, tiles_n, 8, tile_m, tile_n);
    t.a_bytes = reinterpret_cast<const char*>(p->A);
    t.w_bytes = reinterpret_cast<const char*>(p->W);
    t.m0 = tile_m * W4_BM;
    t.n0 = tile_n * W4_BN;
    t.nk = p->K / W4_BK;
    const int lrow = lane >> 3;
    const int schunk = (lane & 7) ^ lrow;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            int r = t.m0 + sub * 64 + (wave + it * 4) * 8 + lrow;
            r = r < p->M ? r : p->M - 1;
            int64_t ar = r;
            if (p->a_seg_rows > 0) {
                const int bi = r / p->a_seg_rows;
                ar = (int64_t)bi * p->a_seg_stride + p->a_seg_off + (r - bi * p->a_seg_rows);
            }
            t.a_off[sub][it] = (uint32_t)((ar * p->lda + schunk * 8) * 2);
        }
        {   // slot row sr = (wave + 4 it) * 8 + lrow -> column (sr >> 5) * 64 + sub * 32 + (sr & 31): it adds 64 columns
            // (N % 256 == 0 is required of this variant: no clamping)
            const int sr = wave * 8 + lrow;
            const int n = t.n0 + sub * 32 + sr;
            t.b_off[sub] = (uint32_t)(((int64_t)n * p->ldw + schunk * 8) * 2);
        }
    }
    t.w_step = (int64_t)64 * p->ldw * 2;
}

__device__ __forceinline__ void w4_stage_a(const W4Tile& t, char* smem, int wave, int slot, int sub, int kt) {
    if (kt >= t.nk) return;
    char* base = smem + slot * W4_AITEM;
    const char* src = t.a_bytes + kt * (W4_BK * 2);
#pragma unroll
    for (int it = 0; it < 2; ++it) p8_dma16(src, t.a_off[sub][it], lds_addr(base + (wave + it * 4) * 1024));
}
__device__ __forceinline__ void w4_stage_w(const W4Tile& t, char* smem, int wave, int slot, int sub, int kt) {
    if (kt >= t.nk) return;
    char* base = smem + W4_W_REGION + slot * W4_WITEM;
    const char* src = t.w_bytes + kt * (W4_BK * 2);
#pragma unroll
    for (int it = 0; it < 4; ++it) p8_dma16(src + it * t.w_step, t.b_off[sub], lds_addr(base + (wave + it * 4) * 1024));
}

}  // namespace

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm4w_kernel(const GemmPair pp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);     // = wave column
    const int bid = blockIdx.x;
    const bool second = bid >= pp.tiles_a;
    W4Tile t;
    {
        const int tiles = second ? (int)gridDim.x - pp.tiles_a : pp.tiles_a;
        const int id = xcd_remap(second ? bid - pp.tiles_a : bid, tiles);
        if (second) w4_setup(t, pp.b, id, wave);
        else w4_setup(t, pp.a, id, wave);
    }
    const int nk = t.nk;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int frag_off[2];
    {
        const int l = p8_lane();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) frag_off[ks] = (l & 15) * 128 + (((ks * 4 + (l >> 4)) ^ (l & 7)) << 4);
    }
    const int b_base = wave * 32 * 128;
    auto read_a = [&](int slot, bf16x8_t (&a)[2][4]) __attribute__((always_inline)) {
        const char* item = smem + slot * W4_AITEM;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) a[ks][i] = *reinterpret_cast<const bf16x8_t*>(item + i * 16 * 128 + frag_off[ks]);
    };
    auto read_b = [&](int slot, bf16x8_t (&b)[2][2]) __attribute__((always_inline)) {
        const char* item = smem + W4_W_REGION + slot * W4_WITEM + b_base;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) b[ks][j] = *reinterpret_cast<const bf16x8_t*>(item + j * 16 * 128 + frag_off[ks]);
    };
    // wait until at most `n` of this wave's DMA instructions are in flight (counted only while every younger item exists)
#define W4_SYNC(N, STEADY)                                                                                              \
    do {                                                                                                                \
        if (STEADY) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");                                               \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        __builtin_amdgcn_s_barrier();                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)
#define W4_COMPUTE(MH, NH, AF, BF)                                                                                      \
    do {                                                                                                                \
        __builtin_amdgcn_s_setprio(1);                                                                                  \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                               \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
                    acc[(MH) * 4 + i][(NH) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                          \
                        BF[ks][j], AF[ks][i], acc[(MH) * 4 + i][(NH) * 2 + j], 0, 0, 0);                                \
        __builtin_amdgcn_s_setprio(0);                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)

    // 96 fragment VGPRs: A sub 0 / A sub 1 and two W sets that swap roles every k-tile (W sub 1 of k-tile k dies after ph2,
    // its registers take W sub 0 of k-tile k+1 in ph3; A sub 0 dies after ph1 and is re-read for k+1 in ph3)
    bf16x8_t af0[2][4], af1[2][4], bx[2][2], by[2][2];
    // k-tile k with slots (s0, s1, s2): A0(k), W0(k) came from s0; A1(k), W1(k) sit in s1; A0(k+1), W0(k+1) in s2
    auto ktile = [&](int k, int s0, int s1, int s2, bf16x8_t (&b0)[2][2], bf16x8_t (&b1)[2][2]) __attribute__((always_inline)) {
        const bool steady = k + 1 < nk;
        W4_SYNC(4, steady);                        // W1(k) landed (younger: A1(k), A0(k+1))
        read_b(s1, b1);
        w4_stage_w(t, smem, wave, s2, 0, k + 1);
        W4_COMPUTE(0, 0, af0, b0);
        W4_SYNC(6, steady);                        // A1(k) landed (younger: A0(k+1), W0(k+1))
        read_a(s1, af1);
        w4_stage_w(t, smem, wave, s0, 1, k + 1);
        W4_COMPUTE(0, 1, af0, b1);
        w4_stage_a(t, smem, wave, s0, 1, k + 1);
        W4_COMPUTE(1, 1, af1, b1);
        W4_SYNC(6, steady);                        // A0(k+1), W0(k+1) landed (younger: W1(k+1), A1(k+1))
        if (steady) {
            read_a(s2, af0);
            read_b(s2, b1);                        // (becomes b0 of the next k-tile)
        }
        w4_stage_a(t, smem, wave, s1, 0, k + 2);
        W4_COMPUTE(1, 0, af1, b0);
    };
    // prologue: A0(0) W0(0) W1(0) A1(0), then the state "ph3 of k-tile -1"
    w4_stage_a(t, smem, wave, 0, 0, 0);
    w4_stage_w(t, smem, wave, 0, 0, 0);
    w4_stage_w(t, smem, wave, 1, 1, 0);
    w4_stage_a(t, smem, wave, 1, 1, 0);
    W4_SYNC(6, true);
    read_a(0, af0);
    read_b(0, bx);
    w4_stage_a(t, smem, wave, 2, 0, 1);
    int s0 = 0, s1 = 1, s2 = 2;
    int k = 0;
#pragma unroll 1
    for (; k + 1 < nk; k += 2) {
        ktile(k, s0, s1, s2, bx, by);
        ktile(k + 1, s2, s0, s1, by, bx);
        const int n0_ = s1, n1_ = s2, n2_ = s0;         // two rotations (s0,s1,s2) -> (s2,s0,s1) -> (s1,s2,s0)
        s0 = n0_; s1 = n1_; s2 = n2_;
    }
    if (k < nk) ktile(k, s0, s1, s2, bx, by);
#undef W4_SYNC
#undef W4_COMPUTE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // every wave is done with the ring: it becomes the epilogue scratch
    {
        const GemmParams& p = second ? pp.b : pp.a;
        p8_epilogue<EPI>(p, acc, t.m0, t.n0 + wave * 64, smem + wave * P8_SCRATCH);
    }
}

template <int EPI>
static int launch4w(const GemmPair& pp, int tiles_total, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm4w_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm4w_kernel<EPI>), dim3(tiles_total), dim3(256), W4_LDS, s, pp);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

static int w4_tiles(const GemmParams& p) { return ((p.M + W4_BM - 1) / W4_BM) * ((p.N + W4_BN - 1) / W4_BN); }

// epilogue class as in gemm8p.hip (rollout classes only: the experiment is measured on the rollout)
static int w4_class(const GemmParams& p) {
    if (p.out_dtype != ADVGRPO_BF16) return EPI_GENERIC;
    int f = 0;
    if (p.bias) f |= F_BIAS;
    if (p.rms_w) f |= F_RMS;
    if (p.act == ACT_GELU_TANH) f |= F_GELU;
    else if (p.act != ACT_NONE) return EPI_GENERIC;
    if (p.aux_in || p.aux_out) return EPI_GENERIC;
    if (p.gate && p.residual) f |= F_GATE_RES;
    else if (p.gate || p.residual) return EPI_GENERIC;
    switch (f) {
        case EPI_BIAS: case EPI_BIAS_RMS: case EPI_BIAS_GELU: case EPI_BIAS_GATE_RES: return f;
    }
    return EPI_GENERIC;
}

int gemm4w_launch_pair(const GemmParams* a, const GemmParams* b, hipStream_t s) {
    GemmPair pp{};
    pp.a = *a;
    pp.b = b ? *b : *a;
    pp.tiles_a = w4_tiles(pp.a);
    const int total = pp.tiles_a + (b ? w4_tiles(pp.b) : 0);
    int epi = w4_class(pp.a);
    if (b && w4_class(pp.b) != epi) epi = EPI_GENERIC;
    switch (epi) {
        case EPI_BIAS: return launch4w<EPI_BIAS>(pp, total, s);
        case EPI_BIAS_RMS: return launch4w<EPI_BIAS_RMS>(pp, total, s);
        case EPI_BIAS_GELU: return launch4w<EPI_BIAS_GELU>(pp, total, s);
        case EPI_BIAS_GATE_RES: return launch4w<EPI_BIAS_GATE_RES>(pp, total, s);
        default: return launch4w<EPI_GENERIC>(pp, total, s);
    }
}

}  // namespace advgrpo
