// gemm.hip -- bf16 MFMA GEMM with fused prologue/epilogue for gfx950 (MI355X).
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )         A, W bf16 (K contiguous), f32 accumulate
//
// This one kernel family carries every dense contraction of the hot path: the MMDiT
// QKV / out-proj / MLP / adaLN linears (reference call sites
// sd3_pipeline_with_logprob_fast.py:630-637 and train_sd3_fast_pickscore.py:235-255, which run
// diffusers' SD3Transformer2DModel), the ViT reward towers, the text encoders, and (through the implicit-GEMM A
// loader, CONV = true) the VAE decoder.
//
// CDNA4 mapping (two-stage kernel, the one the dispatcher uses; the ring / ping-pong kernels below share its pieces)
//   * 512 threads = 8 wave64; block tile BM x BN x 64 (192x128 for the image-stream Linears, 128x128 otherwise), wave
//     tiles of v_mfma_f32_16x16x32_bf16 fragments; 80 / 64 KB of LDS and <= 128 VGPRs so that TWO workgroups share a CU
//     (the register bound is explicit in __launch_bounds__: drifting past it once halved the occupancy).
//   * HBM -> LDS by global_load_lds_dwordx4 (16 B/lane, no VGPR round trip) in whole 128-byte rows, double-buffered;
//     the next tile's DMA is issued before the current tile's fragment reads and MFMAs.
//   * LDS image is lane-linear per wave instruction (8 rows x 128 B); bank conflicts of the column-slice ds_read_b128
//     are removed by an XOR swizzle applied to the per-lane SOURCE chunk (chunk ^= row & 7) and again on the read.
//   * MFMA is issued with operands swapped (W fragment as A, activation fragment as B): a lane owns 4 consecutive output
//     columns of one row; the epilogue then bounces 16-row slabs through LDS to get 8 consecutive columns per lane and
//     whole 128-byte rows per 8 lanes (16-byte loads of bias / gate / residual, 16-byte stores).
//   * block ids are remapped so each XCD (private 4 MiB L2) walks a contiguous range of tiles, in groups of 4 row-tiles.
//   * a launch can carry two problems (GemmPair): the text-stream Linear rides in the tail of the image-stream one.
//   * rows >= M / N are clamped on load and masked on store; K must be a multiple of 64.
#include <stdlib.h>

#include "gemm_device.hpp"

namespace advgrpo {

template <int BM, int BN, int WM, int WN, bool CONV>
__device__ __forceinline__ void gemm_bf16_body(const GemmParams& p, const int bid, const int by, char* smem) {
    constexpr int BK = 64;
    constexpr int NW = WM * WN;                 // waves per workgroup
    constexpr int TM = BM / WM, TN = BN / WN;   // wave tile
    constexpr int FM = TM / 16, FN = TN / 16;   // fragments per wave
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    constexpr int STAGE = A_BYTES + B_BYTES;
    // DMA instructions (8 rows each) per tile and per wave; when the wave count does not divide them (6 waves on a
    // 128 x 192 tile) the last round is issued by the first waves only (wave-uniform guard)
    constexpr int A_TOTAL = BM / 8, B_TOTAL = BN / 8;
    constexpr int A_INST = (A_TOTAL + NW - 1) / NW, B_INST = (B_TOTAL + NW - 1) / NW;
    static_assert(BM % 16 == 0 && BN % 16 == 0 && TM % 16 == 0 && TN % 16 == 0, "tile / wave tile must be fragment multiples");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    const int swz = xcd_remap(bid, nwg);
    int tile_m, tile_n;
    tile_coords(swz, tiles_m, tiles_n, (p.debug >> 8) ? (p.debug >> 8) : 4, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int bz = by / p.splitk, sk = by % p.splitk;

    const bf16_t* __restrict__ A = p.A + (int64_t)bz * p.strideA;
    const bf16_t* __restrict__ W = p.W + (int64_t)bz * p.strideW;

    // ---- per-lane DMA source pointers (row clamped; chunk pre-swizzled)
    const int lrow = lane >> 3;                  // row inside the 8-row instruction
    const int schunk = (lane & 7) ^ lrow;        // source 16-B chunk for this lane's LDS slot
    const bf16_t* a_src[A_INST];
    const bf16_t* b_src[B_INST];
    int a_y[A_INST], a_x[A_INST];        // CONV: output pixel of this lane's row
    int64_t a_img[A_INST];               // CONV: element offset of the image (b) in the input
#pragma unroll
    for (int it = 0; it < A_INST; ++it) {
        int r = m0 + (wave + it * NW) * 8 + lrow;
        r = r < p.M ? r : p.M - 1;
        if (p.debug & 64) r = (wave + it * NW) * 8 + lrow;   // experiment: every tile streams the same rows (all L2 hits)
        if constexpr (CONV) {
            const int hw = p.Hout * p.Wout;
            const int bi = r / hw, rem = r - bi * hw;
            a_y[it] = rem / p.Wout;
            a_x[it] = rem - a_y[it] * p.Wout;
            a_img[it] = (int64_t)bi * (p.Hout >> p.ups) * (p.Wout >> p.ups) * p.Cin;
            a_src[it] = nullptr;
        } else {
            int64_t ar = r;
            if (p.a_seg_rows > 0) {
                const int bi = r / p.a_seg_rows;
                ar = (int64_t)bi * p.a_seg_stride + p.a_seg_off + (r - bi * p.a_seg_rows);
            }
            a_src[it] = A + ar * p.lda + schunk * 8;
        }
    }
#pragma unroll
    for (int it = 0; it < B_INST; ++it) {
        int r = n0 + (wave + it * NW) * 8 + lrow;
        r = r < p.N ? r : p.N - 1;
        if (p.debug & 64) r = (wave + it * NW) * 8 + lrow;
        b_src[it] = W + (int64_t)r * p.ldw + schunk * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE;
        if constexpr (CONV) {
            // one 64-wide k tile lies inside one filter tap (Cin % 64 == 0)
            const int k0 = kt * BK;
            const int tap = k0 / p.Cin, c0 = k0 - tap * p.Cin;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const int win = p.Wout >> p.ups;
#pragma unroll
            for (int it = 0; it < A_INST; ++it) {
                const int yy = a_y[it] + dy, xx = a_x[it] + dx;
                const bool ok = (unsigned)yy < (unsigned)p.Hout && (unsigned)xx < (unsigned)p.Wout;
                const bf16_t* src = ok ? A + a_img[it] + ((int64_t)(yy >> p.ups) * win + (xx >> p.ups)) * p.Cin + c0 +
                                             schunk * 8
                                       : p.zero_page + schunk * 8;
                if (A_TOTAL % NW == 0 || wave + it * NW < A_TOTAL)
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lds_ptr_t)(base + (wave + it * NW) * 1024), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int it = 0; it < A_INST; ++it)
                if (A_TOTAL % NW == 0 || wave + it * NW < A_TOTAL)
                    __builtin_amdgcn_global_load_lds((gptr_t)(a_src[it] + kt * BK),
                                                     (lds_ptr_t)(base + (wave + it * NW) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < B_INST; ++it)
            if (B_TOTAL % NW == 0 || wave + it * NW < B_TOTAL)
                __builtin_amdgcn_global_load_lds((gptr_t)(b_src[it] + kt * BK),
                                                 (lds_ptr_t)(base + A_BYTES + (wave + it * NW) * 1024), 16, 0, 0);
    };

    // ---- fragment read offsets (bytes) inside a tile: row (lane&15), swizzled chunk
    int frag_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        frag_off[ks] = (lane & 15) * 128 + (((ks * 4 + (lane >> 4)) ^ (lane & 7)) << 4);

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // k-tile range of this split (whole range when splitk == 1)
    const int nk_all = p.K / BK;
    const int kt0 = (int)((int64_t)nk_all * sk / p.splitk), kt1 = (int)((int64_t)nk_all * (sk + 1) / p.splitk);
    const int nk = kt1 - kt0;
    if (nk <= 0) return;
    stage(0, kt0);
    __syncthreads();  // hipcc drains the LDS-DMA (vmcnt 0) before the barrier
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk && !(p.debug & 1)) stage(cur ^ 1, kt0 + kt + 1);
        const char* ta = smem + cur * STAGE + wm * TM * 128;
        const char* tb = smem + cur * STAGE + A_BYTES + wn * TN * 128;
        // both k-steps' fragments are requested up front into separate registers, so the LDS latency of the second
        // set is covered by the first set's MFMAs (left alone the compiler reuses one register set and waits for
        // LDS twice per tile); sched_barriers pin the order
        bf16x8_t af[2][FM], bf[2][FN];
        if (!(p.debug & 2) || kt == 0)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
                af[ks][i] = *reinterpret_cast<const bf16x8_t*>(ta + i * 2048 + frag_off[ks]);
#pragma unroll
            for (int j = 0; j < FN; ++j)
                bf[ks][j] = *reinterpret_cast<const bf16x8_t*>(tb + j * 2048 + frag_off[ks]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }

    if (p.debug & 4) { if (acc[0][0][0] != 12345.678f) return; }
    gemm_epilogue<FM, FN, TM, TN, CONV>(p, acc, m0, n0, wm, wn, bz, lane, smem, wave);
}

// waves per SIMD the register allocation must leave room for: two workgroups per CU whenever their LDS fits (<= 80 KB
// each).  Without the bound the 192x128 kernel drifted to 136 VGPRs after an unrelated epilogue change, i.e. 3 waves per
// SIMD = ONE 8-wave workgroup per CU, and lost 25 %.
template <int BM, int BN, int NW>
constexpr int gemm_min_waves_per_simd() { return (2 * (BM + BN) * 64 * 2 <= 81920 ? 2 : 1) * NW / 4; }

template <int BM, int BN, int WM, int WN, bool CONV>
__global__ __launch_bounds__(WM * WN * 64, (gemm_min_waves_per_simd<BM, BN, WM * WN>())) void gemm_bf16_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_bf16_body<BM, BN, WM, WN, CONV>(p, blockIdx.x, blockIdx.y, smem);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, (gemm_min_waves_per_simd<BM, BN, WM * WN>())) void gemm_bf16_pair_kernel(const GemmPair pp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = blockIdx.x;
    if (bid < pp.tiles_a) gemm_bf16_body<BM, BN, WM, WN, false>(pp.a, bid, 0, smem);
    else gemm_bf16_body<BM, BN, WM, WN, false>(pp.b, bid - pp.tiles_a, 0, smem);
}

#ifdef ADVGRPO_EXPERIMENTS
#include "experiments/gemm_variants.inc"
#endif

template <int BM, int BN, int WM, int WN, bool CONV>
static int launch(const GemmParams& p, hipStream_t s) {
    constexpr int lds = 2 * (BM + BN) * 64 * 2;
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<BM, BN, WM, WN, CONV>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, CONV>), dim3(tiles, p.batch * p.splitk), dim3(WM * WN * 64), lds, s, p);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

template <int BM, int BN, int WM, int WN>
static int launch_pair(const GemmParams& a, const GemmParams& b, hipStream_t s) {
    constexpr int lds = 2 * (BM + BN) * 64 * 2;
    GemmPair pp{a, b, ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN)};
    const int tiles_b = ((b.M + BM - 1) / BM) * ((b.N + BN - 1) / BN);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_pair_kernel<BM, BN, WM, WN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_pair_kernel<BM, BN, WM, WN>), dim3(pp.tiles_a + tiles_b), dim3(WM * WN * 64), lds, s, pp);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

// ---- tile variants (ids returned by advgrpo_gemm_variant):
//   two-stage BK=64 kernel: 0 = 128x128 (4 waves), 1 = 128x64, 2 = 64x128, 15 = 128x128 (8 waves 4x2), 26 = 192x128;
//   conv: 5 = 128x64, 18 = 128x128 (8 waves); 30 = the 256x256 eight-phase persistent kernel of gemm8p.hip.
//   An experiments build (make EXPERIMENTS=1 -> libadvgrpo_experiments.so, never the product library) adds 14 / 27 (other
//   wave grids), 4 (conv, 4 waves), 11 / 17 (BK=32 ring kernel), 20 / 21 / 23 (ping-pong kernel), 31 (experiments/gemm4w.hip)
//   and the ADVGRPO_GEMM_* environment overrides used for in-situ A/B runs; the product library reads no environment.
#ifdef ADVGRPO_EXPERIMENTS
struct ExperimentKnobs {
    int force = -1, no8p = 0, use4w = 0, debug = 0, fn = -1, fk = 0, fv = 0;
    ExperimentKnobs() {
        if (const char* e = getenv("ADVGRPO_GEMM_FORCE")) force = atoi(e);
        if (const char* e = getenv("ADVGRPO_GEMM_NO8P")) no8p = atoi(e) ? 1 : 0;
        if (const char* e = getenv("ADVGRPO_GEMM_4W")) use4w = atoi(e);
        if (const char* e = getenv("ADVGRPO_GEMM_DEBUG")) debug = atoi(e);
        if (const char* e = getenv("ADVGRPO_GEMM_FORCE_NK")) sscanf(e, "%d:%d:%d", &fn, &fk, &fv);
    }
};
static const ExperimentKnobs& knobs() { static ExperimentKnobs k; return k; }
#endif

static int cu_count() {   // MI355X: 256; asked once (a host without a device, e.g. a build box calling advgrpo_gemm_variant, gets 256)
    static const int cus = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) {
            (void)hipGetLastError();
            return 256;
        }
        return prop.multiProcessorCount;
    }();
    return cus;
}

static bool small_m_off() {
#ifdef ADVGRPO_EXPERIMENTS
    static const bool off = [] { const char* e = getenv("ADVGRPO_GEMM_SMALLM"); return e && atoi(e) == 0; }();
    return off;
#else
    return false;
#endif
}

static int gemm_variant(int M, int N, int K, int batch, int conv, int plain) {   // batch includes the split-K factor
    if (N <= 64) return conv ? 5 : 1;
#ifdef ADVGRPO_EXPERIMENTS
    if (knobs().force >= 0) return conv ? (knobs().force == 18 ? 18 : 4) : knobs().force;
    if (knobs().force == -3) return conv ? 4 : 0;
    if (plain && M >= 8192 && N >= 1024 && batch == 1 && !knobs().no8p && N % 256 == 0 &&
        (knobs().use4w == 1 || (knobs().use4w == 2 && K <= 2048)))
        return 31;
    if (knobs().no8p && plain && M >= 8192 && N >= 1024) return 26;
#endif
    if (conv) return 18;
    if (M <= 64) return 2;
    // measured (scripts/bench_gemm.py): 8 waves per workgroup (16 waves per CU) beat 4 on every MMDiT / ViT shape.
    // Wide Linears (M >= 8192 rows, N >= 1024): the 256x256 eight-phase kernel; where its preconditions fail
    // (gemm8p_ok), the 192x128 two-stage tile (80 KB of LDS: still two workgroups per CU, 17 % fewer L2->LDS bytes per
    // flop than 128x128), chosen from in-situ runs of the whole rollout step with one shape forced to each candidate.
    if (plain && M >= 8192 && N >= 1024 && batch == 1) return 30;
    if (plain && M >= 8192 && N >= 1024) return 26;
    if (plain && batch == 1 && M >= 1024 && K >= 1024 && !small_m_off()) {
        // A reward tower's Linears (CLIP ViT-H: 8 images x 257 tokens = 2056 rows) are one partial round of the chip: count
        // workgroup slots (two workgroups per CU).  Measured at 2056 / 4112 rows (profiles/r6_vit_tower.txt): a ragged
        // second round of 128x128 tiles costs more than one round of 192x128 tiles (FC1: 680 -> 440 tiles, 53 -> 45 us), and a
        // launch that fills under half the slots runs faster as twice as many 128x64 tiles (out-proj / FC2: 170 -> 340 tiles,
        // 20.7 -> 19.0 and 59.1 -> 52.9 us).
        const long slots = 2L * cu_count();
        const long rows128 = (M + 127) / 128, rows192 = (M + 191) / 192, cols = (N + 127) / 128;
        const long t128 = rows128 * cols, t192 = rows192 * cols;
        if (t128 * 2 <= slots) return 1;
        if (t128 > slots && 3 * ((t192 + slots - 1) / slots) <= 2 * ((t128 + slots - 1) / slots)) return 26;
    }
    return plain ? 15 : 0;
}

// validation shared by the single and the paired launch; returns the tile variant or -1
static int gemm_prepare(GemmParams& p) {
#ifdef ADVGRPO_EXPERIMENTS
    p.debug = knobs().debug;
#endif
    ADVGRPO_CHECK(p.A && p.W && p.C, "gemm: null operand");
    ADVGRPO_CHECK(p.M > 0 && p.N > 0 && p.K > 0 && p.K % 64 == 0, "gemm: need M,N>0 and K %% 64 == 0 (M=%d N=%d K=%d)",
                  p.M, p.N, p.K);
    ADVGRPO_CHECK(p.lda % 8 == 0 && p.ldw % 8 == 0, "gemm: lda/ldw must be multiples of 8 elements (16 B rows)");
    ADVGRPO_CHECK(p.out_dtype == ADVGRPO_BF16 || p.out_dtype == ADVGRPO_F32, "gemm: bad out dtype");
    ADVGRPO_CHECK(p.batch >= 1 && p.splitk >= 1, "gemm: batch / splitk must be >= 1");
    ADVGRPO_CHECK(p.splitk == 1 || (p.out_dtype == ADVGRPO_F32 && !p.bias && !p.gate && !p.residual && p.act == 0 &&
                                    !p.aux_out),
                  "gemm: split-K accumulates atomically into f32 and takes no other epilogue");
    ADVGRPO_CHECK(p.act < ACT_DGELU_TANH || p.aux_in, "gemm: d-activation epilogue needs aux_in");
    ADVGRPO_CHECK(!(p.aux_out || p.aux_in) || ((p.ld_aux & 3) == 0 && (p.N & 3) == 0), "gemm: aux needs N, ld_aux %% 4 == 0");
    int variant = gemm_variant(p.M, p.N, p.K, p.batch * p.splitk, p.conv, p.splitk == 1 && !p.conv);
#ifdef ADVGRPO_EXPERIMENTS   // ADVGRPO_GEMM_FORCE_NK=N:K:variant overrides the tile variant of one Linear shape
    if (knobs().fn == p.N && knobs().fk == p.K && !p.conv && p.splitk == 1 && p.M >= 8192) variant = knobs().fv;
#endif
    if (variant == 30 && !gemm8p_ok(p)) variant = p.M >= 8192 && p.N >= 1024 ? 26 : 15;
    if (p.rms_w && (variant == 27 || variant == 1)) variant = 15;   // the fused QK-norm needs 64-wide wave tiles (one head per wave row)
    if (p.conv) {
        ADVGRPO_CHECK(p.Cin % 64 == 0 && p.K == 9 * p.Cin && p.zero_page && p.batch == 1,
                      "conv3x3: need Cin %% 64 == 0, K == 9*Cin, a zero page and batch 1 (Cin=%d K=%d)", p.Cin, p.K);
        ADVGRPO_CHECK((p.Hout % (1 << p.ups)) == 0 && (p.Wout % (1 << p.ups)) == 0, "conv3x3: bad upsample shape");
    }
    if (p.rms_w) {   // the fused QK-norm lives in the row-coalesced epilogue of the 64-wide wave tiles only
        auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        const bool narrow = variant == 1 || variant == 5 || variant == 14 || variant == 27;
        ADVGRPO_CHECK(!narrow && !p.conv && p.splitk == 1 && p.batch == 1 && p.N % 64 == 0 && p.ldc % 8 == 0 &&
                          p.out_dtype == ADVGRPO_BF16 && a16(p.C) && a16(p.bias) && a16(p.rms_w) && p.rms_nheads > 0 &&
                          p.rms_hpw > 0 && !p.gate && !p.residual && !p.aux_out && !p.aux_in && p.act == ACT_NONE,
                      "gemm: fused QK-norm needs a plain bf16 projection with N %% 64 == 0 and 16-byte aligned rows");
    }
    return variant;
}

int gemm_bf16_pair(const GemmParams& a_in, const GemmParams& b_in, hipStream_t s) {
    GemmParams a = a_in, b = b_in;
    const int va = gemm_prepare(a);
    if (va < 0) return -1;
    const int vb = gemm_prepare(b);
    if (vb < 0) return -1;
    const bool pairable = a.batch == 1 && b.batch == 1 && a.splitk == 1 && b.splitk == 1 && !a.conv && !b.conv &&
                          !(a.debug & 32);
    if (pairable && va == 15) return launch_pair<128, 128, 4, 2>(a, b, s);
#ifdef ADVGRPO_EXPERIMENTS
    if (pairable && va == 17) return launch_pipe_pair<256, 128, 3, 4, 2>(a, b, s);
    if (pairable && va == 27) return launch_pair<128, 192, 2, 4>(a, b, s);
    if (pairable && va == 31 && gemm8p_ok(a) && gemm8p_ok(b) && a.N % 256 == 0 && b.N % 256 == 0 && a.K % 64 == 0 && b.K % 64 == 0)
        return gemm4w_launch_pair(&a, &b, s);
#endif
    if (pairable && va == 26) return launch_pair<192, 128, 4, 2>(a, b, s);
    if (pairable && va == 30) return gemm8p_launch_pair(a, b, s);
    const int rc = gemm_bf16(a_in, s);
    return rc ? rc : gemm_bf16(b_in, s);
}

int gemm_bf16(const GemmParams& p_in, hipStream_t s) {
    GemmParams p = p_in;
    const int variant = gemm_prepare(p);
    if (variant < 0) return -1;
    switch (variant) {
        case 0: return launch<128, 128, 2, 2, false>(p, s);
        case 1: return launch<128, 64, 2, 2, false>(p, s);
        case 2: return launch<64, 128, 2, 2, false>(p, s);
        case 5: return launch<128, 64, 2, 2, true>(p, s);
        case 18: return launch<128, 128, 4, 2, true>(p, s);
        case 15: return launch<128, 128, 4, 2, false>(p, s);
        case 26: return launch<192, 128, 4, 2, false>(p, s);
        case 30: return gemm8p_launch(p, s);
#ifdef ADVGRPO_EXPERIMENTS
        case 4: return launch<128, 128, 2, 2, true>(p, s);
        case 11: return launch_pipe<128, 128, 3, 4, 2>(p, s);
        case 14: return launch<128, 128, 2, 4, false>(p, s);
        case 17: return launch_pipe<256, 128, 3, 4, 2>(p, s);
        case 20: return launch_pp<256, 256, 4, 2, 4>(p, s);
        case 21: return launch_pp<256, 128, 4, 4, 2>(p, s);
        case 23: return launch_pp<256, 128, 3, 4, 2>(p, s);
        case 27: return launch<128, 192, 2, 4, false>(p, s);
        case 31:
            if (gemm8p_ok(p) && p.N % 256 == 0 && p.K % 64 == 0) return gemm4w_launch_pair(&p, nullptr, s);
            return launch<128, 128, 4, 2, false>(p, s);
#endif
    }
    set_error("gemm: bad variant %d", variant);
    return -1;
}

}  // namespace advgrpo

using namespace advgrpo;

extern "C" int advgrpo_gemm_variant(int M, int N, int K, int batch, int conv) {
    return gemm_variant(M, N, K, batch < 1 ? 1 : batch, conv, !conv);
}

extern "C" int advgrpo_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                                 int out_dtype, int M, int N, int K, const void* bias, int act, float alpha,
                                 const void* gate, int64_t gate_stride, int gate_rows, const void* residual,
                                 int64_t ldr, int seg_rows, int64_t seg_stride, int64_t seg_off, int a_seg_rows,
                                 int64_t a_seg_stride, int64_t a_seg_off, int batch, int64_t strideA,
                                 int64_t strideW, int64_t strideC, void* stream) {
    GemmParams p{};
    p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.C = C;
    p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.out_dtype = out_dtype;
    p.M = M; p.N = N; p.K = K;
    p.bias = (const bf16_t*)bias; p.act = act; p.alpha = alpha;
    p.gate = (const bf16_t*)gate; p.gate_stride = gate_stride; p.gate_rows = gate_rows; p.gate_batch_stride = 0;
    p.residual = (const bf16_t*)residual; p.ldr = ldr; p.strideR = strideC;
    p.seg_rows = seg_rows; p.seg_stride = seg_stride; p.seg_off = seg_off;
    p.a_seg_rows = a_seg_rows; p.a_seg_stride = a_seg_stride; p.a_seg_off = a_seg_off;
    p.batch = batch < 1 ? 1 : batch; p.strideA = strideA; p.strideW = strideW; p.strideC = strideC;
    p.splitk = 1;
    return gemm_bf16(p, as_stream(stream));
}

static GemmParams from_desc(const advgrpo_gemm_desc& d) {
    GemmParams p{};
    p.A = (const bf16_t*)d.A; p.W = (const bf16_t*)d.W; p.C = d.C;
    p.lda = d.lda; p.ldw = d.ldw; p.ldc = d.ldc; p.out_dtype = d.out_dtype;
    p.M = d.M; p.N = d.N; p.K = d.K;
    p.bias = (const bf16_t*)d.bias; p.act = d.act; p.alpha = d.alpha;
    p.gate = (const bf16_t*)d.gate; p.gate_stride = d.gate_stride; p.gate_rows = d.gate_rows;
    p.residual = (const bf16_t*)d.residual; p.ldr = d.ldr;
    p.seg_rows = d.seg_rows; p.seg_stride = d.seg_stride; p.seg_off = d.seg_off;
    p.a_seg_rows = d.a_seg_rows; p.a_seg_stride = d.a_seg_stride; p.a_seg_off = d.a_seg_off;
    p.batch = 1; p.splitk = 1;
    p.aux_out = (bf16_t*)d.aux_out; p.aux_in = (const bf16_t*)d.aux_in; p.ld_aux = d.ld_aux;
    p.rms_w = (const bf16_t*)d.rms_weight; p.rms_nheads = d.rms_nheads; p.rms_hpw = d.rms_heads_per_weight;
    p.rms_eps = d.rms_eps; p.rms_rs_out = d.rms_rs_out;
    return p;
}

extern "C" int advgrpo_gemm_grouped(const advgrpo_gemm_desc* descs, int count, void* stream) {
    ADVGRPO_CHECK(descs && (count == 1 || count == 2), "gemm_grouped: need 1 or 2 descriptors");
    if (count == 1) return gemm_bf16(from_desc(descs[0]), as_stream(stream));
    return gemm_bf16_pair(from_desc(descs[0]), from_desc(descs[1]), as_stream(stream));
}

/* fp8 operands (gemm8p_fp8.hip): always the eight-phase kernel, no other tile variant takes them */
extern "C" int advgrpo_gemm_fp8_grouped(const advgrpo_gemm_desc* descs, const advgrpo_fp8_scales* scales, int count, void* stream) {
    ADVGRPO_CHECK(descs && scales && (count == 1 || count == 2), "gemm_fp8_grouped: need 1 or 2 descriptors with their scales");
    GemmParams p[2];
    for (int i = 0; i < count; ++i) {
        p[i] = from_desc(descs[i]);
        p[i].fp8 = 1; p[i].a_scale = scales[i].a_scale; p[i].w_scale = scales[i].w_scale;
        ADVGRPO_CHECK(p[i].out_dtype == ADVGRPO_BF16 && !p[i].aux_in, "gemm_fp8_grouped: bf16 output, no d-activation input");
    }
    if (count == 1) return gemm8p_launch(p[0], as_stream(stream));
    return gemm8p_launch_pair(p[0], p[1], as_stream(stream));
}

/* training variant: + aux pre-activation output / d-activation input, split-K atomic accumulation */
extern "C" int advgrpo_gemm_bf16_train(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc,
                                       int out_dtype, int M, int N, int K, const void* bias, int act, float alpha,
                                       const void* gate, int64_t gate_stride, int gate_rows, const void* residual,
                                       int64_t ldr, void* aux_out, const void* aux_in, int64_t ld_aux, int splitk,
                                       void* stream) {
    GemmParams p{};
    p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.C = C;
    p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.out_dtype = out_dtype;
    p.M = M; p.N = N; p.K = K;
    p.bias = (const bf16_t*)bias; p.act = act; p.alpha = alpha;
    p.gate = (const bf16_t*)gate; p.gate_stride = gate_stride; p.gate_rows = gate_rows;
    p.residual = (const bf16_t*)residual; p.ldr = ldr;
    p.batch = 1; p.splitk = splitk < 1 ? 1 : splitk;
    p.aux_out = (bf16_t*)aux_out; p.aux_in = (const bf16_t*)aux_in; p.ld_aux = ld_aux;
    return gemm_bf16(p, as_stream(stream));
}

/* implicit-GEMM conv3x3 (stride 1, pad 1) over NHWC bf16, optional fused nearest-x2 upsample of the input */
extern "C" int advgrpo_conv3x3_nhwc(const void* x, const void* w, void* y, int out_dtype, int B, int Hout, int Wout,
                                    int Cin, int Cout, int upsample, const void* bias, int act, const void* residual,
                                    const void* zero_page, void* stream) {
    GemmParams p{};
    p.A = (const bf16_t*)x; p.W = (const bf16_t*)w; p.C = y;
    p.lda = Cin; p.ldw = 9 * (int64_t)Cin; p.ldc = Cout; p.out_dtype = out_dtype;
    p.M = B * Hout * Wout; p.N = Cout; p.K = 9 * Cin;
    p.bias = (const bf16_t*)bias; p.act = act; p.alpha = 1.0f;
    p.residual = (const bf16_t*)residual; p.ldr = Cout;
    p.batch = 1;
    p.conv = 1; p.Hout = Hout; p.Wout = Wout; p.Cin = Cin; p.ups = upsample ? 1 : 0;
    p.zero_page = (const bf16_t*)zero_page;
    p.splitk = 1;
    return gemm_bf16(p, as_stream(stream));
}

/* the same convolution over split-bf16 operands (vae.hip: split8): x3 [B, Hin, Win, Cin3] bf16 with Cin3 = 3*C laid out
 * [hi | hi | lo], w3 [Cout, 9*Cin3] with each tap's channels [hi | lo | hi]; bias / residual / y are f32 */
static int conv3x3_two_products(const void* x2, const void* w16, float* y, int B, int Hout, int Wout, int Cin3, int Cout,
                                int upsample, const float* bias, int act, const float* residual, const void* zero_page,
                                float alpha, float* gn_partial, void* stream, int form, void* pair_out = nullptr,
                                float pair_prescale = 1.0f) {
    GemmParams p{};
    p.A = (const bf16_t*)x2; p.W = (const bf16_t*)w16; p.C = y;
    p.lda = Cin3; p.ldw = 3 * (int64_t)Cin3; p.ldc = Cout; p.out_dtype = ADVGRPO_F32;      // weights [Cout, 9 C], C = Cin3 / 3
    p.M = B * Hout * Wout; p.N = Cout; p.K = 3 * Cin3;
    p.bias = (const bf16_t*)bias; p.act = act; p.alpha = alpha;
    p.residual = (const bf16_t*)residual; p.ldr = Cout;
    p.batch = 1;
    p.conv = 1; p.Hout = Hout; p.Wout = Wout; p.Cin = Cin3; p.ups = upsample ? 1 : 0;
    p.zero_page = (const bf16_t*)zero_page;
    p.splitk = 1;
    p.f32_io = 1;
    p.gn_partial = gn_partial;
    p.pair_out = pair_out; p.pair_prescale = pair_prescale;
    ADVGRPO_CHECK(!pair_out || (!gn_partial && Cout % 4 == 0 && (reinterpret_cast<uintptr_t>(pair_out) & 7) == 0),
                  "conv3x3_f16x2: pair output needs Cout %% 4 == 0, 8-byte alignment and no block sums");
    ADVGRPO_CHECK(x2 && w16 && y && zero_page, "conv3x3_f16x2: null pointer");
    ADVGRPO_CHECK(!gn_partial || ((Hout * Wout) % 16 == 0 && Cout % 4 == 0), "conv3x3_f16x2: block sums need 16 | Hout Wout");
    ADVGRPO_CHECK(Cin3 % 192 == 0 && Cout >= 128, "conv3x3_f16x2: Cin3 must be 3 x (a multiple of 64), Cout >= 128 (Cin3=%d Cout=%d)", Cin3, Cout);
    return conv3x3_f16x2_launch(p, as_stream(stream), form);
}

extern "C" int advgrpo_conv3x3_nhwc_f16x2(const void* x2, const void* w16, float* y, int B, int Hout, int Wout, int Cin3, int Cout,
                                          int upsample, const float* bias, int act, const float* residual, const void* zero_page,
                                          float alpha, float* gn_partial, void* stream) {
    return conv3x3_two_products(x2, w16, y, B, Hout, Wout, Cin3, Cout, upsample, bias, act, residual, zero_page, alpha, gn_partial, stream, 0);
}

extern "C" int advgrpo_conv3x3_nhwc_f16x1(const void* x2, const void* w16, float* y, int B, int Hout, int Wout, int Cin3, int Cout,
                                          int upsample, const float* bias, int act, const float* residual, const void* zero_page,
                                          float alpha, float* gn_partial, void* stream) {
    return conv3x3_two_products(x2, w16, y, B, Hout, Wout, Cin3, Cout, upsample, bias, act, residual, zero_page, alpha, gn_partial, stream, 2);
}

extern "C" int advgrpo_conv3x3_nhwc_f16x1_pair(const void* x2, const void* w16, void* pair_out, float pair_prescale, int B, int Hout, int Wout,
                                               int Cin3, int Cout, int upsample, const float* bias, int act, const float* residual,
                                               const void* zero_page, float alpha, void* stream) {
    ADVGRPO_CHECK(pair_out, "conv3x3_f16x1_pair: null output");
    return conv3x3_two_products(x2, w16, reinterpret_cast<float*>(pair_out), B, Hout, Wout, Cin3, Cout, upsample, bias, act, residual, zero_page,
                                alpha, nullptr, stream, 2, pair_out, pair_prescale);
}

extern "C" int advgrpo_conv3x3_nhwc_f16x2_pair(const void* x2, const void* w16, void* pair_out, float pair_prescale, int B, int Hout, int Wout,
                                               int Cin3, int Cout, int upsample, const float* bias, int act, const float* residual,
                                               const void* zero_page, float alpha, void* stream) {
    ADVGRPO_CHECK(pair_out, "conv3x3_f16x2_pair: null output");
    return conv3x3_two_products(x2, w16, reinterpret_cast<float*>(pair_out), B, Hout, Wout, Cin3, Cout, upsample, bias, act, residual, zero_page,
                                alpha, nullptr, stream, 0, pair_out, pair_prescale);
}

extern "C" int advgrpo_conv3x3_nhwc_bf16x2(const void* x2, const void* w16, float* y, int B, int Hout, int Wout, int Cin3, int Cout,
                                           int upsample, const float* bias, int act, const float* residual, const void* zero_page,
                                           float alpha, float* gn_partial, void* stream) {
    return conv3x3_two_products(x2, w16, y, B, Hout, Wout, Cin3, Cout, upsample, bias, act, residual, zero_page, alpha, gn_partial, stream, 1);
}

extern "C" int advgrpo_conv3x3_nhwc_x3(const void* x3, const void* w3, float* y, int B, int Hout, int Wout, int Cin3,
                                       int Cout, int upsample, const float* bias, int act, const float* residual,
                                       const void* zero_page, void* stream) {
    GemmParams p{};
    p.A = (const bf16_t*)x3; p.W = (const bf16_t*)w3; p.C = y;
    p.lda = Cin3; p.ldw = 9 * (int64_t)Cin3; p.ldc = Cout; p.out_dtype = ADVGRPO_F32;
    p.M = B * Hout * Wout; p.N = Cout; p.K = 9 * Cin3;
    p.bias = (const bf16_t*)bias; p.act = act; p.alpha = 1.0f;
    p.residual = (const bf16_t*)residual; p.ldr = Cout;
    p.batch = 1;
    p.conv = 1; p.Hout = Hout; p.Wout = Wout; p.Cin = Cin3; p.ups = upsample ? 1 : 0;
    p.zero_page = (const bf16_t*)zero_page;
    p.splitk = 1;
    p.f32_io = 1;
    ADVGRPO_CHECK(Cin3 % 192 == 0, "conv3x3_x3: Cin3 must be 3 x (a multiple of 64) (Cin3=%d)", Cin3);
    // wide outputs: the dedicated kernel (conv_x3.hip: the four hi / lo pieces staged once per 64 channels, three MFMA
    // products from them); the 3-channel conv_out keeps the 128x64 tile of the tripled-K path
    if (Cout >= 128) return conv3x3_x3_launch(p, as_stream(stream));
    return gemm_bf16(p, as_stream(stream));
}
