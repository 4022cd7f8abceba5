// gemm.hip -- bf16 MFMA GEMM with fused prologue/epilogue for gfx950 (MI355X).
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )         A, W bf16 (K contiguous), f32 accumulate
//
// This one kernel family carries every dense contraction of the hot path: the MMDiT
// QKV / out-proj / MLP / adaLN linears (reference call sites
// sd3_pipeline_with_logprob_fast.py:630-637 and train_sd3_fast_pickscore.py:235-255, which run
// diffusers' SD3Transformer2DModel), the ViT reward towers, and (through the implicit-GEMM A
// loader in conv.hip) the VAE decoder.
//
// CDNA4 mapping
//   * 256 threads = 4 wave64 as 2x2; block tile BM x BN x 64, wave tile (BM/2)x(BN/2) built from
//     v_mfma_f32_16x16x32_bf16 fragments.
//   * HBM -> LDS by global_load_lds_dwordx4 (16 B/lane, no VGPR round trip), double-buffered;
//     the next tile's DMA is issued before the current tile's MFMAs.
//   * LDS image is lane-linear per wave instruction (8 rows x 128 B); bank conflicts of the
//     column-slice ds_read_b128 are removed by an XOR swizzle applied to the per-lane SOURCE
//     chunk (chunk ^= row & 7) and again on the read (guide rule 21).
//   * MFMA is issued with operands swapped (W fragment as A, activation fragment as B) so each
//     lane ends up with 4 CONSECUTIVE output columns of one row: 8-byte bf16x4 stores and
//     vector loads of bias / gate / residual in the fused epilogue.
//   * block ids are remapped so each XCD (private 4 MiB L2) walks a contiguous strip of tiles.
//   * rows >= M / N are clamped on load and masked on store; K must be a multiple of 64.
#include "gemm.hpp"

namespace advgrpo {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

__device__ inline float act_fn(float x, int act) {
    switch (act) {
        case ACT_GELU_TANH: {
            const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
            return 0.5f * x * (1.0f + tanhf(u));
        }
        case ACT_GELU_ERF: return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
        case ACT_SILU: return x / (1.0f + __expf(-x));
        default: return x;
    }
}

// bijective XCD remap: consecutive remapped ids live on one XCD (observed placement b % 8)
__device__ inline int xcd_remap(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int BM, int BN, bool CONV>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const GemmParams p) {
    constexpr int BK = 64;
    constexpr int TM = BM / 2, TN = BN / 2;     // wave tile
    constexpr int FM = TM / 16, FN = TN / 16;   // fragments per wave
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int A_INST = BM / 8 / 4, B_INST = BN / 8 / 4;  // DMA instructions per wave per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    const int swz = xcd_remap(blockIdx.x, nwg);
    const int tile_m = swz / tiles_n, tile_n = swz % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int bz = blockIdx.y;

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
        int r = m0 + (wave + it * 4) * 8 + lrow;
        r = r < p.M ? r : p.M - 1;
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
        int r = n0 + (wave + it * 4) * 8 + lrow;
        r = r < p.N ? r : p.N - 1;
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
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lds_ptr_t)(base + (wave + it * 4) * 1024), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int it = 0; it < A_INST; ++it)
                __builtin_amdgcn_global_load_lds((gptr_t)(a_src[it] + kt * BK),
                                                 (lds_ptr_t)(base + (wave + it * 4) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < B_INST; ++it)
            __builtin_amdgcn_global_load_lds((gptr_t)(b_src[it] + kt * BK),
                                             (lds_ptr_t)(base + A_BYTES + (wave + it * 4) * 1024), 16, 0, 0);
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

    const int nk = p.K / BK;
    stage(0, 0);
    __syncthreads();  // hipcc drains the LDS-DMA (vmcnt 0) before the barrier
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* ta = smem + cur * STAGE + wm * TM * 128;
        const char* tb = smem + cur * STAGE + A_BYTES + wn * TN * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t af[FM], bf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i)
                af[i] = *reinterpret_cast<const bf16x8_t*>(ta + i * 2048 + frag_off[ks]);
#pragma unroll
            for (int j = 0; j < FN; ++j)
                bf[j] = *reinterpret_cast<const bf16x8_t*>(tb + j * 2048 + frag_off[ks]);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- fused epilogue: lane holds C[m][n..n+3], m = frag row (lane&15), n = (lane>>4)*4
    const int mrow = lane & 15, ncol = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * TM + i * 16 + mrow;
        if (m >= p.M) continue;
        // output row mapping (row segments, e.g. image / text tokens into one joint buffer)
        int64_t orow = m;
        int bidx = 0;
        if (p.seg_rows > 0) {
            bidx = m / p.seg_rows;
            orow = (int64_t)bidx * p.seg_stride + p.seg_off + (m - bidx * p.seg_rows);
        }
        const int gb = p.gate_rows > 0 ? m / p.gate_rows : 0;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn * TN + j * 16 + ncol;
            if (n >= p.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            const bool full = (n + 3 < p.N);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (!full && n + r >= p.N) break;
                float y = v[r] * p.alpha;
                if (p.bias) y += bf2f(p.bias[n + r]);
                y = act_fn(y, p.act);
                if (p.gate) y *= bf2f(p.gate[(int64_t)(bz * p.gate_batch_stride) + (int64_t)gb * p.gate_stride + n + r]);
                if (p.residual) y += bf2f(p.residual[(int64_t)bz * p.strideR + orow * p.ldr + n + r]);
                v[r] = y;
            }
            const int64_t o = (int64_t)bz * p.strideC + orow * p.ldc + n;
            if (p.out_dtype == ADVGRPO_BF16) {
                bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
                if (full && ((o & 3) == 0)) {
                    uint2 pk;
                    pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
                    pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
                    *reinterpret_cast<uint2*>(C + o) = pk;
                } else {
                    for (int r = 0; r < 4 && n + r < p.N; ++r) C[o + r] = f2bf(v[r]);
                }
            } else {
                float* C = reinterpret_cast<float*>(p.C);
                if (full && ((o & 3) == 0)) {
                    *reinterpret_cast<float4*>(C + o) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    for (int r = 0; r < 4 && n + r < p.N; ++r) C[o + r] = v[r];
                }
            }
        }
    }
}

template <int BM, int BN, bool CONV>
static int launch(const GemmParams& p, hipStream_t s) {
    constexpr int lds = 2 * (BM + BN) * 64 * 2;
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<BM, BN, CONV>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, CONV>), dim3(tiles, p.batch), dim3(256), lds, s, p);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

// tile variant the dispatcher picks: 0 = 128x128, 1 = 128x64, 2 = 64x128 (+4 for the conv loader)
static int gemm_variant(int M, int N, int batch, int conv) {
    const int64_t t128 = (int64_t)((M + 127) / 128) * ((N + 127) / 128) * batch;
    if (conv) return 4 + (N <= 64 ? 1 : 0);
    if (N <= 64) return 1;
    if (t128 >= 256 || M > 2048) return 0;
    return 2;
}

int gemm_bf16(const GemmParams& p, hipStream_t s) {
    ADVGRPO_CHECK(p.A && p.W && p.C, "gemm: null operand");
    ADVGRPO_CHECK(p.M > 0 && p.N > 0 && p.K > 0 && p.K % 64 == 0, "gemm: need M,N>0 and K %% 64 == 0 (M=%d N=%d K=%d)",
                  p.M, p.N, p.K);
    ADVGRPO_CHECK(p.lda % 8 == 0 && p.ldw % 8 == 0, "gemm: lda/ldw must be multiples of 8 elements (16 B rows)");
    ADVGRPO_CHECK(p.out_dtype == ADVGRPO_BF16 || p.out_dtype == ADVGRPO_F32, "gemm: bad out dtype");
    ADVGRPO_CHECK(p.batch >= 1, "gemm: batch must be >= 1");
    // tile choice: big tiles when they still fill the 256 CUs, smaller ones for skinny problems
    const int variant = gemm_variant(p.M, p.N, p.batch, p.conv);
    if (p.conv) {
        ADVGRPO_CHECK(p.Cin % 64 == 0 && p.K == 9 * p.Cin && p.zero_page && p.batch == 1,
                      "conv3x3: need Cin %% 64 == 0, K == 9*Cin, a zero page and batch 1 (Cin=%d K=%d)", p.Cin, p.K);
        ADVGRPO_CHECK((p.Hout % (1 << p.ups)) == 0 && (p.Wout % (1 << p.ups)) == 0, "conv3x3: bad upsample shape");
        if (variant == 5) return launch<128, 64, true>(p, s);
        return launch<128, 128, true>(p, s);
    }
    if (variant == 1) return launch<128, 64, false>(p, s);
    if (variant == 0) return launch<128, 128, false>(p, s);
    return launch<64, 128, false>(p, s);
}

}  // namespace advgrpo

using namespace advgrpo;

extern "C" int advgrpo_gemm_variant(int M, int N, int batch, int conv) { return gemm_variant(M, N, batch < 1 ? 1 : batch, conv); }

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
    return gemm_bf16(p, as_stream(stream));
}
