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
// k-tile stages the 64-wide pieces ONCE and issues the three MFMA products from them, and the three taps of one kernel row
// share ONE pixel stage:
//   * operands are read where the round-2 layout has them (activations [.., 3C] = [hi | hi | lo], weights per tap
//     [hi | lo | hi]): hi at +0, x_lo at +2C, w_lo at +C; the middle third of an activation row is never read
//   * workgroup 512 threads = 8 waves (4 x 2), tile 192 pixels x 128 channels x 64, wave tile 48 x 64,
//     v_mfma_f32_16x16x32_bf16 with swapped operands (a lane owns 4 consecutive output channels of one pixel: float4 stores
//     in the f32 epilogue); 72 MFMAs per wave and k-tile behind 28 ds_read_b128
//   * k order: group (dy, 64-channel slice) -> taps dx = -1, 0, +1.  A group stages the tile's 192 consecutive output
//     pixels displaced by dy PLUS one pixel on each side (x_hi, x_lo: 2 x 25 KiB, LDS row j = pixel m0 - 1 + j); tap dx
//     reads its fragments one row up or down, and a pixel in the first / last column of its image row (whose shifted row
//     holds the neighbouring image row) gets a zero fragment instead.  One pixel stage per THREE k-tiles: 43 % fewer
//     L2 -> LDS bytes than a stage per tap.
//   * weights (w_hi, w_lo of one tap: 2 x 16 KiB) travel through a ring of three stages -- the tap index is the slot --
//     requested two k-tiles ahead; the single pixel stage is re-requested on the last tap of a group, behind a barrier
//     that follows that k-tile's fragment reads.  LDS: 50 + 96 = 146 KiB, one workgroup per CU.
//   * raw s_barrier and hand-counted s_waitcnt vmcnt(4) (the youngest weight stage stays in flight): __syncthreads() drains
//     every LDS-DMA.  HBM/L2 -> LDS by global_load_lds_dwordx4, lane-linear 1 KiB images, chunk ^ (row & 7) source swizzle.
// Measured (8 x 512^2 decode, same box): round-2 form 83.9 ms; one stage per tap, 128-pixel tile 73.6; 192-pixel tile 69.9
// (the kernel pays a fixed latency per k-tile -- DMA round trip, fragment reads in front of the first MFMA, barrier -- so
// more MFMAs per k-tile pay, fewer bytes alone do not: the shared pixel stage with two weight stages gave 0.6 ms); weight ring
// 66.0; fragment reads issued and waited for by hand, seven at a time in the order the products consume them, 65.5 (hipcc
// had put all 28 reads of a k-tile in front of its first MFMA).  What is left: both waves of a SIMD run the same phase at the
// same time, and a one-workgroup-per-CU, non-persistent tile exposes its prologue and epilogue.
#include "gemm_device.hpp"
#include <cstdlib>

namespace advgrpo {

namespace {

typedef __attribute__((address_space(3))) void* x3_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* x3_gptr_t;

constexpr int X3_BM = 192, X3_BN = 128, X3_BK = 64;
#ifndef X3_GRID_M
#define X3_GRID_M 4
#define X3_GRID_N 2
#endif
constexpr int X3_XINST = (X3_BM + 2 + 7) / 8;        // 25 DMA instructions (8 pixel rows each) per pixel piece: pixels m0-1 .. m0+198
constexpr int X3_XPIECE = X3_XINST * 1024;           // 25 KiB: the tile's pixels and one halo pixel on each side, 64 channels
constexpr int X3_XSTAGE = 2 * X3_XPIECE;             // x_hi, x_lo: ONE stage, reloaded once per three k-tiles
constexpr int X3_WPIECE = X3_BN * X3_BK * 2;         // 16 KiB: 128 output channels x 64 input channels of one tap
constexpr int X3_WSTAGE = 2 * X3_WPIECE;             // w_hi, w_lo
constexpr int X3_LDS = X3_XSTAGE + 3 * X3_WSTAGE;    // 146 KiB: a ring of three weight stages (the tap index IS the ring slot)

// Fragment reads are issued and waited for by hand: hipcc puts s_waitcnt lgkmcnt(0) in front of the first MFMA that consumes an
// LDS read, i.e. every read of a k-tile in front of its first MFMA.  The wait statement takes the registers it releases as
// in/out operands, so nothing that uses them can be scheduled above it (LDS returns data in order; lgkmcnt has 4 bits: at
// most 15 reads are left in flight).
template <int OFF>
__device__ __forceinline__ void x3_lds_read(bf16x8_t& d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int CNT>
__device__ __forceinline__ void x3_wait_reads() {
    asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(CNT));
}
// (volatile statements keep their order: a register tied here cannot be used above the wait in front of it)
template <int N>
__device__ __forceinline__ void x3_tie(bf16x8_t (&a)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(a[i]));
}

// F16 (round 4, "f16x2"): the same kernel for weights that are EXACT in fp16 -- the released SD3 / SD3.5 VAE is stored in fp16
// and only upcast by vae.to(torch.float32) (TP:481), so every weight is one 16-bit piece.  Activations then travel as fp16
// hi = f16(v), lo = f16(v - hi) (22 significant bits: more than the 16 of the bf16 pair) in the same places of the same
// [hi | unwritten | lo] rows, weights as ONE fp16 piece per tap ([Cout, 9 C], no hi / lo), and a product is x_hi w + x_lo w on
// v_mfma_f32_16x16x32_f16: two MFMA products instead of three, half the weight bytes through L2 -> LDS (two thirds of this
// kernel's bytes are weights), the same pipeline.  p.alpha undoes the power-of-two pre-scale of un-normalised inputs (fp16 range).
// PIECE = 2 ("bf16x2"): the same two-product form for weights that are EXACT in bf16 (the released Qwen-Image VAE is a bf16
// checkpoint): activations as the bf16 pair of the three-product path in the same [hi | unwritten | lo] rows (16 significant bits),
// weights as ONE bf16 piece per tap, x_hi w + x_lo w on v_mfma_f32_16x16x32_bf16 -- the three-product form would multiply by a
// weight "lo" that is identically zero.
// PIECE = 3 ("f16x1", round 6): the TF32-CLASS form of f16x2 -- ONE fp16 product per f32 product: the activation's hi half only (11 significant
// bits, what a TF32 operand keeps; the weights are exact), f32 accumulation, f32 between kernels.  The reference runs with allow_tf32 = True
// (config/base.py:22-23, TP:537-538), i.e. its fp32 convolutions round both operands to 10 explicit mantissa bits on Ampere+; this is that
// arithmetic class on the fp16 MFMA.  Same pipeline; the lo pieces are neither requested nor read nor multiplied.  Priced as a leg of the bench
// line, never the default.
// DBG (experiments build only, scripts/probes/x3_decompose.sh): 1 = no DMA after the prologue, 2 = no MFMAs, 3 = fragments read
// once -- WRONG results, used to price the three activities of the k loop against each other
// BN (round 6): output channels per tile.  256 for the one-piece-weight forms where Cout is a multiple of 256 and the launch is many rounds of
// tiles deep: a wave tile of 48 x 128 (96 MFMAs per wave behind one barrier per k-tile instead of 48; the same pixel stage feeds twice the
// channels: 26 % fewer L2 -> LDS bytes and 30 % fewer fragment bytes per flop); the weight ring then holds three one-piece stages of 32 KiB
// (the same 146 KiB).  Same k order and product order per output element as BN = 128: the same bits.
template <int X3_WM, int X3_WN, int DBG = 0, int PIECE = 0, int BN = X3_BN>
__global__ __launch_bounds__(64 * X3_WM * X3_WN, 1) void conv3x3_x3_kernel(const GemmParams p) {
    constexpr bool F16 = PIECE != 0;                  // one-piece weights, two products (1: fp16 pieces; 2: bf16 pieces; 3: fp16 pieces, hi product only)
    constexpr bool X1 = PIECE == 3;
    constexpr int WPI = F16 ? 1 : 2;                  // weight pieces per tap (DMA instructions per 8 output channels)
    constexpr int NW = X3_WM * X3_WN, TM = X3_BM / X3_WM, TN = BN / X3_WN, FM = TM / 16, FN = TN / 16;
    static_assert(BN == X3_BN || PIECE != 0, "the wide tile exists for the one-piece-weight forms");
    constexpr int WPIECE = BN * X3_BK * 2;            // one weight piece of one tap: BN output channels x 64 input channels
    constexpr int WSTAGE = (BN == X3_BN ? 2 : 1) * WPIECE;
    constexpr int XI = (X3_XINST + NW - 1) / NW;     // pixel-piece DMA instructions per wave (the last one on some waves only)
    constexpr int WI = BN / 8 / NW;                  // weight-piece DMA instructions per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / X3_WN, wn = wave % X3_WN;
    const int C = p.Cin / 3;                          // logical channels; pixel pitch of the activations = 3C elements

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + X3_BM - 1) / X3_BM;
    const int swz = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    int tile_m, tile_n;
    tile_coords(swz, tiles_m, tiles_n, 4, tile_m, tile_n);
    const int m0 = tile_m * X3_BM, n0 = tile_n * BN;

    // ---- per-lane DMA sources.  LDS row j of a pixel piece holds output pixel m0 - 1 + j displaced by the group's dy
    const int lrow = lane >> 3, schunk = (lane & 7) ^ lrow;
    const int hw = p.Hout * p.Wout;
    int a_y[XI], a_x[XI];
    int64_t a_img[XI];
#pragma unroll
    for (int it = 0; it < XI; ++it) {
        int q = m0 - 1 + (wave + it * NW) * 8 + lrow;
        q = q < 0 ? 0 : (q < p.M ? q : p.M - 1);
        const int bi = q / hw, rem = q - bi * hw;
        a_y[it] = rem / p.Wout;
        a_x[it] = rem - a_y[it] * p.Wout;
        a_img[it] = (int64_t)bi * (p.Hout >> p.ups) * (p.Wout >> p.ups) * p.Cin;
    }
    const bf16_t* w_src[WI];
#pragma unroll
    for (int it = 0; it < WI; ++it) {
        int n = n0 + (wave + it * NW) * 8 + lrow;
        n = n < p.N ? n : p.N - 1;
        w_src[it] = p.W + (int64_t)n * p.ldw + schunk * 8;
    }
    const int win = p.Wout >> p.ups;
    const int slices = C / X3_BK;
    // group g = (dy, 64-channel slice): one pixel stage, three k-tiles (dx = -1, 0, +1) that each stage their own weights
    auto stage_x = [&](int g) {
        char* base = smem;
        const int dyi = g / slices, c0 = (g - dyi * slices) * X3_BK;
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            if (wave + it * NW >= X3_XINST) break;    // wave-uniform
            const int yy = a_y[it] + dyi - 1;
            const bool ok = (unsigned)yy < (unsigned)p.Hout;
            const bf16_t* hi = ok ? p.A + a_img[it] + ((int64_t)(yy >> p.ups) * win + (a_x[it] >> p.ups)) * p.Cin + c0 + schunk * 8
                                  : p.zero_page + schunk * 8;
            const bf16_t* lo = ok ? hi + 2 * C : hi;                          // [hi | hi | lo]
            char* dst = base + (wave + it * NW) * 1024;
            __builtin_amdgcn_global_load_lds((x3_gptr_t)hi, (x3_lds_ptr_t)dst, 16, 0, 0);
            // (f16x1 reads the hi pieces only.  Every wait of the k loop allows the WPI * WI most recent requests -- a weight stage -- to be outstanding
            // and nothing older, so leaving the lo requests out changes no count)
            if constexpr (!X1) __builtin_amdgcn_global_load_lds((x3_gptr_t)lo, (x3_lds_ptr_t)(dst + X3_XPIECE), 16, 0, 0);
        }
    };
    auto stage_w = [&](int buf, int g, int dxi) {
        char* base = smem + X3_XSTAGE + buf * WSTAGE;
        const int dyi = g / slices, c0 = (g - dyi * slices) * X3_BK;
        const int tap = dyi * 3 + dxi;
#pragma unroll
        for (int it = 0; it < WI; ++it) {
            char* dst = base + (wave + it * NW) * 1024;
            const bf16_t* wh = w_src[it] + (int64_t)tap * (F16 ? C : p.Cin) + c0;   // per tap [hi | lo | hi]; F16: one piece per tap
            __builtin_amdgcn_global_load_lds((x3_gptr_t)wh, (x3_lds_ptr_t)dst, 16, 0, 0);
            if constexpr (!F16) __builtin_amdgcn_global_load_lds((x3_gptr_t)(wh + C), (x3_lds_ptr_t)(dst + WPIECE), 16, 0, 0);
        }
    };

    // fragment offsets: a lane's pixel row of tap dx is LDS row (lane & 15) + 1 + dx of its 16-row block
    int off_x[3][2], off_w[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        off_w[ks] = (lane & 15) * 128 + (((ks * 4 + (lane >> 4)) ^ (lane & 7)) << 4);
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi) {
            const int r = (lane & 15) + dxi;
            off_x[dxi][ks] = r * 128 + (((ks * 4 + (lane >> 4)) ^ (r & 7)) << 4);
        }
    }
    // a pixel in the first / last column of its image row has no left / right neighbour: the shifted LDS row holds the
    // neighbouring image row's pixel there, so the fragment is zeroed instead (bit i: block i of this wave)
    uint32_t no_left = 0, no_right = 0;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        int m = m0 + wm * TM + i * 16 + (lane & 15);
        m = m < p.M ? m : p.M - 1;
        const int x = (m % hw) % p.Wout;
        no_left |= (x == 0 ? 1u : 0u) << i;
        no_right |= (x == p.Wout - 1 ? 1u : 0u) << i;
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Waits are counted by hand (raw s_barrier: __syncthreads() would drain every LDS-DMA): a wave has 2 * WI = 4 weight
    // instructions per k-tile in flight behind whatever the next k-tile reads.
    const int ngroups = 3 * slices, nk = 3 * ngroups;
    stage_x(0);
    stage_w(0, 0, 0);
    stage_w(1, 0, 1);
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(WPI * WI) : "memory");
    __builtin_amdgcn_s_barrier();
    static_assert(FM + FN <= 15 && (F16 || 3 * FN <= 15) && (!F16 || 2 * FM + FN <= 15), "lgkmcnt counts at most 15 reads in flight");
    auto product = [&](const bf16x8_t (&a)[FM], const bf16x8_t (&b)[FN]) __attribute__((always_inline)) {
        if constexpr (DBG != 2) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    if constexpr (PIECE == 1 || PIECE == 3) {
                        typedef _Float16 x3_f16x8 __attribute__((ext_vector_type(8)));
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(x3_f16x8, b[j]), __builtin_bit_cast(x3_f16x8, a[i]),
                                                                           acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
                    }
                }
        } else {
            acc[0][0] += __builtin_bit_cast(f32x4, a[0]) + __builtin_bit_cast(f32x4, b[FN - 1]);
        }
    };
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(smem);
    uint32_t xa[3][2], wa[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        wa[ks] = lds0 + X3_XSTAGE + wn * TN * 128 + off_w[ks];
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi) xa[dxi][ks] = lds0 + wm * TM * 128 + off_x[dxi][ks];
    }
    for (int g = 0; g < ngroups; ++g) {
        static_for<3>([&](auto dxi_c) {
            constexpr int dxi = decltype(dxi_c)::value;      // k-tile kt = 3g + dxi reads weight slot dxi
            const int kt = 3 * g + dxi;
            const uint32_t wb0 = wa[0] + dxi * WSTAGE, wb1 = wa[1] + dxi * WSTAGE;
            auto read_x = [&](uint32_t a, bf16x8_t (&h)[FM], bf16x8_t (&l)[FM], bool hi, bool lo) __attribute__((always_inline)) {
                if (hi) static_for<FM>([&](auto i) { x3_lds_read<decltype(i)::value * 2048>(h[decltype(i)::value], a); });
                if (lo) static_for<FM>([&](auto i) { x3_lds_read<X3_XPIECE + decltype(i)::value * 2048>(l[decltype(i)::value], a); });
            };
            auto read_wh = [&](uint32_t a, bf16x8_t (&b)[FN]) __attribute__((always_inline)) {
                static_for<FN>([&](auto j) { x3_lds_read<decltype(j)::value * 2048>(b[decltype(j)::value], a); });
            };
            auto read_wl = [&](uint32_t a, bf16x8_t (&b)[FN]) __attribute__((always_inline)) {
                static_for<FN>([&](auto j) { x3_lds_read<WPIECE + decltype(j)::value * 2048>(b[decltype(j)::value], a); });
            };
            auto mask = [&](bf16x8_t (&f)[FM]) __attribute__((always_inline)) {
                if constexpr (dxi != 1) {
                    const uint32_t bits = dxi == 0 ? no_left : no_right;
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        const bool z = (bits >> i) & 1u;
                        const bf16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
                        f[i] = z ? zero : f[i];
                    }
                }
            };
            bf16x8_t ah0[FM], al0[FM], bh0[FN], bl0[FN], ah1[FM], al1[FM], bh1[FN], bl1[FN];
            const bool more = kt + 2 < nk;                   // (then group g + 1 exists as well when dxi == 2)
            const bool rd = DBG != 3 || kt == 0;
            if constexpr (X1 && dxi < 2) {
                // hi product only: both 32-deep steps' x_hi and w fragments requested up front
                if (rd) { read_x(xa[dxi][0], ah0, al0, true, false); read_wh(wb0, bh0); read_x(xa[dxi][1], ah1, al1, true, false); read_wh(wb1, bh1); }
                __builtin_amdgcn_sched_barrier(0);
                if (more && DBG != 1) stage_w((dxi + 2) % 3, dxi == 0 ? g : g + 1, (dxi + 2) % 3);
                __builtin_amdgcn_sched_barrier(0);
                if (rd) { x3_wait_reads<FM + FN>(); x3_tie(ah0); x3_tie(bh0); }
                mask(ah0);
                product(ah0, bh0);
                __builtin_amdgcn_sched_barrier(0);
                if (rd) { x3_wait_reads<0>(); x3_tie(ah1); x3_tie(bh1); }
                mask(ah1);
                product(ah1, bh1);
            } else if constexpr (X1) {
                if (rd) { read_x(xa[dxi][0], ah0, al0, true, false); read_x(xa[dxi][1], ah1, al1, true, false); }
                if (rd) { x3_wait_reads<0>(); x3_tie(ah0); x3_tie(ah1); }
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (g + 1 < ngroups && DBG != 1) {
                    stage_x(g + 1);
                    stage_w(1, g + 1, 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (rd) { read_wh(wb0, bh0); read_wh(wb1, bh1); }
                if (rd) { x3_wait_reads<FN>(); x3_tie(bh0); }
                mask(ah0);
                product(ah0, bh0);
                __builtin_amdgcn_sched_barrier(0);
                if (rd) { x3_wait_reads<0>(); x3_tie(bh1); }
                mask(ah1);
                product(ah1, bh1);
            } else if constexpr (F16 && dxi < 2) {
                // one weight piece: per 32-deep step x_hi (FM reads), w (FN), x_lo (FM); products x_hi w, x_lo w
                if (rd) { read_x(xa[dxi][0], ah0, al0, true, false); read_wh(wb0, bh0); read_x(xa[dxi][0], ah0, al0, false, true); }
                __builtin_amdgcn_sched_barrier(0);
                if (more && DBG != 1) stage_w((dxi + 2) % 3, dxi == 0 ? g : g + 1, (dxi + 2) % 3);
                __builtin_amdgcn_sched_barrier(0);
                if (rd) { x3_wait_reads<FM>(); x3_tie(ah0); x3_tie(bh0); }
                mask(ah0);
                product(ah0, bh0);
                __builtin_amdgcn_sched_barrier(0);
                if (rd) { read_x(xa[dxi][1], ah1, al1, true, false); read_wh(wb1, bh1); }
                if (rd) { x3_wait_reads<FM + FN>(); x3_tie(al0); }
                mask(al0);
                product(al0, bh0);
                __builtin_amdgcn_sched_barrier(0);
                if (rd) { read_x(xa[dxi][1], ah1, al1, false, true); }
                if (rd) { x3_wait_reads<FM>(); x3_tie(ah1); x3_tie(bh1); }
                mask(ah1);
                product(ah1, bh1);
                __builtin_amdgcn_sched_barrier(0);
                if (rd) { x3_wait_reads<0>(); x3_tie(al1); }
                mask(al1);
                product(al1, bh1);
            } else if constexpr (F16) {
                // last tap of a group (see below): all twelve pixel reads, barrier, the next group's pixel stage, then the weights
                if (rd) { read_x(xa[dxi][0], ah0, al0, true, true); read_x(xa[dxi][1], ah1, al1, true, true); }
                if (rd) { x3_wait_reads<0>(); x3_tie(ah0); x3_tie(al0); x3_tie(ah1); x3_tie(al1); }
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (g + 1 < ngroups && DBG != 1) {
                    stage_x(g + 1);
                    stage_w(1, g + 1, 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (rd) { read_wh(wb0, bh0); read_wh(wb1, bh1); }
                if (rd) { x3_wait_reads<FN>(); x3_tie(bh0); }
                mask(ah0);
                product(ah0, bh0);
                mask(al0);
                product(al0, bh0);
                __builtin_amdgcn_sched_barrier(0);
                if (rd) { x3_wait_reads<0>(); x3_tie(bh1); }
                mask(ah1);
                product(ah1, bh1);
                mask(al1);
                product(al1, bh1);
            } else if constexpr (dxi < 2) {
                // reads in the order the products need them, seven at a time, the next seven requested before the wait that
                // releases the previous ones; the weights of k-tile kt + 2 go into the slot k-tile kt - 1 read
                if (rd) { read_x(xa[dxi][0], ah0, al0, true, false); read_wl(wb0, bl0); }
                if (rd) { read_x(xa[dxi][0], ah0, al0, false, true); read_wh(wb0, bh0); }
                __builtin_amdgcn_sched_barrier(0);
                if (more && DBG != 1) stage_w((dxi + 2) % 3, dxi == 0 ? g : g + 1, (dxi + 2) % 3);
                __builtin_amdgcn_sched_barrier(0);
                if (rd) { x3_wait_reads<FM + FN>(); x3_tie(ah0); x3_tie(bl0); }
                mask(ah0);
                product(ah0, bl0);
                __builtin_amdgcn_sched_barrier(0);
                if (rd) { read_x(xa[dxi][1], ah1, al1, true, false); read_wl(wb1, bl1); }
                if (rd) { x3_wait_reads<FM + FN>(); x3_tie(al0); x3_tie(bh0); }
                mask(al0);
                product(al0, bh0);
                product(ah0, bh0);
                __builtin_amdgcn_sched_barrier(0);
                if (rd) { read_x(xa[dxi][1], ah1, al1, false, true); read_wh(wb1, bh1); }
                if (rd) { x3_wait_reads<FM + FN>(); x3_tie(ah1); x3_tie(bl1); }
                mask(ah1);
                product(ah1, bl1);
                __builtin_amdgcn_sched_barrier(0);
                if (rd) { x3_wait_reads<0>(); x3_tie(al1); x3_tie(bh1); }
                mask(al1);
                product(al1, bh1);
                product(ah1, bh1);
            } else {
                // last tap of a group: the twelve pixel reads first -- once they have arrived on every wave (barrier) the
                // single pixel stage is free for the next group's pixels, requested BEFORE this k-tile's weight prefetch so
                // that the closing vmcnt(4) covers them
                if (rd) { read_x(xa[dxi][0], ah0, al0, true, true); read_x(xa[dxi][1], ah1, al1, true, true); }
                if (rd) read_wl(wb0, bl0);                    // (more than 15 reads: the hardware holds the rest back until the first return)
                if (rd) { x3_wait_reads<FN>(); x3_tie(ah0); x3_tie(al0); x3_tie(ah1); x3_tie(al1); }
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (g + 1 < ngroups && DBG != 1) {
                    stage_x(g + 1);
                    stage_w(1, g + 1, 1);                    // k-tile kt + 2 = (g + 1, tap 1)
                }
                __builtin_amdgcn_sched_barrier(0);
                if (rd) { read_wh(wb0, bh0); read_wl(wb1, bl1); read_wh(wb1, bh1); }
                if (rd) { x3_wait_reads<3 * FN>(); x3_tie(bl0); }
                mask(ah0);
                product(ah0, bl0);
                __builtin_amdgcn_sched_barrier(0);
                if (rd) { x3_wait_reads<2 * FN>(); x3_tie(bh0); }
                mask(al0);
                product(al0, bh0);
                product(ah0, bh0);
                __builtin_amdgcn_sched_barrier(0);
                if (rd) { x3_wait_reads<FN>(); x3_tie(bl1); }
                mask(ah1);
                product(ah1, bl1);
                __builtin_amdgcn_sched_barrier(0);
                if (rd) { x3_wait_reads<0>(); x3_tie(bh1); }
                mask(al1);
                product(al1, bh1);
                product(ah1, bh1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(WPI * WI) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        });
    }
    if constexpr (F16) {
        if (p.gn_partial) {     // (uniform) the sums of the GroupNorm that reads this output: see GemmParams::gn_partial
            float st[FM][FN][2];
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) { st[i][j][0] = 0.f; st[i][j][1] = 0.f; }
            gemm_epilogue_f32io<FM, FN, TM, TN, true>(p, acc, m0, n0, wm, wn, lane, st);
            // the 16 pixels of a fragment sit in the 16 lanes that share lane >> 4: a fixed xor tree, the same bits wherever the
            // image stands in the batch; every (16-pixel block, 4-channel chunk) belongs to exactly one wave: no LDS, no atomics
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    float sv = st[i][j][0], qv = st[i][j][1];
                    sv += __shfl_xor(sv, 1, 64); qv += __shfl_xor(qv, 1, 64);
                    sv += __shfl_xor(sv, 2, 64); qv += __shfl_xor(qv, 2, 64);
                    sv += __shfl_xor(sv, 4, 64); qv += __shfl_xor(qv, 4, 64);
                    sv += __shfl_xor(sv, 8, 64); qv += __shfl_xor(qv, 8, 64);
                    const int m = m0 + wm * TM + i * 16, n = n0 + wn * TN + j * 16 + (lane >> 4) * 4;
                    if ((lane & 15) == 0 && m < p.M && n < p.N)
                        *reinterpret_cast<float2*>(p.gn_partial + ((int64_t)(m >> 4) * (p.N >> 2) + (n >> 2)) * 2) = float2{sv, qv};
                }
            return;
        }
    }
    gemm_epilogue_f32io<FM, FN, TM, TN>(p, acc, m0, n0, wm, wn, lane);
}

// fewest wide (192 x 256) tiles for which the wide form is dispatched: 4 rounds of the 256 CUs (below that the coarser tile count costs more
// in its ragged last round than the wave tile gains; measured at 256 / 1024 / 2048, profiles/r6_conv_wide_tile.txt).
// ADVGRPO_X3_WIDE_MIN (experiments build): another threshold; a huge one = never.
int64_t x3_wide_min_tiles() {
#ifdef ADVGRPO_EXPERIMENTS
    static const int64_t v = getenv("ADVGRPO_X3_WIDE_MIN") ? atoll(getenv("ADVGRPO_X3_WIDE_MIN")) : 1024;
    return v;
#else
    return 1024;
#endif
}

}  // namespace

// fp16 pair activations x one-piece fp16 weights: p as filled by advgrpo_conv3x3_nhwc_f16x2 (Cin = 3C, lda = 3C, ldw = 9C, f32_io)
int conv3x3_f16x2_launch(const GemmParams& p, hipStream_t s, int form /* 0: f16x2, 1: bf16x2, 2: f16x1 */) {
    ADVGRPO_CHECK(p.conv && p.f32_io && p.Cin % 192 == 0 && p.zero_page && p.batch == 1 && p.splitk == 1, "conv3x3_f16x2: bad parameter block");
    constexpr int WM = X3_GRID_M, WN = X3_GRID_N, WIDE = 2 * X3_BN;
    static bool attr_set = false;
    if (!attr_set) {
        ADVGRPO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_x3_kernel<WM, WN, 0, 1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS) == hipSuccess &&
                      hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_x3_kernel<WM, WN, 0, 2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS) == hipSuccess &&
                      hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_x3_kernel<WM, WN, 0, 3>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS) == hipSuccess &&
                      hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_x3_kernel<WM, WN, 0, 1, WIDE>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS) == hipSuccess &&
                      hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_x3_kernel<WM, WN, 0, 2, WIDE>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS) == hipSuccess &&
                      hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_x3_kernel<WM, WN, 0, 3, WIDE>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS) == hipSuccess,
                      "conv3x3_f16x2: %d bytes of LDS refused", X3_LDS);
        attr_set = true;
    }
    static_assert(X3_XSTAGE + 3 * WIDE * X3_BK * 2 <= X3_LDS, "three one-piece weight stages of the wide tile fit the same LDS");
    // the wide tile where the launch stays several rounds deep (profiles/r6_conv_wide_tile.txt): Cout % 256 == 0 and >= x3_wide_min_tiles() wide tiles
    const int64_t tiles_m = (p.M + X3_BM - 1) / X3_BM;
    bool wide = p.N % WIDE == 0 && tiles_m * (p.N / WIDE) >= x3_wide_min_tiles();
    const dim3 block(64 * WM * WN);
    if (wide) {
        const dim3 grid((unsigned)(tiles_m * (p.N / WIDE)));
        if (form == 1) hipLaunchKernelGGL((conv3x3_x3_kernel<WM, WN, 0, 2, WIDE>), grid, block, X3_LDS, s, p);
        else if (form == 2) hipLaunchKernelGGL((conv3x3_x3_kernel<WM, WN, 0, 3, WIDE>), grid, block, X3_LDS, s, p);
        else hipLaunchKernelGGL((conv3x3_x3_kernel<WM, WN, 0, 1, WIDE>), grid, block, X3_LDS, s, p);
        ADVGRPO_LAUNCH_CHECK();
        return 0;
    }
    const int tiles = ((p.M + X3_BM - 1) / X3_BM) * ((p.N + X3_BN - 1) / X3_BN);
    if (form == 1) hipLaunchKernelGGL((conv3x3_x3_kernel<WM, WN, 0, 2>), dim3(tiles), block, X3_LDS, s, p);
    else if (form == 2) hipLaunchKernelGGL((conv3x3_x3_kernel<WM, WN, 0, 3>), dim3(tiles), block, X3_LDS, s, p);
    else hipLaunchKernelGGL((conv3x3_x3_kernel<WM, WN, 0, 1>), dim3(tiles), block, X3_LDS, s, p);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

// p as filled by advgrpo_conv3x3_nhwc_x3 (gemm.hip): Cin = 3C, K = 9 * 3C, lda = 3C, ldw = 27C, f32_io
int conv3x3_x3_launch(const GemmParams& p, hipStream_t s) {
    ADVGRPO_CHECK(p.conv && p.f32_io && p.Cin % 192 == 0 && p.zero_page && p.batch == 1 && p.splitk == 1,
                  "conv3x3_x3: bad parameter block");
    constexpr int WM = X3_GRID_M, WN = X3_GRID_N;
    static bool attr_set = false;
    if (!attr_set) {
        ADVGRPO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_x3_kernel<WM, WN, 0>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS) == hipSuccess,
                      "conv3x3_x3: %d bytes of LDS refused", X3_LDS);
        attr_set = true;
    }
    const int tiles = ((p.M + X3_BM - 1) / X3_BM) * ((p.N + X3_BN - 1) / X3_BN);
    const dim3 block(64 * WM * WN);
#ifdef ADVGRPO_EXPERIMENTS
    static const int dbg = getenv("ADVGRPO_X3_DBG") ? atoi(getenv("ADVGRPO_X3_DBG")) : 0;
    if (dbg) {
        const void* k = dbg == 1 ? (const void*)conv3x3_x3_kernel<WM, WN, 1> : dbg == 2 ? (const void*)conv3x3_x3_kernel<WM, WN, 2>
                                                                                        : (const void*)conv3x3_x3_kernel<WM, WN, 3>;
        (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
        if (dbg == 1) hipLaunchKernelGGL((conv3x3_x3_kernel<WM, WN, 1>), dim3(tiles), block, X3_LDS, s, p);
        else if (dbg == 2) hipLaunchKernelGGL((conv3x3_x3_kernel<WM, WN, 2>), dim3(tiles), block, X3_LDS, s, p);
        else hipLaunchKernelGGL((conv3x3_x3_kernel<WM, WN, 3>), dim3(tiles), block, X3_LDS, s, p);
        ADVGRPO_LAUNCH_CHECK();
        return 0;
    }
#endif
    hipLaunchKernelGGL((conv3x3_x3_kernel<WM, WN, 0>), dim3(tiles), block, X3_LDS, s, p);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

}  // namespace advgrpo
