// attention_bwd_d128.hip -- attention backward for head dim 128 on gfx950 (bf16 in/out, f32 math): the joint attention of the
// Qwen-Image MMDiT (BASELINE config 5) in the G-step.
//
// Replaces autograd of F.scaled_dot_product_attention behind the transformer call of compute_log_prob
// (scripts/train_sd3_fast_pickscore.py:233-267, loss.backward() :1165) for the Qwen-Image model the reference names but does not
// ship (README.md:75, config/grpo.py:324,330).
//
// Deterministic (no atomics; probabilities recomputed from the forward's base-2 log-sum-exp, the same two-pass arithmetic as
// attention_bwd_pipe.hip); operand tiles staged through LDS by plain loads with the next tile's global loads in flight during the
// products; NOT software-pipelined like the head-dim-64 kernel.  Two kernels: the 16x16x32-MFMA one below (dQ, and the dK/dV
// fallback) and the 32x32x16-MFMA one after it (dK/dV: half the LDS bytes per product).
//   dQ    (DKDV = false): a workgroup owns 128 queries (wave = 32, as two 16-row blocks) and streams the keys in tiles of 32:
//           S^T = K Q^T, P^T = exp2(c S^T - L[q]), dP^T = V dO^T, dS^T = P^T (dP^T - D[q]), dQ^T += K^T dS^T
//   dK/dV (DKDV = true):  a workgroup owns 64 keys (wave = 16) and streams the queries:
//           S = Q K^T, P = exp2(c S - L[q]), dP = dO V^T, dS = P (dP - D[q]), dV^T += dO^T P, dK^T += Q^T dS
// In both, the OWN side is the B operand of every product (lane & 15 = own row), so the score tile comes out of the matrix unit
// with C layout "column = own row, rows = streamed rows (lane >> 4) * 4 + r" -- which IS the B-operand layout of the accumulating
// products once the contraction index is permuted (slot j of k-group g <-> streamed row 16 (j >> 2) + 4 g + (j & 3)): P and dS
// never leave registers, and the A operand (the streamed tile transposed, [d][32 rows] in LDS) is read with the same permutation
// as two 8-byte reads.
#include "attention_bwd.hpp"

namespace advgrpo {

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// B8_CB (template): 16-row blocks of own rows per wave -- every A fragment read from LDS feeds B8_CB MFMAs; a workgroup owns 64 B8_CB rows
constexpr int B8_ROWS = 32;           // streamed rows per tile
constexpr int B8_HD = 128;
constexpr int B8_RP = B8_HD;          // row-major tile pitch (elements): 256 bytes; 16-byte chunk c of row r sits at chunk c ^ (r & 15),
                                      // so the 16 rows x 4 chunks of one fragment read spread evenly over the 64 banks
constexpr int B8_TP = B8_ROWS + 4;    // transposed tile pitch (elements): 72 bytes -- the eight d-rows one wave's transposing
                                      // stores hit are 576 bytes apart: four bank groups (80 bytes put them on two)
constexpr int B8_ROWMAJ = B8_ROWS * B8_RP * 2, B8_TRANS = B8_HD * B8_TP * 2;      // bytes

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2(f32x2_t v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v_t)); }   // v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t b8_pack(float lo, float hi) { return pack2(f32x2_t{lo, hi}); }

template <bool DKDV, int B8_CB>
__global__ __launch_bounds__(256, B8_CB == 1 ? (DKDV ? 3 : 4) : 2) void attn_bwd_d128_kernel(const AttnBwdParams p) {
    constexpr int B8_OWN = 64 * B8_CB;
    __shared__ __attribute__((aligned(16))) char smem[2 * B8_ROWMAJ + 2 * B8_TRANS + 2 * B8_ROWS * 4];
    bf16_t* x0 = reinterpret_cast<bf16_t*>(smem);                        // streamed operand 0 (K | Q), row-major
    bf16_t* x1 = reinterpret_cast<bf16_t*>(smem + B8_ROWMAJ);             // streamed operand 1 (V | dO), row-major
    bf16_t* x0t = reinterpret_cast<bf16_t*>(smem + 2 * B8_ROWMAJ);        // operand 0 transposed [d][row]
    bf16_t* x1t = reinterpret_cast<bf16_t*>(smem + 2 * B8_ROWMAJ + B8_TRANS);
    float* lvec = reinterpret_cast<float*>(smem + 2 * B8_ROWMAJ + 2 * B8_TRANS);      // dK/dV: -L and -D of the tile's queries
    float* dvec = lvec + B8_ROWS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kg = lane >> 4;
    const int n_own = DKDV ? p.Skv : p.Sq, n_str = DKDV ? p.Sq : p.Skv;
    int blk, h, b;
    xcd_local_bh((n_own + B8_OWN - 1) / B8_OWN, p.H, (int)gridDim.x, p.xcd_local, blk, h, b);
    const int own0 = blk * B8_OWN + wave * (16 * B8_CB);
    const int64_t bh = (int64_t)b * p.H + h;
    const bf16_t* b0p = (DKDV ? p.k + (int64_t)b * p.bsk : p.q + (int64_t)b * p.bsq) + h * B8_HD;
    const bf16_t* b1p = (DKDV ? p.v + (int64_t)b * p.bsv : p.d_o + (int64_t)b * p.bsdo) + h * B8_HD;
    const int64_t ld_b0 = DKDV ? p.ldk : p.ldq, ld_b1 = DKDV ? p.ldv : p.lddo;
    const bf16_t* s0p = (DKDV ? p.q + (int64_t)b * p.bsq : p.k + (int64_t)b * p.bsk) + h * B8_HD;
    const bf16_t* s1p = (DKDV ? p.d_o + (int64_t)b * p.bsdo : p.v + (int64_t)b * p.bsv) + h * B8_HD;
    const int64_t ld_s0 = DKDV ? p.ldq : p.ldk, ld_s1 = DKDV ? p.lddo : p.ldv;
    const float* Lp = p.lse + bh * p.Sq;             // base-2 log-sum-exp of the scaled scores
    const float* Dp = p.vec + bh * p.Sq;             // D[q] = sum_d O dO (attn_bwd_delta128_kernel)
    const float c = p.scale_log2e;

    // own-side B fragments: row own0 + cb * 16 + col, k = d = ks * 32 + kg * 8 .. + 7
    bf16x8_t b0[B8_CB][4], b1[B8_CB][4];
    float negL_own[B8_CB], negD_own[B8_CB];
#pragma unroll
    for (int cb = 0; cb < B8_CB; ++cb) {
        const int own_r = min(own0 + cb * 16 + col, n_own - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            b0[cb][ks] = *reinterpret_cast<const bf16x8_t*>(b0p + (int64_t)own_r * ld_b0 + ks * 32 + kg * 8);
            b1[cb][ks] = *reinterpret_cast<const bf16x8_t*>(b1p + (int64_t)own_r * ld_b1 + ks * 32 + kg * 8);
        }
        negL_own[cb] = 0.f; negD_own[cb] = 0.f;
        if constexpr (!DKDV) { negL_own[cb] = -Lp[own_r]; negD_own[cb] = -Dp[own_r]; }
    }

    f32x4_t acc0[B8_CB][8], acc1[B8_CB][8];
#pragma unroll
    for (int cb = 0; cb < B8_CB; ++cb)
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc0[cb][i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc1[cb][i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

    // tile loader: thread -> (row = tid >> 3, 16-byte chunks (tid & 7) and (tid & 7) + 8) of both operands
    const int lrow = tid >> 3, lch = tid & 7;
    uint4 g0[2], g1[2];
    float gl = 0.f, gd = 0.f;
    // (uniform base + 32-bit byte offset per lane -- the launcher checks rows * ld * 2 < 2^31: 64-bit per-lane pointers cost registers
    // that spilled into the tile loop)
    const uint32_t ldb0 = (uint32_t)ld_s0 * 2u, ldb1 = (uint32_t)ld_s1 * 2u;
    auto fetch = [&](int t) {
        const uint32_t r = (uint32_t)min(t * B8_ROWS + lrow, n_str - 1);
        const char* q0 = reinterpret_cast<const char*>(s0p) + (size_t)(r * ldb0 + (uint32_t)lch * 16u);
        const char* q1 = reinterpret_cast<const char*>(s1p) + (size_t)(r * ldb1 + (uint32_t)lch * 16u);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            g0[i] = *reinterpret_cast<const uint4*>(q0 + 128 * i);
            g1[i] = *reinterpret_cast<const uint4*>(q1 + 128 * i);
        }
        if constexpr (DKDV) {
            if (tid < B8_ROWS) {
                const int q = min(t * B8_ROWS + tid, n_str - 1);
                gl = Lp[q];                     // (raw: negated / masked in commit, so that nothing waits on these loads here)
                gd = Dp[q];
            }
        }
    };
    auto commit = [&](int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = lch + 8 * i;
            *reinterpret_cast<uint4*>(x0 + lrow * B8_RP + ((ch ^ (lrow & 15)) * 8)) = g0[i];
            *reinterpret_cast<uint4*>(x1 + lrow * B8_RP + ((ch ^ (lrow & 15)) * 8)) = g1[i];
            // transposed copy [d][row] as 4-byte stores of two neighbouring rows: the lane 8 further on holds row lrow ^ 1 of the same
            // chunk (DPP row_ror:8 swaps the halves of a 16-lane row); the even row's lane writes d = 8 ch + 0..3, the odd row's 4..7
            const uint32_t w0[4] = {g0[i].x, g0[i].y, g0[i].z, g0[i].w}, w1[4] = {g1[i].x, g1[i].y, g1[i].z, g1[i].w};
            const int odd = lrow & 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {                     // dword j (+ 2 for the odd row's lane) = d pair (2j, 2j + 1) (+ 4)
                const uint32_t mine0 = odd ? w0[j + 2] : w0[j], give0 = odd ? w0[j] : w0[j + 2];
                const uint32_t got0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give0, 0x128, 0xf, 0xf, true);
                const uint32_t ev0 = odd ? got0 : mine0, od0 = odd ? mine0 : got0;      // (even row's value, odd row's value) of the d pair
                const int d = ch * 8 + 4 * odd + 2 * j;
                *reinterpret_cast<uint32_t*>(x0t + d * B8_TP + (lrow & ~1)) = (ev0 & 0xffffu) | (od0 << 16);
                *reinterpret_cast<uint32_t*>(x0t + (d + 1) * B8_TP + (lrow & ~1)) = (ev0 >> 16) | (od0 & 0xffff0000u);
                if constexpr (DKDV) {
                    const uint32_t mine1 = odd ? w1[j + 2] : w1[j], give1 = odd ? w1[j] : w1[j + 2];
                    const uint32_t got1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give1, 0x128, 0xf, 0xf, true);
                    const uint32_t ev1 = odd ? got1 : mine1, od1 = odd ? mine1 : got1;
                    *reinterpret_cast<uint32_t*>(x1t + d * B8_TP + (lrow & ~1)) = (ev1 & 0xffffu) | (od1 << 16);
                    *reinterpret_cast<uint32_t*>(x1t + (d + 1) * B8_TP + (lrow & ~1)) = (ev1 >> 16) | (od1 & 0xffff0000u);
                }
            }
        }
        if constexpr (DKDV) {
            if (tid < B8_ROWS) {
                const bool in = t * B8_ROWS + tid < n_str;
                lvec[tid] = in ? -gl : -INFINITY;       // a query past the end: P = 0 exactly
                dvec[tid] = in ? -gd : 0.f;
            }
        }
    };

    const int nt = (n_str + B8_ROWS - 1) / B8_ROWS;
    f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    asm volatile("" : "+v"(zero4));            // (opaque: stays ONE register quad instead of being re-materialised per use)
    fetch(0);
    for (int t = 0; t < nt; ++t) {
        __syncthreads();                      // everyone is done with the previous tile
        commit(t);
        __syncthreads();
        if (t + 1 < nt) fetch(t + 1);         // in flight during the products

        const bool last_ragged = t == nt - 1 && nt * B8_ROWS > n_str;       // (uniform: only the last tile of a ragged sequence masks keys)
        // ---- scores and dP of the two 16-row blocks of the tile x the wave's B8_CB own-row blocks
        f32x4_t sc[2][B8_CB], dp[2][B8_CB];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(x0 + (rb * 16 + col) * B8_RP + (((ks * 4 + kg) ^ col) * 8));
                const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(x1 + (rb * 16 + col) * B8_RP + (((ks * 4 + kg) ^ col) * 8));
#pragma unroll
                for (int cb = 0; cb < B8_CB; ++cb) {
                    // (the first product starts from ONE zero quad kept outside the tile loop: no per-tile clearing of 32 registers)
                    sc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0[cb][ks], ks == 0 ? zero4 : sc[rb][cb], 0, 0, 0);
                    dp[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1[cb][ks], ks == 0 ? zero4 : dp[rb][cb], 0, 0, 0);
                }
            }
        }
        // ---- P and dS in registers: element (rb, r) belongs to streamed row rb * 16 + kg * 4 + r, own row cb * 16 + col
        bf16x8_t dsf[B8_CB], pf[B8_CB];
#pragma unroll
        for (int cb = 0; cb < B8_CB; ++cb) {
            float pv[8], ds[8];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int sr = rb * 16 + kg * 4 + r;
                    float nl, nd;
                    if constexpr (DKDV) { nl = lvec[sr]; nd = dvec[sr]; }
                    else { nl = (last_ragged && t * B8_ROWS + sr >= n_str) ? -INFINITY : negL_own[cb]; nd = negD_own[cb]; }   // a key past the end: P = 0
                    const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[rb][cb][r], c, nl));
                    pv[rb * 4 + r] = pe;
                    ds[rb * 4 + r] = pe * (dp[rb][cb][r] + nd);
                }
            uint4 dsw, pw;
            dsw.x = b8_pack(ds[0], ds[1]); dsw.y = b8_pack(ds[2], ds[3]); dsw.z = b8_pack(ds[4], ds[5]); dsw.w = b8_pack(ds[6], ds[7]);
            pw.x = b8_pack(pv[0], pv[1]); pw.y = b8_pack(pv[2], pv[3]); pw.z = b8_pack(pv[4], pv[5]); pw.w = b8_pack(pv[6], pv[7]);
            dsf[cb] = __builtin_bit_cast(bf16x8_t, dsw);
            pf[cb] = __builtin_bit_cast(bf16x8_t, pw);
        }
        // ---- accumulating products: A = transposed streamed tile, rows d = db * 16 + col, k slots in the permuted order
        uint4 awv[DKDV ? 1 : 8];
        if constexpr (!DKDV) {
            // dQ: all eight A fragments requested up front (the score registers are dead by now): left to itself the compiler issues
            // read, wait, two products, eight times over -- an exposed LDS latency per pair of products
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                const bf16_t* r0 = x0t + (db * 16 + col) * B8_TP + kg * 4;
                const uint2 lo = *reinterpret_cast<const uint2*>(r0), up = *reinterpret_cast<const uint2*>(r0 + 16);
                awv[db].x = lo.x; awv[db].y = lo.y; awv[db].z = up.x; awv[db].w = up.y;
            }
            __builtin_amdgcn_sched_barrier(0);       // (keeps the machine scheduler from sinking the reads back to their uses)
        }
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            uint4 aw;
            if constexpr (!DKDV) {
                aw = awv[db];
            } else {
                const bf16_t* r0 = x0t + (db * 16 + col) * B8_TP + kg * 4;
                const uint2 lo = *reinterpret_cast<const uint2*>(r0), up = *reinterpret_cast<const uint2*>(r0 + 16);
                aw.x = lo.x; aw.y = lo.y; aw.z = up.x; aw.w = up.y;
            }
#pragma unroll
            for (int cb = 0; cb < B8_CB; ++cb)
                acc0[cb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, aw), dsf[cb], acc0[cb][db], 0, 0, 0);
            if constexpr (DKDV) {
                const bf16_t* r1 = x1t + (db * 16 + col) * B8_TP + kg * 4;
                uint4 bw;
                const uint2 lo1 = *reinterpret_cast<const uint2*>(r1), up1 = *reinterpret_cast<const uint2*>(r1 + 16);
                bw.x = lo1.x; bw.y = lo1.y; bw.z = up1.x; bw.w = up1.y;
#pragma unroll
                for (int cb = 0; cb < B8_CB; ++cb)
                    acc1[cb][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bw), pf[cb], acc1[cb][db], 0, 0, 0);
            }
        }
    }

    // ---- store: acc[cb][db] holds (own row cb * 16 + col, d = db * 16 + kg * 4 + r): four consecutive d = 8 bytes per lane and block
#pragma unroll
    for (int cb = 0; cb < B8_CB; ++cb) {
        const int orow = own0 + cb * 16 + col;
        if (orow >= n_own) continue;
        bf16_t* o0 = (DKDV ? p.dk : p.dq) + (int64_t)b * p.bsdq + (int64_t)orow * p.lddq + h * B8_HD;
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            uint2 v;
            v.x = b8_pack(acc0[cb][db][0] * p.scale, acc0[cb][db][1] * p.scale);
            v.y = b8_pack(acc0[cb][db][2] * p.scale, acc0[cb][db][3] * p.scale);
            *reinterpret_cast<uint2*>(o0 + db * 16 + kg * 4) = v;
        }
        if constexpr (DKDV) {
            bf16_t* o1 = p.dv + (int64_t)b * p.bsdq + (int64_t)orow * p.lddq + h * B8_HD;
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                uint2 v;
                v.x = b8_pack(acc1[cb][db][0], acc1[cb][db][1]);
                v.y = b8_pack(acc1[cb][db][2], acc1[cb][db][3]);
                *reinterpret_cast<uint2*>(o1 + db * 16 + kg * 4) = v;
            }
        }
    }
}

// dK/dV on v_mfma_f32_32x32x16_bf16: a workgroup owns 128 keys (wave = 32) and streams the queries in the same 32-row tiles (same LDS
// images, same loader) as the kernel above.  Per operand byte read from LDS the 32x32 product does twice the arithmetic of the 16x16 one:
// with one 16-key block per wave the kernel above reads 32 KB of LDS per wave and tile for 32 MFMA-16 (LDS : matrix time 2 : 1), this one
// 32 KB for 32 MFMA-32 (1 : 1).  Same algebra and the same register-resident P / dS trick: S = Q K^T leaves the matrix unit with
// C layout "column = own key (lane & 31), rows = queries 8 (r >> 2) + 4 (lane >> 5) + (r & 3)", which is the B operand of the
// accumulating products when k-slot j of k-step s and group g = lane >> 5 stands for query 16 s + 8 (j >> 2) + 4 g + (j & 3) = C
// register 8 s + j; the A operand (the transposed tile) is read with that permutation as two 8-byte reads 8 queries apart.
typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <bool DKDV, int AHEAD = 0, bool LATE = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_d128_w32_kernel(const AttnBwdParams p) {
    constexpr int OWN = 128;
    // LDS: streamed tile row-major x 2 | transposed x0 (| transposed x1 | -L, -D of the tile's queries | the workgroup's own V rows)
    constexpr int OFF_T0 = 2 * B8_ROWMAJ, OFF_T1 = OFF_T0 + B8_TRANS, OFF_LD = OFF_T1 + B8_TRANS, OFF_XV = OFF_LD + 2 * B8_ROWS * 4;
    __shared__ __attribute__((aligned(16))) char smem[DKDV ? OFF_XV + OWN * B8_RP * 2 : OFF_T1];
    bf16_t* x0 = reinterpret_cast<bf16_t*>(smem);                        // streamed operand 0 (Q | K), row-major, swizzled chunks
    bf16_t* x1 = reinterpret_cast<bf16_t*>(smem + B8_ROWMAJ);             // streamed operand 1 (dO | V)
    bf16_t* x0t = reinterpret_cast<bf16_t*>(smem + OFF_T0);               // operand 0 transposed [d][row]
    bf16_t* x1t = reinterpret_cast<bf16_t*>(smem + (DKDV ? OFF_T1 : 0));  // (dK/dV only)
    float* lvec = reinterpret_cast<float*>(smem + (DKDV ? OFF_LD : 0));   // (dK/dV only)
    float* dvec = lvec + B8_ROWS;
    bf16_t* xv = reinterpret_cast<bf16_t*>(smem + (DKDV ? OFF_XV : 0));   // (dK/dV only)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, hi = lane >> 5;
    const int n_own = DKDV ? p.Skv : p.Sq, n_str = DKDV ? p.Sq : p.Skv;
    int blk, h, b;
    xcd_local_bh((n_own + OWN - 1) / OWN, p.H, (int)gridDim.x, p.xcd_local, blk, h, b);
    const int own0 = blk * OWN + wave * 32;
    const int64_t bh = (int64_t)b * p.H + h;
    const bf16_t* s0p = (DKDV ? p.q + (int64_t)b * p.bsq : p.k + (int64_t)b * p.bsk) + h * B8_HD;
    const bf16_t* s1p = (DKDV ? p.d_o + (int64_t)b * p.bsdo : p.v + (int64_t)b * p.bsv) + h * B8_HD;
    const int64_t ld_s0 = DKDV ? p.ldq : p.ldk, ld_s1 = DKDV ? p.lddo : p.ldv;
    const float* Lp = p.lse + bh * p.Sq;
    const float* Dp = p.vec + bh * p.Sq;
    const float c = p.scale_log2e;

    // own-side B fragments: row own0 + col, k = d = ks * 16 + hi * 8 .. + 7.  dK/dV: K stays in registers; V (32 more registers: with
    // both, and two accumulator sets, the kernel spills at two waves per SIMD) is read from an LDS image written once.  dQ: Q and dO in
    // registers (one accumulator set)
    bf16x8_t b0[8], b1r[DKDV ? 1 : 8];
    float negL_own = 0.f, negD_own = 0.f;
    {
        const int own_r = min(own0 + col, n_own - 1);
        const bf16_t* o0p = (DKDV ? p.k + (int64_t)b * p.bsk + (int64_t)own_r * p.ldk : p.q + (int64_t)b * p.bsq + (int64_t)own_r * p.ldq) + h * B8_HD + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) b0[ks] = *reinterpret_cast<const bf16x8_t*>(o0p + ks * 16);
        if constexpr (DKDV) {
#pragma unroll
            for (int i = 0; i < OWN * 16 / 256; ++i) {
                const int e = tid + 256 * i, r = e >> 4, ch = e & 15;
                const int vr = min(blk * OWN + r, n_own - 1);
                *reinterpret_cast<uint4*>(xv + r * B8_RP + ((ch ^ (r & 15)) * 8)) =
                    *reinterpret_cast<const uint4*>(p.v + (int64_t)b * p.bsv + (int64_t)vr * p.ldv + h * B8_HD + ch * 8);
            }
        } else {
            const bf16_t* o1p = p.d_o + (int64_t)b * p.bsdo + (int64_t)own_r * p.lddo + h * B8_HD + hi * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) b1r[ks] = *reinterpret_cast<const bf16x8_t*>(o1p + ks * 16);
            negL_own = -Lp[own_r];
            negD_own = -Dp[own_r];
        }
    }
    f32x16_t acc0[4], acc1[DKDV ? 4 : 1];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[i][r] = 0.f; if constexpr (DKDV) acc1[i][r] = 0.f; }

    const int lrow = tid >> 3, lch = tid & 7;
    uint4 g0[2], g1[2];
    float gl = 0.f, gd = 0.f;
    // (uniform base + 32-bit byte offset per lane: the launcher checks rows * ld * 2 < 2^31.  64-bit per-lane pointers cost the registers
    // whose spilling put two scratch round trips into every tile)
    const uint32_t ldb0 = (uint32_t)ld_s0 * 2u, ldb1 = (uint32_t)ld_s1 * 2u;
    auto fetch = [&](int t) {
        const uint32_t r = (uint32_t)min(t * B8_ROWS + lrow, n_str - 1);
        const char* q0 = reinterpret_cast<const char*>(s0p) + (size_t)(r * ldb0 + (uint32_t)lch * 16u);
        const char* q1 = reinterpret_cast<const char*>(s1p) + (size_t)(r * ldb1 + (uint32_t)lch * 16u);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            g0[i] = *reinterpret_cast<const uint4*>(q0 + 128 * i);
            g1[i] = *reinterpret_cast<const uint4*>(q1 + 128 * i);
        }
        if constexpr (DKDV) {
            if (tid < B8_ROWS) {
                const int q = min(t * B8_ROWS + tid, n_str - 1);
                gl = Lp[q];                     // (raw: negated / masked in commit, so that nothing waits on these loads here)
                gd = Dp[q];
            }
        }
    };
    auto commit = [&](int t) {   // (the loader of the kernel above: row-major image + transposed image by DPP-paired 4-byte stores)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = lch + 8 * i;
            *reinterpret_cast<uint4*>(x0 + lrow * B8_RP + ((ch ^ (lrow & 15)) * 8)) = g0[i];
            *reinterpret_cast<uint4*>(x1 + lrow * B8_RP + ((ch ^ (lrow & 15)) * 8)) = g1[i];
            const uint32_t w0[4] = {g0[i].x, g0[i].y, g0[i].z, g0[i].w}, w1[4] = {g1[i].x, g1[i].y, g1[i].z, g1[i].w};
            const int odd = lrow & 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t mine0 = odd ? w0[j + 2] : w0[j], give0 = odd ? w0[j] : w0[j + 2];
                const uint32_t got0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give0, 0x128, 0xf, 0xf, true);
                const uint32_t ev0 = odd ? got0 : mine0, od0 = odd ? mine0 : got0;
                const int d = ch * 8 + 4 * odd + 2 * j;
                *reinterpret_cast<uint32_t*>(x0t + d * B8_TP + (lrow & ~1)) = (ev0 & 0xffffu) | (od0 << 16);
                *reinterpret_cast<uint32_t*>(x0t + (d + 1) * B8_TP + (lrow & ~1)) = (ev0 >> 16) | (od0 & 0xffff0000u);
                if constexpr (DKDV) {
                    const uint32_t mine1 = odd ? w1[j + 2] : w1[j], give1 = odd ? w1[j] : w1[j + 2];
                    const uint32_t got1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give1, 0x128, 0xf, 0xf, true);
                    const uint32_t ev1 = odd ? got1 : mine1, od1 = odd ? mine1 : got1;
                    *reinterpret_cast<uint32_t*>(x1t + d * B8_TP + (lrow & ~1)) = (ev1 & 0xffffu) | (od1 << 16);
                    *reinterpret_cast<uint32_t*>(x1t + (d + 1) * B8_TP + (lrow & ~1)) = (ev1 >> 16) | (od1 & 0xffff0000u);
                }
            }
        }
        if constexpr (DKDV) {
            if (tid < B8_ROWS) {
                const bool in = t * B8_ROWS + tid < n_str;
                lvec[tid] = in ? -gl : -INFINITY;       // a query past the end: P = 0 exactly
                dvec[tid] = in ? -gd : 0.f;
            }
        }
    };

    const int nt = (n_str + B8_ROWS - 1) / B8_ROWS;
    fetch(0);
    for (int t = 0; t < nt; ++t) {
        __syncthreads();
        commit(t);
        __syncthreads();
        if (!LATE && t + 1 < nt) fetch(t + 1);
        const bool last_ragged = t == nt - 1 && nt * B8_ROWS > n_str;       // (uniform; dQ: only the last tile of a ragged sequence masks keys)

        // fragment address of k-step ks = (row base + swizzled chunk of k-step 0) ^ (ks << 5): chunk (2 ks + hi) ^ (col & 15) = (2 ks) ^
        // (hi ^ (col & 15)), and the row base is a multiple of 256.  The base is made opaque once per tile so that the eight addresses
        // are ONE register and an XOR each instead of sixteen loop-invariant registers (they were what pushed this kernel into spilling)
        uint32_t fbase = (uint32_t)(col * (B8_RP * 2) + ((hi ^ (col & 15)) << 4));
        asm volatile("" : "+v"(fbase));
        const char* x0c = reinterpret_cast<const char*>(x0);
        const char* xvc = reinterpret_cast<const char*>(xv) + (wave * 32) * (B8_RP * 2);
        // ---- scores and dP: the tile's 32 rows x the wave's 32 own rows, k = d in 8 steps of 16
        f32x16_t sc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
        if constexpr (AHEAD > 0) {
            // fragments AHEAD k-steps ahead of the products (the tile loads are issued AFTER this phase, so their 16 registers are free here)
            bf16x8_t fa0[8], fa1[8], fb1[DKDV ? 8 : 1];
            auto rd = [&](int ks) {
                fa0[ks] = *reinterpret_cast<const bf16x8_t*>(x0c + (fbase ^ (uint32_t)(ks << 5)));
                fa1[ks] = *reinterpret_cast<const bf16x8_t*>(x0c + B8_ROWMAJ + (fbase ^ (uint32_t)(ks << 5)));
                if constexpr (DKDV) fb1[ks] = *reinterpret_cast<const bf16x8_t*>(xvc + (fbase ^ (uint32_t)(ks << 5)));
            };
            constexpr int NRD = DKDV ? 3 : 2;
#pragma unroll
            for (int ks = 0; ks < AHEAD; ++ks) rd(ks);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + AHEAD < 8) rd(ks + AHEAD);
                const int left = NRD * (7 - ks < AHEAD ? 7 - ks : AHEAD);     // reads that may still be in flight: those of later steps
                if (left == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                else if (left == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                else if (left == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                else if (left == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                else if (left == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                bf16x8_t b1;
                if constexpr (DKDV) b1 = fb1[ks]; else b1 = b1r[ks];
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[ks], b0[ks], sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[ks], b1, dp, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(x0c + (fbase ^ (uint32_t)(ks << 5)));
                const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(x0c + B8_ROWMAJ + (fbase ^ (uint32_t)(ks << 5)));
                bf16x8_t b1;
                if constexpr (DKDV) b1 = *reinterpret_cast<const bf16x8_t*>(xvc + (fbase ^ (uint32_t)(ks << 5)));
                else b1 = b1r[ks];
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0[ks], sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, dp, 0, 0, 0);
            }
        }
        // ---- P and dS in registers: C register r = streamed row 8 (r >> 2) + 4 hi + (r & 3) of the tile, own row own0 + col
        bf16x8_t pf[2], dsf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint32_t pw[4], dw[4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {                       // C registers 8 s + 4 u .. + 3 = streamed rows 16 s + 8 u + 4 hi + 0..3
                f32x4_t nl, nd;
                if constexpr (DKDV) {
                    nl = *reinterpret_cast<const f32x4_t*>(lvec + 16 * s + 8 * u + 4 * hi);
                    nd = *reinterpret_cast<const f32x4_t*>(dvec + 16 * s + 8 * u + 4 * hi);
                } else {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        nl[v] = (last_ragged && t * B8_ROWS + 16 * s + 8 * u + 4 * hi + v >= n_str) ? -INFINITY : negL_own;   // a key past the end: P = 0
                        nd[v] = negD_own;
                    }
                }
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const int r = 8 * s + 4 * u + 2 * v;
                    f32x2_t pe, dd;
                    pe.x = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], c, nl[2 * v]));
                    pe.y = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r + 1], c, nl[2 * v + 1]));
                    dd.x = dp[r] + nd[2 * v];
                    dd.y = dp[r + 1] + nd[2 * v + 1];
                    dd = dd * pe;
                    if constexpr (DKDV) pw[2 * u + v] = pack2(pe);
                    dw[2 * u + v] = pack2(dd);
                }
            }
            uint4 dsw, pww;
            dsw.x = dw[0]; dsw.y = dw[1]; dsw.z = dw[2]; dsw.w = dw[3];
            dsf[s] = __builtin_bit_cast(bf16x8_t, dsw);
            if constexpr (DKDV) {
                pww.x = pw[0]; pww.y = pw[1]; pww.z = pw[2]; pww.w = pw[3];
                pf[s] = __builtin_bit_cast(bf16x8_t, pww);
            }
        }
        if (LATE && t + 1 < nt) fetch(t + 1);
        // ---- accumulating products (dK^T += Q^T dS, dV^T += dO^T P | dQ^T += K^T dS^T): A = transposed tile rows d = db * 32 + col,
        //      k = streamed rows in the permuted order
        if constexpr (DKDV) {
            // fragments of d-block db + 2 are requested while the products of block db issue (ping-pong register sets, the order pinned by
            // sched_barrier: left alone the compiler issues read, wait, product one after the other)
            uint4 fq[2][4];
            auto rdq = [&](int db, uint4* f) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16_t* r0 = x0t + (db * 32 + col) * B8_TP + 16 * s + 4 * hi;
                    const bf16_t* r1 = x1t + (db * 32 + col) * B8_TP + 16 * s + 4 * hi;
                    const uint2 lo0 = *reinterpret_cast<const uint2*>(r0), up0 = *reinterpret_cast<const uint2*>(r0 + 8);
                    const uint2 lo1 = *reinterpret_cast<const uint2*>(r1), up1 = *reinterpret_cast<const uint2*>(r1 + 8);
                    f[2 * s].x = lo0.x; f[2 * s].y = lo0.y; f[2 * s].z = up0.x; f[2 * s].w = up0.y;
                    f[2 * s + 1].x = lo1.x; f[2 * s + 1].y = lo1.y; f[2 * s + 1].z = up1.x; f[2 * s + 1].w = up1.y;
                }
            };
            rdq(0, fq[0]);
            rdq(1, fq[1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int db = 0; db < 4; ++db) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    acc0[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fq[db & 1][2 * s]), dsf[s], acc0[db], 0, 0, 0);
                    acc1[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fq[db & 1][2 * s + 1]), pf[s], acc1[db], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (db + 2 < 4) {
                    rdq(db + 2, fq[db & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
            for (int db = 0; db < 4; ++db) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16_t* r0 = x0t + (db * 32 + col) * B8_TP + 16 * s + 4 * hi;
                    const uint2 lo0 = *reinterpret_cast<const uint2*>(r0), up0 = *reinterpret_cast<const uint2*>(r0 + 8);
                    uint4 aw;
                    aw.x = lo0.x; aw.y = lo0.y; aw.z = up0.x; aw.w = up0.y;
                    acc0[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, aw), dsf[s], acc0[db], 0, 0, 0);
                }
            }
        }
    }

    // ---- store: acc[db][r] = (own row own0 + col, d = db * 32 + 8 (r >> 2) + 4 hi + (r & 3)): four consecutive d = 8 bytes
    const int orow = own0 + col;
    if (orow < n_own) {
        bf16_t* o0 = (DKDV ? p.dk : p.dq) + (int64_t)b * p.bsdq + (int64_t)orow * p.lddq + h * B8_HD;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint2 v;
                v.x = pack2(f32x2_t{acc0[db][4 * q] * p.scale, acc0[db][4 * q + 1] * p.scale});
                v.y = pack2(f32x2_t{acc0[db][4 * q + 2] * p.scale, acc0[db][4 * q + 3] * p.scale});
                *reinterpret_cast<uint2*>(o0 + db * 32 + 8 * q + 4 * hi) = v;
            }
        if constexpr (DKDV) {
            bf16_t* o1 = p.dv + (int64_t)b * p.bsdq + (int64_t)orow * p.lddq + h * B8_HD;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint2 w;
                    w.x = pack2(f32x2_t{acc1[db][4 * q], acc1[db][4 * q + 1]});
                    w.y = pack2(f32x2_t{acc1[db][4 * q + 2], acc1[db][4 * q + 3]});
                    *reinterpret_cast<uint2*>(o1 + db * 32 + 8 * q + 4 * hi) = w;
                }
        }
    }
}

// D[b, h, q] = sum_d O[q, d] dO[q, d]: one wave per (b, q) row, 16 lanes per head
__global__ __launch_bounds__(256) void attn_bwd_delta128_kernel(const AttnBwdParams p, int B) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)B * p.Sq) return;
    const int b = row / p.Sq, q = row % p.Sq;
    const bf16_t* o = p.o + (int64_t)b * p.bso + (int64_t)q * p.ldo;
    const bf16_t* d = p.d_o + (int64_t)b * p.bsdo + (int64_t)q * p.lddo;
    const int sub = lane & 15;
    for (int h0 = 0; h0 < p.H; h0 += 4) {
        const int h = h0 + (lane >> 4);
        float s = 0.f;
        if (h < p.H) {
            const uint4 a = *reinterpret_cast<const uint4*>(o + h * B8_HD + sub * 8);
            const uint4 c = *reinterpret_cast<const uint4*>(d + h * B8_HD + sub * 8);
            const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, cw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                s += bf2f((bf16_t)(aw[k] & 0xffffu)) * bf2f((bf16_t)(cw[k] & 0xffffu)) +
                     bf2f((bf16_t)(aw[k] >> 16)) * bf2f((bf16_t)(cw[k] >> 16));
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 8, 64);
        if (h < p.H && sub == 0) p.vec[((int64_t)b * p.H + h) * p.Sq + q] = s;
    }
}

}  // namespace

int attention_bwd_d128_launch(const AttnBwdParams& p, int B, hipStream_t s) {
    // every operand is addressed with 32-bit byte offsets from a per-(batch, head) base: checked for ALL of them -- q, dO, o, k, v and the
    // dq / dk / dv pitch -- and BEFORE the first launch (the delta kernel reads o and dO with the same arithmetic)
    auto fits = [](int64_t rows, int64_t ld) { return rows * ld * 2 < (1ll << 31); };
    const bool fits32 = fits(p.Sq, p.ldq) && fits(p.Sq, p.lddo) && fits(p.Sq, p.ldo) && fits(p.Skv, p.ldk) && fits(p.Skv, p.ldv) &&
                        fits(p.Sq > p.Skv ? p.Sq : p.Skv, p.lddq);
    ADVGRPO_CHECK(fits32, "attention_bwd (d128): rows * row stride must stay below 2^30 elements (32-bit byte offsets)");
    hipLaunchKernelGGL(attn_bwd_delta128_kernel, dim3((unsigned)(((int64_t)B * p.Sq + 3) / 4)), dim3(256), 0, s, p, B);
    ADVGRPO_LAUNCH_CHECK();
    // dQ: the 16x16 kernel with two own-row blocks per wave (227 registers, two waves per SIMD).  dK/dV: the 32x32 kernel (254 registers, two
    // waves per SIMD).  Same-box kernel times at 16 x 24 x 4224 (rocprofv3; scripts/probes/attn_bwd_d128_variants.sh):
    //   dK/dV  16x16 one block per wave 9.96 ms | 32x32 7.9 - 8.1 ms | 32x32 with the tile loads issued after the score phase 9.8 ms (spills)
    //          | 32x32 + pinned ping-pong fragment reads in the accumulating phase + one-register XOR fragment addresses 7.8 - 7.9 (prev. 7.97)
    //          | the same + score-phase fragments one k-step ahead 8.1 - 8.2
    //   dQ     16x16 two blocks 6.9 ms -> 6.55 ms with the eight accumulating-phase fragments requested up front
    //          | 32x32 8.4 ms | 32x32, tile loads after the score phase, fragments 2 steps ahead 7.0 - 7.6 ms
    // ADVGRPO_ATTN_BWD_D128 (A/B switch of the EXPERIMENTS build, read once): dQ variant << 4 | dK/dV variant; 0 = 16x16, 1 = 32x32 (dQ: the late-load form).  Default 0x01.
    constexpr int CBQ = 2, CBK = 1;
#ifdef ADVGRPO_EXPERIMENTS
    static const int variant = [] { const char* e = getenv("ADVGRPO_ATTN_BWD_D128"); return e ? atoi(e) : 1; }();
#else
    constexpr int variant = 1;                 // (the product library reads no environment: Makefile)
#endif
    const int64_t nq = (int64_t)((p.Sq + 64 * CBQ - 1) / (64 * CBQ)) * p.H * B, nk = (int64_t)((p.Skv + 64 * CBK - 1) / (64 * CBK)) * p.H * B;
    ADVGRPO_CHECK(nq < (1ll << 31) && nk < (1ll << 31), "attention_bwd (d128): grid too large");
    const dim3 gq((unsigned)((int64_t)((p.Sq + 127) / 128) * p.H * B)), gk((unsigned)((int64_t)((p.Skv + 127) / 128) * p.H * B));
    if ((variant >> 4) & 1) hipLaunchKernelGGL((attn_bwd_d128_w32_kernel<false, 2, true>), gq, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((attn_bwd_d128_kernel<false, CBQ>), dim3((unsigned)nq), dim3(256), 0, s, p);
    ADVGRPO_LAUNCH_CHECK();
    if (variant & 1) hipLaunchKernelGGL((attn_bwd_d128_w32_kernel<true, 0, false>), gk, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((attn_bwd_d128_kernel<true, CBK>), dim3((unsigned)nk), dim3(256), 0, s, p);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

}  // namespace advgrpo
