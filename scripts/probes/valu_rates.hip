// Probe: issue cost of the softmax-side VALU instructions of the attention kernels on gfx950, alone, mixed with each
// other, beside MFMAs in the same wave, and beside a partner wave on the same SIMD.
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
// Every wave runs ITERS iterations of a fixed body and reports s_memtime ticks per iteration (wave 0 of workgroup 0 and
// the slowest wave of the grid); one workgroup per CU, 256 threads = one wave per SIMD, 512 = two.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define REP32(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) \
                 X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
#define ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x[i]));
#define ADD2(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(accs[i & 3]) : "v"(x[i]));
#define MAXI(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(accs[i & 3]) : "v"(x[i]));
#define MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(accs[i & 3]) : "v"(x[2 * i]), "v"(x[2 * i + 1]));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x2[i]) : "v"(a2), "v"(b2));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc2[i & 1]) : "v"(x2[i]));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x2[i]) : "v"(a2));
#define CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(x[2 * i]), "v"(x[2 * i + 1]));
#define LDEXP(i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x[i]) : "v"(ia));
#define FLOOR(i) asm volatile("v_floor_f32 %0, %0" : "+v"(x[i]));
#define LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %1, 23, %0" : "+v"(x[i]) : "v"(ia));
#define EXPFMA(i) EXP(i) FMA(i)
#define EXPMUL(i) EXP(i) MUL(i)
#define MFMA(i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc16[i & 3]) : "v"(af), "v"(bfr));
#define MFMA_2EXP(i) MFMA(i) EXP(2 * i) EXP(2 * i + 1)
#define MFMA_2EXP_2FMA(i) MFMA(i) EXP(2 * i) FMA(2 * i) EXP(2 * i + 1) FMA(2 * i + 1)
#define MFMA_4FMA(i) MFMA(i) FMA(2 * i) FMA(2 * i + 1) MUL(2 * i) MUL(2 * i + 1)
#define MFMA_8V(i) MFMA(i) FMA(2 * i) FMA(2 * i + 1) MUL(2 * i) MUL(2 * i + 1) FMA(2 * i) FMA(2 * i + 1) MUL(2 * i) MUL(2 * i + 1)
#define MFMA16(i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc4[i & 7]) : "v"(af), "v"(bfr));
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ __launch_bounds__(512) void probe(int iters, unsigned long long* out, float* sink, float seed) {
    float x[32];
    f32x2 x2[16];
    for (int i = 0; i < 32; ++i) x[i] = seed * (float)(i + 1) + (float)threadIdx.x * 1e-6f;
    for (int i = 0; i < 16; ++i) x2[i] = f32x2{x[2 * i], x[2 * i + 1]};
    float a = 0.999f, b = 1e-3f, acc = 0.f, accs[4] = {0.f, 0.f, 0.f, 0.f};
    f32x2 a2 = {0.999f, 0.999f}, b2 = {1e-3f, 1e-3f}, acc2[2] = {{0.f, 0.f}, {0.f, 0.f}};
    unsigned pk[16];
    for (int i = 0; i < 16; ++i) pk[i] = 0;
    int ia = 0;
    bf16x8_t af, bfr;
    for (int i = 0; i < 8; ++i) { af[i] = (__bf16)(seed * 0.01f); bfr[i] = (__bf16)(seed * 0.02f); }
    f32x16 acc16[4];
    f32x4 acc4[8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc16[i][j] = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc4[i][j] = 0.f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wave >> 2;       // second wave of each SIMD in 512-thread launches
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) { REP32(EXP) }
        if constexpr (MODE == 1) { REP32(FMA) }
        if constexpr (MODE == 2) { REP32(EXPFMA) }
        if constexpr (MODE == 3) { REP16(PKFMA) }
        if constexpr (MODE == 4) { REP16(PKADD) }
        if constexpr (MODE == 5) { REP16(PKMUL) }
        if constexpr (MODE == 6) { REP32(MAXI) }
        if constexpr (MODE == 7) { REP16(MAX3) }
        if constexpr (MODE == 8) { REP16(CVT) }
        if constexpr (MODE == 9) { REP16(MFMA) }
        if constexpr (MODE == 10) { REP16(MFMA_2EXP) }
        if constexpr (MODE == 11) { REP16(MFMA_2EXP_2FMA) }
        if constexpr (MODE == 12) { REP32(ADD) }
        if constexpr (MODE == 13) { REP32(LDEXP) }
        if constexpr (MODE == 14) { REP32(FLOOR) }
        if constexpr (MODE == 15) { REP32(LSHLADD) }
        if constexpr (MODE == 16) { REP32(EXPMUL) }
        if constexpr (MODE == 17) { REP16(MFMA_4FMA) }
        if constexpr (MODE == 18) { REP32(ADD2) }
        if constexpr (MODE == 19) { REP16(MFMA_8V) }
        if constexpr (MODE == 20) {          // partner roles: waves 0-3 MFMA only, waves 4-7 exp only (same SIMDs)
            if (role == 0) { REP16(MFMA) } else { REP32(EXP) }
        }
        if constexpr (MODE == 21) {          // waves 0-3 MFMA only, waves 4-7 fma only
            if (role == 0) { REP16(MFMA) } else { REP32(FMA) REP32(MUL) }
        }
        if constexpr (MODE == 22) {          // waves 0-3 MFMA only, waves 4-7 the softmax mix: 32 exp + 32 fma + 32 add + 32 max + 16 cvt
            if (role == 0) { REP16(MFMA) } else { REP32(MAXI) REP32(EXPFMA) REP32(ADD2) REP16(CVT) }
        }
        if constexpr (MODE == 23) { REP32(MAXI) REP32(EXPFMA) REP32(ADD2) REP16(CVT) }     // the softmax mix alone
        if constexpr (MODE == 24) { REP32(MFMA16) }
        if constexpr (MODE == 25) {          // softmax mix with packed forms: 16 max3 + 16 pk_fma + 32 exp + 16 pk_add + 16 cvt
            REP16(MAX3) REP16(PKFMA)
            for (int i = 0; i < 16; ++i) { x[2 * i] = x2[i][0]; x[2 * i + 1] = x2[i][1]; }
            REP32(EXP)
            for (int i = 0; i < 16; ++i) { x2[i][0] = x[2 * i]; x2[i][1] = x[2 * i + 1]; }
            REP16(PKADD) REP16(CVT)
        }
        if constexpr (MODE == 26) {          // roles: MFMA | packed softmax mix
            if (role == 0) { REP16(MFMA) } else {
                REP16(MAX3) REP16(PKFMA)
                for (int i = 0; i < 16; ++i) { x[2 * i] = x2[i][0]; x[2 * i + 1] = x2[i][1]; }
                REP32(EXP)
                for (int i = 0; i < 16; ++i) { x2[i][0] = x[2 * i]; x2[i][1] = x[2 * i + 1]; }
                REP16(PKADD) REP16(CVT)
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 8 + wave] = t1 - t0;
    float s = acc + accs[0] + accs[1] + accs[2] + accs[3] + acc2[0][0] + acc2[0][1] + acc2[1][0] + acc2[1][1];
    for (int i = 0; i < 32; ++i) s += x[i];
    for (int i = 0; i < 16; ++i) s += x2[i][0] + x2[i][1] + (float)pk[i];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc16[i][j];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc4[i][j];
    if (s == 123.456f) sink[0] = s;
}

template <int MODE>
static void run(const char* name, int threads, int n_inst, unsigned long long* out, float* sink) {
    const int iters = 2000, blocks = 256;
    hipMemset(out, 0, blocks * 8 * 8);
    hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(threads), 0, 0, 10, out, sink, 0.5f);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(threads), 0, 0, iters, out, sink, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), out, blocks * 8 * 8, hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    double lo = 0, hi = 0;     // mean ticks of the first four waves / of waves 4-7
    for (int bl = 0; bl < blocks; ++bl) for (int w = 0; w < nw; ++w) (w < 4 ? lo : hi) += (double)h[bl * 8 + w];
    lo /= blocks * 4.0 * iters; hi = nw > 4 ? hi / (blocks * 4.0 * iters) : 0;
    printf("%-64s thr %3d  ticks/iter waves0-3 %8.1f  waves4-7 %8.1f  per inst %6.2f   wall %7.3f ms -> %7.1f ns/iter\n", name, threads,
           lo, hi, lo / n_inst, ms, ms * 1e6 / iters);
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256 * 8 * 8); hipMalloc(&sink, 4);
    for (int thr : {256, 512}) {
        run<0>("32 v_exp_f32", thr, 32, out, sink);
        run<1>("32 v_fma_f32", thr, 32, out, sink);
        run<2>("32 (v_exp + v_fma)", thr, 64, out, sink);
        run<16>("32 (v_exp + v_mul)", thr, 64, out, sink);
        run<3>("16 v_pk_fma_f32", thr, 16, out, sink);
        run<4>("16 v_pk_add_f32 (2 accumulators)", thr, 16, out, sink);
        run<5>("16 v_pk_mul_f32", thr, 16, out, sink);
        run<6>("32 v_max_f32 (4 accumulators)", thr, 32, out, sink);
        run<7>("16 v_max3_f32 (4 accumulators)", thr, 16, out, sink);
        run<8>("16 v_cvt_pk_bf16_f32", thr, 16, out, sink);
        run<12>("32 v_add_f32 (1 accumulator: dependent chain)", thr, 32, out, sink);
        run<18>("32 v_add_f32 (4 accumulators)", thr, 32, out, sink);
        run<13>("32 v_ldexp_f32", thr, 32, out, sink);
        run<14>("32 v_floor_f32", thr, 32, out, sink);
        run<15>("32 v_lshl_add_u32", thr, 32, out, sink);
        run<9>("16 v_mfma_f32_32x32x16_bf16 (4 accumulators)", thr, 16, out, sink);
        run<24>("32 v_mfma_f32_16x16x32_bf16 (8 accumulators)", thr, 32, out, sink);
        run<10>("16 (mfma32 + 2 v_exp)", thr, 16, out, sink);
        run<11>("16 (mfma32 + 2 v_exp + 2 v_fma)", thr, 16, out, sink);
        run<17>("16 (mfma32 + 2 v_fma + 2 v_mul)", thr, 16, out, sink);
        run<19>("16 (mfma32 + 4 v_fma + 4 v_mul)", thr, 16, out, sink);
        run<23>("softmax mix: 32 max + 32 (exp+fma) + 32 add + 16 cvt = 144", thr, 144, out, sink);
        run<25>("packed mix: 16 max3 + 16 pk_fma + 32 exp + 16 pk_add + 16 cvt = 96", thr, 96, out, sink);
    }
    run<20>("roles: waves0-3 16 mfma32 | waves4-7 32 v_exp", 512, 16, out, sink);
    run<21>("roles: waves0-3 16 mfma32 | waves4-7 32 fma + 32 mul", 512, 16, out, sink);
    run<22>("roles: waves0-3 16 mfma32 | waves4-7 softmax mix (144)", 512, 16, out, sink);
    run<26>("roles: waves0-3 16 mfma32 | waves4-7 packed mix (96)", 512, 16, out, sink);
    return 0;
}
