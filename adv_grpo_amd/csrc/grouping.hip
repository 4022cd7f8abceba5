// grouping.hip -- per-prompt group advantage (float64) and the clipped GRPO surrogate (gfx950).
//
// group advantage: adv_grpo/stat_tracking.py:18-47 (PerPromptStatTracker.update, type='grpo',
// fresh tracker).  N*T*8 bytes (12 KB at N=768,T=2): one workgroup, latency bound.  The point of
// the kernel is bit-exact grouping and numpy-identical float64 arithmetic with no host round
// trip (the reference ships rewards to the host, decodes 768 prompts from token ids and loops
// in numpy): rows are bucketed by an int32 group key with a stable O(N^2/threads) counting
// sort in LDS, and sums follow numpy's orders -- row-sequential for the axis-0 reduction of an
// [N,T>1] array, numpy's 8-lane pairwise tree when T == 1 (numpy collapses [N,1] to 1-D).
// Built with -ffp-contract=off.
#include "common.hpp"

namespace advgrpo {

// numpy's pairwise sum (loops_utils.h.src, PW_BLOCKSIZE = 128) over a contiguous f64 / f32 array
template <class F>
__device__ F np_pairwise_leaf(const F* a, int n) {  // n <= 128
    if (n < 8) {
        F r = 0;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    F r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    F res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}
template <class F>
__device__ F np_pairwise_sum(const F* a, int n) {
    if (n <= 128) return np_pairwise_leaf(a, n);
    // iterative form of the recursion: explicit stack of (ptr, n) halves, summed left to right
    // in the same association order as the recursive definition  f(a,n) = f(a,n2) + f(a+n2,n-n2)
    struct Fr { const F* a; int n; int state; F left; };
    Fr st[32];
    int sp = 0;
    st[0] = {a, n, 0, 0};
    F ret = 0;
    while (sp >= 0) {
        Fr& f = st[sp];
        if (f.n <= 128) { ret = np_pairwise_leaf(f.a, f.n); --sp; continue; }
        int n2 = f.n / 2; n2 -= n2 % 8;
        if (f.state == 0) { f.state = 1; st[++sp] = {f.a, n2, 0, 0}; }
        else if (f.state == 1) { f.left = ret; f.state = 2; st[++sp] = {f.a + n2, f.n - n2, 0, 0}; }
        else { ret = f.left + ret; --sp; }
    }
    return ret;
}

// sum of column j over rows [r0, r0+cnt) of vals[*, T] in numpy's order
__device__ double np_colsum(const double* vals, int T, int j, int r0, int cnt) {
    if (T == 1) return np_pairwise_sum(vals + r0, cnt);
    double s = 0.;
    for (int i = 0; i < cnt; ++i) s += vals[(int64_t)(r0 + i) * T + j];
    return s;
}

constexpr int GA_THREADS = 256;

// per-group np.std of column 0 in arithmetic type F, then (zero_std_ratio, mean std); all threads of the workgroup
template <class F>
__device__ void group_std_stats(const double* vals, double* s_col, double* s_sq, double* s_std, const int32_t* gid, const int* pos,
                                const int* gstart, const int* gcount, const int* leader, int N, int T, double* out_stats) {
    F* col = reinterpret_cast<F*>(s_col);     // column 0, rows permuted so that groups are contiguous
    F* sq = reinterpret_cast<F*>(s_sq);
    F* stds = reinterpret_cast<F*>(s_std);    // indexed by the group's rank in ascending key order
    for (int i = threadIdx.x; i < N; i += blockDim.x) col[pos[i]] = (F)vals[(size_t)pos[i] * T];
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        if (pos[i] != gstart[i]) continue;                 // one thread per group
        const int s = gstart[i], cnt = gcount[i];
        const F mean = np_pairwise_sum<F>(col + s, cnt) / (F)cnt;
        for (int k = 0; k < cnt; ++k) {
            const F x = col[s + k] - mean;
            sq[s + k] = x * x;
        }
        const F var = np_pairwise_sum<F>(sq + s, cnt) / (F)cnt;
        int rank = 0;
        for (int k = 0; k < N; ++k) rank += (leader[k] == k && gid[k] < gid[i]) ? 1 : 0;
        stds[rank] = sqrt(var);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int ng = 0, zero = 0;
        for (int k = 0; k < N; ++k) ng += leader[k] == k ? 1 : 0;
        for (int g = 0; g < ng; ++g) zero += stds[g] == (F)0 ? 1 : 0;
        out_stats[0] = (double)zero / (double)ng;
        out_stats[1] = (double)(np_pairwise_sum<F>(stds, ng) / (F)ng);
    }
}

// dynamic LDS (all f64 arrays are [N*T]):
//   vals  rewards permuted so each group's rows are contiguous (stable in row order)
//   dev   scratch: original-order copy, then squared deviations
//   gmean / gsd   per-group mean / std, stored at the group's first permuted row
//   gstd[T] global std;  pos / gstart / gcount / tmp : int[N]
__global__ __launch_bounds__(GA_THREADS) void group_advantage_kernel(
    const void* __restrict__ rewards, int r_dt, const int32_t* __restrict__ gid, int N, int T,
    int global_std, double* __restrict__ out, double* __restrict__ out_stats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const size_t nt = (size_t)N * T;
    double* vals = reinterpret_cast<double*>(smem);
    double* dev = vals + nt;
    double* gmean = dev + nt;
    double* gsd = gmean + nt;
    double* gstd = gsd + nt;
    int* pos = reinterpret_cast<int*>(gstd + T);
    int* gstart = pos + N;
    int* gcount = gstart + N;
    int* tmp = gcount + N;

    auto rd = [&](int i, int j) -> double {
        return r_dt == ADVGRPO_F64 ? reinterpret_cast<const double*>(rewards)[(int64_t)i * T + j]
                                   : (double)reinterpret_cast<const float*>(rewards)[(int64_t)i * T + j];
    };
    // 1. stable bucket by key.  leader = first row with the same key; rank = position inside
    //    the group; groups are laid out in order of their leader row.
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const int g = gid[i];
        int leader = i, rank = 0, cnt = 0;
        for (int k = 0; k < N; ++k) {
            if (gid[k] == g) {
                if (k < leader) leader = k;
                if (k < i) ++rank;
                ++cnt;
            }
        }
        tmp[i] = leader;
        gcount[i] = cnt;
        pos[i] = rank;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const int leader = tmp[i];
        int off = 0;  // rows that belong to groups whose leader comes earlier
        for (int k = 0; k < N; ++k) off += (tmp[k] < leader) ? 1 : 0;
        gstart[i] = off;
        pos[i] += off;
    }
    __syncthreads();
    for (int w = threadIdx.x; w < N * T; w += blockDim.x) {
        const int i = w / T, j = w % T;
        const double r = rd(i, j);
        vals[(size_t)pos[i] * T + j] = r;
        dev[w] = r;  // original order, for the global std
    }
    __syncthreads();
    // 2. global std per column: np.std(rewards, axis=0) = sqrt(mean((r - mean(r))^2))
    if (global_std) {
        if ((int)threadIdx.x < T) gstd[threadIdx.x] = np_colsum(dev, T, threadIdx.x, 0, N) / (double)N;
        __syncthreads();
        for (int w = threadIdx.x; w < N * T; w += blockDim.x) {
            const double d = dev[w] - gstd[w % T];
            dev[w] = d * d;
        }
        __syncthreads();
        if ((int)threadIdx.x < T)
            gstd[threadIdx.x] = sqrt(np_colsum(dev, T, threadIdx.x, 0, N) / (double)N) + 1e-4;
        __syncthreads();
    }
    // 3. per-group mean: one thread per (group leader, column)
    for (int w = threadIdx.x; w < N * T; w += blockDim.x) {
        const int i = w / T, j = w % T;
        if (pos[i] != gstart[i]) continue;
        gmean[(size_t)gstart[i] * T + j] = np_colsum(vals, T, j, gstart[i], gcount[i]) / (double)gcount[i];
    }
    __syncthreads();
    if (!global_std) {
        // 4. per-group std from squared deviations (permuted order, kept in dev)
        for (int w = threadIdx.x; w < N * T; w += blockDim.x) {
            const int i = w / T, j = w % T;
            const double d = vals[(size_t)pos[i] * T + j] - gmean[(size_t)gstart[i] * T + j];
            dev[(size_t)pos[i] * T + j] = d * d;
        }
        __syncthreads();
        for (int w = threadIdx.x; w < N * T; w += blockDim.x) {
            const int i = w / T, j = w % T;
            if (pos[i] != gstart[i]) continue;
            gsd[(size_t)gstart[i] * T + j] =
                sqrt(np_colsum(dev, T, j, gstart[i], gcount[i]) / (double)gcount[i]) + 1e-4;
        }
        __syncthreads();
    }
    for (int w = threadIdx.x; w < N * T; w += blockDim.x) {
        const int i = w / T, j = w % T;
        const double m = gmean[(size_t)gstart[i] * T + j];
        const double sd = global_std ? gstd[j] : gsd[(size_t)gstart[i] * T + j];
        out[w] = (vals[(size_t)pos[i] * T + j] - m) / sd;
    }
    // calculate_zero_std_ratio (train_sd3_fast_pickscore.py:195-229) on column 0 (the reference's 'ori_avg' is the
    // pre-repeat [N] reward): np.std of every group IN THE INPUT DTYPE (f32 as gathered: numpy keeps float32 through
    // mean / multiply / sum / sqrt), groups in ascending key order like np.unique, then the fraction of exact zeros and the
    // mean of the group stds (pairwise sums).  Rows inside a group keep their input order (np.argsort of the inverse
    // indices is stable for the sizes np.std's pairwise order is sensitive to; checked against the goldens).
    if (out_stats) {
        __syncthreads();
        if (r_dt == ADVGRPO_F64) group_std_stats<double>(vals, dev, gsd, gmean, gid, pos, gstart, gcount, tmp, N, T, out_stats);
        else group_std_stats<float>(vals, dev, gsd, gmean, gid, pos, gstart, gcount, tmp, N, T, out_stats);
    }
}

// ---- GRPO clipped surrogate, train_sd3_fast_pickscore.py:1111-1162
__global__ void grpo_loss_kernel(const float* __restrict__ lp, const float* __restrict__ old,
                                 const float* __restrict__ adv, int B, float adv_clip, float clip,
                                 float* __restrict__ scalars, float* __restrict__ grad) {
    __shared__ float red[5][4];
    float s_loss = 0.f, s_kl = 0.f, s_cf = 0.f, s_gt = 0.f, s_lt = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const float a = fminf(fmaxf(adv[i], -adv_clip), adv_clip);
        const float diff = lp[i] - old[i];
        const float ratio = expf(diff);
        const float lo = 1.0f - clip, hi = 1.0f + clip;
        const float rc = fminf(fmaxf(ratio, lo), hi);
        const float un = -a * ratio, cl = -a * rc;
        s_loss += fmaxf(un, cl);
        s_kl += diff * diff;
        s_cf += (fabsf(ratio - 1.0f) > clip) ? 1.f : 0.f;
        s_gt += (ratio - 1.0f > clip) ? 1.f : 0.f;
        s_lt += (1.0f - ratio > clip) ? 1.f : 0.f;
        if (grad) {
            // torch.maximum routes the gradient to the larger branch (half/half on ties); the
            // clamp passes gradient inside [lo, hi] inclusive.
            const bool inside = ratio >= lo && ratio <= hi;
            float w_un = un > cl ? 1.f : (un == cl ? 0.5f : 0.f);
            float w_cl = 1.f - w_un;
            if (un != un || cl != cl) { w_un = 1.f; w_cl = 1.f; }  // NaN propagates
            const float g = (w_un * (-a * ratio) + (inside ? w_cl * (-a * ratio) : 0.f)) / (float)B;
            grad[i] = g;
        }
    }
    float v[5] = {s_loss, s_kl, s_cf, s_gt, s_lt};
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        v[k] = wave_sum(v[k]);
        if (lane == 0) red[k][w] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t[5];
        for (int k = 0; k < 5; ++k) t[k] = red[k][0] + red[k][1] + red[k][2] + red[k][3];
        const float inv = 1.0f / (float)B;
        scalars[0] = t[0] * inv;         // loss (beta == 0)
        scalars[1] = 0.5f * t[1] * inv;  // approx_kl
        scalars[2] = t[2] * inv;         // clipfrac
        scalars[3] = t[3] * inv;         // clipfrac_gt_one
        scalars[4] = t[4] * inv;         // clipfrac_lt_one
        scalars[5] = t[0] * inv;         // policy_loss
    }
}

}  // namespace advgrpo

using namespace advgrpo;

static int group_advantage_launch(const void* rewards, int rewards_dtype, const int32_t* group_id, int N, int T,
                                  int global_std, double* out_adv, double* out_stats, void* stream) {
    ADVGRPO_CHECK(rewards && group_id && out_adv, "group_advantage: null argument");
    ADVGRPO_CHECK(rewards_dtype == ADVGRPO_F32 || rewards_dtype == ADVGRPO_F64, "group_advantage: bad dtype %d",
                  rewards_dtype);
    ADVGRPO_CHECK(N > 0 && T > 0, "group_advantage: need N>0, T>0 (N=%d T=%d)", N, T);
    const size_t nt = (size_t)N * T;
    // vals, dev, gmean, gsd (f64 [N*T] each) + gstd[T] + 4 int[N]
    const size_t bytes = 4 * nt * 8 + (size_t)T * 8 + (size_t)4 * N * 4;
    ADVGRPO_CHECK(bytes <= 160 * 1024, "group_advantage: N*T=%zu exceeds the single-workgroup LDS budget", nt);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(group_advantage_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    hipLaunchKernelGGL(group_advantage_kernel, dim3(1), dim3(GA_THREADS), bytes, as_stream(stream), rewards,
                       rewards_dtype, group_id, N, T, global_std, out_adv, out_stats);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

extern "C" int advgrpo_group_advantage(const void* rewards, int rewards_dtype, const int32_t* group_id, int N, int T,
                                       int global_std, double* out_adv, void* stream) {
    return group_advantage_launch(rewards, rewards_dtype, group_id, N, T, global_std, out_adv, nullptr, stream);
}

extern "C" int advgrpo_group_advantage_stats(const void* rewards, int rewards_dtype, const int32_t* group_id, int N, int T,
                                             int global_std, double* out_adv, double* out_stats, void* stream) {
    ADVGRPO_CHECK(out_stats, "group_advantage_stats: null out_stats");
    return group_advantage_launch(rewards, rewards_dtype, group_id, N, T, global_std, out_adv, out_stats, stream);
}

extern "C" int advgrpo_grpo_loss(const float* log_prob, const float* old_log_prob, const float* advantages, int B,
                                 float adv_clip_max, float clip_range, float* out_scalars, float* out_grad_log_prob,
                                 void* stream) {
    ADVGRPO_CHECK(log_prob && old_log_prob && advantages && out_scalars, "grpo_loss: null argument");
    ADVGRPO_CHECK(B > 0, "grpo_loss: B must be positive");
    hipLaunchKernelGGL(grpo_loss_kernel, dim3(1), dim3(256), 0, as_stream(stream), log_prob, old_log_prob,
                       advantages, B, adv_clip_max, clip_range, out_scalars, out_grad_log_prob);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}
