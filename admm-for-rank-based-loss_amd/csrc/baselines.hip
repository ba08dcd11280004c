// baselines.hip - the reference's competitor baselines on the device:
//   SGDmethod   (SGD_solver.py:9-96)   -> StochasticSubgradientMethod, existing_methods/lerm_main/src/optim/algorithms.py:54-98
//   LSVRGmethod (LSVRG_solver.py:9-98) -> LSVRG,                       existing_methods/lerm_main/src/optim/algorithms.py:150-253
// on the competitor's objective (existing_methods/lerm_main/src/optim/objective.py:41-112): risk = dot(alphas,
// sort(losses)), differentiated in closed form:  d risk / d w = sum_i a_{rank(i)} loss_i'(x_i . w) x_i  (stable
// ranks; EHRM: alpha or beta by the sorted loss against lossB).  The reference's peculiarities are kept (labels in
// {0, 1} for the hinge loss too, b-sample weights for a mini-batch of b rows, alphas[row index] in the uniform
// LSVRG step, one float32 random sign per step for the l1 subgradient at 0); CPU restatement, pinned by golden
// vectors of the real reference: oracle/baselines.py.
//
// These are sequential small-step methods (64 rows x d per SGD step, ONE row per LSVRG step): a step is a few
// microseconds of work for one workgroup, so an epoch (100 steps) is ONE launch of a single-workgroup kernel that
// loops over the steps with w in global memory (the workgroup is its only reader and writer); only LSVRG's
// checkpoint (full-batch losses, stable sort, X^T c) uses the whole chip, through the library's sweep and sort
// kernels.  The random index / sign streams are the reference's own host generators (torch.randperm, RandomState,
// numpy.random.choice, torch.rand) and arrive as data.
#include "rbl_internal.h"
#include "device_math.h"

#include <cstring>
#include <vector>

struct rbl_baseline {
    int64_t n = 0, d = 0, ld = 0;
    int loss = 0, has_B = 0, device = 0, num_cu = 256;
    double lossB = 0.0, l2 = 0.0, l1 = 0.0;
    hipStream_t stream = nullptr;
    double *X = nullptr, *y01 = nullptr, *w = nullptr, *w_chk = nullptr, *g_chk = nullptr;
    double *z = nullptr, *c = nullptr, *alphas = nullptr, *betas = nullptr, *slab = nullptr;
    double *ab = nullptr, *bb = nullptr;   // mini-batch weights (<= 1024)
    int* idx = nullptr;                    // order / samples of one epoch
    float* rands = nullptr;
    size_t idx_cap = 0;
    SortWorkspace sw{};
};

namespace {

constexpr int BL_THREADS = 256;
constexpr int BL_MAX_BATCH = 1024;   // one thread of the SGD workgroup per mini-batch row (the reference's drivers use 64; SGD_solver.py:9 takes any size)

template <int LOSS>
__device__ inline double bl_loss(double z, double y) {
    // competitor objective.py:11-15 (BCE with logits) / :36-37 (hinge on y in {0, 1}, SGD_solver.py:13-14)
    return LOSS == 0 ? rbl::softplus(z) - y * z : fmax(1.0 - y * z, 0.0);
}
template <int LOSS>
__device__ inline double bl_dloss(double z, double y) {
    if (LOSS == 0) return rbl::sigmoid1(z) - y;
    const double t = 1.0 - y * z;
    return t > 0.0 ? -y : (t == 0.0 ? -0.5 * y : 0.0);   // torch.maximum splits the gradient at a tie
}
// the regulariser's share of a direction (objective.py:97-106): l2 w / n and the float32 sign term of l1
__device__ inline double bl_reg(double wj, double l2, double l1, double n, float rnd) {
    double g = 0.0;
    if (l2 != 0.0) g += l2 * wj / n;
    if (l1 != 0.0) {
        float res = wj > 0.0 ? 1.0f : (wj < 0.0 ? -1.0f : rnd * 2.0f - 1.0f);
        g += (double)((res * (float)l1) / (float)(2.0 * n));
    }
    return g;
}

// one epoch of StochasticSubgradientMethod.step (algorithms.py:84-93): `steps` mini-batches of `batch` rows.
// One workgroup of 16 waves; a step is latency (two dependent phases over 64 rows x d), so both phases keep
// several independent loads in flight: a wave computes 4 logits at once, a thread accumulates its column over
// 8 rows at a time.
constexpr int BL_SGD_THREADS = 1024;
template <int LOSS>
__global__ __launch_bounds__(BL_SGD_THREADS) void k_bl_sgd_epoch(const double* __restrict__ X, long long ld, long long d,
                                                                  long long n, const double* __restrict__ y01, double* w,
                                                                  const int* __restrict__ order, int steps, int batch,
                                                                  const double* __restrict__ ab, const double* __restrict__ bb,
                                                                  int has_B, double lossB, double lr, double l2, double l1,
                                                                  const float* __restrict__ rands) {
    __shared__ double s_z[BL_MAX_BATCH], s_l[BL_MAX_BATCH], s_c[BL_MAX_BATCH];
    __shared__ long long s_off[BL_MAX_BATCH];                // row offsets into X
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = BL_SGD_THREADS / 64;
    for (int s = 0; s < steps; ++s) {
        long long b0 = (long long)s * batch, b1 = b0 + batch;
        if (b1 > n) b1 = n;
        const int b = (int)(b1 - b0);                        // rows of this mini-batch (algorithms.py:85-88)
        if (tid < b) s_off[tid] = (long long)order[b0 + tid] * ld;
        __syncthreads();
        for (int r0 = wave * 4; r0 < b; r0 += NW * 4) {      // logits: 4 rows per wave at a time
            const double* x0 = X + s_off[r0];
            const double* x1 = X + s_off[r0 + 1 < b ? r0 + 1 : r0];
            const double* x2 = X + s_off[r0 + 2 < b ? r0 + 2 : r0];
            const double* x3 = X + s_off[r0 + 3 < b ? r0 + 3 : r0];
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            for (long long j = lane; j < d; j += 64) {
                const double wj = w[j];
                a0 = __builtin_fma(x0[j], wj, a0);
                a1 = __builtin_fma(x1[j], wj, a1);
                a2 = __builtin_fma(x2[j], wj, a2);
                a3 = __builtin_fma(x3[j], wj, a3);
            }
            a0 = rbl::wave_sum_all(a0);
            a1 = rbl::wave_sum_all(a1);
            a2 = rbl::wave_sum_all(a2);
            a3 = rbl::wave_sum_all(a3);
            if (lane == 0) {
                s_z[r0] = a0;
                if (r0 + 1 < b) s_z[r0 + 1] = a1;
                if (r0 + 2 < b) s_z[r0 + 2] = a2;
                if (r0 + 3 < b) s_z[r0 + 3] = a3;
            }
        }
        __syncthreads();
        if (tid < b) s_l[tid] = bl_loss<LOSS>(s_z[tid], y01[s_off[tid] / ld]);
        __syncthreads();
        if (tid < b) {
            const double lt = s_l[tid];
            int rank = 0;                                    // stable rank: ties keep the batch order
            for (int j = 0; j < b; ++j) rank += (s_l[j] < lt || (s_l[j] == lt && j < tid)) ? 1 : 0;
            const double wk = (has_B && !(lt <= lossB)) ? bb[rank] : ab[rank];   // objective.py:84-88
            s_c[tid] = wk * bl_dloss<LOSS>(s_z[tid], y01[s_off[tid] / ld]);
        }
        __syncthreads();
        const float rnd = (l1 != 0.0 && rands) ? rands[s] : 0.0f;
        for (long long j = tid; j < d; j += BL_SGD_THREADS) {
            double g = 0.0;
            int r = 0;
            for (; r + 8 <= b; r += 8) {                     // 8 independent loads in flight, summed in row order
                double xv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) xv[k] = X[s_off[r + k] + j];
#pragma unroll
                for (int k = 0; k < 8; ++k) g = __builtin_fma(s_c[r + k], xv[k], g);
            }
            for (; r < b; ++r) g = __builtin_fma(s_c[r], X[s_off[r] + j], g);
            const double wj = w[j];
            w[j] = wj - lr * (g + bl_reg(wj, l2, l1, (double)n, rnd));
        }
        __syncthreads();
    }
}

// per-row loss of the whole data set at w (z = X w given): keys for the stable sort
template <int LOSS>
__global__ void k_bl_losses(long long n, const double* __restrict__ z, const double* __restrict__ y01,
                            double* __restrict__ losses) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        losses[i] = bl_loss<LOSS>(z[i], y01[i]);
}

// sorted position k -> coefficient of its row in the checkpoint subgradient (algorithms.py:186-194)
template <int LOSS>
__global__ void k_bl_rank_coef(long long n, const u64* __restrict__ keys_sorted, const u32* __restrict__ perm,
                               const double* __restrict__ alphas, const double* __restrict__ betas, int has_B,
                               double lossB, const double* __restrict__ z, const double* __restrict__ y01,
                               double* __restrict__ c) {
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        const long long row = perm[k];
        const double srt = rbl::unflip_key(keys_sorted[k]);
        const double wk = (has_B && !(srt <= lossB)) ? betas[k] : alphas[k];
        c[row] = wk * bl_dloss<LOSS>(z[row], y01[row]);
    }
}

// `steps` LSVRG.step calls (algorithms.py:198-253); samples[s]: a row (uniform) or a rank of the checkpoint order
template <int LOSS>
__global__ __launch_bounds__(BL_THREADS) void k_bl_lsvrg_steps(const double* __restrict__ X, long long ld, long long d,
                                                                long long n, const double* __restrict__ y01, double* w,
                                                                const double* __restrict__ w_chk,
                                                                const double* __restrict__ g_chk,
                                                                const u32* __restrict__ perm, const int* __restrict__ samples,
                                                                int steps, int uniform, const double* __restrict__ alphas,
                                                                const double* __restrict__ betas, int has_B, double lossB,
                                                                double lr, double l2, double l1,
                                                                const float* __restrict__ rands) {
    __shared__ double smem[2 * BL_THREADS / 64];
    const int tid = threadIdx.x;
    for (int s = 0; s < steps; ++s) {
        const int i = samples[s];
        const long long row = uniform ? (long long)i : (long long)perm[i];
        const double* x = X + row * ld;
        double acc[2] = {0.0, 0.0};
        for (long long j = tid; j < d; j += BL_THREADS) {
            const double xj = x[j];
            acc[0] = __builtin_fma(xj, w[j], acc[0]);
            acc[1] = __builtin_fma(xj, w_chk[j], acc[1]);
        }
        rbl::block_sum<2, BL_THREADS>(acc, smem);
        const double yy = y01[row];
        const double diff = bl_dloss<LOSS>(acc[0], yy) - bl_dloss<LOSS>(acc[1], yy);
        double scale = 1.0;                                            // non-uniform: g - g_chk + subgrad (:240)
        if (uniform) {
            const bool use_beta = has_B && !(bl_loss<LOSS>(acc[0], yy) <= lossB);
            scale = (double)n * (use_beta ? betas[i] : alphas[i]);     // alphas[ROW index] (:232-238)
        }
        const float rnd = (l1 != 0.0 && rands) ? rands[s] : 0.0f;
        __syncthreads();    // every thread has its sums before w changes
        for (long long j = tid; j < d; j += BL_THREADS) {
            const double wj = w[j];
            const double dir = scale * diff * x[j] + g_chk[j] + bl_reg(wj, l2, l1, (double)n, rnd);
            w[j] = wj - lr * dir;
        }
        __syncthreads();
    }
}

template <typename T>
int bl_alloc(T** p, size_t count) {
    *p = nullptr;
    if (hipMalloc((void**)p, (count ? count : 1) * sizeof(T)) != hipSuccess) {
        rbl_set_error("baselines: hipMalloc of %zu bytes failed", count * sizeof(T));
        return RBL_ERR_NOMEM;
    }
    return RBL_OK;
}

int bl_ensure_idx(rbl_baseline* h, size_t count) {
    if (count <= h->idx_cap) return RBL_OK;
    if (h->idx) (void)hipFree(h->idx);
    if (h->rands) (void)hipFree(h->rands);
    h->idx = nullptr;
    h->rands = nullptr;
    RBL_TRY(bl_alloc(&h->idx, count));
    RBL_TRY(bl_alloc(&h->rands, count));
    h->idx_cap = count;
    return RBL_OK;
}

}  // namespace

extern "C" {

int rbl_bl_destroy(rbl_baseline* h) {
    if (!h) return RBL_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void* ptrs[] = {h->X, h->y01, h->w, h->w_chk, h->g_chk, h->z, h->c, h->alphas, h->betas, h->slab, h->ab, h->bb, h->idx,
                    h->rands, h->sw.keys[0], h->sw.keys[1], h->sw.vals[0], h->sw.vals[1], h->sw.spine, h->sw.bin_total,
                    h->sw.bin_base, h->sw.ghist};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return RBL_OK;
}

int rbl_bl_create(int64_t n, int64_t d, const double* X, const double* y01, int loss, int has_lossB, double lossB,
                  double l2_reg, double l1_reg, int device, rbl_baseline** out) {
    if (!out || !X || !y01 || n <= 0 || d <= 0 || n >= (1LL << 31) || (loss != RBL_LOSS_BCE && loss != RBL_LOSS_HINGE)) {
        rbl_set_error("baselines: bad arguments (n=%lld d=%lld loss=%d)", (long long)n, (long long)d, loss);
        return RBL_ERR_INVALID;
    }
    *out = nullptr;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) {
        (void)hipGetLastError();
        rbl_set_error("no HIP device available (librbl has no CPU fallback)");
        return RBL_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= cnt) {
        rbl_set_error("device %d out of range (%d devices)", device, cnt);
        return RBL_ERR_INVALID;
    }
    RBL_HIP(hipSetDevice(device));
    rbl_baseline* h = new rbl_baseline();
    h->n = n;
    h->d = d;
    h->ld = (d + 3) / 4 * 4;
    h->loss = loss;
    h->has_B = has_lossB ? 1 : 0;
    h->lossB = lossB;
    h->l2 = l2_reg;
    h->l1 = l1_reg;
    h->device = device;
    hipDeviceProp_t prop;
    int rc = RBL_OK;
    do {
        if (hipGetDeviceProperties(&prop, device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
            rbl_set_error("baselines: device set-up failed");
            rc = RBL_ERR_HIP;
            break;
        }
        h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        const size_t nn = (size_t)n, ld = (size_t)h->ld;
        if ((rc = bl_alloc(&h->X, nn * ld)) || (rc = bl_alloc(&h->y01, nn)) || (rc = bl_alloc(&h->w, ld)) ||
            (rc = bl_alloc(&h->w_chk, ld)) || (rc = bl_alloc(&h->g_chk, ld)) || (rc = bl_alloc(&h->z, nn)) ||
            (rc = bl_alloc(&h->c, nn)) || (rc = bl_alloc(&h->alphas, nn)) || (rc = bl_alloc(&h->betas, nn)) ||
            (rc = bl_alloc(&h->ab, (size_t)BL_MAX_BATCH)) || (rc = bl_alloc(&h->bb, (size_t)BL_MAX_BATCH)) ||
            (rc = bl_alloc(&h->slab, (size_t)gemvt_slab_rows(h->num_cu) * ld * 2)) || (rc = bl_alloc(&h->sw.keys[0], nn)) ||
            (rc = bl_alloc(&h->sw.keys[1], nn)) || (rc = bl_alloc(&h->sw.vals[0], nn)) || (rc = bl_alloc(&h->sw.vals[1], nn)) ||
            (rc = bl_alloc((unsigned char**)&h->sw.spine, sort_spine_bytes())) || (rc = bl_alloc(&h->sw.bin_total, (size_t)256)) ||
            (rc = bl_alloc(&h->sw.bin_base, (size_t)256)) || (rc = bl_alloc((unsigned char**)&h->sw.ghist, sort_ghist_bytes())))
            break;
        bool ok = hipMemset(h->sw.ghist, 0, sort_ghist_bytes()) == hipSuccess && hipMemset(h->w, 0, sizeof(double) * ld) == hipSuccess &&
                  hipMemset(h->X, 0, sizeof(double) * nn * ld) == hipSuccess;
        ok = ok && hipMemcpy2D(h->X, sizeof(double) * ld, X, sizeof(double) * (size_t)d, sizeof(double) * (size_t)d, nn,
                               hipMemcpyHostToDevice) == hipSuccess;
        ok = ok && hipMemcpy(h->y01, y01, sizeof(double) * nn, hipMemcpyHostToDevice) == hipSuccess;
        if (!ok) {
            rbl_set_error("baselines: upload failed: %s", hipGetErrorString(hipGetLastError()));
            rc = RBL_ERR_HIP;
        }
    } while (0);
    if (rc != RBL_OK) {
        rbl_bl_destroy(h);
        return rc;
    }
    *out = h;
    return RBL_OK;
}

int rbl_bl_set_w(rbl_baseline* h, const double* w) {
    if (!h || !w) return RBL_ERR_INVALID;
    RBL_HIP(hipSetDevice(h->device));
    RBL_HIP(hipStreamSynchronize(h->stream));
    RBL_HIP(hipMemset(h->w, 0, sizeof(double) * h->ld));
    RBL_HIP(hipMemcpy(h->w, w, sizeof(double) * h->d, hipMemcpyHostToDevice));
    return RBL_OK;
}

int rbl_bl_get_w(rbl_baseline* h, double* w) {
    if (!h || !w) return RBL_ERR_INVALID;
    RBL_HIP(hipSetDevice(h->device));
    RBL_HIP(hipStreamSynchronize(h->stream));
    RBL_HIP(hipMemcpy(w, h->w, sizeof(double) * h->d, hipMemcpyDeviceToHost));
    return RBL_OK;
}

int rbl_bl_sgd_epoch(rbl_baseline* h, const int32_t* order, int steps, int batch, const double* alphas_b,
                     const double* betas_b, double lr, const float* rands) {
    if (!h || !order || !alphas_b || steps < 0 || batch < 1 || batch > BL_MAX_BATCH || (h->has_B && !betas_b)) {
        rbl_set_error("sgd_epoch: bad arguments (batch must be 1..%d)", BL_MAX_BATCH);
        return RBL_ERR_INVALID;
    }
    if ((long long)(steps - 1) * batch >= h->n && steps > 0) {
        rbl_set_error("sgd_epoch: %d mini-batches of %d rows exceed the %lld rows", steps, batch, (long long)h->n);
        return RBL_ERR_INVALID;
    }
    RBL_HIP(hipSetDevice(h->device));
    if (steps == 0) return RBL_OK;
    hipStream_t s = h->stream;
    long long cnt = (long long)steps * batch;
    if (cnt > h->n) cnt = h->n;
    for (long long i = 0; i < cnt; ++i)
        if (order[i] < 0 || order[i] >= h->n) {
            rbl_set_error("sgd_epoch: row index %d out of range", order[i]);
            return RBL_ERR_INVALID;
        }
    RBL_TRY(bl_ensure_idx(h, (size_t)(cnt > steps ? cnt : steps)));
    RBL_HIP(hipMemcpyAsync(h->idx, order, sizeof(int) * (size_t)cnt, hipMemcpyHostToDevice, s));
    RBL_HIP(hipMemcpyAsync(h->ab, alphas_b, sizeof(double) * batch, hipMemcpyHostToDevice, s));
    if (betas_b) RBL_HIP(hipMemcpyAsync(h->bb, betas_b, sizeof(double) * batch, hipMemcpyHostToDevice, s));
    if (rands) RBL_HIP(hipMemcpyAsync(h->rands, rands, sizeof(float) * steps, hipMemcpyHostToDevice, s));
    if (h->loss == RBL_LOSS_BCE)
        hipLaunchKernelGGL(k_bl_sgd_epoch<0>, dim3(1), dim3(BL_SGD_THREADS), 0, s, h->X, (long long)h->ld, (long long)h->d,
                           (long long)h->n, h->y01, h->w, h->idx, steps, batch, h->ab, h->bb, h->has_B, h->lossB, lr, h->l2, h->l1,
                           rands ? h->rands : (const float*)nullptr);
    else
        hipLaunchKernelGGL(k_bl_sgd_epoch<1>, dim3(1), dim3(BL_SGD_THREADS), 0, s, h->X, (long long)h->ld, (long long)h->d,
                           (long long)h->n, h->y01, h->w, h->idx, steps, batch, h->ab, h->bb, h->has_B, h->lossB, lr, h->l2, h->l1,
                           rands ? h->rands : (const float*)nullptr);
    RBL_HIP(hipGetLastError());
    RBL_HIP(hipStreamSynchronize(s));   // the caller's arrays may go away
    return RBL_OK;
}

int rbl_bl_lsvrg_epoch(rbl_baseline* h, const double* alphas, const double* betas, const int32_t* samples, int steps,
                       int uniform, double lr, const float* rands) {
    if (!h || !alphas || !samples || steps < 0 || (h->has_B && !betas)) {
        rbl_set_error("lsvrg_epoch: bad arguments");
        return RBL_ERR_INVALID;
    }
    for (int i = 0; i < steps; ++i)
        if (samples[i] < 0 || samples[i] >= h->n) {
            rbl_set_error("lsvrg_epoch: sample %d out of range", samples[i]);
            return RBL_ERR_INVALID;
        }
    RBL_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const long long n = h->n, ld = h->ld;
    RBL_HIP(hipMemcpyAsync(h->alphas, alphas, sizeof(double) * n, hipMemcpyHostToDevice, s));
    if (betas) RBL_HIP(hipMemcpyAsync(h->betas, betas, sizeof(double) * n, hipMemcpyHostToDevice, s));
    RBL_TRY(bl_ensure_idx(h, (size_t)(steps > 0 ? steps : 1)));
    if (steps > 0) RBL_HIP(hipMemcpyAsync(h->idx, samples, sizeof(int) * steps, hipMemcpyHostToDevice, s));
    if (rands && steps > 0) RBL_HIP(hipMemcpyAsync(h->rands, rands, sizeof(float) * steps, hipMemcpyHostToDevice, s));
    // checkpoint (LSVRG.start_epoch, algorithms.py:183-196): losses of all rows, their stable order, the full-batch
    // subgradient X^T c with c_i = weight(rank i) * loss_i' - the only part of these baselines that uses the whole chip
    RBL_TRY(launch_gemv(RBL_STORE_F64, h->X, n, ld, h->w, h->z, h->num_cu, s));
    double* losses = (double*)h->sw.keys[1];   // scratch until the sort needs it
    const unsigned g = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (h->loss == RBL_LOSS_BCE) hipLaunchKernelGGL(k_bl_losses<0>, dim3(g), dim3(256), 0, s, n, h->z, h->y01, losses);
    else hipLaunchKernelGGL(k_bl_losses<1>, dim3(g), dim3(256), 0, s, n, h->z, h->y01, losses);
    RBL_TRY(launch_keys_from_m(n, losses, h->sw.keys[0], h->sw.vals[0], s));
    RBL_TRY(launch_radix_sort(h->sw, n, true, s));
    if (h->loss == RBL_LOSS_BCE)
        hipLaunchKernelGGL(k_bl_rank_coef<0>, dim3(g), dim3(256), 0, s, n, h->sw.keys[0], h->sw.vals[0], h->alphas, h->betas, h->has_B,
                           h->lossB, h->z, h->y01, h->c);
    else
        hipLaunchKernelGGL(k_bl_rank_coef<1>, dim3(g), dim3(256), 0, s, n, h->sw.keys[0], h->sw.vals[0], h->alphas, h->betas, h->has_B,
                           h->lossB, h->z, h->y01, h->c);
    RBL_TRY(launch_gemvt(RBL_STORE_F64, h->X, n, ld, h->c, h->slab, h->g_chk, h->num_cu, s));
    RBL_HIP(hipMemcpyAsync(h->w_chk, h->w, sizeof(double) * ld, hipMemcpyDeviceToDevice, s));
    if (steps > 0) {
        if (h->loss == RBL_LOSS_BCE)
            hipLaunchKernelGGL(k_bl_lsvrg_steps<0>, dim3(1), dim3(BL_THREADS), 0, s, h->X, ld, (long long)h->d, n, h->y01, h->w, h->w_chk,
                               h->g_chk, h->sw.vals[0], h->idx, steps, uniform ? 1 : 0, h->alphas, h->betas, h->has_B, h->lossB, lr,
                               h->l2, h->l1, rands ? h->rands : (const float*)nullptr);
        else
            hipLaunchKernelGGL(k_bl_lsvrg_steps<1>, dim3(1), dim3(BL_THREADS), 0, s, h->X, ld, (long long)h->d, n, h->y01, h->w, h->w_chk,
                               h->g_chk, h->sw.vals[0], h->idx, steps, uniform ? 1 : 0, h->alphas, h->betas, h->has_B, h->lossB, lr,
                               h->l2, h->l1, rands ? h->rands : (const float*)nullptr);
    }
    RBL_HIP(hipGetLastError());
    RBL_HIP(hipStreamSynchronize(s));
    return RBL_OK;
}

}  // extern "C"
