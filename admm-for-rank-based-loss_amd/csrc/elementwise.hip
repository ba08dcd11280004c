// elementwise.hip - n-vector kernels of the ADMM iteration: element prox (z-step for
// erm), m = D w - lambda/rho, dual update + residual/objective partial sums, sigma
// generators, and the objective from the cached v = D w.  All reductions are two-stage
// with a fixed summation order (deterministic, no float atomics).
#include "rbl_internal.h"
#include "device_math.h"

namespace {

constexpr int EW_THREADS = 256;
constexpr int RED_BLOCKS = 1024;  // fixed: partial sums do not depend on n or the device

inline unsigned ew_grid(long long n) {
    long long g = (n + EW_THREADS - 1) / EW_THREADS;
    if (g < 1) g = 1;
    if (g > 1 << 20) g = 1 << 20;
    return (unsigned)g;
}

// erm: sigma is constant, the prox is monotone in m, so no sort and no PAV are needed:
// z_i = prox(m_i) is the isotonic solution.  Fuses algorithms.py:89
// (m), individual_solver.py:112-123 (prox) and the w-step's c = z + lambda/rho (:192).
template <int LOSS>
__global__ __launch_bounds__(EW_THREADS) void k_erm_zc(long long n, double sigma0, double rho,
                                                        const double* __restrict__ v,
                                                        const double* __restrict__ lam, double* __restrict__ m,
                                                        double* __restrict__ z, double* __restrict__ c) {
    const double inv_rho = 1.0 / rho;
    for (long long i = (long long)blockIdx.x * EW_THREADS + threadIdx.x; i < n;
         i += (long long)gridDim.x * EW_THREADS) {
        double lr = lam[i] / rho;
        double mi = v[i] - lr;
        double zi = rbl::prox<LOSS>(sigma0, rho, mi);
        m[i] = mi;
        z[i] = zi;
        c[i] = zi + lr;
    }
    (void)inv_rho;
}

__global__ void k_make_m_keys(long long n, double rho, const double* __restrict__ v, const double* __restrict__ lam,
                              double* __restrict__ m, u64* __restrict__ keys, u32* __restrict__ idx, u32 idx_off) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const double x = v[i] - lam[i] / rho;  // algorithms.py:89
        m[i] = x;
        keys[i] = rbl::flip_key(x);
        idx[i] = (u32)i + idx_off;
    }
}

// m = v - lambda/rho (algorithms.py:89) and the range of m for the 32-bit sort keys: every block leaves the smallest
// and the largest order-preserving key of its rows in mm[2 b], mm[2 b + 1] (an integer min / max: order independent);
// k_keys32 reduces those S32_RANGE_BLOCKS pairs again in every one of its blocks.  (Atomic max on two global words
// from every wave was tried first: 16 384 contended atomics = 380 us at 6.25 M rows.)
constexpr int S32_RANGE_BLOCKS = 1024;
__global__ __launch_bounds__(EW_THREADS) void k_make_m_range(long long n, double rho, const double* __restrict__ v,
                                                              const double* __restrict__ lam, double* __restrict__ m,
                                                              u64* __restrict__ mm) {
    __shared__ u64 s_lo[EW_THREADS / 64], s_hi[EW_THREADS / 64];
    u64 lo = ~0ull, hi = 0ull;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double x = v[i] - lam[i] / rho;
        m[i] = x;
        const u64 k = rbl::flip_key(x);
        lo = k < lo ? k : lo;
        hi = k > hi ? k : hi;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const u64 a = __shfl_xor(lo, off, 64), b = __shfl_xor(hi, off, 64);
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        s_lo[threadIdx.x >> 6] = lo;
        s_hi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < EW_THREADS / 64; ++w) {
            lo = s_lo[w] < lo ? s_lo[w] : lo;
            hi = s_hi[w] > hi ? s_hi[w] : hi;
        }
        mm[2 * blockIdx.x] = lo;          // (a block without rows leaves the neutral pair)
        mm[2 * blockIdx.x + 1] = hi;
    }
}

// 32-bit sort keys: the monotone fixed-point image of m on [min m, max m] (a subtraction, a multiplication by a
// positive constant and a truncation, each non-decreasing under rounding), payload = global row id.  Rows whose m the
// 32 bits cannot tell apart (spacing of the images: range / 2^32) come out of the stable sort in row order; k_sort32_fix
// puts such runs in (m, row) order.  A degenerate range (all m equal, or not finite) gives every row key 0: one run of
// n rows, which the fix-up reports - the caller then sorts 64-bit keys.
__global__ __launch_bounds__(256) void k_keys32(long long n, const double* __restrict__ m, const u64* __restrict__ mm, int nparts,
                                                u32* __restrict__ keys, u32* __restrict__ idx, u32 idx_off) {
    __shared__ u64 s_lo[4], s_hi[4];
    u64 klo = ~0ull, khi = 0ull;
    for (int p = threadIdx.x; p < nparts; p += 256) {
        const u64 a = mm[2 * p], b = mm[2 * p + 1];
        klo = a < klo ? a : klo;
        khi = b > khi ? b : khi;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const u64 a = __shfl_xor(klo, off, 64), b = __shfl_xor(khi, off, 64);
        klo = a < klo ? a : klo;
        khi = b > khi ? b : khi;
    }
    if ((threadIdx.x & 63) == 0) {
        s_lo[threadIdx.x >> 6] = klo;
        s_hi[threadIdx.x >> 6] = khi;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        klo = s_lo[w] < klo ? s_lo[w] : klo;
        khi = s_hi[w] > khi ? s_hi[w] : khi;
    }
    const double lo = rbl::unflip_key(klo), hi = rbl::unflip_key(khi);
    double scale = 4294967295.0 / (hi - lo);
    if (!(hi > lo) || !(scale < 1.7e308)) scale = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double t = (m[i] - lo) * scale;
        keys[i] = t >= 4294967295.0 ? 0xffffffffu : (t > 0.0 ? (u32)t : 0u);
        idx[i] = (u32)i + idx_off;
    }
}

// After the 32-bit sort: ms[p] = m of the row at sorted position p, ids_out[p] = its row id, with every run of equal
// 32-bit keys rearranged into (m, row id) order - each element of a run counts the run's elements that precede it in
// that order (runs are pairs, rarely triples: the images of neighbouring m differ by ~2^32 / n).  A run longer than
// S32_MAX_RUN sets *flag; the values written for it are then meaningless and the caller redoes the z-step with 64-bit
// keys.  (Equal m keep their row order, as in the stable 64-bit sort: SURVEY 3.4-e.)
constexpr int S32_MAX_RUN = 32;
__global__ void k_sort32_fix(long long n, const u32* __restrict__ keys, const u32* __restrict__ ids,
                             const double* __restrict__ m, u32 off, double* __restrict__ ms, u32* __restrict__ ids_out,
                             int* __restrict__ flag) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const u32 k = keys[i], id = ids[i];
        const double x = m[id - off];
        const bool prev = i > 0 && keys[i - 1] == k, next = i + 1 < n && keys[i + 1] == k;
        if (!prev && !next) {
            ms[i] = x;
            ids_out[i] = id;
            continue;
        }
        long long s0 = i, e0 = i;   // the run [s0, e0]
        while (s0 > 0 && i - s0 < S32_MAX_RUN && keys[s0 - 1] == k) --s0;
        while (e0 + 1 < n && e0 - i < S32_MAX_RUN && keys[e0 + 1] == k) ++e0;
        if (e0 - s0 + 1 > S32_MAX_RUN) {
            *flag = 1;
            ms[i] = x;
            ids_out[i] = id;
            continue;
        }
        int rank = 0;
        for (long long j = s0; j <= e0; ++j) {
            if (j == i) continue;
            const u32 idj = ids[j];
            const double xj = m[idj - off];
            rank += (xj < x || (xj == x && idj < id)) ? 1 : 0;
        }
        ms[s0 + rank] = x;
        ids_out[s0 + rank] = id;
    }
}

__global__ void k_keys_from_m(long long n, const double* __restrict__ m, u64* __restrict__ keys,
                              u32* __restrict__ idx) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        keys[i] = rbl::flip_key(m[i]);
        if (idx) idx[i] = (u32)i;
    }
}

template <int LOSS>
__global__ __launch_bounds__(EW_THREADS) void k_prox(long long n, const double* __restrict__ sigma, double rho,
                                                      const double* __restrict__ m, double* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * EW_THREADS + threadIdx.x; i < n;
         i += (long long)gridDim.x * EW_THREADS)
        out[i] = rbl::prox<LOSS>(sigma[i], rho, m[i]);
}

// lambda += rho (z - v) (algorithms.py:132); partials: sum (z-v)^2 (:135), sum loss(v)
template <int LOSS>
__global__ __launch_bounds__(EW_THREADS) void k_dual(long long n, double rho, const double* __restrict__ z,
                                                      const double* __restrict__ v, double* __restrict__ lam,
                                                      double* __restrict__ partials) {
    __shared__ double smem[2 * EW_THREADS / 64];
    double acc[2] = {0.0, 0.0};
    for (long long i = (long long)blockIdx.x * EW_THREADS + threadIdx.x; i < n;
         i += (long long)gridDim.x * EW_THREADS) {
        double vi = v[i];
        double r = z[i] - vi;
        lam[i] = lam[i] + rho * r;
        acc[0] += r * r;
        acc[1] += rbl::sample_loss<LOSS>(vi);
    }
    rbl::block_sum<2, EW_THREADS>(acc, smem);
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 2 + 0] = acc[0];
        partials[blockIdx.x * 2 + 1] = acc[1];
    }
}

// out[k] = sum_b partials[b*K + k], one block, fixed order
__global__ __launch_bounds__(256) void k_sum_partials(const double* __restrict__ partials, int nblocks, int K,
                                                       double* __restrict__ out) {
    __shared__ double smem[4];
    for (int k = 0; k < K; ++k) {
        double acc[1] = {0.0};
        for (int b = threadIdx.x; b < nblocks; b += 256) acc[0] += partials[(long long)b * K + k];
        rbl::block_sum<1, 256>(acc, smem);
        if (threadIdx.x == 0) out[k] = acc[0];
        __syncthreads();
    }
}

__global__ void k_loss_keys(long long n, const double* __restrict__ v, u64* __restrict__ keys) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        keys[i] = rbl::flip_key(v[i]);
}

// sum_i sigma_i * loss(v_(i)) over ascending v (objective.py:73-81: losses are monotone
// in v, so sorting v sorts the losses)
template <int LOSS>
__global__ __launch_bounds__(EW_THREADS) void k_sorted_loss_dot(long long n, const u64* __restrict__ keys,
                                                                 const double* __restrict__ sigma,
                                                                 double* __restrict__ partials) {
    __shared__ double smem[EW_THREADS / 64];
    double acc[1] = {0.0};
    for (long long i = (long long)blockIdx.x * EW_THREADS + threadIdx.x; i < n;
         i += (long long)gridDim.x * EW_THREADS)
        acc[0] += sigma[i] * rbl::sample_loss<LOSS>(rbl::unflip_key(keys[i]));
    rbl::block_sum<1, EW_THREADS>(acc, smem);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

template <int LOSS>
__global__ __launch_bounds__(EW_THREADS) void k_loss_sum(long long n, const double* __restrict__ v, double scale,
                                                          double* __restrict__ partials) {
    __shared__ double smem[EW_THREADS / 64];
    double acc[1] = {0.0};
    for (long long i = (long long)blockIdx.x * EW_THREADS + threadIdx.x; i < n;
         i += (long long)gridDim.x * EW_THREADS)
        acc[0] += rbl::sample_loss<LOSS>(v[i]);
    acc[0] *= scale;
    rbl::block_sum<1, EW_THREADS>(acc, smem);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

// correct predictions of the linear classifier from v = D w = -y (x.w)
// (reference: src/util/calculate_acc.py:3-19).  tau = logit(threshold).
//   binary_cross_entropy: predict +1 iff sigmoid(x.w) >= threshold  <=>  x.w >= tau
//   hinge: the reference maps BOTH outcomes of (x.w >= 0) to +1 (calculate_acc.py:13-15),
//          i.e. its accuracy is the fraction of y == +1; mirrored as is.
template <int LOSS>
__global__ __launch_bounds__(EW_THREADS) void k_accuracy(long long n, const double* __restrict__ v,
                                                          const signed char* __restrict__ ysign, double tau,
                                                          double* __restrict__ partials) {
    __shared__ double smem[EW_THREADS / 64];
    double acc[1] = {0.0};
    for (long long i = (long long)blockIdx.x * EW_THREADS + threadIdx.x; i < n;
         i += (long long)gridDim.x * EW_THREADS) {
        const int y = ysign[i];
        bool ok;
        if (LOSS == 0) {
            const double xw = -(double)y * v[i];
            const int pred = (xw >= tau) ? 1 : -1;
            ok = pred == y;
        } else {
            ok = y == 1;
        }
        acc[0] += ok ? 1.0 : 0.0;
    }
    rbl::block_sum<1, EW_THREADS>(acc, smem);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

// Group-wise confusion counts and the Theil-index sums for the fairness statistics the EHRM
// driver prints (reference: src/util/fair_metric.py:3-41, called from run_EHRM.py:41).
// partials[b*14 + k]: k = 6*g + {rows, predicted +, TP, FN, TN, FP} for group g in {0,1};
// k = 12: sum b, k = 13: sum b*log(b) with b = prob - y01 + 1  (fair_metric.py:36-39).
__global__ __launch_bounds__(EW_THREADS) void k_fair_counts(long long n, const double* __restrict__ v,
                                                             const signed char* __restrict__ ysign,
                                                             const double* __restrict__ group, double threshold,
                                                             double* __restrict__ partials) {
    __shared__ double smem[14 * EW_THREADS / 64];
    double acc[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) acc[k] = 0.0;
    for (long long i = (long long)blockIdx.x * EW_THREADS + threadIdx.x; i < n;
         i += (long long)gridDim.x * EW_THREADS) {
        const int y = ysign[i];
        const double xw = -(double)y * v[i];
        const double prob = rbl::sigmoid1(xw);                 // fair_metric.py:5-7
        const int pred = prob >= threshold ? 1 : 0;            // :8
        const int y01 = y > 0 ? 1 : 0;                         // :9-10
        const int g = group[i] == 0.0 ? 0 : (group[i] == 1.0 ? 1 : -1);
        if (g >= 0) {
            const int o = 6 * g;
            acc[o + 0] += 1.0;
            acc[o + 1] += pred;
            acc[o + 2] += (pred == 1 && y01 == 1);
            acc[o + 3] += (pred == 0 && y01 == 1);
            acc[o + 4] += (pred == 0 && y01 == 0);
            acc[o + 5] += (pred == 1 && y01 == 0);
        }
        const double b = prob - (double)y01 + 1.0;
        acc[12] += b;
        acc[13] += b * log(b);
    }
    rbl::block_sum<14, EW_THREADS>(acc, smem);
    if (threadIdx.x == 0)
        for (int k = 0; k < 14; ++k) partials[blockIdx.x * 14 + k] = acc[k];
}

// ---- sigma generators: src/optim/objective.py:97-164 ---------------------------------
struct WeightParams {
    int wf;
    long long n;
    double a0, a1;       // args
    long long i0, i1;    // integer break points
    double wa, wb;       // boundary / plateau weights
};

__device__ inline double distort(double p, double gamma) {
    // objective.py:148-150
    double pg = pow(p, gamma);
    return pg / pow(pg + pow(1.0 - p, gamma), 1.0 / gamma);
}

__global__ void k_weights(WeightParams P, double* __restrict__ alphas, double* __restrict__ betas) {
    const double nn = (double)P.n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < P.n;
         i += (long long)gridDim.x * blockDim.x) {
        double x = (double)i, a = 0.0, b;
        switch (P.wf) {
            case RBL_W_ERM: a = 1.0 / nn; break;                                           // :97-98
            case RBL_W_EXTREMILE: a = (pow(x + 1.0, P.a0) - pow(x, P.a0)) / pow(nn, P.a0); break;  // :101-105
            case RBL_W_SUPERQUANTILE: a = (i < P.i0) ? 0.0 : (i == P.i0 ? P.wa : P.wb); break;     // :108-117
            case RBL_W_ESRM:                                                                // :120-123
                a = exp(-P.a0) * (exp(P.a0 * ((x + 1.0) / nn)) - exp(P.a0 * (x / nn))) / (1.0 - exp(-P.a0));
                break;
            case RBL_W_AORR: a = (i < P.i0 || i >= P.i1) ? 0.0 : (i == P.i0 ? P.wa : P.wb); break;  // :126-136
            case RBL_W_AORR_DC:                                                             // :139-145
                a = (i > P.i1 && i < P.i0) ? P.wb : 0.0;
                if (i == P.i0 + 1) a = P.wa;
                break;
            default: break;
        }
        b = a;
        if (P.wf == RBL_W_EHRM) {
            a = distort((x + 1.0) / nn, 0.69) - distort(x / nn, 0.69);                // :153-157
            b = distort((nn - x) / nn, 0.61) - distort((nn - x - 1.0) / nn, 0.61);    // :160-164
        }
        alphas[i] = a;
        if (betas) betas[i] = b;
    }
}

}  // namespace

int reduce_blocks() { return RED_BLOCKS; }

#define LAUNCH_LOSS(KERN, loss, grid, block, stream, ...)                                        \
    do {                                                                                         \
        if ((loss) == RBL_LOSS_BCE)                                                              \
            hipLaunchKernelGGL((KERN<0>), dim3(grid), dim3(block), 0, stream, __VA_ARGS__);      \
        else                                                                                     \
            hipLaunchKernelGGL((KERN<1>), dim3(grid), dim3(block), 0, stream, __VA_ARGS__);      \
    } while (0)

int launch_erm_zc(int loss, int64_t n, double sigma0, double rho, const double* v, const double* lam, double* m,
                  double* z, double* c, hipStream_t s) {
    if (n <= 0) return RBL_OK;
    LAUNCH_LOSS(k_erm_zc, loss, ew_grid(n), EW_THREADS, s, (long long)n, sigma0, rho, v, lam, m, z, c);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int launch_make_m_keys(int64_t n, double rho, const double* v, const double* lam, double* m, u64* keys, u32* idx,
                       u32 idx_off, hipStream_t s) {
    if (n <= 0) return RBL_OK;
    hipLaunchKernelGGL(k_make_m_keys, dim3(ew_grid(n)), dim3(256), 0, s, (long long)n, rho, v, lam, m, keys, idx, idx_off);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int s32_range_words() { return 2 * S32_RANGE_BLOCKS; }

int launch_make_m_range(int64_t n, double rho, const double* v, const double* lam, double* m, u64* mm, hipStream_t s) {
    if (n <= 0) return RBL_OK;
    hipLaunchKernelGGL(k_make_m_range, dim3(S32_RANGE_BLOCKS), dim3(EW_THREADS), 0, s, (long long)n, rho, v, lam, m, mm);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int launch_keys32(int64_t n, const double* m, const u64* mm, u32* keys, u32* idx, u32 idx_off, hipStream_t s) {
    if (n <= 0) return RBL_OK;
    long long g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(k_keys32, dim3((unsigned)g), dim3(256), 0, s, (long long)n, m, mm, S32_RANGE_BLOCKS, keys, idx, idx_off);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int launch_sort32_fix(int64_t n, const u32* keys, const u32* ids, const double* m, u32 off, double* ms, u32* ids_out, int* flag,
                      hipStream_t s) {
    RBL_HIP(hipMemsetAsync(flag, 0, sizeof(int), s));
    if (n <= 0) return RBL_OK;
    long long g = (n + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(k_sort32_fix, dim3((unsigned)g), dim3(256), 0, s, (long long)n, keys, ids, m, off, ms, ids_out, flag);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int launch_keys_from_m(int64_t n, const double* m, u64* keys, u32* idx, hipStream_t s) {
    if (n <= 0) return RBL_OK;
    hipLaunchKernelGGL(k_keys_from_m, dim3(ew_grid(n)), dim3(256), 0, s, (long long)n, m, keys, idx);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int launch_prox(int loss, int64_t n, const double* sigma, double rho, const double* m, double* out, hipStream_t s) {
    if (n <= 0) return RBL_OK;
    LAUNCH_LOSS(k_prox, loss, ew_grid(n), EW_THREADS, s, (long long)n, sigma, rho, m, out);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int launch_sum_partials(const double* partials, int nblocks, int K, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, s, partials, nblocks, K, out);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int launch_dual(int loss, int64_t n, double rho, const double* z, const double* v, double* lam, double* partials,
                double* red, hipStream_t s) {
    LAUNCH_LOSS(k_dual, loss, RED_BLOCKS, EW_THREADS, s, (long long)n, rho, z, v, lam, partials);
    RBL_HIP(hipGetLastError());
    return launch_sum_partials(partials, RED_BLOCKS, 2, red, s);
}

int launch_loss_keys(int64_t n, const double* v, u64* keys, hipStream_t s) {
    if (n <= 0) return RBL_OK;
    hipLaunchKernelGGL(k_loss_keys, dim3(ew_grid(n)), dim3(256), 0, s, (long long)n, v, keys);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int launch_sorted_loss_dot(int loss, int64_t n, const u64* sorted_keys, const double* sigma, double* partials,
                           double* out, hipStream_t s) {
    LAUNCH_LOSS(k_sorted_loss_dot, loss, RED_BLOCKS, EW_THREADS, s, (long long)n, sorted_keys, sigma, partials);
    RBL_HIP(hipGetLastError());
    return launch_sum_partials(partials, RED_BLOCKS, 1, out, s);
}

int launch_loss_sum(int loss, int64_t n, const double* v, double scale, double* partials, double* out,
                    hipStream_t s) {
    LAUNCH_LOSS(k_loss_sum, loss, RED_BLOCKS, EW_THREADS, s, (long long)n, v, scale, partials);
    RBL_HIP(hipGetLastError());
    return launch_sum_partials(partials, RED_BLOCKS, 1, out, s);
}

int launch_accuracy(int loss, int64_t n, const double* v, const signed char* ysign, double tau, double* partials,
                    double* out, hipStream_t s) {
    LAUNCH_LOSS(k_accuracy, loss, RED_BLOCKS, EW_THREADS, s, (long long)n, v, ysign, tau, partials);
    RBL_HIP(hipGetLastError());
    return launch_sum_partials(partials, RED_BLOCKS, 1, out, s);
}

int fair_partial_blocks() { return 256; }

int launch_fair_counts(int64_t n, const double* v, const signed char* ysign, const double* group, double threshold,
                       double* partials, double* out14, hipStream_t s) {
    hipLaunchKernelGGL(k_fair_counts, dim3(fair_partial_blocks()), dim3(EW_THREADS), 0, s, (long long)n, v, ysign,
                       group, threshold, partials);
    RBL_HIP(hipGetLastError());
    return launch_sum_partials(partials, fair_partial_blocks(), 14, out14, s);
}

int launch_weights(int wf, int64_t n, const double* args, double* alphas, double* betas, hipStream_t s) {
    WeightParams P;
    P.wf = wf;
    P.n = n;
    P.a0 = args ? args[0] : 0.0;
    P.a1 = args ? args[1] : 0.0;
    P.i0 = P.i1 = 0;
    P.wa = P.wb = 0.0;
    const double nn = (double)n;
    if (wf == RBL_W_SUPERQUANTILE) {  // objective.py:108-117
        double q = P.a0;
        long long idx = (long long)floor(nn * q);
        double frac = 1.0 - (nn - (double)idx - 1.0) / (nn * (1.0 - q));
        P.i0 = idx;
        if (frac > 1e-12) {
            P.wa = frac;
            P.wb = 1.0 / (nn * (1.0 - q));
        } else {
            P.wa = P.wb = 1.0 / (nn - (double)idx);
        }
    } else if (wf == RBL_W_AORR) {  // objective.py:126-136
        double ql = P.a0, qu = P.a1;
        long long lo = (long long)floor(nn * ql), up = (long long)floor(nn * qu);
        double frac = 1.0 - ((double)up - (double)lo - 1.0) / (nn * (qu - ql));
        P.i0 = lo;
        P.i1 = up;
        if (frac > 1e-12) {
            P.wa = frac;
            P.wb = 1.0 / (nn * (qu - ql));
        } else {
            P.wa = P.wb = 1.0 / ((double)up - (double)lo);
        }
    } else if (wf == RBL_W_AORR_DC) {  // objective.py:139-145 (k = args[0], m = args[1])
        long long k = (long long)P.a0, mm = (long long)P.a1;
        if (k <= mm) {
            rbl_set_error("need args[0] > args[1]!");
            return RBL_ERR_INVALID;
        }
        if (k + 1 >= n) {
            rbl_set_error("aorr_dc needs args[0] + 1 < n");
            return RBL_ERR_INVALID;
        }
        P.i0 = k;
        P.i1 = mm;
        P.wb = 1.0 / (double)(k - mm);
        P.wa = 1.0 - (double)(k - mm - 1) / (double)(k - mm);
    }
    hipLaunchKernelGGL(k_weights, dim3(ew_grid(n)), dim3(256), 0, s, P, alphas, betas);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
