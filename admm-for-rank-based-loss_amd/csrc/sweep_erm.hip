// sweep_erm.hip - ONE pass over D per ADMM iteration for erm weights.
//
// With constant sigma the z-step is element-wise (z_i = prox(m_i)), so the
// tail of iteration k and the head of iteration k+1 touch every row of D exactly once each:
//     v_i      = D_i . w_{k+1}                               (algorithms.py:132,135)
//     lambda_i += rho_k (z_i - v_i);  primal^2 += (z_i - v_i)^2
//     m_i      = v_i - lambda_i / rho_{k+1}                  (algorithms.py:89, next iteration)
//     z'_i     = prox(sigma0, rho_{k+1}, m_i)                (individual_solver.py:112-123)
//     q       += (z'_i + lambda_i / rho_{k+1}) * D_i         (the w-step's D^T c)
// This kernel does all of it while the row sits in registers: a wave owns R rows at a time
// (P 16-byte packets per lane and row, as k_gemv), reduces the R dot products with xor
// shuffles, lanes 0..R-1 do the row-wise update + warm-started prox, the R coefficients are
// broadcast back and the rows are accumulated into per-lane column sums.  The next batch's
// loads are issued before the current batch is processed (two register buffers).
// rho_{k+1} depends on the GLOBAL primal residual of iteration k; it is predicted in d-space
// before the pass (k_predict_rho: ||z - D w||^2 = ||z||^2 - 2 (D^T z)'w + w'Gw) and verified
// after it with the exact residual this kernel accumulates; on a misprediction the host
// simply recomputes the z-step / q with the unfused kernels (api.hip).
// Algorithmic bytes = n*ld*sizeof(T): half of the unfused iteration's.
#include "rbl_internal.h"
#include "device_math.h"

namespace {

template <typename T> struct Pk;
template <> struct Pk<float> {
    static constexpr int E = 4;
    typedef float4 type;
    __device__ static inline double at(const float4& p, int k) {
        return k == 0 ? (double)p.x : (k == 1 ? (double)p.y : (k == 2 ? (double)p.z : (double)p.w));
    }
};
template <> struct Pk<double> {
    static constexpr int E = 2;
    typedef double2 type;
    __device__ static inline double at(const double2& p, int k) { return k == 0 ? p.x : p.y; }
};

constexpr int SE_THREADS = 256;

template <int R, int P>
__device__ inline void opaque(float4 (&buf)[R][P]) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int p = 0; p < P; ++p)
            asm volatile("" : "+v"(buf[r][p].x), "+v"(buf[r][p].y), "+v"(buf[r][p].z), "+v"(buf[r][p].w));
}
template <int R, int P>
__device__ inline void opaque(double2 (&)[R][P]) {}

// pred[0] = rho_{k+1} (predicted), read on the device so no host round trip is needed
template <typename T, int LOSS, int P, int R>
__global__ __launch_bounds__(SE_THREADS, 2) void k_sweep_erm(
    const T* __restrict__ D, long long n, long long ld, const double* __restrict__ w, const double* __restrict__ z_old,
    double* __restrict__ lam, double* __restrict__ v, double* __restrict__ z_new, double sigma0, double rho,
    const double* __restrict__ pred, double* __restrict__ slab, double* __restrict__ partials, int want_obj) {
    typedef typename Pk<T>::type pkt_t;
    constexpr int E = Pk<T>::E;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long PK = ld / E;
    const double rho_next = pred[0];

    double wr[P][E], acc[P][E];
    long long pk[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        pk[p] = lane + 64LL * p;
        const bool ok = pk[p] < PK;
#pragma unroll
        for (int k = 0; k < E; ++k) {
            wr[p][k] = ok ? w[pk[p] * E + k] : 0.0;
            acc[p][k] = 0.0;
        }
        if (!ok) pk[p] = PK - 1;  // tail lanes re-read the last packet; w == 0 and the sums are dropped
    }
    double s_prim = 0.0, s_loss = 0.0, s_zz = 0.0;

    const long long nbatch = (n + R - 1) / R;
    const long long gw = (long long)blockIdx.x * (SE_THREADS / 64) + wave;
    const long long GW = (long long)gridDim.x * (SE_THREADS / 64);

    // Every load of batch b (rows, z_old, lambda) is issued before any load of batch b+1:
    // s_waitcnt vmcnt counts in issue order, so waiting for batch b then leaves the whole of
    // batch b+1 in flight (a row-wise load issued later would drain the prefetch).
    auto load_batch = [&](long long b, pkt_t (&buf)[R][P], double& zo, double& lm) {
        const long long myrow = b * R + lane;
        const bool mine = lane < R && myrow < n;
        zo = mine ? z_old[myrow] : 0.0;
        lm = mine ? lam[myrow] : 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            long long row = b * R + r;
            if (row >= n) row = n - 1;
            const pkt_t* rp = reinterpret_cast<const pkt_t*>(D + row * ld);
#pragma unroll
            for (int p = 0; p < P; ++p) buf[r][p] = rp[pk[p]];
        }
    };

    auto process = [&](long long b, pkt_t (&buf)[R][P], double zo, double lm) {
        double dot[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double a = 0.0;
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int k = 0; k < E; ++k) a = __builtin_fma(Pk<T>::at(buf[r][p], k), wr[p][k], a);
            dot[r] = rbl::wave_sum_all(a);   // DPP butterfly + scalar readlanes: no LDS round trips
        }
        // lane r owns row b*R + r
        double myv = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) myv = (lane == r) ? dot[r] : myv;
        double c = 0.0;
        const long long row = b * R + lane;
        if (lane < R && row < n) {
            const double res = zo - myv;
            const double l = lm + rho * res;                       // algorithms.py:132
            s_prim += res * res;                                   // algorithms.py:135
            if (want_obj) s_loss += rbl::sample_loss<LOSS>(myv);   // objective.py:11-24
            const double lr = l / rho_next;
            const double m = myv - lr;                             // algorithms.py:89 (next iteration)
            const double zn = (LOSS == 0) ? rbl::prox_bce_warm(sigma0, rho_next, m, zo) : rbl::prox_hinge(sigma0, rho_next, m);
            s_zz += zn * zn;
            lam[row] = l;
            v[row] = myv;
            z_new[row] = zn;
            c = zn + lr;
        }
        // the fp32 -> fp64 widening is redone for the accumulation: keeping the widened copies of
        // the dot phase alive across the prox would cost 2 VGPRs per element and spill
        opaque(buf);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double cr = rbl::readlane_d(c, r);
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int k = 0; k < E; ++k) acc[p][k] = __builtin_fma(Pk<T>::at(buf[r][p], k), cr, acc[p][k]);
        }
    };

    pkt_t bufA[R][P], bufB[R][P];
    double zoA = 0.0, lmA = 0.0, zoB = 0.0, lmB = 0.0;
    long long b = gw;
    if (b < nbatch) load_batch(b, bufA, zoA, lmA);
    while (b < nbatch) {
        const long long b1 = b + GW;
        if (b1 < nbatch) load_batch(b1, bufB, zoB, lmB);
        process(b, bufA, zoA, lmA);
        if (b1 >= nbatch) break;
        const long long b2 = b1 + GW;
        if (b2 < nbatch) load_batch(b2, bufA, zoA, lmA);
        process(b1, bufB, zoB, lmB);
        b = b2;
    }

    // fold the 4 waves' column sums in LDS, one slab row per block
    __shared__ double red[SE_THREADS / 64][64 * P * E];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int k = 0; k < E; ++k) red[wave][(p * 64 + lane) * E + k] = acc[p][k];
    __syncthreads();
    for (int i = tid; i < 64 * P * E; i += SE_THREADS) {
        const long long col = i;  // packet (p*64+lane), element k  ->  column (p*64+lane)*E + k
        if (col < ld) {
            double s = 0.0;
#pragma unroll
            for (int wv = 0; wv < SE_THREADS / 64; ++wv) s += red[wv][i];
            slab[(long long)blockIdx.x * ld + col] = s;
        }
    }
    __syncthreads();
    double sums[3] = {s_prim, s_loss, s_zz};
    __shared__ double smem[3 * SE_THREADS / 64];
    rbl::block_sum<3, SE_THREADS>(sums, smem);
    if (tid == 0) {
        partials[blockIdx.x * 3 + 0] = sums[0];
        partials[blockIdx.x * 3 + 1] = sums[1];
        partials[blockIdx.x * 3 + 2] = sums[2];
    }
}

// q[j] = sum_b slab[b][j]  (same as sweep.hip's k_colreduce; kept local to this file)
__global__ __launch_bounds__(1024) void k_colreduce2(const double* __restrict__ slab, int nb, long long ld,
                                                       double* __restrict__ q) {
    __shared__ double red[16][64];
    const int cx = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long long col = (long long)blockIdx.x * 64 + cx;
    double acc = 0.0;
    if (col < ld)
        for (int b = g; b < nb; b += 16) acc += slab[(long long)b * ld + col];
    red[g][cx] = acc;
    __syncthreads();
    if (g == 0 && col < ld) {
        double sacc = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) sacc += red[k][cx];
        q[col] = sacc;
    }
}

// red[0] = sum primal^2, red[1] = sum loss, zz_out[0] = sum z'^2
__global__ __launch_bounds__(256) void k_sum3(const double* __restrict__ partials, int nb, double* __restrict__ red,
                                               double* __restrict__ zz_out) {
    __shared__ double smem[3 * 4];
    double a[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nb; b += 256) {
        a[0] += partials[b * 3 + 0];
        a[1] += partials[b * 3 + 1];
        a[2] += partials[b * 3 + 2];
    }
    rbl::block_sum<3, 256>(a, smem);
    if (threadIdx.x == 0) {
        red[0] = a[0];
        red[1] = a[1];
        zz_out[0] = a[2];
    }
}

// Predict rho_{k+1} before the pass.  Gw = G w_{k+1} (k_symv), q = D^T(z + lambda/rho) summed over
// ranks, p = D^T lambda (kept by the recurrence below), zz = ||z||^2:
//   D^T z = q - p/rho ;  ||z - D w||^2 = zz - 2 (D^T z)'w + w'Gw ;  p <- p + rho (D^T z - G w)
__global__ __launch_bounds__(1024) void k_predict_rho(long long ld, const double* __restrict__ q,
                                                       double* __restrict__ p, const double* __restrict__ w,
                                                       const double* __restrict__ Gw, const double* __restrict__ zz,
                                                       double rho, double cap, double* __restrict__ pred) {
    __shared__ double smem[2 * 16];
    double a[2] = {0.0, 0.0};
    for (long long j = threadIdx.x; j < ld; j += 1024) {
        const double dz = q[j] - p[j] / rho;
        a[0] += dz * w[j];
        a[1] += w[j] * Gw[j];
        p[j] = p[j] + rho * (dz - Gw[j]);
    }
    rbl::block_sum<2, 1024>(a, smem);
    if (threadIdx.x == 0) {
        double pr2 = zz[0] - 2.0 * a[0] + a[1];
        if (pr2 < 0.0) pr2 = 0.0;
        const double primal = sqrt(pr2);
        double rn = rho * (primal > 1e-2 ? 1.02 : 1.07);  // algorithms.py:154-157
        if (rn > cap) rn = cap;
        pred[0] = rn;
        pred[1] = primal;
    }
}

__global__ __launch_bounds__(256) void k_sumsq(long long n, const double* __restrict__ x, double* __restrict__ partials) {
    __shared__ double smem[4];
    double a[1] = {0.0};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) a[0] += x[i] * x[i];
    rbl::block_sum<1, 256>(a, smem);
    if (threadIdx.x == 0) partials[blockIdx.x] = a[0];
}

int g_want_obj = 1;  // set per launch by launch_sweep_erm (host side, single caller thread per handle)

template <typename T, int LOSS, int P, int R>
int launch_one(const T* D, long long n, long long ld, const double* w, const double* z_old, double* lam, double* v,
               double* z_new, double sigma0, double rho, const double* pred, double* slab, double* partials, int grid,
               hipStream_t s) {
    hipLaunchKernelGGL((k_sweep_erm<T, LOSS, P, R>), dim3(grid), dim3(SE_THREADS), 0, s, D, n, ld, w, z_old, lam, v,
                       z_new, sigma0, rho, pred, slab, partials, g_want_obj);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

template <typename T, int LOSS>
int launch_T(const T* D, long long n, long long ld, const double* w, const double* z_old, double* lam, double* v,
             double* z_new, double sigma0, double rho, const double* pred, double* slab, double* partials, int grid,
             hipStream_t s) {
    const long long PK = ld / Pk<T>::E;
    const long long passes = (PK + 63) / 64;
    if (passes == 1) return launch_one<T, LOSS, 1, 8>(D, n, ld, w, z_old, lam, v, z_new, sigma0, rho, pred, slab, partials, grid, s);
    if (passes == 2) return launch_one<T, LOSS, 2, 4>(D, n, ld, w, z_old, lam, v, z_new, sigma0, rho, pred, slab, partials, grid, s);
    if (passes <= 4) return launch_one<T, LOSS, 4, 2>(D, n, ld, w, z_old, lam, v, z_new, sigma0, rho, pred, slab, partials, grid, s);
    return launch_one<T, LOSS, 8, 1>(D, n, ld, w, z_old, lam, v, z_new, sigma0, rho, pred, slab, partials, grid, s);
}

}  // namespace

bool sweep_erm_supported(int storage, int64_t ld) {
    const int64_t PK = ld / (storage == RBL_STORE_F32 ? 4 : 2);
    // d <= 1024 in both storage types (the wider fp32 variant would spill registers)
    return PK > 32 && PK <= (storage == RBL_STORE_F32 ? 256 : 512);
}

int sweep_erm_blocks(int num_cu) { return num_cu * 2; }  // 2 blocks of 4 waves per CU (2 waves per SIMD)

int launch_sweep_erm(int storage, int loss, const void* D, int64_t n, int64_t ld, const double* w, const double* z_old,
                     double* lam, double* v, double* z_new, double sigma0, double rho, const double* pred_dev,
                     double* slab, double* partials, double* q, double* red, double* zz_out, int num_cu, hipStream_t s,
                     hipEvent_t main_done, int want_obj) {
    g_want_obj = want_obj;
    const int grid = sweep_erm_blocks(num_cu);
    int rc;
    if (storage == RBL_STORE_F32) {
        rc = (loss == RBL_LOSS_BCE)
                 ? launch_T<float, 0>((const float*)D, n, ld, w, z_old, lam, v, z_new, sigma0, rho, pred_dev, slab, partials, grid, s)
                 : launch_T<float, 1>((const float*)D, n, ld, w, z_old, lam, v, z_new, sigma0, rho, pred_dev, slab, partials, grid, s);
    } else {
        rc = (loss == RBL_LOSS_BCE)
                 ? launch_T<double, 0>((const double*)D, n, ld, w, z_old, lam, v, z_new, sigma0, rho, pred_dev, slab, partials, grid, s)
                 : launch_T<double, 1>((const double*)D, n, ld, w, z_old, lam, v, z_new, sigma0, rho, pred_dev, slab, partials, grid, s);
    }
    RBL_TRY(rc);
    if (main_done) RBL_HIP(hipEventRecord(main_done, s));
    hipLaunchKernelGGL(k_colreduce2, dim3((unsigned)((ld + 63) / 64)), dim3(1024), 0, s, slab, grid, (long long)ld, q);
    hipLaunchKernelGGL(k_sum3, dim3(1), dim3(256), 0, s, partials, grid, red, zz_out);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int launch_predict_rho(int64_t ld, const double* q, double* p, const double* w, const double* Gw, const double* zz,
                       double rho, double cap, double* pred, hipStream_t s) {
    hipLaunchKernelGGL(k_predict_rho, dim3(1), dim3(1024), 0, s, (long long)ld, q, p, w, Gw, zz, rho, cap, pred);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int launch_sumsq(int64_t n, const double* x, double* partials, double* out, hipStream_t s) {
    const int nb = reduce_blocks();
    hipLaunchKernelGGL(k_sumsq, dim3(nb), dim3(256), 0, s, (long long)n, x, partials);
    RBL_HIP(hipGetLastError());
    return launch_sum_partials(partials, nb, 1, out, s);
}
