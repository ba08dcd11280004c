// gram.hip - G = D^T D (src/optim/algorithms.py:24), the one dense contraction of the
// path and the only MFMA-shaped work: 2*n*d^2 flops, done once per problem.
// fp64 matrix cores (v_mfma_f64_16x16x4_f64): D is widened to fp64 in registers so G is
// exact to fp64 accumulation regardless of the storage type.
// Tiling: 128 x 128 output tile per 4-wave workgroup (each wave a 64 x 64 quadrant =
// 4 x 4 MFMA tiles, 128 accumulator VGPRs), only tile pairs ti <= tj (symmetry), the row
// (K) dimension split over blockIdx.y; partial tiles go to a slab and are summed in a
// fixed order (deterministic), then mirrored.
#include "rbl_internal.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int GT = 128;  // block tile

template <typename T>
__global__ __launch_bounds__(256) void k_gram(const T* __restrict__ D, long long n, long long ld, int ntiles,
                                               long long rows_per_split, double* __restrict__ slab) {
    int p = blockIdx.x, ti = 0;
    while (p >= ntiles - ti) {
        p -= ntiles - ti;
        ++ti;
    }
    const int tj = ti + p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const long long ci0 = (long long)ti * GT + wi * 64;
    const long long cj0 = (long long)tj * GT + wj * 64;
    const int lk = lane >> 4;  // k (row inside the 4-row step) held by this lane
    const int lc = lane & 15;  // column inside a 16-wide MFMA operand
    const long long r_begin = (long long)blockIdx.y * rows_per_split;
    long long r_end = r_begin + rows_per_split;
    if (r_end > n) r_end = n;

    v4d acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};

    bool cia[4], cjb[4];
    long long coli[4], colj[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        coli[s] = ci0 + 16 * s + lc;
        colj[s] = cj0 + 16 * s + lc;
        cia[s] = coli[s] < ld;
        cjb[s] = colj[s] < ld;
        if (!cia[s]) coli[s] = 0;
        if (!cjb[s]) colj[s] = 0;
    }

    auto load_step = [&](long long r, double (&a)[4], double (&b)[4]) {
        long long row = r + lk;
        const bool rv = row < r_end;
        if (!rv) row = r_begin;
        const T* rp = D + row * ld;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            a[s] = (rv && cia[s]) ? (double)rp[coli[s]] : 0.0;
            b[s] = (rv && cjb[s]) ? (double)rp[colj[s]] : 0.0;
        }
    };

    if (r_begin < r_end) {
        double a0[4], b0[4], a1[4], b1[4];
        load_step(r_begin, a0, b0);
        for (long long r = r_begin; r < r_end; r += 8) {
            load_step(r + 4, a1, b1);  // rows beyond r_end load zeros
#pragma unroll
            for (int si = 0; si < 4; ++si)
#pragma unroll
                for (int sj = 0; sj < 4; ++sj)
                    acc[si][sj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[si], b0[sj], acc[si][sj], 0, 0, 0);
            load_step(r + 8, a0, b0);
#pragma unroll
            for (int si = 0; si < 4; ++si)
#pragma unroll
                for (int sj = 0; sj < 4; ++sj)
                    acc[si][sj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[si], b1[sj], acc[si][sj], 0, 0, 0);
        }
    }

    // accumulator layout of v_mfma_f64_16x16x4_f64: register i of a lane holds row
    // 4*i + lane/16, column lane%16 (checked against numpy in tests/test_gpu_kernels.py)
    double* tile = slab + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * (GT * GT);
#pragma unroll
    for (int si = 0; si < 4; ++si)
#pragma unroll
        for (int sj = 0; sj < 4; ++sj)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int lr = wi * 64 + 16 * si + 4 * i + lk;
                const int lcol = wj * 64 + 16 * sj + lc;
                tile[lr * GT + lcol] = acc[si][sj][i];
            }
}

__global__ void k_gram_reduce(const double* __restrict__ slab, int npairs, int ksplit, int ntiles, long long ld,
                              double* __restrict__ G) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)npairs * GT * GT) return;
    const int pair = (int)(idx / (GT * GT));
    const int e = (int)(idx % (GT * GT));
    int p = pair, ti = 0;
    while (p >= ntiles - ti) {
        p -= ntiles - ti;
        ++ti;
    }
    const int tj = ti + p;
    const long long gi = (long long)ti * GT + e / GT, gj = (long long)tj * GT + e % GT;
    if (gi >= ld || gj >= ld) return;
    if (ti == tj && gi > gj) return;  // diagonal tiles: upper triangle only, mirrored below
    double acc = 0.0;
    for (int s = 0; s < ksplit; ++s) acc += slab[((long long)s * npairs + pair) * (GT * GT) + e];
    G[gi * ld + gj] = acc;
    G[gj * ld + gi] = acc;
}

struct GramPlan {
    int ntiles, npairs, ksplit;
    long long rows_per_split;
};

GramPlan gram_plan(long long ld, long long n, int num_cu) {
    GramPlan g;
    g.ntiles = (int)((ld + GT - 1) / GT);
    g.npairs = g.ntiles * (g.ntiles + 1) / 2;
    long long want = ((long long)num_cu * 4 + g.npairs - 1) / g.npairs;
    long long maxsplit = (n + 255) / 256;  // at least 256 rows per split
    if (want > maxsplit) want = maxsplit;
    if (want < 1) want = 1;
    g.ksplit = (int)want;
    long long rps = (n + g.ksplit - 1) / g.ksplit;
    rps = (rps + 7) / 8 * 8;
    if (rps < 8) rps = 8;
    g.rows_per_split = rps;
    g.ksplit = (int)((n + rps - 1) / rps);
    if (g.ksplit < 1) g.ksplit = 1;
    return g;
}

}  // namespace

size_t gram_slab_bytes(int64_t ld, int num_cu, int64_t n) {
    GramPlan g = gram_plan(ld, n, num_cu);
    return (size_t)g.ksplit * g.npairs * GT * GT * sizeof(double);
}

int launch_gram(int storage, const void* D, int64_t n, int64_t ld, int64_t d, double* slab, double* G, int num_cu,
                hipStream_t s) {
    (void)d;
    RBL_HIP(hipMemsetAsync(G, 0, sizeof(double) * ld * ld, s));
    if (n <= 0) return RBL_OK;
    GramPlan g = gram_plan(ld, n, num_cu);
    dim3 grid(g.npairs, g.ksplit);
    if (storage == RBL_STORE_F32)
        hipLaunchKernelGGL(k_gram<float>, grid, dim3(256), 0, s, (const float*)D, (long long)n, (long long)ld,
                           g.ntiles, g.rows_per_split, slab);
    else
        hipLaunchKernelGGL(k_gram<double>, grid, dim3(256), 0, s, (const double*)D, (long long)n, (long long)ld,
                           g.ntiles, g.rows_per_split, slab);
    const long long total = (long long)g.npairs * GT * GT;
    hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, slab, g.npairs, g.ksplit,
                       g.ntiles, (long long)ld, G);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
