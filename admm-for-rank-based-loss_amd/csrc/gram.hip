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
    // XCD-aware placement: workgroups are dealt round-robin over the 8 XCDs by their linear id, and
    // every XCD has its own L2.  The tile pairs of one row split read the SAME rows of D, so they
    // are given linear ids that are equal mod 8 (gridDim.y is a multiple of 8): one XCD's L2 then
    // serves a row to all the pairs that need it instead of every XCD fetching it from HBM.
    const int npairs = gridDim.x;
    const int lin = blockIdx.y * npairs + blockIdx.x;
    const int in_xcd = lin >> 3;
    const int split = (lin & 7) + 8 * (in_xcd / npairs);
    const int pair = in_xcd % npairs;
    int p = pair, ti = 0;
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
    const long long r_begin = (long long)split * rows_per_split;
    long long r_end = r_begin + rows_per_split;
    if (r_end > n) r_end = n;

    v4d acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};

    // Operands come straight from global memory (a lane needs D[r + lane/16][c0 + 16 s + lane%16]:
    // 64-byte row segments, L1/L2 absorb the reuse between the 4 waves), as buffer loads:
    // scalar descriptor = the step's first row, per-lane offset = (lane/16) rows + column, and
    // num_records = the bytes left in this split, so rows past its end - and columns past ld,
    // whose offset is pushed out of range - read as 0 with no per-load address arithmetic.
    // DEPTH steps are kept in flight (raw storage type in registers, widened at use): one step
    // is 16 MFMAs = ~1000 cycles of matrix pipe per wave, loaded HBM latency is several times that.
    constexpr int DEPTH = 4;
    constexpr unsigned OOB = 0x7fffffffu;
    unsigned offa[4], offb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const long long ca = ci0 + 16 * s + lc, cb = cj0 + 16 * s + lc;
        offa[s] = ca < ld ? (unsigned)(((long long)lk * ld + ca) * (long long)sizeof(T)) : OOB;
        offb[s] = cb < ld ? (unsigned)(((long long)lk * ld + cb) * (long long)sizeof(T)) : OOB;
    }
    const long long row_bytes = ld * (long long)sizeof(T);

    T ra[DEPTH][4], rb[DEPTH][4];
    auto load_step = [&](long long r, T (&a)[4], T (&b)[4]) {
        long long left = (r_end - r) * row_bytes;   // bytes of this split from row r on (<= 0: nothing)
        if (left < 0) left = 0;
        if (left > 4 * row_bytes) left = 4 * row_bytes;
        const long long rr = r < r_end ? r : r_begin;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(D + rr * ld), 0, (int)left, 0x00020000);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr (sizeof(T) == 4) {
                a[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)offa[s], 0, 0));
                b[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)offb[s], 0, 0));
            } else {
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 va = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)offa[s], 0, 0);
                const u32x2 vb = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)offb[s], 0, 0);
                a[s] = __hiloint2double((int)va[1], (int)va[0]);
                b[s] = __hiloint2double((int)vb[1], (int)vb[0]);
            }
        }
    };
    auto mfma_step = [&](T (&a)[4], T (&b)[4]) {
        double da[4], db[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            // opaque use of the raw registers HERE: otherwise the compiler widens all DEPTH steps at the
            // top of the loop body and its s_waitcnt vmcnt(0) there drains the whole prefetch
            asm volatile("" : "+v"(a[s]), "+v"(b[s]));
            da[s] = (double)a[s];
            db[s] = (double)b[s];
        }
#pragma unroll
        for (int si = 0; si < 4; ++si)
#pragma unroll
            for (int sj = 0; sj < 4; ++sj)
                acc[si][sj] = __builtin_amdgcn_mfma_f64_16x16x4f64(da[si], db[sj], acc[si][sj], 0, 0, 0);
    };

    if (r_begin < r_end) {
#pragma unroll
        for (int k = 0; k < DEPTH - 1; ++k) load_step(r_begin + 4 * k, ra[k], rb[k]);
        for (long long r = r_begin; r < r_end; r += 4 * DEPTH) {
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) {
                // the step DEPTH-1 ahead goes into the slot consumed in the previous sub-step
                load_step(r + 4 * (k + DEPTH - 1), ra[(k + DEPTH - 1) % DEPTH], rb[(k + DEPTH - 1) % DEPTH]);
                mfma_step(ra[k], rb[k]);   // rows past r_end were loaded as zeros
            }
        }
    }

    // accumulator layout of v_mfma_f64_16x16x4_f64: register i of a lane holds row
    // 4*i + lane/16, column lane%16 (checked against numpy in tests/test_gpu_kernels.py)
    double* tile = slab + ((long long)split * npairs + pair) * (GT * GT);
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
    // XCD-aware placement (k_gram) deals the splits over the 8 XCDs: a multiple of 8 of them, all
    // with rows (a split count of 1 would put the whole matrix on one XCD)
    want = (want + 7) / 8 * 8;
    long long rps = (n + want - 1) / want;
    rps = (rps + 7) / 8 * 8;
    if (rps < 8) rps = 8;
    g.rows_per_split = rps;
    g.ksplit = (int)want;   // splits past the last row (tiny n) write zero tiles
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
