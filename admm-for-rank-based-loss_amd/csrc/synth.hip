// synth.hip - synthetic two-class data generated in place on the device, never on the
// host (BASELINE configs C2-C5 do not fit host-side generation + upload).  Reproduces the
// STATISTICS of the reference's synthetic branch, src/util/load_data.py:101-116
// (sklearn make_classification defaults: 2 informative + 2 redundant columns, the other
// d-4 columns N(0,1) noise, 2 clusters per class on hypercube vertices - cluster k belongs to
// class k % 2, its points are standard normal draws times a random matrix A_k with entries
// uniform in (-1, 1), shifted to its vertex at +-class_sep -, redundant = informative @ B,
// flip_y label noise, shuffled columns) followed by preprocessing.scale (done by
// launch_colstats + launch_standardize_negy in sweep.hip).  Counter-based Philox4x32-10,
// keyed by the seed and indexed by (global row, column packet): any sharding of the rows
// over GPUs yields the same matrix.  This kernel writes the RAW matrix (storage type T)
// and the labels; standardisation and the -y scaling follow.
#include "rbl_internal.h"

namespace {

struct U4 { unsigned x, y, z, w; };

__device__ inline U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) {
    const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
        unsigned hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
        U4 n;
        n.x = hi1 ^ c.y ^ k0;
        n.y = lo1;
        n.z = hi0 ^ c.w ^ k1;
        n.w = lo0;
        c = n;
        k0 += W0;
        k1 += W1;
    }
    return c;
}

__device__ inline float u01(unsigned x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)

__device__ inline void box_muller(unsigned a, unsigned b, float& n0, float& n1) {
    float r = sqrtf(-2.0f * __logf(u01(a)));
    float s, c;
    __sincosf(6.28318530717958647692f * u01(b), &s, &c);
    n0 = r * c;
    n1 = r * s;
}

struct SynthParams {
    long long n, ld, d, row_offset;
    unsigned k0, k1;
    float class_sep, flip_y;
    int special[4];   // output columns that hold the 2 informative + 2 redundant features
    float mix[4];     // redundant = informative @ mix (2 x 2)
    float A[4][4];    // cluster k: informative = (g0, g1) @ A_k (2 x 2, row-major) + centroid_k
    float cen[4][2];  // centroid of cluster k: its hypercube vertex at +-class_sep
};

template <typename T>
__global__ __launch_bounds__(256) void k_synth(T* __restrict__ D, SynthParams P, signed char* __restrict__ ysign) {
    const long long packets = P.ld / 4;
    const long long total = P.n * packets;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / packets, pk = i - r * packets;
        const unsigned long long gr = (unsigned long long)(r + P.row_offset);
        U4 ctr = {(unsigned)gr, (unsigned)(gr >> 32), (unsigned)pk, 1u};
        U4 rnd = philox4x32_10(ctr, P.k0, P.k1);
        float x[4];
        box_muller(rnd.x, rnd.y, x[0], x[1]);
        box_muller(rnd.z, rnd.w, x[2], x[3]);
        // per-row draw: label, cluster, flip, informative noise
        U4 rc = {(unsigned)gr, (unsigned)(gr >> 32), 0xFFFFFFFFu, 0u};
        U4 rr = philox4x32_10(rc, P.k0, P.k1);
        const int y01 = rr.x & 1u, cl = (rr.x >> 1) & 1u;
        int ylab = y01;
        if (u01(rr.y) < P.flip_y) ylab = (rr.x >> 2) & 1u;  // sklearn: flipped rows get a random class
        bool touches = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) touches |= (P.special[k] >> 2) == pk;
        if (touches) {
            float g0, g1;
            box_muller(rr.z, rr.w, g0, g1);
            const int c = (cl << 1) | y01;                             // cluster: class = c % 2 as in make_classification
            const float f0 = g0 * P.A[c][0] + g1 * P.A[c][2] + P.cen[c][0];
            const float f1 = g0 * P.A[c][1] + g1 * P.A[c][3] + P.cen[c][1];
            const float feat[4] = {f0, f1, f0 * P.mix[0] + f1 * P.mix[2], f0 * P.mix[1] + f1 * P.mix[3]};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((P.special[k] >> 2) == pk) x[P.special[k] & 3] = feat[k];
        }
        T* dst = D + r * P.ld + pk * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = (pk * 4 + k < P.d) ? (T)x[k] : (T)0;
        if (pk == 0) ysign[r] = ylab ? 1 : -1;
    }
}

}  // namespace

int launch_synth(int storage, void* D, int64_t n, int64_t ld, int64_t d, int64_t row_offset, u64 seed,
                 double class_sep, double flip_y, const int* special, const double* mix, const double* A16,
                 const int* vertex, signed char* ysign, hipStream_t s) {
    SynthParams P;
    P.n = n;
    P.ld = ld;
    P.d = d;
    P.row_offset = row_offset;
    P.k0 = (unsigned)seed;
    P.k1 = (unsigned)(seed >> 32);
    P.class_sep = (float)class_sep;
    P.flip_y = (float)flip_y;
    for (int k = 0; k < 4; ++k) {
        P.special[k] = special[k];
        P.mix[k] = (float)mix[k];
        for (int j = 0; j < 4; ++j) P.A[k][j] = (float)A16[4 * k + j];
        P.cen[k][0] = (float)class_sep * ((vertex[k] & 1) ? 1.0f : -1.0f);
        P.cen[k][1] = (float)class_sep * ((vertex[k] & 2) ? 1.0f : -1.0f);
    }
    long long total = n * (ld / 4);
    if (total <= 0) return RBL_OK;
    long long g = (total + 255) / 256;
    if (g > (1 << 22)) g = 1 << 22;
    if (storage == RBL_STORE_F32)
        hipLaunchKernelGGL(k_synth<float>, dim3((unsigned)g), dim3(256), 0, s, (float*)D, P, ysign);
    else
        hipLaunchKernelGGL(k_synth<double>, dim3((unsigned)g), dim3(256), 0, s, (double*)D, P, ysign);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
