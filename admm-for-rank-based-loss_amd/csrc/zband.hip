// zband.hip - the z-step (and the logged objective) for PIECEWISE-CONSTANT rank weights without a sort
// (superquantile, aorr, aorr_dc: up to two single-rank bands between two bands of several ranks).
//
// Reference path: src/optim/algorithms.py:96-104 (z_subproblem: argsort m, PAV_solver, unsort) with
// src/util/pav.py:84-161.  With weights that are constant on a few rank bands
//     sigma = [ s_0 x n_0 | s_1 x n_1 | ... ]         (src/optim/objective.py:108-145)
// the isotonic solution has a closed structure.  Inside a band the element prox u_i = prox_{s_j l / rho}(m_i) is
// non-decreasing in m_i, so the pool-adjacent-violators pass can only pool ACROSS a band edge, and what it pools
// there is one block: the top of the band below (u_i > x), the single-rank bands in between, the bottom of the band
// above (u_i < x), with the block value x the root of the pooled derivative
//     psi(x) = sum_{block(x)} sigma_i l'(x) + rho (x - m_i)      (monotone in x; pav.py:134-140 for a fixed block).
// Every element is then z_i = clamp(u_i, lo_band, hi_band) with the block values as clamps.  None of this needs
// the sorted order, only
//   1. the keys at a handful of ranks (the last / first rank of each band): an MSD radix SELECT, 11 bits per pass
//      over the 64-bit keys (integer histograms: exact and order-independent);
//   2. the root of psi: passes that evaluate the block sums for 16 candidate values at once (thresholds in m-space,
//      u_i > x  <=>  m_i > x + (s/rho) l'(x)); the first pass packs its candidates around a prediction from the last
//      three block values, the others shrink the bracket 15x each; once at most 2048 elements are undecided they are
//      gathered and one workgroup settles the block exactly (its value is pav.py's own formula);
//   3. one element-wise pass.
// All sums are accumulated per thread in a fixed element order and reduced in a fixed order: bit-reproducible.
// Whatever the fast path cannot certify (keys tied across a band edge, a block that swallows a whole band or stays
// on one side of a single-rank band, an unresolved bracket) is REPORTED through a pinned status word; the caller
// (api.hip: zb_resolve) then runs the sort + merge-tree PAV for that iteration.  6M rows, superquantile: ~0.35 ms
// instead of ~1.1 ms for sort + PAV + unsort.  The same select gives the logged objective sum_k sigma_k loss(v_(k))
// without sorting v (k_zb_risk).  CPU restatement of the structure and of the certification rules: oracle/zband.py.
#include "rbl_internal.h"
#include "device_math.h"

namespace {

constexpr int ZB_THREADS = 256;
constexpr int ZB_BINS = 1 << ZB_BITS;
constexpr int ZB_HTHREADS = 1024;   // select passes: one block per CU (every block ends with global atomics on the same few bins)

__device__ __forceinline__ int zb_pass_bits(int pass) { return pass < 5 ? ZB_BITS : 64 - 5 * ZB_BITS; }
__device__ __forceinline__ int zb_pass_shift(int pass) { return pass < 5 ? 64 - (pass + 1) * ZB_BITS : 0; }

// ------------------------------------------------------------------------------------------ select
__global__ void k_zb_init(ZbState* __restrict__ st, ZbConfig cfg, u32* __restrict__ hist) {
    const int t = threadIdx.x;
    if (t < ZB_MAX_TARGETS) {
        st->prefix[t] = 0;
        st->rem[t] = t < cfg.ntargets ? cfg.target_rank[t] : 0;
        st->group[t] = 0;
        st->gprefix[t] = 0;
        st->key[t] = 0;
    }
    if (t < ZB_MAX_CLUSTERS) {
        st->done[t] = 0;
        st->has_block[t] = 0;
        st->x[t] = 0.0;
        st->und[t] = 1e300;
        st->gcount[t] = 0;
    }
    if (t == 0) {
        st->ngroups = 1;
        st->status = ZB_OK;
    }
    for (int i = t; i < ZB_MAX_GROUPS * ZB_BINS; i += blockDim.x) hist[i] = 0;
}

// histogram of the pass's digit over the keys that still share a target's prefix
__global__ __launch_bounds__(ZB_HTHREADS) void k_zb_hist(const u64* __restrict__ keys, long long n,
                                                         const ZbState* __restrict__ st, u32* __restrict__ hist, int pass) {
    __shared__ u32 lh[ZB_MAX_GROUPS * ZB_BINS];
    __shared__ u64 gp[ZB_MAX_GROUPS];
    const int G = st->ngroups;
    if (st->status != ZB_OK) return;
    for (int i = threadIdx.x; i < G * ZB_BINS; i += ZB_HTHREADS) lh[i] = 0;
    if (threadIdx.x < G) gp[threadIdx.x] = st->gprefix[threadIdx.x];
    __syncthreads();
    const int bits = zb_pass_bits(pass), shift = zb_pass_shift(pass);
    const u32 mask = (1u << bits) - 1u;
    const long long stride = (long long)gridDim.x * ZB_HTHREADS;
    const long long n4 = n >> 2;
    const ulonglong2* __restrict__ k2 = reinterpret_cast<const ulonglong2*>(keys);
    if (pass == 0) {
        // sign + 10 exponent bits: a wave's 64 keys carry a handful of distinct digits, and 64 LDS atomics on one
        // counter serialise - one atomic per distinct digit and wave instead
        auto add = [&](u64 k, bool valid) {
            const u32 d = (u32)(k >> shift) & mask;
            u64 todo = __ballot(valid);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const u32 dl = (u32)__shfl((int)d, leader, 64);
                const u64 same = __ballot(valid && d == dl) & todo;
                if ((int)(threadIdx.x & 63) == leader) atomicAdd(&lh[dl], (u32)__popcll(same));
                todo &= ~same;
            }
        };
        // (uniform trip count: every lane of a wave runs the same number of steps, the ballots see whole waves)
        const long long first = (long long)blockIdx.x * ZB_HTHREADS + threadIdx.x;
        const long long steps = (n4 + stride - 1) / stride;
        for (long long it = 0; it < steps; ++it) {
            const long long i = first + it * stride;
            const bool v = i < n4;
            ulonglong2 a = {0, 0}, b = {0, 0};
            if (v) {
                a = k2[2 * i];
                b = k2[2 * i + 1];
            }
            add(a.x, v);
            add(a.y, v);
            add(b.x, v);
            add(b.y, v);
        }
        if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) atomicAdd(&lh[(u32)(keys[4 * n4 + threadIdx.x] >> shift) & mask], 1u);
    } else {
        const int up = shift + bits;
        auto add = [&](u64 k) {
            const u64 hi = k >> up;
            for (int g = 0; g < G; ++g)
                if (hi == gp[g]) atomicAdd(&lh[g * ZB_BINS + ((u32)(k >> shift) & mask)], 1u);
        };
        for (long long i = (long long)blockIdx.x * ZB_HTHREADS + threadIdx.x; i < n4; i += stride) {
            const ulonglong2 a = k2[2 * i], b = k2[2 * i + 1];
            add(a.x);
            add(a.y);
            add(b.x);
            add(b.y);
        }
        if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) add(keys[4 * n4 + threadIdx.x]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < G * ZB_BINS; i += ZB_HTHREADS) {
        const u32 c = lh[i];
        if (c) atomicAdd(&hist[i], c);
    }
}

// u_i > x  <=>  m_i > zb_theta_gt(x);   u_i < x  <=>  m_i < zb_theta_lt(x)     (u_i = prox_{s l / rho}(m_i))
template <int LOSS>
__device__ inline double zb_theta_gt(double s_over_rho, double x) {
    if (LOSS == 0) return x + s_over_rho * rbl::sigmoid1(x);
    return x >= -1.0 ? x + s_over_rho : x;
}
template <int LOSS>
__device__ inline double zb_theta_lt(double s_over_rho, double x) {
    if (LOSS == 0) return x + s_over_rho * rbl::sigmoid1(x);
    return x > -1.0 ? x + s_over_rho : x;
}
template <int LOSS>
__device__ inline double zb_block_value(double A, double M, double cnt, double rho) {   // pav.py:134-140
    return rbl::prox<LOSS>(A / cnt, rho, M / cnt);
}
template <int LOSS>
__device__ inline double zb_psi(double A, double M, double cnt, double rho, double x) {
    if (!(cnt > 0.0)) return 0.0;
    if (LOSS == 0) return A * rbl::sigmoid1(x) + rho * (cnt * x - M);
    return x - zb_block_value<1>(A, M, cnt, rho);
}

// candidates a = x_0 < ... < x_{C-1} = b.  Hinge: the prox has a plateau at the kink (u = -1 for a whole range of m), so a
// block value of exactly -1 with thousands of elements tied at it is the common case; psi then jumps across 0 AT -1.
// With -1 and its two floating-point neighbours as consecutive candidates the jump shows up as a sign change between
// two adjacent doubles, which k_zb_refine accepts as the root -1.
template <int LOSS>
__device__ inline void zb_candidates(double* cand, double a, double b) {
    for (int c = 0; c < ZB_C; ++c) cand[c] = c == ZB_C - 1 ? b : a + (b - a) * ((double)c / (ZB_C - 1));
    if (LOSS == 1 && a <= -1.0 && -1.0 <= b && a < b) {
        const double km = nextafter(-1.0, -2.0), kp = nextafter(-1.0, 0.0);
        if (a == -1.0) {          // the chain starts on the plateau: only the side above the kink is open
            if (kp < cand[2]) cand[1] = kp;
        } else if (b == -1.0) {
            if (km > cand[ZB_C - 3]) cand[ZB_C - 2] = km;
        } else {
            int c = (int)((-1.0 - a) / (b - a) * (ZB_C - 1) + 0.5);
            c = c < 2 ? 2 : (c > ZB_C - 3 ? ZB_C - 3 : c);
            // (a neighbour may coincide with a bracket end: duplicates are harmless, the sign change is looked for at the
            // first candidate with psi >= 0)
            if (cand[c - 2] <= km && kp <= cand[c + 2]) {
                cand[c - 1] = fmax(km, cand[c - 2]);
                cand[c] = -1.0;
                cand[c + 1] = fmin(kp, cand[c + 2]);
            }
        }
    }
}

// first pass of an iteration: the block value moves smoothly from one ADMM iteration to the next (rho grows by 2 %,
// w by a CG step), so the candidates are packed around the prediction xp (quadratic through the last three values,
// linear through two) at distances d, 2d, ... 64d on both sides, d a quarter of the last prediction's miss (or
// |dx|/256), with the chain bounds a, b as the outermost candidates: the bracket is valid whatever the prediction is
// worth, and a good one leaves a few hundred undecided elements after ONE pass over the keys.
template <int LOSS>
__device__ inline bool zb_candidates_warm(double* cand, double a, double b, const double* xh, int nh) {
    if (nh < 2) return false;
    double xp, d;
    if (nh >= 3) {
        xp = 3.0 * (xh[0] - xh[1]) + xh[2];
        d = 0.25 * fabs(xh[0] - 2.0 * xh[1] + xh[2]);
    } else {
        xp = 2.0 * xh[0] - xh[1];
        d = fabs(xh[0] - xh[1]) * (1.0 / 256.0);
    }
    d = fmax(d, fabs(xh[0] - xh[1]) * 1e-4);
    d = fmax(d, (b - a) * 1e-9);
    d = fmin(d, fmin(xp - a, b - xp) * (1.0 / 128.0));
    if (!(d > 0.0) || !(xp > a && xp < b)) return false;
    if (LOSS == 1 && a <= -1.0 && -1.0 <= b) return false;   // the kink has to be a candidate: uniform grid
    cand[0] = a;
    cand[ZB_C - 1] = b;
    double w = d;
    for (int i = 0; i < 7; ++i, w *= 2.0) {
        cand[7 - i] = xp - w;
        cand[8 + i] = xp + w;
    }
    for (int c = 1; c < ZB_C; ++c)
        if (!(cand[c] > cand[c - 1])) return false;
    return true;
}

// after the last select pass: keys of all targets are known -> which clusters can pool at all, and the first bracket
template <int LOSS>
__device__ void zb_cluster_setup(ZbState* st, const ZbConfig& cfg, double rho) {
    // consecutive ranks must carry strictly increasing keys: band membership is then a key comparison
    for (int t = 0; t + 1 < cfg.ntargets; ++t)
        if (cfg.target_rank[t + 1] == cfg.target_rank[t] + 1 && !(st->key[t] < st->key[t + 1])) {
            st->status = ZB_TIE;
            return;
        }
    for (int k = 0; k < cfg.nclusters; ++k) {
        const int L = cfg.cl_L[k], R = cfg.cl_R[k];
        if (!cfg.cl_root[k]) {
            st->done[k] = 1;
            continue;
        }
        // chain of element prox values across the edge: last of L, the single-rank bands, first of R
        double prev = rbl::prox<LOSS>(cfg.sigma[L], rho, rbl::unflip_key(st->key[cfg.last_t[L]]));
        double lo = prev, hi = prev;
        bool violated = false;
        for (int j = L + 1; j <= R; ++j) {
            const double u = rbl::prox<LOSS>(cfg.sigma[j], rho, rbl::unflip_key(st->key[cfg.first_t[j]]));
            if (u < prev) violated = true;
            lo = fmin(lo, u);
            hi = fmax(hi, u);
            prev = u;
        }
        if (!violated) {
            st->done[k] = 1;   // nothing to pool at this edge
            st->nh[k] = 0;
            continue;
        }
        if (!zb_candidates_warm<LOSS>(st->cand[k], lo, hi, st->xh[k], st->nh[k]))
            zb_candidates<LOSS>(st->cand[k], lo, hi);
    }
}

// one block: the digit of every target inside its bucket, new prefixes, regrouping; clears the histogram
template <int LOSS>
__global__ __launch_bounds__(1024) void k_zb_scan(ZbState* __restrict__ st, ZbConfig cfg, u32* __restrict__ hist, int pass,
                                                   double rho, int zstep) {
    __shared__ u64 newp[ZB_MAX_TARGETS];
    __shared__ long long newr[ZB_MAX_TARGETS];
    __shared__ int bad;
    if (st->status != ZB_OK) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bits = zb_pass_bits(pass);
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    if (wave < cfg.ntargets) {
        const int t = wave, g = st->group[t];
        const u32* hg = hist + g * ZB_BINS;
        const long long r = st->rem[t];
        constexpr int PER = ZB_BINS / 64;
        long long mine = 0;
        for (int b = 0; b < PER; ++b) mine += hg[lane * PER + b];
        long long incl = mine;   // inclusive scan over the lanes
        for (int off = 1; off < 64; off <<= 1) {
            const long long o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        const long long excl = incl - mine;
        const bool here = r >= excl && r < incl;
        const u64 who = __ballot(here);
        if (who == 0) {
            if (lane == 0) bad = 1;   // rank outside the bucket: cannot happen with a consistent histogram
        } else if (here) {
            long long cum = excl;
            int bin = 0;
            for (int b = 0; b < PER; ++b) {
                const long long c = hg[lane * PER + b];
                if (r < cum + c) {
                    bin = lane * PER + b;
                    break;
                }
                cum += c;
            }
            newp[t] = (st->prefix[t] << bits) | (u64)bin;
            newr[t] = r - cum;
            if (pass == 5) st->eq[t] = hg[bin];   // keys equal to the target's key; newr = the target's rank among them
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (bad) {
            st->status = ZB_BAD;
        } else {
            int G = 0;
            for (int t = 0; t < cfg.ntargets; ++t) {
                st->prefix[t] = newp[t];
                st->rem[t] = newr[t];
                int g = -1;
                for (int q = 0; q < G; ++q)
                    if (st->gprefix[q] == newp[t]) g = q;
                if (g < 0) {
                    if (G == ZB_MAX_GROUPS) {
                        st->status = ZB_GROUPS;
                        break;
                    }
                    st->gprefix[G] = newp[t];
                    g = G++;
                }
                st->group[t] = g;
            }
            st->ngroups = G;
            if (pass == 5 && st->status == ZB_OK) {
                for (int t = 0; t < cfg.ntargets; ++t) st->key[t] = st->prefix[t];
                if (zstep) zb_cluster_setup<LOSS>(st, cfg, rho);
            }
        }
    }
    for (int i = threadIdx.x; i < ZB_MAX_GROUPS * ZB_BINS; i += blockDim.x) hist[i] = 0;
}

// What a certified block may look like.  A whole band may be pooled only where nothing lies beyond it (the first / the
// last band).  The root was computed for "top of L + every single-rank band + bottom of R"; that IS the pooled block
// of pav.py iff every prefix of it pools to a value >= x (else the prefix would stay a block of its own):
//   no single-rank band:   both sides must be present (one band alone never pools);
//   one (e):               both sides present - always; only the top: u_e <= x; only the bottom: u_e >= x;
//   two (e1, e2; aorr_dc): both sides present - the prefix "top of L + e1" must pool to >= x (equivalently
//                          "e2 + bottom of R" to <= x); only the top: u_e2 <= x; only the bottom or neither: u_e1 >= x.
// cT, mT / cB, mB: count and sum of m of the elements of L / R in the block.
template <int LOSS>
__device__ void zb_accept(ZbState* st, const ZbConfig& cfg, int k, double rho, double x, double cT, double mT, double cB,
                          double mB) {
    const int L = cfg.cl_L[k], R = cfg.cl_R[k];
    const double sizeL = (double)(cfg.start[L + 1] - cfg.start[L]), sizeR = (double)(cfg.start[R + 1] - cfg.start[R]);
    if (!(cT < sizeL || L == 0)) {
        st->status = ZB_SWALLOW_L;
        return;
    }
    if (!(cB < sizeR || R == cfg.nbands - 1)) {
        st->status = ZB_SWALLOW_R;
        return;
    }
    const int nt = R - L - 1;
    bool ok;
    if (nt == 0) {
        ok = cT > 0.0 && cB > 0.0;
    } else {
        const double s1 = cfg.sigma[L + 1], m1 = rbl::unflip_key(st->key[cfg.first_t[L + 1]]);
        const double s2 = cfg.sigma[R - 1], m2 = rbl::unflip_key(st->key[cfg.first_t[R - 1]]);   // (nt == 1: the same band)
        const double u1 = rbl::prox<LOSS>(s1, rho, m1), u2 = rbl::prox<LOSS>(s2, rho, m2);
        if (cT > 0.0 && cB > 0.0) {
            ok = true;
            if (nt == 2) {
                const double xl = zb_block_value<LOSS>(cfg.sigma[L] * cT + s1, mT + m1, cT + 1.0, rho);
                const double xr = zb_block_value<LOSS>(s2 + cfg.sigma[R] * cB, m2 + mB, 1.0 + cB, rho);
                ok = xl >= x && xr <= x;
            }
        } else if (cT > 0.0) {
            ok = u2 <= x;
        } else {
            ok = u1 >= x;
        }
    }
    if (!ok) {
        st->status = ZB_ONESIDED;
        return;
    }
    st->x[k] = x;
    st->xh[k][2] = st->xh[k][1];
    st->xh[k][1] = st->xh[k][0];
    st->xh[k][0] = x;
    st->nh[k] = st->nh[k] < 3 ? st->nh[k] + 1 : 3;
    st->has_block[k] = 1;
    st->done[k] = 1;
}

// ------------------------------------------------------------------------------------------ root of psi
// band j holds the keys in (lo_key, hi_key]
__device__ inline void zb_band_keys(const ZbState* st, const ZbConfig& cfg, int j, u64& lo_excl, bool& has_lo, u64& hi_incl) {
    has_lo = j > 0;
    lo_excl = j > 0 ? st->key[cfg.last_t[j - 1]] : 0ull;
    hi_incl = j < cfg.nbands - 1 ? st->key[cfg.last_t[j]] : ~0ull;
}

// partials[block][4][ZB_C]: sum m / count of the top part of band L (u > x_c), of the bottom part of band R (u < x_c)
template <int LOSS>
__global__ __launch_bounds__(ZB_THREADS) void k_zb_eval(const u64* __restrict__ keys, long long n,
                                                         const ZbState* __restrict__ st, ZbConfig cfg, int k, double rho,
                                                         double* __restrict__ partials) {
    if (st->status != ZB_OK || st->done[k] || st->und[k] <= (double)ZB_GCAP) return;
    __shared__ double red[(ZB_THREADS / 64) * 4 * ZB_C];
    const int L = cfg.cl_L[k], R = cfg.cl_R[k];
    u64 Llo, Lhi, Rlo, Rhi;
    bool Lhas, Rhas;
    zb_band_keys(st, cfg, L, Llo, Lhas, Lhi);
    zb_band_keys(st, cfg, R, Rlo, Rhas, Rhi);
    // thresholds in m-space (uniform): LDS, only the outermost ones stay in registers for the fast paths
    __shared__ double thT[ZB_C], thB[ZB_C];
    const double sl = cfg.sigma[L] / rho, sr = cfg.sigma[R] / rho;
    if (threadIdx.x < ZB_C) {
        const double x = st->cand[k][threadIdx.x];
        thT[threadIdx.x] = zb_theta_gt<LOSS>(sl, x);
        thB[threadIdx.x] = zb_theta_lt<LOSS>(sr, x);
    }
    __syncthreads();
    // the outer thresholds stay in registers: with a warm-started first pass (candidates packed around the predicted
    // root, zb_candidates_warm) nearly every element falls into one of four classes that need no loop
    const double tT0 = thT[0], tT1 = thT[1], tT2 = thT[ZB_C - 2], tT3 = thT[ZB_C - 1];
    const double tB0 = thB[0], tB1 = thB[1], tB2 = thB[ZB_C - 2], tB3 = thB[ZB_C - 1];
    double MT[ZB_C], MB[ZB_C], NT[ZB_C], NB[ZB_C];
#pragma unroll
    for (int c = 0; c < ZB_C; ++c) MT[c] = MB[c] = NT[c] = NB[c] = 0.0;
    // top side: all candidates / all but the last / only the first; bottom side: all / all but the first / only the last
    double MTall = 0.0, NTall = 0.0, MTabl = 0.0, NTabl = 0.0, MTfst = 0.0, NTfst = 0.0;
    double MBall = 0.0, NBall = 0.0, MBabf = 0.0, NBabf = 0.0, MBlst = 0.0, NBlst = 0.0;
    auto one = [&](u64 key) {
        if (key <= Lhi && (!Lhas || key > Llo)) {
            const double m = rbl::unflip_key(key);
            if (m > tT3) {
                MTall += m;
                NTall += 1.0;
            } else if (m > tT2) {
                MTabl += m;
                NTabl += 1.0;
            } else if (m > tT1) {
#pragma unroll
                for (int c = 0; c < ZB_C; ++c) {
                    const bool in = m > thT[c];
                    MT[c] += in ? m : 0.0;
                    NT[c] += in ? 1.0 : 0.0;
                }
            } else if (m > tT0) {
                MTfst += m;
                NTfst += 1.0;
            }
        } else if (key > Rlo && key <= Rhi) {   // R > 0 always: Rlo is a real key
            const double m = rbl::unflip_key(key);
            if (m < tB0) {
                MBall += m;
                NBall += 1.0;
            } else if (m < tB1) {
                MBabf += m;
                NBabf += 1.0;
            } else if (m < tB2) {
#pragma unroll
                for (int c = 0; c < ZB_C; ++c) {
                    const bool in = m < thB[c];
                    MB[c] += in ? m : 0.0;
                    NB[c] += in ? 1.0 : 0.0;
                }
            } else if (m < tB3) {
                MBlst += m;
                NBlst += 1.0;
            }
        }
    };
    // 4 keys per thread and step (two 16-byte loads in flight), tail one by one; the element order per thread is fixed
    const long long n4 = n >> 2;
    const ulonglong2* __restrict__ k2 = reinterpret_cast<const ulonglong2*>(keys);
    const long long stride = (long long)gridDim.x * ZB_THREADS;
    for (long long i = (long long)blockIdx.x * ZB_THREADS + threadIdx.x; i < n4; i += stride) {
        const ulonglong2 a = k2[2 * i], b = k2[2 * i + 1];
        one(a.x);
        one(a.y);
        one(b.x);
        one(b.y);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) one(keys[4 * n4 + threadIdx.x]);
#pragma unroll
    for (int c = 0; c < ZB_C; ++c) {
        MT[c] += MTall + (c < ZB_C - 1 ? MTabl : 0.0) + (c == 0 ? MTfst : 0.0);
        NT[c] += NTall + (c < ZB_C - 1 ? NTabl : 0.0) + (c == 0 ? NTfst : 0.0);
        MB[c] += MBall + (c > 0 ? MBabf : 0.0) + (c == ZB_C - 1 ? MBlst : 0.0);
        NB[c] += NBall + (c > 0 ? NBabf : 0.0) + (c == ZB_C - 1 ? NBlst : 0.0);
    }
    // fixed-order reduction: lanes (butterfly), waves (in order), blocks (k_zb_refine, in order)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < ZB_C; ++c) {
        double a = MT[c], b = NT[c], e = MB[c], f = NB[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_xor(a, off, 64);
            b += __shfl_xor(b, off, 64);
            e += __shfl_xor(e, off, 64);
            f += __shfl_xor(f, off, 64);
        }
        if (lane == 0) {
            red[(wave * 4 + 0) * ZB_C + c] = a;
            red[(wave * 4 + 1) * ZB_C + c] = b;
            red[(wave * 4 + 2) * ZB_C + c] = e;
            red[(wave * 4 + 3) * ZB_C + c] = f;
        }
    }
    __syncthreads();
    if (threadIdx.x < 4 * ZB_C) {
        double s = 0.0;
        for (int w = 0; w < ZB_THREADS / 64; ++w) s += red[w * 4 * ZB_C + threadIdx.x];
        partials[(size_t)blockIdx.x * 4 * ZB_C + threadIdx.x] = s;
    }
}

// psi at the candidates of one root pass (tot: the summed block sums), the next bracket or the block value - one thread
template <int LOSS>
__device__ inline void zb_decide(ZbState* __restrict__ st, const ZbConfig& cfg, int k, double rho, const double* tot, int last) {
    const int L = cfg.cl_L[k], R = cfg.cl_R[k];
    const double* MT = tot;
    const double* NT = tot + ZB_C;
    const double* MB = tot + 2 * ZB_C;
    const double* NB = tot + 3 * ZB_C;
    // the single-rank bands between L and R belong to every candidate block
    double At = 0.0, Mt = 0.0, nt = 0.0;
    for (int j = L + 1; j < R; ++j) {
        At += cfg.sigma[j];
        Mt += rbl::unflip_key(st->key[cfg.first_t[j]]);
        nt += 1.0;
    }
    const double sL = cfg.sigma[L], sR = cfg.sigma[R];
    double psi[ZB_C];
    for (int c = 0; c < ZB_C; ++c)
        psi[c] = zb_psi<LOSS>(sL * NT[c] + At + sR * NB[c], MT[c] + Mt + MB[c], NT[c] + nt + NB[c], rho, st->cand[k][c]);
    int c1 = -1;   // first candidate with psi >= 0
    for (int c = 0; c < ZB_C; ++c)
        if (psi[c] >= 0.0) {
            c1 = c;
            break;
        }
    double x = 0.0, cT = 0.0, cB = 0.0, mT = 0.0, mB = 0.0;
    bool found = false;
    if (c1 < 0 || (c1 == 0 && psi[0] > 0.0)) {
        st->status = ZB_BRACKET;   // the root is not inside the bracket: cannot happen for a consistent chain
        return;
    }
    const double kink_m = nextafter(-1.0, -2.0), kink_p = nextafter(-1.0, 0.0);
    if (LOSS == 1 && c1 > 0 && psi[c1] > 0.0 &&
        ((st->cand[k][c1 - 1] == -1.0 && st->cand[k][c1] == kink_p) || (st->cand[k][c1 - 1] == kink_m && st->cand[k][c1] == -1.0))) {
        // psi < 0 just below, > 0 just above: 0 lies in the subdifferential at the kink, the block value is -1 and the
        // elements whose prox sits on the plateau are part of the block (top part as left of -1, bottom part as right of it)
        x = -1.0;
        cT = NT[c1 - 1];
        mT = MT[c1 - 1];
        cB = NB[c1];
        mB = MB[c1];
        found = true;
    } else if (psi[c1] == 0.0) {
        x = st->cand[k][c1];
        cT = NT[c1];
        mT = MT[c1];
        cB = NB[c1];
        mB = MB[c1];
        found = true;
    } else {
        const int c0 = c1 - 1;
        const double undecided = (NT[c0] - NT[c1]) + (NB[c1] - NB[c0]);
        if (undecided == 0.0) {
            // no element changes sides inside (x_c0, x_c1): top part as at x_c1, bottom part as at x_c0
            cT = NT[c1];
            mT = MT[c1];
            cB = NB[c0];
            mB = MB[c0];
            const double cnt = cT + nt + cB;
            x = zb_block_value<LOSS>(sL * cT + At + sR * cB, mT + Mt + mB, cnt, rho);
            if (!(x >= st->cand[k][c0] && x <= st->cand[k][c1])) {
                st->status = ZB_BRACKET;
                return;
            }
            found = true;
        } else {
            // still undecided elements inside (a, b): remember the bracket and what lies outside of it for certain; few
            // enough: k_zb_gather / k_zb_finish settle it exactly, otherwise the next pass subdivides the bracket
            const double a = st->cand[k][c0], b = st->cand[k][c1];
            st->br[k][0] = a;
            st->br[k][1] = b;
            st->frozen[k][0] = NT[c1];
            st->frozen[k][1] = MT[c1];
            st->frozen[k][2] = NB[c0];
            st->frozen[k][3] = MB[c0];
            st->und[k] = undecided;
            if (undecided > (double)ZB_GCAP) {
                if (last || !(b > a)) {
                    st->status = ZB_UNRESOLVED;   // too many elements tied at the root (hinge plateau) or too dense
                    return;
                }
                zb_candidates<LOSS>(st->cand[k], a, b);
            }
        }
    }
    if (found) zb_accept<LOSS>(st, cfg, k, rho, x, cT, mT, cB, mB);
}

// one block: sums over the eval blocks, psi at the candidates, the next bracket or the block value
template <int LOSS>
// Across GPUs the two halves run as separate launches with an all-reduce in between: tot_out != NULL stops after the
// local sums (written there, zeros when this cluster is settled), tot_in != NULL starts from the summed totals.
__global__ __launch_bounds__(1024) void k_zb_refine(ZbState* __restrict__ st, ZbConfig cfg, int k, double rho,
                                                     const double* __restrict__ partials, int nblocks, int last,
                                                     double* __restrict__ tot_out, const double* __restrict__ tot_in,
                                                     int* pin_settled = nullptr, int dseq = 0) {
    // pin_settled (multi-GPU driver, pinned host memory): [1] = "this cluster needs no further root pass" (settled, few
    // enough undecided elements for the gather, or the step has failed), then [0] = dseq.  The state is a function of the
    // SUMMED totals, so every rank reads the same answer and stops issuing all-reduces after the pass that settles.
    const bool idle = st->status != ZB_OK || st->done[k] || st->und[k] <= (double)ZB_GCAP;
    if (idle) {
        if (tot_out && threadIdx.x < 4 * ZB_C) tot_out[threadIdx.x] = 0.0;
        if (pin_settled && threadIdx.x == 0) {
            pin_settled[1] = 1;
            __threadfence_system();
            *reinterpret_cast<volatile int*>(pin_settled) = dseq;
        }
        return;
    }
    __shared__ double tot[4 * ZB_C];
    if (tot_in) {
        if (threadIdx.x < 4 * ZB_C) tot[threadIdx.x] = tot_in[threadIdx.x];
        __syncthreads();
    } else {
        // 16 threads per value (coalesced over the 64 values): thread (v, part) sums blocks part, part + 16, ... in
        // order, then the 16 partial sums in order
        const int v = threadIdx.x & 63, part = threadIdx.x >> 6;
        double s = 0.0;
#pragma unroll 4
        for (int b = part; b < nblocks; b += 16) s += partials[(size_t)b * 4 * ZB_C + v];
        __shared__ double tmp[4 * ZB_C * 16];
        tmp[part * 64 + v] = s;
        __syncthreads();
        if (part == 0) {
            double a = 0.0;
            for (int p = 0; p < 16; ++p) a += tmp[p * 64 + v];
            tot[v] = a;
        }
        __syncthreads();
    }
    if (tot_out) {
        if (threadIdx.x < 4 * ZB_C) tot_out[threadIdx.x] = tot[threadIdx.x];
        return;
    }
    if (threadIdx.x != 0) return;
    zb_decide<LOSS>(st, cfg, k, rho, tot, last);
    if (pin_settled) {
        pin_settled[1] = (st->status != ZB_OK || st->done[k] || st->und[k] <= (double)ZB_GCAP) ? 1 : 0;
        __threadfence_system();
        *reinterpret_cast<volatile int*>(pin_settled) = dseq;
    }
}

// ------------------------------------------------------------------------------------------ gather + finish
// the undecided elements of the bracket (at most ZB_GCAP): their m, in any order (k_zb_finish sorts them)
template <int LOSS>
__global__ __launch_bounds__(ZB_THREADS) void k_zb_gather(const u64* __restrict__ keys, long long n, ZbState* __restrict__ st,
                                                           ZbConfig cfg, int k, double rho, double* __restrict__ list) {
    if (st->status != ZB_OK || st->done[k]) return;
    if (st->und[k] > (double)ZB_GCAP) return;   // k_zb_finish reports it
    const int L = cfg.cl_L[k], R = cfg.cl_R[k];
    u64 Llo, Lhi, Rlo, Rhi;
    bool Lhas, Rhas;
    zb_band_keys(st, cfg, L, Llo, Lhas, Lhi);
    zb_band_keys(st, cfg, R, Rlo, Rhas, Rhi);
    const double a = st->br[k][0], b = st->br[k][1];
    const double sl = cfg.sigma[L] / rho, sr = cfg.sigma[R] / rho;
    const double tTa = zb_theta_gt<LOSS>(sl, a), tTb = zb_theta_gt<LOSS>(sl, b);   // top: in the block at a, not at b
    const double tBa = zb_theta_lt<LOSS>(sr, a), tBb = zb_theta_lt<LOSS>(sr, b);   // bottom: in the block at b, not at a
    auto one = [&](u64 key) {
        bool take = false;
        double m = 0.0;
        if (key <= Lhi && (!Lhas || key > Llo)) {
            m = rbl::unflip_key(key);
            take = m > tTa && !(m > tTb);
        } else if (key > Rlo && key <= Rhi) {
            m = rbl::unflip_key(key);
            take = m < tBb && !(m < tBa);
        }
        if (take) {
            const int slot = atomicAdd(&st->gcount[k], 1);
            if (slot < ZB_GCAP) list[(size_t)k * ZB_GCAP + slot] = m;
        }
    };
    const long long n4 = n >> 2;
    const ulonglong2* __restrict__ k2 = reinterpret_cast<const ulonglong2*>(keys);
    const long long stride = (long long)gridDim.x * ZB_THREADS;
    for (long long i = (long long)blockIdx.x * ZB_THREADS + threadIdx.x; i < n4; i += stride) {
        const ulonglong2 p = k2[2 * i], q = k2[2 * i + 1];
        one(p.x);
        one(p.y);
        one(q.x);
        one(q.y);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) one(keys[4 * n4 + threadIdx.x]);
}

// inclusive scan of v[0 .. ZB_GCAP) in LDS, 2 entries per thread, fixed order
__device__ inline void zb_block_scan(double* v, double* wsum) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const double a = v[2 * t], b = a + v[2 * t + 1];
    double incl = b;
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    double base = 0.0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    const double excl = base + (incl - b);
    v[2 * t] = excl + a;
    v[2 * t + 1] = excl + b;
    __syncthreads();
}

// one block: the undecided elements sorted by their prox value, psi at every one of them from prefix sums, the block
template <int LOSS>
__global__ __launch_bounds__(1024) void k_zb_finish(ZbState* __restrict__ st, ZbConfig cfg, int k, double rho,
                                                     const double* __restrict__ list) {
    static_assert(ZB_GCAP == 2048, "two entries per thread of a 1024-thread block");
    if (st->status != ZB_OK || st->done[k]) return;
    __shared__ double su[ZB_GCAP], sm[ZB_GCAP], pTc[ZB_GCAP], pTm[ZB_GCAP], pBc[ZB_GCAP], pBm[ZB_GCAP];
    __shared__ double wsum[16];
    __shared__ int pstar;
    const int t = threadIdx.x;
    const int cnt = st->gcount[k];
    if (st->und[k] > (double)ZB_GCAP || cnt > ZB_GCAP || (double)cnt != st->und[k]) {
        if (t == 0) st->status = st->und[k] > (double)ZB_GCAP ? ZB_UNRESOLVED : ZB_BAD;
        return;
    }
    const int L = cfg.cl_L[k], R = cfg.cl_R[k];
    const double sL = cfg.sigma[L], sR = cfg.sigma[R];
    const double mLhi = rbl::unflip_key(st->key[cfg.last_t[L]]);   // m <= mLhi: an element of band L (top side)
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    for (int i = t; i < ZB_GCAP; i += 1024) {
        double m = 0.0, u = inf;
        if (i < cnt) {
            m = list[(size_t)k * ZB_GCAP + i];
            u = rbl::prox<LOSS>(m <= mLhi ? sL : sR, rho, m);
        }
        su[i] = u;
        sm[i] = m;
    }
    if (t == 0) pstar = ZB_GCAP;
    __syncthreads();
    // bitonic sort by (u, m) of the first N >= cnt entries (the rest is +inf already): the gather order is arbitrary,
    // the sums below must not depend on it
    int N = 2;
    while (N < cnt) N <<= 1;
    for (int size = 2; size <= N; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (2 * t < N) {
                const int i = 2 * t - (t & (stride - 1));   // lower index of this thread's pair
                const int j = i + stride;
                const bool up = (i & size) == 0;
                const double ui = su[i], uj = su[j], mi = sm[i], mj = sm[j];
                const bool gt = ui > uj || (ui == uj && mi > mj);
                if (gt == up) {
                    su[i] = uj;
                    su[j] = ui;
                    sm[i] = mj;
                    sm[j] = mi;
                }
            }
            __syncthreads();
        }
    for (int i = t; i < ZB_GCAP; i += 1024) {
        const bool valid = i < cnt, top = valid && sm[i] <= mLhi, bot = valid && !top;
        pTc[i] = top ? 1.0 : 0.0;
        pTm[i] = top ? sm[i] : 0.0;
        pBc[i] = bot ? 1.0 : 0.0;
        pBm[i] = bot ? sm[i] : 0.0;
    }
    __syncthreads();
    zb_block_scan(pTc, wsum);
    zb_block_scan(pTm, wsum);
    zb_block_scan(pBc, wsum);
    zb_block_scan(pBm, wsum);
    double At = 0.0, Mt = 0.0, nt = 0.0;
    for (int j = L + 1; j < R; ++j) {
        At += cfg.sigma[j];
        Mt += rbl::unflip_key(st->key[cfg.first_t[j]]);
        nt += 1.0;
    }
    const double fTc = st->frozen[k][0], fTm = st->frozen[k][1], fBc = st->frozen[k][2], fBm = st->frozen[k][3];
    const double totTc = cnt ? pTc[cnt - 1] : 0.0, totTm = cnt ? pTm[cnt - 1] : 0.0;
    // psi at x = su[p]: top entries with u > x (behind p's group of equal u), bottom entries with u < x (before the group)
    auto sets_at = [&](int p, bool with_group, double& cT, double& mT, double& cB, double& mB) {
        int lo = p, hi = p;
        while (lo > 0 && su[lo - 1] == su[p]) --lo;
        while (hi + 1 < cnt && su[hi + 1] == su[p]) ++hi;
        const int tfrom = with_group ? lo : hi + 1;   // top entries at positions >= tfrom
        cT = fTc + totTc - (tfrom > 0 ? pTc[tfrom - 1] : 0.0);
        mT = fTm + totTm - (tfrom > 0 ? pTm[tfrom - 1] : 0.0);
        cB = fBc + (lo > 0 ? pBc[lo - 1] : 0.0);
        mB = fBm + (lo > 0 ? pBm[lo - 1] : 0.0);
        return lo;
    };
    for (int p = t; p < cnt; p += 1024) {
        double cT, mT, cB, mB;
        sets_at(p, false, cT, mT, cB, mB);
        const double psi = zb_psi<LOSS>(sL * cT + At + sR * cB, mT + Mt + mB, cT + nt + cB, rho, su[p]);
        if (psi >= 0.0) atomicMin(&pstar, p);
    }
    __syncthreads();
    if (t != 0) return;
    const double a = st->br[k][0], b = st->br[k][1];
    double x, cT, mT, cB, mB, lo_x, hi_x;
    if (pstar == ZB_GCAP) {
        // psi < 0 at every undecided element: the root lies behind the last one - no top entry, every bottom entry
        cT = fTc;
        mT = fTm;
        cB = fBc + (cnt ? pBc[cnt - 1] : 0.0);
        mB = fBm + (cnt ? pBm[cnt - 1] : 0.0);
        lo_x = cnt ? su[cnt - 1] : a;
        hi_x = b;
        x = zb_block_value<LOSS>(sL * cT + At + sR * cB, mT + Mt + mB, cT + nt + cB, rho);
    } else {
        const int p = pstar;
        sets_at(p, false, cT, mT, cB, mB);
        if (zb_psi<LOSS>(sL * cT + At + sR * cB, mT + Mt + mB, cT + nt + cB, rho, su[p]) == 0.0) {
            x = lo_x = hi_x = su[p];
        } else {
            // the root lies before su[p]: the group at su[p] is still part of the top side, nothing of it of the bottom side
            const int lo = sets_at(p, true, cT, mT, cB, mB);
            lo_x = lo > 0 ? su[lo - 1] : a;
            hi_x = su[p];
            x = zb_block_value<LOSS>(sL * cT + At + sR * cB, mT + Mt + mB, cT + nt + cB, rho);
        }
    }
    // (closed interval: a hinge block on the plateau returns exactly -1, the value its tied elements sit at)
    if (!(x >= lo_x && x <= hi_x && x >= a && x <= b)) {
        st->status = ZB_BRACKET;
        return;
    }
    zb_accept<LOSS>(st, cfg, k, rho, x, cT, mT, cB, mB);
}

// ---- across GPUs: the undecided elements of all ranks
// pack[0] = how many this rank gathered, pack[1 ...] = their m
__global__ void k_zb_pack(const ZbState* __restrict__ st, int k, const double* __restrict__ list, double* __restrict__ pack) {
    const bool live = st->status == ZB_OK && !st->done[k] && st->und[k] <= (double)ZB_GCAP;
    const int cnt = live ? min(st->gcount[k], ZB_GCAP) : 0;
    if (threadIdx.x == 0 && blockIdx.x == 0) pack[0] = (double)cnt;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ZB_GCAP; i += gridDim.x * blockDim.x)
        pack[1 + i] = i < cnt ? list[(size_t)k * ZB_GCAP + i] : 0.0;
}
// the gathered packs of all ranks -> one list in rank order (the union holds und[k] <= ZB_GCAP elements)
__global__ __launch_bounds__(1024) void k_zb_union(ZbState* __restrict__ st, int k, const double* __restrict__ packs, int world,
                                                    double* __restrict__ list) {
    if (st->status != ZB_OK || st->done[k] || st->und[k] > (double)ZB_GCAP) return;
    __shared__ int off[65];
    if (threadIdx.x == 0) {
        int o = 0;
        for (int r = 0; r < world; ++r) {
            off[r] = o;
            o += (int)packs[(size_t)r * (ZB_GCAP + 1)];
        }
        off[world] = o;
        st->gcount[k] = o;
    }
    __syncthreads();
    if (off[world] > ZB_GCAP) return;   // k_zb_finish reports the inconsistency
    for (int r = 0; r < world; ++r) {
        const int c = off[r + 1] - off[r];
        for (int i = threadIdx.x; i < c; i += blockDim.x)
            list[(size_t)k * ZB_GCAP + off[r] + i] = packs[(size_t)r * (ZB_GCAP + 1) + 1 + i];
    }
}

// ------------------------------------------------------------------------------------------ objective
// sum_k sigma_k loss(v_(k)) (objective.py:73-81) for banded sigma: the band sums of the losses need the keys at the last
// rank of every band (the same select) and one pass; elements tied with a band-edge key are accounted for by their
// count (their losses are equal, so which of them falls on which side of the edge does not matter).
template <int LOSS>
__global__ __launch_bounds__(ZB_THREADS) void k_zb_risk(const u64* __restrict__ keys, long long n, const ZbState* __restrict__ st,
                                                         ZbConfig cfg, double* __restrict__ partials) {
    __shared__ u64 bk[ZB_MAX_BANDS];
    __shared__ double red[(ZB_THREADS / 64) * ZB_MAX_BANDS];
    const int B = cfg.nbands;
    if (threadIdx.x < B - 1) bk[threadIdx.x] = st->key[cfg.last_t[threadIdx.x]];
    __syncthreads();
    double acc[ZB_MAX_BANDS];
#pragma unroll
    for (int j = 0; j < ZB_MAX_BANDS; ++j) acc[j] = 0.0;
    auto one = [&](u64 key) {
        int j = 0;
        bool tie = false;
        for (int q = 0; q < B - 1; ++q) {
            j += key > bk[q] ? 1 : 0;
            tie = tie || key == bk[q];
        }
        if (tie) return;
        const double l = rbl::sample_loss<LOSS>(rbl::unflip_key(key));
#pragma unroll
        for (int q = 0; q < ZB_MAX_BANDS; ++q) acc[q] += q == j ? l : 0.0;
    };
    const long long n4 = n >> 2;
    const ulonglong2* __restrict__ k2 = reinterpret_cast<const ulonglong2*>(keys);
    const long long stride = (long long)gridDim.x * ZB_THREADS;
    for (long long i = (long long)blockIdx.x * ZB_THREADS + threadIdx.x; i < n4; i += stride) {
        const ulonglong2 a = k2[2 * i], b = k2[2 * i + 1];
        one(a.x);
        one(a.y);
        one(b.x);
        one(b.y);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) one(keys[4 * n4 + threadIdx.x]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < ZB_MAX_BANDS; ++j) {
        double a = acc[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
        if (lane == 0) red[wave * ZB_MAX_BANDS + j] = a;
    }
    __syncthreads();
    if (threadIdx.x < ZB_MAX_BANDS) {
        double s = 0.0;
        for (int w = 0; w < ZB_THREADS / 64; ++w) s += red[w * ZB_MAX_BANDS + threadIdx.x];
        partials[(size_t)blockIdx.x * ZB_MAX_BANDS + threadIdx.x] = s;
    }
}

template <int LOSS>
__global__ __launch_bounds__(256) void k_zb_risk_finish(const ZbState* __restrict__ st, ZbConfig cfg,
                                                         const double* __restrict__ partials, int nblocks, double* __restrict__ out) {
    __shared__ double tmp[32 * ZB_MAX_BANDS], tot[ZB_MAX_BANDS];
    const int v = threadIdx.x & (ZB_MAX_BANDS - 1), part = threadIdx.x / ZB_MAX_BANDS;   // 32 threads per band, in order
    double s = 0.0;
    for (int b = part; b < nblocks; b += 32) s += partials[(size_t)b * ZB_MAX_BANDS + v];
    tmp[part * ZB_MAX_BANDS + v] = s;
    __syncthreads();
    if (part == 0) {
        double a = 0.0;
        for (int p = 0; p < 32; ++p) a += tmp[p * ZB_MAX_BANDS + v];
        tot[v] = a;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double risk = 0.0;
    for (int j = 0; j < cfg.nbands; ++j) risk += cfg.sigma[j] * tot[j];
    // the elements tied with a band-edge key: ranks [first, first + eq) all carry the loss of that key
    u64 prev = 0;
    bool have_prev = false;
    for (int j = 0; j < cfg.nbands - 1; ++j) {
        const int t = cfg.last_t[j];
        const u64 key = st->key[t];
        if (have_prev && key == prev) continue;
        prev = key;
        have_prev = true;
        const long long first = cfg.target_rank[t] - st->rem[t], last = first + st->eq[t];
        const double l = rbl::sample_loss<LOSS>(rbl::unflip_key(key));
        double wsum = 0.0;
        for (int q = 0; q < cfg.nbands; ++q) {
            const long long a = first > cfg.start[q] ? first : cfg.start[q];
            const long long b = last < cfg.start[q + 1] ? last : cfg.start[q + 1];
            if (b > a) wsum += cfg.sigma[q] * (double)(b - a);
        }
        risk += wsum * l;
    }
    out[0] = st->status == ZB_OK ? risk : __longlong_as_double(0x7ff8000000000000ll);
}

// ------------------------------------------------------------------------------------------ apply
template <int LOSS>
__global__ __launch_bounds__(ZB_THREADS) void k_zb_apply(const double* __restrict__ m, long long n,
                                                          const ZbState* __restrict__ st, ZbConfig cfg, double rho,
                                                          double* __restrict__ z, const double* __restrict__ lam,
                                                          double* __restrict__ c, int* __restrict__ pin, int seq,
                                                          u32* __restrict__ counters) {
    __shared__ u64 bhi[ZB_MAX_BANDS];
    __shared__ double lo[ZB_MAX_BANDS], hi[ZB_MAX_BANDS], mlo[ZB_MAX_BANDS], mhi[ZB_MAX_BANDS], sg[ZB_MAX_BANDS];
    __shared__ int status;
    if (threadIdx.x == 0) {
        int s = st->status;
        for (int k = 0; k < cfg.nclusters; ++k)
            if (s == ZB_OK && !st->done[k]) s = ZB_UNRESOLVED;
        status = s;
        const double inf = __longlong_as_double(0x7ff0000000000000ll);
        for (int j = 0; j < cfg.nbands; ++j) {
            bhi[j] = j < cfg.nbands - 1 ? st->key[cfg.last_t[j]] : ~0ull;
            sg[j] = cfg.sigma[j];
            lo[j] = -inf;
            hi[j] = inf;
        }
        int blocks = 0;
        for (int k = 0; k < cfg.nclusters; ++k)
            if (st->has_block[k]) {
                const double x = st->x[k];
                hi[cfg.cl_L[k]] = x;
                lo[cfg.cl_R[k]] = x;
                for (int j = cfg.cl_L[k] + 1; j < cfg.cl_R[k]; ++j) lo[j] = hi[j] = x;
                ++blocks;
            }
        for (int j = 0; j < cfg.nbands; ++j)
            if (s == ZB_OK && !(lo[j] <= hi[j])) s = ZB_OVERLAP;   // the blocks at both ends of a band would meet
        status = s;
        for (int j = 0; j < cfg.nbands; ++j) {
            const double sr = cfg.sigma[j] / rho;
            mhi[j] = hi[j] < inf ? zb_theta_gt<LOSS>(sr, hi[j]) : inf;      // m above: u > hi -> z = hi
            mlo[j] = lo[j] > -inf ? zb_theta_lt<LOSS>(sr, lo[j]) : -inf;    // m below: u < lo -> z = lo
        }
        if (blockIdx.x == 0) {
            if (counters) counters[0] = (u32)blocks;
            __threadfence_system();
            reinterpret_cast<volatile int*>(pin)[1] = s;
            __threadfence_system();
            reinterpret_cast<volatile int*>(pin)[0] = seq;   // written last: the host polls this word
        }
    }
    __syncthreads();
    if (status != ZB_OK) return;   // the caller redoes the z-step with the sort
    const int B = cfg.nbands;
    auto one = [&](double mi) -> double {
        const u64 key = rbl::flip_key(mi);
        int j = 0;
        for (int q = 0; q < B - 1; ++q) j += key > bhi[q] ? 1 : 0;
        if (mi > mhi[j]) return hi[j];
        if (mi < mlo[j]) return lo[j];
        return fmin(fmax(sg[j] == 0.0 ? mi : rbl::prox_est<LOSS>(sg[j], rho, mi), lo[j]), hi[j]);
    };
    // z and, in the same pass, c = z + lambda/rho (algorithms.py:192; the sorted path forms it in rbl_phase_q)
    const long long n2 = n >> 1;
    const double2* __restrict__ m2 = reinterpret_cast<const double2*>(m);
    const double2* __restrict__ l2 = reinterpret_cast<const double2*>(lam);
    double2* __restrict__ z2 = reinterpret_cast<double2*>(z);
    double2* __restrict__ c2 = reinterpret_cast<double2*>(c);
    const long long stride = (long long)gridDim.x * ZB_THREADS;
    for (long long i = (long long)blockIdx.x * ZB_THREADS + threadIdx.x; i < n2; i += stride) {
        const double2 v = m2[i], l = l2[i];
        double2 r, cc;
        r.x = one(v.x);
        r.y = one(v.y);
        cc.x = r.x + l.x / rho;
        cc.y = r.y + l.y / rho;
        z2[i] = r;
        c2[i] = cc;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        z[n - 1] = one(m[n - 1]);
        c[n - 1] = z[n - 1] + lam[n - 1] / rho;
    }
}

// positions where sigma changes (setup)
__global__ void k_zb_edges(const double* __restrict__ sigma, long long n, long long* __restrict__ pos, int* __restrict__ counter,
                           int cap) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x + 1; i < n; i += (long long)gridDim.x * blockDim.x)
        if (sigma[i] != sigma[i - 1]) {
            const int slot = atomicAdd(counter, 1);
            if (slot < cap) pos[slot] = i;
        }
}

}  // namespace

size_t zb_hist_bytes() { return sizeof(u32) * ZB_MAX_GROUPS * ZB_BINS; }
int zb_eval_blocks(int64_t n) {   // root passes: two blocks of 256 threads per CU (the accumulators take 216 VGPRs)
    const int64_t b = (n + ZB_THREADS * 8 - 1) / (ZB_THREADS * 8);
    return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}
static int zb_hist_blocks(int64_t n) {
    const int64_t b = (n + ZB_HTHREADS * 8 - 1) / (ZB_HTHREADS * 8);
    return (int)(b < 1 ? 1 : (b > 256 ? 256 : b));
}
// eval partials (512 blocks x 64 values) followed by the gather lists of the clusters
size_t zb_partials_bytes() { return sizeof(double) * (1024 * 4 * ZB_C + ZB_MAX_CLUSTERS * ZB_GCAP); }

int launch_zb_edges(const double* sigma, int64_t n, long long* pos, int* counter, int cap, hipStream_t s) {
    RBL_HIP(hipMemsetAsync(counter, 0, sizeof(int), s));
    if (n > 1) {
        const int64_t b = (n + 255) / 256;
        hipLaunchKernelGGL(k_zb_edges, dim3((unsigned)(b > 4096 ? 4096 : b)), dim3(256), 0, s, sigma, (long long)n, pos, counter, cap);
        RBL_HIP(hipGetLastError());
    }
    return RBL_OK;
}

// the whole banded z-step: keys (of m, any payload) -> z in row order; the status lands in pin[1] with pin[0] = seq
int launch_zband(int loss, const ZbConfig& cfg, int64_t n, double rho, const u64* keys, const double* m, double* z,
                 const double* lam, double* c, ZbState* st, u32* hist, double* partials, int* pin, int seq, u32* counters, hipStream_t s) {
    hipLaunchKernelGGL(k_zb_init, dim3(1), dim3(1024), 0, s, st, cfg, hist);
    const int hb = zb_eval_blocks(n), sb = zb_hist_blocks(n);
    double* glist = partials + 1024 * 4 * ZB_C;
    for (int pass = 0; pass < 6; ++pass) {
        hipLaunchKernelGGL(k_zb_hist, dim3(sb), dim3(ZB_HTHREADS), 0, s, keys, (long long)n, (const ZbState*)st, hist, pass);
        if (loss == RBL_LOSS_BCE)
            hipLaunchKernelGGL(k_zb_scan<0>, dim3(1), dim3(1024), 0, s, st, cfg, hist, pass, rho, 1);
        else
            hipLaunchKernelGGL(k_zb_scan<1>, dim3(1), dim3(1024), 0, s, st, cfg, hist, pass, rho, 1);
    }
    for (int k = 0; k < cfg.nclusters; ++k) {
        if (!cfg.cl_root[k]) continue;
        for (int r = 0; r < ZB_ROOT_PASSES; ++r) {
            const int last = r == ZB_ROOT_PASSES - 1;
            if (loss == RBL_LOSS_BCE) {
                hipLaunchKernelGGL(k_zb_eval<0>, dim3(hb), dim3(ZB_THREADS), 0, s, keys, (long long)n, (const ZbState*)st, cfg, k,
                                   rho, partials);
                hipLaunchKernelGGL(k_zb_refine<0>, dim3(1), dim3(1024), 0, s, st, cfg, k, rho, (const double*)partials, hb, last,
                                   (double*)nullptr, (const double*)nullptr);
            } else {
                hipLaunchKernelGGL(k_zb_eval<1>, dim3(hb), dim3(ZB_THREADS), 0, s, keys, (long long)n, (const ZbState*)st, cfg, k,
                                   rho, partials);
                hipLaunchKernelGGL(k_zb_refine<1>, dim3(1), dim3(1024), 0, s, st, cfg, k, rho, (const double*)partials, hb, last,
                                   (double*)nullptr, (const double*)nullptr);
            }
        }
        if (loss == RBL_LOSS_BCE) {
            hipLaunchKernelGGL(k_zb_gather<0>, dim3(hb), dim3(ZB_THREADS), 0, s, keys, (long long)n, st, cfg, k, rho, glist);
            hipLaunchKernelGGL(k_zb_finish<0>, dim3(1), dim3(1024), 0, s, st, cfg, k, rho, (const double*)glist);
        } else {
            hipLaunchKernelGGL(k_zb_gather<1>, dim3(hb), dim3(ZB_THREADS), 0, s, keys, (long long)n, st, cfg, k, rho, glist);
            hipLaunchKernelGGL(k_zb_finish<1>, dim3(1), dim3(1024), 0, s, st, cfg, k, rho, (const double*)glist);
        }
    }
    const int64_t ab = (n + ZB_THREADS * 8 - 1) / (ZB_THREADS * 8);
    const unsigned ag = (unsigned)(ab < 1 ? 1 : (ab > 2048 ? 2048 : ab));
    if (loss == RBL_LOSS_BCE)
        hipLaunchKernelGGL(k_zb_apply<0>, dim3(ag), dim3(ZB_THREADS), 0, s, m, (long long)n, (const ZbState*)st, cfg, rho, z, lam, c, pin,
                           seq, counters);
    else
        hipLaunchKernelGGL(k_zb_apply<1>, dim3(ag), dim3(ZB_THREADS), 0, s, m, (long long)n, (const ZbState*)st, cfg, rho, z, lam, c, pin,
                           seq, counters);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

// sum_k sigma_k loss(v_(k)) for banded sigma from the keys of v -> out_dev[0] (no sort; always exact)
int launch_zband_risk(int loss, const ZbConfig& cfg, int64_t n, const u64* keys, ZbState* st, u32* hist, double* partials,
                      double* out_dev, hipStream_t s) {
    hipLaunchKernelGGL(k_zb_init, dim3(1), dim3(1024), 0, s, st, cfg, hist);
    const int hb = zb_eval_blocks(n), sb = zb_hist_blocks(n);
    for (int pass = 0; pass < 6; ++pass) {
        hipLaunchKernelGGL(k_zb_hist, dim3(sb), dim3(ZB_HTHREADS), 0, s, keys, (long long)n, (const ZbState*)st, hist, pass);
        hipLaunchKernelGGL(k_zb_scan<0>, dim3(1), dim3(1024), 0, s, st, cfg, hist, pass, 1.0, 0);
    }
    if (loss == RBL_LOSS_BCE) {
        hipLaunchKernelGGL(k_zb_risk<0>, dim3(hb), dim3(ZB_THREADS), 0, s, keys, (long long)n, (const ZbState*)st, cfg, partials);
        hipLaunchKernelGGL(k_zb_risk_finish<0>, dim3(1), dim3(256), 0, s, (const ZbState*)st, cfg, (const double*)partials, hb, out_dev);
    } else {
        hipLaunchKernelGGL(k_zb_risk<1>, dim3(hb), dim3(ZB_THREADS), 0, s, keys, (long long)n, (const ZbState*)st, cfg, partials);
        hipLaunchKernelGGL(k_zb_risk_finish<1>, dim3(1), dim3(256), 0, s, (const ZbState*)st, cfg, (const double*)partials, hb, out_dev);
    }
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

// ---- the same steps one by one, for the multi-GPU driver (collectives in between; include/rbl.h: rbl_zbd_*)
int launch_zbd_init(const ZbConfig& cfg, ZbState* st, u32* hist, hipStream_t s) {
    hipLaunchKernelGGL(k_zb_init, dim3(1), dim3(1024), 0, s, st, cfg, hist);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
int launch_zbd_hist(int64_t n, const u64* keys, ZbState* st, u32* hist, int pass, hipStream_t s) {
    hipLaunchKernelGGL(k_zb_hist, dim3(zb_hist_blocks(n)), dim3(ZB_HTHREADS), 0, s, keys, (long long)n, (const ZbState*)st, hist, pass);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
int launch_zbd_scan(int loss, const ZbConfig& cfg, ZbState* st, u32* hist, int pass, double rho, hipStream_t s) {
    if (loss == RBL_LOSS_BCE)
        hipLaunchKernelGGL(k_zb_scan<0>, dim3(1), dim3(1024), 0, s, st, cfg, hist, pass, rho, 1);
    else
        hipLaunchKernelGGL(k_zb_scan<1>, dim3(1), dim3(1024), 0, s, st, cfg, hist, pass, rho, 1);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
// local block sums of one root pass -> tot (4 * ZB_C doubles, to be summed over the ranks)
int launch_zbd_eval(int loss, const ZbConfig& cfg, int64_t n, const u64* keys, ZbState* st, int k, double rho, double* partials,
                    double* tot, hipStream_t s) {
    const int hb = zb_eval_blocks(n);
    if (loss == RBL_LOSS_BCE) {
        hipLaunchKernelGGL(k_zb_eval<0>, dim3(hb), dim3(ZB_THREADS), 0, s, keys, (long long)n, (const ZbState*)st, cfg, k, rho, partials);
        hipLaunchKernelGGL(k_zb_refine<0>, dim3(1), dim3(1024), 0, s, st, cfg, k, rho, (const double*)partials, hb, 0, tot,
                           (const double*)nullptr);
    } else {
        hipLaunchKernelGGL(k_zb_eval<1>, dim3(hb), dim3(ZB_THREADS), 0, s, keys, (long long)n, (const ZbState*)st, cfg, k, rho, partials);
        hipLaunchKernelGGL(k_zb_refine<1>, dim3(1), dim3(1024), 0, s, st, cfg, k, rho, (const double*)partials, hb, 0, tot,
                           (const double*)nullptr);
    }
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
int launch_zbd_decide(int loss, const ZbConfig& cfg, ZbState* st, int k, double rho, const double* tot, int last, hipStream_t s,
                      int* pin_settled, int dseq) {
    if (loss == RBL_LOSS_BCE)
        hipLaunchKernelGGL(k_zb_refine<0>, dim3(1), dim3(1024), 0, s, st, cfg, k, rho, (const double*)nullptr, 0, last,
                           (double*)nullptr, tot, pin_settled, dseq);
    else
        hipLaunchKernelGGL(k_zb_refine<1>, dim3(1), dim3(1024), 0, s, st, cfg, k, rho, (const double*)nullptr, 0, last,
                           (double*)nullptr, tot, pin_settled, dseq);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
int launch_zbd_gather(int loss, const ZbConfig& cfg, int64_t n, const u64* keys, ZbState* st, int k, double rho, double* partials,
                      double* pack, hipStream_t s) {
    double* glist = partials + 1024 * 4 * ZB_C;
    const int hb = zb_eval_blocks(n);
    if (loss == RBL_LOSS_BCE)
        hipLaunchKernelGGL(k_zb_gather<0>, dim3(hb), dim3(ZB_THREADS), 0, s, keys, (long long)n, st, cfg, k, rho, glist);
    else
        hipLaunchKernelGGL(k_zb_gather<1>, dim3(hb), dim3(ZB_THREADS), 0, s, keys, (long long)n, st, cfg, k, rho, glist);
    hipLaunchKernelGGL(k_zb_pack, dim3(8), dim3(256), 0, s, (const ZbState*)st, k, (const double*)glist, pack);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
int launch_zbd_finish(int loss, const ZbConfig& cfg, ZbState* st, int k, double rho, double* partials, const double* packs_all,
                      int world, hipStream_t s) {
    double* glist = partials + 1024 * 4 * ZB_C;
    hipLaunchKernelGGL(k_zb_union, dim3(1), dim3(1024), 0, s, st, k, packs_all, world, glist);
    if (loss == RBL_LOSS_BCE)
        hipLaunchKernelGGL(k_zb_finish<0>, dim3(1), dim3(1024), 0, s, st, cfg, k, rho, (const double*)glist);
    else
        hipLaunchKernelGGL(k_zb_finish<1>, dim3(1), dim3(1024), 0, s, st, cfg, k, rho, (const double*)glist);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
int launch_zbd_apply(int loss, const ZbConfig& cfg, int64_t n, double rho, const double* m, double* z, const double* lam, double* c,
                     ZbState* st, int* pin, int seq, u32* counters, hipStream_t s) {
    const int64_t ab = (n + ZB_THREADS * 8 - 1) / (ZB_THREADS * 8);
    const unsigned ag = (unsigned)(ab < 1 ? 1 : (ab > 2048 ? 2048 : ab));
    if (loss == RBL_LOSS_BCE)
        hipLaunchKernelGGL(k_zb_apply<0>, dim3(ag), dim3(ZB_THREADS), 0, s, m, (long long)n, (const ZbState*)st, cfg, rho, z, lam, c, pin,
                           seq, counters);
    else
        hipLaunchKernelGGL(k_zb_apply<1>, dim3(ag), dim3(ZB_THREADS), 0, s, m, (long long)n, (const ZbState*)st, cfg, rho, z, lam, c, pin,
                           seq, counters);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
