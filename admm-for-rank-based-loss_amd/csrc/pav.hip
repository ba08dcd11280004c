// pav.hip - exact generalised pool-adjacent-violators on the GPU (z-step,
// src/optim/algorithms.py:95-104 -> src/util/pav.py:93-178 / src/util/PAV_cpt.py:234-293).
//
// The reference merges decreasing runs sweep after sweep (O(n) sweeps in bad cases,
// SURVEY 3.4-i).  The isotonic solution is unique, so this file computes it with a
// bottom-up merge tree instead (restated on the CPU in oracle/pav.py:pav_tree_exact):
//   * u[i] starts as the element prox of sorted position i (every position is a solved
//     segment of length 1);
//   * level L joins the solved segments [k*2^L, k*2^L + 2^(L-1)) and the next 2^(L-1)
//     positions.  Joining two solved neighbours pools exactly ONE block around the seam:
//     the left positions with u > x* and the right positions with u < x*, where x* is the
//     root of the increasing, continuous function
//         Psi(t) = sum_{i in A(t)} f_i'(t),   A(t) = {left: u_i > t} U {right: u_i < t},
//         f_i'(t) = sigma_i * loss'(t) + rho * (t - m_i).
//     The extents are found by binary search on the sign of Psi at existing values of u
//     (each evaluation: two binary searches on the monotone halves + prefix-sum lookups),
//     the pooled block is solved once, and u is overwritten on the pooled range.
// Block sums come from two-level prefix sums (1024-position chunks, chunk totals kept in
// double-double) so a small block's sum is not a difference of two n-sized prefixes.
#include "rbl_internal.h"
#include "device_math.h"
#include "grid_sync.h"

namespace {

constexpr int PV_THREADS = 256;

inline unsigned pv_grid(long long n, int threads = PV_THREADS, long long cap = 1 << 20) {
    long long g = (n + threads - 1) / threads;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// ---------------------------------------------------------------- prefix sums
// locx[i] = sum of x over [chunk_start(i), i); chunk_tot[c] = sum of chunk c.  One block
// per 1024-chunk, 256 threads x 4 consecutive elements.  Positions 0..n inclusive.
// KEYS: x is the sorted key array of the radix sort; the values are recovered on the fly (and stored
// to ms_out: the sorted m) instead of by a streaming pass of their own
template <bool KEYS>
__global__ __launch_bounds__(256) void k_chunk_scan(const double* __restrict__ x, long long n,
                                                     double* __restrict__ locx, double* __restrict__ chunk_tot,
                                                     double* __restrict__ ms_out) {
    __shared__ double wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long base = (long long)blockIdx.x * PAV_CHUNK + tid * 4;
    double a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = 0.0;
        if (base + k < n) {
            if (KEYS) {
                a[k] = rbl::unflip_key(reinterpret_cast<const u64*>(x)[base + k]);
                ms_out[base + k] = a[k];
            } else {
                a[k] = x[base + k];
            }
        }
    }
    double tsum = (a[0] + a[1]) + (a[2] + a[3]);
    double incl = tsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        double y = __shfl_up(incl, off, 64);
        if (lane >= off) incl += y;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    double pre = incl - tsum;
    for (int w = 0; w < wave; ++w) pre += wsum[w];
    double run = pre;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k <= n) locx[base + k] = run;
        run += a[k];
    }
    if (tid == 255) chunk_tot[blockIdx.x] = run;
}

__device__ inline void two_sum(double a, double b, double& s, double& e) {
    s = a + b;
    double bb = s - a;
    e = (a - (s - bb)) + (b - bb);
}
__device__ inline void dd_add(double ah, double al, double bh, double bl, double& ch, double& cl) {
    double s, e;
    two_sum(ah, bh, s, e);
    e += al + bl;
    ch = s + e;
    cl = e - (ch - s);
}

// double-double exclusive scan of the chunk totals, one block of 1024 threads
__global__ __launch_bounds__(1024) void k_chunk_prefix_dd(const double* __restrict__ chunk_tot, long long nc,
                                                           double* __restrict__ cph, double* __restrict__ cpl) {
    __shared__ double sh[1024], sl[1024];
    const int tid = threadIdx.x;
    const long long per = (nc + 1023) / 1024;
    const long long b = tid * per;
    long long e = b + per;
    if (e > nc) e = nc;
    double h = 0.0, l = 0.0;
    for (long long i = b; i < e; ++i) dd_add(h, l, chunk_tot[i], 0.0, h, l);
    sh[tid] = h;
    sl[tid] = l;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        double oh = 0.0, ol = 0.0;
        if (tid >= off) {
            oh = sh[tid - off];
            ol = sl[tid - off];
        }
        __syncthreads();
        if (tid >= off) {
            dd_add(sh[tid], sl[tid], oh, ol, h, l);
            sh[tid] = h;
            sl[tid] = l;
        }
        __syncthreads();
    }
    // exclusive prefix of this thread's first chunk
    double ph = 0.0, pl = 0.0;
    if (tid > 0) {
        ph = sh[tid - 1];
        pl = sl[tid - 1];
    }
    for (long long i = b; i < e; ++i) {
        cph[i] = ph;
        cpl[i] = pl;
        dd_add(ph, pl, chunk_tot[i], 0.0, ph, pl);
    }
}

__device__ inline double range_sum(const Prefix& p, long long s, long long e1) {
    // sum over sorted positions [s, e1)
    const long long cs = s >> PAV_CHUNK_LOG, ce = e1 >> PAV_CHUNK_LOG;
    double r = p.locx[e1] - p.locx[s];
    if (ce != cs) r += (p.cph[ce] - p.cph[cs]) + (p.cpl[ce] - p.cpl[cs]);
    return r;
}

// ---------------------------------------------------------------- block value / sign
template <int LOSS>
__device__ inline double block_value(double ssig, double sm, double cnt, double rho) {
    // src/util/pav.py:134-140: solve with (sum sigma / len, sum m / len).  The safeguarded Newton starts from the
    // expansion around the mean of m (device_math.h: prox_bce_est - 2-3 exponentials per block) instead of cold from
    // x = m (6-8, with IEEE divisions, until the iterate repeats): same root to rounding.  With rank weights that pool
    // everywhere (EHRM's CPT weights) these solves, not level 0, are what k_pav_bottom / k_pav_upper spend their time on.
    return rbl::prox_est<LOSS>(ssig / cnt, rho, sm / cnt);
}

// ---------------------------------------------------------------- one seam merge
// Accessors: where u and the prefix sums of sigma / m live.  Global memory for the upper
// levels, LDS (tile-local indices and plain tile-local prefixes) for the bottom levels.
struct GlobalAcc {
    const double* u;
    Prefix pa, pm;
    __device__ inline double val(long long i) const { return u[i]; }
    __device__ inline double sum_a(long long s, long long e1) const { return range_sum(pa, s, e1); }
    __device__ inline double sum_m(long long s, long long e1) const { return range_sum(pm, s, e1); }
};
struct LdsAcc {
    const double* u;   // tile values
    const double* pa;  // exclusive prefix of sigma inside the tile (C+1 entries)
    const double* pm;  // exclusive prefix of m inside the tile
    __device__ inline double val(long long i) const { return u[i]; }
    __device__ inline double sum_a(long long s, long long e1) const { return pa[e1] - pa[s]; }
    __device__ inline double sum_m(long long s, long long e1) const { return pm[e1] - pm[s]; }
};

// sign of the pooled derivative of positions [s, e] at t:  >0 <=> pooled value < t
template <int LOSS, typename Acc>
__device__ inline double psi_sign(const Acc& ac, double rho, long long s, long long e, double t) {
    if (e < s) return 0.0;
    const double A = ac.sum_a(s, e + 1), M = ac.sum_m(s, e + 1), cnt = (double)(e + 1 - s);
    if (LOSS == 0) return A * rbl::sigmoid1(t) + rho * (cnt * t - M);
    return t - block_value<1>(A, M, cnt, rho);
}

template <typename Acc>
__device__ inline long long upper_bound_gt(const Acc& ac, long long lo, long long hi, double t) {
    // first i in [lo, hi) with u[i] > t (hi if none); u non-decreasing on [lo, hi)
    while (lo < hi) {
        long long mid = lo + ((hi - lo) >> 1);
        if (ac.val(mid) > t) hi = mid; else lo = mid + 1;
    }
    return lo;
}
template <typename Acc>
__device__ inline long long lower_bound_ge(const Acc& ac, long long lo, long long hi, double t) {
    // first j in [lo, hi) with u[j] >= t (hi if none)
    while (lo < hi) {
        long long mid = lo + ((hi - lo) >> 1);
        if (ac.val(mid) >= t) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// Wave-cooperative accessor for the upper levels: a whole wave runs seam_merge() on ONE seam with
// identical arguments in all 64 lanes; the inner binary searches become 64-ary (each step samples
// 64 positions with one load per lane and a ballot), so a search over 3M positions takes 4
// dependent loads instead of 22.  Reads of single values / prefix sums are wave-uniform loads.
struct WaveAcc {
    const double* u;
    Prefix pa, pm;
    __device__ inline double val(long long i) const { return u[i]; }
    __device__ inline double sum_a(long long s, long long e1) const { return range_sum(pa, s, e1); }
    __device__ inline double sum_m(long long s, long long e1) const { return range_sum(pm, s, e1); }
};
template <bool STRICT>   // STRICT: first i with u[i] > t;  else first i with u[i] >= t
__device__ inline long long wave_first(const double* __restrict__ u, long long lo, long long hi, double t) {
    const int lane = threadIdx.x & 63;
    // invariant: the answer is the first position in [lo, hi) that satisfies the predicate, or hi
    while (hi - lo > 64) {
        const long long step = (hi - lo + 63) >> 6;
        const long long p = lo + (long long)lane * step + (step - 1);   // last position of chunk `lane`
        bool pred = true;                                               // chunks past hi count as satisfied
        if (p < hi) {
            const double x = u[p];
            pred = STRICT ? (x > t) : (x >= t);
        }
        const unsigned long long mask = __ballot(pred);
        if (mask == 0ull) return hi;
        const int j = __ffsll((long long)mask) - 1;   // first chunk whose last position satisfies it
        const long long nlo = lo + (long long)j * step;
        const long long pj = nlo + step - 1;
        lo = nlo;
        if (pj < hi) hi = pj;   // u[pj] satisfies the predicate: it is the answer unless an earlier one does
    }
    const long long p = lo + lane;
    bool pred = true;
    if (p < hi) {
        const double x = u[p];
        pred = STRICT ? (x > t) : (x >= t);
    }
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return hi;   // exactly 64 positions left and none satisfies the predicate
    const long long r = lo + (__ffsll((long long)mask) - 1);
    return r < hi ? r : hi;
}
__device__ inline long long upper_bound_gt(const WaveAcc& ac, long long lo, long long hi, double t) {
    return lo < hi ? wave_first<true>(ac.u, lo, hi, t) : lo;
}
__device__ inline long long lower_bound_ge(const WaveAcc& ac, long long lo, long long hi, double t) {
    return lo < hi ? wave_first<false>(ac.u, lo, hi, t) : lo;
}

// the same for a tile in LDS (top levels of k_pav_bottom: 4, 2, 1 seams per tile, one wave each)
struct WaveLdsAcc {
    const double* u;
    const double* pa;
    const double* pm;
    __device__ inline double val(long long i) const { return u[i]; }
    __device__ inline double sum_a(long long s, long long e1) const { return pa[e1] - pa[s]; }
    __device__ inline double sum_m(long long s, long long e1) const { return pm[e1] - pm[s]; }
};
__device__ inline long long upper_bound_gt(const WaveLdsAcc& ac, long long lo, long long hi, double t) {
    return lo < hi ? wave_first<true>(ac.u, lo, hi, t) : lo;
}
__device__ inline long long lower_bound_ge(const WaveLdsAcc& ac, long long lo, long long hi, double t) {
    return lo < hi ? wave_first<false>(ac.u, lo, hi, t) : lo;
}

// Joins the solved segments [L0, seam) and [seam, R1) (seam violates: u[seam-1] > u[seam]).
// Returns the pooled range [s*, e*] and its value.
// hint_s / hint_e (optional, -1 = none): the extents this seam pooled in the previous ADMM iteration.  The block
// structure moves little from one iteration to the next, so the searches start there (a few probes around the
// old boundary) instead of galloping away from the seam; any value is safe - a hint only picks the first probe.
template <int LOSS, typename Acc>
__device__ inline void seam_merge(const Acc& ac, long long L0, long long seam, long long R1, double rho,
                                  long long& s_star, long long& e_star, double& x, long long hint_s = -1,
                                  long long hint_e = -1, double hint_x = __builtin_nan("")) {
    // Both boundary functions are monotone in t:  s(t) = first left position with u > t,
    // e(t) = last right position with u < t.  Every evaluation of Psi therefore narrows the
    // ranges in which later evaluations have to search ([s_lb,s_ub], [e_lb,e_ub]).
    long long s_lb = L0, s_ub = seam, e_lb = seam - 1, e_ub = R1 - 1;
    auto psi_at = [&](double t, long long s_from, long long e_to) -> double {
        const long long a = s_from > s_lb ? s_from : s_lb;
        const long long s = upper_bound_gt(ac, a < s_ub ? a : s_ub, s_ub, t);
        const long long eb = (e_to < e_ub ? e_to : e_ub) + 1;
        const long long e = lower_bound_ge(ac, e_lb + 1 < eb ? e_lb + 1 : eb, eb, t) - 1;
        const double sg = psi_sign<LOSS>(ac, rho, s, e, t);
        if (sg > 0.0) {  // x* < t: later probes use smaller t
            s_ub = s;
            e_ub = e;
        } else if (sg < 0.0) {  // x* > t: later probes use larger t
            s_lb = s;
            e_lb = e;
        }
        return sg;
    };
    // Fast path with a hint for the pooled VALUE (last iteration's): x* is the fixed point of
    // t -> T(t) = block value of A(t), and sign(T(t) - t) = -sign Psi(t).  From a t close to x* the set A(t) is
    // (almost) the final one, T(t) lands on x* to rounding, and the iteration stops as soon as A(T(t)) == A(t):
    // typically 2-3 evaluations (two searches + prefix look-ups each) where the positional search below needs
    // 10-20.  Bracketed in value space (Psi is increasing); anything unexpected falls through to that search.
    if (hint_x == hint_x) {
        double t_lo = ac.val(seam), t_hi = ac.val(seam - 1);      // Psi(t_lo) < 0 < Psi(t_hi): the seam violates
        double t = hint_x;
        if (!(t > t_lo && t < t_hi)) t = 0.5 * (t_lo + t_hi);
        if (t > t_lo && t < t_hi) {
            auto sets_at = [&](double tt, long long& ss, long long& ee, double& xx) -> double {
                ss = upper_bound_gt(ac, s_lb < s_ub ? s_lb : s_ub, s_ub, tt);
                const long long eb = e_ub + 1;
                ee = lower_bound_ge(ac, e_lb + 1 < eb ? e_lb + 1 : eb, eb, tt) - 1;
                const double A = ac.sum_a(ss, ee + 1), M = ac.sum_m(ss, ee + 1), cnt = (double)(ee + 1 - ss);
                xx = block_value<LOSS>(A, M, cnt, rho);
                const double sg = (LOSS == 0) ? A * rbl::sigmoid1(tt) + rho * (cnt * tt - M) : tt - xx;
                if (sg > 0.0) {
                    s_ub = ss;
                    e_ub = ee;
                } else if (sg < 0.0) {
                    s_lb = ss;
                    e_lb = ee;
                }
                return sg;
            };
            long long s1, e1, s2, e2;
            double x1, x2;
            double sg = sets_at(t, s1, e1, x1);
            for (int it = 0; it < 10; ++it) {
                if (sg > 0.0) t_hi = t; else if (sg < 0.0) t_lo = t;
                double tn = x1;
                const bool newton = tn > t_lo && tn < t_hi;
                if (!newton) tn = 0.5 * (t_lo + t_hi);
                if (!(tn > t_lo && tn < t_hi)) break;          // the bracket cannot be split any further
                const double sg2 = sets_at(tn, s2, e2, x2);
                if (newton && s2 == s1 && e2 == e1) {          // A(T(t)) == A(t): consistent, x1 is the root
                    s_star = s1;
                    e_star = e1;
                    x = x1;
                    return;
                }
                t = tn;
                s1 = s2;
                e1 = e2;
                x1 = x2;
                sg = sg2;
            }
            // not settled (rare): the positional search below runs exactly as without a value hint
            s_lb = L0;
            s_ub = seam;
            e_lb = seam - 1;
            e_ub = R1 - 1;
        }
    }
    // s* = first left position whose value exceeds x*  <=>  first i with Psi(u[i]) > 0.
    // Psi(u[seam-1]) > 0 is known (the seam violates); gallop leftwards from the seam (x8 per
    // step), then close the bracket by interpolating on the VALUES of Psi (monotone and smooth
    // along the positions), falling back to bisection whenever a step fails to halve the
    // bracket.  A cascade over k positions costs ~log8(k) + a handful of evaluations.
    long long hi = seam - 1, lo = L0 - 1;  // pred(hi) true, pred(lo) false (L0-1: virtual)
    double f_hi = 1.0, f_lo = 0.0;
    bool have_lo = false, have_hi = false;
    long long from = seam - 1;             // the gallop leaves from here (leftwards)
    bool gallop_left = true;
    if (hint_s >= L0 && hint_s < seam - 1) {
        const double f = psi_at(ac.val(hint_s), hint_s + 1, R1 - 1);
        if (f > 0.0) {                     // still pooled: s* <= hint, continue leftwards from it
            hi = hint_s;
            f_hi = f;
            have_hi = true;
            from = hint_s;
        } else {                           // s* moved towards the seam: gallop rightwards from the hint
            lo = hint_s;
            f_lo = f;
            have_lo = true;
            gallop_left = false;
            for (long long off = 1;; off <<= 3) {
                const long long p = hint_s + off;
                if (p >= hi) break;
                const double g = psi_at(ac.val(p), p + 1, R1 - 1);
                if (g > 0.0) {
                    hi = p;
                    f_hi = g;
                    have_hi = true;
                    break;
                }
                lo = p;
                f_lo = g;
            }
        }
    }
    if (gallop_left)
    for (long long off = 1; hi > L0; off <<= 3) {
        long long p = from - off;
        if (p < L0) p = L0;
        const double f = psi_at(ac.val(p), p + 1, R1 - 1);
        if (f > 0.0) {
            hi = p;
            f_hi = f;
            have_hi = true;
        } else {
            lo = p;
            f_lo = f;
            have_lo = true;
            break;
        }
    }
    bool bisect = false;
    while (hi - lo > 1) {
        const long long span = hi - lo;
        long long mid = lo + (span >> 1);
        if (!bisect && have_lo && have_hi) {
            const double frac = -f_lo / (f_hi - f_lo);
            long long st = (long long)(frac * (double)span);
            if (st < 1) st = 1;
            if (st > span - 1) st = span - 1;
            mid = lo + st;
        }
        const double f = psi_at(ac.val(mid), mid + 1, R1 - 1);
        if (f > 0.0) {
            hi = mid;
            f_hi = f;
            have_hi = true;
        } else {
            lo = mid;
            f_lo = f;
            have_lo = true;
        }
        bisect = (hi - lo) * 2 > span;  // interpolation did not halve the bracket: bisect next
    }
    s_star = hi;
    // e* = last right position whose value is below x*  <=>  last j with Psi(u[j]) < 0.
    // Psi(u[seam]) < 0 is known; gallop rightwards.  The upper bounds found so far stay valid
    // (a probe above them still sees exactly the pooled sets of the bounding evaluation, whose
    // sign it inherits); the lower bounds do not (the first probes lie below them): reset.
    s_lb = L0;
    e_lb = seam - 1;
    lo = seam;
    hi = R1;  // pred(lo) true, pred(hi) false (R1: virtual)
    have_lo = have_hi = false;
    from = seam;
    bool gallop_right = true;
    if (hint_e > seam && hint_e <= R1 - 1) {
        const double f = psi_at(ac.val(hint_e), L0, hint_e - 1);
        if (f < 0.0) {                     // still pooled: e* >= hint, continue rightwards from it
            lo = hint_e;
            f_lo = f;
            have_lo = true;
            from = hint_e;
        } else {                           // e* moved towards the seam: gallop leftwards from the hint
            hi = hint_e;
            f_hi = f;
            have_hi = true;
            gallop_right = false;
            for (long long off = 1;; off <<= 3) {
                const long long p = hint_e - off;
                if (p <= lo) break;
                const double g = psi_at(ac.val(p), L0, p - 1);
                if (g < 0.0) {
                    lo = p;
                    f_lo = g;
                    have_lo = true;
                    break;
                }
                hi = p;
                f_hi = g;
            }
        }
    }
    if (gallop_right)
    for (long long off = 1; lo < R1 - 1; off <<= 3) {
        long long p = from + off;
        if (p > R1 - 1) p = R1 - 1;
        const double f = psi_at(ac.val(p), L0, p - 1);
        if (f < 0.0) {
            lo = p;
            f_lo = f;
            have_lo = true;
        } else {
            hi = p;
            f_hi = f;
            have_hi = true;
            break;
        }
    }
    bisect = false;
    while (hi - lo > 1) {
        const long long span = hi - lo;
        long long mid = lo + (span >> 1);
        if (!bisect && have_lo && have_hi) {
            const double frac = -f_lo / (f_hi - f_lo);
            long long st = (long long)(frac * (double)span);
            if (st < 1) st = 1;
            if (st > span - 1) st = span - 1;
            mid = lo + st;
        }
        const double f = psi_at(ac.val(mid), L0, mid - 1);
        if (f < 0.0) {
            lo = mid;
            f_lo = f;
            have_lo = true;
        } else {
            hi = mid;
            f_hi = f;
            have_hi = true;
        }
        bisect = (hi - lo) * 2 > span;
    }
    e_star = lo;
    const double A = ac.sum_a(s_star, e_star + 1), M = ac.sum_m(s_star, e_star + 1);
    x = block_value<LOSS>(A, M, (double)(e_star + 1 - s_star), rho);
}

// ---------------------------------------------------------------- bottom of the tree, in LDS
// One workgroup per tile of PB_TILE sorted positions: element prox (the tree's level 0),
// tile-local prefix sums and the levels with segments up to PB_TILE, all in LDS; one global
// read of (m, sigma) and one write of u per position.
//
// Levels 1-3 (segments of PB_PER = 8 positions) are ONE step (round 3): every thread runs the classic sequential
// stack PAV over its own 8 consecutive positions - block sums from the tile's prefix arrays, the stack a bit mask of
// block starts - which leaves exactly what three tree levels leave (the isotonic solution of a segment is unique;
// the block values are the same block_value() of the same prefix differences).  With smooth rank weights (extremile,
// esrm, EHRM's CPT weights) the solution has thousands of short blocks (pairs, triples) spread over the whole upper
// part of the order: as tree merges each of them cost one divergent seam_merge() - searches included - per wave and
// level (k_pav_bottom 381 us at 6.25 M EHRM positions against 141 us for a superquantile problem of the same size);
// sequentially a pair is one comparison and one block solve.  (A second sequential stage over 64-position segments - one
// thread per segment walking the blocks of stage 1 through a 64-bit mask - was built and measured: 32 active lanes per
// tile cost more than the three tree levels they replace, z-step 0.87 -> 0.93 ms at 6.25 M EHRM positions; removed.)
//
// EHRM with a SPECULATED branch (SPEC): the singleton-stage scalar test of PAV_cpt.py:205-226 needs
// f1 = sum phi_a(min(prox_a(m), B)) and f2 = sum phi_b(max(prox_b(m), B)) over all positions.  Round 2 computed both
// element prox vectors in a pass of its own (k_ehrm_fvals, 182 us at 6.25 M), stored both, and the tree read the one
// the test chose.  The branch is the same from one ADMM iteration to the next (b on every trajectory seen, SURVEY
// 3.4-b), so the tree is built for the branch the PREVIOUS iteration took while this kernel accumulates f1 and f2 on
// the side: the speculated branch's prox is level 0 anyway, and the other branch's clipped value is B itself - no
// Newton solve - wherever its prox lies beyond B, i.e. m >= B + sigma_a sigmoid(B) / rho for branch a (phi_a'(B) <= 0),
// m <= B + sigma_b sigmoid(B) / rho for branch b.  k_ehrm_pick then forms the exact test; if it contradicts the
// speculation, the plain kernel (guard = the speculated branch: a no-op launch otherwise) rebuilds the tile for the
// other branch before the upper levels run.
constexpr int PB_TILE_LOG = 11;
constexpr int PB_TILE = 1 << PB_TILE_LOG;  // 2048 positions, 48 KB of LDS
constexpr int PB_PER = PB_TILE / PV_THREADS;
static_assert(PB_PER == 8, "the sequential step keeps its block starts in the low 8 bits of a mask");

// tools/pav_lab.hip builds this file with PAV_LAB defined: block 0's wall-clock stamps at the stages of k_pav_bottom
#ifdef PAV_LAB
#define PAV_LAB_STAMP(i)                                                             \
    do {                                                                             \
        if (threadIdx.x == 0 && blockIdx.x < 8) pav_lab_stamps[blockIdx.x * 32 + (i)] = (long long)wall_clock64(); \
    } while (0)
#else
#define PAV_LAB_STAMP(i) do { } while (0)
#endif
template <int LOSS, bool SPEC>
__global__ __launch_bounds__(PV_THREADS) void k_pav_bottom(const double* __restrict__ ms, const double* __restrict__ sa,
                                                            const double* __restrict__ sb, const int* __restrict__ branch,
                                                            double rho, long long n, double* u_out,
                                                            u32* __restrict__ merge_counter, const double* u0a,
                                                            const double* u0b, int bflags, int skip_if_branch,
                                                            double B, int spec, double* __restrict__ fpart) {
    // u0a / u0b != NULL (EHRM, distributed z-step): level 0 was computed by k_ehrm_fvals; u0a may alias u_out (a block
    // reads and writes only its own tile).  skip_if_branch >= 0: nothing to do when *branch says so (see above).
    // SPEC: sa / sb / B / spec as described above, fpart[2 * block + {0, 1}] receive this tile's share of f1 / f2.
    if (!SPEC && skip_if_branch >= 0 && branch && *branch == skip_if_branch) return;
    PAV_LAB_STAMP(0);
    const int wave_top = bflags & 1;
    const bool seq_levels = !(bflags & 2);
    __shared__ double su[PB_TILE];
    __shared__ double spa[PB_TILE + 1];
    __shared__ double spm[PB_TILE + 1];
    __shared__ double wsum[2][PV_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Tiles are taken from the HIGH end of the order first: with rank weights that grow towards the top (EHRM's CPT weights,
    // extremile, esrm) the pooling - chains of dependent block solves - concentrates in the last tiles, and a tile that
    // starts last is the kernel's tail (EHRM, 6.25M positions: the average wave lives 13 us, the kernel took 296 us).
    const long long tile = (long long)gridDim.x - 1 - blockIdx.x;
    const long long base = tile * PB_TILE;
    const bool use_b = SPEC ? (spec != 0) : (branch && *branch);
    const double* sg = use_b ? sb : sa;
    const double* u0 = (!SPEC && u0a) ? (use_b ? u0b : u0a) : nullptr;
    long long nt = n - base;  // valid positions of this tile
    if (nt > PB_TILE) nt = PB_TILE;

    // level 0 + thread-local sums (PB_PER consecutive positions per thread).  ONE copy of the element code (the loop is not
    // unrolled: eight inlined copies of the two safeguarded Newton solves made the speculating kernel 19 000 instructions
    // long); the position's sigma and m are parked in the prefix arrays until the prefixes replace them below.
    double ts = 0.0, tm = 0.0;
    double f12[2] = {0.0, 0.0};
    const double sigB = SPEC ? rbl::sigmoid1(B) : 0.0, spB = SPEC ? rbl::softplus(B) : 0.0;
#pragma clang loop unroll(disable)
    for (int k = 0; k < PB_PER; ++k) {
        const int i = tid * PB_PER + k;
        const bool ok = i < nt;
        const double ls_k = ok ? sg[base + i] : 0.0;
        const double lm_k = ok ? ms[base + i] : 0.0;
        spa[i] = ls_k;
        spm[i] = lm_k;
        if (SPEC) {
            double x = 0.0;
            if (ok) {
                // one exponential serves sigmoid(m) and softplus(m); the two objective values are expanded around m
                // (device_math.h: softplus_near), the two prox problems start from the O(a^4) estimate
                const double m = lm_k, em = exp(-fabs(m)), inv = 1.0 / (1.0 + em);
                const double sm = (m > 0.0) ? inv : em * inv, spm = fmax(m, 0.0) + log1p(em);
                x = rbl::prox_bce_est(ls_k, rho, m, sm);                    // level 0 of the speculated branch
                const double so = (use_b ? sa : sb)[base + i];               // the weight of the other branch
                // speculated branch: clipped at B from its own side; other branch: B itself wherever its prox lies beyond B
                double os = x, oo;
                if (use_b) {
                    if (os <= B) os = B;                                     // PAV_cpt.py:218
                    oo = (m >= B + so * sigB / rho) ? B : fmin(rbl::prox_bce_est(so, rho, m, sm), B);   // :211
                } else {
                    if (os > B) os = B;
                    oo = (m <= B + so * sigB / rho) ? B : fmax(rbl::prox_bce_est(so, rho, m, sm), B);
                }
                const double fs = ls_k * (os == B ? spB : rbl::softplus_near(os, m, spm, sm)) + 0.5 * rho * (os - m) * (os - m);
                const double fo = so * (oo == B ? spB : rbl::softplus_near(oo, m, spm, sm)) + 0.5 * rho * (oo - m) * (oo - m);
                f12[0] += use_b ? fo : fs;                                   // f1 belongs to branch a, f2 to branch b
                f12[1] += use_b ? fs : fo;
            }
            su[i] = x;
        } else {
            su[i] = ok ? (u0 ? u0[base + i] : rbl::prox_est<LOSS>(ls_k, rho, lm_k)) : 0.0;
        }
        ts += ls_k;
        tm += lm_k;
    }
    double is = ts, im = tm;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double ys = __shfl_up(is, off, 64), ym = __shfl_up(im, off, 64);
        if (lane >= off) {
            is += ys;
            im += ym;
        }
    }
    if (lane == 63) {
        wsum[0][wave] = is;
        wsum[1][wave] = im;
    }
    __syncthreads();
    double ps = is - ts, pm_ = im - tm;
    for (int w = 0; w < wave; ++w) {
        ps += wsum[0][w];
        pm_ += wsum[1][w];
    }
#pragma unroll
    for (int k = 0; k < PB_PER; ++k) {
        const int i = tid * PB_PER + k;
        const double ls_k = spa[i], lm_k = spm[i];   // parked by this thread above
        spa[i] = ps;
        spm[i] = pm_;
        ps += ls_k;
        pm_ += lm_k;
    }
    if (tid == PV_THREADS - 1) {
        spa[PB_TILE] = ps;
        spm[PB_TILE] = pm_;
    }
    __syncthreads();
    if (SPEC) {
        // this tile's share of the two singleton-stage sums (fixed order: deterministic)
        rbl::block_sum<2, PV_THREADS>(f12, &wsum[0][0]);
        if (tid == 0) {
            fpart[2 * tile + 0] = f12[0];
            fpart[2 * tile + 1] = f12[1];
        }
        __syncthreads();
    }

    PAV_LAB_STAMP(1);
    const LdsAcc ac{su, spa, spm};
    u32 merges = 0;
    // Levels 1-3: sequential PAV over the thread's own PB_PER positions (see the header comment).  `starts`: bit j set
    // = position j of the segment starts a block; all positions of a block hold its value.
    if (seq_levels) {
        const int b0 = tid * PB_PER;
        const int cnt = (int)(nt - b0 < PB_PER ? nt - b0 : PB_PER);
        u32 starts = cnt > 0 ? 1u : 0u;
        if (cnt > 1) {
            double cur = su[b0];
            int cur_s = 0;
            for (int i = 1; i < cnt; ++i) {
                const double xi = su[b0 + i];
                if (cur <= xi) {                     // pav.py:105: only a strict decrease violates
                    starts |= 1u << i;
                    cur = xi;
                    cur_s = i;
                    continue;
                }
                int s0 = cur_s;
                double x = block_value<LOSS>(spa[b0 + i + 1] - spa[b0 + s0], spm[b0 + i + 1] - spm[b0 + s0], (double)(i + 1 - s0), rho);
                ++merges;
                while (s0 > 0) {
                    const int ps0 = 31 - __clz((int)(starts & ((1u << s0) - 1u)));   // start of the block before
                    if (su[b0 + ps0] <= x) break;
                    starts &= ~(1u << s0);
                    s0 = ps0;
                    x = block_value<LOSS>(spa[b0 + i + 1] - spa[b0 + s0], spm[b0 + i + 1] - spm[b0 + s0], (double)(i + 1 - s0), rho);
                    ++merges;
                }
                for (int j = s0; j <= i; ++j) su[b0 + j] = x;
                cur = x;
                cur_s = s0;
            }
        }
        __syncthreads();
    }
    PAV_LAB_STAMP(2);
    // Levels with short segments: the thread that merged a seam writes the pooled range itself.
    // From PB_COOP on (at most PB_TILE / (2 PB_COOP) seams per level) the pooled ranges get long
    // (up to the whole tile) and a single thread writing them would serialise the level: the
    // seam threads only record (s*, e*, x) and the whole workgroup writes.
    constexpr int PB_COOP = 32;
    __shared__ int rec_s[PB_TILE / (2 * PB_COOP)], rec_e[PB_TILE / (2 * PB_COOP)];
    __shared__ double rec_x[PB_TILE / (2 * PB_COOP)];
    for (int half = seq_levels ? PB_PER : 1; half < PB_TILE; half <<= 1) {
        const int nseams = PB_TILE / (2 * half);
        const bool coop = half >= PB_COOP;
        if (wave_top && nseams <= PV_THREADS / 64) {
            // top levels of the tile: one WAVE per seam, 64-ary searches in LDS
            const int k = wave;
            if (k < nseams) {
                const long long seam = (2LL * k + 1) * half;
                bool act = seam < nt;
                if (act) act = su[seam - 1] > su[seam];   // pav.py:105: only a strict decrease violates
                if (lane == 0) rec_s[k] = -1;
                if (act) {
                    long long R1 = seam + half;
                    if (R1 > nt) R1 = nt;
                    long long s_star, e_star;
                    double x;
                    const WaveLdsAcc wac{su, spa, spm};
                    seam_merge<LOSS>(wac, seam - half, seam, R1, rho, s_star, e_star, x);
                    if (lane == 0) {
                        rec_s[k] = (int)s_star;
                        rec_e[k] = (int)e_star;
                        rec_x[k] = x;
                        ++merges;
                    }
                }
            }
        } else
        for (int k = tid; k < nseams; k += PV_THREADS) {
            const long long seam = (2LL * k + 1) * half;
            if (coop) rec_s[k] = -1;
            if (seam >= nt) continue;
            if (su[seam - 1] <= su[seam]) continue;  // pav.py:105: only a strict decrease violates
            long long R1 = seam + half;
            if (R1 > nt) R1 = nt;
            long long s_star, e_star;
            double x;
            seam_merge<LOSS>(ac, seam - half, seam, R1, rho, s_star, e_star, x);
            if (coop) {
                rec_s[k] = (int)s_star;
                rec_e[k] = (int)e_star;
                rec_x[k] = x;
            } else {
                for (long long i = s_star; i <= e_star; ++i) su[i] = x;  // disjoint from other seams' segments
            }
            ++merges;
        }
        __syncthreads();
        if (coop) {
            const int shift = 31 - __clz(2 * half);   // log2 of the segment length
            for (int i = tid; i < nt; i += PV_THREADS) {
                const int k = i >> shift;
                const int s0 = rec_s[k];
                if (s0 >= 0 && i >= s0 && i <= rec_e[k]) su[i] = rec_x[k];
            }
            __syncthreads();
        }
        PAV_LAB_STAMP(3 + (31 - __clz(half)));      // after the level that joins segments of `half` positions
    }
    for (int i = tid; i < nt; i += PV_THREADS) u_out[base + i] = su[i];
    PAV_LAB_STAMP(20);
    if (merges) atomicAdd(merge_counter, merges);
}

// ---------------------------------------------------------------- upper levels, global memory
// One thread per seam of a level.  half = 2^(L-1) >= PB_TILE.
template <int LOSS>
__global__ __launch_bounds__(PV_THREADS) void k_pav_seam(double* __restrict__ u, long long n, long long half,
                                                          Prefix pa_, Prefix pb_, Prefix pm, const int* branch,
                                                          double rho, SeamRec* __restrict__ recs,
                                                          long long nseams, u32* __restrict__ merge_counter) {
    const long long k = (long long)blockIdx.x * PV_THREADS + threadIdx.x;
    if (k >= nseams) return;
    const long long seam = (2 * k + 1) * half;
    if (seam >= n || u[seam - 1] <= u[seam]) {  // pav.py:105 - only a strict decrease is a violation
        recs[k].s = -1;
        return;
    }
    const GlobalAcc ac{u, (branch && *branch) ? pb_ : pa_, pm};
    long long R1 = seam + half;
    if (R1 > n) R1 = n;
    long long s_star, e_star;
    double x;
    seam_merge<LOSS>(ac, seam - half, seam, R1, rho, s_star, e_star, x);
    atomicAdd(merge_counter, 1u);
    recs[k].s = s_star;
    recs[k].e = e_star;
    recs[k].x = x;
}

// One WAVE per seam (the levels where seams are few and long): see WaveAcc.
template <int LOSS>
__global__ __launch_bounds__(PV_THREADS) void k_pav_seam_wave(double* __restrict__ u, long long n, long long half,
                                                               Prefix pa_, Prefix pb_, Prefix pm, const int* branch,
                                                               double rho, SeamRec* __restrict__ recs,
                                                               long long nseams, u32* __restrict__ merge_counter,
                                                               SeamRec* __restrict__ hints) {
    const long long k = ((long long)blockIdx.x * PV_THREADS + threadIdx.x) >> 6;   // wave-uniform
    const int lane = threadIdx.x & 63;
    if (k >= nseams) return;
    const long long seam = (2 * k + 1) * half;
    if (seam >= n || u[seam - 1] <= u[seam]) {  // pav.py:105 - only a strict decrease is a violation
        if (lane == 0) recs[k].s = -1;
        return;
    }
    const WaveAcc ac{u, (branch && *branch) ? pb_ : pa_, pm};
    long long R1 = seam + half;
    if (R1 > n) R1 = n;
    long long s_star, e_star;
    double x;
    long long hs = -1, he = -1;
    double hx = __builtin_nan("");
    if (hints) {   // what this seam pooled in the previous iteration (wave-uniform loads)
        hs = hints[k].s;
        he = hints[k].e;
        if (hs >= 0) hx = hints[k].x;
    }
    seam_merge<LOSS>(ac, seam - half, seam, R1, rho, s_star, e_star, x, hs, he, hx);
    if (lane == 0) {
        atomicAdd(merge_counter, 1u);
        recs[k].s = s_star;
        recs[k].e = e_star;
        recs[k].x = x;
        if (hints) {
            hints[k].s = s_star;
            hints[k].e = e_star;
            hints[k].x = x;
        }
    }
}

// One block per 4096 positions: it looks at the record of the seam its positions belong to and
// returns at once unless the pooled range reaches into them (most blocks of most levels).
constexpr int PF_CHUNK = 4096;
__global__ __launch_bounds__(256) void k_pav_fill(double* __restrict__ u, long long n, int level_shift,
                                                   const SeamRec* __restrict__ recs) {
    const long long b0 = (long long)blockIdx.x * PF_CHUNK;
    if (b0 >= n) return;
    const SeamRec r = recs[b0 >> level_shift];   // level_shift >= 12: a chunk never straddles two segments
    if (r.s < 0) return;
    long long lo = b0 > r.s ? b0 : r.s, hi = b0 + PF_CHUNK - 1;
    if (hi > r.e) hi = r.e;
    if (hi >= n) hi = n - 1;
    for (long long i = lo + threadIdx.x; i <= hi; i += 256) u[i] = r.x;
}

// ---------------------------------------------------------------- upper levels in ONE launch (round 3)
// Round 2 ran two kernels per upper level (k_pav_seam_wave + k_pav_fill: 12 levels at 6 M positions = 24 launches,
// ~22 us per level with the boundaries, 0.23-0.27 ms per z-step) although in the steady state of a solve only a few
// of the ~n / 2048 upper seams violate at all, and what they pool are short blocks.  One persistent launch instead:
//   phase 0   every upper seam of every level is looked at once (u[seam - 1] > u[seam]?): a bit per level that has a
//             violating seam (`dirty`, agent-scope atomic OR);
//   level l   is SKIPPED by every block when its bit is clear - no work, no barrier.  Otherwise one wave per seam as
//             before (seam_merge with the 64-ary searches and the hints of the previous ADMM iteration); the merging
//             wave writes the pooled range itself when it is short, long ranges (the first iterations pool most rows
//             into one block) go to a list all blocks fill together after the level's barrier; a pooled range that
//             reaches an end of its segment can make the seam of a HIGHER level at that end violate: its level's bit
//             is set.  One device-wide barrier (grid_sync.h) per level that did work, a second one behind a
//             cooperative fill.
// A level's bit can only be set by phase 0 or by merges of lower levels, all of which lie before the barrier after
// which the bit is read, so every block takes the same decisions and passes the same barriers.
constexpr long long PU_DIRECT_FILL = 8192;    // positions a merging wave writes itself
constexpr int PU_BIG_CAP = 4096;              // capacity of the cooperative-fill list

struct PavUpperArgs {
    SeamRec* hints;        // per level, concatenated (NULL: cold searches)
    SeamRec* big;          // cooperative-fill list
    u32* counters;         // [0] merges, [1] dirty levels, [2] entries of `big`, [3] status (1: a wait gave up / list overflow)
    unsigned* bar;
    int parity, nlevels;
};

template <int LOSS>
__global__ __launch_bounds__(PV_THREADS) void k_pav_upper(double* __restrict__ u, long long n, Prefix pa_, Prefix pb_, Prefix pm,
                                                           const int* branch, double rho, PavUpperArgs A) {
    __shared__ int s_flag;
    const int lane = threadIdx.x & 63;
    const long long wave_id = ((long long)blockIdx.x * PV_THREADS + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * PV_THREADS) >> 6;
    const long long tid_g = (long long)blockIdx.x * PV_THREADS + threadIdx.x, nthreads = (long long)gridDim.x * PV_THREADS;
    rbl::GridBarrier gb = rbl::gs_init(A.bar, A.parity);
    typedef unsigned gu32 __attribute__((address_space(1)));
    gu32* dirty = (gu32*)(A.counters + 1);
    gu32* bigcnt = (gu32*)(A.counters + 2);
    // ---- phase 0: which levels have a violating seam
    for (int l = 0; l < A.nlevels; ++l) {
        const long long half = (long long)PB_TILE << l, nseams = (n + 2 * half - 1) / (2 * half);
        bool any = false;
        for (long long k = tid_g; k < nseams; k += nthreads) {
            const long long seam = (2 * k + 1) * half;
            if (seam < n && u[seam - 1] > u[seam]) any = true;
        }
        if (__ballot(any) != 0ull && lane == 0) (void)__hip_atomic_fetch_or(dirty, 1u << l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    bool ok = rbl::gs_barrier(gb, &s_flag);
    u32 mask = __hip_atomic_load(dirty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u32 merges = 0, big_done = 0;
    const WaveAcc ac{u, (branch && *branch) ? pb_ : pa_, pm};
    long long hint_off = 0;
    for (int l = 0; ok && l < A.nlevels; ++l) {
        const long long half = (long long)PB_TILE << l, nseams = (n + 2 * half - 1) / (2 * half);
        SeamRec* hints = A.hints ? A.hints + hint_off : nullptr;
        hint_off += nseams;
        if (!((mask >> l) & 1u)) continue;
        for (long long k = wave_id; k < nseams; k += nwaves) {
            const long long seam = (2 * k + 1) * half;
            if (seam >= n || u[seam - 1] <= u[seam]) continue;   // pav.py:105 - only a strict decrease is a violation
            long long R1 = seam + half;
            if (R1 > n) R1 = n;
            const long long L0 = seam - half;
            long long s_star, e_star, hs = -1, he = -1;
            double x, hx = __builtin_nan("");
            if (hints) {
                hs = hints[k].s;
                he = hints[k].e;
                if (hs >= 0) hx = hints[k].x;
            }
            seam_merge<LOSS>(ac, L0, seam, R1, rho, s_star, e_star, x, hs, he, hx);
            if (lane == 0) {
                ++merges;
                if (hints) {
                    hints[k].s = s_star;
                    hints[k].e = e_star;
                    hints[k].x = x;
                }
                // a pooled range that reaches an end of its segment changes the value next to a higher level's seam
                u32 up = 0;
                if (s_star == L0 && L0 > 0) up |= 1u << __builtin_ctzll((unsigned long long)(L0 >> PB_TILE_LOG));
                if (e_star == R1 - 1 && R1 < n) up |= 1u << __builtin_ctzll((unsigned long long)(R1 >> PB_TILE_LOG));
                if (up) (void)__hip_atomic_fetch_or(dirty, up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (e_star - s_star + 1 <= PU_DIRECT_FILL) {
                for (long long i = s_star + lane; i <= e_star; i += 64) u[i] = x;
            } else if (lane == 0) {
                const u32 slot = __hip_atomic_fetch_add(bigcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (slot < (u32)PU_BIG_CAP) {
                    A.big[slot].s = s_star;
                    A.big[slot].e = e_star;
                    A.big[slot].x = x;
                } else {
                    __hip_atomic_store((gu32*)(A.counters + 3), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        ok = rbl::gs_barrier(gb, &s_flag);
        if (!ok) break;
        u32 nbig = __hip_atomic_load(bigcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (nbig > (u32)PU_BIG_CAP) nbig = PU_BIG_CAP;
        if (nbig > big_done) {
            for (u32 r = big_done; r < nbig; ++r) {
                const SeamRec rec = A.big[r];
                for (long long i = rec.s + tid_g; i <= rec.e; i += nthreads) u[i] = rec.x;
            }
            big_done = nbig;
            ok = rbl::gs_barrier(gb, &s_flag);
            if (!ok) break;
        }
        mask |= __hip_atomic_load(dirty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0 && merges) atomicAdd(A.counters, merges);
    if (!ok && threadIdx.x == 0) __hip_atomic_store((gu32*)(A.counters + 3), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------- setup kernels
__global__ void k_unflip(long long n, const u64* __restrict__ keys, double* __restrict__ ms) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        ms[i] = rbl::unflip_key(keys[i]);
}

// EHRM singleton-stage scalar test (src/util/PAV_cpt.py:205-226): opt1 = min(prox_a, B),
// opt2 = max(prox_b, B); fval_k = sum sigma_k*log(1+exp(opt_k)) + rho/2 ||opt_k - m||^2
__global__ __launch_bounds__(PV_THREADS) void k_ehrm_fvals(long long n, const double* __restrict__ sa,
                                                            const double* __restrict__ sb, double B, double rho,
                                                            const double* __restrict__ ms,
                                                            double* __restrict__ partials, double* __restrict__ u0a,
                                                            double* __restrict__ u0b) {
    // u0a / u0b (optional): the element prox of both branches, which is also level 0 of the PAV tree of
    // whichever branch wins - kept so that k_pav_bottom does not solve n Newton problems again
    __shared__ double smem[2 * PV_THREADS / 64];
    double acc[2] = {0.0, 0.0};
    for (long long i = (long long)blockIdx.x * PV_THREADS + threadIdx.x; i < n;
         i += (long long)gridDim.x * PV_THREADS) {
        const double m = ms[i];
        const double sm = rbl::sigmoid1(m);          // shared by the two element prox problems of this position
        double o1 = rbl::prox_bce_est(sa[i], rho, m, sm);
        double o2 = rbl::prox_bce_est(sb[i], rho, m, sm);
        if (u0a) {
            u0a[i] = o1;
            u0b[i] = o2;
        }
        if (o1 > B) o1 = B;
        if (o2 <= B) o2 = B;
        acc[0] += sa[i] * rbl::softplus(o1) + 0.5 * rho * (o1 - m) * (o1 - m);
        acc[1] += sb[i] * rbl::softplus(o2) + 0.5 * rho * (o2 - m) * (o2 - m);
    }
    rbl::block_sum<2, PV_THREADS>(acc, smem);
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 2 + 0] = acc[0];
        partials[blockIdx.x * 2 + 1] = acc[1];
    }
}

__global__ void k_ehrm_pick(const double* __restrict__ partials, int nblocks, int forced, int* __restrict__ branch) {
    __shared__ double smem[8];
    double acc[2] = {0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += 256) {
        acc[0] += partials[b * 2 + 0];
        acc[1] += partials[b * 2 + 1];
    }
    rbl::block_sum<2, 256>(acc, smem);
    if (threadIdx.x == 0) *branch = (forced >= 0) ? forced : ((acc[0] <= acc[1]) ? 0 : 1);  // PAV_cpt.py:225-226
}

// z[perm[i]] = clip(u[i]); c = z + lambda/rho  (algorithms.py:103-104 and :192)
__global__ void k_scatter_z(long long n, const double* __restrict__ u, const u32* __restrict__ perm,
                            const int* __restrict__ branch, double B, int has_B, double rho,
                            const double* __restrict__ lam, double* __restrict__ z, double* __restrict__ c,
                            long long off, long long nloc) {
    const int br = (has_B && branch) ? *branch : -1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const long long g = perm[i];
        if (g < off || g >= off + nloc) continue;
        double x = u[i];
        if (br == 0) x = fmin(x, B);       // branch a: all z <= B  (PAV_cpt.py:211)
        else if (br == 1) x = fmax(x, B);  // branch b: all z >= B  (PAV_cpt.py:218)
        const long long l = g - off;
        z[l] = x;
        if (c) c[l] = x + lam[l] / rho;
    }
}


// ================================================================ merge tree over RANKS
// Distributed z-step (several GPUs, rank weights): every rank owns a contiguous range of the
// globally sorted order and has solved it (launch_pav_tree above).  The chunks are joined by a
// merge tree over ranks; the seam rule is seam_merge()'s, but a probe of Psi(t) needs only
// (count, sum sigma, sum m) of {left: u > t} / {right: u < t}, which every rank computes on its
// own chunk (k_zd_eval) and the host sums over ranks (one small all-reduce).  The extents are
// found by a K-ary search: per round every rank proposes K of its still undecided u values
// (k_zd_update_propose), all candidates of a seam are evaluated together and every rank narrows
// its undecided index range [lo, hi) with the signs.  CPU restatement: oracle/zdist.py.
__device__ inline long long zd_upper_gt(const double* __restrict__ u, long long n, double t) {
    long long lo = 0, hi = n;   // first i with u[i] > t
    while (lo < hi) {
        const long long mid = lo + ((hi - lo) >> 1);
        if (u[mid] > t) hi = mid; else lo = mid + 1;
    }
    return lo;
}
__device__ inline long long zd_lower_ge(const double* __restrict__ u, long long n, double t) {
    long long lo = 0, hi = n;   // first i with u[i] >= t
    while (lo < hi) {
        const long long mid = lo + ((hi - lo) >> 1);
        if (u[mid] >= t) hi = mid; else lo = mid + 1;
    }
    return lo;
}

__global__ void k_zd_sample(const u64* __restrict__ keys, long long n, int ns, double* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ns) return;
    double r = __longlong_as_double(0x7ff8000000000000ll);   // NaN = no sample
    if (n > ns) r = rbl::unflip_key(keys[((long long)(j + 1) * n) / (ns + 1)]);
    else if (j < n) r = rbl::unflip_key(keys[j]);
    out[j] = r;
}

// bounds[j] = first sorted position whose key is >= the key of splitter j
__global__ void k_zd_split_bounds(const u64* __restrict__ keys, long long n, const double* __restrict__ split,
                                  int nsplit, long long* __restrict__ bounds) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nsplit) return;
    const u64 key = rbl::flip_key(split[j]);
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = lo + ((hi - lo) >> 1);
        if (keys[mid] >= key) hi = mid; else lo = mid + 1;
    }
    bounds[j] = lo;
}

// split positions -> how many sorted rows go to each part (equal splitters give empty parts); stays on the device
__global__ void k_zd_counts_from_bounds(const long long* __restrict__ bounds, int nparts, long long n,
                                        long long* __restrict__ counts) {
    if (threadIdx.x || blockIdx.x) return;
    long long prev = 0;
    for (int j = 0; j < nparts; ++j) {
        long long b = (j == nparts - 1) ? n : bounds[j];
        if (b < prev) b = prev;
        counts[j] = b - prev;
        prev = b;
    }
}

__global__ void k_zd_bounds(const double* __restrict__ u, long long n, double* __restrict__ out3) {
    if (threadIdx.x || blockIdx.x) return;
    out3[0] = n > 0 ? u[0] : 0.0;
    out3[1] = n > 0 ? u[n - 1] : 0.0;
    out3[2] = (double)n;
}

// oracle/zdist.py: seam_of + RankChunk.seam_setup
__global__ void k_zd_seam_setup(int rank, int world, int level, const double* __restrict__ bounds_all, long long n,
                                ZdSeam* __restrict__ st) {
    if (threadIdx.x || blockIdx.x) return;
    ZdSeam s;
    s.active = 0; s.k = 0; s.side = 0; s.a0 = 0; s.b1 = 0; s.lo = 0; s.hi = n; s.n = n; s.x = 0.0; s.cnt = 0.0;
    const int half = 1 << (level - 1);
    const int k = (rank / half) / 2;
    const int a0 = 2 * k * half, b0 = a0 + half;
    if (b0 < world) {
        const int b1 = (b0 + half < world) ? b0 + half : world;
        int last_left = -1, first_right = -1;
        for (int r = a0; r < b0; ++r)
            if (bounds_all[3 * r + 2] > 0.0) last_left = r;
        for (int r = b1 - 1; r >= b0; --r)
            if (bounds_all[3 * r + 2] > 0.0) first_right = r;
        if (last_left >= 0 && first_right >= 0 &&
            bounds_all[3 * last_left + 1] > bounds_all[3 * first_right + 0]) {   // pav.py:105: strict decrease only
            s.active = 1;
            s.k = k;
            s.side = rank < b0 ? 0 : 1;
            s.a0 = a0;
            s.b1 = b1;
        }
    }
    *st = s;
}

template <int LOSS>
__device__ inline double zd_psi(double cnt, double A, double M, double rho, double t) {
    if (!(cnt > 0.0)) return 0.0;
    if (LOSS == 0) return A * rbl::sigmoid1(t) + rho * (cnt * t - M);
    return t - block_value<1>(A, M, cnt, rho);
}

// narrow [lo, hi) with the summed (count, sum sigma, sum m) of the previous round's candidates
// (oracle/zdist.py: RankChunk.update), then propose K new candidates (RankChunk.propose)
template <int LOSS>
__global__ __launch_bounds__(256) void k_zd_update_propose(ZdSeam* __restrict__ st, const double* __restrict__ u, int K,
                                                            int world, const double* __restrict__ cand_prev,
                                                            const double* __restrict__ part_prev, double rho,
                                                            double* __restrict__ cand_out) {
    __shared__ long long s_lo, s_hi;
    const ZdSeam s = *st;
    if (threadIdx.x == 0) {
        s_lo = s.lo;
        s_hi = s.hi;
    }
    __syncthreads();
    if (s.active && cand_prev && part_prev) {
        for (int c = threadIdx.x; c < world * K; c += 256) {
            const double t = cand_prev[c];
            const int r = c / K;
            if (t != t || r < s.a0 || r >= s.b1) continue;
            const double sg = zd_psi<LOSS>(part_prev[3 * c], part_prev[3 * c + 1], part_prev[3 * c + 2], rho, t);
            if (s.side == 0) {   // s* = first left position with Psi(u[i]) > 0
                if (sg > 0.0) atomicMin(&s_hi, zd_lower_ge(u, s.n, t));
                else atomicMax(&s_lo, zd_upper_gt(u, s.n, t));
            } else {             // e* = last right position with Psi(u[j]) < 0
                if (sg < 0.0) atomicMax(&s_lo, zd_upper_gt(u, s.n, t));
                else atomicMin(&s_hi, zd_lower_ge(u, s.n, t));
            }
        }
    }
    __syncthreads();
    long long lo = s_lo, hi = s_hi;
    if (hi < lo) hi = lo;
    if (threadIdx.x == 0) {
        st->lo = lo;
        st->hi = hi;
    }
    if (cand_out && threadIdx.x < K) {
        const int j = threadIdx.x;
        double r = __longlong_as_double(0x7ff8000000000000ll);
        const long long sz = hi - lo;
        if (s.active && sz > 0) {
            if (sz <= K) {
                if (j < sz) r = u[lo + j];
            } else {
                r = u[lo + ((long long)(j + 1) * sz) / (K + 1)];
            }
        }
        cand_out[j] = r;
    }
}

// partial (count, sum sigma, sum m) of this rank for every candidate of its seam
__global__ __launch_bounds__(256) void k_zd_eval(const ZdSeam* __restrict__ st, const double* __restrict__ u, Prefix pa_,
                                                  Prefix pb_, Prefix pm, const int* __restrict__ branch, int K, int world,
                                                  const double* __restrict__ cand_all, double* __restrict__ part) {
    const ZdSeam s = *st;
    const Prefix pa = (branch && *branch) ? pb_ : pa_;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < world * K; c += gridDim.x * 256) {
        double cnt = 0.0, A = 0.0, M = 0.0;
        const double t = cand_all[c];
        const int r = c / K;
        if (s.active && t == t && r >= s.a0 && r < s.b1) {
            long long b, e;
            if (s.side == 0) {   // left group: positions with u > t (a suffix)
                b = zd_upper_gt(u, s.n, t);
                e = s.n;
            } else {             // right group: positions with u < t (a prefix)
                b = 0;
                e = zd_lower_ge(u, s.n, t);
            }
            if (e > b) {
                cnt = (double)(e - b);
                A = range_sum(pa, b, e);
                M = range_sum(pm, b, e);
            }
        }
        part[3 * c] = cnt;
        part[3 * c + 1] = A;
        part[3 * c + 2] = M;
    }
}

// (count, sum sigma, sum m) of this rank's pooled positions into its seam's slot
__global__ void k_zd_pooled(const ZdSeam* __restrict__ st, Prefix pa_, Prefix pb_, Prefix pm, const int* __restrict__ branch,
                            int nseams, double* __restrict__ sums, int* __restrict__ err) {
    const int i = threadIdx.x;
    if (i < 3 * nseams) sums[i] = 0.0;
    __syncthreads();
    if (i != 0) return;
    const ZdSeam s = *st;
    if (!s.active) return;
    if (s.lo != s.hi) {
        *err = 1;   // the search did not finish: more rounds needed (host raises)
        return;
    }
    const Prefix pa = (branch && *branch) ? pb_ : pa_;
    const long long b = s.side == 0 ? s.hi : 0, e = s.side == 0 ? s.n : s.lo;
    if (e > b) {
        sums[3 * s.k] = (double)(e - b);
        sums[3 * s.k + 1] = range_sum(pa, b, e);
        sums[3 * s.k + 2] = range_sum(pm, b, e);
    }
}

template <int LOSS>
__global__ void k_zd_fill_value(ZdSeam* __restrict__ st, const double* __restrict__ sums_total, double rho) {
    if (threadIdx.x || blockIdx.x) return;
    ZdSeam s = *st;
    s.cnt = 0.0;
    if (s.active) {
        const double cnt = sums_total[3 * s.k], A = sums_total[3 * s.k + 1], M = sums_total[3 * s.k + 2];
        if (cnt > 0.0) {
            s.cnt = cnt;
            s.x = block_value<LOSS>(A, M, cnt, rho);
        }
    }
    *st = s;
}

__global__ void k_zd_fill_range(const ZdSeam* __restrict__ st, double* __restrict__ u) {
    const ZdSeam s = *st;
    if (!s.active || !(s.cnt > 0.0)) return;
    const long long b = s.side == 0 ? s.hi : 0, e = s.side == 0 ? s.n : s.lo;
    for (long long i = b + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (long long)gridDim.x * blockDim.x)
        u[i] = s.x;
}

// return path: keys = row ids (for the sort by owner), vals = chunk positions
__global__ void k_zd_ids_to_keys(long long n, const u32* ids, u64* __restrict__ keys, u32* pos) {   // pos may alias ids
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        keys[i] = (u64)ids[i];
        pos[i] = (u32)i;
    }
}
__global__ void k_zd_gather_back(long long n, const u64* __restrict__ keys_sorted, const u32* __restrict__ pos,
                                 const double* __restrict__ u, u32* __restrict__ out_ids, double* __restrict__ out_u) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        out_ids[i] = (u32)keys_sorted[i];
        out_u[i] = u[pos[i]];
    }
}
__global__ void k_zd_owner_bounds(const u64* __restrict__ keys_sorted, long long n, long long nmax, int world,
                                  long long* __restrict__ bounds) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;   // bounds[j] = first position with id >= j * nmax
    if (j > world) return;
    const u64 key = (u64)j * (u64)nmax;
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = lo + ((hi - lo) >> 1);
        if (keys_sorted[mid] >= key) hi = mid; else lo = mid + 1;
    }
    bounds[j] = j == world ? n : lo;
}

// z[id - off] = clip(u); c = z + lambda/rho  (algorithms.py:103-104 and :192), rows received back
__global__ void k_zd_scatter(long long n, const u32* __restrict__ ids, const double* __restrict__ uu,
                             const int* __restrict__ branch, double B, int has_B, double rho,
                             const double* __restrict__ lam, double* __restrict__ z, double* __restrict__ c,
                             long long off, long long nloc) {
    const int br = (has_B && branch) ? *branch : -1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long l = (long long)ids[i] - off;
        if (l < 0 || l >= nloc) continue;
        double x = uu[i];
        if (br == 0) x = fmin(x, B);
        else if (br == 1) x = fmax(x, B);
        z[l] = x;
        if (c) c[l] = x + lam[l] / rho;
    }
}

}  // namespace

int64_t pav_num_chunks(int64_t n) { return n / PAV_CHUNK + 1; }
// per-level scratch records [0, L) followed by the hints of every upper level (sum over levels <= 2 L + 64)
static int64_t pav_level_recs(int64_t n) { return n / (2 * PB_TILE) + 2; }
int64_t pav_num_recs(int64_t n) { return 3 * pav_level_recs(n) + 64; }

int launch_prefix(const double* x, int64_t n, double* locx, double* chunk_tot, double* cph, double* cpl,
                  hipStream_t s) {
    const long long nc = pav_num_chunks(n);
    hipLaunchKernelGGL(k_chunk_scan<false>, dim3((unsigned)nc), dim3(256), 0, s, x, (long long)n, locx, chunk_tot,
                       (double*)nullptr);
    hipLaunchKernelGGL(k_chunk_prefix_dd, dim3(1), dim3(1024), 0, s, chunk_tot, nc, cph, cpl);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

// sorted keys -> sorted m (ms) and its two-level prefix sums in one pass
int launch_unflip_prefix(const u64* keys, int64_t n, double* ms, double* locx, double* chunk_tot, double* cph, double* cpl,
                         hipStream_t s) {
    const long long nc = pav_num_chunks(n);
    hipLaunchKernelGGL(k_chunk_scan<true>, dim3((unsigned)nc), dim3(256), 0, s, reinterpret_cast<const double*>(keys),
                       (long long)n, locx, chunk_tot, ms);
    hipLaunchKernelGGL(k_chunk_prefix_dd, dim3(1), dim3(1024), 0, s, chunk_tot, nc, cph, cpl);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int launch_unflip_keys(int64_t n, const u64* keys, double* ms, hipStream_t s) {
    if (n <= 0) return RBL_OK;
    hipLaunchKernelGGL(k_unflip, dim3(pv_grid(n)), dim3(256), 0, s, (long long)n, keys, ms);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int launch_ehrm_branch(int64_t n, const double* sa, const double* sb, double B, double rho, const double* ms,
                       double* partials, int* branch, int forced, hipStream_t s, double* u0a, double* u0b) {
    const int nb = reduce_blocks();
    hipLaunchKernelGGL(k_ehrm_fvals, dim3(nb), dim3(PV_THREADS), 0, s, (long long)n, sa, sb, B, rho, ms, partials, u0a, u0b);
    hipLaunchKernelGGL(k_ehrm_pick, dim3(1), dim3(256), 0, s, partials, nb, forced, branch);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

// the two singleton-stage sums of this chunk alone -> out2 (summed over ranks by the caller)
int launch_ehrm_fvals(int64_t n, const double* sa, const double* sb, double B, double rho, const double* ms,
                      double* partials, double* out2, hipStream_t s, double* u0a, double* u0b) {
    const int nb = reduce_blocks();
    hipLaunchKernelGGL(k_ehrm_fvals, dim3(nb), dim3(PV_THREADS), 0, s, (long long)n, sa, sb, B, rho, ms, partials, u0a, u0b);
    RBL_HIP(hipGetLastError());
    return launch_sum_partials(partials, nb, 2, out2, s);
}
// branch from the sums over all ranks (PAV_cpt.py:225-226)
int launch_ehrm_pick(const double* fvals_total, int* branch, hipStream_t s) {
    hipLaunchKernelGGL(k_ehrm_pick, dim3(1), dim3(256), 0, s, fvals_total, 1, -1, branch);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

namespace {
__global__ void k_add_u32(long long n, u32* __restrict__ x, u32 add) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        x[i] += add;
}
}  // namespace
namespace {
__global__ void k_make_c(long long n, const double* __restrict__ z, const double* __restrict__ lam, double rho,
                         double* __restrict__ c) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        c[i] = z[i] + lam[i] / rho;   // algorithms.py:192
}
}  // namespace
int launch_make_c(int64_t n, const double* z, const double* lam, double rho, double* c, hipStream_t s) {
    if (n <= 0) return RBL_OK;
    hipLaunchKernelGGL(k_make_c, dim3(pv_grid(n, 256, 4096)), dim3(256), 0, s, (long long)n, z, lam, rho, c);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int launch_add_u32(int64_t n, u32* x, u32 add, hipStream_t s) {
    if (n <= 0) return RBL_OK;
    hipLaunchKernelGGL(k_add_u32, dim3(pv_grid(n)), dim3(256), 0, s, (long long)n, x, add);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

size_t pav_bar_uints() { return (size_t)rbl::GS_UINTS; }
int64_t pav_big_recs() { return PU_BIG_CAP; }
int64_t pav_fpart_doubles(int64_t n) { return 2 * ((n + PB_TILE - 1) / PB_TILE) + 2; }

int launch_pav_tree(int loss, int64_t n, double rho, const double* ms, const double* sa, const double* sb, double* u,
                    Prefix pa, Prefix pb, Prefix pm, const int* branch, SeamRec* recs, u32* merge_counter,
                    hipStream_t s, const double* u0a, const double* u0b, PavExtras* ex) {
    RBL_HIP(hipMemsetAsync(merge_counter, 0, 4 * sizeof(u32), s));   // merges | dirty levels | long fills | status
    if (n <= 0) return RBL_OK;
    static const bool thread_seams = [] {
        const char* e = getenv("RBL_PAV_THREAD_SEAMS");   // one thread per seam everywhere, for comparison
        return e && e[0] == '1';
    }();
    static const bool no_seq = [] {
        const char* e = getenv("RBL_PAV_NO_SEQ");          // tree merges from segments of one position on (round 2), for comparison
        return e && e[0] == '1';
    }();
    static const bool no_upper = [] {
        const char* e = getenv("RBL_PAV_UPPER_PERSIST");   // =0: two launches per upper level (round 2), for comparison
        return e && e[0] == '0';
    }();
    const int bflags = (thread_seams ? 0 : 1) | (no_seq ? 2 : 0);
    // levels 0 .. log2(PB_TILE): prox + in-LDS merges, one tile per workgroup
    const unsigned tiles = (unsigned)((n + PB_TILE - 1) / PB_TILE);
    if (ex && ex->fpart && loss == RBL_LOSS_BCE && branch) {
        // EHRM, branch speculated (see k_pav_bottom): tree of the speculated branch + the two singleton-stage sums, the
        // exact test, and the other branch's tree only if the test says so (a no-op launch otherwise)
        int* br = const_cast<int*>(branch);
        hipLaunchKernelGGL((k_pav_bottom<0, true>), dim3(tiles), dim3(PV_THREADS), 0, s, ms, sa, sb, (const int*)nullptr, rho,
                           (long long)n, u, merge_counter, (const double*)nullptr, (const double*)nullptr, bflags, -1, ex->B,
                           ex->spec, ex->fpart);
        hipLaunchKernelGGL(k_ehrm_pick, dim3(1), dim3(256), 0, s, (const double*)ex->fpart, (int)tiles, -1, br);
        hipLaunchKernelGGL((k_pav_bottom<0, false>), dim3(tiles), dim3(PV_THREADS), 0, s, ms, sa, sb, branch, rho, (long long)n, u,
                           merge_counter, (const double*)nullptr, (const double*)nullptr, bflags, ex->spec, 0.0, 0,
                           (double*)nullptr);
    } else if (loss == RBL_LOSS_BCE) {
        hipLaunchKernelGGL((k_pav_bottom<0, false>), dim3(tiles), dim3(PV_THREADS), 0, s, ms, sa, sb, branch, rho, (long long)n, u,
                           merge_counter, u0a, u0b, bflags, -1, 0.0, 0, (double*)nullptr);
    } else {
        hipLaunchKernelGGL((k_pav_bottom<1, false>), dim3(tiles), dim3(PV_THREADS), 0, s, ms, sa, sb, branch, rho, (long long)n, u,
                           merge_counter, u0a, u0b, bflags, -1, 0.0, 0, (double*)nullptr);
    }
    // upper levels: one WAVE per seam (64-ary inner searches)
    static const bool no_hints = [] {
        const char* e = getenv("RBL_PAV_NO_HINTS");       // cold searches every iteration, for comparison
        return e && e[0] == '1';
    }();
    SeamRec* hint_base = no_hints ? nullptr : recs + pav_level_recs(n);
    if (ex && ex->bar && ex->big && !thread_seams && !no_upper && (long long)PB_TILE < n) {
        // ... all of them in one persistent launch (k_pav_upper): at most one block per CU
        PavUpperArgs A;
        A.hints = hint_base;
        A.big = ex->big;
        A.counters = merge_counter;
        A.bar = ex->bar;
        A.parity = ex->bar_parity;
        ex->bar_parity ^= 1;
        A.nlevels = 0;
        for (long long half = PB_TILE; half < n; half <<= 1) ++A.nlevels;
        long long want = ((n + 2LL * PB_TILE - 1) / (2LL * PB_TILE) + 3) / 4;    // one wave per seam of the lowest level
        int grid = ex->num_cu > 0 ? ex->num_cu : 64;
        if (want < grid) grid = (int)(want < 1 ? 1 : want);
        if (loss == RBL_LOSS_BCE)
            hipLaunchKernelGGL(k_pav_upper<0>, dim3(grid), dim3(PV_THREADS), 0, s, u, (long long)n, pa, pb, pm, branch, rho, A);
        else
            hipLaunchKernelGGL(k_pav_upper<1>, dim3(grid), dim3(PV_THREADS), 0, s, u, (long long)n, pa, pb, pm, branch, rho, A);
        RBL_HIP(hipGetLastError());
        return RBL_OK;
    }
    // ... or two launches per level (round 2): the seams of a level, then a fill pass over the pooled ranges
    int level = PB_TILE_LOG + 1;
    for (long long half = PB_TILE; half < n; half <<= 1, ++level) {
        const long long nseams = (n + 2 * half - 1) / (2 * half);
        SeamRec* hints = hint_base;
        if (hint_base) hint_base += nseams;
        if (thread_seams) {
            const unsigned grid = pv_grid(nseams, PV_THREADS, 1LL << 30);
            if (loss == RBL_LOSS_BCE)
                hipLaunchKernelGGL(k_pav_seam<0>, dim3(grid), dim3(PV_THREADS), 0, s, u, (long long)n, half, pa, pb, pm,
                                   branch, rho, recs, nseams, merge_counter);
            else
                hipLaunchKernelGGL(k_pav_seam<1>, dim3(grid), dim3(PV_THREADS), 0, s, u, (long long)n, half, pa, pb, pm,
                                   branch, rho, recs, nseams, merge_counter);
        } else {
            const unsigned grid = pv_grid(nseams * 64, PV_THREADS, 1LL << 30);
            if (loss == RBL_LOSS_BCE)
                hipLaunchKernelGGL(k_pav_seam_wave<0>, dim3(grid), dim3(PV_THREADS), 0, s, u, (long long)n, half, pa, pb,
                                   pm, branch, rho, recs, nseams, merge_counter, hints);
            else
                hipLaunchKernelGGL(k_pav_seam_wave<1>, dim3(grid), dim3(PV_THREADS), 0, s, u, (long long)n, half, pa, pb,
                                   pm, branch, rho, recs, nseams, merge_counter, hints);
        }
        hipLaunchKernelGGL(k_pav_fill, dim3((unsigned)((n + PF_CHUNK - 1) / PF_CHUNK)), dim3(256), 0, s, u, (long long)n, level,
                           recs);
    }
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int launch_scatter_z(int64_t n, const double* u, const u32* perm, const int* branch, double B, int has_B,
                     double rho, const double* lam, double* z, double* c, int64_t off, int64_t nloc,
                     hipStream_t s) {
    if (n <= 0) return RBL_OK;
    hipLaunchKernelGGL(k_scatter_z, dim3(pv_grid(n)), dim3(256), 0, s, (long long)n, u, perm, branch, B, has_B, rho,
                       lam, z, c, (long long)off, (long long)nloc);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

// ---------------------------------------------------------------- distributed z-step launchers
int launch_zd_sample(const u64* keys, int64_t n, int ns, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_zd_sample, dim3((ns + 255) / 256), dim3(256), 0, s, keys, (long long)n, ns, out);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
int launch_zd_split_bounds(const u64* keys, int64_t n, const double* split, int nsplit, long long* bounds, hipStream_t s) {
    if (nsplit <= 0) return RBL_OK;
    hipLaunchKernelGGL(k_zd_split_bounds, dim3((nsplit + 63) / 64), dim3(64), 0, s, keys, (long long)n, split, nsplit, bounds);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
int launch_zd_counts_from_bounds(const long long* bounds, int nparts, int64_t n, long long* counts, hipStream_t s) {
    hipLaunchKernelGGL(k_zd_counts_from_bounds, dim3(1), dim3(64), 0, s, bounds, nparts, (long long)n, counts);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
int launch_zd_bounds(const double* u, int64_t n, double* out3, hipStream_t s) {
    hipLaunchKernelGGL(k_zd_bounds, dim3(1), dim3(64), 0, s, u, (long long)n, out3);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
int launch_zd_seam_setup(int rank, int world, int level, const double* bounds_all, int64_t n, ZdSeam* st, hipStream_t s) {
    hipLaunchKernelGGL(k_zd_seam_setup, dim3(1), dim3(64), 0, s, rank, world, level, bounds_all, (long long)n, st);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
int launch_zd_update_propose(int loss, ZdSeam* st, const double* u, int K, int world, const double* cand_prev,
                             const double* part_prev, double rho, double* cand_out, hipStream_t s) {
    if (loss == RBL_LOSS_BCE)
        hipLaunchKernelGGL(k_zd_update_propose<0>, dim3(1), dim3(256), 0, s, st, u, K, world, cand_prev, part_prev, rho, cand_out);
    else
        hipLaunchKernelGGL(k_zd_update_propose<1>, dim3(1), dim3(256), 0, s, st, u, K, world, cand_prev, part_prev, rho, cand_out);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
int launch_zd_eval(const ZdSeam* st, const double* u, Prefix pa, Prefix pb, Prefix pm, const int* branch, int K, int world,
                   const double* cand_all, double* part, hipStream_t s) {
    const int total = K * world;
    hipLaunchKernelGGL(k_zd_eval, dim3((total + 255) / 256), dim3(256), 0, s, st, u, pa, pb, pm, branch, K, world, cand_all, part);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
int launch_zd_pooled(const ZdSeam* st, Prefix pa, Prefix pb, Prefix pm, const int* branch, int nseams, double* sums, int* err,
                     hipStream_t s) {
    hipLaunchKernelGGL(k_zd_pooled, dim3(1), dim3(256), 0, s, st, pa, pb, pm, branch, nseams, sums, err);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
int launch_zd_fill(int loss, ZdSeam* st, const double* sums_total, double rho, double* u, int64_t n, hipStream_t s) {
    if (loss == RBL_LOSS_BCE) hipLaunchKernelGGL(k_zd_fill_value<0>, dim3(1), dim3(64), 0, s, st, sums_total, rho);
    else hipLaunchKernelGGL(k_zd_fill_value<1>, dim3(1), dim3(64), 0, s, st, sums_total, rho);
    hipLaunchKernelGGL(k_zd_fill_range, dim3(pv_grid(n > 0 ? n : 1, 256, 4096)), dim3(256), 0, s, st, u);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
int launch_zd_ids_to_keys(int64_t n, const u32* ids, u64* keys, u32* pos, hipStream_t s) {
    if (n <= 0) return RBL_OK;
    hipLaunchKernelGGL(k_zd_ids_to_keys, dim3(pv_grid(n)), dim3(256), 0, s, (long long)n, ids, keys, pos);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
int launch_zd_gather_back(int64_t n, const u64* keys_sorted, const u32* pos, const double* u, u32* out_ids, double* out_u,
                          hipStream_t s) {
    if (n <= 0) return RBL_OK;
    hipLaunchKernelGGL(k_zd_gather_back, dim3(pv_grid(n)), dim3(256), 0, s, (long long)n, keys_sorted, pos, u, out_ids, out_u);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
int launch_zd_owner_bounds(const u64* keys_sorted, int64_t n, int64_t nmax, int world, long long* bounds, hipStream_t s) {
    hipLaunchKernelGGL(k_zd_owner_bounds, dim3((world + 64) / 64), dim3(64), 0, s, keys_sorted, (long long)n, (long long)nmax,
                       world, bounds);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
int launch_zd_scatter(int64_t n, const u32* ids, const double* uu, const int* branch, double B, int has_B, double rho,
                      const double* lam, double* z, double* c, int64_t off, int64_t nloc, hipStream_t s) {
    if (n <= 0) return RBL_OK;
    hipLaunchKernelGGL(k_zd_scatter, dim3(pv_grid(n)), dim3(256), 0, s, (long long)n, ids, uu, branch, B, has_B, rho, lam, z,
                       c, (long long)off, (long long)nloc);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}
