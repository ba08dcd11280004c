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
// (P 16-byte packets per lane and row, as k_gemv), reduces the R dot products with DPP
// butterflies, the R owner lanes do the row-wise update + warm-started prox, the R coefficients
// are broadcast back and the rows are accumulated into per-lane column sums.  The next
// sub-batch's loads are issued before the current one is processed (two register buffers);
// S sub-batches form a super-batch of 16 consecutive rows whose row-wise state is read and
// written as whole 128-byte lines.
// rho_{k+1} depends on the GLOBAL primal residual of iteration k; it is predicted in d-space
// before the pass (k_predict_rho: ||z - D w||^2 = ||z||^2 - 2 (D^T z)'w + w'Gw) and verified
// after it with the exact residual this kernel accumulates; on a misprediction the host
// simply recomputes the z-step / q with the unfused kernels (api.hip).
// Algorithmic bytes = n*ld*sizeof(T): half of the unfused iteration's.
#include "rbl_internal.h"
#ifndef RBL_D_AUX
#define RBL_D_AUX 2   // cache policy of the streaming loads of D: 2 = nt (non-temporal) on gfx950; 0 = default policy
#endif
#include "device_math.h"

namespace {

// a 16-byte packet as loaded by buffer_load_dwordx4
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Pk;
template <> struct Pk<float> {
    static constexpr int E = 4;
    __device__ static inline double at(const u32x4& p, int k) { return (double)__uint_as_float(p[k]); }
};
template <> struct Pk<double> {
    static constexpr int E = 2;
    __device__ static inline double at(const u32x4& p, int k) { return __hiloint2double((int)p[2 * k + 1], (int)p[2 * k]); }
};

constexpr int SE_THREADS = 256;

// Row-wise stores of the single-sweep kernels (lambda, v, z': 8 bytes per row and array, whole 128-byte lines per
// super-batch) are WRITE-THROUGH (sc1).  They are 0.4 % of the pass's bytes but cost 8 % of its time as plain stores
// (tools/sweep_lab.hip, 6M x 1000 fp32: v-only pass 3.81 ms, 3.39 without them, 3.40 when they go to a ring that stays
// in L2 - so it is not their issue but their way to HBM: dirty lines evicted from the write-back L2 by the streaming
// reads); written through they leave in order: 3.59 ms (nt 3.70; sc0 / sc1 nt / buffer-store forms the same as sc1).
// The EXP bits 256 / 512 / 2048 / 4096 exist for tools/sweep_lab.hip only (256: into a 1024-row ring, 512: non-temporal,
// 2048 + aux in bits 13-17: raw buffer store with that cache-policy field, 4096: plain stores as in round 2).
template <int EXP>
__device__ inline void row_store(double* __restrict__ base, long long row, double x) {
    if (EXP & 256) row &= 1023;
    if (EXP & 2048) {
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
        v2u val;
        val.x = (unsigned)b;
        val.y = (unsigned)(b >> 32);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b64(val, rs, (int)(row * 8), 0, (EXP >> 13) & 31);
    } else if (EXP & 512) __builtin_nontemporal_store(x, base + row);
    else if (EXP & 4096) base[row] = x;
    else
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(base + row), __builtin_bit_cast(unsigned long long, x), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

// fp32 storage: hide the packets from the optimiser between the dot phase and the accumulation
// phase (otherwise the widened fp64 copies of the dot phase are kept alive: 2 VGPRs per element)
template <typename T, int R, int P>
__device__ inline void opaque(u32x4 (&buf)[R][P]) {
    if (sizeof(T) == 4) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int p = 0; p < P; ++p) asm volatile("" : "+v"(buf[r][p]));
    }
}

// pred[0] = rho_{k+1} (predicted), read on the device so no host round trip is needed
// EXP selects what is left out.  The library instantiates EXP == 0 (everything) and
// EXP == SE_VONLY (rank-weighted problems: v = D w, the lambda update and the primal residual
// only - the z-step needs the global sort and q a second pass); the other values exist for
// tools/sweep_lab.hip (ablation timings: which phase costs what).
constexpr int SE_VONLY = 2 | 4 | 128;   // no accumulation phase, no prox, no z' / slab output
// q = D^T c alone (round 3; replaces k_gemvt where the row width fits): the row's coefficient is READ (c = z + lambda/rho,
// passed in the z_old slot) instead of computed - no dot product, no wave reduction, no prox, no row-wise stores, no
// residual sums; the same loads in the same order, the same accumulation into per-lane column sums, one slab row per
// block.  The thread-per-column-packet k_gemvt reads the same bytes at 6.0 TB/s, this kernel at the single-sweep
// kernel's 6.2-6.5 (buffer loads through scalar row descriptors, a whole sub-batch of rows in flight per wave).
constexpr int SE_QONLY = 1 | 4 | 8 | 16 | 64;
template <typename T, int LOSS, int P, int R, int S, bool WL, int EXP = 0, bool ONE = false, int OCC = 2>
__global__ __launch_bounds__(SE_THREADS, OCC) void k_sweep_erm(
    const T* __restrict__ D, long long n, long long ld, const double* __restrict__ w, const double* __restrict__ z_old,
    double* __restrict__ lam, double* __restrict__ v, double* __restrict__ z_new, double sigma0, double rho,
    const double* __restrict__ pred, double* __restrict__ slab, double* __restrict__ partials) {
    constexpr int E = Pk<T>::E;
    // The wave index goes through readfirstlane: super-batch numbers, row numbers and the row
    // descriptors are then scalar registers, and a load is buffer_load(SGPR descriptor, 32-bit
    // per-lane offset) - no 64-bit per-lane addresses, which is what keeps two R x P packet
    // buffers inside the 256-VGPR budget of 2 waves per SIMD.
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int PK = (int)(ld / E);
    const unsigned row_bytes = (unsigned)ld * (unsigned)sizeof(T);
    const double rho_next = (EXP & (128 | 64)) ? 1.0 : pred[0];

    // w: in registers, or (WL) in LDS, one copy per block, laid out [p][lane][k] so that a
    // lane's E values are contiguous - frees 2*P*E VGPRs for a second pair of row buffers
    double wr[WL ? 1 : P][E], acc[P][E];
    __shared__ double sw[WL ? 64 * P * E : 1];
    int boff[P];   // byte offset of this lane's packet p inside a row
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int pkp = lane + 64 * p;
        const bool ok = pkp < PK;
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const double wv = ok ? w[(long long)pkp * E + k] : 0.0;
            if (WL) {
                if (wave == 0) sw[(p * 64 + lane) * E + k] = wv;
            } else {
                wr[p][k] = wv;
            }
            acc[p][k] = 0.0;
        }
        // packets past the row end: an offset outside the row descriptor - the range check returns 0
        // without a memory access (w == 0 there and the column sums are dropped)
        boff[p] = ok ? pkp * 16 : 0x7ffffff0;
    }
    if (WL) __syncthreads();
    double s_prim = 0.0, s_zz = 0.0;

    // A wave works on super-batches of S*R consecutive rows (S sub-batches of R rows, two
    // register buffers alternating); lane l < S*R owns row q*S*R + l of super-batch q, so the
    // row-wise reads (z_old, lambda) and writes (lambda, v, z') of a super-batch are S*R*8
    // contiguous bytes per array - 128 B for S*R = 16 - instead of R*8-byte fragments (partial
    // cache-line writes cost ~8% of the pass, tools/sweep_lab.hip).
    // Super-batch numbers are 32-bit (n < 2^31 * S*R rows) so that all the loop control is SALU:
    // there is no scalar 64-bit compare.
    constexpr int SR = S * R;
    const int nsuper = (int)((n + SR - 1) / SR);
    const int live_last = (int)(n - (long long)(nsuper - 1) * SR);   // rows of the last super-batch
    const int gw = (int)blockIdx.x * (SE_THREADS / 64) + wave;
    const int GW = (int)gridDim.x * (SE_THREADS / 64);

    // Loads are issued in the order they are consumed: s_waitcnt vmcnt counts in issue order,
    // so waiting for sub-batch t then leaves the whole of sub-batch t+1 in flight (a row-wise
    // load issued later than the prefetch would drain it).
    auto load_side = [&](int q, double& zo, double& lm) {
        const int live = q == nsuper - 1 ? live_last : SR;
        const bool mine = lane < live && !(EXP & 32);
        const long long myrow = (long long)q * SR + lane;
        zo = mine ? z_old[myrow] : 0.0;
        lm = (mine && !(EXP & 64)) ? lam[myrow] : 0.0;
    };
    auto load_rows = [&](int q, int sub, u32x4 (&buf)[R][P]) {
        const int live = q == nsuper - 1 ? live_last : SR;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int i = sub * R + r;
            if (i >= live) i = live - 1;   // rows past n re-read the last row; their coefficient is 0
            const long long row = (long long)q * SR + i;
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(D + row * ld), 0, (int)row_bytes, 0x00020000);
#pragma unroll
            for (int p = 0; p < P; ++p) buf[r][p] = __builtin_amdgcn_raw_buffer_load_b128(rs, boff[p], 0, RBL_D_AUX);
        }
    };

    double l_out = 0.0, v_out = 0.0, z_out = 0.0;   // this lane's row of the current super-batch
    auto process = [&](int live, int sub, u32x4 (&buf)[R][P], double zo, double lm) {
        double dot[R];
#pragma unroll
        for (int r = 0; r < R; ++r) dot[r] = (EXP & 16) ? Pk<T>::at(buf[r][0], 0) : 0.0;
        // the LDS offset is made opaque so that the (loop-invariant) reads of w are not hoisted
        // back into 2*P*E registers
        int woff = lane * E;
        if (WL) asm volatile("" : "+v"(woff));
#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (EXP & 16) break;
            double wv[E];
#pragma unroll
            for (int k = 0; k < E; ++k) wv[k] = WL ? sw[p * 64 * E + woff + k] : wr[WL ? 0 : p][k];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int k = 0; k < E; ++k) dot[r] = __builtin_fma(Pk<T>::at(buf[r][p], k), wv[k], dot[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (!(EXP & 8)) dot[r] = rbl::wave_sum_all(dot[r]);   // DPP butterfly + readlanes: no LDS trips
        double myv = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) myv = (lane == sub * R + r) ? dot[r] : myv;
        double c = 0.0;
        if (EXP & 64) {
            if (lane >= sub * R && lane < sub * R + R && lane < live) c = zo;    // the coefficient was read, not computed
        } else if (lane >= sub * R && lane < sub * R + R && lane < live) {
            const double res = zo - myv;
            const double l = lm + rho * res;                       // algorithms.py:132
            s_prim += res * res;                                   // algorithms.py:135
            const double lr = l / rho_next;
            const double m = myv - lr;                             // algorithms.py:89 (next iteration)
            const double zn = (EXP & 4) ? m : ((LOSS == 0) ? rbl::prox_bce_warm(sigma0, rho_next, m, zo) : rbl::prox_hinge(sigma0, rho_next, m));
            s_zz += zn * zn;
            l_out = l;
            v_out = myv;
            z_out = zn;
            c = zn + lr;
        }
        opaque<T>(buf);
        if (EXP & 2) {
            s_zz += c;
            return;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double cr = rbl::readlane_d(c, sub * R + r);
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int k = 0; k < E; ++k) acc[p][k] = __builtin_fma(Pk<T>::at(buf[r][p], k), cr, acc[p][k]);
        }
    };

    u32x4 bufA[R][P], bufB[R][P];
    double zo = 0.0, lm = 0.0, zoN = 0.0, lmN = 0.0;
    int q = gw, sub = 0;
    if (ONE) {
        // One copy of the row-wise code, as in k_sweep_erm_wide: every load lands in bufB; at the
        // top of an iteration bufB (issued one whole process() earlier) is moved to bufA, the next
        // sub-batch is requested into bufB and the arithmetic runs on bufA, whose registers are
        // never the target of a load - nothing in process() waits for the prefetch in flight.
        if (q < nsuper) {
            load_side(q, zoN, lmN);
            load_rows(q, 0, bufB);
        }
#pragma clang loop unroll(disable)
        while (q < nsuper) {
            const int live = q == nsuper - 1 ? live_last : SR;
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int p = 0; p < P; ++p) bufA[r][p] = bufB[r][p];
            if (sub == 0) {
                zo = zoN;
                lm = lmN;
            }
            const bool last = sub + 1 == S;
            const int qn = last ? q + GW : q;
            const int subn = last ? 0 : sub + 1;
            __builtin_amdgcn_sched_barrier(0);
            if (qn < nsuper) {
                if (last) load_side(qn, zoN, lmN);
                load_rows(qn, subn, bufB);
            }
            __builtin_amdgcn_sched_barrier(0);   // the prefetch stays above the arithmetic
            process(live, sub, bufA, zo, lm);
            if (last && lane < live && !(EXP & 1)) {
                const long long row = (long long)q * SR + lane;
                row_store<EXP>(lam, row, l_out);
                if (v) row_store<EXP>(v, row, v_out);   // NULL: nobody reads v before the next pass (no objective logging)
                if (!(EXP & 128)) row_store<EXP>(z_new, row, z_out);
            }
            q = qn;
            sub = subn;
        }
    } else {
    if (q < nsuper) {
        load_side(q, zo, lm);
        load_rows(q, 0, bufA);
    }
    // one flat loop over pairs of sub-batches ((q, sub) in bufA, (q, sub+1) in bufB); q and sub
    // are scalar registers.  (A nested "for sub" loop lets the compiler hoist the next
    // super-batch's load addresses out of it and spill them.)
    while (q < nsuper) {
        const int live = q == nsuper - 1 ? live_last : SR;
        load_rows(q, sub + 1, bufB);
        process(live, sub, bufA, zo, lm);
        const bool last = sub + 2 == S;
        const int qn = last ? q + GW : q;
        const int subn = last ? 0 : sub + 2;
        if (qn < nsuper) {
            if (last) load_side(qn, zoN, lmN);
            load_rows(qn, subn, bufA);
        }
        process(live, sub + 1, bufB, zo, lm);
        if (last) {
            if (lane < live && !(EXP & 1)) {
                const long long row = (long long)q * SR + lane;
                lam[row] = l_out;
                if (v) v[row] = v_out;   // NULL: nobody reads v before the next pass (no objective logging)
                if (!(EXP & 128)) z_new[row] = z_out;
            }
            zo = zoN;
            lm = lmN;
        }
        q = qn;
        sub = subn;
    }
    }

    // fold the 4 waves' column sums in LDS, one slab row per block
    __shared__ double red[(EXP & 128) ? 1 : SE_THREADS / 64][(EXP & 128) ? 1 : 64 * P * E];
    if (!(EXP & 128)) {
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
    }
    if (EXP & 64) return;                  // q-only: no residual sums
    double sums[3] = {s_prim, 0.0, s_zz};   // slot 1: the loss sum, filled by k_loss_sum when wanted
    __shared__ double smem[3 * SE_THREADS / 64];
    rbl::block_sum<3, SE_THREADS>(sums, smem);
    if (tid == 0) {
        partials[blockIdx.x * 3 + 0] = sums[0];
        partials[blockIdx.x * 3 + 1] = sums[1];
        partials[blockIdx.x * 3 + 2] = sums[2];
    }
}

// ---- wide rows (more than 8 x 64 packets, e.g. d = 10 000): one WORKGROUP of 8 waves per row
// batch.  Thread t owns packets t, t+512, ... of every row (PT packets), so the column sums stay in
// registers exactly as in the wave-per-row kernel; the dot product is folded across the 8 waves
// through LDS (fixed order), the owner lane of each row does the row-wise update + prox, and the
// coefficients come back through LDS: two workgroup barriers per batch of R rows, hidden behind the
// next batch's loads already in flight.  Super-batches of S*R consecutive rows (16-24), same arithmetic
// per row, one slab row per workgroup.  w lives in LDS ([p][thread][k], up to 128 KB: one workgroup per
// CU) and so do the row-wise values of the super-batch in flight (z_old, lambda, the three outputs and the
// two running sums of each row's owner thread), which leaves the registers to the two row buffers and
// the column sums: R is the largest number of rows whose two buffers fit (launch_T).
constexpr int SEW_THREADS = 512;

template <typename T, int LOSS, int PT, int R, int S, int NT = SEW_THREADS>
__global__ __launch_bounds__(NT, 1) void k_sweep_erm_wide(
    const T* __restrict__ D, long long n, long long ld, const double* __restrict__ w, const double* __restrict__ z_old,
    double* __restrict__ lam, double* __restrict__ v, double* __restrict__ z_new, double sigma0, double rho,
    const double* __restrict__ pred, double* __restrict__ slab, double* __restrict__ partials) {
    constexpr int E = Pk<T>::E;
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int PK = (int)(ld / E);
    const unsigned row_bytes = (unsigned)ld * (unsigned)sizeof(T);
    const double rho_next = pred[0];

    extern __shared__ __align__(16) double sw_wide[];   // PT * NT * E doubles
    double acc[PT][E];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        int pkp = tid + NT * p;
        const bool ok = pkp < PK;
#pragma unroll
        for (int k = 0; k < E; ++k) {
            sw_wide[(p * NT + tid) * E + k] = ok ? w[(long long)pkp * E + k] : 0.0;
            acc[p][k] = 0.0;
        }
    }
    // Per-lane byte offsets of the packets inside a row: tid * 16 + p * NT * 16 - one VGPR and an immediate per packet.
    // Only the LAST packet group can reach past the row end (PK > (PT - 1) * NT by the choice of PT): its lanes past the
    // end use an offset outside the row descriptor - the range check returns 0 without a memory access (w == 0 there
    // and the column sums are dropped).
    const int voff = tid * 16;
    const int voff_last = (tid + NT * (PT - 1) < PK) ? voff + NT * (PT - 1) * 16 : 0x7ffffff0;

    constexpr int SR = S * R;
    const int nsuper = (int)((n + SR - 1) / SR);
    const int live_last = (int)(n - (long long)(nsuper - 1) * SR);
    const int GW = (int)gridDim.x;

    __shared__ double part[NW][R];   // per-wave partial dot products of the batch
    __shared__ double cshare[R];     // coefficients of the batch's rows
    // Row-wise values of the super-batch in flight, one slot per row, written and read by the SAME thread (thread i < SR
    // owns row i of the super-batch, so no synchronisation): registers moved to LDS by hand - z_old and lambda of the
    // row, its three outputs and the thread's two running sums, 14 VGPRs.  At d = 10 000 that is what lets a THIRD row per
    // sub-batch fit without a spill (three rows of 5 packets in flight per wave instead of two: C5shard 124.6 -> 132
    // it/s); the LDS round trips on the owner's path between the two barriers cost ~1.5 % at equal R.
    __shared__ double rw_zo[SR], rw_lm[SR], rw_l[SR], rw_v[SR], rw_z[SR], rw_prim[SR], rw_zz[SR];
    if (tid < SR) {
        rw_prim[tid] = 0.0;
        rw_zz[tid] = 0.0;
    }

    auto load_side = [&](int q, double& zo, double& lm) {
        const int live = q == nsuper - 1 ? live_last : SR;
        const bool mine = tid < live;
        const long long myrow = (long long)q * SR + tid;
        zo = mine ? z_old[myrow] : 0.0;
        lm = mine ? lam[myrow] : 0.0;
    };
    auto load_rows = [&](int q, int sub, u32x4 (&buf)[R][PT]) {
        const int live = q == nsuper - 1 ? live_last : SR;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int i = sub * R + r;
            if (i >= live) i = live - 1;
            const long long row = (long long)q * SR + i;
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(D + row * ld), 0, (int)row_bytes, 0x00020000);
#pragma unroll
            for (int p = 0; p < PT; ++p)
                buf[r][p] = __builtin_amdgcn_raw_buffer_load_b128(rs, p == PT - 1 ? voff_last : voff + NT * p * 16, 0, RBL_D_AUX);
        }
    };

    auto process = [&](int live, int sub, u32x4 (&buf)[R][PT]) {
        double dot[R];
#pragma unroll
        for (int r = 0; r < R; ++r) dot[r] = 0.0;
        int woff = tid * E;   // opaque: keeps the loop-invariant LDS reads of w from being hoisted into registers
        asm volatile("" : "+v"(woff));
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            double wv[E];
#pragma unroll
            for (int k = 0; k < E; ++k) wv[k] = sw_wide[p * NT * E + woff + k];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int k = 0; k < E; ++k) dot[r] = __builtin_fma(Pk<T>::at(buf[r][p], k), wv[k], dot[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            dot[r] = rbl::wave_sum_all(dot[r]);
            if (lane == 0) part[wave][r] = dot[r];
        }
        __syncthreads();
        double c = 0.0;
        if (tid >= sub * R && tid < sub * R + R && tid < live) {   // owner of row q*SR + tid (wave 0)
            const int r = tid - sub * R;
            double myv = 0.0;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) myv += part[wv][r];
            const double zo = rw_zo[tid];
            const double res = zo - myv;
            const double l = rw_lm[tid] + rho * res;               // algorithms.py:132
            rw_prim[tid] += res * res;                             // algorithms.py:135
            const double lr = l / rho_next;
            const double m = myv - lr;                             // algorithms.py:89 (next iteration)
            const double zn = (LOSS == 0) ? rbl::prox_bce_warm(sigma0, rho_next, m, zo) : rbl::prox_hinge(sigma0, rho_next, m);
            rw_zz[tid] += zn * zn;
            rw_l[tid] = l;
            rw_v[tid] = myv;
            rw_z[tid] = zn;
            c = zn + lr;
        }
        if (tid >= sub * R && tid < sub * R + R) cshare[tid - sub * R] = c;   // 0 for rows past n
        opaque<T>(buf);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double cr = cshare[r];
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
                for (int k = 0; k < E; ++k) acc[p][k] = __builtin_fma(Pk<T>::at(buf[r][p], k), cr, acc[p][k]);
        }
    };

    // One copy of the row-wise code (unrolling by two would duplicate the inlined prox and its
    // constants and spill): every load lands in bufB; at the top of an iteration bufB - complete by
    // then, it was issued one whole process() earlier - is moved to bufA (R*PT register moves), the
    // NEXT sub-batch is requested into bufB, and the arithmetic runs on bufA, whose registers are
    // never the target of a load (so nothing in process() waits for the prefetch in flight).
    u32x4 bufA[R][PT], bufB[R][PT];
    double zoN = 0.0, lmN = 0.0;
    int q = (int)blockIdx.x, sub = 0;
    if (q < nsuper) {
        load_side(q, zoN, lmN);
        load_rows(q, 0, bufB);
    }
#pragma clang loop unroll(disable)
    while (q < nsuper) {
        const int live = q == nsuper - 1 ? live_last : SR;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int p = 0; p < PT; ++p) bufA[r][p] = bufB[r][p];
        if (sub == 0 && tid < SR) {
            rw_zo[tid] = zoN;
            rw_lm[tid] = lmN;
        }
        const bool last = sub + 1 == S;
        const int qn = last ? q + GW : q;
        const int subn = last ? 0 : sub + 1;
        __builtin_amdgcn_sched_barrier(0);
        if (qn < nsuper) {
            if (last) load_side(qn, zoN, lmN);
            load_rows(qn, subn, bufB);
        }
        __builtin_amdgcn_sched_barrier(0);   // the prefetch stays above the arithmetic
        process(live, sub, bufA);
        if (last) {
            if (tid < live) {
                const long long row = (long long)q * SR + tid;
                row_store<0>(lam, row, rw_l[tid]);
                if (v) row_store<0>(v, row, rw_v[tid]);   // NULL: nobody reads v before the next pass (no objective logging)
                row_store<0>(z_new, row, rw_z[tid]);
            }
        }
        q = qn;
        sub = subn;
    }
    const double s_prim = tid < SR ? rw_prim[tid] : 0.0, s_zz = tid < SR ? rw_zz[tid] : 0.0;

    // every thread owns its columns: the slab row of this workgroup is written directly
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int pkp = tid + NT * p;
        if (pkp < PK) {
#pragma unroll
            for (int k = 0; k < E; ++k) slab[(long long)blockIdx.x * ld + (long long)pkp * E + k] = acc[p][k];
        }
    }
    __syncthreads();
    double sums[3] = {s_prim, 0.0, s_zz};
    __shared__ double smem[3 * NW];
    rbl::block_sum<3, NT>(sums, smem);
    if (tid == 0) {
        partials[blockIdx.x * 3 + 0] = sums[0];
        partials[blockIdx.x * 3 + 1] = sums[1];
        partials[blockIdx.x * 3 + 2] = sums[2];
    }
}

// Column sums of the nb slab rows in two steps: CR_SLICES x ceil(ld/64) blocks each fold their share
// of the rows into part[slice][ld] (fixed order), k_finish_sweep adds the slices.
constexpr int CR_SLICES = 8;
__global__ __launch_bounds__(256) void k_colreduce2(const double* __restrict__ slab, int nb, long long ld,
                                                      double* __restrict__ part) {
    __shared__ double red[4][64];
    const int cx = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long long col = (long long)blockIdx.x * 64 + cx;
    const int slice = blockIdx.y;
    const int per = (nb + CR_SLICES - 1) / CR_SLICES;
    const int b0 = slice * per, b1 = min(nb, b0 + per);
    double acc = 0.0;
    if (col < ld)
        for (int b = b0 + g; b < b1; b += 4) acc += slab[(long long)b * ld + col];
    red[g][cx] = acc;
    __syncthreads();
    if (g == 0 && col < ld) part[(long long)slice * ld + col] = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
}

// q[j] = sum_slices part[slice][j];  red[0] = sum primal^2, red[1] = 0 (the loss sum comes from
// k_loss_sum when wanted), zz_out[0] = sum z'^2
__global__ __launch_bounds__(256) void k_finish_sweep(const double* __restrict__ part, long long ld,
                                                       double* __restrict__ q, const double* __restrict__ partials,
                                                       int nb, double* __restrict__ red, double* __restrict__ zz_out) {
    for (long long j = threadIdx.x; j < ld; j += 256) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < CR_SLICES; ++k) s += part[(long long)k * ld + j];
        q[j] = s;
    }
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

// After the w-step, before the pass: the dual residual / regulariser sums of the new w
//   wstats[0] = ||w - w_prev||^2 (algorithms.py:136), wstats[1] = sum w^2, wstats[2] = ||w||_1 (objective.py:83-86)
// and the prediction of rho_{k+1}.  Gw = G w_{k+1} (k_symv), q = D^T(z + lambda/rho) summed over
// ranks, p = D^T lambda (kept by the recurrence below, written to p_out), zz = ||z||^2:
//   D^T z = q - p/rho ;  ||z - D w||^2 = zz - 2 (D^T z)'w + w'Gw ;  p_out = p + rho (D^T z - G w)
__global__ __launch_bounds__(1024) void k_predict_rho(long long ld, const double* __restrict__ q,
                                                       const double* __restrict__ p, double* __restrict__ p_out,
                                                       const double* __restrict__ w, const double* __restrict__ w_prev,
                                                       const double* __restrict__ Gw, const double* __restrict__ zz,
                                                       double rho_val, const double* rho_dev, double cap,
                                                       double* pred, double* __restrict__ wstats) {
    __shared__ double smem[5 * 16];
    const double rho = rho_dev ? rho_dev[0] : rho_val;   // rho_dev may alias pred: read before the barrier below
    double a[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (long long j = threadIdx.x; j < ld; j += 1024) {
        const double wj = w[j];
        const double dz = q[j] - p[j] / rho;
        a[0] += dz * wj;
        a[1] += wj * Gw[j];
        p_out[j] = p[j] + rho * (dz - Gw[j]);
        const double t = wj - w_prev[j];
        a[2] += t * t;
        a[3] += wj * wj;
        a[4] += fabs(wj);
    }
    rbl::block_sum<5, 1024>(a, smem);
    if (threadIdx.x == 0) {
        double pr2 = zz[0] - 2.0 * a[0] + a[1];
        if (pr2 < 0.0) pr2 = 0.0;
        const double primal = sqrt(pr2);
        double rn = rho * (primal > 1e-2 ? 1.02 : 1.07);  // algorithms.py:154-157
        if (rn > cap) rn = cap;
        pred[0] = rn;
        pred[1] = primal;
        wstats[0] = a[2];
        wstats[1] = a[3];
        wstats[2] = a[4];
    }
}

__global__ __launch_bounds__(256) void k_sumsq(long long n, const double* __restrict__ x, double* __restrict__ partials) {
    __shared__ double smem[4];
    double a[1] = {0.0};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) a[0] += x[i] * x[i];
    rbl::block_sum<1, 256>(a, smem);
    if (threadIdx.x == 0) partials[blockIdx.x] = a[0];
}

template <typename T, int LOSS, int P, int R, int S, bool WL, int OCC = 2>
int launch_one(const T* D, long long n, long long ld, const double* w, const double* z_old, double* lam, double* v,
               double* z_new, double sigma0, double rho, const double* pred, double* slab, double* partials, int grid,
               hipStream_t s) {
    hipLaunchKernelGGL((k_sweep_erm<T, LOSS, P, R, S, WL, 0, true, OCC>), dim3(grid), dim3(SE_THREADS), 0, s, D, n, ld, w, z_old, lam, v,
                       z_new, sigma0, rho, pred, slab, partials);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

template <typename T, int LOSS>
int launch_T(const T* D, long long n, long long ld, const double* w, const double* z_old, double* lam, double* v,
             double* z_new, double sigma0, double rho, const double* pred, double* slab, double* partials, int grid,
             int num_cu, int* used, hipStream_t s) {
    const long long PK = ld / Pk<T>::E;
    const long long passes = (PK + 63) / 64;
    *used = grid;
    // 3-4 (fp64 storage: 5-8) packets per lane: ONE block of 4 waves per CU, four (two) rows per sub-batch and two
    // sub-batch buffers - 32 KB of loads in flight per wave - with w in LDS.  Round 2's shape (two rows per sub-batch,
    // two blocks per CU, w in registers) is 5 % slower at 6M x 1000 and ramps over its first ~25 launches after an idle
    // gap (3.93 -> 3.66 ms; this shape 3.57 -> 3.48): interleaved on one box, C2 266-271 it/s against 281-283
    // (profiles/r03_sweep_shapes.txt).  Fewer waves, each with more rows in flight, stream better than more waves:
    // with 8 waves per CU the same rows-in-flight give 276-278.  RBL_SWEEP_SHAPE=0 is round 2's shape, 1 the
    // two-blocks-per-CU form of the default, 8 eight rows per sub-batch (same speed as the default).
    static const int shape = [] {
        const char* e = getenv("RBL_SWEEP_SHAPE");
        return e ? atoi(e) : -1;
    }();
    static const int bpc1 = [] {
        const char* e = getenv("RBL_SWEEP_BLOCKS_PER_CU");
        const int v = e ? atoi(e) : 1;
        return (v >= 1 && v <= 2) ? v : 1;
    }();
#define RBL_ONE(P_, R_, S_, WL_)                                                                                              \
    do {                                                                                                                      \
        *used = num_cu * bpc1;                                                                                                \
        return launch_one<T, LOSS, P_, R_, S_, WL_, 1>(D, n, ld, w, z_old, lam, v, z_new, sigma0, rho, pred, slab, partials, \
                                                       num_cu * bpc1, s);                                                     \
    } while (0)
    // 1 and 2 packets per lane (fp32 storage: d <= 256 / 512).  One packet: the per-row work (wave reduction, prox on one
    // lane) weighs most, more waves hide it - 16 rows per sub-batch, still two blocks per CU: 24M x 250 201.5 -> 224 it/s
    // (one block per CU: 184-189).  Two packets: 8 rows per sub-batch, super-batches of 64 rows (512-byte row-wise
    // segments), one block per CU: 12M x 500 255.5 -> 269 (profiles/r03_sweep_shapes.txt block 8).
    if (passes == 1) {
        if (shape == 0) return launch_one<T, LOSS, 1, 8, 2, false>(D, n, ld, w, z_old, lam, v, z_new, sigma0, rho, pred, slab, partials, grid, s);
        return launch_one<T, LOSS, 1, 16, 2, false>(D, n, ld, w, z_old, lam, v, z_new, sigma0, rho, pred, slab, partials, grid, s);
    }
    if (passes == 2) {
        if (shape == 0) return launch_one<T, LOSS, 2, 4, 4, false>(D, n, ld, w, z_old, lam, v, z_new, sigma0, rho, pred, slab, partials, grid, s);
        RBL_ONE(2, 8, 8, false);
    }
    if (passes <= 4) {
        if (shape == 0) return launch_one<T, LOSS, 4, 2, 8, false>(D, n, ld, w, z_old, lam, v, z_new, sigma0, rho, pred, slab, partials, grid, s);
        if (shape == 1) return launch_one<T, LOSS, 4, 4, 4, true>(D, n, ld, w, z_old, lam, v, z_new, sigma0, rho, pred, slab, partials, grid, s);
        if (shape == 8) RBL_ONE(4, 8, 2, true);
        RBL_ONE(4, 4, 4, true);
    }
    if (passes <= 8) {
        // 5-8 packets per lane and row (fp64 storage d <= 1024, fp32 storage d <= 2048).  fp64, d = 1000: four rows of 8
        // packets per sub-batch (64 KB in flight per wave) 143.7 it/s against round 2's one row per sub-batch with two
        // blocks per CU 136.4-138.6 (RBL_SWEEP_SHAPE=0), two rows 140.5-143.5.  fp32 storage took the workgroup-per-row
        // kernel from d = 1025 on until round 3 (4M x 2000: 171.8 it/s, 5.5 TB/s); RBL_SWEEP_SHAPE=0 keeps that.
        if constexpr (sizeof(T) == 8) {
            if (shape == 0) return launch_one<T, LOSS, 8, 1, 8, false>(D, n, ld, w, z_old, lam, v, z_new, sigma0, rho, pred, slab, partials, grid, s);
        }
        if (sizeof(T) == 8 || shape != 0) {
            if (shape == 9) RBL_ONE(8, 2, 8, true);
            RBL_ONE(8, 4, 4, true);
        }
    }
#undef RBL_ONE
    // wider rows: one workgroup of 512 threads per row batch (grid = one workgroup per CU)
    const long long pt = (PK + SEW_THREADS - 1) / SEW_THREADS;
    const int wgrid = grid / 2;
    *used = wgrid;
#define RBL_WIDE(PT_, R_, S_) RBL_WIDE_NT(PT_, R_, S_, SEW_THREADS)
#define RBL_WIDE_NT(PT_, R_, S_, NT_)                                                                                  \
    do {                                                                                                               \
        auto kfn = k_sweep_erm_wide<T, LOSS, PT_, R_, S_, NT_>;                                                        \
        const size_t lds = (size_t)PT_ * NT_ * Pk<T>::E * sizeof(double);                                              \
        static bool attr_set = false;                                                                                  \
        if (!attr_set) {                                                                                               \
            RBL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)lds));                                                                    \
            attr_set = true;                                                                                           \
        }                                                                                                              \
        hipLaunchKernelGGL(kfn, dim3(wgrid), dim3(NT_), lds, s, D, n, ld, w, z_old, lam, v, z_new, sigma0, rho,          \
                           pred, slab, partials);                                                                      \
        RBL_HIP(hipGetLastError());                                                                                    \
        return RBL_OK;                                                                                                 \
    } while (0)
    // Rows per sub-batch: as many as the 256 registers of a wave (8 waves per workgroup, one workgroup per CU) hold in two
    // buffers beside the column sums - the pass is bound by the bytes each wave keeps in flight.  fp32 storage, 30-32 GB
    // per pass, it/s (profiles/r03_sweep_shapes.txt):  d = 4000: 4 rows 177.5, 6 rows 196.5, 8 rows 204.1;  d = 6000:
    // 4 rows 203.2, 5 rows 208.0, 6 rows 210.6;  d = 8000: 2 rows 178.7, 3 rows 200.4, 4 rows 209.8;  d = 10 000: 1 row
    // 80.0, 2 rows 124.6, 3 rows 132.3 (4 rows spill: 68);  d = 12 000: 1 row 130.7, 2 rows 200.6;  d = 16 000: 1 row
    // 147.1 (two spill).  RBL_WIDE_SHAPE=2: round 2's shapes.
    static const int wshape = [] {
        const char* e = getenv("RBL_WIDE_SHAPE");
        return e ? atoi(e) : 0;
    }();
    if (pt <= 1) RBL_WIDE(1, 16, 1);   // (fp32 storage reaches this kernel from 9 packets per lane on: pt >= 2)
    if (pt <= 2) {
        if (wshape == 2) RBL_WIDE(2, 4, 4);
        RBL_WIDE(2, 8, 2);
    }
    if (pt <= 3) {
        if (wshape == 2) RBL_WIDE(3, 4, 4);
        RBL_WIDE(3, 6, 4);
    }
    if (pt <= 4) {
        if (wshape == 2) RBL_WIDE(4, 2, 8);
        RBL_WIDE(4, 4, 4);
    }
    if (pt <= 5) {
        if (wshape == 2) RBL_WIDE(5, 2, 8);
        RBL_WIDE(5, 3, 8);
    }
    if (pt <= 6) {
        if (wshape == 2) RBL_WIDE(6, 1, 16);
        RBL_WIDE(6, 2, 8);
    }
    if (pt <= 8) RBL_WIDE(8, 1, 16);   // (two rows spill with fp32 storage)
#undef RBL_WIDE
#undef RBL_WIDE_NT
    rbl_set_error("single-sweep kernel: d=%lld too wide", (long long)ld);
    return RBL_ERR_INVALID;
}

}  // namespace

bool sweep_erm_supported(int storage, int64_t ld) {
    const int64_t PK = ld / (storage == RBL_STORE_F32 ? 4 : 2);
    // wave-per-row kernel up to 4 (fp32) / 8 (fp64) passes of 64 packets, workgroup-per-row
    // kernel up to 8 packets per thread: d <= 16384 (fp32) / 8192 (fp64)
    return PK > 32 && PK <= 8 * SEW_THREADS;
}
// upper bound of the blocks a single-sweep launch uses (slab rows, partial triples): 2 per CU; the shapes that run one
// block per CU (launch_T) use the first half
int sweep_erm_blocks(int num_cu) { return num_cu * 2; }
int sweep_erm_slab_rows(int num_cu) { return sweep_erm_blocks(num_cu) + CR_SLICES; }
// the v-only and q-only passes (one half of the fused pass each) run best with ONE block of 4 waves per CU: interleaved
// on one box C2sq 134.6 -> 137.4 it/s (the fused pass with two rows per sub-batch the other way round, 270 it/s with two
// blocks and 236 with one; with four rows per sub-batch it too runs best with one block, launch_T);
// RBL_SWEEPVQ_BLOCKS_PER_CU=2 for experiments
static int sweep_vq_blocks(int num_cu) {
    static const int per_cu = [] {
        const char* e = getenv("RBL_SWEEPVQ_BLOCKS_PER_CU");
        const int v = e ? atoi(e) : 1;
        return (v == 1 || v == 2) ? v : 1;
    }();
    return num_cu * per_cu;
}

int launch_sweep_erm(int storage, int loss, const void* D, int64_t n, int64_t ld, const double* w, const double* z_old,
                     double* lam, double* v, double* z_new, double sigma0, double rho, const double* pred_dev,
                     double* slab, double* partials, double* q, double* red, double* zz_out, int num_cu, hipStream_t s,
                     hipEvent_t main_done, int want_obj) {
    const int grid = sweep_erm_blocks(num_cu);
    int nrows = grid;   // slab rows / partial triples produced = blocks launched
    double* const v_for_loss = v;
    if (!want_obj) v = nullptr;   // 8 of the 24 B/row of row-wise stores: only the loss sum reads v after the pass
    int rc;
    if (storage == RBL_STORE_F32) {
        rc = (loss == RBL_LOSS_BCE)
                 ? launch_T<float, 0>((const float*)D, n, ld, w, z_old, lam, v, z_new, sigma0, rho, pred_dev, slab, partials, grid, num_cu, &nrows, s)
                 : launch_T<float, 1>((const float*)D, n, ld, w, z_old, lam, v, z_new, sigma0, rho, pred_dev, slab, partials, grid, num_cu, &nrows, s);
    } else {
        rc = (loss == RBL_LOSS_BCE)
                 ? launch_T<double, 0>((const double*)D, n, ld, w, z_old, lam, v, z_new, sigma0, rho, pred_dev, slab, partials, grid, num_cu, &nrows, s)
                 : launch_T<double, 1>((const double*)D, n, ld, w, z_old, lam, v, z_new, sigma0, rho, pred_dev, slab, partials, grid, num_cu, &nrows, s);
    }
    RBL_TRY(rc);
    if (main_done) RBL_HIP(hipEventRecord(main_done, s));
    double* part = slab + (size_t)grid * ld;   // CR_SLICES rows behind the grid rows of the slab
    hipLaunchKernelGGL(k_colreduce2, dim3((unsigned)((ld + 63) / 64), CR_SLICES), dim3(256), 0, s, slab, nrows, (long long)ld,
                       part);
    hipLaunchKernelGGL(k_finish_sweep, dim3(1), dim3(256), 0, s, part, (long long)ld, q, partials, nrows, red, zz_out);
    RBL_HIP(hipGetLastError());
    // objective.py:11-24: the per-sample losses are summed from v in a pass of their own (8 B per
    // row) - exp/log1p inside the sweep cost registers on its critical path
    if (want_obj) RBL_TRY(launch_loss_sum(loss, n, v_for_loss, 1.0, partials, red + 1, s));
    return RBL_OK;
}

namespace {
// red[0] = sum primal^2, red[1] = 0 (k_loss_sum fills it when wanted)
__global__ __launch_bounds__(256) void k_finish_v(const double* __restrict__ partials, int nb, double* __restrict__ red) {
    __shared__ double smem[4];
    double a[1] = {0.0};
    for (int b = threadIdx.x; b < nb; b += 256) a[0] += partials[b * 3];
    rbl::block_sum<1, 256>(a, smem);
    if (threadIdx.x == 0) {
        red[0] = a[0];
        red[1] = 0.0;
    }
}

template <typename T>
int launch_v_T(const T* D, long long n, long long ld, const double* w, const double* z, double* lam, double* v, double rho,
               double* partials, int grid, hipStream_t s) {
    const long long PK = ld / Pk<T>::E;
    const long long passes = (PK + 63) / 64;
#define RBL_V(P_, R_, S_) RBL_V2(P_, R_, S_, false, 2)
#define RBL_V2(P_, R_, S_, WL_, OCC_)                                                                               \
    do {                                                                                                            \
        hipLaunchKernelGGL((k_sweep_erm<T, 1, P_, R_, S_, WL_, SE_VONLY, true, OCC_>), dim3(grid), dim3(SE_THREADS), 0, s, D, n, \
                           ld, w, z, lam, v, (double*)nullptr, 0.0, rho, (const double*)nullptr, (double*)nullptr,  \
                           partials);                                                                               \
        RBL_HIP(hipGetLastError());                                                                                 \
        return RBL_OK;                                                                                              \
    } while (0)
    if (passes == 1) RBL_V(1, 8, 2);
    if (passes == 2) RBL_V(2, 4, 4);
    if (passes <= 4) RBL_V(4, 2, 8);   // (four rows per sub-batch: no gain here, 3.50 against 3.46-3.50 ms at 6M x 1000)
    // 5-8 packets per lane (fp64 d <= 1024, fp32 d <= 2048): two rows per sub-batch, w in LDS, the one block per CU may
    // use up to 512 registers per lane
    if (passes <= 8) RBL_V2(8, 2, 8, true, 1);
#undef RBL_V
#undef RBL_V2
    return RBL_ERR_INVALID;
}
}  // namespace

// rank-weighted problems: v = D w, lambda += rho (z - v), red[0] = sum (z - v)^2 in one pass
// (replaces k_gemv + k_dual when the row width fits the wave-per-row kernel)
bool sweep_v_supported(int storage, int64_t ld) {
    const int64_t PK = ld / (storage == RBL_STORE_F32 ? 4 : 2);
    return PK > 32 && PK <= 512;
}

int launch_sweep_v(int storage, const void* D, int64_t n, int64_t ld, const double* w, const double* z, double* lam,
                   double* v, double rho, double* partials, double* red, int num_cu, hipStream_t s, hipEvent_t main_done) {
    const int grid = sweep_vq_blocks(num_cu);
    if (storage == RBL_STORE_F32)
        RBL_TRY(launch_v_T<float>((const float*)D, n, ld, w, z, lam, v, rho, partials, grid, s));
    else
        RBL_TRY(launch_v_T<double>((const double*)D, n, ld, w, z, lam, v, rho, partials, grid, s));
    if (main_done) RBL_HIP(hipEventRecord(main_done, s));
    hipLaunchKernelGGL(k_finish_v, dim3(1), dim3(256), 0, s, partials, grid, red);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

namespace {
__global__ __launch_bounds__(256) void k_finish_q(const double* __restrict__ part, long long ld, double* __restrict__ q) {
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < ld; j += (long long)gridDim.x * 256) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < CR_SLICES; ++k) s += part[(long long)k * ld + j];
        q[j] = s;
    }
}

template <typename T>
int launch_q_T(const T* D, long long n, long long ld, const double* c, double* slab, int grid, hipStream_t s) {
    const long long PK = ld / Pk<T>::E;
    const long long passes = (PK + 63) / 64;
#define RBL_Q(P_, R_, S_) RBL_Q2(P_, R_, S_, 2)
#define RBL_Q2(P_, R_, S_, OCC_)                                                                                     \
    do {                                                                                                             \
        hipLaunchKernelGGL((k_sweep_erm<T, 1, P_, R_, S_, false, SE_QONLY, true, OCC_>), dim3(grid), dim3(SE_THREADS), 0, s, D, n,  \
                           ld, (const double*)nullptr, c, (double*)nullptr, (double*)nullptr, (double*)nullptr, 0.0, 1.0,     \
                           (const double*)nullptr, slab, (double*)nullptr);                                          \
        RBL_HIP(hipGetLastError());                                                                                  \
        return RBL_OK;                                                                                               \
    } while (0)
    if (passes == 1) RBL_Q(1, 8, 2);
    if (passes == 2) RBL_Q(2, 4, 4);
    if (passes <= 4) RBL_Q(4, 2, 8);
    if (passes <= 8) RBL_Q2(8, 2, 8, 1);
#undef RBL_Q
#undef RBL_Q2
    return RBL_ERR_INVALID;
}
}  // namespace

// q = D^T c through the single-sweep kernel's loads (SE_QONLY); same widths as launch_sweep_v
bool sweep_q_supported(int storage, int64_t ld) {
    static const bool on = [] {
        const char* e = getenv("RBL_GEMVT_SWEEP");     // =0: k_gemvt everywhere (round 2), for comparison
        return !(e && e[0] == '0');
    }();
    return on && sweep_v_supported(storage, ld);
}

int launch_sweep_q(int storage, const void* D, int64_t n, int64_t ld, const double* c, double* slab, double* q, int num_cu,
                   hipStream_t s, hipEvent_t main_done) {
    const int grid = sweep_vq_blocks(num_cu);
    if (storage == RBL_STORE_F32)
        RBL_TRY(launch_q_T<float>((const float*)D, n, ld, c, slab, grid, s));
    else
        RBL_TRY(launch_q_T<double>((const double*)D, n, ld, c, slab, grid, s));
    if (main_done) RBL_HIP(hipEventRecord(main_done, s));
    double* part = slab + (size_t)grid * ld;
    hipLaunchKernelGGL(k_colreduce2, dim3((unsigned)((ld + 63) / 64), CR_SLICES), dim3(256), 0, s, slab, grid, (long long)ld, part);
    hipLaunchKernelGGL(k_finish_q, dim3((unsigned)((ld + 255) / 256)), dim3(256), 0, s, part, (long long)ld, q);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int launch_predict_rho(int64_t ld, const double* q, const double* p, double* p_out, const double* w, const double* w_prev,
                       const double* Gw, const double* zz, double rho, double cap, double* pred, double* wstats,
                       hipStream_t s, const double* rho_dev) {
    hipLaunchKernelGGL(k_predict_rho, dim3(1), dim3(1024), 0, s, (long long)ld, q, p, p_out, w, w_prev, Gw, zz, rho, rho_dev,
                       cap, pred, wstats);
    RBL_HIP(hipGetLastError());
    return RBL_OK;
}

int launch_sumsq(int64_t n, const double* x, double* partials, double* out, hipStream_t s) {
    const int nb = reduce_blocks();
    hipLaunchKernelGGL(k_sumsq, dim3(nb), dim3(256), 0, s, (long long)n, x, partials);
    RBL_HIP(hipGetLastError());
    return launch_sum_partials(partials, nb, 1, out, s);
}
