// rbl_internal.h - shared declarations of librbl.so (gfx950 only, no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>
#include "../../include/rbl.h"

void rbl_set_error(const char* fmt, ...);

// Host wait for a word in pinned memory that a kernel writes last (after __threadfence_system):
// spins on the word; after 2 s without a change the stream is waited for instead, so that a
// failed launch cannot hang the host.
void rbl_spin_wait(const volatile int* word, int sentinel, hipStream_t stream);
// every host wait for the device inside an iteration is counted (rbl_stats.host_syncs); rbl_spin_wait counts itself
void rbl_note_host_sync();

#define RBL_HIP(x)                                                                         \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            rbl_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            return RBL_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define RBL_TRY(x)                 \
    do {                           \
        int r_ = (x);              \
        if (r_ != RBL_OK) return r_; \
    } while (0)

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int RBL_WAVE = 64;
constexpr int PAV_CHUNK_LOG = 10;              // prefix-sum chunk = 1024 sorted positions
constexpr int PAV_CHUNK = 1 << PAV_CHUNK_LOG;

// ---- two-level prefix sums over sorted positions (pav.hip) -------------------------
// P(i) = sum_{j<i} x_j = (cph[i>>10] + cpl[i>>10]) + locx[i]; locx restarts at every
// 1024-chunk so that small-block sums do not cancel against n-sized prefixes.
struct Prefix {
    const double* locx;  // n+1 entries, exclusive prefix inside the chunk
    const double* cph;   // chunk-level exclusive prefix, double-double high part
    const double* cpl;   //                               double-double low part
};

struct SeamRec {
    long long s;  // first pooled position, -1: no merge
    long long e;  // last pooled position
    double x;     // pooled value
};

// ---- sweep.hip ------------------------------------------------------------------------
// v = D w : D is n x ld row-major (ld % 4 == 0, columns >= d are zero)
int launch_gemv(int storage, const void* D, int64_t n, int64_t ld, const double* w, double* v,
                int num_cu, hipStream_t s);
// q = D^T c : partial column sums go to slab (gemvt_slab_rows() x ld doubles), then q
int gemvt_slab_rows(int num_cu);
int launch_gemvt(int storage, const void* D, int64_t n, int64_t ld, const double* c, double* slab,
                 double* q, int num_cu, hipStream_t s, hipEvent_t main_done = nullptr);
// D[r0+i][j] = -y[i] * X[i][j] for a chunk of rows already on the device (fp64 staging)
int launch_form_D(int storage, void* D, int64_t ld, int64_t row0, const double* Xdev, int64_t ldx,
                  const double* ydev, int64_t rows, int64_t d, hipStream_t s);
int launch_D_to_f64(int storage, const void* D, int64_t ld, int64_t n, int64_t d, double* out,
                    hipStream_t s);
// column sums / sums of squares (slab-reduced, deterministic) and in-place standardisation
int launch_colstats(int storage, const void* D, int64_t n, int64_t ld, double* slab, double* sum,
                    double* sumsq, int num_cu, hipStream_t s);
int launch_standardize_negy(int storage, void* D, int64_t n, int64_t ld, int64_t d, const double* mean,
                            const double* inv_std, const signed char* ysign, hipStream_t s);

// ---- elementwise.hip --------------------------------------------------------------------
int launch_erm_zc(int loss, int64_t n, double sigma0, double rho, const double* v, const double* lam,
                  double* m, double* z, double* c, hipStream_t s);
// m = v - lambda/rho together with the sort's input: keys[i] = order-preserving transform of m[i], idx[i] = i + idx_off
int launch_make_m_keys(int64_t n, double rho, const double* v, const double* lam, double* m, u64* keys, u32* idx,
                       u32 idx_off, hipStream_t s);
int launch_keys_from_m(int64_t n, const double* m, u64* keys, u32* idx, hipStream_t s);
// z-step with 32-bit sort keys (round 3): m and its range; the fixed-point keys; after the sort, sorted m / row ids with
// the runs of equal keys put in (m, row) order (*flag = 1: a run too long - sort 64-bit keys instead)
int s32_range_words();   // u64 words of the range buffer mm
int launch_make_m_range(int64_t n, double rho, const double* v, const double* lam, double* m, u64* mm, hipStream_t s);
int launch_keys32(int64_t n, const double* m, const u64* mm, u32* keys, u32* idx, u32 idx_off, hipStream_t s);
int launch_sort32_fix(int64_t n, const u32* keys, const u32* ids, const double* m, u32 off, double* ms, u32* ids_out, int* flag,
                      hipStream_t s);
int launch_prox(int loss, int64_t n, const double* sigma, double rho, const double* m, double* out,
                hipStream_t s);
// lambda += rho (z - v); partial sums {sum (z-v)^2, sum loss(v)} -> red[0..1]
int launch_dual(int loss, int64_t n, double rho, const double* z, const double* v, double* lam,
                double* partials, double* red, hipStream_t s);
int launch_accuracy(int loss, int64_t n, const double* v, const signed char* ysign, double tau, double* partials,
                    double* out, hipStream_t s);
int launch_fair_counts(int64_t n, const double* v, const signed char* ysign, const double* group, double threshold,
                       double* partials, double* out14, hipStream_t s);
int launch_weights(int wf, int64_t n, const double* args, double* alphas, double* betas, hipStream_t s);
// generic deterministic two-stage reduction helpers
int reduce_blocks();
int launch_sum_partials(const double* partials, int nblocks, int K, double* out, hipStream_t s);
// losses of v (for the objective), optionally as sortable keys
int launch_loss_keys(int64_t n, const double* v, u64* keys, hipStream_t s);
int launch_sorted_loss_dot(int loss, int64_t n, const u64* sorted_keys, const double* sigma,
                           double* partials, double* out, hipStream_t s);
int launch_loss_sum(int loss, int64_t n, const double* v, double scale, double* partials, double* out,
                    hipStream_t s);

// ---- sort.hip -------------------------------------------------------------------------
struct SortWorkspace {
    u64* keys[2];
    u32* vals[2];
    u32* spine;      // 256 * RS_MAX_BLOCKS
    u32* bin_total;  // 256
    u32* bin_base;   // 256
    u32* ghist;      // the spine scan's done-counter (zero between launches)
};
size_t sort_spine_bytes();
size_t sort_ghist_bytes();
// sorts keys[0]/vals[0] ascending (stable); result ends in keys[0]/vals[0]
// key_bits: number of low key bits that can differ (digits above are skipped)
int launch_radix_sort(SortWorkspace& ws, int64_t n, bool with_vals, hipStream_t s, int key_bits = 64);
int launch_radix_sort32(SortWorkspace& ws, int64_t n, hipStream_t s);

// ---- pav.hip ----------------------------------------------------------------------------
// ex (round 3, optional): the upper levels in one persistent launch (bar / big / num_cu) and, for a single-handle EHRM
// z-step, the branch speculated from the previous iteration with the singleton-stage sums formed inside the bottom
// kernel (fpart / spec / B; `branch` is then WRITTEN by the exact test before the upper levels read it).
struct PavExtras {
    unsigned* bar;      // pav_bar_uints() counters, zeroed at allocation
    int bar_parity;     // flips per launch
    SeamRec* big;       // pav_big_recs() entries: long pooled ranges all blocks fill together
    int num_cu;
    double* fpart;      // pav_fpart_doubles(n): per-tile shares of the two sums (NULL: no speculation)
    int spec;           // speculated branch (0 = a, 1 = b)
    double B;
};
size_t pav_bar_uints();
int rbl_live_handles(int device);   // api.hip: solver handles of this process alive on a device
int64_t pav_big_recs();
int64_t pav_fpart_doubles(int64_t n);
struct PavWorkspace {
    double* ms;        // n   sorted m
    double* u;         // n   current block values by sorted position
    double* locx_m;    // n+1
    double* chunk_m;   // nchunks
    double* cph_m;     // nchunks
    double* cpl_m;     // nchunks
    SeamRec* recs;     // seams of the upper levels
    u32* counters;     // [0] merges, [1] dirty upper levels, [2] long fills, [3] status of the persistent upper-level kernel
    double* partials;  // reduce scratch
    int* branch;       // EHRM branch flag on the device (0 = a, 1 = b)
    PavExtras ex;      // round-3 paths of launch_pav_tree (buffers owned by this workspace)
};
int64_t pav_num_chunks(int64_t n);
int64_t pav_num_recs(int64_t n);
int launch_prefix(const double* x, int64_t n, double* locx, double* chunk_tot, double* cph, double* cpl,
                  hipStream_t s);
int launch_unflip_keys(int64_t n, const u64* keys, double* ms, hipStream_t s);
int launch_unflip_prefix(const u64* keys, int64_t n, double* ms, double* locx, double* chunk_tot, double* cph, double* cpl,
                         hipStream_t s);
// EHRM: scalar branch test (PAV_cpt.py:205-226) -> *branch
// u0a / u0b (optional): the element prox of both branches is kept for launch_pav_tree
int launch_ehrm_branch(int64_t n, const double* sa, const double* sb, double B, double rho, const double* ms,
                       double* partials, int* branch, int forced, hipStream_t s, double* u0a = nullptr,
                       double* u0b = nullptr);
// element prox (level 0) + merge tree -> u.  sigma = sa, or sb when *branch != 0 (EHRM); the
// matching prefix sums are pa / pb.
// u0a / u0b != NULL: level 0 (element prox of branch a / b) was computed already; u0a may alias u
int launch_pav_tree(int loss, int64_t n, double rho, const double* ms, const double* sa, const double* sb, double* u,
                    Prefix pa, Prefix pb, Prefix pm, const int* branch, SeamRec* recs, u32* merge_counter,
                    hipStream_t s, const double* u0a = nullptr, const double* u0b = nullptr, PavExtras* ex = nullptr);
// z[perm[i]] = clip(u[i]); c[perm[i]] = z + lam[perm[i]]/rho  (local slice [off, off+nloc))
// ---- distributed z-step (merge tree over ranks; pav.hip, CPU restatement oracle/zdist.py)
struct ZdSeam {
    int active, k, side, a0, b1;   // this rank's seam at the current level: id, 0 = left / 1 = right group, rank range
    long long lo, hi, n;           // undecided index range [lo, hi) of the chunk (n positions)
    double x, cnt;                 // pooled block: value and length
};
int launch_ehrm_fvals(int64_t n, const double* sa, const double* sb, double B, double rho, const double* ms,
                      double* partials, double* out2, hipStream_t s, double* u0a = nullptr, double* u0b = nullptr);
int launch_ehrm_pick(const double* fvals_total, int* branch, hipStream_t s);
int launch_add_u32(int64_t n, u32* x, u32 add, hipStream_t s);
int launch_make_c(int64_t n, const double* z, const double* lam, double rho, double* c, hipStream_t s);
int launch_zd_sample(const u64* keys, int64_t n, int ns, double* out, hipStream_t s);
int launch_zd_split_bounds(const u64* keys, int64_t n, const double* split, int nsplit, long long* bounds, hipStream_t s);
int launch_zd_counts_from_bounds(const long long* bounds, int nparts, int64_t n, long long* counts, hipStream_t s);
int launch_zd_bounds(const double* u, int64_t n, double* out3, hipStream_t s);
int launch_zd_seam_setup(int rank, int world, int level, const double* bounds_all, int64_t n, ZdSeam* st, hipStream_t s);
int launch_zd_update_propose(int loss, ZdSeam* st, const double* u, int K, int world, const double* cand_prev,
                             const double* part_prev, double rho, double* cand_out, hipStream_t s);
int launch_zd_eval(const ZdSeam* st, const double* u, Prefix pa, Prefix pb, Prefix pm, const int* branch, int K, int world,
                   const double* cand_all, double* part, hipStream_t s);
int launch_zd_pooled(const ZdSeam* st, Prefix pa, Prefix pb, Prefix pm, const int* branch, int nseams, double* sums, int* err,
                     hipStream_t s);
int launch_zd_fill(int loss, ZdSeam* st, const double* sums_total, double rho, double* u, int64_t n, hipStream_t s);
int launch_zd_ids_to_keys(int64_t n, const u32* ids, u64* keys, u32* pos, hipStream_t s);
int launch_zd_gather_back(int64_t n, const u64* keys_sorted, const u32* pos, const double* u, u32* out_ids, double* out_u,
                          hipStream_t s);
int launch_zd_owner_bounds(const u64* keys_sorted, int64_t n, int64_t nmax, int world, long long* bounds, hipStream_t s);
int launch_zd_scatter(int64_t n, const u32* ids, const double* uu, const int* branch, double B, int has_B, double rho,
                      const double* lam, double* z, double* c, int64_t off, int64_t nloc, hipStream_t s);
int launch_scatter_z(int64_t n, const double* u, const u32* perm, const int* branch, double B, int has_B,
                     double rho, const double* lam, double* z, double* c, int64_t off, int64_t nloc,
                     hipStream_t s);

// ---- wstep.hip --------------------------------------------------------------------------
struct WstepWorkspace {
    double* yk;     // d
    double* Gy;     // d
    double* wn;     // d
    double* r;      // d (CG)
    double* p;      // d (CG)
    double* scal;   // small device scalars: [0]=t, [1]=rr, ...
    int* flags;     // [0]=done, [1]=iters
    int* pin;       // host-pinned, device-visible: [0..3] = status block of the active-set lasso kernel, [4..5] = CG (done, iterations)
    // l2 w-step: G = V diag(lambda) V^T computed once by rbl_gram_finish (eig.hip); NULL / false: CG
    double *eig_Vt, *eig_V, *eig_lambda;
    bool eig_ok;
    int last_iters; // CG iterations of the previous w-step (sizes the next batch)
    int last_fista; // the same for FISTA; pin[6..7] = its (done, iterations)
    // persistent one-launch w-steps (wstep.hip: k_cg_persist / k_ncg_persist): two sets of barrier counters used
    // alternately (a launch clears the set of the next one), pin[4..5] / pin[8..9] = their (status, iterations)
    unsigned* bar;  // [0]: abort word of the persistent kernels (a block that gave up waiting)
    double* xch;    // the two exchange buffers of the matrix-vector products: 8-byte granules {tag | 32 bits}, two per double
    int launch_seq; // launches so far (the tags of a launch: launch_seq << 12 | exchange number)
    bool gw_valid;  // ws.Gy holds G w of the w the last run_wstep returned
    int form;       // how the last run_wstep ran (rbl_stats.wstep_form)
    int ncg_skip, ncg_backoff;   // persistent nonlinear CG: w-steps left without / length of the pause of its linear first phase
};
constexpr int WSTEP_BAR_UINTS = 2 * 10 * 32;
constexpr int WSTEP_PERSIST_MAX_LD = 2048;                                  // 8 vector elements per thread of a 256-thread block
constexpr int WSTEP_XCH_GRANULES = 2 * (WSTEP_PERSIST_MAX_LD + 16);        // granules of one exchange buffer
constexpr int WSTEP_XCH_DOUBLES = 2 * WSTEP_XCH_GRANULES + 16;            // two buffers (+ debug stamps at the end)
int launch_power_iteration(const double* G, int64_t d, double* tmp1, double* tmp2, double* scal, int iters,
                           double* lambda_host, hipStream_t s);
// lasso / smoothed-l1 by FISTA with restart, ridge by CG; w is updated in place.
// If fs_pending != NULL and the w-step is the lasso, only the active-set kernel is enqueued and
// *fs_pending = true is returned: the caller may enqueue work that assumes success, then calls
// finish_wstep_l1(), which waits for the kernel's status and runs FISTA when it did not converge
// (*fell_back = true: w changed again, work enqueued in between must be redone).
int run_wstep(int wstep, const double* G, int64_t d, const double* q, double rho, double reg, double smooth_t,
              double L, double tol, int max_inner, double* w, WstepWorkspace& ws, int* iters_host,
              hipStream_t s, bool* fs_pending = nullptr, const double* rho_dev = nullptr,
              double* w_prev_out = nullptr, bool want_Gw = false);   // want_Gw: the lasso kernel leaves G w in ws.Gy
int finish_wstep_l1(const double* G, int64_t d, const double* q, double rho, double reg, double L, double tol,
                    int max_inner, double* w, WstepWorkspace& ws, int* iters_host, hipStream_t s, bool* fell_back);
int launch_w_stats(int64_t d, const double* w, const double* w_prev, double* out3, hipStream_t s);
// lasso_fs.hip: exact active-set (feature-sign) lasso in one workgroup; out_dev = 4 ints
// rho_dev != NULL: kappa = reg / (2 rho_dev[0]) is formed on the device; w_prev_out != NULL: the warm start is saved there
int launch_lasso_fs(const double* G, int64_t ld, int64_t d, const double* q, double* w, double kappa, int* out_dev,
                    hipStream_t s, const double* rho_dev = nullptr, double reg = 0.0, double* w_prev_out = nullptr,
                    double* Gw_out = nullptr);   // Gw_out: G w of the solution (valid when the status is 0)
int launch_reg_terms(int64_t d, const double* w, double* out2 /* [sum w^2, sum |w|] */, hipStream_t s);
int launch_soft_threshold(int64_t d, double* w, double t, hipStream_t s);

// ---- sweep_erm.hip: one pass over D per iteration for erm (fused dual update + next z-step + D^T c)
bool sweep_erm_supported(int storage, int64_t ld);
int sweep_erm_blocks(int num_cu);
int launch_sweep_erm(int storage, int loss, const void* D, int64_t n, int64_t ld, const double* w, const double* z_old,
                     double* lam, double* v, double* z_new, double sigma0, double rho, const double* pred_dev,
                     double* slab, double* partials, double* q, double* red, double* zz_out, int num_cu, hipStream_t s,
                     hipEvent_t main_done, int want_obj);
// rho_{k+1} prediction + the w statistics (||w - w_prev||^2, sum w^2, ||w||_1 -> wstats[0..2]); p_out != p
// rho_dev != NULL: rho is read from rho_dev[0] on the device (may alias pred)
int launch_predict_rho(int64_t ld, const double* q, const double* p, double* p_out, const double* w, const double* w_prev,
                       const double* Gw, const double* zz, double rho, double cap, double* pred, double* wstats,
                       hipStream_t s, const double* rho_dev = nullptr);
int launch_sumsq(int64_t n, const double* x, double* partials, double* out, hipStream_t s);
// v = D w, lambda += rho (z - v), red[0] = sum (z - v)^2 in one pass (rows up to 4 / 8 passes of 64 packets)
bool sweep_v_supported(int storage, int64_t ld);
// q = D^T c with the single-sweep kernel's row-streaming loads (sweep_erm.hip: SE_QONLY); launch_gemvt routes to it
bool sweep_q_supported(int storage, int64_t ld);
int launch_sweep_q(int storage, const void* D, int64_t n, int64_t ld, const double* c, double* slab, double* q, int num_cu,
                   hipStream_t s, hipEvent_t main_done);
int launch_sweep_v(int storage, const void* D, int64_t n, int64_t ld, const double* w, const double* z, double* lam,
                   double* v, double rho, double* partials, double* red, int num_cu, hipStream_t s, hipEvent_t main_done);
int launch_symv(const double* G, int64_t ld, const double* x, double* y, hipStream_t s);
int launch_symv_ab(const double* G, int64_t ld, const double* x, double* y, double alpha, double beta, hipStream_t s);   // y = alpha G x + beta x

// ---- eig.hip: one-time eigendecomposition of G for the l2 w-step
int launch_eig_jacobi(const double* G, int64_t ld, int64_t d, double* Bt, double* Vt, double* V, double* lambda,
                      unsigned long long* offmax_dev, hipStream_t s, int* sweeps_out);
int launch_ridge_eig(const double* G, const double* Vt, const double* V, const double* lambda, int64_t ld, const double* q,
                     double rho, double reg, double* w, double* tmp1, double* tmp2, hipStream_t s);

// ---- gram.hip ---------------------------------------------------------------------------
size_t gram_slab_bytes(int64_t d, int num_cu, int64_t n);
int launch_gram(int storage, const void* D, int64_t n, int64_t ld, int64_t d, double* slab, double* G,
                int num_cu, hipStream_t s);

// ---- synth.hip --------------------------------------------------------------------------
int launch_synth(int storage, void* D, int64_t n, int64_t ld, int64_t d, int64_t row_offset, u64 seed,
                 double class_sep, double flip_y, const int* special, const double* mix, const double* A16,
                 const int* vertex, signed char* ysign, hipStream_t s);

// ---- zband.hip: z-step for piecewise-constant rank weights without a sort -------------------
constexpr int ZB_BITS = 11;          // radix-select digit (last pass: the remaining 9 bits)
constexpr int ZB_C = 16;             // candidate block values per root pass
constexpr int ZB_ROOT_PASSES = 4;    // bracket shrinks 15x per pass (the first one far more when the prediction is good)
constexpr int ZB_GCAP = 2048;        // undecided elements the finishing kernel settles exactly
constexpr int ZB_MAX_BANDS = 8;
constexpr int ZB_MAX_TARGETS = 12;
constexpr int ZB_MAX_GROUPS = 6;     // distinct key prefixes among the targets in one select pass
constexpr int ZB_MAX_CLUSTERS = 4;
enum { ZB_OK = 0, ZB_TIE = 1, ZB_BAD = 2, ZB_GROUPS = 3, ZB_BRACKET = 4, ZB_UNRESOLVED = 5, ZB_SWALLOW_L = 6, ZB_SWALLOW_R = 7,
       ZB_ONESIDED = 8, ZB_OVERLAP = 9 };

struct ZbConfig {                        // built once from sigma (host), passed to the kernels by value
    int nbands;
    long long start[ZB_MAX_BANDS + 1];   // first rank of band j; start[nbands] = n
    double sigma[ZB_MAX_BANDS];
    int ntargets;
    long long target_rank[ZB_MAX_TARGETS];   // ascending, unique
    int last_t[ZB_MAX_BANDS];            // target index of band j's last rank  (j < nbands - 1)
    int first_t[ZB_MAX_BANDS];           // target index of band j's first rank (j > 0)
    int nclusters;                       // edge clusters: band L | single-rank bands | band R
    int cl_L[ZB_MAX_CLUSTERS], cl_R[ZB_MAX_CLUSTERS];
    int cl_root[ZB_MAX_CLUSTERS];        // sigma increases somewhere along the chain: pooling is possible
};
struct ZbState {                         // device scratch of one z-step
    u64 prefix[ZB_MAX_TARGETS];
    long long rem[ZB_MAX_TARGETS];
    int group[ZB_MAX_TARGETS];
    u64 gprefix[ZB_MAX_TARGETS];
    u64 key[ZB_MAX_TARGETS];
    long long eq[ZB_MAX_TARGETS];         // keys equal to key[t] (rem[t] ends as the target's rank among them)
    int ngroups;
    int status;
    double cand[ZB_MAX_CLUSTERS][ZB_C];
    int done[ZB_MAX_CLUSTERS];
    int has_block[ZB_MAX_CLUSTERS];
    double x[ZB_MAX_CLUSTERS];
    double xh[ZB_MAX_CLUSTERS][3];        // block values of the last three certified iterations (kept across z-steps)
    int nh[ZB_MAX_CLUSTERS];
    double br[ZB_MAX_CLUSTERS][2];        // bracket of the root after the last pass
    double frozen[ZB_MAX_CLUSTERS][4];    // outside the bracket for certain: top count / sum m, bottom count / sum m
    double und[ZB_MAX_CLUSTERS];          // elements whose membership changes inside the bracket
    int gcount[ZB_MAX_CLUSTERS];          // gathered so far
};
size_t zb_hist_bytes();
size_t zb_partials_bytes();
int launch_zb_edges(const double* sigma, int64_t n, long long* pos, int* counter, int cap, hipStream_t s);
// z = the z-step, c = z + lambda/rho in the same pass
int launch_zband(int loss, const ZbConfig& cfg, int64_t n, double rho, const u64* keys, const double* m, double* z,
                 const double* lam, double* c, ZbState* st, u32* hist, double* partials, int* pin, int seq, u32* counters, hipStream_t s);
int launch_zband_risk(int loss, const ZbConfig& cfg, int64_t n, const u64* keys, ZbState* st, u32* hist, double* partials,
                      double* out_dev, hipStream_t s);
// the steps of launch_zband one by one (multi-GPU driver: collectives in between)
int launch_zbd_init(const ZbConfig& cfg, ZbState* st, u32* hist, hipStream_t s);
int launch_zbd_hist(int64_t n, const u64* keys, ZbState* st, u32* hist, int pass, hipStream_t s);
int launch_zbd_scan(int loss, const ZbConfig& cfg, ZbState* st, u32* hist, int pass, double rho, hipStream_t s);
int launch_zbd_eval(int loss, const ZbConfig& cfg, int64_t n, const u64* keys, ZbState* st, int k, double rho, double* partials,
                    double* tot, hipStream_t s);
int launch_zbd_decide(int loss, const ZbConfig& cfg, ZbState* st, int k, double rho, const double* tot, int last, hipStream_t s,
                      int* pin_settled = nullptr, int dseq = 0);
int launch_zbd_gather(int loss, const ZbConfig& cfg, int64_t n, const u64* keys, ZbState* st, int k, double rho, double* partials,
                      double* pack, hipStream_t s);
int launch_zbd_finish(int loss, const ZbConfig& cfg, ZbState* st, int k, double rho, double* partials, const double* packs_all,
                      int world, hipStream_t s);
int launch_zbd_apply(int loss, const ZbConfig& cfg, int64_t n, double rho, const double* m, double* z, const double* lam, double* c,
                     ZbState* st, int* pin, int seq, u32* counters, hipStream_t s);
