/* rbl.h - C ABI of librbl.so: the ADMM inner iteration for rank-based loss
 * minimisation on AMD Instinct MI355X (gfx950), hand-written HIP.
 *
 * This is the drop-in boundary for the hot path of RufengXiao/ADMM-for-rank-based-loss
 * (reference paths below are relative to that repository).  The reference has no
 * FFI layer - its boundary is the Python class API of src/optim/algorithms.py - so
 * every entry point names the reference interface it stands behind; the Python
 * mirror of that class API (admm-for-rank-based-loss_amd/src/optim/algorithms.py)
 * binds these symbols with ctypes (see INTEGRATION.md for the stub).
 *
 * Conventions: plain pointers and sizes only; every function returns an int status
 * (0 = ok, <0 = error, message via rbl_last_error()); no exceptions cross the
 * boundary; the library owns all device memory; the caller owns all host buffers;
 * no callbacks.  One solver handle per host thread.  Host arrays are row-major
 * float64 unless stated.  There is NO CPU fallback: every compute entry point
 * fails with RBL_ERR_NO_DEVICE when no gfx950 device is present.
 */
#ifndef RBL_H
#define RBL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RBL_VERSION 106

/* status codes */
enum {
    RBL_OK = 0,
    RBL_ERR_INVALID = -1,    /* bad argument / bad configuration (Python side raises ValueError) */
    RBL_ERR_NO_DEVICE = -2,  /* no HIP device: the product path has no CPU fallback */
    RBL_ERR_HIP = -3,        /* a HIP runtime call failed */
    RBL_ERR_STATE = -4,      /* call order violated (e.g. step before set_data) */
    RBL_ERR_NOMEM = -5
};

/* loss: src/optim/objective.py:27-37 (get_loss); the ADMM z-step supports these two
 * (src/util/individual_solver.py:112-123) */
enum { RBL_LOSS_BCE = 0, RBL_LOSS_HINGE = 1 };

/* weight_function: src/optim/objective.py:166-187 (get_weights) */
enum {
    RBL_W_ERM = 0, RBL_W_EXTREMILE = 1, RBL_W_SUPERQUANTILE = 2, RBL_W_ESRM = 3,
    RBL_W_AORR = 4, RBL_W_AORR_DC = 5, RBL_W_EHRM = 6
};

/* w-step flavour: src/optim/algorithms.py:57-60,190-207 (ADMMmethod) and :238-246
 * (smoothADMMmethod) */
enum { RBL_WSTEP_L1 = 1, RBL_WSTEP_L2 = 2, RBL_WSTEP_SMOOTH_L1 = 3 };

/* element type D = -y*X is stored in (accumulation is always float64) */
enum { RBL_STORE_F32 = 0, RBL_STORE_F64 = 1 };

/* Constructor arguments of Optimizer.__init__ (src/optim/algorithms.py:20-75). */
typedef struct rbl_config {
    int64_t n;              /* rows held by THIS process (its shard of the sample axis) */
    int64_t d;              /* features */
    int64_t n_total;        /* rows of the whole problem (== n on one GPU) */
    int64_t row_offset;     /* global index of this shard's first row */
    int32_t loss;           /* RBL_LOSS_* */
    int32_t weight_function;/* RBL_W_* */
    double  weight_args[2]; /* args list (objective.py:174-183); unused entries 0 */
    int32_t n_weight_args;  /* 0 = args is None */
    int32_t has_B;          /* B is not None (ehrm only, algorithms.py:64-68) */
    double  B;
    int32_t wstep;          /* RBL_WSTEP_* */
    double  reg;            /* l1_reg or l2_reg (algorithms.py:30) */
    double  smooth_t;       /* smoothADMMmethod t (algorithms.py:225,228) */
    double  rho0;           /* <= 0: reference default by weight_function (algorithms.py:47-52) */
    double  tol;            /* stop tolerance (algorithms.py:44,137; reference default 1e-4); taken literally:
                               tol <= 0 never reports convergence (a fixed number of iterations) */
    double  w_tol;          /* inner w-step tolerance; <= 0: library default 1e-13 */
    int32_t max_iter;       /* algorithms.py:45 */
    int32_t storage;        /* RBL_STORE_* */
    int32_t device;         /* HIP device ordinal */
    int32_t objective_only; /* 1: handle used only for rbl_objective (rankbasedObjective) */
} rbl_config;

/* Per-iteration report of Optimizer.main_loop (src/optim/algorithms.py:119-164). */
typedef struct rbl_stats {
    int64_t iter;            /* iterations completed */
    double  primal;          /* ||z - D w||_2           (algorithms.py:135) */
    double  dual;            /* ||w - w_prev||_2        (algorithms.py:136) */
    double  rho;             /* rho used in this iteration */
    double  rho_next;        /* rho after the schedule  (algorithms.py:154-157) */
    double  objective;       /* F(w) after the iteration (objective.py:71-87); NaN if not computed */
    int32_t converged;       /* both residuals < tol    (algorithms.py:137) */
    int32_t inner_iters;     /* w-step inner iterations */
    int32_t ehrm_branch;     /* 0 = a (z<=B), 1 = b (z>=B), -1 = n/a (PAV_cpt.py:222-226) */
    int32_t pav_merges;      /* seam merges performed by the PAV tree, -1 = n/a */
    float   ms_z, ms_q, ms_w, ms_v, ms_total;  /* device time of the phases, HIP events (rbl_profile_kernels level 2) */
    int32_t fused;           /* 1: this iteration's dual update ran in the single-sweep erm kernel */
    int32_t mispredicted;    /* 1: rho was mispredicted, the next z-step is redone unfused */
    int32_t fused_v;         /* 1: rank-weighted iteration whose v = D w, lambda update and primal residual ran in one
                                pass (the v-only mode of the single-sweep kernel) instead of k_gemv + k_dual */
    int32_t host_syncs;      /* times the host waited for the device inside this iteration's library calls
                                (stream waits, blocking copies, the spin on the pinned statistics block) */
    int32_t sort_passes;     /* radix-sort passes the z-step executed (digits shared by all keys are skipped), -1 = n/a */
    int32_t zband;           /* rank-weighted z-step: 0 = sort + merge-tree PAV, 1 = sort-free banded path (piecewise-constant
                                weights), 2 = banded path not certified, redone with the sort; -1 = n/a */
    int32_t wstep_form;      /* how the w-step ran: 0 = batches of launches (CG / nonlinear CG / FISTA), 1 = ONE persistent
                                launch (k_cg_persist / k_ncg_persist), 2 = the exact active-set lasso kernel, 3 = the
                                eigen-decomposition ridge, -1 = an overridden w_subproblem (rbl_phase_w_external) */
} rbl_stats;

typedef struct rbl_solver rbl_solver;

/* ---- lifetime ------------------------------------------------------------------ */
int  rbl_version(void);
/* sizeof(rbl_config) (which = 0) / sizeof(rbl_stats) (which = 1) as THIS library was compiled: a binding whose
 * structure definitions are older or newer than the library checks them at load time instead of letting rbl_step
 * write past its buffer; -1 for any other `which` */
int  rbl_sizeof(int which);
const char* rbl_last_error(void);
int  rbl_device_count(void);
/* Optimizer.__init__ / rankbasedObjective.__init__ */
int  rbl_create(const rbl_config* cfg, rbl_solver** out);
int  rbl_destroy(rbl_solver* h);
/* run the library's kernels on this hipStream_t: NULL is the (legacy) default stream,
 * (void*)-1 goes back to the handle's own non-blocking stream (the initial setting) */
int  rbl_set_stream(rbl_solver* h, void* hip_stream);

/* ---- data: D = -y * X (algorithms.py:23), G = D^T D (algorithms.py:24) ---------- */
/* X: n x d host rows with leading dimension ldx, y: n labels (+-1). */
int  rbl_set_data(rbl_solver* h, const double* X, const double* y, int64_t ldx);
/* Synthetic two-class data generated on the device (statistics of
 * src/util/load_data.py:101-116), never materialised on the host. */
int  rbl_generate_synthetic(rbl_solver* h, uint64_t seed, double class_sep, double flip_y);
/* sharded form: raw local rows + local column sums (RBL_BUF_COLSTATS, to be summed over
 * ranks), then standardise and scale by -y.  Same matrix for any sharding of the rows. */
int  rbl_synth_local(rbl_solver* h, uint64_t seed, double class_sep, double flip_y);
int  rbl_synth_finish(rbl_solver* h);
/* labels of the generated rows (+-1), n doubles */
int  rbl_get_labels(rbl_solver* h, double* y_out);
/* Build G and the w-step constants.  For multi-GPU runs call rbl_gram_local(), sum
 * the d*d buffer across ranks (rbl_buffer RBL_BUF_G) and then rbl_gram_finish(). */
int  rbl_gram_local(rbl_solver* h);
int  rbl_gram_finish(rbl_solver* h);
int  rbl_get_D(rbl_solver* h, double* out /* n x d */);

/* ---- state: w, z, lambda, rho (algorithms.py:32-52); NULL pointers are skipped ---- */
int  rbl_get_state(rbl_solver* h, double* w, double* z, double* lam, double* rho, int64_t* iter, double* smooth_t);
int  rbl_set_state(rbl_solver* h, const double* w, const double* z, const double* lam, const double* rho, const int64_t* iter, const double* smooth_t);
int  rbl_get_sigma(rbl_solver* h, double* alphas, double* betas /* n_total each */);

/* ---- the hot path ---------------------------------------------------------------- */
/* One ADMM iteration = Optimizer.main_loop(i, ...) (algorithms.py:119-164).
 * want_objective != 0 also evaluates F(w_{k+1}) from the cached D w (the `store`
 * logging of algorithms.py:159-161). */
int  rbl_step(rbl_solver* h, int want_objective, rbl_stats* out);
/* ADMMmethod.main_loop / smoothADMMmethod.main_loop (algorithms.py:209-216, 248-260):
 * up to max_iter steps, stops on convergence; history arrays (may be NULL) receive
 * one entry per iteration, cap entries each. */
int  rbl_solve(rbl_solver* h, int max_iter, int want_objective, rbl_stats* last,
               double* hist_objective, double* hist_primal, double* hist_dual, double* hist_rho,
               double* hist_time_s, int64_t cap);
/* smoothADMMmethod's final soft-threshold of w by t (algorithms.py:257-258) */
int  rbl_finalize_smooth(rbl_solver* h);
/* rankbasedObjective.get_arrogate_loss(w) (objective.py:71-87); w: d host doubles */
int  rbl_objective(rbl_solver* h, const double* w, int include_reg, double* out);

/* calculate_accuracy(w, X, y, threshold, loss) of src/util/calculate_acc.py:3-19 on this handle's
 * rows (hinge mirrors the reference's quirk: every prediction is +1) */
int  rbl_accuracy(rbl_solver* h, const double* w, double threshold, double* out);
/* calculate_statistics(w, X, label, group, threshold) of src/util/fair_metric.py:3-41 on this
 * handle's rows: out6 = {SPD, DI, EOD, AOD, TI, FNRD}; group: n doubles (0 / 1) */
int  rbl_fair_statistics(rbl_solver* h, const double* w, const double* group, double threshold, double* out6);

/* ---- phase API (one process per GPU; the host does the collectives in between) ---- */
/* A: m = D w - lambda/rho for the local rows (algorithms.py:89) -> RBL_BUF_M */
int  rbl_phase_m(rbl_solver* h);
/* B: z-step (algorithms.py:92-104).  m_all_dev: device pointer to the n_total gathered
 * m values (rank order) or NULL when n == n_total or weight_function == erm. */
int  rbl_phase_z(rbl_solver* h, const void* m_all_dev);
/* B': the caller's own z-step (an overridden Optimizer.z_subproblem that returns an array, algorithms.py:88-106 /
 * :186-188): z (n host doubles, this process's rows) replaces the library's; everything the later phases derive
 * from z is rebuilt.  Runs rbl_phase_m first if the iteration has not been opened yet. */
int  rbl_phase_z_external(rbl_solver* h, const double* z);
/* D': the caller's own w-step (an overridden w_subproblem, algorithms.py:109-116): w (d host doubles, identical
 * on every rank) becomes w_{k+1}; the previous iterate is kept for the dual residual.  Call after rbl_phase_q. */
int  rbl_phase_w_external(rbl_solver* h, const double* w);
/* C: local q = D^T (z + lambda/rho) -> RBL_BUF_Q (d doubles, to be summed over ranks) */
int  rbl_phase_q(rbl_solver* h);
/* D: replicated w-step from the summed q (algorithms.py:109-116,190-207) */
int  rbl_phase_w(rbl_solver* h);
/* E: v = D w, lambda += rho (z - v), local partial sums -> RBL_BUF_RED (to be summed) */
int  rbl_phase_dual(rbl_solver* h, int want_objective);
/* F: residual norms, stop test, rho schedule (algorithms.py:135-157) */
int  rbl_phase_finish(rbl_solver* h, rbl_stats* out);

/* device buffers the host may pass to a collective */
enum { RBL_BUF_M = 0, RBL_BUF_Q = 1, RBL_BUF_RED = 2, RBL_BUF_G = 3, RBL_BUF_V = 4, RBL_BUF_Z = 5,
       RBL_BUF_LAM = 6, RBL_BUF_W = 7, RBL_BUF_COLSTATS = 8,
       /* distributed z-step; the count returned is in ELEMENTS of the type given here */
       RBL_BUF_ZD_SKEYS = 16,  /* int64 x n       sorted keys of the local rows (send)            */
       RBL_BUF_ZD_SIDS = 17,   /* int32 x n       their global row ids (send)                      */
       RBL_BUF_ZD_RKEYS = 18,  /* int64 x n_total keys received for the own range (capacity)       */
       RBL_BUF_ZD_RIDS = 19,   /* int32 x n_total row ids received                                 */
       RBL_BUF_ZD_SMALL = 20,  /* double x 16384  samples [0,256) | bounds [256,259) | EHRM sums
                                  [260,262) | candidates [320,384) | partial sums [512,12800) |
                                  seam sums [12800, ...)                                           */
       RBL_BUF_ZD_BIDS = 21,   /* int32 x n_total row ids of the chunk, grouped by owner (send back) */
       RBL_BUF_ZD_BU = 22,     /* double x n_total their block values (send back)                  */
       RBL_BUF_ZD_ZIDS = 23,   /* int32 x n       row ids received back                            */
       RBL_BUF_ZD_ZU = 24,     /* double x n      block values received back                       */
       RBL_BUF_ZD_COUNTS = 25, /* int64 x 64      rows of the local sorted run that go to each rank (rbl_zd_partition) */
       /* sort-free z-step for banded rank weights, sharded rows (rbl_zbd_*) */
       RBL_BUF_ZB_HIST = 26,   /* int32 x 12288   digit histograms of one select pass (to be SUMMED over the ranks)   */
       RBL_BUF_ZB_TOT = 27,    /* double x 64     block sums of one root pass (to be SUMMED over the ranks)           */
       RBL_BUF_ZB_PACK = 28    /* double x 2049   [count | undecided elements] of this rank (to be ALL-GATHERED)       */ };
int  rbl_buffer(rbl_solver* h, int which, void** dev_ptr, int64_t* n_doubles);
/* ---- distributed z-step for rank-weighted problems on several GPUs ---------------------------
 * (no reference counterpart: the reference is single-process; what is distributed is
 * algorithms.py:88-106.  Driver: admm-for-rank-based-loss_amd/dist.py:_z_distributed; CPU
 * restatement of every call: oracle/zdist.py.)  After rbl_phase_m:
 *   rbl_zd_sort_local        sort the local m (payload: global row id); ZD_SMALL[0,nsamples) = regular samples (NaN = none)
 *   rbl_zd_partition         splitters (nparts-1 doubles, device) -> how many sorted rows go to each rank: left in
 *                            RBL_BUF_ZD_COUNTS on the device (all-gathered there: ONE host wait for the whole count
 *                            matrix); send_counts != NULL additionally downloads them (a host wait of its own)
 *   [all-to-all of ZD_SKEYS / ZD_SIDS into ZD_RKEYS / ZD_RIDS]
 *   rbl_zd_prepare           sort the received chunk (n_recv rows, first sorted position sigma_off), prefix sums;
 *                            ZD_SMALL[260,262) = this chunk's EHRM branch sums (to be summed over ranks)
 *   rbl_zd_pav               exact PAV of the chunk (EHRM: branch from the summed values)
 *   per level of the merge tree over ranks:
 *     rbl_zd_bounds          ZD_SMALL[256,259) = (u_first, u_last, count)          [all-gather]
 *     rbl_zd_seam_setup      this rank's seam / side / violation from all bounds
 *     rounds x { rbl_zd_seam_propose -> ZD_SMALL[320,320+K)                         [all-gather]
 *                rbl_zd_seam_eval    -> ZD_SMALL[512,512+3*world*K)                 [all-reduce] }
 *     rbl_zd_seam_sums       ZD_SMALL[12800,12800+3*nseams)                        [all-reduce]
 *     rbl_zd_seam_fill       pooled block value onto this rank's pooled positions
 *   rbl_zd_return_partition  (row id, value) grouped by owner into ZD_BIDS / ZD_BU.  counts == NULL: no host wait - the
 *                            count matrix of the return trip is the transpose of the forward one, which the driver
 *                            holds; counts != NULL downloads them (and the seam-search error flag, otherwise
 *                            reported by rbl_phase_finish)
 *   [all-to-all into ZD_ZIDS / ZD_ZU]
 *   rbl_zd_scatter           z of the local rows; then rbl_phase_q as usual.
 * Logged objective of rank weights (sum_i sigma_i loss_(i), objective.py:73-82) by the same sample sort:
 *   rbl_zd_sort_losses       sort the local per-sample losses (keys only) + samples; rbl_zd_partition;
 *   [all-to-all of ZD_SKEYS into ZD_RKEYS]; rbl_zd_risk -> ZD_SMALL[264] = this chunk's share [all-reduce]. */
int  rbl_zd_sort_local(rbl_solver* h, int nsamples);
int  rbl_zd_partition(rbl_solver* h, const void* splitters_dev, int nparts, int64_t* send_counts);
int  rbl_zd_sort_losses(rbl_solver* h, int nsamples);
int  rbl_zd_risk(rbl_solver* h, int64_t n_recv, int64_t sigma_off);
int  rbl_zd_prepare(rbl_solver* h, int64_t n_recv, int64_t sigma_off);
int  rbl_zd_pav(rbl_solver* h, const void* fvals_total_dev);
int  rbl_zd_bounds(rbl_solver* h);
int  rbl_zd_seam_setup(rbl_solver* h, int rank, int world, int level, const void* bounds_all_dev);
int  rbl_zd_seam_propose(rbl_solver* h, int K, const void* cand_all_prev_dev, const void* part_sum_prev_dev);
int  rbl_zd_seam_eval(rbl_solver* h, int K, const void* cand_all_dev);
int  rbl_zd_seam_sums(rbl_solver* h, int K, const void* cand_all_prev_dev, const void* part_sum_prev_dev, int nseams);
int  rbl_zd_seam_fill(rbl_solver* h, const void* sums_total_dev);
int  rbl_zd_return_partition(rbl_solver* h, int64_t nmax, int world, int64_t* counts);
int  rbl_zd_scatter(rbl_solver* h, int64_t n_back);

/* Distributed z-step WITHOUT a sort for rank weights that are constant on a few bands (superquantile, aorr, aorr_dc;
 * src/optim/objective.py:108-145) - the reference's z_subproblem (algorithms.py:96-104) for row-sharded m.  No sample
 * sort, no all-to-all, no merge tree: the keys at the band edges by a radix select on histograms summed over the ranks,
 * the pooled block's value as the root of the pooled derivative from sums summed over the ranks, the last undecided
 * elements gathered and settled identically on every rank.  After rbl_phase_m:
 *   rbl_zbd_begin     *applicable = 0: not such weights / iteration 0 / pausing after an uncertified step -> rbl_zd_*.
 *                     *root_clusters: bit k set = band edge k can pool.
 *   for pass 0..5:    rbl_zbd_hist(pass); SUM RBL_BUF_ZB_HIST over the ranks; rbl_zbd_scan(pass)
 *   for every set bit k, up to rbl_zbd_root_passes() times:  rbl_zbd_eval(k); SUM RBL_BUF_ZB_TOT;
 *                     rbl_zbd_decide(k, last = final time, &settled) - settled != 0 (the same on every rank: the state is
 *                     a function of the summed totals; one host wait on a pinned word): no further pass for this k.
 *                     In steady state the first pass settles (its candidates sit around a prediction from the last
 *                     block values).  settled = NULL: no host wait, the caller issues all passes (spare ones are idle).
 *                     then rbl_zbd_gather(k); ALL-GATHER RBL_BUF_ZB_PACK; rbl_zbd_finish(k, gathered, world)
 *   rbl_zbd_apply     z and c = z + lambda/rho of the local rows; *status = 0: certified (go on with rbl_phase_q),
 *                     otherwise every rank got the same non-zero status: run rbl_zd_* for this iteration.            */
int  rbl_zbd_begin(rbl_solver* h, int* applicable, int* root_clusters);
int  rbl_zbd_hist(rbl_solver* h, int pass);
int  rbl_zbd_scan(rbl_solver* h, int pass);
int  rbl_zbd_eval(rbl_solver* h, int k);
int  rbl_zbd_decide(rbl_solver* h, int k, int last, int* settled);
int  rbl_zbd_root_passes(void);
int  rbl_zbd_gather(rbl_solver* h, int k);
int  rbl_zbd_finish(rbl_solver* h, int k, const void* packs_all_dev, int world);
int  rbl_zbd_apply(rbl_solver* h, int* status);

/* RBL_BUF_Q is the whole exchange buffer [q (ld) | D^T lambda seed (ld) | ||z||^2 | primal^2 |
 * sum loss]; RBL_BUF_RED is its 2-double tail.  After rbl_phase_q and after rbl_phase_dual this
 * tells which part awaits the sum over ranks: bit 0 = the first 2 ld + 1 doubles, bit 1 = the
 * tail; mask 3 = one collective over the whole buffer (single-sweep erm iterations). */
int  rbl_pending_reduce(rbl_solver* h, int* mask);
/* sum_i sigma_i * loss_(i) of n_total gathered values of v = D w on the device
 * (objective.py:73-81 without the regulariser) */
int  rbl_risk_from_v(rbl_solver* h, const void* v_all_dev, double* out);
/* padded leading dimension of D / G / q / w buffers, CU count, Lipschitz constant of G */
int  rbl_info(rbl_solver* h, int64_t* ld, int* num_cu, double* lipschitz);

/* ---- measurement ------------------------------------------------------------------ */
/* accumulated HIP-event time of the two n x d sweep kernels since the last reset */
enum { RBL_KERNEL_GEMV = 0, RBL_KERNEL_GEMVT = 1, RBL_KERNEL_SWEEP_ERM = 2 };
int  rbl_kernel_time(rbl_solver* h, int which, double* total_ms, int64_t* launches);
int  rbl_reset_kernel_times(rbl_solver* h);
/* The timed launches of one kernel since the last reset, one by one in launch order (milliseconds): the first
 * min(cap, *count) of them are written to out_ms, *count is how many there are.  bench.py reports the first and
 * the last of the timed region and the median of its last third beside the mean (the reference keeps a
 * cumulative time stamp per iteration, algorithms.py:162; this is its per-launch counterpart). */
int  rbl_kernel_samples(rbl_solver* h, int which, double* out_ms, int64_t cap, int64_t* count);
/* enable: 0 = no HIP events inside the iteration (default: an event record costs ~5 us of stream
 * time), 1 = events around the sweep kernels (rbl_kernel_time), 2 = also around the phases (the
 * ms_* fields of rbl_stats, 0 otherwise) */
int  rbl_profile_kernels(rbl_solver* h, int enable);
/* kernel events (level >= 1) on every `every`-th iteration only; default 1.  With several GPUs the pass
 * of a rank is short (0.55 ms at 8 x 750 000 rows) and two events per iteration are 2 % of it. */
int  rbl_profile_sampling(rbl_solver* h, int every);

/* ---- kernel-level entry points over host buffers (parity tests call these) -------- */
/* element prox (src/util/individual_solver.py:112-123) */
int  rbl_k_prox(int loss, int64_t n, const double* sigma, double rho, const double* m, double* out);
/* stable ascending sort of float64 keys with index payload (algorithms.py:92-93) */
int  rbl_k_sort(int64_t n, const double* keys, double* sorted_keys, uint32_t* perm);
/* generalised PAV on sorted m (src/util/pav.py:93-178) */
int  rbl_k_pav(int loss, int64_t n, const double* sigma, double rho, const double* m_sorted,
               double* out, int64_t* n_merges);
/* EHRM z-step on sorted m (src/util/PAV_cpt.py:169-293): branch -1 = choose by the
 * singleton-stage scalar test, 0 = a, 1 = b; *branch_out receives the choice */
int  rbl_k_pav_ehrm(int64_t n, const double* sigma_a, const double* sigma_b, double B, double rho,
                    const double* m_sorted, int branch, double* out, int* branch_out);
/* v = D w and q = D^T c on a host matrix (storage: RBL_STORE_*) */
int  rbl_k_gemv(int storage, int64_t n, int64_t d, const double* D, const double* w, double* v);
int  rbl_k_gemvt(int storage, int64_t n, int64_t d, const double* D, const double* c, double* q);
/* G = D^T D (MFMA f64) */
int  rbl_k_gram(int storage, int64_t n, int64_t d, const double* D, double* G);
/* w-steps in Gram space: lasso / ridge / smoothed-l1 */
int  rbl_k_wstep(int wstep, int64_t d, const double* G, const double* q, double rho, double reg,
                 double smooth_t, const double* w0, double tol, double* w_out, int* iters);
/* sigma generators (src/optim/objective.py:97-164) */
int  rbl_k_weights(int weight_function, int64_t n, const double* args, int n_args,
                   double* alphas, double* betas);

/* ---- the reference's competitor baselines ---------------------------------------------------
 * SGDmethod (SGD_solver.py:9-96 -> StochasticSubgradientMethod, existing_methods/lerm_main/src/optim/algorithms.py:54-98)
 * and LSVRGmethod (LSVRG_solver.py:9-98 -> LSVRG, algorithms.py:150-253) on the competitor's objective
 * (existing_methods/lerm_main/src/optim/objective.py:41-112).  One call = one epoch.  The index and random-sign
 * streams are the reference's own host generators (torch.randperm / numpy RandomState / numpy.random.choice /
 * torch.rand) and are passed in as data; the Python mirrors SGD_solver.py / LSVRG_solver.py of the package draw
 * them exactly as the reference does.  X: n x d host rows (float64), y01: labels in {0, 1} (the reference maps -1
 * to 0 for both losses, SGD_solver.py:13-14). */
typedef struct rbl_baseline rbl_baseline;
int  rbl_bl_create(int64_t n, int64_t d, const double* X, const double* y01, int loss, int has_lossB, double lossB,
                   double l2_reg, double l1_reg, int device, rbl_baseline** out);
int  rbl_bl_destroy(rbl_baseline* h);
int  rbl_bl_set_w(rbl_baseline* h, const double* w);
int  rbl_bl_get_w(rbl_baseline* h, double* w);
/* StochasticSubgradientMethod.start_epoch + `steps` x step (algorithms.py:80-93): mini-batch s = rows
 * order[s*batch .. min(n, (s+1)*batch)); alphas_b / betas_b: the `batch`-sample weights (objective.py:72-75;
 * betas_b NULL unless EHRM); rands: one torch.rand(1) per step for the l1 subgradient at 0 (NULL without l1_reg) */
int  rbl_bl_sgd_epoch(rbl_baseline* h, const int32_t* order, int steps, int batch, const double* alphas_b,
                      const double* betas_b, double lr, const float* rands);
/* LSVRG.start_epoch + `steps` x step (algorithms.py:183-253): alphas / betas are the n-sample weights; samples[s]
 * is a row index (uniform != 0: RandomState.randint) or a rank of the checkpoint's sorted order (uniform == 0:
 * numpy.random.choice(n, p=alphas)) */
int  rbl_bl_lsvrg_epoch(rbl_baseline* h, const double* alphas, const double* betas, const int32_t* samples, int steps,
                        int uniform, double lr, const float* rands);

#ifdef __cplusplus
}
#endif
#endif /* RBL_H */
