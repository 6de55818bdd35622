/*
 * pcabo.h - C ABI of libpcabo.so: the MI355X (gfx950) implementation of the PCA_BO inner loop.
 *
 * The reference (IvanBanny/para-ortho-pca-bo) is pure Python and has no FFI; the four places
 * where its hot loop hands arithmetic to third-party CPU libraries are the boundary this
 * library replaces.  Each entry point below names the reference call
 * site (file:line under /root/reference) whose work it performs.
 *
 * Conventions
 *   - fp64 everywhere, row-major, caller owns every buffer passed in.
 *   - Pointers are HOST pointers unless the context is switched with pcabo_set_pointer_mode()
 *     to PCABO_PTR_DEVICE, in which case the bulk inputs/outputs (marked [bulk]) are device
 *     pointers on the context's device (e.g. torch tensor .data_ptr()); small scalar/vector
 *     results (marked [host]) are always written to host memory.
 *   - Every call returns 0 on success or a negative pcabo_status; nothing throws across the ABI.
 *     pcabo_last_error() returns a human-readable message for the last failure on a context.
 *   - A context owns one HIP stream plus all workspaces, sized at creation for (max_n, max_d,
 *     max_q).  A context is single-threaded; distinct contexts are independent (one per
 *     concurrent BO run; ctypes releases the GIL around calls).
 *   - There is NO CPU fallback: without a usable HIP device pcabo_ctx_create() fails with
 *     PCABO_ERR_HIP.
 */
#ifndef PCABO_H
#define PCABO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pcabo_ctx pcabo_ctx;

typedef enum {
  PCABO_OK = 0,
  PCABO_ERR_ARG = -1,      /* bad argument / size exceeds context capacity / call order */
  PCABO_ERR_NOT_PD = -2,   /* K + s2 I not positive definite after the jitter retries */
  PCABO_ERR_HIP = -3,      /* HIP runtime error (message in pcabo_last_error) */
  PCABO_ERR_NAN = -4,      /* NaN met in an acquisition gradient (botorch raises here) */
  PCABO_ERR_TIMEOUT = -5   /* device did not answer within the watchdog interval */
} pcabo_status;

enum { PCABO_KERNEL_MATERN52 = 0, PCABO_KERNEL_RBF = 1 };
enum { PCABO_ACQ_LOG_EI = 0, PCABO_ACQ_PI = 1 };
enum { PCABO_PTR_HOST = 0, PCABO_PTR_DEVICE = 1 };

/* ABI version of this header (checked by the Python loader). */
int pcabo_abi_version(void);

/* Number of HIP devices visible (0 when none; never fails). */
int pcabo_device_count(void);

/* Create / destroy a per-run context on `device`.
 * max_n: largest number of evaluated points (budget); max_d: ambient dimension;
 * max_q: largest number of query points per pcabo_acq_eval call (>= raw_samples). */
int pcabo_ctx_create(int device, int max_n, int max_d, int max_q, pcabo_ctx** out);
int pcabo_ctx_destroy(pcabo_ctx* ctx);
int pcabo_set_pointer_mode(pcabo_ctx* ctx, int mode);
int pcabo_last_error(pcabo_ctx* ctx, char* buf, int buflen);

/* Per-context switches.
 * PCABO_OPT_RESIDENT (default 1): pcabo_optimize_acqf may serve all evaluations of a call from ONE resident launch of
 *   the acquisition kernel (query points through a mailbox); 0 = one launch per evaluation.  Same arithmetic, same bits.
 * PCABO_OPT_BESTF_F32 (default 1): best_f is rounded to float32 before use, as botorch does when the reference hands it
 *   a Python float (`torch.as_tensor(best_f)`, PCA_BO.py:199-203); 0 keeps all 64 bits (what botorch does when the
 *   objective returned a numpy float64 scalar).
 * PCABO_OPT_GROUP_ACQ (default 0; 1 for the contexts of a batch): value+gradient evaluations of up to 32 points run through the
 *   throughput kernel (one work-group per restart group of <= 5 points and 64-row slab) instead of the per-query
 *   latency kernels; implies one launch per evaluation.  Same formulas, another summation order (~1e-15 relative), so a
 *   run is bit-reproducible within a mode, not across modes. */
enum { PCABO_OPT_RESIDENT = 0, PCABO_OPT_BESTF_F32 = 1, PCABO_OPT_GROUP_ACQ = 2,
       PCABO_OPT_DEVICE_LBFGSB = 3, PCABO_OPT_LBFGSB_CUS = 4 /* batches only, see pcabo_batch_set_option */ };
int pcabo_set_option(pcabo_ctx* ctx, int option, int value);

/* Rows A-C (+D,J): rank-weighted PCA of the evaluated points.
 * Replaces PCA_BO._calculate_weights + _transform_points_to_reduced_space
 * (Algorithms/BayesianOptimization/PCA_BO.py:316-408), i.e. numpy + sklearn PCA().fit/transform.
 *   X[n*d]      [bulk] evaluated points
 *   f[n]        [bulk] objective values; used only when ranks == NULL (device ranking, ties
 *               broken by index)
 *   ranks[n]    [bulk] 1-based ranks (best = 1) as numpy's argsort(argsort(.))+1 gives them, or NULL
 *   noise[n*d]  [bulk] the N(0,1e-8) draw of PCA_BO.py:376 (numpy global RNG), or NULL for none
 *   n_components > 0 fixes k; otherwise k = #(cumsum(evr) <= var_threshold) + 1 clamped to [1, min(n,d)]
 * Outputs (any may be NULL):
 *   data_mean[d], pca_mean[d] [host]; comps[r*d] [host] all r=min(n,d) components (sklearn
 *   components_, rows sorted by decreasing variance, sign: max-|.| entry of each row positive);
 *   evr[r] [host]; *k [host]; Z[n*k] [bulk] reduced coordinates (PCA_BO.py:407).
 * The context keeps X-mean, the k leading components and Z on the device for the calls below. */
int pcabo_wpca(pcabo_ctx* ctx, const double* X, const double* f, const int64_t* ranks, int n, int d,
               int maximize, double var_threshold, int n_components, const double* noise,
               double* data_mean, double* pca_mean, double* comps, double* evr, int* k, double* Z);

/* Rows D-H: condition the exact GP on (Z, y).
 * Replaces PCA_BO._initialize_model (PCA_BO.py:502-545: SingleTaskGP(MaternKernel(2.5),
 * Standardize, Normalize)) and the lazy Gram/Cholesky/alpha/root-inverse that gpytorch performs
 * on the first posterior call inside optimize_acqf (PCA_BO.py:607).
 *   Z[n*k]           [bulk] reduced points, or NULL to use the Z of the last pcabo_wpca call
 *   y[n]             [bulk] objective values (un-standardised)
 *   norm_bounds[2*k] [host] Normalize bounds (lo row, hi row) or NULL for PCA_BO.py:514-518
 *   lengthscale, noise: model constants (reference values: ln 2 and exp(-5)); kernel: PCABO_KERNEL_*
 * On Cholesky breakdown jitter 1e-8, 1e-7, 1e-6 is added (psd_safe_cholesky) before PCABO_ERR_NOT_PD. */
int pcabo_gp_condition(pcabo_ctx* ctx, const double* Z, const double* y, int n, int k,
                       const double* norm_bounds, double lengthscale, double noise, int kernel);

/* The same in two halves: _begin stages the inputs and enqueues every kernel without waiting, _end waits,
 * checks the factorisation and performs the jitter retries.  Lets the host prepare the next phase (e.g. the
 * Sobol engine of the initial-condition draw) while the device conditions the GP. */
int pcabo_gp_condition_begin(pcabo_ctx* ctx, const double* Z, const double* y, int n, int k,
                             const double* norm_bounds, double lengthscale, double noise, int kernel);
int pcabo_gp_condition_end(pcabo_ctx* ctx);

/* pcabo_gp_condition_end followed by pcabo_acq_eval(values only) with the evaluation enqueued behind the conditioning
 * instead of after the host has seen it finish (the 512 raw samples of botorch's gen_batch_initial_conditions,
 * PCA_BO.py:607-614, are drawn from the search box, which is known long before the factorisation ends).
 * Same results as the two calls. Xq[q*k] [bulk], val[q] [bulk] (device-pointer mode simply runs the two calls). */
int pcabo_gp_condition_end_eval(pcabo_ctx* ctx, const double* Xq, int q, double best_f, int maximize, int acq,
                                double* val);

/* Rows A-H as ONE enqueue: pcabo_wpca immediately followed by pcabo_gp_condition_begin(Z = NULL,
 * norm_bounds = NULL), i.e. PCA_BO._transform_points_to_reduced_space + _initialize_model of one iteration
 * (PCA_BO.py:343-408 and :502-545) without the host round trip between them: the conditioning launches are queued
 * behind the projection before the host has seen k (the kernels read it on the device).  Arguments as in the two
 * calls (gp_noise = the likelihood noise, `noise` = the PCA_BO.py:376 draw).  Returns when the wPCA results are on the
 * host, with the conditioning still in flight: finish with pcabo_gp_condition_end().  With all five output pointers NULL
 * it returns right after the enqueue; pcabo_wpca_results() then waits for and delivers the wPCA results. */
int pcabo_wpca_gp_condition_begin(pcabo_ctx* ctx, const double* X, const double* f, const int64_t* ranks, int n,
                                  int d, int maximize, double var_threshold, int n_components, const double* noise,
                                  const double* y, double lengthscale, double gp_noise, int kernel,
                                  double* data_mean, double* pca_mean, double* comps, double* evr, int* k);

/* Second half of pcabo_wpca_gp_condition_begin when that was called with all five output pointers NULL (enqueue only):
 * waits for the wPCA results - the conditioning keeps running - and hands them out.  Between the two calls the host can
 * do work that only needs a guess of k (PCA_BO builds the scrambled Sobol engine of botorch's initial-condition draw with
 * the previous iteration's k while the eigen-decomposition runs, and rebuilds it in the rare case that k changed).
 * Must precede pcabo_acq_bounds / pcabo_gp_condition_end(_eval). */
int pcabo_wpca_results(pcabo_ctx* ctx, double* data_mean, double* pca_mean, double* comps, double* evr, int* k);

/* Row J: search box of the acquisition optimiser, PCA_BO.py:558-573. bounds[2*k] [host] (lo row, hi row).
 * May be called between pcabo_gp_condition_begin and _end: it then waits only for the statistics kernel, so the
 * raw samples of the initial-condition draw can be generated while the factorisation is still running. */
int pcabo_acq_bounds(pcabo_ctx* ctx, double* bounds);

/* Row I: batched acquisition value (+ gradient) at q query points of the reduced space.
 * Replaces LogExpectedImprovement / ProbabilityOfImprovement .forward + torch autograd as driven
 * by botorch (PCA_BO.py:199-203, 607-614).
 *   Xq[q*k] [bulk]; best_f as the reference passes it (current_best; it is rounded to float32
 *   like torch.as_tensor(python float) does); val[q] [bulk]; grad[q*k] [bulk] or NULL. */
int pcabo_acq_eval(pcabo_ctx* ctx, const double* Xq, int q, double best_f, int maximize, int acq,
                   double* val, double* grad);
/* Name suggested by SURVEY.md 8b; identical to pcabo_acq_eval(..., PCABO_ACQ_LOG_EI, ...). */
int pcabo_logei(pcabo_ctx* ctx, const double* Xq, int q, double best_f, int maximize,
                double* val, double* grad);

/* Rows M-N: multi-start L-BFGS-B over the acquisition (botorch gen_candidates_scipy semantics:
 * restarts are optimised jointly in groups of `batch_limit`, objective -sum_j acq(x_j), scipy
 * L-BFGS-B defaults m=10, factr=1e7 (ftol 2.22e-9), pgtol=1e-5, maxls=20, maxfun=15000).
 * Replaces botorch.optim.optimize_acqf's optimisation stage (PCA_BO.py:607-614).
 *   ics[num_restarts*k] [host] initial conditions; bounds[2*k] [host];
 *   cand[num_restarts*k] [host] clamped final points; vals[num_restarts] [host] acquisition there;
 *   info[4*ngroups] [host] or NULL: per group {iterations, function evaluations, warnflag, task};
 *   returns PCABO_OK; *failed (may be NULL) is set to 1 when a group ended abnormally (the case in
 *   which botorch re-draws initial conditions and retries once). */
int pcabo_optimize_acqf(pcabo_ctx* ctx, const double* ics, int num_restarts, int batch_limit,
                        const double* bounds, int maxiter, double best_f, int maximize, int acq,
                        double* cand, double* vals, int* info, int* failed);

/* Row O: x = z Ck + pca_mean + data_mean (PCA_BO.py:410-434). z[k] [host] -> x[d] [host]. */
int pcabo_inverse_map(pcabo_ctx* ctx, const double* z, double* x);

/* Introspection used by the parity tests (all [host] outputs, row-major n x n / vectors). */
int pcabo_get_gp_state(pcabo_ctx* ctx, double* K_chol /*n*n lower*/, double* Rinv /*n*n lower*/,
                       double* alpha /*n*/, double* y_mean_std /*2*/, double* norm_bounds /*2*k*/);
int pcabo_get_gram(pcabo_ctx* ctx, double* K /*n*n, symmetric, incl. noise*/);

/* Host-only L-BFGS-B (no device work): minimise a callback objective with box bounds, same
 * algorithm and defaults as scipy.optimize.minimize(method="L-BFGS-B").  Used by the CPU tests
 * to pin this library's optimiser against scipy itself.
 *   fg(x, g, user) returns f and fills g.  Returns warnflag (0 converged, 1 limit, 2 abnormal). */
typedef double (*pcabo_fg_callback)(const double* x, double* g, void* user);
int pcabo_lbfgsb_minimize(int nvar, double* x, const double* lower, const double* upper,
                          pcabo_fg_callback fg, void* user, int m, double factr, double pgtol,
                          int maxiter, int maxfun, int maxls, double* f_out, int* nit, int* nfev,
                          int* task_out);
/* The O(m n) loops of the host L-BFGS-B have AVX2 forms that take, bit for bit, the scalar loops' iterates (same operands,
 * same order, no fma); on by default where the CPU has AVX2.  0 selects the scalar loops (process-wide; the test that
 * compares the two uses it), 1 the default again.  Returns the previous setting. */
int pcabo_lbfgsb_set_vector_kernels(int enabled);
/* Summation order of the sums over the variables (d'd, g'd, r'r, W'd ...) in optimisers started by pcabo_lbfgsb_minimize from now
 * on: 0 (default) the published order - scipy's iterates; 1 the 64-lane tree order in which the device-resident optimiser
 * (PCABO_OPT_DEVICE_LBFGSB) steps: lane l adds terms l, l + 64, ..., then a balanced tree of adjacent pairs.  With 1 the host class
 * is the device's twin bit for bit (tests/test_gpu_device_lbfgsb.py); against scipy it is one more rounding of the same sums
 * (tests/test_lbfgsb_vs_scipy.py, tests/test_lbfgsb_divergence.py state what that does).  Returns the previous setting. */
int pcabo_lbfgsb_set_sum_order(int order);

/* Host-only helper for the initial-condition draw (botorch -> torch.quasirandom.SobolEngine, row K):
 * the matrix scramble of torch's `_sobol_engine_scramble_` on state[k*30] (in/out) with the k lower-
 * triangular 30x30 0/1 matrices ltm[k*30*30] (as drawn by torch.randint(...); entries on and above the diagonal are not
 * read - torch's .tril() need not be applied); bit-identical to torch, ~100x faster than torch's accessor loop.  The random
 * bits themselves still come from torch's generator. */
int pcabo_sobol_scramble(int64_t* state, const int64_t* ltm, int k);
/* The draw of such an engine (fresh: nothing generated yet), bit-identical to SobolEngine.draw(n, dtype=float64) followed by
 * botorch's map into the box, out[i*k + j] = lo[j] + rng[j] * u[i][j] (lo = rng = NULL: u itself).  state[k*30] after
 * pcabo_sobol_scramble, shift[k] = sum_b bit_b 2^b of the k x 30 shift bits (torch's `shift`). */
int pcabo_sobol_draw(const int64_t* state, const int64_t* shift, int k, int n, const double* lo, const double* rng, double* out);
/* The same for every run of a lock-step batch in one call: run r draws n points (n x ks[r]) into outs[r], mapped into the box
 * [lo(k), hi(k)] found at boxes + r * box_stride (the packing of pcabo_batch_acq_bounds); states[r] == NULL skips run r. */
int pcabo_sobol_draw_rows(const int64_t* const* states, const int64_t* const* shifts, const int* ks, int rows, int n,
                          const double* boxes, long long box_stride, double* const* outs);
/* torch's CPU generator restated for the host's pacing thread (csrc/host_entry.cpp): `blob` = the generator's state exactly as
 * torch.Generator.get_state() exports it (5056 bytes), read and ADVANCED in place - set_state(blob) afterwards leaves torch's
 * generator where torch's own call would have left it.
 * pcabo_torch_randint2: torch.randint(2, (count,), generator=g) - the Sobol scramble bits of botorch's draw_sobol_samples.
 * pcabo_torch_multinomial_rows: torch.multinomial(weights[r], n_pick, replacement=False, generator=g_r) for `rows` rows, one
 *   generator each (blobs[r] == NULL: row skipped) - the Boltzmann pick of botorch's initialize_q_batch; out[rows][n_pick].
 * pcabo/hostrng.py verifies both against torch at import and keeps torch's own calls if they ever differ. */
int pcabo_torch_randint2(void* blob, int64_t count, int64_t* out);
int pcabo_torch_multinomial_rows(void* const* blobs, const double* weights, int rows, int n, int n_pick, int64_t* out);
/* botorch's initialize_q_batch (Boltzmann pick of the restarts' initial conditions) for `rows` runs in one call: vals[rows][n] raw-sample
 * scores, eta the temperature; out[rows][n_pick]; flags[rows]: 0 picked, 1 all values equal (nothing drawn: the caller takes the
 * random-permutation path), 2 skipped (blobs[r] == NULL). */
int pcabo_boltzmann_pick_rows(void* const* blobs, const double* vals, int rows, int n, int n_pick, double eta, int64_t* out, int* flags);

/* Device-time accounting: accumulated HIP-event time (ms, events recorded on the context's own
 * stream), launch count and ALGORITHMIC bytes / flops of the kernel groups since the last reset.
 * which: 0 wpca (5 kernels), 1 normalise+Gram (3 kernels), 2 Cholesky (2 kernels per 64-wide panel),
 * 3 root inverse + alpha (5 launches), 4 acquisition kernel, single-launch evaluations (<= 32 queries: the L-BFGS-B
 * rounds), 5 acquisition, large batches (two launches: partials + combine).
 * Enabled by pcabo_set_profiling(ctx, 1) (adds an event pair per bracketed group of launches; the reading of two events
 * recorded back to back, calibrated when profiling is switched on, is subtracted from every pair).
 * pcabo_get_profile_calibration: that reading (ms) and the reading of an event pair around an EMPTY kernel (ms), both medians
 * of 64, so that a caller can state raw event times next to the calibrated ones (bench.py does). */
int pcabo_set_profiling(pcabo_ctx* ctx, int enabled);
int pcabo_get_profile_calibration(pcabo_ctx* ctx, double* pair_ms, double* empty_kernel_ms);
int pcabo_get_profile(pcabo_ctx* ctx, int which, double* ms, int64_t* launches, double* bytes, double* flops);
int pcabo_reset_profile(pcabo_ctx* ctx);

/* ---- Batched contexts: B independent BO runs advancing in lock-step ---------------------------------------------------
 * The reference's outer loop is a list of independent runs (Algorithms/Experiment/ExperimentRunner.py:137-183: 30 instances
 * x functions x dimensions, one optimiser object each).  A batch holds B per-run contexts of equal capacity side by side in
 * one device slab and advances them together: every phase of rows A-H is ONE launch sequence with blockIdx.z = run (the
 * work-groups of all runs fill the chip together), the raw-sample scoring is one launch for all runs, and the L-BFGS-B
 * rounds of all runs are fed to shared acquisition launches (one per gang of runs and round; an active-query table names the
 * (run, query) pairs of the round).  The kernels and their per-run grids are those of the single context, so every run's
 * numbers are bit-identical to the same run in a context of its own.
 * All runs of a call share n and d (same budget / DoE sizes: the runs of one (function, dimension) cell or of several cells
 * with the same dimension); k and best_f are per run.  Per-run arrays are laid out [B][...] with the strides named below.
 * status[b] receives the pcabo_status of run b where a call can fail per run; the return value reports argument / HIP errors.
 * pcabo_batch_ctx(batch, b) is run b's context: the single-context calls above work on it (introspection, the rare retry of
 * one run), but not pcabo_ctx_destroy. */
typedef struct pcabo_batch pcabo_batch;
int pcabo_batch_create(int device, int B, int max_n, int max_d, int max_q, pcabo_batch** out);
int pcabo_batch_destroy(pcabo_batch* batch);
/* Worker threads of the L-BFGS-B phase (one gang of runs and one HIP stream each; default min(8, B)).
 * A worker spins while its launch is in flight: when several batches of one process advance side by side (one host thread
 * per batch - the reference's cells of different dimension, or two halves of one cell) give each its share of the cores.
 * Results do not depend on the number.  Not during a call on this batch. */
int pcabo_batch_set_workers(pcabo_batch* batch, int workers);
/* PCABO_OPT_GROUP_ACQ (default 1): the L-BFGS-B rounds of the batch go through the throughput kernel (k_acq_group); 0: through
 * the per-query kernels a stand-alone context uses by default - a run of the batch is then bit-identical to the same run in a
 * context of its own with default options (with 1 it is bit-identical to such a context with PCABO_OPT_GROUP_ACQ set).
 * PCABO_OPT_DEVICE_LBFGSB (default 0): 1 = pcabo_batch_optimize_acqf runs every restart group's whole L-BFGS-B optimisation
 *   (gen_candidates_scipy of PCA_BO.py:607-614) inside ONE kernel launch, a work-group per group: evaluate, step, next point
 *   without a host round trip (csrc/kernels_lbfgsb.hip; needs n <= 512, k <= 40, batch_limit <= 5 and finite bounds - other
 *   calls take the host-paced path).  Its evaluation sums in an order of its own (a third arithmetic mode, ~1e-15 relative from
 *   the other two).  2 = the same evaluation kernel driven by the HOST's L-BFGS-B, one launch per round: slow, the reference
 *   the device stepping is compared with bit for bit (tests/test_gpu_device_lbfgsb.py).
 * PCABO_OPT_LBFGSB_CUS (default 0 = the whole chip): the launches of the device-resident optimiser go to a stream confined to the
 *   first `value` compute units (hipExtStreamCreateWithCUMask).  Its work-groups hold a whole CU each for milliseconds; with
 *   several batches of one process in flight the rest of the chip stays free for the short kernels of the others.  Results do
 *   not depend on it. */
int pcabo_batch_set_option(pcabo_batch* batch, int option, int value);
/* Shapes the device-resident optimiser (PCABO_OPT_DEVICE_LBFGSB = 1) covers: points n <= *max_n, reduced dimension k <= *max_k,
 * batch_limit <= *max_group.  A call beyond them is NOT served by it: pcabo_batch_optimize_acqf takes the host-paced path for that
 * call (pcabo_batch_optimize_acqf_begin returns 1), i.e. ANOTHER arithmetic mode.  A driver that promises one trajectory per seed
 * therefore decides per RUN, before it starts, from (budget, dimension) - pcabo/batchrun.py refuses acq_kernel="device" for a
 * run that could leave these limits, Algorithms/Experiment/ExperimentRunner.py resolves the mode per dimension of the experiment.
 * Host code; no device needed. */
int pcabo_device_lbfgsb_limits(int* max_n, int* max_k, int* max_group);
int pcabo_batch_last_error(pcabo_batch* batch, char* buf, int buflen);
pcabo_ctx* pcabo_batch_ctx(pcabo_batch* batch, int b);

/* Rows A-H of all runs as one enqueue (pcabo_wpca_gp_condition_begin for B runs).  X[B][n*d], ranks[B][n], noise[B][n*d] or
 * NULL, y[B][n].  Returns after the enqueue; pcabo_batch_wpca_results waits for the wPCA part only. */
/* Doubles between the blocks of two runs in the X / noise / y arrays handed to pcabo_batch_wpca_gp_condition_begin (0 = dense,
 * the default: n*d, n*d, n).  A driver that keeps X as [B][budget][d] passes budget*d and saves a dense copy per iteration. */
int pcabo_batch_set_input_strides(pcabo_batch* batch, size_t x_stride, size_t noise_stride, size_t y_stride);
int pcabo_batch_wpca_gp_condition_begin(pcabo_batch* batch, const double* X, const int64_t* ranks, const double* noise,
                                        const double* y, int n, int d, int maximize, double var_threshold,
                                        int n_components, double lengthscale, double gp_noise, int kernel);
/* Rows D-H of all runs without the weighted PCA (pcabo_gp_condition_begin for B runs): the reference's Vanilla_BO conditions its
 * GP on the raw d-dimensional points with Normalize switched off (Vanilla_BO.py:166-196) - the caller passes identity bounds.
 * Z[B][n*k], y[B][n] (strides: pcabo_batch_set_input_strides), norm_bounds[2*k] (lo[k], hi[k], the same for every run) or NULL
 * (bounds from the data, as for PCA_BO).  Continue with pcabo_batch_gp_condition_end_eval / pcabo_batch_optimize_acqf. */
int pcabo_batch_gp_condition_begin(pcabo_batch* batch, const double* Z, const double* y, int n, int k, const double* norm_bounds,
                                   double lengthscale, double gp_noise, int kernel);
/* data_mean[B][d], pca_mean[B][d], comps[B][d*d] (the first min(n,d)*d entries of each block), evr[B][d], k[B]; any may be NULL */
int pcabo_batch_wpca_results(pcabo_batch* batch, double* data_mean, double* pca_mean, double* comps, double* evr, int* k);
/* Row J for every run: bounds[B][2*max_d], run b's block holds lo[k_b] then hi[k_b]. */
int pcabo_batch_acq_bounds(pcabo_batch* batch, double* bounds);
/* Wait for the conditioning of all runs and score q points per run (values only; the raw samples of
 * gen_batch_initial_conditions), enqueued behind the conditioning.  Xq[B][q*max_d]: run b's block holds a dense q x k_b array;
 * best_f[B]; val[B][q]; status[B] (PCABO_ERR_NOT_PD for a run whose factorisation failed after the jitter retries). */
int pcabo_batch_gp_condition_end_eval(pcabo_batch* batch, const double* Xq, int q, const double* best_f, int maximize,
                                      int acq, double* val, int* status);
/* Rows M-N for every run (pcabo_optimize_acqf semantics per run).  ics[B][num_restarts*max_d] (dense num_restarts x k_b per
 * block), bounds[B][2*max_d] as pcabo_batch_acq_bounds gives them, best_f[B]; cand[B][num_restarts*max_d] (dense per block),
 * vals[B][num_restarts], info[B][4*ngroups] or NULL, failed[B], status[B]. */
int pcabo_batch_optimize_acqf(pcabo_batch* batch, const double* ics, int num_restarts, int batch_limit,
                              const double* bounds, int maxiter, const double* best_f, int maximize, int acq,
                              double* cand, double* vals, int* info, int* failed, int* status);
/* The two waiting calls of an iteration in halves, for a caller that advances SEVERAL batches from one thread (the reference's
 * outer loop over runs, ExperimentRunner.py:137-183, interleaved instead of threaded): _begin enqueues and returns, _end waits
 * and collects, pcabo_batch_busy tells whether the batch's stream still has work (1) or a waiting call would return at once (0).
 * pcabo_batch_optimize_acqf_begin needs PCABO_OPT_DEVICE_LBFGSB = 1 (the whole optimisation is one launch); it returns 1 -
 * nothing enqueued - when the call does not qualify for it, and the caller then uses pcabo_batch_optimize_acqf.  The arrays
 * handed to a _begin are read before it returns. */
int pcabo_batch_busy(pcabo_batch* batch);
int pcabo_batch_gp_condition_end_eval_begin(pcabo_batch* batch, const double* Xq, int q, const double* best_f, int maximize, int acq);
int pcabo_batch_gp_condition_end_eval_end(pcabo_batch* batch, const double* Xq, int q, const double* best_f, int maximize,
                                          int acq, double* val, int* status);
int pcabo_batch_optimize_acqf_begin(pcabo_batch* batch, const double* ics, int num_restarts, int batch_limit,
                                    const double* bounds, int maxiter, const double* best_f, int maximize, int acq);
int pcabo_batch_optimize_acqf_end(pcabo_batch* batch, int num_restarts, int batch_limit, double* cand, double* vals, int* info,
                                  int* failed, int* status);
int pcabo_batch_inverse_map_begin(pcabo_batch* batch, const double* z);
int pcabo_batch_inverse_map_end(pcabo_batch* batch, double* x);
/* Park / unpark runs: active[B]; a parked run still goes through the lock-step launches of rows A-K but is skipped by
 * pcabo_batch_optimize_acqf (status PCABO_ERR_ARG).  For runs that failed the way the reference's would (botorch raises on a
 * NaN acquisition gradient, reached once the reference's unclipped out-of-box candidates have blown up the search box). */
int pcabo_batch_set_active(pcabo_batch* batch, const int* active);
/* Device time of the phases of the last pcabo_batch_wpca_gp_condition_begin (HIP events on the batch's stream; call once
 * that conditioning has been waited for): ms[0] rows A-C, ms[1] Normalize + Gram, ms[2] Cholesky, ms[3] root inverse + alpha. */
int pcabo_batch_set_profiling(pcabo_batch* batch, int enabled);
int pcabo_batch_get_profile(pcabo_batch* batch, double* ms);
/* Row O for every run: z[B][max_d] (k_b entries used) -> x[B][d]. */
int pcabo_batch_inverse_map(pcabo_batch* batch, const double* z, double* x);
/* Value and gradient of the acquisition (botorch LogExpectedImprovement / ProbabilityOfImprovement + autograd, PCA_BO.py:199-203,
 * 681-682) at q <= 32 points per run through the evaluation of the device-resident optimiser (PCABO_OPT_DEVICE_LBFGSB on;
 * n <= 512, k <= 40).  Xq[B][q*max_d]: run b's points packed with stride k_b; val[B][q]; grad[B][q*max_d], same packing.
 * For diagnostics and parity tests: pcabo_batch_optimize_acqf evaluates inside its own launch. */
int pcabo_batch_device_acq_eval(pcabo_batch* batch, const double* Xq, int q, const double* best_f, int maximize, int acq,
                                double* val, double* grad);

/* ---- BBOB f15-f24 objectives on the device, for runs that advance in lock-step ---------------------------------------
 * The reference evaluates `problem(x)` on the host, one candidate per BO iteration (PCA_BO.py:263; the problems come from
 * ioh, ExperimentRunner.py:90).  With B runs in lock-step their B candidates are evaluated in one launch.
 * tables[B][pcabo_bbob_table_doubles(d)]: per run [x_opt(d) | R(d*d) | M(d*d) | aux], generated on the host by the seeded
 * legacy generators (pcabo/bbob.py builds them; layout in pcabo/bbob_device.py).  fid[B] in 15..24.
 * pcabo_bbob_eval: X[B][d] candidates -> raw[B] (value WITHOUT f_opt; `penalty` for a candidate outside [lb, ub]^d, which is
 * not evaluated: PCA_BO.py:248-263) and oob[B] (may be NULL). */
typedef struct pcabo_objective pcabo_objective;
int pcabo_bbob_table_doubles(int d);
int pcabo_bbob_create(int device, int B, int d, const int* fid, const double* tables, pcabo_objective** out);
int pcabo_bbob_destroy(pcabo_objective* obj);
int pcabo_bbob_eval(pcabo_objective* obj, const double* X, double lb, double ub, double penalty, double* raw, int* oob);

/* ---- Final gather of best-so-far values across the GPUs of a node -------------------------------------
 * Runs are independent (ExperimentRunner.py:137-183: one optimiser object per (function, dimension, instance)), so the
 * multi-GPU path has NO collective inside the loop; the one exchange of the design is an all-gather of each rank's
 * best-so-far values after its runs, over RCCL (xGMI inside a node).  One process per GPU:
 *   rank 0:      pcabo_comm_unique_id(id)            128 bytes (ncclUniqueId); the CALLER hands them to the other ranks
 *                                                    (a file, the launcher's environment, its own store)
 *   every rank:  pcabo_comm_create(id, world, rank, device, &comm)       (collective: returns once all ranks have joined)
 *                pcabo_gather_best(comm, local, n_local, all)            all[world * n_local] [host], rank-major; n_local equal on all ranks
 *                pcabo_comm_destroy(comm)
 * librccl.so is opened on first use (no link-time dependency).  pcabo/distributed.py offers the same gather through
 * torch.distributed for callers that already run under it (bench.py does). */
typedef struct pcabo_comm pcabo_comm;
int pcabo_comm_unique_id(char* id128);
int pcabo_comm_create(const char* id128, int world, int rank, int device, pcabo_comm** out);
int pcabo_gather_best(pcabo_comm* comm, const double* local, int n_local, double* all);
int pcabo_comm_last_error(pcabo_comm* comm, char* buf, int buflen);
int pcabo_comm_destroy(pcabo_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* PCABO_H */
