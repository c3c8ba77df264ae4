/* blueice_hip.h -- C ABI of libblueice_hip.so: the MI355X (gfx950) binned-likelihood hot path.
 *
 * The reference (JelleAalbers/blueice v1.2.1) is pure Python and has no FFI; its extension
 * points for this path are Python-level.  Every entry point below cites the reference
 * interface (file:line under /root/reference) whose work it takes over.  Python binds this
 * header with ctypes (blueice_amd/_capi.py); INTEGRATION.md shows the stub a blueice
 * maintainer would add.
 *
 * Conventions
 *   - All host buffers are borrowed for the duration of the call; the library copies what it
 *     needs.  Device memory is owned by the context until bi_destroy.
 *   - Return value: 0 = BI_OK, negative = error (message via bi_last_error).  Nothing throws
 *     across the boundary.  Numerical edge cases are VALUES (-inf / nan as the reference
 *     produces them), not errors.
 *   - One context = one device + one HIP stream.  A context is not re-entrant; distinct
 *     contexts are independent (one per GPU / per process in multi-GPU runs).
 *   - All floating point is IEEE binary64.  Layouts are C order.
 */
#ifndef BLUEICE_HIP_H
#define BLUEICE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: these are its only exports */
#endif

typedef struct bi_ctx bi_ctx;

enum {
    BI_OK = 0,
    BI_ERR_INVALID = -1, /* bad argument / shape mismatch */
    BI_ERR_HIP = -2,     /* HIP runtime failure (message has the hipError string) */
    BI_ERR_STATE = -3,   /* model or data not uploaded yet (reference: NotPreparedException,
                            blueice/likelihood.py:30-50) */
    BI_ERR_NOMEM = -4
};

/* per-point status bits written by bi_eval (reference behaviour in brackets) */
enum {
    BI_ST_OUT_OF_BOUNDS = 1, /* z outside the anchor box or nan [-inf, likelihood.py:345-347] */
    BI_ST_UNPHYSICAL = 2,    /* rates not in [0, inf) [-inf or ValueError, likelihood.py:397-415] */
    BI_ST_BB_ROOT1 = 4,      /* Beeston-Barlow `assert all(A1 <= 0)` would fail [likelihood.py:649] */
    BI_ST_BB_NEG = 8,        /* Beeston-Barlow `assert all(0 <= A)` would fail [likelihood.py:655] */
    BI_ST_BAD_DATASET = 16,  /* dataset index out of range */
    BI_ST_INTERNAL = 32      /* the in-launch reduction gave up waiting for a partial sum (result nan): a device fault */
};

/* ---- lifetime ------------------------------------------------------------------------- */
int bi_create(int device, bi_ctx** out);
void bi_destroy(bi_ctx* ctx);
const char* bi_last_error(const bi_ctx* ctx); /* ctx == NULL: last error of a failed bi_create */
const char* bi_version(void);
/* name (<= len bytes), compute units, total HBM bytes, gfx arch string (<= len bytes) */
int bi_device_info(bi_ctx* ctx, char* name, char* arch, int len, int* n_cu, int64_t* hbm_bytes);

/* ---- model: the anchor tensors built by GridInterpolator.make_interpolator ---------------
 * (blueice/pdf_morphers.py:57-67 fills `anchor_scores[A_0..A_{d-1}, *extra_dims]` and wraps it
 * in a RegularGridInterpolator; BinnedLogLikelihood.prepare builds three of them,
 * blueice/likelihood.py:248-251 (mus), :593-596 (ps), :598-601 (n_model_events)).
 *
 *   d          number of shape parameters (0 allowed: no morphing, likelihood.py:359-363)
 *   n_anchor   [d]   anchors per axis
 *   anchor_z   concatenated, ascending per axis (pdf_morphers.py:48)
 *   S, B       sources, total analysis bins (product of the analysis-space shape)
 *   ps         [A_0..A_{d-1}][S][B]
 *   mus        [A_0..A_{d-1}][S]
 *   n_model    NULL or [A_0..A_{d-1}][S][B]; only row `bb_source` is kept (likelihood.py:643)
 *   bb_source  -1 = no Beeston-Barlow, else the 'bb_single_source' index (likelihood.py:625-630)
 */
int bi_upload_model(bi_ctx* ctx, int d, const int32_t* n_anchor, const double* anchor_z, int S, int64_t B,
                    const double* ps, const double* mus, const double* n_model, int bb_source);

/* Streaming variant of the same: declare the grid, then hand over one anchor model at a time in
 * any order -- exactly the loop of pdf_morphers.py:62-65 -- so no dense host tensor is needed. */
int bi_model_begin(bi_ctx* ctx, int d, const int32_t* n_anchor, const double* anchor_z, int S, int64_t B,
                   int bb_source);
/* anchor_index: C-order linear index into the anchor grid; ps [S][B]; mus [S];
 * n_model_row NULL or [B] (= n_model_events[bb_source]) */
int bi_model_set_anchor(bi_ctx* ctx, int64_t anchor_index, const double* ps, const double* mus,
                        const double* n_model_row);
int bi_model_end(bi_ctx* ctx);

/* Bin-sharded models (a tensor too large for one GPU: every rank holds a slice of the bins of every row and
 * the per-rank log likelihoods add up).  Everything is additive over bins except the Beeston-Barlow
 * normalisation N_c = sum over ALL bins of n_model[c, bb_source, :] (likelihood.py:645): read the local sums,
 * all-reduce them across ranks, and write the global ones back before evaluating.  totals: [A]. */
int bi_get_bb_totals(bi_ctx* ctx, double* totals);
int bi_set_bb_totals(bi_ctx* ctx, const double* totals);

/* per-source `allow_negative` flags (likelihood.py:82-83,397-415); default all 0 */
int bi_set_allow_negative(bi_ctx* ctx, const int32_t* allow /*[S]*/);

/* ---- data: what BinnedLogLikelihood.set_data leaves in data_events_per_bin.histogram ------
 * (blueice/likelihood.py:603-609).  T datasets of B float64 counts each (toy MC: T > 1). */
int bi_upload_counts(bi_ctx* ctx, int64_t T, const double* counts /*[T][B]*/);

/* set_data on the device: declare the analysis space once (config['analysis_space'] = [[name, edges], ...],
 * blueice/likelihood.py:607), then bin N events into dataset 0 there -- numpy.histogramdd semantics, which is
 * what multihist's Histdd.add applies in likelihood.py:608-609 (right-most edge inclusive, events outside the
 * range dropped).  coords: [k][N] (one column per analysis dimension).  Replaces the context's data. */
int bi_set_analysis_space(bi_ctx* ctx, int k, const int32_t* n_edges /*[k]*/, const double* edges /*concatenated*/);
int bi_upload_events(bi_ctx* ctx, int64_t N, const double* coords);

/* Template building: the same binning as a stand-alone service, for the histograms the sources fill while the anchor
 * models are built -- DensityEstimatingSource.build_histogram's `mh.add(...)`, blueice/source.py:287-299, i.e.
 * numpy.histogramdd of N events in k dimensions (about a second per 10^6 three-dimensional events on a host core, times
 * sources x anchor models; ~3 ms here, most of it PCIe).  Needs no model and touches none of the context's state:
 * coords [k][N] and the edges are borrowed for the call, counts [prod(n_edges - 1)] (C order, host) is OVERWRITTEN with
 * the number of events per bin.  Unweighted events only. */
int bi_histogram_events(bi_ctx* ctx, int k, const int32_t* n_edges /*[k]*/, const double* edges /*concatenated*/,
                        int64_t N, const double* coords, double* counts);

/* Extended unbinned likelihood on the same machinery (UnbinnedLogLikelihood, blueice/likelihood.py:528-573;
 * extended_loglikelihood :678-690).  Upload the model with B = number of events and `ps` = the pdf values
 * of every source at every event for every anchor (what `Model.score_events(d)` returns,
 * likelihood.py:557-560, model.py:97-99), then switch the context:
 *     ll = -sum_s mu_s + sum_events log( sum_s mu_s p_s(x_e) ),
 * events whose summed density is not > 0 get `outlier_likelihood` instead when it is non-zero
 * (config 'outlier_likelihood', default 1e-12, likelihood.py:573,687-689).  No counts are needed;
 * bi_eval / bi_plan_* / bi_interpolate / bi_eval_full work as for the binned case.  A nan pdf value drops that
 * source's term for that event (numpy.nansum, likelihood.py:686).  B = 0 (no events) is allowed. */
int bi_set_unbinned(bi_ctx* ctx, double outlier_likelihood);

/* set_data of the unbinned likelihood on the device, for sources whose pdf is a histogram (HistogramPdfSource.pdf,
 * blueice/source.py:218-243): `templates` holds the DENSITY histograms of every source at every anchor as its model
 * rows ([A..][S][B], uploaded once like a binned model); this call evaluates all of them at N events and leaves the
 * result -- the [A..][S][N] tensor `Model.score_events(d)` would have produced anchor by anchor on the host,
 * likelihood.py:557-560 -- as the model of `target` (same device), copies the expected-event table, and switches
 * `target` to the unbinned likelihood as bi_set_unbinned does.  Only the k x N event coordinates cross PCIe.
 *   method 0 'piecewise': the density of the bin the event falls in; `grid` = the bin EDGES of every axis; an event
 *            on an edge belongs to the bin on its left, events outside take the first / last bin (the lookup of the
 *            package's Histdd);
 *   method 1 'linear':    scipy's RegularGridInterpolator over the bin CENTRES (`grid` = the centres of every axis,
 *            as the host computed them; >= 2 per axis), in its operation order; coords must already be clipped to
 *            [first centre, last centre] and finite, as source.py:232-241 does.
 * n_grid[k], grid concatenated; coords [k][N]. */
int bi_score_events(bi_ctx* templates, bi_ctx* target, int method, int k, const int32_t* n_grid, const double* grid,
                    int64_t N, const double* coords, double outlier_likelihood);

/* Event-level toy Monte Carlo on the device, for the unbinned likelihood with histogram-pdf sources: what
 * `Model.simulate` (blueice/model.py:69-91) does source by source on the host -- N_s ~ Poisson(mu_s) events, each drawn
 * from the source's pdf, for a histogram pdf a bin with probability density x volume and a uniform position inside it
 * (`HistogramPdfSource.simulate`, blueice/source.py:248-264) -- followed by `set_data` (likelihood.py:531-563), without
 * the events ever leaving HBM.  `templates` holds the density histograms of every source at every anchor (as for
 * bi_score_events); the events are drawn at parameter point (z, rate_scale) from the MORPHED densities, scored at every
 * anchor model, and become the (unbinned) data of `target`.  Philox4x32-10 keyed by the seed; counters are (event within
 * its source, source), so a toy does not depend on launch geometry.
 *   method 0 / 1   the sources' pdf_interpolation_method 'piecewise' / 'linear' (as bi_score_events; with 'linear' the
 *                  coordinates are clipped to the outer bin centres before scoring, source.py:232-241)
 *   n_edges [k], edges concatenated: the BIN EDGES of the analysis space
 *   n_per_source [S] or NULL: receives the number of events drawn per source
 * bi_download_events copies the simulated coordinates [k][N] (and the source index of every event) to the host,
 * N = bi_simulated_event_count. */
int bi_simulate_events(bi_ctx* templates, bi_ctx* target, const double* z, const double* rate_scale, int method, int k,
                       const int32_t* n_edges, const double* edges, uint64_t seed, double outlier_likelihood,
                       int64_t* n_per_source);
int bi_download_events(bi_ctx* target, double* coords /*[k][N] or NULL*/, int32_t* source /*[N] or NULL*/);
int64_t bi_simulated_event_count(const bi_ctx* target);

/* Toy-MC datasets generated on the device: n_{t,b} ~ Poisson(mu_b), mu_b = sum_s r_s p_{s,b}(z) -- the binned
 * equivalent of Model.simulate (blueice/model.py:69-91: Poisson number of events per source, each drawn from
 * the source's pdf) followed by set_data's binning (blueice/likelihood.py:603-609).  Philox4x32-10 keyed by
 * (seed; dataset, bin): reproducible and independent of launch geometry.  The T datasets REPLACE the
 * context's data and are stored as non-empty-bin lists only (no [T][B] array, no host transfer);
 * bi_eval_datasets works on them directly, point evaluations when the compacted templates fit the budget. */
int bi_generate_toys(bi_ctx* ctx, const double* z, const double* rate_scale, int64_t T, uint64_t seed);
/* Expands device-generated toys (non-empty-bin lists) into the dense [T][B] counts array as well, on the device: what
 * the paths that visit every bin need -- Beeston-Barlow point evaluations and gradients (likelihood.py:618-660 read n in
 * every bin), sparse = 0.  T * B * 8 bytes of HBM; a no-op when the counts are dense already. */
int bi_counts_to_dense(bi_ctx* ctx);
/* dense counts [B] of dataset t, from whatever form is resident */
int bi_download_counts(bi_ctx* ctx, int64_t t, double* out);

/* ---- the hot path -----------------------------------------------------------------------
 * P independent evaluations of LogLikelihoodBase.__call__ (blueice/likelihood.py:318-427)
 * without the Python-callable priors:
 *   mus = mus_interpolator(z); ps = ps_interpolator(z) [; n_model_events_interpolator(z)]
 *   (:355-357, scipy RegularGridInterpolator semantics, pdf_morphers.py:67-70)
 *   mus *= rate_scale (rate multiplier :366-368, livetime :374-382, efficiency :385-393)
 *   unphysical check (:397-415) -> -inf
 *   adjust_expectations ('bb_single', :618-660)
 *   _compute_likelihood = sum_bins poisson.logpmf(n | sum_s mus_s ps_s) (:662-675)
 *
 *   z          [P][d]   (ignored when d == 0)
 *   rate_scale [P][S]   or NULL (= all ones)
 *   dataset    [P]      or NULL (= dataset 0)
 *   out        [P]      log likelihoods (-inf / nan exactly where the reference gives them)
 *   status     [P]      or NULL; BI_ST_* bits
 * Points are grouped by (grid cell, dataset) internally so that corner templates are read once
 * per group. */
int bi_eval(bi_ctx* ctx, int64_t P, const double* z, const double* rate_scale, const int64_t* dataset,
            double* out, int32_t* status);

/* bi_eval(P = 1) in two halves, for a caller that evaluates several contexts at once -- a sum of likelihoods with
 * one context per term (LogLikelihoodSum, blueice/likelihood.py:867-955): bi_eval_begin on every context, then
 * bi_eval_end on every context, and the launches overlap.  One bi_eval_begin may be outstanding per context; any
 * other call on that context in between is a caller error, except bi_eval_end. */
int bi_eval_begin(bi_ctx* ctx, const double* z, const double* rate_scale, int64_t dataset);
int bi_eval_end(bi_ctx* ctx, double* out, int32_t* status);

/* Value and analytic gradient in ONE pass over the templates (no counterpart in the reference, whose
 * fits differentiate `make_objective`'s f numerically: blueice/inference.py:111-124,153-155).
 *   ll   [P]          as bi_eval
 *   grad [P][d + S]   d ll / d z_i (i < d; through the morph weights AND through mus(z)), then
 *                     d ll / d rate_scale_s.  Inside a grid cell ll is smooth; on an anchor the
 *                     derivative is the one of the cell the point is assigned to.  NaN where ll = -inf.
 * Needs 1 + d + S <= 16.
 * With Beeston-Barlow (bb_source >= 0; blueice/likelihood.py:618-660,693-712) the chain rule runs through the per-bin root:
 * mu_b = U_b + A_b p_b with p_b = r_i P_b / a_b, d mu = dU + p dA + A dp, dA from the root formula's partial derivatives;
 * in bins with U_b == 0 exactly the derivative of the reference's special case A = (n + a) / (1 + p_cal) is taken (on that
 * measure-zero set the two branches of the reference differ, so ll itself is not differentiable across it).  d <= 7 there.
 * status carries the Beeston-Barlow assertion bits of the value, as bi_eval does.
 * Extended unbinned likelihood (bi_set_unbinned / bi_score_events; blueice/likelihood.py:678-690): the same call returns
 *   d ll = -sum_s d mu_s + sum_events (sum_s d(mu_s p_s(x_e))) / (sum_s mu_s p_s(x_e));
 * an event that takes the outlier likelihood (density not > 0) is a constant and contributes no slope, a source whose term is
 * nan at an event is dropped from value and slopes alike (numpy.nansum, likelihood.py:686); `dataset` is ignored. */
int bi_eval_grad(bi_ctx* ctx, int64_t P, const double* z, const double* rate_scale, const int64_t* dataset, double* ll,
                 double* grad, int32_t* status);

/* The batched profile-fit engine's inner loop (host code): P minimisations of F variables each advance in lock-step, every
 * optimiser iteration ONE evaluation call over the problems still running -- what replaces the reference's loops of sequential
 * scipy fits (bestfit_scipy, blueice/inference.py:131-178, inside one_parameter_interval / plot_likelihood_ratio, :332-443).
 * BFGS per problem (Hessian estimate in direct form, exact reduced steps at bounds), Armijo backtracking with a ladder of
 * three trial steps per call, box constraints lo / hi [F] (+-inf = none); `kinks` (n_kinks [F] counts, values concatenated,
 * ascending; NULL = none): points along a variable where the function has a kink -- the interior anchors of a shape parameter,
 * the morph is linear between them (pdf_morphers.py:67-70) -- at which both one-sided slopes are taken and steps end.
 *   bi_minimize_batched   over a caller's objective: fun(user, n, F, x [n][F], rows [n], f [n], g [n][F]) evaluates f and its
 *                         gradient for problems `rows` at `x`; non-zero return aborts.  Needs no context and no GPU.
 *   bi_fit_batched        with the context's likelihood as the objective, f = -ll (bi_eval_grad), no caller code between the
 *                         iterations: variable j is shape parameter var_index[j] (var_kind 0: z_i = x_j) or the rate
 *                         multiplier of source var_index[j] (var_kind 1: rate_scale_s = x_j * unit[p][s]); z0 [P][d] and
 *                         scale0 [P][S] are the problems' other settings, dataset [P] or NULL.  Points at which a
 *                         Beeston-Barlow assertion would fire count as nan (to be stepped around), rejected points as +inf.
 * x_out [P][F], f_out [P]; flags_out [P]: 1 converged (projected gradient below gtol, or the rounding floor), 2 stalled (no
 * descent step / crawling across a kink), 4 failed (the start was not finite); counters [4]: iterations, evaluation calls,
 * of which one-sided-slope calls, evaluations (bi_fit_batched) -- or NULL. */
typedef int (*bi_objective_fn)(void* user, int64_t n, int F, const double* x, const int64_t* rows, double* f, double* g);
int bi_minimize_batched(bi_objective_fn fun, void* user, int64_t P, int F, const double* x0, const double* lo, const double* hi,
                        const int32_t* n_kinks, const double* kinks, double gtol, int max_iter, double* x_out, double* f_out,
                        int32_t* flags_out, int64_t* counters);
int bi_fit_batched(bi_ctx* ctx, int64_t P, int F, const int32_t* var_kind, const int32_t* var_index, const double* z0,
                   const double* scale0, const double* unit, const int64_t* dataset, const double* x0, const double* lo,
                   const double* hi, const int32_t* n_kinks, const double* kinks, double gtol, int max_iter, double* x_out,
                   double* f_out, int32_t* flags_out, int64_t* counters);

/* One parameter point against datasets [t0, t1): the toy-MC form.  mu_b / log mu_b are computed
 * once and every dataset reduces sum_b xlogy(n, mu) against them.  Not available with
 * Beeston-Barlow (mu then depends on the data).  out [t1 - t0]. */
int bi_eval_datasets(bi_ctx* ctx, const double* z, const double* rate_scale, int64_t t0, int64_t t1,
                     double* out, int32_t* status /*[1] or NULL*/);
/* the same with the results left in HBM (out_dev: t1 - t0 doubles of device memory, e.g. the send buffer of the
 * gather that ends a toy-MC run sharded over GPUs); returns when the kernels have finished */
int bi_eval_datasets_device(bi_ctx* ctx, const double* z, const double* rate_scale, int64_t t0, int64_t t1,
                            double* out_dev, int32_t* status /*[1] or NULL*/);

/* The toy-MC form over SEVERAL parameter points: P hypotheses against datasets [t0, t1) in one call -- what the reference's
 * toy-MC users run as a double loop, every simulated dataset (blueice/model.py:69-91 + set_data) evaluated at every hypothesis
 * (the loops of blueice/inference.py:392-443), P x T likelihood calls.  Here the points are ordered by grid cell and taken four
 * at a time: the points of a pass that share a grid cell share one pass over its 2^d S template rows (hypotheses that differ
 * in their rates only always do), and ONE pass over the datasets' non-empty-bin lists serves all four -- their log mu tiles sit
 * side by side in LDS.  Same values as P calls of bi_eval_datasets to rounding (the partial sums are grouped by tiles of 4096
 * instead of 8192 bins); every result is a sum in a fixed order, so a call is bitwise reproducible.
 *   z [P][d], rate_scale [P][S] or NULL; out [P][t1 - t0], row p = point p; status [P] or NULL: a point outside the anchor box
 *   or with unphysical rates has BI_ST_OUT_OF_BOUNDS / BI_ST_UNPHYSICAL and a row of -inf (likelihood.py:345-347, 397-415).
 * Data that the multi-point kernels do not cover (dense counts only, fewer than 64 datasets, counts that fit no list format)
 * are answered point by point through bi_eval_datasets; not available with Beeston-Barlow.  P <= 65535.
 * _device: out_dev [P][t1 - t0] in HBM (the send buffer of a gather when the hypotheses are dealt over GPUs). */
int bi_eval_datasets_points(bi_ctx* ctx, int64_t P, const double* z, const double* rate_scale, int64_t t0, int64_t t1,
                            double* out, int32_t* status /*[P] or NULL*/);
int bi_eval_datasets_points_device(bi_ctx* ctx, int64_t P, const double* z, const double* rate_scale, int64_t t0, int64_t t1,
                                   double* out_dev, int32_t* status /*[P] or NULL*/);

/* ---- compatibility mode: materialise what the morpher closures return ---------------------
 * `ps_interpolator(zs)` / `mus_interpolator(zs)` / `n_model_events_interpolator(zs)`
 * (pdf_morphers.py:70) and `full_output=True` (likelihood.py:424-425).
 *   which: 0 = ps [S][B], 1 = mus [S], 2 = n_model row [B]
 * returns BI_ERR_INVALID for out-of-bounds z (scipy raises ValueError, bounds_error=True). */
int bi_interpolate(bi_ctx* ctx, int which, const double* z, double* out);
/* adjusted (mus [S], ps [S][B]) after rate scaling and Beeston-Barlow, as full_output returns */
int bi_eval_full(bi_ctx* ctx, const double* z, const double* rate_scale, int64_t dataset, double* ll,
                 double* mus_out, double* ps_out, int32_t* status);

/* ---- asynchronous / device-resident form (inputs already in HBM, no host round trip) -------
 * bi_plan_points does the host-side part of bi_eval once (cell lookup, rates, grouping) and
 * keeps the plan on the device; bi_run_plan launches the kernels on the context stream and
 * writes to a DEVICE buffer of P doubles (e.g. a torch / RCCL tensor's data pointer);
 * bi_sync waits for the stream. */
typedef struct bi_plan bi_plan;
int bi_plan_points(bi_ctx* ctx, int64_t P, const double* z, const double* rate_scale, const int64_t* dataset,
                   bi_plan** out);
int bi_run_plan(bi_ctx* ctx, bi_plan* plan, double* out_dev /* NULL: internal buffer */);
/* A scan dealt over several GPUs ("sort/group by grid cell first, then deal cells to ranks"; the
 * reference's counterpart is the double loop of blueice/inference.py:424-432).  Every rank passes the SAME P points; the
 * device planner's (cell, dataset) sort IS the dealing: rank `share_rank` of `share_world` takes a contiguous, balanced
 * range [lo, hi) of the sorted list of valid points, so the points of a grid cell stay together (a cell is split over at
 * most two neighbouring ranks) and no host pass over the points is needed.  bi_run_plan then writes hi - lo results in
 * SORTED order to out_dev[0 .. hi - lo); after the ranks' vectors have been gathered into [share_world][stride]
 * (stride >= ceil(n_valid / share_world)), bi_plan_unsort scatters them into the caller's point order, full_dev [P]
 * (rejected points: -inf), on the context's stream.  Beeston-Barlow models too (blueice/likelihood.py:618-660: work items of
 * bb_max_group points, N(z) from the per-anchor totals), unless some point of the batch can have a bin with U_b == 0 -- there
 * the reference's first-root assertion hangs on the last bits of N, which only the host planner's pass in numpy's summation
 * order reproduces: BI_ERR_INVALID then, as for infinite rate scales (the caller deals such a scan on the host). */
int bi_plan_points_share(bi_ctx* ctx, int64_t P, const double* z, const double* rate_scale, const int64_t* dataset,
                         int share_rank, int share_world, bi_plan** out);
int bi_plan_share_info(const bi_plan* plan, int64_t* n_valid, int64_t* lo, int64_t* hi);
/* The same with the points ALREADY IN HBM: z_dev [P][d], rate_scale_dev [P][S] or NULL, dataset_dev [P] or NULL are device
 * arrays on the context's GPU (bi_device_alloc + bi_memcpy_to_device, or any producer that has finished writing them) and are
 * read where they lie -- no host pass, no copy; they may be freed or overwritten as soon as the call returns.
 * share_world = 1: the whole batch (bi_plan_points); > 1: this rank's share of a dealt scan (bi_plan_points_share).
 * Always planned on the device, so the restrictions of the device planner apply as errors instead of a silent host path:
 * no point with an infinite rate of a source that may go negative (likelihood.py:403-415: those are answered on the host),
 * no Beeston-Barlow point that needs exact totals (see bi_plan_points_share), P <= 2^30; BI_ERR_INVALID as well for a
 * pointer that is not device memory of this GPU.
 * The batched callers of the reference hold their points in numpy arrays (inference.py:424-432), so this entry has no
 * counterpart there: it is for drivers that produce hypotheses on the device or reuse one grid for many datasets. */
int bi_plan_points_resident(bi_ctx* ctx, int64_t P, const double* z_dev, const double* rate_scale_dev,
                            const int64_t* dataset_dev, int share_rank, int share_world, bi_plan** out);
int bi_plan_unsort(bi_ctx* ctx, bi_plan* plan, const double* gathered_dev, int64_t stride, double* full_dev);
int bi_plan_read(bi_ctx* ctx, bi_plan* plan, double* out /*[P]*/, int32_t* status /*[P] or NULL*/);
/* The bitwise OR of the plan's per-point status words after the last bi_run_plan (waits for the stream): what a caller
 * that leaves the results in HBM (bi_run_plan(out_dev) -> collective) must look at before it trusts them.
 * BI_ST_INTERNAL in it means a launch gave up collecting a partial sum (result nan): the context's mailbox is
 * emptied again here, so the next launch starts clean. */
int bi_plan_status(bi_ctx* ctx, bi_plan* plan, int32_t* status_or);
int64_t bi_plan_bytes(const bi_plan* plan);    /* algorithmic HBM bytes one bi_run_plan moves */
int64_t bi_plan_launches(const bi_plan* plan); /* morph+reduce launches per bi_run_plan */
void bi_plan_destroy(bi_ctx* ctx, bi_plan* plan);
int bi_sync(bi_ctx* ctx);
void* bi_stream(bi_ctx* ctx); /* the hipStream_t the context launches on */

/* Plain device buffers on the context's GPU, for callers that keep results in HBM between bi_run_plan and a
 * collective (the per-rank result vectors that RCCL gathers over xGMI at the end of a sharded scan: SURVEY.md
 * section 8e; the reference has no counterpart, its scans are Python loops, blueice/inference.py:424-432).
 * The copies run on the context stream and return when the data has arrived. */
int bi_device_alloc(bi_ctx* ctx, int64_t bytes, void** out);
int bi_device_free(bi_ctx* ctx, void* p);
int bi_memcpy_to_host(bi_ctx* ctx, void* dst_host, const void* src_dev, int64_t bytes);
int bi_memcpy_to_device(bi_ctx* ctx, void* dst_dev, const void* src_host, int64_t bytes);

/* self-test of the device logarithm used in the per-bin terms: out[i] = log(x[i]) as the kernels compute it */
int bi_selftest_log(bi_ctx* ctx, int64_t n, const double* x, double* out);

/* self-tests of the library's own device-wide primitives (csrc/tu_prim.hip: the stable radix sort of (64-bit key, value) pairs over
 * the key bits [begin_bit, end_bit) and the scans behind the device planner, the list builders and toy generation), on caller data:
 *   sort kind 0: uint64 keys, int64 values | 1: int64 keys, int32 values | 2: double keys (numpy's sort order, -nan < -inf ... +inf < +nan), int32 values
 *   scan kind 0: inclusive running maximum of int64 | 1: inclusive sum of int64 | 2: inclusive sum of double | 3: exclusive sum of int64, from `init` */
int bi_selftest_sort(bi_ctx* ctx, int kind, int64_t n, const void* keys, const void* vals, int begin_bit, int end_bit, void* keys_out,
                     void* vals_out);
int bi_selftest_scan(bi_ctx* ctx, int kind, int64_t n, const void* in, int64_t init, void* out);

/* ---- measurement ---------------------------------------------------------------------------
 * While enabled, every morph+reduce launch is bracketed by HIP events on the context stream;
 * bi_profile_read drains them: number of launches and summed GPU time in milliseconds.
 * bi_measure_read_bandwidth: the read-only streaming ceiling of the device (SURVEY.md 8d asks for it beside the
 * 8 TB/s vendor figure) -- a plain 16-byte-load sum over the resident template tensor, best of `reps` passes,
 * in GB/s; `nontemporal` selects the load hint, `blocks_per_cu` the grid.
 * bi_measure_copy_bandwidth: the device-to-device copy rate (bytes read + bytes written per second, GB/s) of the runtime's
 * own copy over the first `bytes` of the template tensor into a scratch buffer -- the other usual ceiling. */
int bi_measure_read_bandwidth(bi_ctx* ctx, int nontemporal, int blocks_per_cu, int reps, double* gb_per_s);
int bi_measure_copy_bandwidth(bi_ctx* ctx, int64_t bytes, int reps, double* gb_per_s);
/* the same read-only ceiling with the morph kernel's own access pattern: `items` work items that each stream `rows`
 * template rows concurrently (16 bytes per lane per row, XCD-aware tile order, the grid a batched launch would get)
 * and do nothing with them -- what k_morph_reduce's GB/s is to be held against; best of `reps` passes */
int bi_measure_stream_bandwidth(bi_ctx* ctx, int items, int rows, int nontemporal, int blocks_per_cu, int reps, double* gb_per_s);
int bi_profile_enable(bi_ctx* ctx, int on);
int bi_profile_read(bi_ctx* ctx, int64_t* n_launches, double* total_ms);
/* Tunables (bi_set_param; all have measured defaults) and read-only counters (bi_get_param):
 *   sparse            0 every bin visited | 1 non-empty-bin form when exact (templates >= 0, default) | 2 force
 *   max_group         points per cell pass: 1, 2, 4, 8, 16 (default 16)
 *   blocks_per_cu     resident blocks per CU the launch shapes aim for (8)
 *   nt_loads          0 never | 1 always | 2 nontemporal template loads when no two items share an anchor (default)
 *   tile_chunks       XCD-aware tile order: contiguous regions per row (8; 1 = plain order)
 *   single_kernel, fuse_max_blocks   one-launch path of single evaluations, in-launch finish up to this many blocks
 *   single_blocks_per_cu   single evaluations: blocks per CU the launch shape aims for, equal tiles per block (4)
 *   fuse_finish       batched launches of up to 256 work items whose partial sums fit the context's mailbox (2^20 slots):
 *                     the last block of an item collects the item's partials inside the launch, no finish launch follows (1)
 *   keep_rows         single dense evaluations repeated in one cell: stream rows that keep the default cache policy so
 *                     they stay in the Infinity Cache between calls (-1 = as many as fit, default; 0 = none)
 *   poll_result       single evaluations: poll the pinned result word instead of a stream synchronise (1)
 *   xcd_affine        round block counts to multiples of 8
 *   device_plan_min   batches of at least this many points are planned on the device (512)
 *   scan_mfma, scan_min_items, scan_cb, scan_waves_per_cu   the matrix-core scan kernel: on/off, items per cell from
 *                     which it is used (4; x2 for dense data), strip width (0 = by the data), waves per CU the
 *                     strips of all cells are spread over (0 = sized by the kernel's occupancy so that the blocks run in
 *                     full rounds, default)
 *   grad_mfma, grad_mfma_min, grad_slices   bi_eval_grad: batches of >= grad_mfma_min (2048) points of one dataset (plain binned likelihood, up to 32
 *                     streams) run on the fp64 matrix cores -- points grouped by grid cell, mu = rows x coefficients and
 *                     G = (n / mu) x rows^T as two chained matrix products per 16-bin block, the derivative coefficients
 *                     contracted per point afterwards (k_grad_mfma; 1, default; 0 = one work item per point); grad_slices:
 *                     slices a cell's 16-bin blocks are split into (0 = by the batch).  Read-only n_grad_mfma_launches
 *   host_threads      host threads the planner may start per call (PROCESS-wide): 0 = what the process may run on (its
 *                     scheduling affinity), at most 16 (default); a launcher of N ranks per node sets its share, so that the ranks
 *                     together never run more planner threads than the node has cores for them
 *   scan_xcd          k_scan_sorted (rows in count order): how the (cell, block) pairs of a launch are dealt to the 8 XCDs, whose
 *                     L2s each keep the coefficient lists of the cells they work on: 1 contiguous, equal ranges of the cell-major
 *                     list (default), 2 cell g on XCD g mod 8, 0 plain launch order (every cell on every XCD)
 *   bb_exact          single-point Beeston-Barlow evaluations: N(z) = sum_b n_model[i, b] in numpy's own summation order (one more
 *                     pass over the 2^d rows of MC counts), so that the root formula sees the reference's bits: 0 never, 1
 *                     always, 2 when some bin can have U_b == 0 at the point (default)
 *   toy_events        bi_generate_toys: sparse expectations (sum mu < B/8, templates and rates >= 0) are drawn event by event --
 *                     N ~ Poisson(sum mu), bins by bisection in the cumulative sums, sorted and run-length encoded per toy (1,
 *                     default); 0 = always one Poisson draw per bin.  Read-only `last_toy_method`: 1 / 0 for the last call.
 *   dot_tiled         bi_eval_datasets over non-empty-bin lists: batches of >= 64 datasets take the kernel that stages bin tiles
 *                     of log mu in LDS over tile-major lists (1, default); 0 = always one block per dataset
 *   dot_lanes         ... lanes per (dataset, tile) run of that kernel: 0 = the best measured for the lists' entry width (default: 8
 *                     lanes x three 16-byte loads = 96 entry slots with four-byte entries, 4 lanes x three loads = 96 slots with
 *                     two-byte ones), or 4 / 8 / 16 (16: 128 slots, round 3's shape)
 *   dot_entry16       ... the tile-major lists hold TWO-byte entries (bin within the tile, count) where every count of every dataset is
 *                     at most 7 -- the kernel is bound by reading the lists --, four-byte entries otherwise (1, default; 0 = always
 *                     four bytes; read-only `tm_entry_bytes`: the width of the lists as built).  With two-byte entries dot_lanes
 *                     8 and 16 mean 128 entry slots per run (8 lanes, two loads each), 4 means 96 (three loads per lane)
 *   dot_blocks_per_cu ... blocks of that kernel per CU its split of the datasets aims at: 0 = as many as are resident at once (one
 *                     round of blocks; default), n = n per CU
 *   score_sorted      bi_score_events / bi_simulate_events (set on the TARGET context): from 4096 events on, the events are ordered
 *                     by histogram cell before their densities are gathered, so the gathers stream through the histograms; the
 *                     tensor's columns are then in that order, which only sums over events can see -- bi_interpolate and
 *                     bi_eval_full return per-event values in the caller's order (1, default; read-only `events_sorted`)
 *   toy_points_pp     bi_eval_datasets_points: parameter points per pass over the datasets' lists: 0 = by the batch (4; 2 for a
 *                     call of two points), 2, 4, or 1 = point by point through bi_eval_datasets.  toy_points_lanes: lanes per
 *                     (dataset, tile) run of its kernel, 0 = the measured default, or 2 / 4 / 8.  Read-only n_toy_points_passes
 *                     (passes made so far) and tmm_entry_bytes (entry width of the 4096-bin tile lists as built, 0 = none)
 *   toy_fast_call     bi_eval_datasets, bits: 1 = the point's descriptors travel in the kernel arguments (no copy ahead of the launch),
 *                     2 = the tiled kernel's partial sums are finished 64 datasets per block, tiles split over its four waves,
 *                     4 = results of up to 4 MB: the call polls a completion word behind them instead of synchronising the
 *                     stream (needs 2 and poll_result; every 256th call synchronises anyway).  7 (default)
 *   scan_share_slow   k_scan_sorted: a strip whose 64 bins do not carry one count (two runs meet, the end of the data) is worked item by
 *                     item, ~25 times the cost of a uniform strip; with 1 (default) the items of such strips are dealt over ALL waves
 *                     of the cell instead of staying with the wave that owns the strip
 *   scan_bb, scan_bb_min   Beeston-Barlow batches planned on the device (no bin can have U_b == 0, at least scan_bb_min = 64
 *                     points) run on the fp64 matrix cores: four 16-point work items of a grid cell share the rows of a bin
 *                     tile through LDS, U, P and a are three chains of matrix products, the per-bin roots and Poisson terms follow
 *                     on the vector ALU in the reference's operation order (k_scan_bb; 1, default; 0 = k_morph_reduce<bb_max_group,
 *                     true> as for small batches).  Read-only n_bb_scan_launches
 *   plan_tables       device planner: where the (cell, dataset) keys are few (anchors x datasets < 65 536) the group structure of the
 *                     sorted points comes from tables over the KEYS (one small kernel) instead of three scans over the points (1, default)
 *   toy_points_overlap   bi_eval_datasets_points over more than 8 points: the log mu pass of the next group of passes and the finish
 *                     of the last one on a second, low-priority stream beside the dot kernel (1); measured: the kernels run side
 *                     by side but slow each other by as much -- no gain, hence off (0, default).  The same bits either way
 *   plan_count_sort   device planner: keys of at most 1024 values (a profile scan: one dataset, up to 1023 grid cells) are ordered by
 *                     a one-pass counting sort instead of the radix sort's pass per 4 bits; the same stable order (1, default)
 *   scan_chunk        matrix-core scans over few grid cells with long lists of work items (a rank's share of a dealt scan): the
 *                     item lists are cut into chunks, each worked as a group of its own, so that the chip is filled (1, default)
 *   scan_sparse_max_items   non-empty-bin form: items per grid cell up to which the matrix-core scan kernel takes the compacted rows
 *   scan_split        scans with sparse = 0 over mostly empty data: non-empty-bin pass + validity pass of every bin on the
 *                     matrix cores (k_scan_valid) instead of the per-bin terms in every bin (1)
 *   compact_budget    bytes of device memory the compacted templates of the non-empty-bin form may take
 *   toy_offset        bi_generate_toys: toy t of a call is dataset toy_offset + t of the seed's random stream, so ranks
 *                     that each generate a range of one toy-MC ensemble draw the same toys as one process would (0)
 *   scan_pow          scans over dense data on the matrix cores run on a copy of the template rows whose bins are ordered by
 *                     their count (one dataset, within compact_budget; built on first use per data upload): a lane's bins then
 *                     carry one count n, and sum n log mu over them is n log of their product -- one logarithm per lane, work
 *                     item and 32-bin strip instead of eight (1, default; 0 = rows in bin order, a logarithm per bin).
 *                     Read-only n_sorted_scans counts the scans that ran on the copy
 *   mail_timeout_ms   in-launch finish: how long an item's collecting block waits for a sibling's partial sum before it gives
 *                     up with BI_ST_INTERNAL (2000)
 *   single_timing_reset   (write) zero the single-call wall-time accumulators below
 *   bb_max_group      points per work item of a Beeston-Barlow batch (the points of a grid cell share a pass over the streams):
 *                     1, 2, 4, 8 (default) or 16 -- sixteen keep their accumulators partly in AGPRs, one wave per SIMD
 *   drop_recycle_cache   (write) give the device buffers the context keeps for reuse (plan and scratch buffers of up to
 *                     1 GiB each, 4 GiB in total; read-only recycle_cache_bytes) back to the driver now; an allocation that
 *                     fails does the same before it gives up
 *   debug_skip_post, debug_late_post   (write; fault injection for tests) block k of the NEXT launch that finishes through the
 *                     mailbox never posts its partial sum / posts it after the collector has given up; consumed by that launch
 * read-only: tile_bins, padded_bins, n_scan_launches, n_toy_polled (bi_eval_datasets calls that returned on the completion word), tm_entry_bytes, events_sorted, n_valid_launches, n_sorted_scans, n_bb_exact, n_mail_resets, user_allocations, csr_ready, compact_ready, compact_sorted (the compacted copy is ordered by count), split_ready, ps_nonneg, nnz_total;
 *            last_scan_nslots / last_valid_nslots / last_scan_resident (waves per cell the planner chose for the scan kernels of
 *            the last plan, and the resident blocks per CU it sized them by), last_toy_method (1 = event by event);
 *   single_calls, single_ns_host, single_ns_launch, single_ns_wait   wall time (ns, summed over single_calls calls) of
 *                     bi_eval(P = 1): host half (geometry, rates, descriptors), launch calls, wait for the result */
int bi_set_param(bi_ctx* ctx, const char* name, int64_t value);   /* unknown or read-only name: BI_ERR_INVALID */
int64_t bi_get_param(bi_ctx* ctx, const char* name);               /* unknown or write-only name: INT64_MIN (no parameter
                                                                      can hold it) and a message in bi_last_error */
/* every parameter name, one per line as "<name> <rw|r|w>\n", NUL-terminated, into buf (truncated to len); returns
 * the buffer size the whole list needs */
int bi_list_params(char* buf, int len);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* BLUEICE_HIP_H */
