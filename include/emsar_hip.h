/* emsar_hip.h -- C ABI of libemsar_hip.so, the MI355X (gfx950) abundance-estimation core.
 *
 * Drop-in boundary: the reference has no plugin API; the seam is the call
 *     run_MLE_threads();                                   /root/reference/src/emsar_main.c:446
 * bracketed by construct_EUMAps() (emsar_main.c:436) and construct_FPKMfinal(round) (emsar_main.c:448),
 * which reads the globals CT, ReadCount, EUMAps, SC, ST (emsar.h:129-170) and writes FPKM[0..max_tid].
 * This library replaces that call, plus compute_iEUMA (emsar_functions.c:3218-3232) and the numeric part
 * of print_FPKMfinal (emsar_functions.c:3176-3207).  Plain C types only: a C host (ours: emsar_amd/csrc/host,
 * or the reference's emsar_main.c with the stub shown in INTEGRATION.md) links it directly.
 *
 * Conventions: every entry point returns 0 on success or a negative emsar_hip_status; nothing exits the
 * process (the reference exit(1)s, e.g. emsar_functions.c:3133).  The caller keeps ownership of all host
 * arrays; they may be freed as soon as the call returns.  One context per GPU; contexts share nothing, so
 * the -M multi-sample path (emsar_main.c:380-488) runs one host thread or process per device.
 *
 * Matrix convention: row c = one segment (a distinct tid multiset, CT[c], emsar.h:129) or one read;
 * columns = transcript ids; a tid may repeat inside a row and then counts twice, exactly as lambdap and
 * compute_iEUMA count it (emsar_functions.c:2969-2973, 3226-3228).
 */
#ifndef EMSAR_HIP_H
#define EMSAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct emsar_hip_ctx emsar_hip_ctx; /* opaque; one per GPU */

typedef enum {
    EMSAR_HIP_OK = 0,
    EMSAR_HIP_ERR_ARG = -1,        /* NULL / out-of-range argument, malformed CSR (tid outside [0,n_tx)) */
    EMSAR_HIP_ERR_NO_DEVICE = -2,  /* no HIP device, or device_id out of range */
    EMSAR_HIP_ERR_OOM = -3,        /* host or device allocation failed */
    EMSAR_HIP_ERR_HIP = -4,        /* a HIP runtime call or kernel failed (see emsar_hip_last_error) */
    EMSAR_HIP_ERR_STATE = -5,      /* call order: upload_structure -> upload_sample -> solve */
    EMSAR_HIP_ERR_NUMERIC = -6     /* NaN/Inf met in theta (infeasible input such as R>0 on an all-zero row) */
} emsar_hip_status;

/* How the structure is laid out in HBM (DESIGN.md "Data layout"). */
typedef enum {
    EMSAR_LAYOUT_AUTO = 0,   /* TILED when it applies, else CSR */
    EMSAR_LAYOUT_CSR = 1,    /* rows as given; lane-per-row walk, FP64 atomics straight to HBM/L2 */
    /* 2 was the WINDOWED layout of round 1 (4x slower than TILED, removed) */
    EMSAR_LAYOUT_TILED = 3    /* units of <= 8 slices x 768 rows with a unit-local dictionary of <= 240 transcripts, kept in LDS as the 16
                                 subset sums of every block of 4 neighbouring slots; a stored operand is 10 bits (three to a dword)
                                 and names a block and a subset of it, so one LDS gather serves every transcript of the block that
                                 a row hits; forward index for the E-step and a per-slice transposed index for the M-step (no
                                 atomics in the inner loops); single-tid rows folded into a per-transcript count; rows longer than
                                 237 tids go to a small CSR of their own; transcripts numbered by co-occurrence inside when the
                                 caller's numbering packs poorly (ids at this ABI stay the caller's) */
} emsar_hip_layout;

/* OR-ed into the layout argument of emsar_hip_upload_structure (TILED only): store rows with the same tid multiset
 * once, weighted by the sum of their members' weights -- the read -> segment collapse the reference does while it
 * counts reads (update_ReadCounts, emsar_functions.c:838-943).  Exact up to summation order; per-row inputs of
 * upload_sample / ieuma keep referring to the caller's rows. */
#define EMSAR_LAYOUT_FLAG_MERGE_ROWS 0x100

/* Replaces the solver knobs -e/-r/-i/-l/-n of the reference (emsar_main.c:86-91): the pattern search's
 * step/F epsilons have no meaning for an EM; they map to tol / max_iter. */
typedef struct {
    int32_t max_iter;     /* cap on EM passes (one pass = one sweep over the matrix); <=0 -> 100000 */
    int32_t accel;        /* 0 plain EM; 1 SQUAREM (3 passes per cycle, likelihood-safeguarded) */
    double  tol;          /* stop when max_t |dtheta_t| / (theta_t + abs_floor) < tol; <=0 -> 1e-10 */
    double  abs_floor;    /* <=0 -> 1e-6, the %lf print quantum of the reference's .fpkm */
    int32_t check_every;  /* host looks at the device convergence word every this many cycles; <=0 -> 8 */
    int32_t set_mode;     /* 0 = split the problem into its connected sets (run_MLE_threads' unit of work, emsar_main.c:446-474):
                             sets that fit a CU's LDS are solved by one workgroup each with no kernel launch per pass,
                             one-transcript sets in closed form, only the rest by the streaming passes;
                             1 = streaming passes over the whole matrix only */
    double  count_floor;  /* optional second floor, in READS: the floor of transcript t becomes max(abs_floor, count_floor/den_t).
                             Transcripts whose optimum is the boundary theta = 0 with zero gradient decay like 1/k; a floor of
                             e.g. 1e-3 inferred reads stops the solve once only such components still move.  0 = off. */
    double  zero_cut;     /* > 0: a component that is below this value AND still decreasing no longer holds the solve up.  The
                             reference prints FPKM with "%lf" (emsar_functions.c:3207): anything below 5e-7 is written as
                             0.000000, so with zero_cut = 2.5e-7 the .fpkm file is the same while the 1/k decay of boundary
                             components (optimum theta = 0 with zero gradient) stops costing tens of thousands of passes.
                             <= 0 = off (every component must meet tol). */
    double  abs_step;     /* > 0: a component counts as converged, whatever its relative change, once its change in one plain
                             EM step (in FPKM) is below abs_step * 200000 / K at pass K (K >= 1000).  Nearly flat directions
                             converge sublinearly (|dtheta_k| ~ k^-p, p >= 2): what is left after stopping at |dtheta| < a in
                             pass K is at most ~a*K, so the rule bounds the remaining drift by abs_step * 2e5 at every K --
                             2e-8 FPKM, a fiftieth of the .fpkm print quantum, for 1e-13 -- while the relative rule at 1e-10
                             keeps such components going for 10^5 passes.  <= 0 = off. */
    /* (zero_cut and abs_step apply to the streaming passes, and to the resident sets only when newton_after < 0: a set that gets
       Newton steps reaches a boundary optimum in a few of them and is held to the strict rule) */
    int32_t newton_after; /* set_mode 0, resident sets: a set that has not converged after this many passes gets one safeguarded
                             projected-Newton step (direction by matrix-free conjugate gradients, accepted only if F does not
                             fall) after every SQUAREM cycle.  0 -> 60, < 0 = never (EM / SQUAREM only). */
    int32_t reserved0;
} emsar_em_params;

typedef struct {
    int32_t iters;            /* EM passes executed */
    int32_t converged;        /* 1 if tol was met */
    double  final_delta;      /* last max_t |dtheta|/(theta+abs_floor) */
    double  loglik;           /* F(theta) = sum_c R_c log(E_c S_c) - E_c S_c, rows with E_c != 0 (Fp, emsar_functions.c:2946) */
    double  solve_ms;         /* wall time of the solve, upload excluded */
    double  kernel_ms;        /* device time of all EM passes (HIP events on the context's stream) */
    int64_t bytes_per_pass;   /* algorithmic bytes of one pass, SURVEY.md 8d: 4 nnz + P (rows+1) + W rows + 32 T */
    int64_t stored_bytes_per_pass; /* bytes the chosen layout actually streams per pass */
    /* set_mode 0 only (else 0): how the transcripts were split and what the LDS-resident sets cost */
    int32_t sets_resident;    /* connected sets solved inside one workgroup's LDS */
    int32_t sets_streamed;    /* connected sets too large for that, solved by the streaming passes */
    int32_t set_passes_max;   /* EM passes of the slowest resident set (iters = streaming passes + this) */
    int32_t sets_unconverged; /* resident sets that hit max_iter */
    int64_t set_passes_sum;   /* EM passes summed over the resident sets */
    double  sets_build_ms;    /* host time spent finding and packing the sets (once per upload_sample) */
    double  sets_kernel_ms;   /* device time of the resident-set kernels (clusters included) */
    int32_t sets_cluster;     /* connected sets solved by a cluster of 2-8 workgroups inside one launch (too large for one workgroup's LDS) */
    int32_t cluster_passes_max; /* EM passes of the slowest of them (set_passes_max covers them too) */
    double  cluster_kernel_ms;  /* device time from the first to the last cluster launch */
} emsar_em_stats;

/* ---- lifetime ---------------------------------------------------------------------------------- */
int  emsar_hip_create(emsar_hip_ctx **out, int device_id);
void emsar_hip_destroy(emsar_hip_ctx *ctx);
const char *emsar_hip_strerror(int status);
const char *emsar_hip_last_error(const emsar_hip_ctx *ctx); /* text of the last HIP failure, "" if none */

/* ---- inputs ------------------------------------------------------------------------------------
 * upload_structure: the incidence CT (emsar.h:129, built by scan_rshbucket emsar_functions.c:2135-2192) as
 * CSR; called once per rsh.  The library validates 0 <= col_idx < n_tx and monotone row_ptr on the host
 * before anything reaches a kernel. */
int emsar_hip_upload_structure(emsar_hip_ctx *ctx, int64_t n_rows, int32_t n_tx,
                               const uint64_t *row_ptr /* n_rows+1 */, const int32_t *col_idx /* nnz */,
                               int layout /* emsar_hip_layout */);

/* upload_sample: per-sample vectors; called once per alignment file (the body of the loop emsar_main.c:380).
 *   row_weight = ReadCount[c] (emsar.h:142), NULL = every row counts 1 (read-level matrix)
 *   row_E      = EUMAps[c]   (construct_EUMAps, emsar_functions.c:3148-3154), NULL = 1.0 everywhere;
 *                rows with E == 0 are outside the likelihood (emsar_functions.c:2952)
 *   den        = optional precomputed sum_c m_ct E_c per transcript; NULL = computed on the device */
int emsar_hip_upload_sample(emsar_hip_ctx *ctx, const int32_t *row_weight, const double *row_E,
                            const double *den);

/* ---- the hot path: replaces run_MLE_threads() (emsar_main.c:446) --------------------------------
 * Starts from the uniform interior point (theta = 1 where den > 0, else 0), runs EM to tol and copies
 * theta (= FPKM[], emsar.h:160) to fpkm_out[n_tx].  stats may be NULL. */
int emsar_hip_solve(emsar_hip_ctx *ctx, const emsar_em_params *p, double *fpkm_out, emsar_em_stats *stats);

/* Lower-level stepping, used by bench.py and the parity tests:
 *   reset        theta <- uniform start
 *   set/get      move theta between host and device
 *   run_passes   n plain EM passes back to back on the context's stream, no host synchronisation inside;
 *                *elapsed_ms (may be NULL) = device time between HIP events recorded on that stream around
 *                the n passes, *last_loglik_terms (may be NULL) = sum_c R_c log S_c at the input of the last pass */
int emsar_hip_reset_theta(emsar_hip_ctx *ctx);
int emsar_hip_set_theta(emsar_hip_ctx *ctx, const double *theta /* n_tx */);
int emsar_hip_get_theta(emsar_hip_ctx *ctx, double *theta /* n_tx */);
int emsar_hip_run_passes(emsar_hip_ctx *ctx, int32_t n_passes, float *elapsed_ms, double *last_loglik_terms);

/* ---- post-processing: compute_iEUMA + print_FPKMfinal arithmetic --------------------------------
 * ieuma[t] = sum over ALL rows of m_ct * row_L[c]  (adjEUMA, emsar_functions.c:3224-3231). */
int emsar_hip_ieuma(emsar_hip_ctx *ctx, const double *row_L /* n_rows */, double *ieuma_out /* n_tx */);
/* From the mean FPKM: TPM = mean*1e6/sum(mean); iReadcount = ieuma/1e3 * mean * N/1e6  (emsar_functions.c:3203,3207). */
int emsar_hip_normalise(emsar_hip_ctx *ctx, const double *mean_fpkm, const double *ieuma, int64_t total_read_count,
                        double *tpm_out, double *ireadcount_out, int32_t *ireadcount_int_out);

/* ---- per-sample effective lengths: compute_adjEUMA (emsar_functions.c:2517-2523) ----------------
 * upload_euma: EUMA_c[i], the effective position counts per fragment length of every row (emsar.h:141, read from the rsh
 *   by construct_rsh_from_rshfile, emsar_functions.c:1351-1510), row-major [n_rows][nfl], 0 where a row has no entry;
 *   once per rsh, after upload_structure.  Stored transposed in HBM ([nfl][n_rows]) so that one lane owns one row.
 * adj_euma: L_c = sum_i Wf[i] * (double)EUMA_c[i], i ascending, multiply and add rounded separately -- the reference's
 *   loop, operation for operation, so L is bit-identical to the host's.  Wf = the sample's normalised fragment-length
 *   histogram (transfer_fraglendist_to_Wf, emsar_functions.c:2503-2513).  HBM-bound: 4 * n_rows * nfl bytes per call. */
int emsar_hip_upload_euma(emsar_hip_ctx *ctx, const int32_t *euma /* n_rows * nfl */, int32_t nfl);
int emsar_hip_adj_euma(emsar_hip_ctx *ctx, const double *wf /* nfl */, double *adj_euma_out /* n_rows */);

/* ---- read -> segment collapse: the integer core of update_ReadCounts (emsar_functions.c:838-943) ----------------
 * Rows with the same MULTISET of transcript ids (order inside a row does not matter, repeats do: SURVEY.md A2) become
 * one row whose weight is the sum of its members' weights (row_weight NULL = 1 each; rows with weight 0 and empty
 * rows vanish).  Output rows are numbered by first occurrence -- the order in which the reference meets the segments
 * -- with their ids sorted ascending.  Exact: hash matches are confirmed by comparing the rows themselves.
 * The caller provides the output arrays at worst-case size (row_ptr_out n_rows+1, col_idx_out nnz, weight_out n_rows,
 * row_map_out n_rows or NULL: original row -> output row, -1 for vanished rows).  Error if a sum exceeds INT32_MAX. */
typedef struct {
    double  kernel_ms;          /* device time of the collapse kernels (HIP events), transfers excluded */
    double  total_ms;           /* wall time of the call */
    int64_t n_rows, nnz, n_unique, nnz_unique;
    int64_t table_slots;        /* LDS table slots over all partitions (one workgroup each) of the largest round */
    int64_t algorithmic_bytes;  /* CSR read twice (hash, compare) + weights + the unique rows written */
    int64_t rounds;             /* 1 unless rows had to be hashed again (a 64-bit hash collision, a crowded partition table) */
} emsar_hip_collapse_stats;
int emsar_hip_collapse_rows(emsar_hip_ctx *ctx, int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                            const int32_t *row_weight, int64_t *n_unique_out, uint64_t *row_ptr_out, int32_t *col_idx_out,
                            int32_t *weight_out, int32_t *row_map_out, emsar_hip_collapse_stats *stats);

/* ---- deterministic mode ------------------------------------------------------------------------------------------
 * The streaming passes add with floating atomics, so two runs of the same solve agree to ~1e-9 relative, not bit for bit (the
 * per-set solver of set_mode 0 has no atomics and is reproducible either way).  With the mode on, every sum that workgroups share
 * is kept as a 64-bit integer in fixed point (integer adds commute): the M-step accumulators hold the reads assigned to a
 * transcript at a resolution of N * 2^-61 reads (N = the sample's total weight), the log-likelihood sums at N * 2^-51, the
 * SQUAREM norms are added up in a fixed order.  Two solves of the same input are then bit-identical, whatever the layout's
 * tile order.  Costs a few percent of a pass.  Default: off, or EMSAR_HIP_DETERMINISTIC=1 in the environment at create time. */
int emsar_hip_set_deterministic(emsar_hip_ctx *ctx, int on);

/* ---- introspection ------------------------------------------------------------------------------ */
typedef struct {
    int64_t n_rows, nnz;
    int32_t n_tx;
    int32_t layout;            /* layout in use (flags included) */
    int64_t n_chunks;          /* TILED: tiles (one workgroup each, or one per pair of tiles) */
    int64_t n_slices;          /* TILED: 768-row slices (one wavefront at a time) */
    int64_t padded_entries;    /* stored forward slots incl. padding */
    int64_t far_entries;       /* entries outside their tile's contiguous tid range */
    int32_t window;            /* transcripts per dictionary */
    int32_t device_id;
    int64_t bytes_per_pass;        /* SURVEY.md 8d formula */
    int64_t stored_bytes_per_pass; /* what the layout streams */
    int64_t tiled_entries;         /* TILED: stored operands without padding (an operand = a block of 3 dictionary slots + a subset) */
    int64_t tiled_ids;             /* TILED: transcript ids of the tiled rows; tiled_ids / tiled_entries = ids served per operand */
    int64_t n_units;               /* TILED: workgroups of k_pass_tiled_unit (tiles that share a dictionary) */
    int32_t renumbered;            /* TILED: 1 = the library numbered the transcripts by co-occurrence (theta / den are mapped at this ABI) */
    int32_t reserved0;
} emsar_hip_info;
int emsar_hip_get_info(const emsar_hip_ctx *ctx, emsar_hip_info *out);

/* Host-only diagnostic (no HIP call, works without a GPU): build the TILED layout for a CSR (forward index, transposed index,
 * dictionaries, folded and leftover rows), check every descriptor against the arrays it indexes, decode the layout
 * again and check that it stores exactly the input rows. */
int emsar_hip_layout_selfcheck_tiled(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                                     int merge_rows, emsar_hip_info *info_out);

/* The same for the set-resident solver's records (set_mode 0): find the connected sets of the rows with weight > 0
 * (row_weight NULL = every row counts 1), pack the ones that fit a workgroup's LDS and check the records against the CSR. */
typedef struct {
    int64_t n_components;        /* connected sets with at least two transcripts */
    int64_t sets_resident[3];    /* by workgroup class: 64 / 256 / 512 threads */
    int64_t max_lds_bytes[3];    /* largest LDS footprint in each class */
    int64_t sets_streamed;       /* sets too large for one workgroup */
    int64_t tids_closed, tids_resident, tids_streamed;
    int64_t rows_in, rows_stored; /* weighted multi-transcript rows before / after merging identical ones */
    int64_t sets_cluster, tids_cluster, max_lds_cluster;   /* sets packed for a cluster of workgroups, their transcripts, the largest LDS footprint of one of their workgroups */
} emsar_hip_sets_info;
int emsar_hip_sets_selfcheck(int64_t n_rows, int32_t n_tx, const uint64_t *row_ptr, const int32_t *col_idx,
                             const int32_t *row_weight, emsar_hip_sets_info *info_out);

#ifdef __cplusplus
}
#endif
#endif /* EMSAR_HIP_H */
