/*
 * egdst.h -- C ABI of the MI355X-native egdst hot path (one shared library PER MODEL).
 *
 * The reference builds three MEX files per model (compile.m:781,793,805) and calls them as
 *     [M,D,dbgout] = egdst_solver(model)        @egdstmodel/egdstmodel.m:1170  -> egdst_solver.c:143 mexFunction
 *     sims         = egdst_simulator(model,rnd) @egdstmodel/egdstmodel.m:1268  -> egdst_simulator.c:47 mexFunction
 * Every entry point below is what a MEX (or any FFI) shim for those two gateways binds; the model
 * plugin (utility, budget, trpr ... compile.m:183-655) is compiled INTO the library as gfx950
 * device code, exactly as the reference compiles modelspec.c into each MEX file.  Plain pointers
 * and sizes only; all buffers are caller-owned host memory unless a name ends in _dev.
 *
 * Error convention: the reference fills a global err[300] (egdst_lib.c:299-302), returns partial
 * results and warns (egdst_solver.c:237).  Here every call returns 0 or an EGDST_E_* code, solve
 * reports a per-draw status, unsolved cells have length 0, egdst_strerror() gives the reference's
 * message for a code and egdst_last_error() the text of the last failing call of this thread.
 */
#ifndef EGDST_H
#define EGDST_H

#ifdef __cplusplus
extern "C" {
#endif

/* Run-time scalars read by parseModel on every call (egdst_lib.c:34-62) plus the quadrature
 * array built by solve (egdstmodel.m:1157-1160): ny weights followed by ny Gauss-Legendre
 * abscissae on [0,1]; the library applies the inverse normal cdf itself (egdst_solver.c:164). */
typedef struct egdst_desc {
    int t0, T;          /* first and last period; nt = T-t0+1 */
    int ngridm;         /* standard number of endogenous grid points */
    int ngridmax;       /* capacity of a cell, rows (without the a0 row) */
    int nthrhmax;       /* capacity of a threshold list */
    int ny;             /* quadrature order */
    double mmax, a0;    /* largest cash-in-hand, credit constraint */
    const double *quadrature; /* host, [2*ny] */
} egdst_desc;

/* Constants baked into this model's library (compile-time in the reference too). */
typedef struct egdst_model_info {
    int nst, nd, nnst, nnd, nparam, neq, distrib;
    int optim_MUnoD, optim_UnoD, optim_UasD, optim_TRPRnoSH;
    double tolerance, zeroconsumption, doublepoint_delta; /* cflags, egdstmodel.m:420-424 */
    const char *label;
} egdst_model_info;

typedef struct egdst_handle egdst_handle; /* device-resident batch of independent solves */

enum {
    EGDST_OK = 0,
    EGDST_E_ARG = 1,            /* bad argument / descriptor */
    EGDST_E_HIP = 2,            /* HIP runtime failure (text in egdst_last_error) */
    EGDST_E_NOGPU = 3,          /* no gfx950 device visible: there is no CPU fallback */
    /* solver errors; same conditions and texts as the reference's error() calls */
    EGDST_E_INTERP2 = 10,       /* egdst_lib.c:172 */
    EGDST_E_TRPR_SUM = 11,      /* egdst_solver.c:580 */
    EGDST_E_NO_SAVINGS = 12,    /* egdst_solver.c:589 */
    EGDST_E_GRID_FULL = 13,     /* egdst_solver.c:662,1336,1378,1413,1456,1890,1912 */
    EGDST_E_EMPTY_CHOICESET = 14, /* egdst_solver.c:700 */
    EGDST_E_ALL_NEG_INF = 15,   /* egdst_solver.c:707 */
    EGDST_E_ENVELOPE = 16,      /* egdst_solver.c:726 */
    EGDST_E_ENV2_SPACE = 17,    /* egdst_solver.c:823,884 */
    EGDST_E_ENV2_SEGMENTS = 18, /* egdst_solver.c:832 */
    EGDST_E_ADRAW_INIT = 19,    /* egdst_solver.c:1018 */
    EGDST_E_THRH_FULL = 20,     /* egdst_solver.c:1327,1891 */
    EGDST_E_TWO_ANALYTIC = 21,  /* egdst_solver.c:1691 */
    EGDST_E_BRACKET_OUT = 22,   /* egdst_solver.c:1938 */
    EGDST_E_BRACKET_REV = 23,   /* egdst_solver.c:1943 */
    EGDST_E_BUDGET_INVERT = 24, /* egdst_lib.c:293 */
    EGDST_E_TRPR_CASES = 25,    /* compile.m:541-544 */
    EGDST_E_RESEND_IN_GRID = 26,/* internal guard: a c1<=0 resend inside the grid stage (egdst_solver.c:1080-1099)
                                   reached the envelope without having been redone sequentially by k_fixup */
    EGDST_E_INTERNAL = 27,      /* scratch exhausted (crossing stack) */
    EGDST_E_CAPACITY = 28,      /* compact handles only: the draw needs more rows than rows_cap (no reference
                                   counterpart; solve the draw again with exact capacities) */
    /* simulator gateway (egdst_simulator.c:54-75,155,165) */
    EGDST_E_NOT_SOLVED = 40,
    EGDST_E_RAND_SHORT = 41,
    EGDST_E_SIM_STATE = 42
};

int egdst_get_model_info(egdst_model_info *out);
const char *egdst_strerror(int code);
const char *egdst_last_error(void);

/* Create a batch of `ndraw` independent solves of this model.  keep_history=1 keeps every period's
 * tables resident (needed for cell export and simulation); 0 keeps two ping-pong periods only.
 * stream: a hipStream_t (NULL = the library creates its own non-blocking stream). */
int egdst_create(const egdst_desc *desc, int ndraw, int keep_history, void *stream, egdst_handle **out);
/* The same with PHYSICAL row capacity rows_cap < ngridmax for every device list and table (0 = exact).  The
 * reference sizes each cell for ngridmax rows (egdst_solver.c:198-217) although a solved cell holds about ngridm;
 * with thousands of draws resident that sparsity costs TLB reach and cache (measured).  All of the
 * reference's limits (ngridmax in the runaway guard :963, error :662 ...) keep their logical values; a draw that
 * would write row rows_cap stops with EGDST_E_CAPACITY instead and must be solved again on an exact handle
 * (egdst_amd.runtime.Solver does that transparently).  Results of the draws that fit are bit-identical. */
int egdst_create_compact(const egdst_desc *desc, int ndraw, int keep_history, int rows_cap, void *stream,
                         egdst_handle **out);
int egdst_destroy(egdst_handle *h);
/* Draw groups.  The draws of a handle are split into `ngroups` contiguous ranges whose per-period kernels are
 * enqueued on separate HIP streams (forked from and joined to the handle's stream), so that one draw with a long
 * sequential stretch -- the reference's guess generator re-bases point after point on some parameter draws --
 * holds up its own group only.  Results do not depend on the grouping.  Default with GPU_MAX_HW_QUEUES >= 10: 4 groups from
 * 64 (draw, state) cells, 8 from 512, and 16 from 1024 cells with >= 20 queues; with the runtime's default of 4
 * hardware queues: 4 groups from 1024 cells (environment EGDST_GROUPS overrides).  At most 32, and at most ndraw. */
int egdst_set_groups(egdst_handle *h, int ngroups);
/* History-based scheduling (on by default with more than one group): after a solve, the draws whose guess streams
 * re-based more than 1000 times (degenerate parameter draws: one such stream is ~75 ms of strictly sequential work)
 * are scheduled on up to 8 extra streams of their own in the next solves, as far as hardware queues are left,
 * so that they hold up each other instead of a whole group.  Results do not depend on it. */
int egdst_set_adaptive(egdst_handle *h, int on);
/* Current schedule: regular groups, extra straggler lanes, draws currently treated as stragglers. */
int egdst_get_schedule(egdst_handle *h, int *groups, int *lanes, int *stragglers);
/* Re-basing calls of every draw's guess streams in the last solve (the straggler measure). */
int egdst_get_work(egdst_handle *h, unsigned *out /* [ndraw] */);
/* Guess streams of every draw that had to be regenerated sequentially in the last solve (zero-consumption signal inside
 * the grid stage, egdst_solver.c:1080-1099): a diagnostic of where a batch spends sequential time. */
int egdst_get_regenerations(egdst_handle *h, unsigned *out /* [ndraw] */);
/* Physical geometry of the handle: rows per list and row stride of the device tables (egdst_device_tables). */
int egdst_geometry(egdst_handle *h, int *rows_cap, int *table_stride);

/* Parameter vectors, one row per draw: params[draw*nparam + k] (loadparameters, compile.m:469-475).
 * F8 of SURVEY.md: the batch-of-draws surface is new; ndraw==1 is the reference's setparam+solve. */
int egdst_set_params(egdst_handle *h, const double *params, int ndraw);
int egdst_set_params_dev(egdst_handle *h, const double *params_dev, int ndraw);
int egdst_get_params(egdst_handle *h, double *params /* host, [ndraw*nparam] */);

/* Backward induction for all draws (egdst_solver.c:258-339).  _async only enqueues on the handle's
 * stream; egdst_sync waits and returns the first non-zero per-draw status (or 0). */
int egdst_solve_async(egdst_handle *h);
int egdst_sync(egdst_handle *h);
int egdst_solve(egdst_handle *h);
int egdst_get_status(egdst_handle *h, int *status /* [ndraw] */, int *where /* [2*ndraw] (it,ist) or NULL */);

/* EGM evaluations the reference would have executed (body at egdst_solver.c:548-570), summed over draws. */
int egdst_get_evals(egdst_handle *h, long long *total, long long *per_draw /* [ndraw] or NULL */);
/* The part of those counts that was accounted for WITHOUT being executed: when the guess generator's first stage gets
 * M = +inf back for A = mmax it asks for mmax again until its runaway guard (egdst_solver.c:963-978); the expectation is a
 * pure function of the guess, so the device executes it once and credits the repeats.  Throughput figures subtract this. */
int egdst_get_evals_credited(egdst_handle *h, long long *total, long long *per_draw /* [ndraw] or NULL */);

/* Cell export in the reference's wire layout (saveoutput, egdst_solver.c:917-952):
 *   M cell: (len x 4) column-major [M C A V], row 0 = (a0, 0, a0, evf(a0));  D cell: (thlen x 2) [D TH].
 * Requires keep_history=1.  len==0: the cell was not solved (infeasible state or earlier error). */
int egdst_cell_dims(egdst_handle *h, int draw, int it, int ist, int *len, int *thlen);
int egdst_get_cell_M(egdst_handle *h, int draw, int it, int ist, double *out /* [len*4] */);
int egdst_get_cell_D(egdst_handle *h, int draw, int it, int ist, double *out /* [thlen*2] */);
/* Bulk export of one draw: lens/thlens [nt*nst], M/C/V [nt*nst*(ngridmax+1)], D/TH [nt*nst*nthrhmax];
 * slot = it*nst+ist.  Any pointer may be NULL. */
int egdst_get_solution(egdst_handle *h, int draw, int *lens, int *thlens, double *M, double *C, double *V,
                       double *D, double *TH);

/* Solution import, the inverse of the export calls.  The reference's simulator and accessor gateways read the cell arrays
 * M and D from the model object on every call (egdst_simulator.c:61-68, egdst_call.c:28-34; the class is ConstructOnLoad,
 * egdstmodel.m:1, so a saved model is simulated without solving again): a handle that never ran egdst_solve is given the
 * cells here and then serves egdst_simulate*, egdst_call, egdst_get_cell_* and egdst_get_checksums exactly as the handle
 * that solved them.  Same layouts as the export: M cell (len x 4) column-major [M C A V] (A is ignored, it is M - C),
 * D cell (thlen x 2) [D TH]; len == 0 marks the cell unsolved (the reference's empty cell).  Rows past len are zeroed.
 * Requires keep_history=1 and egdst_set_params for the draw (the model functions read the parameters).  The draw's
 * status becomes 0.  EGDST_E_ARG if a cell exceeds ngridmax+1 rows or nthrhmax thresholds. */
int egdst_set_cell_M(egdst_handle *h, int draw, int it, int ist, int len, const double *in /* [len*4] */);
int egdst_set_cell_D(egdst_handle *h, int draw, int it, int ist, int thlen, const double *in /* [thlen*2] */);
/* Bulk import of one draw, argument for argument the arrays egdst_get_solution fills (none may be NULL). */
int egdst_set_solution(egdst_handle *h, int draw, const int *lens, const int *thlens, const double *M, const double *C,
                       const double *V, const double *D, const double *TH);

/* Forward simulation of draw `draw` (egdst_simulator.c:47-117): init [nsim x 2] column-major (state index
 * base-1, cash-in-hand), randstream uniforms, rndtype 1 = every agent reuses the head of the stream.
 * sims: [nsimout x nt x nsim] column-major, nsimout = 11+nnst+nnd+neq, NaN where the agent is dead. */
int egdst_simulate(egdst_handle *h, int draw, const double *init, int nsim, const double *randstream,
                   long long nrand, int rndtype, double *sims);

/* Simulated moments (new surface): the same simulation, but only the per-period means of the
 * simulated columns leave the device -- means[col + nout*it] over the agents that have a value in period it, counts the
 * number of those agents (nout = 11+nnst+nnd+neq columns as in egdst_simulate).  Deterministic summation order. */
int egdst_simulate_moments(egdst_handle *h, int draw, const double *init, int nsim, const double *randstream,
                           long long nrand, int rndtype, double *means /* [nout*nt] */, int *counts /* [nout*nt] */);

/* The estimation step on the device: EVERY draw of the handle is simulated with the same nsim agents
 * (init as in egdst_simulate) and the same uniforms -- common random numbers -- and per draw only moments and an
 * objective leave the kernels:
 *   means_dev [ndraw][nout*nt], counts_dev [ndraw][nout*nt]   as egdst_simulate_moments, DEVICE buffers (may be NULL)
 *   obj_dev   [ndraw]   sum over the cells with weight != 0 of weight[k] * (mean[k] - target[k])^2, k = col + nout*it
 *                       (target, weight: host, [nout*nt]); NaN for a draw that failed to solve or has an empty weighted
 *                       cell.  DEVICE buffer (e.g. a torch tensor): it is what the cross-GPU reduce (RCCL) takes.
 * Uniforms: randstream_dev (device, layout and length rules of egdst_simulate) or NULL: generated on the device by the
 * counter-based generator egdst_uniform(seed, k) below, k the index a randstream would have.  Requires keep_history=1. */
int egdst_simulate_batch_moments(egdst_handle *h, const double *init, int nsim, const double *randstream_dev,
                                 long long nrand, unsigned long long seed, int rndtype, const double *target,
                                 const double *weight, double *means_dev, int *counts_dev, double *obj_dev);
/* Uniform number k of stream `seed` (host replay of the device generator): with z = seed + (k+1)*0x9E3779B97F4A7C15,
 * z = (z ^ z>>30)*0xBF58476D1CE4E5B9, z = (z ^ z>>27)*0x94D049BB133111EB, z ^= z>>31 (splitmix64): (z >> 11) * 2^-53. */
double egdst_uniform(unsigned long long seed, unsigned long long k);

/* Model-function accessor behind egdstmodel.call (egdst_call.c:17-164; egdstmodel.m:1181-1207).  sw: 1 utility
 * (it, ist, id, consumption), 2 marginal utility (same), 3 discount (it, ist), 4 budget (it, ist, id, savings, ist1,
 * shock), 5 marginal budget (same), 6 value function from the solved tables (it, ist, cash).  args: host, [narg x ncol]
 * column-major with MATLAB's 1-based it/ist/id; res: host, [narg].  The gateway's conventions are kept: the first row
 * with an out-of-range index makes that row and every later one NaN, a wrong column count leaves zeros.  Requires
 * keep_history=1 (the value function reads the period's table). */
int egdst_call(egdst_handle *h, int draw, int sw, int narg, int ncol, const double *args, double *res);

/* Objective contributions of an estimation loop (new surface): out_dev[2*draw+{0,1}] =
 * value and consumption at the first endogenous grid point of (it=0, ist=0); NaN for failed draws.  The buffer
 * is device memory (e.g. a torch tensor) so that the cross-GPU reduce (RCCL) needs no host copy.  Enqueued on
 * the handle's stream. */
int egdst_objective_dev(egdst_handle *h, double *out_dev);
int egdst_get_objective(egdst_handle *h, double *out /* host, [2*ndraw] */);

/* Measurement: with profiling on, the launches of a solve are bracketed by HIP events on the stream they are
 * launched on (the draw groups' streams).  egdst_get_profile returns, for the nine classes
 *   0 k_probe / k_terminal      1 the grid kernel alone (k_grid_lds, k_grid_wide or k_grid)      2 k_envelope
 *   3 regeneration of guess streams (k_fixup_scan + k_fixup)
 *   4 k_tp_prep   5, 6 k_tp_sort stage 0, 1   7, 8 k_tp_walk stage 0, 1   (the envelope step's throughput path; with it on,
 *     class 2 is the k_envelope launch for the cells the path left over)
 * the summed device time in ms and the number of bracketed launches of the LAST solve (kernels of a class that was not
 * launched: 0), and the algorithmic table bytes of that solve summed over draws (24 B per table row read once per period,
 * 24 B per row written, 16 B per threshold). */
int egdst_set_profile(egdst_handle *h, int on);
int egdst_get_profile(egdst_handle *h, double *ms /* [9] */, int *launches /* [9] */, long long *algbytes);
/* The same events read per draw group (profiling on): for each of the groups of the last solve, ms from the start of the FIRST
 * group's first kernel to the end of this group's last kernel (out_ms[g]) -- how evenly the groups finish -- and the sum of this
 * group's bracketed kernel time per class (out_class_ms[g * 9 + k], may be NULL).  Returns the number of groups (<= max_groups). */
int egdst_get_group_profile(egdst_handle *h, int max_groups, double *out_ms, double *out_class_ms);

/* Diagnostics of a tripped internal guard (EGDST_E_INTERNAL and 27xx codes): 16 ints, meaning is internal. */
int egdst_get_debug(egdst_handle *h, int draw, int *out16);

/* Third output of the solver gateway, [M,D,dbgout] = egdst_solver(model) (egdst_solver.c:178-181, filled in thresholds()
 * :1866-1879, DEBUGOUT is always on in the reference): one row per kink the secondary and primary envelopes record, in the
 * order they are recorded -- it, ist, the choice whose secondary envelope it is (-1: primary), the threshold, consumption
 * left and right of it (-.9999 where it is the zero-consumption marker), |right - left|.  egdst_set_dbgout(h, 1) before
 * the solve makes the kernels keep the log (device memory: nt*ndraw*nst*min((nd+1)*nthrhmax, cap)*32 B, so meant for the
 * single-draw handle of the gateway); egdst_get_dbgout copies one draw's rows into out [cap x 7] column-major with
 * cap = nt*nst*nd*2*nt, zero rows after the *nrows recorded ones, rows beyond cap dropped, as the reference does. */
int egdst_set_dbgout(egdst_handle *h, int on);
int egdst_get_dbgout(egdst_handle *h, int draw, double *out /* [nt*nst*nd*2*nt * 7] */, int *nrows);

/* Checksums of one draw's solution, computed on the device: out[(it*nst+ist)*5 + k] = wrapping 64-bit sum over the rows i of
 * bits(x_i) * (2 i + 1) for column k in {M, C, V} of the cell's rows and {TH, D} of its thresholds (the odd weight pins the
 * order of the rows).  Lets a caller (and the parity tests at BASELINE.json's full sizes) compare whole solutions without
 * exporting them.  Requires keep_history=1. */
int egdst_get_checksums(egdst_handle *h, int draw, unsigned long long *out /* [nt*nst*5] */);

/* Diagnostics: this library's device exp (fn 0), log (1), pow (2) on host arrays x, y (y only for pow), n values.
 * include/egdst_math.h restates glibc's algorithms so that these equal the host libm bit for bit.
 * fn 3, 4: the interpolation of linter (egdst_lib.c:186-189), f0 (g1 - x) / (g1 - g0) + f1 (x - g0) / (g1 - g0), in the form the grid
 * kernels use (3: one refined reciprocal serves both quotients) and as written (4); x = [x | g0 | g1] (3n values), y = [f0 | f1] (2n). */
int egdst_math_eval(int fn, int n, const double *x, const double *y, double *out);

/* Envelope walks of the last solve per draw: out[2*draw] = walks that were cut into segments (one wave each) and merged,
 * out[2*draw+1] = walks whose segment predictions failed the check and were redone by one wave (results are the same
 * either way; environment EGDST_NOSEG=1 at create turns the cutting off). */
int egdst_get_walk_stats(egdst_handle *h, unsigned *out /* [2*ndraw] */);

/* Envelope step of the last solve per draw: out[2*draw] = (state, period) cells completed by the throughput path (five lean
 * kernels with one wave per envelope walk, used for batches of >= 512 cells per period; environment EGDST_ENV_TP=0/1 at
 * solve overrides), out[2*draw+1] = cells it left to the general kernel (an error condition, more than 64 monotone pieces
 * in a choice list, a failed draw, an infeasible state).  Results are the same either way. */
int egdst_get_tp_stats(egdst_handle *h, unsigned *out /* [2*ndraw] */);

/* Raw device views for callers that keep data resident (bench, estimation loops). */
int egdst_device_tables(egdst_handle *h, int it, const double **M_dev, const double **C_dev, const double **V_dev,
                        const int **len_dev);

#ifdef __cplusplus
}
#endif
#endif
