// egdst_device.h -- device-side numerics of the EGM step for gfx950 (wave64).
//
// Everything here is fp64 and follows the arithmetic of the reference expression by expression
// (two divisions per interpolation etc.), so that results stay within 1e-10 relative of the CPU
// path; this translation unit is built with -ffp-contract=off (no FMA contraction) because a
// contracted a*b+c can flip discrete decisions at ties in the envelopes.
//
// Reference lines implemented by each function are cited inline as file:line of
// /root/reference/@egdstmodel/.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define MS_FN static __device__ __forceinline__
#define MS_TABLE static __device__ const
#include "modelspec.h"  // pulls in include/egdst_math.h (MS_EXP/MS_LOG/MS_POW)

#define EG_TOL MS_TOLERANCE
#define EG_ZEROC MS_ZEROCONSUMPTION
#define EG_DPD MS_DOUBLEPOINT_DELTA
#define EG_A0T 0.0  // egdst_solver.c:49

// ---- problem geometry shared by all kernels -------------------------------------------------
struct Geom {
    int t0, T, nt, ngridm, ngridmax, nthrhmax, ny, ndraw, nslots, S;  // S = ngridmax+1 (logical rows of a table)
    // PHYSICAL capacities of the device arrays.  Exact mode: Cp = ngridmax, Sp = S.  Compact mode
    // (egdst_create_compact): Cp < ngridmax rows per list, so the arrays of many draws stay dense in memory; a
    // draw that needs more rows than Cp stops with EGDST_E_CAPACITY and is solved again in exact mode by the host.
    int Cp, Sp;
    double mmax, a0;
};

// Device-resident state of a batch (all pointers are device memory).
struct Batch {
    Geom g;
    int draw0;           // first schedule slot of the group this launch works on (groups run on their own streams)
    const int *order;    // [ndraw] schedule: slot -> draw (identity until the host re-balances, egdst_host.inc)
    struct Env1Tile *e1tiles;  // single-choice models: per (slot, state, tile) descriptors of k_env1
    unsigned *e1done;          // [ndraw*MS_NST] tiles of the cell that have finished (counts on over the periods of a solve)
    int *tsorted;        // [ndraw*MS_NST] stamp of the last period in which k_sortcheck found the next-period table of (draw, state) out of order
    int *negflag;        // [(draw*MS_NST+ist)*MS_ND+id] a grid point of the stream signalled c1<=0 (set by k_grid)
    int *fixn;           // [MAX_GROUPS * nt] streams listed for k_fixup per (group, period)
    int *fixlist;        // [ndraw*MS_NST*MS_ND] the lists, a group's at its first slot
    unsigned *work;      // [ndraw] re-basing calls of the draw's guess streams in this solve (straggler detection)
    int sorted_valid;    // != 0: k_sortcheck runs before the kernels of every period and stamps tsorted[cell] with sorted_valid + it
                         // when the next-period table of the cell is NOT in order (eg_tab_sorted); the base differs from solve to solve
    unsigned *nregen;    // [ndraw] guess streams of the draw that k_fixup regenerated in this solve (schedule: such draws share groups)
    const double *par;   // [ndraw][MS_NPARAM]
    const double *qw;    // [ny] weights
    const double *qz;    // [ny] standard-normal nodes (Acklam of the GL abscissae, egdst_solver.c:164)
    // period tables: index ((slot*ndraw+draw)*MS_NST+ist)
    double *tM, *tC, *tV;   // * Sp
    double *tD, *tTH;       // * nthrhmax
    int *tlen, *tthlen;     // rows incl. the a0 row (0 = unsolved), thresholds
    int *defer;             // [ndraw*MS_NST] cell left for the large-LDS pass of k_envelope
    int *thw, *thhw;        // rows / thresholds of the cell that may be non-zero (>= tlen, tthlen)
    // candidates of the EGM step: index (((draw*MS_NST+ist)*MS_ND+id)*Cp + n); n=0 is the probe's point
    double *cM, *cC, *cV;       // M (for a point that is not normal: what it reports back to the guess generator), C, V
    int *cSt;                   // status and evaluations done for the point, packed (egdst_kernels.hip: eg_sc_pack)
    struct ProbeOut *probe;     // [(draw*MS_NST+ist)*MS_ND+id]
    // envelope workspaces per (draw,ist): W = (MS_ND+1)*Cp entries
    double *pM, *pC, *pV;  int *pF;    // per-choice lists, consecutive (the reference's mgridvecs)
    double *sM, *sC, *sV;  int *sF;    // secondary-envelope input (pieces with their extrapolation points)
    double *qM, *qC, *qV;  int *qF;    // points sorted in comp1 order
    int *rank;                          // [W] sorted position of each input point
    int *gcls;                          // [W] class words of a stream that is sorted in global memory
    double *eM, *eV, *eC;               // [Cp] output of a secondary envelope
    double *eTH, *eIX;                  // [MS_ND][nthrhmax] thresholds of a secondary envelope (not used afterwards)
    // hand-over from the per-choice workgroups of k_envelope (part 1) to the primary envelope (part 2), per (cell, choice):
    int *secn, *secact, *secerr;        // rows of the choice's list, choice active, first error of the job
    double *secev;                      // expected value at a0 of the choice
    unsigned long long *secevals;       // evaluations counted by the job
    // throughput path of the envelope step (k_tp_*, batches with many cells per period): per (cell, choice) and per cell records
    struct TpRec *tprec;                // [ndraw*MS_NST*MS_ND]
    struct TpCell *tpcell;              // [ndraw*MS_NST]
    unsigned *tpstat;                   // [2*ndraw] cells of the draw the throughput path completed / left to k_envelope, this solve
    int *tpn, *tplist;                  // [MAX_GROUPS * nt] cells per (group, period) left to k_envelope; [ndraw*MS_NST] their slots, a group's at its first
    int *tpbign, *tpbiglist;            // the same for the cells of the second tier of stage 1 (lists beyond the regular stream budget, k_tp_*_big)
    // status
    int *status;          // [ndraw] first error code
    int *where;           // [2*ndraw] (it, ist) of that error
    unsigned long long *evals;  // [ndraw]
    unsigned long long *credited;  // [ndraw] part of evals that was accounted for without being executed (stage-0 fixed point)
    int *dbg;             // [16*ndraw] diagnostics of a tripped internal guard
    double *klog;         // kink log (the gateway's dbgout): [(it*ndraw+draw)*MS_NST+ist][kcap][4], or nullptr (egdst_set_dbgout)
    int *kcnt, kcap;      // kinks recorded per cell; capacity of a cell's slice
    unsigned *segstat;    // [2*ndraw] walks of the draw that were cut into segments and merged / that fell back to one wave
    int noseg;            // 1: envelope walks are never cut into segments (environment EGDST_NOSEG at create; diagnostics)
    double *obj;          // [2*ndraw] staging of k_objective for the host-returning entry point
    unsigned long long *algbytes;  // [ndraw] compulsory table traffic: 24 B per next-period row read once per
                                   // period + 24 B per row written + 16 B per threshold
};

// Kernels do not take the Batch by value: ~600 bytes and ~70 pointers in the kernel-argument segment are loaded into SGPRs
// at kernel entry and stay live (hipcc 7.2: 765 SGPR spills in k_envelope, 267 in k_fixup, 127 in k_probe -- every spilled
// SGPR is a lane of a VGPR in kernels that have no VGPR to spare).  The host keeps one copy per draw group in device memory
// (egdst_host.inc: upload_batches) and kernels read it through the CONSTANT address space: every field is a scalar load
// (s_load) issued where it is used, invariant, so the register allocator re-loads instead of spilling.
#ifdef EGDST_EMU
typedef const Batch &BatchRef;
#define EG_BATCH_REF(bp) (*(bp))
#else
typedef const __attribute__((address_space(4))) Batch &BatchRef;
#define EG_BATCH_REF(bp) (*(const __attribute__((address_space(4))) Batch *)(bp))
#endif

struct Env1Tile;
#define TP_NF 64   // functions (choices, or the monotone pieces of one choice list) the throughput path of the envelope step keeps
struct TpRec {     // what k_tp_prep found for one (cell, choice); k_tp_walk (secondary) updates cnt
    int active;    // the choice is in the choice set
    int cnt;       // rows of the choice's list in its slice of the p arrays
    int nfold;     // > 0: the list folds back nfold times -- its pieces (with their extrapolation points) are in the s slice
    int fused;     // the sort classified the stream (blk_rank_sort: *fused)
    double evfa0;
    unsigned long long evals;
    int fstart[TP_NF];  // first point of every piece in the s slice (function ids id .. id+nfold)
};
struct TpCell {    // primary envelope of a cell: what k_tp_sort hands to k_tp_walk
    int npts, fused;
};
struct ProbeOut {
    int active;       // choice is in the choice set (and the state feasible)
    int seq;          // 1: the whole stream was generated sequentially by k_fixup (candidates 0..np-1, all kept)
    int np;           // 1 if the probe stored a kept point at candidate index 0
    int grid;         // 1 if the guess generator entered the grid stage (points n>=1 may be requested)
    int ncalls;       // calls made before the first grid point (for the runaway guard, egdst_solver.c:963)
    int ntogenerate;
    int probe_evals;
    double lim1, lim2, lim3, lim3p, k3;   // egdst_solver.c:1051-1099
    double A0;        // last guess before the grid stage
    double M0;        // M returned for it
    double evfa0;     // egdst_solver.c:430,593,643
};

// ---- inverse normal cdf (Acklam), egdst_lib.c:435-519 ---------------------------------------
__host__ __device__ inline double eg_inv_normal_cdf(double p)
{
    const double a0c = -3.969683028665376e+01, a1c = 2.209460984245205e+02, a2c = -2.759285104469687e+02,
                 a3c = 1.383577518672690e+02, a4c = -3.066479806614716e+01, a5c = 2.506628277459239e+00;
    const double b0c = -5.447609879822406e+01, b1c = 1.615858368580409e+02, b2c = -1.556989798598866e+02,
                 b3c = 6.680131188771972e+01, b4c = -1.328068155288572e+01;
    const double c0c = -7.784894002430293e-03, c1c = -3.223964580411365e-01, c2c = -2.400758277161838e+00,
                 c3c = -2.549732539343734e+00, c4c = 4.374664141464968e+00, c5c = 2.938163982698783e+00;
    const double d0c = 7.784695709041462e-03, d1c = 3.224671290700398e-01, d2c = 2.445134137142996e+00,
                 d3c = 3.754408661907416e+00;
    if (p < 0 || p > 1) return 0.0;
    if (p == 0) return -HUGE_VAL;
    if (p == 1) return HUGE_VAL;
    if (p < 0.02425) {
        double q = sqrt(-2 * MS_LOG(p));
        return (((((c0c * q + c1c) * q + c2c) * q + c3c) * q + c4c) * q + c5c) /
               ((((d0c * q + d1c) * q + d2c) * q + d3c) * q + 1);
    }
    if (p > 0.97575) {
        double q = sqrt(-2 * MS_LOG(1 - p));
        return -(((((c0c * q + c1c) * q + c2c) * q + c3c) * q + c4c) * q + c5c) /
               ((((d0c * q + d1c) * q + d2c) * q + d3c) * q + 1);
    }
    double q = p - 0.5, r = q * q;
    return (((((a0c * r + a1c) * r + a2c) * r + a3c) * r + a4c) * r + a5c) * q /
           (((((b0c * r + b1c) * r + b2c) * r + b3c) * r + b4c) * r + 1);
}

// ---- shocks (egdst_lib.c:66-100) -------------------------------------------------------------
MS_FN double eg_shock_node(const ms_env *E, const ms_pv *cur, const ms_pv *nxt, double z)
{
#if MS_DISTRIB == 1
    return MS_EXP(ms_mu(E, cur, nxt) + z * ms_sigma(E, cur, nxt));
#else
    return ms_mu(E, cur, nxt) + z * ms_sigma(E, cur, nxt);
#endif
}
MS_FN double eg_shock_mean(const ms_env *E, const ms_pv *cur, const ms_pv *nxt)
{
#if MS_DISTRIB == 1
    return MS_EXP(ms_mu(E, cur, nxt) + ms_sigma(E, cur, nxt) * ms_sigma(E, cur, nxt) / 2);
#else
    return ms_mu(E, cur, nxt);
#endif
}
MS_FN double eg_shock_uniform(double u, double mu, double sigma)
{
#if MS_DISTRIB == 1
    return MS_EXP(sigma * eg_inv_normal_cdf(u) + mu);
#else
    return sigma * eg_inv_normal_cdf(u) + mu;
#endif
}

#ifdef EGDST_EMU
#define EG_LDS_AS
#else
#define EG_LDS_AS __attribute__((address_space(3)))
#endif
typedef EG_LDS_AS double eg_ldsd;
typedef EG_LDS_AS int eg_ldsi;
typedef EG_LDS_AS unsigned short eg_ldss;  // sorted positions and function ids of an LDS-resident stream (< 65536)

// ---- bracket search + interpolation (egdst_lib.c:123-206) -----------------------------------
// kind 0: bracket for interpolation/extrapolation; kind 1: last threshold <= x.
template <class P> static __device__ __forceinline__ int eg_bracket(double x, P g, int n, int kind)
{
    if (x < g[1]) return 0;
    if (kind == 0 && x >= g[n - 2]) return n - 2;
    if (kind == 1 && x >= g[n - 1]) return n - 1;
    int lo = 1, hi = n - 2;
    const bool asc = g[0] <= g[n - 1];
    while (hi - lo > 1) {
        int mid = (hi + lo) / 2;
        if (asc && g[mid] > x)
            hi = mid;
        else
            lo = mid;
    }
    return lo;
}

// eg_bracket(x, g, n, 0) on a NON-DECREASING column without NaN (the caller checked the order): bxsearch_common (egdst_lib.c:136-166)
// returns 0 below g[1], n-2 from g[n-2] on, and between them the last row that is not above x -- on an ordered column that is
// "the last row i in [1, n-2] with g[i] <= x, or 0", however it is found.  Here: the branch-free form of the bisection -- the
// range [base, base+len) shrinks by len/2 per step whatever the comparison says, so every lane makes the SAME number of steps (n
// is uniform over the workgroup) and a step is one LDS read, one compare and one select.  The reference's loop compiled to ~22
// instructions per step, 13 of them scalar bookkeeping of the lanes that had finished (profiles/r04_*: k_grid_lds_cv 5876
// instructions, 2981 scalar).  A NaN x fails every comparison of the reference and its bisection runs up to row n-3: the same here.
template <class P> static __device__ __forceinline__ int eg_last_le(double x, P g, int nrows)  // last row of [0, nrows) with g <= x, or 0
{
    P p = g;  // (the position as a pointer: one add per step, no index-to-address shift)
    for (int len = nrows; len > 1;) {
        const int half = len >> 1;
        P q = p + half;
        const double gj = *q;
        p = (gj <= x) ? q : p;
        len -= half;
    }
    return (int)(p - g);
}
template <class P> static __device__ __forceinline__ int eg_bracket_sorted(double x, P g, int n)
{
    const int i = eg_last_le(x, g, n - 1);  // rows 0 .. n-2
    return (x != x) ? n - 3 : i;
}

static __device__ __forceinline__ double eg_lerp(double x, double g0, double g1, double f0, double f1)
{
    return f1 * (x - g0) / (g1 - g0) + f0 * (g1 - x) / (g1 - g0);
}

// The same expression for the grid kernels, which are bound by the instructions they issue: its two divisions have ONE divisor.
// hipcc expands an fp64 division into v_div_scale x2, v_rcp, two Newton-Raphson steps on the reciprocal (four fma), a product, a
// residual, v_div_fmas and v_div_fixup -- thirteen instructions, correctly rounded; the scale and fix-up steps are the identity
// unless an operand is zero, infinite or NaN, the residual would underflow (|numerator| < 2^-969) or the quotient or the reciprocal
// leave the normal range.  For a divisor of at least 2^-300 (rows of an M column are at most a few thousand apart and, where they
// differ at all, at least an ulp of an O(1) number apart) the refined reciprocal is computed ONCE here and each quotient is the
// product, the residual and the final fma of that very sequence: the same operations on the same operands, so the same bits
// (checked against the oracle's IEEE divisions by every parity test: ~10^10 quotients per run; a numerator is f * (x - g) with f a
// consumption or a finite value and x - g exact zero or at least an ulp of O(1): never a nonzero number below 2^-969).  A smaller,
// zero, negative or NaN divisor and any result that is not finite take the plain expression.  (A zero numerator gives +0 where the
// division gives the numerator's sign: the sum of the two terms, the test c1 <= 0 and the sums the value enters do not see it.)
#ifndef EG_LERP_SHARED
#define EG_LERP_SHARED 1
#endif
static __device__ __forceinline__ double eg_lerp_fast(double x, double g0, double g1, double f0, double f1)
{
#if defined(EGDST_EMU) || !EG_LERP_SHARED
    return eg_lerp(x, g0, g1, f0, f1);
#else
    const double d = g1 - g0, n1 = f1 * (x - g0), n0 = f0 * (g1 - x);
    double r = __builtin_amdgcn_rcp(d);  // (computed whatever d is: nothing traps, and the test below throws a bad one away)
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    const double q1 = n1 * r, q0 = n0 * r;
    double res = __builtin_fma(__builtin_fma(-d, q1, n1), r, q1) + __builtin_fma(__builtin_fma(-d, q0, n0), r, q0);
    if (__builtin_expect(!(d >= 0x1p-300) || !(fabs(res) <= 0x1p1023), 0)) res = n1 / d + n0 / d;
    return res;
#endif
}

// One next-period table (state ist1 of period it+1), row 0 = (a0, 0, evf(a0)).
template <class PD> struct TabT {
    PD M, C, V;
    const double *TH, *D;
    int len, thlen;  // rows incl. the a0 row; thresholds
};
typedef TabT<const double *> Tab;    // in global memory
typedef TabT<const eg_ldsd *> TabL;  // M, C, V staged in LDS (k_fixup's sequential stretches)

// the M column of the next-period table of (draw, ist) is known to be in order in period `it` (k_sortcheck; Batch::sorted_valid)
static __device__ __forceinline__ int eg_tab_sorted(BatchRef b, int it, int draw, int ist)
{
    return b.sorted_valid != 0 && b.tsorted[(size_t)draw * MS_NST + ist] != b.sorted_valid + it;
}
static __device__ __forceinline__ Tab eg_tab(BatchRef b, int slot, int draw, int ist)
{
    size_t k = ((size_t)slot * b.g.ndraw + draw) * MS_NST + ist;
    Tab t;
    t.M = b.tM + k * b.g.Sp;
    t.C = b.tC + k * b.g.Sp;
    t.V = b.tV + k * b.g.Sp;
    t.TH = b.tTH + k * b.g.nthrhmax;
    t.D = b.tD + k * b.g.nthrhmax;
    t.len = b.tlen[k];
    t.thlen = b.tthlen[k];
    return t;
}

// Next-period value at nxt->cash (valuefunc, egdst_solver.c:755-772 with linter_extrap, egdst_lib.c:179-206).
// ibr: -1, or the bracket linter's search found for the same x over (M, len) when the column is known to be non-decreasing
// and len >= 4: valuefunc's search over the column shifted by one row then follows from it (k_grid_lds, eg_second_bracket).
// verr: set when the table has fewer than two rows beside the a0 row and the analytic branch does not apply: linter_extrap then
// reports "At least two points are required for interpolation!" (egdst_lib.c:183) and the solver returns at once
// (egdst_solver.c:567-568).  A table of one row is what the envelope of a guess stream that ended in the generator's resend fixed
// point leaves behind (C2 with a0 = -5: ~6 % of the parameter draws).
template <class TT> static __device__ __forceinline__ double eg_next_value(const ms_env *E, const TT &t, const ms_pv *nxt, int ibr = -1, int *verr = nullptr)
{
    const double evf1 = t.V[0], a0 = E->a0, x = nxt->cash;
    if (x < t.M[1] && evf1 > -INFINITY) return ms_utility(E, nxt, x - a0) + ms_discount(E, nxt) * evf1;
    const auto g = t.M + 1, f = t.V + 1;
    const int n = t.len - 1;
    if (n < 2) {
        if (verr) *verr = 1;
        return -1.0;
    }
    int i = (ibr >= 0) ? ((x < t.M[2]) ? 0 : ((x >= t.M[t.len - 2]) ? t.len - 3 : ibr - 1)) : eg_bracket(x, g, n, 0);
    double f0 = f[i], f1 = f[i + 1];
    if (!isfinite(f0)) return f0;
    if (!isfinite(f1)) return f1;
    double g0 = g[i], g1 = g[i + 1];
    if (x > a0 && (x > g[n - 1] || x < g[0])) {
        double tx = ms_tr(E, nxt, x - a0), t0 = ms_tr(E, nxt, g0 - a0), t1 = ms_tr(E, nxt, g1 - a0);
        return f1 * (tx - t0) / (t1 - t0) + f0 * (t1 - tx) / (t1 - t0);
    }
    return eg_lerp(x, g0, g1, f0, f1);
}

// One (next state, shock node) term of the expectation: the body at egdst_solver.c:548-570.
// Returns c1; on c1>0 fills the two weighted terms.  nxt->ist/shock must be set; fills nxt->cash, nxt->id.
// sorted: the table's M column is known to be non-decreasing (k_sortcheck) -- one bracket search serves both interpolations.
template <class TT>
static __device__ __forceinline__ double eg_term(const ms_env *E, const TT &t, const ms_pv *cur, ms_pv *nxt,
                                                 double pr1, int keep, double *t_rhs, double *t_evf, int sorted = 0, int *verr = nullptr)
{
    nxt->cash = ms_cashinhand(E, cur, nxt);
    const int n1 = t.len;
    // (a column known to be in order, four rows or more: the branch-free bisection -- the same bracket, see eg_bracket_sorted)
    int i = (sorted && n1 >= 4) ? eg_bracket_sorted(nxt->cash, t.M, n1) : eg_bracket(nxt->cash, t.M, n1, 0);
    double c1 = eg_lerp(nxt->cash, t.M[i], t.M[i + 1], t.C[i], t.C[i + 1]);
    if (nxt->cash > t.M[n1 - 1]) c1 = MS_MAX(c1, t.C[n1 - 1]);  // constant extrapolation, :554
    *t_rhs = 0;
    *t_evf = 0;
    if (c1 <= 0) return c1;
    if (!MS_OPTIM_MUNOD || (!MS_OPTIM_UNOD && keep == 1 && nxt->cash < t.M[1]))
        nxt->id = (int)t.D[eg_bracket(nxt->cash, t.TH, t.thlen, 1)];  // optimd, egdst_lib.c:129-132
    else
        nxt->id = 0;
    *t_rhs = pr1 * ms_utility_marginal(E, nxt, c1) * ms_cashinhand_marginal(E, cur, nxt);
    if (keep == 1) *t_evf = pr1 * eg_next_value(E, t, nxt, (sorted && n1 >= 4) ? i : -1, verr);
    return c1;
}

// Newton inversion of the budget (cashinhandinverse, egdst_lib.c:275-296); *err set on failure.
static __device__ __forceinline__ double eg_invert_budget(const ms_env *E, ms_pv cur, ms_pv nxt, double target, int *err)
{
    int cnt = 0;
    nxt.savings = target;
    while (fabs(ms_cashinhand(E, &cur, &nxt) - target) > EG_ZEROC / 10) {
        nxt.savings -= (ms_cashinhand(E, &cur, &nxt) - target) / ms_cashinhand_marginal(E, &cur, &nxt);
        if (++cnt >= 100) {
            *err = 24;
            return -1.0;
        }
    }
    return nxt.savings;
}
