/*
 * egdst_oracle.c -- CPU restatement of the reference's backward-induction DC-EGM solver and
 * forward simulator.  THIS IS TEST INFRASTRUCTURE (the parity oracle), not the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build or call it.
 *
 * What it restates (all paths relative to the reference tree):
 *   @egdstmodel/egdst_solver.c   solver/egmbellman/adraw/valuefunc/envelope2/envelop/
 *                                funcvalue/comp1/linter2/thresholds/brsolve/saveoutput
 *   @egdstmodel/egdst_simulator.c simulator/policy/simsoutput (discrete-state models)
 *   @egdstmodel/egdst_lib.c      bxsearch_common/linter/linter_extrap/optimd/rescale/
 *                                expectation/cdfinv/cdfni/cashinhandinverse
 * The model plugin (utility, budget, trpr, ...) is the generated modelspec.h (the counterpart
 * of compile.m's modelspec.c), included below; one oracle library is built per model.
 *
 * Pinning: the reference cannot be built in this image under the round's rules (it needs
 * MATLAB's mex.h/matrix.h and MATLAB-generated modelspec.c), and it ships no tests or golden
 * vectors.  The oracle is therefore pinned against the outputs of the reference
 * that the survey recorded in SURVEY.md §8(c) (row counts, thresholds, value/consumption rows,
 * column sums of the five shipped example models): tests/test_oracle_known_answers.py.
 *
 * Deliberate deviation: the reference reads the local `evf` while keep==0
 * although it is only assigned when keep==1 (egdst_solver.c:382,495,572,577,583).  Here evf is
 * 0.0 at the start of every A-guess, which is what the survey's working builds amount to.
 *
 * Math: built with -DEGDST_NATIVE_MATH the transcendental functions are glibc's, as in the reference (this is
 * the build that is pinned to the reference's recorded outputs); without it they are the bit-reproducible
 * ones of include/egdst_math.h, which the GPU path uses too, so GPU and oracle can be compared bit for bit.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared [-DEGDST_NATIVE_MATH] -I<dir of modelspec.h> -I include egdst_oracle.c -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MS_FN static inline
#define MS_TABLE static const
#include "modelspec.h"

#define TOL MS_TOLERANCE
#define ZEROC MS_ZEROCONSUMPTION
#define DPD MS_DOUBLEPOINT_DELTA
#define A0T 0.0 /* egdst_solver.c:49 (END2) */
#define MAXSEG 10000 /* egdst_solver.c:808,832 */

typedef struct {
    int t0, T, ngridm, ngridmax, nthrhmax, ny;
    double mmax, a0;
    const double *quadrature; /* [2*ny]: weights, then abscissae on [0,1] */
} orc_desc;

typedef struct {
    /* all caller-allocated; period-major: slot = it*nst + ist */
    double *M, *C, *V; /* [nt*nst*(ngridmax+1)], row 0 = (a0, 0, evf(a0)) */
    double *D, *TH;    /* [nt*nst*nthrhmax] */
    int *len;          /* [nt*nst] number of rows incl. the a0 row; 0 = not solved */
    int *thlen;        /* [nt*nst] */
    long long nevals;  /* executions of the body at egdst_solver.c:548-570 */
    char err[300];
    /* third output of the solver gateway (DEBUGOUT, egdst_solver.c:39,178-181,1866-1879): one row per kink the envelopes
       record, in the order they are recorded -- columns it, ist, choice whose secondary envelope it is (-1: primary),
       threshold, consumption left and right of it (-.9999 for the zero-consumption marker), |jump|.  Column-major
       [dbgcap x 7], dbgcap = nt*nst*nd*2*nt as the gateway sizes it; rows beyond the capacity are dropped.  NULL: off. */
    double *dbgout;
    int dbgcap, dbgn;
} orc_solution;

/* ------------------------------------------------------------------------------------------ */
/* numerics shared by solver and simulator (egdst_lib.c) */

/* Acklam's rational approximation of the inverse normal cdf (egdst_lib.c:435-519). */
static double inv_normal_cdf(double p)
{
    static const double ca[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                                1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
    static const double cb[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                                6.680131188771972e+01, -1.328068155288572e+01};
    static const double cc[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                                -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
    static const double cd[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00,
                                3.754408661907416e+00};
    double q, r;
    if (p < 0 || p > 1) return 0.0;
    if (p == 0) return -HUGE_VAL;
    if (p == 1) return HUGE_VAL;
    if (p < 0.02425) {
        q = sqrt(-2 * MS_LOG(p));
        return (((((cc[0] * q + cc[1]) * q + cc[2]) * q + cc[3]) * q + cc[4]) * q + cc[5]) /
               ((((cd[0] * q + cd[1]) * q + cd[2]) * q + cd[3]) * q + 1);
    }
    if (p > 0.97575) {
        q = sqrt(-2 * MS_LOG(1 - p));
        return -(((((cc[0] * q + cc[1]) * q + cc[2]) * q + cc[3]) * q + cc[4]) * q + cc[5]) /
               ((((cd[0] * q + cd[1]) * q + cd[2]) * q + cd[3]) * q + 1);
    }
    q = p - 0.5;
    r = q * q;
    return (((((ca[0] * r + ca[1]) * r + ca[2]) * r + ca[3]) * r + ca[4]) * r + ca[5]) * q /
           (((((cb[0] * r + cb[1]) * r + cb[2]) * r + cb[3]) * r + cb[4]) * r + 1);
}

/* shock for a standard-normal node / its degenerate expectation / inverse cdf draw
 * (egdst_lib.c:66-100, DISTRIB 1 = lognormal, 2 = normal) */
static double shock_from_node(const ms_env *E, const ms_pv *cur, const ms_pv *nxt, double z)
{
#if MS_DISTRIB == 1
    return MS_EXP(ms_mu(E, cur, nxt) + z * ms_sigma(E, cur, nxt));
#else
    return ms_mu(E, cur, nxt) + z * ms_sigma(E, cur, nxt);
#endif
}
static double shock_expectation(const ms_env *E, const ms_pv *cur, const ms_pv *nxt)
{
#if MS_DISTRIB == 1
    return MS_EXP(ms_mu(E, cur, nxt) + ms_sigma(E, cur, nxt) * ms_sigma(E, cur, nxt) / 2);
#else
    return ms_mu(E, cur, nxt);
#endif
}
static double shock_from_uniform(double u, double mu, double sigma)
{
#if MS_DISTRIB == 1
    return MS_EXP(sigma * inv_normal_cdf(u) + mu);
#else
    return sigma * inv_normal_cdf(u) + mu;
#endif
}

/* Bracket index (egdst_lib.c:136-166).  kind 0: interpolation bracket, kind 1: policy lookup. */
static int bracket(double x, const double *g, int n, int kind)
{
    int lo, hi, mid;
    if (x < g[1]) return 0;
    if (kind == 0 && x >= g[n - 2]) return n - 2;
    if (kind == 1 && x >= g[n - 1]) return n - 1;
    lo = 1;
    hi = n - 2;
    while (hi - lo > 1) {
        mid = (hi + lo) / 2;
        if (g[0] <= g[n - 1] && g[mid] > x)
            hi = mid;
        else
            lo = mid;
    }
    return lo;
}

/* Linear interpolation with linear extrapolation on both sides (egdst_lib.c:169-176). */
static double interp_lin(double x, int n, const double *g, const double *f)
{
    int i = bracket(x, g, n, 0);
    return f[i + 1] * (x - g[i]) / (g[i + 1] - g[i]) + f[i] * (g[i + 1] - x) / (g[i + 1] - g[i]);
}

/* Interpolation of the value function; outside the grid (and above a0) the weights are taken in
 * transformed coordinates tr(x-a0) (egdst_lib.c:179-206). */
static double interp_value(const ms_env *E, const ms_pv *prd, double x, int n, const double *g, const double *f)
{
    int i = bracket(x, g, n, 0);
    double a0 = E->a0;
    if (!isfinite(f[i])) return f[i];
    if (!isfinite(f[i + 1])) return f[i + 1];
    if (x > a0 && (x > g[n - 1] || x < g[0])) {
        double tx = ms_tr(E, prd, x - a0), t0 = ms_tr(E, prd, g[i] - a0), t1 = ms_tr(E, prd, g[i + 1] - a0);
        return f[i + 1] * (tx - t0) / (t1 - t0) + f[i] * (t1 - tx) / (t1 - t0);
    }
    return f[i + 1] * (x - g[i]) / (g[i + 1] - g[i]) + f[i] * (g[i + 1] - x) / (g[i + 1] - g[i]);
}

/* ------------------------------------------------------------------------------------------ */
/* solver state */

typedef struct {
    ms_env E;
    const orc_desc *d;
    orc_solution *sol;
    int nt, stride; /* stride = ngridmax+1 */
    double *qw, *qz; /* weights, standard-normal nodes */
    char *err;
    int dbg_id; /* dbgoutd of the reference: the choice whose secondary envelope runs, -1 in the primary one (:718,804) */
} ctx_t;

/* next-period table accessors: slot of (it+1, ist1) */
#define SLOT(c, it, ist) ((size_t)(it) * MS_NST + (ist))
#define TAB_M(c, it, ist) ((c)->sol->M + SLOT(c, it, ist) * (c)->stride)
#define TAB_C(c, it, ist) ((c)->sol->C + SLOT(c, it, ist) * (c)->stride)
#define TAB_V(c, it, ist) ((c)->sol->V + SLOT(c, it, ist) * (c)->stride)
#define TAB_D(c, it, ist) ((c)->sol->D + SLOT(c, it, ist) * (c)->d->nthrhmax)
#define TAB_TH(c, it, ist) ((c)->sol->TH + SLOT(c, it, ist) * (c)->d->nthrhmax)

static void fail(ctx_t *c, const char *msg)
{
    snprintf(c->err, 300, "Error:\n%s", msg);
}

/* Newton inversion of the budget: savings such that cashinhand == target (egdst_lib.c:275-296). */
static double invert_budget(ctx_t *c, ms_pv cur, ms_pv nxt, double target)
{
    int cnt = 0;
    nxt.savings = target;
    while (fabs(ms_cashinhand(&c->E, &cur, &nxt) - target) > ZEROC / 10) {
        nxt.savings -= (ms_cashinhand(&c->E, &cur, &nxt) - target) / ms_cashinhand_marginal(&c->E, &cur, &nxt);
        if (++cnt >= 100) {
            fail(c, "Did not manage to invert the intertemporal budget (cashinhand) after performing many-many iterations!");
            return -1.0;
        }
    }
    return nxt.savings;
}

/* Next-period value at nxt->cash in state nxt->ist (egdst_solver.c:755-772). */
static double next_value(ctx_t *c, const ms_pv *nxt)
{
    int slot_it = nxt->it, ist1 = nxt->ist;
    const double *gm = TAB_M(c, slot_it, ist1), *gv = TAB_V(c, slot_it, ist1);
    int n1 = c->sol->len[SLOT(c, slot_it, ist1)] - 1; /* points without the a0 row */
    double evf1 = gv[0];
    if (nxt->cash < gm[1] && evf1 > -INFINITY)
        return ms_utility(&c->E, nxt, nxt->cash - c->E.a0) + ms_discount(&c->E, nxt) * evf1;
    if (n1 < 2) { /* linter_extrap, egdst_lib.c:183; the caller returns on it (egdst_solver.c:567-568) */
        fail(c, "Error: At least two points are required for interpolation!");
        return -1.0;
    }
    return interp_value(&c->E, nxt, nxt->cash, n1, gm + 1, gv + 1);
}

/* One A-guess: the double loop over next states and shock nodes (egdst_solver.c:494-574).
 * status: 0 normal, 1 c1<=0, 2 evf==-inf, -1 hard error.  nxt keeps ist/shock/cash of the break. */
static int expectation_at(ctx_t *c, const ms_pv *cur, ms_pv *nxt, int keep, double *rhs_out, double *evf_out)
{
    const ms_env *E = &c->E;
    int ny = c->d->ny, niy, iy, terr = 0;
    /* evf and c1 are defined at the start of every A-guess; c1 starts positive so that
     * "no evaluation yet" never reads as "negative consumption" */
    double rhs = 0, evf = 0, c1 = 1.0, checksum = 0, pr1, pr1pre = 0;
    for (nxt->ist = 0; nxt->ist < MS_NST; nxt->ist++) {
        if (ms_feasible(E, nxt) != 1) continue;
        if (MS_OPTIM_TRPRNOSH) {
            pr1pre = ms_trpr(E, cur, nxt, &terr);
            if (pr1pre == 0.0) {
                if (terr) break;
                continue;
            }
        }
        niy = (ms_sigma(E, cur, nxt) <= 0 || ny == 1) ? 1 : ny;
        for (iy = 0; iy < niy; iy++) {
            if (niy == 1) {
                nxt->shock = shock_expectation(E, cur, nxt);
                pr1 = MS_OPTIM_TRPRNOSH ? pr1pre : ms_trpr(E, cur, nxt, &terr);
            } else {
                nxt->shock = shock_from_node(E, cur, nxt, c->qz[iy]);
                pr1 = MS_OPTIM_TRPRNOSH ? pr1pre : ms_trpr(E, cur, nxt, &terr);
                pr1 *= c->qw[iy];
            }
            if (pr1 == 0.0) continue;
            checksum += pr1;
            nxt->cash = ms_cashinhand(E, cur, nxt);
            {
                size_t sl = SLOT(c, nxt->it, nxt->ist);
                int n1 = c->sol->len[sl]; /* rows incl. a0 */
                const double *gm = TAB_M(c, nxt->it, nxt->ist), *gc = TAB_C(c, nxt->it, nxt->ist);
                c->sol->nevals++;
                if (n1 < 2) {
                    fail(c, "Error: At least two points are required for interpolation!");
                    return -1;
                }
                c1 = interp_lin(nxt->cash, n1, gm, gc);
                if (nxt->cash > gm[n1 - 1]) c1 = MS_MAX(c1, gc[n1 - 1]); /* constant extrapolation :554 */
                if (c1 <= 0) break;
                if (!MS_OPTIM_MUNOD || (!MS_OPTIM_UNOD && keep == 1 && nxt->cash < gm[1])) {
                    int nth = c->sol->thlen[sl];
                    nxt->id = (int)TAB_D(c, nxt->it, nxt->ist)[bracket(nxt->cash, TAB_TH(c, nxt->it, nxt->ist), nth, 1)];
                } else
                    nxt->id = 0;
                rhs += pr1 * ms_utility_marginal(E, nxt, c1) * ms_cashinhand_marginal(E, cur, nxt);
                if (keep == 1) {
                    evf += pr1 * next_value(c, nxt);
                    if (c->err[0]) return -1; /* egdst_solver.c:568 */
                    if (evf == -INFINITY) break;
                }
            }
        }
        if (c1 <= 0 || evf == -INFINITY) break;
    }
    if (terr) {
        fail(c, "Error in trpr: unknown combination of current state and decision (the set of cases is not complete)!");
        return -1;
    }
    *rhs_out = rhs;
    *evf_out = evf;
    if (c1 > 0 && evf > -INFINITY && fabs(checksum - 1) > TOL) {
        fail(c, "Transition probabilities don't sum up! Check model specification!");
        return -1;
    }
    if (c1 <= 0) return 1;
    if (evf == -INFINITY) return 2;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* the stream of end-of-period asset guesses (egdst_solver.c:350-367,955-1159) */

typedef struct {
    int ntogenerate, ngenerated, ncalls, keep;
    double baseM, baseA, lim1, lim2, lim2p, lim3, lim3p, k3, last, M, M1;
    const ms_pv *cur;
} agen_t;

static double next_guess(ctx_t *c, agen_t *g)
{
    const ms_env *E = &c->E;
    double a0 = E->a0, mmax = E->mmax, aa, bb, step;
    g->ncalls += 1;
    if (g->ncalls >= c->d->ngridmax) { /* runaway guard :963-978 (a warning, not an error) */
        g->last = -INFINITY;
        return g->last;
    }
    if (g->ngenerated == 0) { /* stage 0: find a base point with M(A)<=mmax */
        g->keep = 0;
        if (g->M == INFINITY)
            g->last = mmax;
        else if (g->M <= mmax) {
            g->baseA = g->last;
            g->baseM = g->M;
            g->ngenerated = 1;
            g->keep = 1;
            g->last = a0;
            g->k3 = 0;
            g->M1 = a0 - 1;
        } else {
            if (g->last - a0 < TOL) {
                fail(c, "Could not complete initial stage in adraw()..\nSeems like M(a0)>mmax! Increase mmax!");
                return -1.0;
            }
            g->last = (g->last + a0) / 2;
        }
        return g->last;
    }
    aa = (g->M - g->baseM) / (g->last - g->baseA);
    bb = g->baseM - aa * g->baseA; /* (the `upper` forecast of :1035-1046 is dead with rescaling switched off) */
    g->M1 = g->M;
    if (g->ngenerated == 1 && g->k3 == 0) { /* limits after the first kept call :1051-1078 */
        g->ntogenerate = c->d->ngridm;
        g->lim2p = MS_MIN(mmax, (mmax - bb) / aa);
        g->lim3p = -bb / aa;
        if (a0 < 0 && a0 < g->lim3p)
            g->k3 = MS_MAX(floor(g->ntogenerate * (g->lim3p - a0) / (g->lim2p - a0)), 2.0);
        else {
            g->lim3p = a0;
            g->k3 = 1.0;
        }
        g->lim1 = ms_tr(E, g->cur, g->lim3p - a0);
        g->lim2 = ms_tr(E, g->cur, g->lim2p - g->lim3p);
        g->lim3 = ms_tr(E, g->cur, 0);
    }
    if (g->M <= a0 - 1 + TOL) { /* c1<=0 signal: resend the prepared point :1080-1099 */
        g->keep = 1;
        aa = (a0 - g->baseM) / (a0 - g->baseA);
        bb = g->baseM - aa * g->baseA;
        g->lim2p = MS_MIN(mmax, (mmax - bb) / aa);
        g->lim3p = g->last - ZEROC;
        g->k3 = 1.0;
        g->lim1 = ms_tr(E, g->cur, g->lim3p - a0);
        g->lim2 = ms_tr(E, g->cur, g->lim2p - g->lim3p);
        g->lim3 = ms_tr(E, g->cur, 0);
    } else if (g->M < mmax && g->ngenerated < g->ntogenerate) {
        g->keep = 1;
        if ((int)g->ngenerated < (int)g->k3 - 1)
            step = -ms_trinv(E, g->cur, g->lim3 + (g->k3 - 1 - g->ngenerated) * (g->lim1 - g->lim3) / (g->k3 - 1)) +
                   g->lim3p - g->last;
        else
            step = ms_trinv(E, g->cur, g->lim3 + (g->ngenerated - g->k3 + 1) * (g->lim2 - g->lim3) / (g->ntogenerate - g->k3)) +
                   g->lim3p - g->last;
        if (step < 0) step = MS_MAX(step, 1e-5);
        g->last += step;
        g->ngenerated += 1;
    } else
        g->last = -INFINITY;
    return g->last;
}

/* ------------------------------------------------------------------------------------------ */
/* upper envelopes (egdst_solver.c:776-913,1165-1968) */

typedef struct {
    double m, c, v;
    int f; /* function (choice or segment) index */
} pt_t;

typedef struct {
    ctx_t *c;
    int it, ist, nf;
    pt_t *p;      /* sorted points */
    int **at;     /* at[f][k]: sorted position of the k-th point of function f */
    int *dims, *cur, *mark;
    const double *evfa0;
    double *og, *ov, *oc, *oth, *oix;
    int oi, oj;
} env_t;

static int cmp_pts(const void *a, const void *b) /* egdst_solver.c:1570-1582 */
{
    const pt_t *x = a, *y = b;
    if (x->m > y->m) return 1;
    if (x->m < y->m) return -1;
    if (x->v > y->v) return -1;
    if (x->v < y->v) return 1;
    if (x->f > y->f) return 1;
    if (x->f < y->f) return -1;
    return 0;
}

/* value of `which` (0: v, 1: c) of function f on the segment starting at its k-th point; no
 * extrapolation (egdst_solver.c:1585-1593) */
static double seg_val(const env_t *e, int f, int k, double x, int which)
{
    const pt_t *a = &e->p[e->at[f][k]], *b = &e->p[e->at[f][k + 1]];
    double fa = which ? a->c : a->v, fb = which ? b->c : b->v;
    if (x == a->m) return fa;
    if (x < a->m) return -INFINITY;
    if (x > b->m) return -INFINITY;
    return fb * (x - a->m) / (b->m - a->m) + fa * (b->m - x) / (b->m - a->m);
}

static double analytic_val(const env_t *e, int f, double x)
{
    ms_pv cv;
    memset(&cv, 0, sizeof cv); /* (byval = 0: the solver reads states and decisions from the tables) */
    cv.it = e->it;
    cv.ist = e->ist;
    cv.id = f;
    return ms_utility(&e->c->E, &cv, x - e->c->E.a0) + ms_discount(&e->c->E, &cv) * e->evfa0[f];
}

static double fn_val(const env_t *e, int f, double x) /* egdst_solver.c:1553-1567 */
{
    if (e->cur[f] >= 0) return seg_val(e, f, e->cur[f], x, 0);
    if (e->evfa0[f] == -INFINITY) return -INFINITY;
    return analytic_val(e, f, x);
}

static double policy_val(const env_t *e, int f, double x) /* consumption of f at x (:1406-1408,1859-1864) */
{
    if (e->cur[f] >= 0) return seg_val(e, f, e->cur[f], x, 1);
    if (e->evfa0[f] == -INFINITY) return ZEROC;
    return x - e->c->E.a0;
}

static double sgn(double x) { return x > 0 ? 1.0 : -1.0; }

/* bisection for the crossing of an analytic value function with a linear segment (:1918-1968) */
static void bisect(env_t *e, double *b0, double *b1, int fl, int kl, int fa)
{
    for (;;) {
        double f0 = analytic_val(e, fa, *b0), f1 = analytic_val(e, fa, *b1), mid, fm;
        if (sgn(f0 - seg_val(e, fl, kl, *b0, 0)) == sgn(f1 - seg_val(e, fl, kl, *b1, 0))) {
            fail(e->c, "Fatal error in braketing module! Solution is outside of brackets.");
            return;
        }
        if (*b0 > *b1) {
            fail(e->c, "Fatal error in braketing module! Bracket limits reversed.");
            return;
        }
        if (fabs(*b0 - *b1) < 2 * DPD || fabs(f0 - f1) < DPD) {
            *b0 = (*b0 + *b1) / 2;
            return;
        }
        mid = (*b0 + *b1) / 2;
        fm = analytic_val(e, fa, mid) - seg_val(e, fl, kl, mid, 0);
        if (sgn(f0 - seg_val(e, fl, kl, *b0, 0)) == sgn(fm))
            *b0 = mid;
        else if (sgn(f1 - seg_val(e, fl, kl, *b1, 0)) == sgn(fm))
            *b1 = mid;
        else
            return;
    }
}

/* crossing of the previous max `pri` with `nwi`; records kink points and the threshold (:1596-1915) */
static void crossing(env_t *e, int pri, int nwi, int mode)
{
    ctx_t *c = e->c;
    double a0 = c->E.a0, x = 0, top = 0, t, pol[2];
    int k, best, cp, cn;
    e->mark[pri] = 1;
    e->mark[nwi] = 1;
    cp = e->cur[pri];
    cn = e->cur[nwi];
    if (cp == -1 && cn != -1) {
        if (e->evfa0[pri] == -INFINITY)
            x = e->p[e->at[pri][0]].m;
        else {
            double br0 = e->p[e->at[nwi][cn]].m, br1 = MS_MIN(e->p[e->at[pri][0]].m, e->p[e->at[nwi][cn + 1]].m);
            bisect(e, &br0, &br1, nwi, cn, pri);
            if (c->err[0]) return;
            x = br0;
        }
        top = seg_val(e, nwi, cn, x, 0);
    } else if (cp != -1 && cn == -1) {
        if (e->evfa0[nwi] == -INFINITY)
            x = e->p[e->at[nwi][0]].m;
        else {
            double br0 = e->p[e->at[pri][cp]].m, br1 = MS_MIN(e->p[e->at[nwi][0]].m, e->p[e->at[pri][cp + 1]].m);
            bisect(e, &br0, &br1, pri, cp, nwi);
            if (c->err[0]) return;
            x = br0;
        }
        top = seg_val(e, pri, cp, x, 0);
    } else if (cp == -1 && cn == -1) {
        fail(c, "Fatal error in threshold module. Two analytical value functions seem to intersect. Utility is not additively separable in consumption and discrete choices.");
        return;
    } else {
        const pt_t *p0 = &e->p[e->at[pri][cp]], *p1 = &e->p[e->at[pri][cp + 1]];
        const pt_t *n0 = &e->p[e->at[nwi][cn]], *n1 = &e->p[e->at[nwi][cn + 1]];
        double icn = (n0->v * n1->m - n1->v * n0->m) / (n1->m - n0->m); /* intercept of nwi's segment */
        double icp = (p0->v * p1->m - p1->v * p0->m) / (p1->m - p0->m);
        if (p1->m == p0->m) { /* pri vertical */
            x = p0->m;
            top = (x * (n1->v - n0->v) / (n1->m - n0->m)) + icn;
        } else if (n1->m == n0->m) { /* nwi vertical */
            x = n0->m;
            top = (x * (p1->v - p0->v) / (p1->m - p0->m)) + icp;
        } else if (((n1->v - n0->v) / (n1->m - n0->m)) == ((p1->v - p0->v) / (p1->m - p0->m))) { /* parallel */
            x = (p0->m + p1->m + n0->m + n1->m) / 4;
            top = (x * (n1->v - n0->v) / (n1->m - n0->m)) + icn;
        } else {
            x = (icp - icn) / (((n1->v - n0->v) / (n1->m - n0->m)) - ((p1->v - p0->v) / (p1->m - p0->m)));
            top = (x * (n1->v - n0->v) / (n1->m - n0->m)) + icn;
        }
    }
    /* is a third function higher at the crossing? */
    best = -1;
    for (k = 0; k < e->nf; k++) {
        if (e->mark[k] == 1) continue;
        /* u(.)+beta*(-inf) is -inf: skip the call so that segment indices never reach the model's id tables */
        t = (e->cur[k] >= 0) ? seg_val(e, k, e->cur[k], x, 0)
                             : (e->evfa0[k] == -INFINITY ? -INFINITY : analytic_val(e, k, x));
        if (top < t) {
            top = t;
            best = k;
            if (mode == 0) break;
        }
    }
    if (best != -1) {
        crossing(e, pri, best, mode);
        if (mode != 0 && !c->err[0]) crossing(e, best, nwi, mode);
        return;
    }
    e->og[e->oi] = x;
    e->ov[e->oi] = top;
    pol[0] = policy_val(e, pri, x);
    pol[1] = policy_val(e, nwi, x);
    e->oc[e->oi] = (pol[0] + pol[1]) / 2;
    if (c->sol->dbgout && c->sol->dbgn < c->sol->dbgcap) { /* :1866-1879 */
        double *o = c->sol->dbgout + c->sol->dbgn;
        const int n = c->sol->dbgcap;
        o[0] = e->it;
        o[n] = e->ist;
        o[2 * n] = c->dbg_id;
        o[3 * n] = x;
        o[4 * n] = (pol[0] == ZEROC) ? -.9999 : pol[0];
        o[5 * n] = (pol[1] == ZEROC) ? -.9999 : pol[1];
        o[6 * n] = fabs(pol[1] - pol[0]);
        c->sol->dbgn++;
    }
    e->oth[e->oj] = x;
    e->oix[e->oj] = nwi;
    e->oi += 1;
    e->oj += 1;
    if (e->oi >= c->d->ngridmax) {
        fail(c, "Not enough space for endogenous grid. Increase max number of grid points for M!");
        return;
    }
    if (e->oj >= c->d->nthrhmax) {
        fail(c, "Not enough space for thresholds. Increase max number of threshold points!");
        return;
    }
    if (e->evfa0[nwi] == -INFINITY && e->cur[nwi] == -1) { /* entering function starts from -inf :1892-1900 */
        e->oc[e->oi - 1] = pol[0];
        e->og[e->oi - 1] = e->og[e->oi - 1] - TOL;
    } else if (DPD > 0) { /* double point :1902-1913 */
        e->oc[e->oi - 1] = pol[0];
        e->og[e->oi] = x + DPD;
        e->ov[e->oi] = top;
        e->oc[e->oi] = pol[1];
        e->oi += 1;
        if (e->oi >= c->d->ngridmax) {
            fail(c, "Not enough space for endogenous grid. Increase max number of grid points for M!");
            return;
        }
    }
}

static void reset_marks(env_t *e)
{
    int l;
    for (l = 0; l < e->nf; l++) e->mark[l] = (e->dims[l] > 0 ? 0 : 1);
}

static void push_point(env_t *e, double g, double v, double cc)
{
    e->og[e->oi] = g;
    e->ov[e->oi] = v;
    e->oc[e->oi] = cc;
    e->oi++;
}

#define GRID_FULL "Not enough space for endogenous grid. Increase max number of grid points for M!"

/* Upper envelope of nf tabulated functions (egdst_solver.c:1165-1550).  pts is sorted in place. */
static void upper_envelope(ctx_t *c, int it, int ist, int nf, int npts, pt_t *pts, const double *evfa0,
                           double *og, double *ov, double *oc, double *oth, double *oix, int *outn, int *outm)
{
    env_t e;
    int i, j, f, ci = 0, cj, above;
    double bound, x, fv, t;
    int *store = calloc((size_t)nf * 3 + (size_t)npts + 1, sizeof(int));
    int **at = calloc((size_t)nf, sizeof(int *));
    e.c = c;
    e.it = it;
    e.ist = ist;
    e.nf = nf;
    e.p = pts;
    e.dims = store;
    e.cur = store + nf;
    e.mark = store + 2 * nf;
    e.at = at;
    e.evfa0 = evfa0;
    e.og = og;
    e.ov = ov;
    e.oc = oc;
    e.oth = oth;
    e.oix = oix;
    e.oi = e.oj = 0;
    qsort(pts, (size_t)npts, sizeof(pt_t), cmp_pts);
    for (i = 0; i < npts; i++) e.dims[pts[i].f]++;
    {
        int *pos = store + 3 * nf, off = 0;
        for (f = 0; f < nf; f++) {
            at[f] = pos + off;
            off += e.dims[f];
            e.dims[f] = 0;
        }
        for (i = 0; i < npts; i++) at[pts[i].f][e.dims[pts[i].f]++] = i;
    }
    for (f = 0; f < nf; f++) e.cur[f] = -1;
    bound = INFINITY; /* min over functions of their last grid point :1266-1271 */
    for (f = 0; f < nf; f++)
        if (e.dims[f] > 0 && pts[at[f][e.dims[f] - 1]].m < bound) bound = pts[at[f][e.dims[f] - 1]].m;

    for (i = 0; i < npts && pts[i].m <= bound; i++) {
        f = pts[i].f;
        x = pts[i].m;
        if (e.oi > 0 && og[e.oi - 1] == x) { /* duplicate grid point :1290-1298 */
            e.cur[f]++;
            continue;
        }
        fv = pts[at[f][e.cur[f] + 1]].v;
        if (e.oj == 0) { /* first point of the common grid :1303-1347 */
            t = fv;
            ci = f;
            for (j = 0; j < nf; j++) {
                if (e.dims[j] <= 0 || j == f) continue;
                fv = fn_val(&e, j, x);
                if (fv > t) t = fv, ci = j;
                if (fv == t && ci > j) ci = j;
            }
            oth[e.oj] = c->E.a0;
            oix[e.oj] = ci;
            e.oj++;
            if (e.oj >= c->d->nthrhmax) {
                fail(c, "Not enough space for thresholds. Increase max number of threshold points!");
                goto done;
            }
            if (ci == f) {
                push_point(&e, x, t, pts[at[f][e.cur[f] + 1]].c);
                if (e.oi >= c->d->ngridmax) {
                    fail(c, GRID_FULL);
                    goto done;
                }
            }
        } else if ((int)oix[e.oj - 1] == f) { /* point of the function that is currently the max :1348-1416 */
            above = 0;
            for (j = 0; j < nf; j++) {
                if (e.dims[j] <= 0 || j == f) continue;
                t = fn_val(&e, j, x);
                if (fv < t) {
                    above = 1;
                    if (x != bound) break;
                    fv = t;
                    ci = j;
                }
            }
            if (!above) {
                push_point(&e, x, fv, pts[at[f][e.cur[f] + 1]].c);
                if (e.oi == c->d->ngridmax) {
                    fail(c, GRID_FULL);
                    goto done;
                }
            } else if (x != bound) {
                reset_marks(&e);
                crossing(&e, f, j, 0);
                if (c->err[0]) goto done;
            } else {
                int jj = j; /* the reference tests evfa0[j] with the loop variable left at nf (:1407) */
                reset_marks(&e);
                crossing(&e, f, ci, 1);
                if (c->err[0]) goto done;
                og[e.oi] = x;
                ov[e.oi] = fn_val(&e, ci, x);
                if (e.cur[ci] >= 0)
                    oc[e.oi] = seg_val(&e, ci, e.cur[ci], x, 1);
                else if (jj < nf && evfa0[jj] == -INFINITY)
                    oc[e.oi] = ZEROC;
                else
                    oc[e.oi] = x - c->E.a0;
                e.oi++;
                if (e.oi >= c->d->ngridmax) {
                    fail(c, GRID_FULL);
                    goto done;
                }
            }
        } else { /* point of another function :1417-1516 */
            ci = (int)oix[e.oj - 1];
            t = fn_val(&e, ci, x);
            if (t < fv) {
                cj = -1;
                for (j = 0; j < nf; j++) {
                    if (e.dims[j] <= 0 || j == f || j == ci) continue;
                    t = fn_val(&e, j, x);
                    if ((fv < t) || (fv == t && j < cj)) fv = t, cj = j;
                }
                reset_marks(&e);
                if (cj == -1) {
                    crossing(&e, ci, f, 1);
                    if (c->err[0]) goto done;
                    push_point(&e, x, fv, pts[at[f][e.cur[f] + 1]].c);
                    if (e.oi >= c->d->ngridmax) {
                        fail(c, GRID_FULL);
                        goto done;
                    }
                } else {
                    crossing(&e, ci, cj, 1);
                    if (c->err[0]) goto done;
                    if (x == bound) {
                        og[e.oi] = x;
                        ov[e.oi] = fn_val(&e, cj, x);
                        oc[e.oi] = policy_val(&e, cj, x);
                        e.oi++;
                    }
                }
            } else if (x == bound) {
                og[e.oi] = x;
                ov[e.oi] = fn_val(&e, ci, x);
                oc[e.oi] = policy_val(&e, ci, x);
                e.oi++;
            }
        }
        e.cur[f] = MS_MIN(e.cur[f] + 1, e.dims[f] - 2);
    }
done:
    *outn = e.oi;
    *outm = e.oj;
    free(store);
    free(at);
}

/* Secondary envelope within one choice: split at fold-backs, take the envelope of the pieces
 * (egdst_solver.c:776-913).  pts has room for the added constant-extrapolation points.
 * Returns the number of points dropped (negative when kink points were added); errors via c->err. */
static int secondary_envelope(ctx_t *c, const ms_pv *cur, pt_t *pts, int n, double evfa0_id)
{
    int ngridmax = c->d->ngridmax, i, seg = cur->id, nadd = 0, total = 1, nout = n, mout;
    double *evf = calloc(MAXSEG, sizeof(double));
    pts[0].f = seg;
    evf[seg] = evfa0_id;
    for (i = 1; i < n; i++) {
        if (pts[i - 1].m > pts[i].m || pts[i - 1].v > pts[i].v) {
            if (total >= ngridmax) {
                fail(c, "Not enough space for endogenous grid in envelop2()");
                free(evf);
                return -1;
            }
            pts[n + nadd].m = 1.5 * c->E.mmax;
            pts[n + nadd].c = pts[i - 1].c;
            pts[n + nadd].v = pts[i - 1].v;
            pts[n + nadd].f = seg;
            total++;
            seg++;
            nadd++;
            if (seg >= MAXSEG) {
                fail(c, "10000 is not enough in envelop2()");
                free(evf);
                return -1;
            }
            evf[seg] = -INFINITY;
        }
        pts[i].f = seg;
        total++;
    }
    if (nadd > 0) {
        double *o = calloc((size_t)5 * ngridmax, sizeof(double));
        c->dbg_id = cur->id;
        upper_envelope(c, cur->it, cur->ist, seg + 1, total, pts, evf, o, o + ngridmax, o + 2 * ngridmax,
                       o + 3 * ngridmax, o + 4 * ngridmax, &nout, &mout);
        if (!c->err[0] && nout >= ngridmax) fail(c, "Not enough space for endogenous grid in envelop2()");
        if (c->err[0]) {
            free(o);
            free(evf);
            return -1;
        }
        for (i = 0; i < nout; i++) {
            pts[i].m = o[i];
            pts[i].v = o[ngridmax + i];
            pts[i].c = o[2 * ngridmax + i];
        }
        free(o);
    }
    for (i = 0; i < nout; i++) pts[i].f = cur->id;
    free(evf);
    return n - nout;
}

/* ------------------------------------------------------------------------------------------ */
/* one (it, ist): EGM step for every choice, envelopes, output (egdst_solver.c:370-752,917-952) */

static void solve_state(ctx_t *c, int it, int ist)
{
    const ms_env *E = &c->E;
    const orc_desc *d = c->d;
    int nlast = c->nt - 1, ngridmax = d->ngridmax, id, i, nall = 0, nid, any = 0, skipped, st;
    double evfa0[MS_ND], rhs, evf;
    pt_t *pts = calloc((size_t)MS_ND * ngridmax + ngridmax, sizeof(pt_t));
    ms_pv cur, nxt;
    size_t sl = SLOT(c, it, ist);
    double *oM = TAB_M(c, it, ist), *oC = TAB_C(c, it, ist), *oV = TAB_V(c, it, ist);
    int outn = 0, outm = 0;
    memset(&cur, 0, sizeof cur);
    memset(&nxt, 0, sizeof nxt);
    cur.it = it;
    cur.ist = ist;
    nxt.it = it + 1;
    for (id = 0; id < MS_ND; id++) evfa0[id] = 0.0;
    for (id = 0; id < MS_ND; id++) {
        cur.id = id;
        if (ms_inchoiceset(E, &cur) != 1) continue;
        any = 1;
        nid = 0;
        evfa0[id] = 0.0;
        if (it == nlast) { /* terminal period :433-476 */
            double m1 = ms_tr(E, &cur, ZEROC - A0T), m2 = ms_tr(E, &cur, E->mmax - A0T);
            evfa0[id] = -INFINITY;
            for (i = 0; i < d->ngridm; i++) {
                pt_t *p = &pts[nall];
                p->m = ms_trinv(E, &cur, m1 + i * (m2 - m1) / (d->ngridm - 1)) + A0T;
                p->c = p->m - A0T;
                p->v = ms_utility(E, &cur, p->c);
                p->f = id;
                nid++;
                nall++;
            }
            continue;
        }
        {
            agen_t g;
            memset(&g, 0, sizeof g);
            g.ntogenerate = d->ngridm;
            g.M = INFINITY;
            g.cur = &cur;
            while (nxt.savings = next_guess(c, &g), nxt.savings != -INFINITY) {
                if (c->err[0]) goto out;
                st = expectation_at(c, &cur, &nxt, g.keep, &rhs, &evf);
                if (st < 0) goto out;
                if (st > 0) { /* emergency: negative consumption or -inf value :583-627 */
                    if (g.ngenerated == 0) {
                        fail(c, "Failed to find any value of savings to result in positive consumption next period! Increase mmax.");
                        goto out;
                    }
                    evfa0[id] = -INFINITY;
                    g.M = nxt.cash;
                    if (st == 1) {
                        double evf1 = TAB_V(c, it + 1, nxt.ist)[0];
                        g.M = E->a0 - 1;
                        if (evf1 > -INFINITY)
                            g.last = invert_budget(c, cur, nxt, E->a0) + ZEROC;
                        else
                            g.last = invert_budget(c, cur, nxt, TAB_M(c, it + 1, nxt.ist)[1]) + ZEROC;
                        if (c->err[0]) goto out;
                    }
                    continue;
                }
                rhs *= ms_discount(E, &cur);
                g.M = nxt.savings + ms_utility_marginal_inverse(E, &cur, rhs);
                if (g.keep == 1 && isfinite(g.M)) {
                    pt_t *p = &pts[nall];
                    if (fabs(g.last - E->a0) < TOL && evfa0[id] > -INFINITY) evfa0[id] = evf;
                    p->m = g.M;
                    p->c = g.M - nxt.savings;
                    p->v = ms_utility(E, &cur, p->c) + ms_discount(E, &cur) * evf;
                    p->f = id;
                    nid++;
                    nall++;
                    if (nid >= ngridmax) {
                        fail(c, GRID_FULL);
                        goto out;
                    }
                }
            }
            if (c->err[0]) goto out;
            if (nid > 0) {
                skipped = secondary_envelope(c, &cur, &pts[nall - nid], nid, evfa0[id]);
                if (c->err[0]) goto out; /* skipped may be negative: kinks add double points */
                nall -= skipped;
            }
        }
    }
    if (!any) {
        fail(c, "Empty choiceset encountered! Check model specifications!");
        goto out;
    }
    if (nall == 0) {
        fail(c, "All of the choices lead to -inf value functions for all values of money-at-hand!");
        goto out;
    }
    c->dbg_id = -1;
    upper_envelope(c, it, ist, MS_ND, nall, pts, evfa0, oM + 1, oV + 1, oC + 1, TAB_TH(c, it, ist), TAB_D(c, it, ist),
                   &outn, &outm);
    if (c->err[0]) goto out;
    if (outn == 0) {
        fail(c, "Failed to compute upper envelope, most likely individual grids don't overlap!");
        goto out;
    }
    /* row 0 and bookkeeping (saveoutput :917-952, evf(a0) :730) */
    oM[0] = E->a0;
    oC[0] = 0;
    oV[0] = evfa0[(int)TAB_D(c, it, ist)[0]];
    c->sol->len[sl] = outn + 1;
    c->sol->thlen[sl] = outm;
out:
    free(pts);
}

int egdst_oracle_solve(const orc_desc *d, const double *par, orc_solution *sol)
{
    ctx_t c;
    int it, ist, i, nt = d->T - d->t0 + 1;
    memset(&c, 0, sizeof c);
    c.d = d;
    c.sol = sol;
    c.nt = nt;
    c.stride = d->ngridmax + 1;
    c.err = sol->err;
    sol->err[0] = 0;
    sol->nevals = 0;
    sol->dbgn = 0;
    c.E.t0 = d->t0;
    c.E.T = d->T;
    c.E.ngridm = d->ngridm;
    c.E.ngridmax = d->ngridmax;
    c.E.nthrhmax = d->nthrhmax;
    c.E.ny = d->ny;
    c.E.mmax = d->mmax;
    c.E.a0 = d->a0;
    c.E.par = par;
    c.qw = malloc(sizeof(double) * 2 * (size_t)d->ny);
    c.qz = c.qw + d->ny;
    for (i = 0; i < d->ny; i++) {
        c.qw[i] = d->quadrature[i];
        c.qz[i] = inv_normal_cdf(d->quadrature[d->ny + i]); /* egdst_solver.c:164 */
    }
    for (i = 0; i < nt * MS_NST; i++) sol->len[i] = sol->thlen[i] = 0;
    for (it = nt - 1; it >= 0 && !sol->err[0]; it--)
        for (ist = 0; ist < MS_NST && !sol->err[0]; ist++) {
            ms_pv cur;
            memset(&cur, 0, sizeof cur);
            cur.it = it;
            cur.ist = ist;
            if (ms_feasible(&c.E, &cur) == 1) solve_state(&c, it, ist);
        }
    free(c.qw);
    return sol->err[0] ? 1 : 0;
}

/* ------------------------------------------------------------------------------------------ */
/* simulator (egdst_simulator.c:47-383), models with continuous states included (:313-372) */

typedef struct {
    ms_pv pv;
    double c, vf, mu, sigma, eqs[MS_NEQ + 1];
} simrow_t;

static void sim_policy(const ms_env *E, const orc_desc *d, const orc_solution *sol, simrow_t *r)
{
    size_t sl = (size_t)r->pv.it * MS_NST + r->pv.ist;
    int stride = d->ngridmax + 1, nm = sol->len[sl], nth = sol->thlen[sl], ith = 0;
    const double *gm = sol->M + sl * stride, *gc = sol->C + sl * stride, *gv = sol->V + sl * stride;
    const double *th = sol->TH + sl * d->nthrhmax, *dd = sol->D + sl * d->nthrhmax;
    double ma0 = gm[1], evf = gv[0];
    r->c = interp_lin(r->pv.cash, nm, gm, gc);
    r->pv.savings = r->pv.cash - r->c;
    while (ith < nth && r->pv.cash >= th[ith]) ith++;
    r->pv.id = (int)dd[ith - 1];
    if (r->pv.cash < ma0 && evf > -INFINITY)
        r->vf = ms_utility(E, &r->pv, r->c) + ms_discount(E, &r->pv) * evf;
    else
        r->vf = interp_lin(r->pv.cash, nm, gm, gv);
}

/* sims: [nsimout x nt x nsim] column-major, nsimout = 11+nnst+nnd+neq; init: [nsim x 2] column-major
 * (ist base-1, m0); rndtype 1 = every agent reuses the head of randstream.  Returns 0, or a
 * negative code for the hard errors of the gateway (egdst_simulator.c:54-75). */
int egdst_oracle_sim(const orc_desc *d, const double *par, const orc_solution *sol, const double *init, int nsim,
                     const double *randstream, long long nrand, int rndtype, double *sims)
{
    ms_env E;
    int nt = d->T - d->t0 + 1, nout = 11 + MS_NNST + MS_NND + MS_NEQ, isim, it, i, terr = 0;
    long long k, total = (long long)nout * nt * nsim;
    E.t0 = d->t0;
    E.T = d->T;
    E.ngridm = d->ngridm;
    E.ngridmax = d->ngridmax;
    E.nthrhmax = d->nthrhmax;
    E.ny = d->ny;
    E.mmax = d->mmax;
    E.a0 = d->a0;
    E.par = par;
    if (rndtype == 1 && nrand < 4LL * nt) return -2;
    if (rndtype == 0 && nrand < 4LL * nsim * nt) return -3;
    for (k = 1; k < total; k++) sims[k] = NAN; /* NaN fill starts at element 1 (:105); element 0 stays 0 */
    if (total > 0) sims[0] = 0.0;
    for (isim = 0; isim < nsim; isim++) {
        const double *rs = rndtype == 1 ? randstream : randstream + 4LL * nt * isim;
        int ist0 = (int)init[isim] - 1, irnd = 0;
        double m0 = init[nsim + isim];
        simrow_t sp[2], *cp = &sp[0], *np_;
        memset(sp, 0, sizeof sp);
        if (ist0 < 0 || ist0 >= MS_NST) continue;
        if (m0 < d->a0 || m0 > d->mmax) continue;
        for (it = 0; it < nt; it++) {
            if (it == 0) {
                cp->pv.it = 0;
                cp->pv.ist = ist0;
                cp->pv.cash = m0;
#if MS_NCONT > 0
                sp[0].pv.byval = sp[1].pv.byval = 1; /* :91-92: model functions read st[] and dc[] by value */
                for (i = 0; i < MS_NNST; i++) cp->pv.st[i] = ms_states[ist0 + i * MS_NST];
#endif
                if (!ms_feasible(&E, &cp->pv)) break;
                cp->mu = cp->sigma = cp->pv.shock = NAN;
                ms_eqs_sim(&E, &cp->pv, &cp->pv, 0, cp->eqs);
            } else {
                double r0, r1, r2, pr = 0;
                np_ = (cp == &sp[0]) ? &sp[1] : &sp[0];
                np_->pv.it = it;
                np_->pv.savings = cp->pv.savings;
                r0 = rs[irnd++];
                r1 = rs[irnd++];
                r2 = rs[irnd++];
                if (r2 > ms_survival(&E, &cp->pv)) break;
                for (np_->pv.ist = 0; np_->pv.ist < MS_NST; np_->pv.ist++) {
#if MS_NCONT > 0
                    /* only the first grid point of every continuous state is looked at: the state index keeps the
                       discrete variables, the continuous ones are carried by value (:274-280) */
                    for (i = 0; i < MS_NNST; i++)
                        if (ms_stcont[i] && (np_->pv.ist / ms_ststride[i]) % ms_stsize[i] != 0) break;
                    if (i < MS_NNST) continue;
                    for (i = 0; i < MS_NNST; i++)
                        if (!ms_stcont[i]) np_->pv.st[i] = ms_states[np_->pv.ist + i * MS_NST];
                    ms_trpr_cont(&E, &cp->pv, &np_->pv);
#endif
                    if (!ms_feasible(&E, &np_->pv)) continue;
                    if (MS_OPTIM_TRPRNOSH)
                        pr = ms_trpr_discrete(&E, &cp->pv, &np_->pv, &terr);
                    else {
                        np_->mu = ms_mu(&E, &cp->pv, &np_->pv);
                        np_->sigma = ms_sigma(&E, &cp->pv, &np_->pv);
                        if (np_->sigma <= 0)
                            np_->pv.shock = shock_expectation(&E, &cp->pv, &np_->pv);
                        else
                            np_->pv.shock = shock_from_uniform(r1, np_->mu, np_->sigma);
                        pr = ms_trpr_discrete(&E, &cp->pv, &np_->pv, &terr);
                    }
                    r0 -= pr;
                    if (r0 <= 0) break;
                }
                if (np_->pv.ist >= MS_NST) return -6; /* probabilities did not cover the draw (reference: out-of-range state) */
                if (MS_OPTIM_TRPRNOSH) {
                    np_->mu = ms_mu(&E, &cp->pv, &np_->pv);
                    np_->sigma = ms_sigma(&E, &cp->pv, &np_->pv);
                    if (np_->sigma <= 0)
                        np_->pv.shock = shock_expectation(&E, &cp->pv, &np_->pv);
                    else
                        np_->pv.shock = shock_from_uniform(r1, np_->mu, np_->sigma);
                }
                np_->pv.cash = ms_cashinhand(&E, &cp->pv, &np_->pv);
                ms_eqs_sim(&E, &cp->pv, &np_->pv, 1, np_->eqs);
                cp = np_;
            }
#if MS_NCONT > 0
            {   /* consumption interpolated over the 2^k grid corners around the continuous states, the (state, decision)
                   pair drawn among the corners by their weights with the fixed "random" number .5 (:318-372).  policy()
                   is called with dovf = 0 there, so the value function column is never computed: the reference adds up
                   whatever its uninitialised period buffer holds; a zeroed buffer gives 0, which is what is written here */
                int nw = 1 << MS_NCONT, ii, q, ist_base = cp->pv.ist, ist1 = -1, id1 = 0;
                double wts[1 << MS_NCONT], wc = 0, rr = .5;
                int wist[1 << MS_NCONT];
                for (ii = 0; ii < nw; ii++) wts[ii] = 1, wist[ii] = ist_base;
                for (q = 0, i = 0; i < MS_NNST; i++) {
                    const double *g;
                    int j1;
                    if (!ms_stcont[i]) continue;
                    g = ms_stgrid(i);
                    j1 = ms_bxsearch(cp->pv.st[i], g, ms_stsize[i]);
                    for (ii = 0; ii < nw; ii++) {
                        if ((ii >> q) % 2 == 0) {
                            wts[ii] *= (g[j1 + 1] - cp->pv.st[i]) / (g[j1 + 1] - g[j1]);
                            wist[ii] += ms_ststride[i] * j1;
                        } else {
                            wts[ii] *= (cp->pv.st[i] - g[j1]) / (g[j1 + 1] - g[j1]);
                            wist[ii] += ms_ststride[i] * (j1 + 1);
                        }
                    }
                    q++;
                }
                for (ii = 0; ii < nw; ii++) {
                    if (!(wts[ii] > 0)) continue;
                    cp->pv.ist = wist[ii];
                    if (cp->pv.ist < 0 || cp->pv.ist >= MS_NST || sol->len[(size_t)it * MS_NST + cp->pv.ist] < 2) return -4;
                    sim_policy(&E, d, sol, cp);
                    wc += cp->c * wts[ii];
                    rr -= wts[ii];
                    if (rr < 0 && ist1 == -1) ist1 = wist[ii], id1 = cp->pv.id;
                }
                cp->c = MS_MIN(wc, cp->pv.cash - d->a0);
                cp->pv.savings = cp->pv.cash - cp->c;
                cp->vf = 0.0;
                cp->pv.ist = ist1;
                cp->pv.id = id1;
                if (ist1 < 0) return -4;
                for (i = 0; i < MS_NND; i++) cp->pv.dc[i] = ms_decisions[cp->pv.id + i * MS_ND];
            }
#else
            if (cp->pv.ist >= MS_NST || sol->len[(size_t)it * MS_NST + cp->pv.ist] < 2) return -4; /* "Solution not found" */
            sim_policy(&E, d, sol, cp);
#endif
            {
                double *o = sims + ((size_t)isim * nt + it) * nout;
                o[0] = cp->pv.cash;
                o[1] = cp->c;
                o[2] = cp->pv.savings;
                o[3] = cp->vf;
                o[4] = (double)cp->pv.id;
                o[5] = (double)cp->pv.ist;
                o[6] = cp->mu;
                o[7] = cp->sigma;
                o[8] = cp->pv.shock;
                o[9] = ms_utility(&E, &cp->pv, cp->c);
                o[10] = ms_discount(&E, &cp->pv);
#if MS_NCONT > 0
                for (i = 0; i < MS_NNST; i++) o[11 + i] = cp->pv.st[i];   /* exact values of the continuous states (:138) */
#else
                for (i = 0; i < MS_NNST; i++) o[11 + i] = ms_states[cp->pv.ist + i * MS_NST];
#endif
                for (i = 0; i < MS_NND; i++) o[11 + MS_NNST + i] = ms_decisions[cp->pv.id + i * MS_ND];
                for (i = 0; i < MS_NEQ; i++) o[11 + MS_NNST + MS_NND + i] = cp->eqs[i];
            }
        }
    }
    return terr ? -5 : 0;
}

/* model constants for the harness */
void egdst_oracle_info(int *out)
{
    out[0] = MS_NST;
    out[1] = MS_ND;
    out[2] = MS_NNST;
    out[3] = MS_NND;
    out[4] = MS_NPARAM;
    out[5] = MS_NEQ;
}


/* ------------------------------------------------------------------------------------------ */
/* egdst_call.c:17-164 -- model-function accessor behind egdstmodel.call (egdstmodel.m:1181-1207):
 *   sw 1 utility(it,ist,id,c)  2 marginal utility  3 discount(it,ist)  4 budget(it,ist,id,savings,ist1,shock)
 *   5 marginal budget  6 value function(it,ist,cash) from the solved tables (vf(), :127-164).
 * args: [narg x ncol] column-major, 1-based it/ist/id as in MATLAB; res[narg] starts as zeros (mxCreateDoubleMatrix).
 * Kept from the reference: an out-of-range it/ist/id sets the switch to -1 for the rest of the call (that row and all
 * later rows are NaN); a wrong column count returns at the first row, leaving zeros; a bad ist1 leaves its own row
 * zero.  Differences where the reference reads past an array or an uninitialised variable: ist, id, ist1 equal to
 * nst+1 / nd+1 pass its `>` checks (:52,56,97) -- rejected here; `value` at the terminal period uses curr.id without
 * setting it (:121) -- id 1 here; a cash value below the first threshold indexes D[-1] (:152) -- D[0] here. */
int egdst_oracle_call(const orc_desc *d, const double *par, const orc_solution *sol, int sw, int narg, int ncol,
                      const double *args, double *res)
{
    ms_env E;
    int i, nt = d->T - d->t0 + 1, stride = d->ngridmax + 1;
    E.t0 = d->t0, E.T = d->T, E.ngridm = d->ngridm, E.ngridmax = d->ngridmax, E.nthrhmax = d->nthrhmax, E.ny = d->ny;
    E.mmax = d->mmax, E.a0 = d->a0, E.par = par;
    for (i = 0; i < narg; i++) res[i] = 0.0;
    for (i = 0; i < narg; i++) {
        const double *arg = args + i;
        ms_pv cur, nxt;
        memset(&cur, 0, sizeof cur);
        memset(&nxt, 0, sizeof nxt);
        cur.it = (int)arg[0 * (size_t)narg] - d->t0;
        if (cur.it < 0 || cur.it > nt - 1) sw = -1;
        cur.ist = (ncol > 1) ? (int)arg[1 * (size_t)narg] - 1 : -1;
        if (cur.ist < 0 || cur.ist >= MS_NST) sw = -1;
        if (ncol > 2 && sw != 6) {
            cur.id = (int)arg[2 * (size_t)narg] - 1;
            if (cur.id < 0 || cur.id >= MS_ND) sw = -1;
        }
        switch (sw) {
        case 1:
        case 2:
            if (ncol != 4) return 0;
            if (arg[3 * (size_t)narg] > d->mmax - d->a0)
                res[i] = NAN;
            else
                res[i] = sw == 1 ? ms_utility(&E, &cur, arg[3 * (size_t)narg]) : ms_utility_marginal(&E, &cur, arg[3 * (size_t)narg]);
            break;
        case 3:
            if (ncol != 2) return 0;
            res[i] = ms_discount(&E, &cur);
            break;
        case 4:
        case 5:
            if (ncol != 6) return 0;
            nxt.it = cur.it + 1;
            nxt.savings = arg[3 * (size_t)narg];
            nxt.ist = (int)arg[4 * (size_t)narg] - 1;
            nxt.shock = arg[5 * (size_t)narg];
            if (nxt.it < 0 || nxt.it > nt - 1)
                res[i] = NAN;
            else if (nxt.savings < d->a0)
                res[i] = NAN;
            else if (nxt.ist < 0 || nxt.ist >= MS_NST)
                sw = -1; /* (this row keeps its zero) */
            else
                res[i] = sw == 4 ? ms_cashinhand(&E, &cur, &nxt) : ms_cashinhand_marginal(&E, &cur, &nxt);
            break;
        case 6:
            if (ncol != 3) return 0;
            cur.cash = arg[2 * (size_t)narg];
            if (cur.cash > d->mmax)
                res[i] = NAN;
            else if (cur.it == nt - 1)
                res[i] = ms_utility(&E, &cur, MS_MAX(0, cur.cash));
            else {
                size_t sl = (size_t)cur.it * MS_NST + cur.ist;
                int nm = sol->len[sl], nth = sol->thlen[sl], ith = 0;
                const double *gm = sol->M + sl * stride, *gc = sol->C + sl * stride, *gv = sol->V + sl * stride;
                const double *th = sol->TH + sl * d->nthrhmax, *dd = sol->D + sl * d->nthrhmax;
                if (nm <= 0)
                    res[i] = NAN; /* "Solution missing for given it,ist.." */
                else if (nm < 2)
                    res[i] = -1.0; /* linter's error return (egdst_lib.c:166) */
                else {
                    double ma0 = gm[1], evf = gv[0], c = interp_lin(cur.cash, nm, gm, gc);
                    cur.savings = cur.cash - c;
                    while (ith < nth && cur.cash >= th[ith]) ith++;
                    cur.id = (int)dd[ith > 0 ? ith - 1 : 0];
                    if (cur.cash < ma0 && evf > -INFINITY)
                        res[i] = ms_utility(&E, &cur, c) + ms_discount(&E, &cur) * evf;
                    else if (cur.cash < ma0 && evf == -INFINITY)
                        res[i] = -INFINITY;
                    else
                        res[i] = interp_lin(cur.cash, nm, gm, gv);
                }
            }
            break;
        default:
            res[i] = NAN;
        }
    }
    return 0;
}
