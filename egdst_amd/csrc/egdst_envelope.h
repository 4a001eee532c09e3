// egdst_envelope.h -- upper envelopes of tabulated value functions on the device.
//
// Restates envelop/funcvalue/linter2/thresholds/brsolve (egdst_solver.c:1165-1968) over SoA arrays:
// the points are already sorted by (M asc, V desc, function asc) (comp1, :1570-1582) by the
// rank-merge of the calling kernel; `rank[fstart[f]+k]` is the sorted position of the k-th point
// of function f.  The walk is data-dependent: one wave runs it, 64 sorted positions per step where nothing
// irregular happens (env_walk_wave) and a wave-cooperative generic step at the events (env_step_wave); the plain
// sequential restatement (env_step, env_crossing) is kept for the diagnostic build EGDST_SEQ_WALK.  The recursion
// of thresholds() is unrolled onto an explicit stack of pending (previous, entering) function pairs.
#pragma once
#include "egdst_device.h"

#ifndef EG_WAVE
#define EG_WAVE 64
#endif

// Address spaces are kept in the TYPES: the sorted stream is either LDS- or global-resident (template
// parameter L, see EgMem), the small per-function arrays and evf(a0) are always LDS.  Generic (flat) pointers into LDS are
// avoided on purpose -- selecting between LDS- and global-derived generic pointers made hipcc's backend fail
// ("Illegal instruction detected ... src_shared_base") on one model, and typed LDS accesses are ds_* ops anyway.
// (EG_LDS_AS, eg_ldsd, eg_ldsi: typed LDS pointers, defined in egdst_device.h)
// L = 0: the stream in global memory;  1: in LDS;  2: in LDS except the consumption column, which the walk only copies to
// its output rows (and reads at a kink) -- 24 B per point instead of 32, for the one-wave walks of the throughput path.
template <int L> struct EgMem {
    typedef double D;
    typedef double DC;  // the consumption column
    typedef int I;
    typedef int S;  // function ids and sorted positions
};
template <> struct EgMem<1> {
    typedef eg_ldsd D;
    typedef eg_ldsd DC;
    typedef eg_ldsi I;
    typedef eg_ldss S;  // 16 bits in LDS: 32 B per point instead of 36
};
template <> struct EgMem<2> {
    typedef eg_ldsd D;
    typedef double DC;
    typedef eg_ldsi I;
    typedef eg_ldss S;
};

template <int L> struct EnvCtxT {
    const ms_env *E;
    int it, ist, nf;
    const typename EgMem<L>::D *m, *v;      // sorted points
    const typename EgMem<L>::DC *c;
    const typename EgMem<L>::S *f;
    const typename EgMem<L>::S *rank;       // position lists
    const eg_ldsi *fstart;
    eg_ldsi *dims, *cur, *mark;
    const typename EgMem<L>::I *cls;  // per sorted position: pre-classification word (env_preclass)
    eg_ldsi *stack;
    int stackcap;
    // expected value at a0 per function: primary -> evfa0[f]; secondary -> (f==sec_id ? sec_ev : -inf)
    const eg_ldsd *evfa0;
    int sec_id;
    double sec_ev;
    double *og, *ov, *oc, *oth, *oix;
    int oi, oj, ocap, e13, nthrhmax;  // ocap rows fit the output; filling it is error e13 (13, or CAPACITY when ocap is physical)
    int cap, npts;  // capacity of rank[] and number of sorted points (bounds guard)
    double bound;   // min over functions of the last grid point
    double lastg;   // grid value of the last output row (og[oi-1]); kept in a register so the walk never reads og back
    int pm;         // function that is currently the max ((int)oix[oj-1])
    int ci;         // the reference's `ci` (persists across iterations)
    int *dbg;       // [16] diagnostics of a tripped guard
    double *klog;   // kink log of this cell (the gateway's dbgout, egdst_solver.c:1866-1879): 4 doubles per recorded kink
    int *kcnt;      // -- (choice whose secondary envelope runs or -1, threshold, consumption left, right); nullptr: off
    int kcap;
    int later;      // 1: this context walks a LATER SEGMENT of the stream (run_walk): rows and thresholds exist before it
    int iend;       // first sorted position the walk did not consume (env_walk_wave)
    int err;
};

// one row of the solver gateway's third output (lane 0 of the walking wave calls this)
template <int L> static __device__ __forceinline__ void env_log_kink(EnvCtxT<L> &e, double x, double pol0, double pol1)
{
    if (!e.klog) return;
    const int n = *e.kcnt;
    if (n < e.kcap) {
        double *o = e.klog + 4 * (size_t)n;
        o[0] = (double)e.sec_id;  // (-1 in the primary envelope: dbgoutd, :718,804)
        o[1] = x;
        o[2] = pol0;
        o[3] = pol1;
    }
    *e.kcnt = n + 1;
}

template <int L> static __device__ __forceinline__ double env_evf(const EnvCtxT<L> &e, int f)
{
    if (e.sec_id >= 0) return f == e.sec_id ? e.sec_ev : -INFINITY;
    return e.evfa0[f];
}
template <int L> static __device__ __forceinline__ int env_at(EnvCtxT<L> &e, int f, int k)
{
    const int o = e.fstart[f] + k;
    if (o < 0 || o >= e.cap) {  // never expected; turns a wild access into an error code
        if (!e.err && e.dbg && atomicCAS(&e.dbg[0], 0, 2701) == 0) {
            int n0 = 0, n1 = 0;
            for (int q = 0; q < e.npts; q++) n0 += (e.f[q] == 0), n1 += (e.f[q] == 1);
            e.dbg[1] = f, e.dbg[2] = k, e.dbg[3] = e.ist, e.dbg[4] = e.it, e.dbg[5] = e.nf;
            e.dbg[6] = e.npts, e.dbg[7] = e.sec_id, e.dbg[8] = e.oi, e.dbg[9] = e.oj;
            e.dbg[10] = e.fstart[0], e.dbg[11] = e.fstart[1], e.dbg[12] = e.dims[0], e.dbg[13] = e.dims[1];
            e.dbg[14] = n0, e.dbg[15] = n1;
        }
        e.err = 2701;
        return 0;
    }
    const int r = e.rank[o];
    if (r < 0 || r >= e.npts) {
        e.err = 2702;
        return 0;
    }
    return r;
}

// value (which=0) or consumption (which=1) of f on the segment that starts at its k-th point; -inf outside
// the segment: "No extrapolation allowed: this is essential for the correct envelop" (linter2, :1585-1593)
template <int L> static __device__ __forceinline__ double env_seg(EnvCtxT<L> &e, int f, int k, double x, int which)
{
    int a = env_at(e, f, k), b = env_at(e, f, k + 1);
    double ga = e.m[a], gb = e.m[b];
    double fa, fb;  // (two loads each way: the columns may live in different address spaces)
    if (which)
        fa = e.c[a], fb = e.c[b];
    else
        fa = e.v[a], fb = e.v[b];
    if (x == ga) return fa;
    if (x < ga) return -INFINITY;
    if (x > gb) return -INFINITY;
    return fb * (x - ga) / (gb - ga) + fa * (gb - x) / (gb - ga);
}

template <int L> static __device__ __forceinline__ double env_analytic(const EnvCtxT<L> &e, int f, double x)
{
    ms_pv cv;
    cv.it = e.it;
    cv.ist = e.ist;
    cv.id = f;
    cv.cash = cv.savings = cv.shock = 0;
    return ms_utility(e.E, &cv, x - e.E->a0) + ms_discount(e.E, &cv) * env_evf(e, f);
}

template <int L> static __device__ __forceinline__ double env_fn(EnvCtxT<L> &e, int f, double x)  // funcvalue, :1553-1567
{
    if (e.cur[f] >= 0) return env_seg(e, f, e.cur[f], x, 0);
    if (env_evf(e, f) == -INFINITY) return -INFINITY;
    return env_analytic(e, f, x);
}

template <int L> static __device__ __forceinline__ double env_policy(EnvCtxT<L> &e, int f, double x)  // :1406-1408, :1859-1864
{
    if (e.cur[f] >= 0) return env_seg(e, f, e.cur[f], x, 1);
    if (env_evf(e, f) == -INFINITY) return EG_ZEROC;
    return x - e.E->a0;
}

static __device__ __forceinline__ double env_sgn(double x) { return x > 0 ? 1.0 : -1.0; }

// brsolve (:1918-1968): bisection between an analytic value function `fa` and the segment (fl,kl)
template <int L> static __device__ __forceinline__ void env_bisect(EnvCtxT<L> &e, double *b0, double *b1, int fl, int kl, int fa)
{
    // The reference re-evaluates both bracket ends at every level of its recursion; they are pure functions of
    // the bracket, so the values are carried along instead (bit-identical, a third of the evaluations).
    // d[q] = analytic(b[q]) - segment(b[q]), a[q] = analytic(b[q]);  q = 0,1 the bracket ends, q = 2 the midpoint.
    double bq[3] = {*b0, *b1, 0.0}, a[3], d[3];
    int have = 0;  // ends already evaluated
    for (;;) {
        for (int q = have; q < 2; q++) {
            a[q] = env_analytic(e, fa, bq[q]);
            d[q] = a[q] - env_seg(e, fl, kl, bq[q], 0);
        }
        have = 2;
        const double s0 = env_sgn(d[0]), s1 = env_sgn(d[1]);
        if (s0 == s1) {
            e.err = 22;
            return;
        }
        if (bq[0] > bq[1]) {
            e.err = 23;
            return;
        }
        if (fabs(bq[0] - bq[1]) < 2 * EG_DPD || fabs(a[0] - a[1]) < EG_DPD) {
            *b0 = (bq[0] + bq[1]) / 2;
            *b1 = bq[1];
            return;
        }
        bq[2] = (bq[0] + bq[1]) / 2;
        a[2] = env_analytic(e, fa, bq[2]);
        d[2] = a[2] - env_seg(e, fl, kl, bq[2], 0);
        const double sm = env_sgn(d[2]);
        if (s0 == sm)
            bq[0] = bq[2], a[0] = a[2], d[0] = d[2];
        else if (s1 == sm)
            bq[1] = bq[2], a[1] = a[2], d[1] = d[2];
        else {
            *b0 = bq[0];
            *b1 = bq[1];
            return;
        }
    }
}

// thresholds (:1596-1915)
template <int L> static __device__ __forceinline__ void env_crossing(EnvCtxT<L> &e, int pri0, int nwi0, int mode)
{
    const double a0 = e.E->a0;
    int sp = 0;
    e.stack[0] = pri0;
    e.stack[1] = nwi0;
    sp = 1;
    while (sp > 0 && !e.err) {
        sp--;
        const int pri = e.stack[2 * sp], nwi = e.stack[2 * sp + 1];
        e.mark[pri] = 1;
        e.mark[nwi] = 1;
        const int cp = e.cur[pri], cn = e.cur[nwi];
        double x = 0, top = 0;
        if ((cp == -1) != (cn == -1)) {  // exactly one of the two is still in its analytic region (:1648-1688)
            const int ana = (cp == -1) ? pri : nwi, lin = (cp == -1) ? nwi : pri, kl = (cp == -1) ? cn : cp;
            if (env_evf(e, ana) == -INFINITY)
                x = e.m[env_at(e, ana, 0)];
            else {
                double br0 = e.m[env_at(e, lin, kl)];
                double br1 = MS_MIN(e.m[env_at(e, ana, 0)], e.m[env_at(e, lin, kl + 1)]);
                env_bisect(e, &br0, &br1, lin, kl, ana);
                if (e.err) return;
                x = br0;
            }
            top = env_seg(e, lin, kl, x, 0);
        } else if (cp == -1 && cn == -1) {
            e.err = 21;
            return;
        } else {
            const int ip0 = env_at(e, pri, cp), ip1 = env_at(e, pri, cp + 1);
            const int in0 = env_at(e, nwi, cn), in1 = env_at(e, nwi, cn + 1);
            const double p0m = e.m[ip0], p1m = e.m[ip1], p0v = e.v[ip0], p1v = e.v[ip1];
            const double n0m = e.m[in0], n1m = e.m[in1], n0v = e.v[in0], n1v = e.v[in1];
            const double icn = (n0v * n1m - n1v * n0m) / (n1m - n0m);  // intercepts
            const double icp = (p0v * p1m - p1v * p0m) / (p1m - p0m);
            if (p1m == p0m) {  // previous max is vertical
                x = p0m;
                top = (x * (n1v - n0v) / (n1m - n0m)) + icn;
            } else if (n1m == n0m) {  // entering function is vertical
                x = n0m;
                top = (x * (p1v - p0v) / (p1m - p0m)) + icp;
            } else if (((n1v - n0v) / (n1m - n0m)) == ((p1v - p0v) / (p1m - p0m))) {  // identical slopes
                x = (p0m + p1m + n0m + n1m) / 4;
                top = (x * (n1v - n0v) / (n1m - n0m)) + icn;
            } else {
                x = (icp - icn) / (((n1v - n0v) / (n1m - n0m)) - ((p1v - p0v) / (p1m - p0m)));
                top = (x * (n1v - n0v) / (n1m - n0m)) + icn;
            }
        }
        int best = -1;
        for (int k = 0; k < e.nf; k++) {
            if (e.mark[k] == 1) continue;
            // u(.)+beta*(-inf) is -inf; skipping the call keeps segment indices out of the model's id tables
            double t = (e.cur[k] >= 0) ? env_seg(e, k, e.cur[k], x, 0)
                                       : (env_evf(e, k) == -INFINITY ? -INFINITY : env_analytic(e, k, x));
            if (top < t) {
                top = t;
                best = k;
                if (mode == 0) break;
            }
        }
        if (best != -1) {  // a third function is higher at the crossing: split (:1827-1845)
            if (2 * (sp + 2) > e.stackcap) {
                e.err = 2703;
                return;
            }
            if (mode != 0) {
                e.stack[2 * sp] = best;
                e.stack[2 * sp + 1] = nwi;
                sp++;
            }
            e.stack[2 * sp] = pri;
            e.stack[2 * sp + 1] = best;
            sp++;
            continue;
        }
        const double pol0 = env_policy(e, pri, x), pol1 = env_policy(e, nwi, x);
        double gx = x;  // grid value of the row written last (kept for the duplicate test)
        e.og[e.oi] = x;
        e.ov[e.oi] = top;
        e.oc[e.oi] = (pol0 + pol1) / 2;
        env_log_kink(e, x, pol0, pol1);
        e.oth[e.oj] = x;
        e.oix[e.oj] = nwi;
        e.pm = nwi;
        e.oi += 1;
        e.oj += 1;
        if (e.oi >= e.ocap) {
            e.err = e.e13;
            return;
        }
        if (e.oj >= e.nthrhmax) {
            e.err = 20;
            return;
        }
        if (env_evf(e, nwi) == -INFINITY && e.cur[nwi] == -1) {  // :1892-1900
            e.oc[e.oi - 1] = pol0;
            gx = x - EG_TOL;
            e.og[e.oi - 1] = gx;
        } else if (EG_DPD > 0) {  // double point at the kink, :1902-1913
            e.oc[e.oi - 1] = pol0;
            gx = x + EG_DPD;
            e.og[e.oi] = gx;
            e.ov[e.oi] = top;
            e.oc[e.oi] = pol1;
            e.oi += 1;
            if (e.oi >= e.ocap) {
                e.err = e.e13;
                return;
            }
        }
        e.lastg = gx;
    }
}

template <int L> static __device__ __forceinline__ void env_reset_marks(EnvCtxT<L> &e)
{
    for (int l = 0; l < e.nf; l++) e.mark[l] = (e.dims[l] > 0 ? 0 : 1);
}
template <int L> static __device__ __forceinline__ void env_push(EnvCtxT<L> &e, double g, double v, double c)
{
    e.og[e.oi] = g;
    e.ov[e.oi] = v;
    e.oc[e.oi] = c;
    e.oi++;
    e.lastg = g;
}

// ---- the merge walk (:1262-1550) ------------------------------------------------------------------------
// env_begin: state before the first point; env_step: one iteration of the reference's while loop for the
// sorted position i (returns false when the walk must stop).  dims[] must hold the points per function.
template <int L> static __device__ __forceinline__ void env_begin(EnvCtxT<L> &e)
{
    for (int f = 0; f < e.nf; f++) e.cur[f] = -1;
    double bound = INFINITY;  // min over functions of the last grid point (:1266-1271)
    for (int f = 0; f < e.nf; f++)
        if (e.dims[f] > 0) {
            double last = e.m[env_at(e, f, e.dims[f] - 1)];
            if (last < bound) bound = last;
        }
    e.bound = bound;
    e.oi = e.oj = 0;
    e.ci = 0;
    e.lastg = -INFINITY;
    e.pm = -1;
}

template <int L> static __device__ __forceinline__ bool env_step(EnvCtxT<L> &e, int i)
{
    const double a0 = e.E->a0, bound = e.bound;
    const int f = e.f[i];
    const double x = e.m[i];
    if (f < 0 || f >= e.nf || e.dims[f] <= 0) {  // sorted stream inconsistent with the per-function lists
        if (e.dbg && atomicCAS(&e.dbg[0], 0, 2708) == 0)
            e.dbg[1] = f, e.dbg[2] = i, e.dbg[3] = e.npts, e.dbg[4] = e.nf, e.dbg[5] = e.sec_id, e.dbg[6] = e.ist;
        e.err = 2708;
        return false;
    }
    if ((e.oi > 0 || e.later) && e.lastg == x) {  // duplicate grid point (:1290-1298)
        e.cur[f]++;
        return true;
    }
    const int self = env_at(e, f, e.cur[f] + 1);
    double fv = e.v[self];
    if (e.oj == 0 && !e.later) {  // first point of the common grid (:1303-1347)
        double t = fv;
        e.ci = f;
        for (int j = 0; j < e.nf; j++) {
            if (e.dims[j] <= 0 || j == f) continue;
            fv = env_fn(e, j, x);
            if (fv > t) t = fv, e.ci = j;
            if (fv == t && e.ci > j) e.ci = j;
        }
        e.oth[e.oj] = a0;
        e.oix[e.oj] = e.ci;
        e.pm = e.ci;
        e.oj++;
        if (e.oj >= e.nthrhmax) {
            e.err = 20;
            return false;
        }
        if (e.ci == f) {
            env_push(e, x, t, e.c[self]);
            if (e.oi >= e.ocap) {
                e.err = e.e13;
                return false;
            }
        }
    } else {
        // Both remaining cases of the reference (:1348-1416 the point belongs to the current max function,
        // :1417-1516 it belongs to another one) end in at most one thresholds() call followed by a small
        // case-specific epilogue; the call is written once (code size: it inlines the whole crossing search).
        int xa = -1, xb = -1, xmode = 0, post = 0;  // post: 0 nothing, 1 last row of ci, 2 push own point, 3 last row of cj
        int cj = -1;
        if (e.pm == f) {  // point of the current max function
            int above = 0, j;
            double t;
            for (j = 0; j < e.nf; j++) {
                if (e.dims[j] <= 0 || j == f) continue;
                t = env_fn(e, j, x);
                if (fv < t) {
                    above = 1;
                    if (x != bound) break;
                    fv = t;
                    e.ci = j;
                }
            }
            if (!above) {
                env_push(e, x, fv, e.c[self]);
                if (e.oi == e.ocap) {
                    e.err = e.e13;
                    return false;
                }
            } else if (x != bound) {
                xa = f, xb = j, xmode = 0, post = 0;
            } else {
                xa = f, xb = e.ci, xmode = 1, post = 1;
            }
        } else {  // point of another function
            e.ci = e.pm;
            double t = env_fn(e, e.ci, x);
            if (t < fv) {
                for (int j = 0; j < e.nf; j++) {
                    if (e.dims[j] <= 0 || j == f || j == e.ci) continue;
                    t = env_fn(e, j, x);
                    if ((fv < t) || (fv == t && j < cj)) fv = t, cj = j;
                }
                if (cj == -1)
                    xa = e.ci, xb = f, xmode = 1, post = 2;
                else
                    xa = e.ci, xb = cj, xmode = 1, post = (x == bound) ? 3 : 0;
            } else if (x == bound) {
                double vv = env_fn(e, e.ci, x), pp = env_policy(e, e.ci, x);
                env_push(e, x, vv, pp);
            }
        }
        if (xa >= 0) {
            env_reset_marks(e);
            env_crossing(e, xa, xb, xmode);
            if (e.err) return false;
            if (post == 1) {
                e.lastg = x;
                e.og[e.oi] = x;
                e.ov[e.oi] = env_fn(e, e.ci, x);
                // (:1406-1408; the reference indexes evfa0 with the exhausted loop variable there)
                e.oc[e.oi] = (e.cur[e.ci] >= 0) ? env_seg(e, e.ci, e.cur[e.ci], x, 1) : x - a0;
                e.oi++;
                if (e.oi >= e.ocap) {
                    e.err = e.e13;
                    return false;
                }
            } else if (post == 2) {
                env_push(e, x, fv, e.c[self]);
                if (e.oi >= e.ocap) {
                    e.err = e.e13;
                    return false;
                }
            } else if (post == 3) {
                double vv = env_fn(e, cj, x), pp = env_policy(e, cj, x);
                env_push(e, x, vv, pp);
            }
        }
    }
    e.cur[f] = MS_MIN(e.cur[f] + 1, e.dims[f] - 2);
    return true;
}

// ---- wave-cooperative generic step ----------------------------------------------------------------------
// env_step()/env_crossing() executed by ALL lanes of the walking wave with uniform control flow: the scalar state of
// the walk (oi, oj, ci, pm, lastg, err ...) lives in every lane's registers and is updated identically; the shared
// arrays (cur, mark, stack, the output rows) are written by lane 0 only; and every loop over the functions
// ("is a function above this point", "which function is highest", the third-function test at a crossing) gives one
// function to each lane and combines the lanes with the reference's own tie rules (strictly-greater wins, ties go
// to the smaller index; NaN candidates never win, as in the reference's comparisons).  Same arithmetic per function
// as the sequential code, so the result is bit-identical -- the CPU harness checks that.
#ifdef EGDST_EMU
#define EG_WSYNC() ((void)__ballot(1))   // the harness runs lanes as threads: order lane 0's stores before the readers
#else
#define EG_WSYNC() __builtin_amdgcn_wave_barrier()  // same wave, in-order LDS: only the compiler must not reorder
#endif

// (value, index) with the larger value, ties to the smaller index; idx < 0 = absent
// The candidates sit on the first min(nf, wave) lanes: a butterfly over the next power of two, then a broadcast from
// lane 0 (2 rounds for the 2 functions of a typical primary envelope instead of 6).
static __device__ __forceinline__ int env_wave_width(int nf)
{
    int w = 1;
    while (w < nf && w < EG_WAVE) w <<= 1;
    return w;
}
static __device__ __forceinline__ void env_wave_argmax(double *v, int *idx, int nf)
{
    for (int o = env_wave_width(nf) >> 1; o > 0; o >>= 1) {
        const double ov = __shfl_xor(*v, o);
        const int oi = __shfl_xor(*idx, o);
        if (oi >= 0 && (*idx < 0 || ov > *v || (ov == *v && oi < *idx))) *v = ov, *idx = oi;
    }
    *v = __shfl(*v, 0);
    *idx = __shfl(*idx, 0);
}
static __device__ __forceinline__ int env_wave_min(int x, int nf)
{
    for (int o = env_wave_width(nf) >> 1; o > 0; o >>= 1) x = min(x, __shfl_xor(x, o));
    return __shfl(x, 0);
}

// Over the functions j with dims[j] > 0, j != ex1, j != ex2 (and, with marks, mark[j] != 1):
//   first != 0: the smallest j with thr < value_j(x);   first == 0: the j of the largest value_j(x) > thr,
//   smallest index among equals.  Returns j (or -1) and its value in *val.
template <int L>
static __device__ __forceinline__ int env_wave_pick(EnvCtxT<L> &e, double x, double thr, int ex1, int ex2, int marks, int first,
                                                    double *val)
{
    const int lane = threadIdx.x & (EG_WAVE - 1);
    double bv = 0;
    int bj = -1;
    for (int j = lane; j < e.nf; j += EG_WAVE) {
        if (marks ? (e.mark[j] == 1) : (e.dims[j] <= 0 || j == ex1 || j == ex2)) continue;
        if (first && bj >= 0) break;
        const double t = env_fn(e, j, x);
        if (thr < t && (bj < 0 || t > bv)) bv = t, bj = j;  // ascending j: an equal value keeps the earlier index
    }
    if (first) {
        const int j0 = env_wave_min(bj < 0 ? 0x7fffffff : bj, e.nf);
        if (j0 == 0x7fffffff) return -1;
        *val = __shfl(bv, j0 & (EG_WAVE - 1));
        return j0;
    }
    env_wave_argmax(&bv, &bj, e.nf);
    if (bj >= 0) *val = bv;
    return bj;
}

template <int L> static __device__ __forceinline__ void env_crossing_wave(EnvCtxT<L> &e, int pri0, int nwi0, int mode)
{
    const int lane = threadIdx.x & (EG_WAVE - 1);
    const double a0 = e.E->a0;
    int sp = 1;
    if (lane == 0) e.stack[0] = pri0, e.stack[1] = nwi0;
    while (sp > 0 && !e.err) {
        EG_WSYNC();
        sp--;
        const int pri = e.stack[2 * sp], nwi = e.stack[2 * sp + 1];
        EG_WSYNC();
        if (lane == 0) e.mark[pri] = 1, e.mark[nwi] = 1;
        const int cp = e.cur[pri], cn = e.cur[nwi];
        double x = 0, top = 0;
        if ((cp == -1) != (cn == -1)) {  // exactly one of the two is still in its analytic region (:1648-1688)
            const int ana = (cp == -1) ? pri : nwi, lin = (cp == -1) ? nwi : pri, kl = (cp == -1) ? cn : cp;
            if (env_evf(e, ana) == -INFINITY)
                x = e.m[env_at(e, ana, 0)];
            else {
                double br0 = e.m[env_at(e, lin, kl)];
                double br1 = MS_MIN(e.m[env_at(e, ana, 0)], e.m[env_at(e, lin, kl + 1)]);
                env_bisect(e, &br0, &br1, lin, kl, ana);
                if (e.err) return;
                x = br0;
            }
            top = env_seg(e, lin, kl, x, 0);
        } else if (cp == -1 && cn == -1) {
            e.err = 21;
            return;
        } else {
            const int ip0 = env_at(e, pri, cp), ip1 = env_at(e, pri, cp + 1);
            const int in0 = env_at(e, nwi, cn), in1 = env_at(e, nwi, cn + 1);
            const double p0m = e.m[ip0], p1m = e.m[ip1], p0v = e.v[ip0], p1v = e.v[ip1];
            const double n0m = e.m[in0], n1m = e.m[in1], n0v = e.v[in0], n1v = e.v[in1];
            const double icn = (n0v * n1m - n1v * n0m) / (n1m - n0m);  // intercepts
            const double icp = (p0v * p1m - p1v * p0m) / (p1m - p0m);
            if (p1m == p0m) {  // previous max is vertical
                x = p0m;
                top = (x * (n1v - n0v) / (n1m - n0m)) + icn;
            } else if (n1m == n0m) {  // entering function is vertical
                x = n0m;
                top = (x * (p1v - p0v) / (p1m - p0m)) + icp;
            } else if (((n1v - n0v) / (n1m - n0m)) == ((p1v - p0v) / (p1m - p0m))) {  // identical slopes
                x = (p0m + p1m + n0m + n1m) / 4;
                top = (x * (n1v - n0v) / (n1m - n0m)) + icn;
            } else {
                x = (icp - icn) / (((n1v - n0v) / (n1m - n0m)) - ((p1v - p0v) / (p1m - p0m)));
                top = (x * (n1v - n0v) / (n1m - n0m)) + icn;
            }
        }
        EG_WSYNC();  // the marks of pri and nwi are set
        const int best = env_wave_pick(e, x, top, -1, -1, 1, mode == 0, &top);
        if (best != -1) {  // a third function is higher at the crossing: split (:1827-1845)
            if (2 * (sp + 2) > e.stackcap) {
                e.err = 2703;
                return;
            }
            if (mode != 0) {
                if (lane == 0) e.stack[2 * sp] = best, e.stack[2 * sp + 1] = nwi;
                sp++;
            }
            if (lane == 0) e.stack[2 * sp] = pri, e.stack[2 * sp + 1] = best;
            sp++;
            continue;
        }
        const double pol0 = env_policy(e, pri, x), pol1 = env_policy(e, nwi, x);
        double gx = x;  // grid value of the row written last (kept for the duplicate test)
        if (lane == 0) {
            e.og[e.oi] = x;
            e.ov[e.oi] = top;
            e.oc[e.oi] = (pol0 + pol1) / 2;
            env_log_kink(e, x, pol0, pol1);
            e.oth[e.oj] = x;
            e.oix[e.oj] = nwi;
        }
        e.pm = nwi;
        e.oi += 1;
        e.oj += 1;
        if (e.oi >= e.ocap) {
            e.err = e.e13;
            return;
        }
        if (e.oj >= e.nthrhmax) {
            e.err = 20;
            return;
        }
        if (env_evf(e, nwi) == -INFINITY && e.cur[nwi] == -1) {  // :1892-1900
            gx = x - EG_TOL;
            if (lane == 0) {
                e.oc[e.oi - 1] = pol0;
                e.og[e.oi - 1] = gx;
            }
        } else if (EG_DPD > 0) {  // double point at the kink, :1902-1913
            gx = x + EG_DPD;
            if (lane == 0) {
                e.oc[e.oi - 1] = pol0;
                e.og[e.oi] = gx;
                e.ov[e.oi] = top;
                e.oc[e.oi] = pol1;
            }
            e.oi += 1;
            if (e.oi >= e.ocap) {
                e.err = e.e13;
                return;
            }
        }
        e.lastg = gx;
    }
}

template <int L> static __device__ __forceinline__ void env_push_wave(EnvCtxT<L> &e, double g, double v, double c, int lane)
{
    if (lane == 0) {
        e.og[e.oi] = g;
        e.ov[e.oi] = v;
        e.oc[e.oi] = c;
    }
    e.oi++;
    e.lastg = g;
}

template <int L> static __device__ __forceinline__ bool env_step_wave(EnvCtxT<L> &e, int i)
{
    const int lane = threadIdx.x & (EG_WAVE - 1);
    const double a0 = e.E->a0, bound = e.bound;
    EG_WSYNC();  // cursors written by the previous step / the rebuild are in place
    const int f = e.f[i];
    const double x = e.m[i];
    if (f < 0 || f >= e.nf || e.dims[f] <= 0) {  // sorted stream inconsistent with the per-function lists
        if (e.dbg && atomicCAS(&e.dbg[0], 0, 2708) == 0)
            e.dbg[1] = f, e.dbg[2] = i, e.dbg[3] = e.npts, e.dbg[4] = e.nf, e.dbg[5] = e.sec_id, e.dbg[6] = e.ist;
        e.err = 2708;
        return false;
    }
    const int curf = e.cur[f];
    if ((e.oi > 0 || e.later) && e.lastg == x) {  // duplicate grid point (:1290-1298)
        EG_WSYNC();
        if (lane == 0) e.cur[f] = curf + 1;
        return true;
    }
    const int self = env_at(e, f, curf + 1);
    double fv = e.v[self];
    if (e.oj == 0 && !e.later) {  // first point of the common grid (:1303-1347)
        // the highest function at x, own value included, smallest index among equals
        double t = fv;
        int cj = (lane == (f & (EG_WAVE - 1))) ? f : -1;
        if (t != t) cj = -1;  // (a NaN own value never loses in the reference: handled below)
        {
            double bv = t;
            int bj = cj;
            for (int j = lane; j < e.nf; j += EG_WAVE) {
                if (e.dims[j] <= 0 || j == f) continue;
                const double tj = env_fn(e, j, x);
                if (tj != tj) continue;
                if (bj < 0 || tj > bv || (tj == bv && j < bj)) bv = tj, bj = j;
            }
            env_wave_argmax(&bv, &bj, e.nf);
            if (fv != fv)
                e.ci = f;  // every comparison with NaN is false: nothing replaces the own point
            else
                t = bv, e.ci = bj;
        }
        if (lane == 0) {
            e.oth[e.oj] = a0;
            e.oix[e.oj] = e.ci;
        }
        e.pm = e.ci;
        e.oj++;
        if (e.oj >= e.nthrhmax) {
            e.err = 20;
            return false;
        }
        if (e.ci == f) {
            env_push_wave(e, x, t, e.c[self], lane);
            if (e.oi >= e.ocap) {
                e.err = e.e13;
                return false;
            }
        }
    } else {
        int xa = -1, xb = -1, xmode = 0, post = 0;  // post: 0 nothing, 1 last row of ci, 2 push own point, 3 last row of cj
        int cj = -1;
        if (e.pm == f) {  // point of the current max function
            double t = 0;
            const int j = env_wave_pick(e, x, fv, f, -1, 0, x != bound, &t);
            if (j < 0) {
                env_push_wave(e, x, fv, e.c[self], lane);
                if (e.oi == e.ocap) {
                    e.err = e.e13;
                    return false;
                }
            } else if (x != bound) {
                xa = f, xb = j, xmode = 0, post = 0;
            } else {
                fv = t;
                e.ci = j;
                xa = f, xb = e.ci, xmode = 1, post = 1;
            }
        } else {  // point of another function
            e.ci = e.pm;
            double t = env_fn(e, e.ci, x);
            if (t < fv) {
                double tv = 0;
                cj = env_wave_pick(e, x, fv, f, e.ci, 0, 0, &tv);
                if (cj == -1)
                    xa = e.ci, xb = f, xmode = 1, post = 2;
                else {
                    fv = tv;
                    xa = e.ci, xb = cj, xmode = 1, post = (x == bound) ? 3 : 0;
                }
            } else if (x == bound) {
                double vv = env_fn(e, e.ci, x), pp = env_policy(e, e.ci, x);
                env_push_wave(e, x, vv, pp, lane);
            }
        }
        if (xa >= 0) {
            for (int l = lane; l < e.nf; l += EG_WAVE) e.mark[l] = (e.dims[l] > 0 ? 0 : 1);
            env_crossing_wave(e, xa, xb, xmode);
            if (e.err) return false;
            if (post == 1) {
                e.lastg = x;
                const double vv = env_fn(e, e.ci, x);
                // (:1406-1408; the reference indexes evfa0 with the exhausted loop variable there)
                const double pp = (e.cur[e.ci] >= 0) ? env_seg(e, e.ci, e.cur[e.ci], x, 1) : x - a0;
                if (lane == 0) {
                    e.og[e.oi] = x;
                    e.ov[e.oi] = vv;
                    e.oc[e.oi] = pp;
                }
                e.oi++;
                if (e.oi >= e.ocap) {
                    e.err = e.e13;
                    return false;
                }
            } else if (post == 2) {
                env_push_wave(e, x, fv, e.c[self], lane);
                if (e.oi >= e.ocap) {
                    e.err = e.e13;
                    return false;
                }
            } else if (post == 3) {
                double vv = env_fn(e, cj, x), pp = env_policy(e, cj, x);
                env_push_wave(e, x, vv, pp, lane);
            }
        }
    }
    EG_WSYNC();  // every lane has read the cursors it needs
    if (lane == 0) e.cur[f] = MS_MIN(curf + 1, e.dims[f] - 2);
    return true;
}

// ---- generic step with the functions' state on the lanes -------------------------------------------------
// env_step_wave reaches every segment it evaluates through a chain of LDS reads (cursor -> start of the function's position
// list -> sorted position -> point), five or six times per step, one after the other: a step took 5-6 us (profiles/r03_*),
// nearly all of it waiting.  Here lane j loads the state of function j ONCE per step -- cursor, the two points of its current
// segment, its value at a0 -- and everything the step needs afterwards is a lane read (v_readlane with a uniform index) or the
// lane's own registers: the "which function is highest" loops are one ballot, the crossing reads its eight numbers from two
// lanes, the marks of thresholds() are a 64-bit mask.  Same control flow and the same arithmetic per function as
// env_step_wave (which stays for walks of more than a wave's worth of functions, and for any step whose lane state is not
// clean -- the guards of env_at); the CPU harness and the parity tests compare the two bit for bit (-DENV_LANE_STEP=0).
#ifndef ENV_LANE_STEP
#define ENV_LANE_STEP 1
#endif
// Which walks take it: those of the throughput path (L == 2, k_tp_walk: 168 VGPRs, nothing spilled).  k_envelope does not
// (-DENV_LANE_STEP_ENV=1 turns it on for its LDS-resident streams, L == 1): the kernel is at its register limit, and the second
// implementation of the step costs it spills -- 81 VGPRs against 30 in the default build, 569 against 222 in the 168-VGPR
// build for big batches.  Measured: C2 single solve 8.75 -> 8.45 ms, but C3 single solve 22.1 -> 25.0 ms, the leftover cells of
// the C2 a0 = -5 batch 357 -> 448 ms per step, and for streams in global memory (L == 0, never) the lanes load the segments of
// ALL functions at every step where env_step_wave touches the two or three it needs: C5 x 128 k_envelope 1.76 -> 1.98 s.
#ifndef ENV_LANE_STEP_ENV
#define ENV_LANE_STEP_ENV 0
#endif
#define ENV_LANES(L) (ENV_LANE_STEP && ((L) == 2 || ((L) == 1 && ENV_LANE_STEP_ENV)))
// Vector memory operations retire in order and share one counter (loads and stores alike, gfx9): where a register that a
// global load of the generic step wrote MAY still be pending, the compiler waits for the counter to reach zero before the
// register is written again -- and found such a place in every regular batch, which then waited for the row stores of the batch
// before it (a microsecond under load, for a loop iteration of a few hundred cycles).  An explicit wait after every generic step
// (15 per cell instead of 63) leaves the regular batches with nothing pending but their own stores, which nobody waits for.
#ifdef EGDST_EMU
#define EG_VM_DRAIN() ((void)0)
#else
#define EG_VM_DRAIN() __builtin_amdgcn_s_waitcnt(0x0F70)  // vmcnt(0), nothing else
#endif
#ifndef ENV_STEP_CHAIN
#define ENV_STEP_CHAIN 1  // consecutive irregular positions are stepped through without going back to the batches (env_walk_wave); 0: tests
#endif
#ifndef ENV_DUPRUN
#define ENV_DUPRUN 1  // runs of repeated grid values at the bound are consumed 64 positions at a time (env_walk_wave); 0: one generic step each (tests)
#endif
#ifndef ENV_CDEFER_ON
#define ENV_CDEFER_ON 1  // walks with the consumption column in global memory copy it to the kept rows after the walk (env_walk_wave)
#endif
#define ENV_CDEFER(L) ((L) == 2 && ENV_CDEFER_ON)
#ifdef EGDST_EMU
#define EG_RLI(v, l) __shfl((int)(v), (l))
#define EG_RLD(v, l) __shfl((double)(v), (l))
#else
static __device__ __forceinline__ int eg_rli_(int v, int l) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l)); }
static __device__ __forceinline__ double eg_rld_(double v, int l)
{
    const int sl = __builtin_amdgcn_readfirstlane(l);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), sl), __builtin_amdgcn_readlane(__double2loint(v), sl));
}
#define EG_RLI(v, l) eg_rli_((v), (l))
#define EG_RLD(v, l) eg_rld_((v), (l))
#endif

struct EnvLaneFn {  // function `lane` at a generic step
    int cur, dim;        // e.cur[lane], e.dims[lane]
    int pa, pb;          // sorted positions of its points cur and cur + 1 (the current segment; pa only when cur >= 0)
    double ga, gb, va, vb;
    double ca, cb;       // consumption at the two points (loaded with the rest: the column may live in global memory, and the loads
                         // are long done when a kink or a kept row needs them)
    double evf;          // env_evf(e, lane)
};

static __device__ __forceinline__ double env_seg_regs(double ga, double gb, double fa, double fb, double x)  // env_seg
{
    if (x == ga) return fa;
    if (x < ga) return -INFINITY;
    if (x > gb) return -INFINITY;
    return fb * (x - ga) / (gb - ga) + fa * (gb - x) / (gb - ga);
}
template <int L> static __device__ __forceinline__ double env_analytic_ev(const EnvCtxT<L> &e, int f, double x, double ev)  // env_analytic
{
    ms_pv cv;
    cv.it = e.it;
    cv.ist = e.ist;
    cv.id = f;
    cv.cash = cv.savings = cv.shock = 0;
    return ms_utility(e.E, &cv, x - e.E->a0) + ms_discount(e.E, &cv) * ev;
}
// env_fn of the lane's own function
template <int L> static __device__ __forceinline__ double env_fn_own(const EnvCtxT<L> &e, const EnvLaneFn &F, int lane, double x)
{
    if (F.cur >= 0) return env_seg_regs(F.ga, F.gb, F.va, F.vb, x);
    if (F.evf == -INFINITY) return -INFINITY;
    return env_analytic_ev(e, lane, x, F.evf);
}
// env_fn of function j (uniform), on every lane
template <int L> static __device__ __forceinline__ double env_fn_of(const EnvCtxT<L> &e, const EnvLaneFn &F, int j, double x)
{
    if (EG_RLI(F.cur, j) >= 0) return env_seg_regs(EG_RLD(F.ga, j), EG_RLD(F.gb, j), EG_RLD(F.va, j), EG_RLD(F.vb, j), x);
    const double ev = EG_RLD(F.evf, j);
    if (ev == -INFINITY) return -INFINITY;
    return env_analytic_ev(e, j, x, ev);
}
// env_policy of function j (uniform)
template <int L> static __device__ __forceinline__ double env_policy_of(const EnvCtxT<L> &e, const EnvLaneFn &F, int j, double x)
{
    if (EG_RLI(F.cur, j) >= 0) {
        return env_seg_regs(EG_RLD(F.ga, j), EG_RLD(F.gb, j), EG_RLD(F.ca, j), EG_RLD(F.cb, j), x);
    }
    if (EG_RLD(F.evf, j) == -INFINITY) return EG_ZEROC;
    return x - e.E->a0;
}
// env_wave_pick: over the functions whose bit is clear in `skip`: first != 0: the smallest j with thr < value_j(x);
// first == 0: the j of the largest value_j(x) > thr, smallest index among equals.  Returns j (or -1), its value in *val.
template <int L>
static __device__ __forceinline__ int env_lane_pick(const EnvCtxT<L> &e, const EnvLaneFn &F, int lane, double x, double thr,
                                                    unsigned long long skip, int first, double *val)
{
    double t = 0;
    bool cand = false;
    if (lane < e.nf && !((skip >> lane) & 1ull)) {
        t = env_fn_own(e, F, lane, x);
        cand = thr < t;
    }
    unsigned long long mk = __ballot(cand);
    if (!mk) return -1;
    int bj = __ffsll((long long)mk) - 1;
    double bv = EG_RLD(t, bj);
    if (!first) {
        mk &= mk - 1ull;
        while (mk) {  // ascending j: an equal value keeps the earlier index
            const int j = __ffsll((long long)mk) - 1;
            mk &= mk - 1ull;
            const double tj = EG_RLD(t, j);
            if (tj > bv) bv = tj, bj = j;
        }
    }
    *val = bv;
    return bj;
}

// brsolve (env_bisect) between the analytic value of function fa (value at a0: ev) and a segment held in registers
template <int L>
static __device__ __forceinline__ void env_bisect_regs(EnvCtxT<L> &e, double *b0, double *b1, double ga, double gb, double va, double vb,
                                                       int fa, double ev)
{
    double bq[3] = {*b0, *b1, 0.0}, a[3], d[3];
    int have = 0;
    for (;;) {
        for (int q = have; q < 2; q++) {
            a[q] = env_analytic_ev(e, fa, bq[q], ev);
            d[q] = a[q] - env_seg_regs(ga, gb, va, vb, bq[q]);
        }
        have = 2;
        const double s0 = env_sgn(d[0]), s1 = env_sgn(d[1]);
        if (s0 == s1) {
            e.err = 22;
            return;
        }
        if (bq[0] > bq[1]) {
            e.err = 23;
            return;
        }
        if (fabs(bq[0] - bq[1]) < 2 * EG_DPD || fabs(a[0] - a[1]) < EG_DPD) {
            *b0 = (bq[0] + bq[1]) / 2;
            *b1 = bq[1];
            return;
        }
        bq[2] = (bq[0] + bq[1]) / 2;
        a[2] = env_analytic_ev(e, fa, bq[2], ev);
        d[2] = a[2] - env_seg_regs(ga, gb, va, vb, bq[2]);
        const double sm = env_sgn(d[2]);
        if (s0 == sm)
            bq[0] = bq[2], a[0] = a[2], d[0] = d[2];
        else if (s1 == sm)
            bq[1] = bq[2], a[1] = a[2], d[1] = d[2];
        else {
            *b0 = bq[0];
            *b1 = bq[1];
            return;
        }
    }
}

// thresholds (env_crossing_wave).  The pair being worked on stays in registers; the stack in LDS holds only the pairs a
// split leaves for later (the reference's recursion, :1827-1845: (pri, best) is done before (best, nwi)).
template <int L>
static __device__ __forceinline__ void env_crossing_lanes(EnvCtxT<L> &e, const EnvLaneFn &F, int lane, unsigned long long marked, int pri,
                                                          int nwi, int mode)
{
    const double a0 = e.E->a0;
    int sp = 0;  // pairs on the LDS stack
    while (!e.err) {
        marked |= (1ull << pri) | (1ull << nwi);
        const int cp = EG_RLI(F.cur, pri), cn = EG_RLI(F.cur, nwi);
        double x = 0, top = 0;
        if ((cp == -1) != (cn == -1)) {  // exactly one of the two is still in its analytic region (:1648-1688)
            const int ana = (cp == -1) ? pri : nwi, lin = (cp == -1) ? nwi : pri;
            const double lga = EG_RLD(F.ga, lin), lgb = EG_RLD(F.gb, lin), lva = EG_RLD(F.va, lin), lvb = EG_RLD(F.vb, lin);
            const double ev = EG_RLD(F.evf, ana), am0 = EG_RLD(F.gb, ana);  // (cursor -1: the lane's second point is the function's first)
            if (ev == -INFINITY)
                x = am0;
            else {
                double br0 = lga;
                double br1 = MS_MIN(am0, lgb);
                env_bisect_regs(e, &br0, &br1, lga, lgb, lva, lvb, ana, ev);
                if (e.err) return;
                x = br0;
            }
            top = env_seg_regs(lga, lgb, lva, lvb, x);
        } else if (cp == -1 && cn == -1) {
            e.err = 21;
            return;
        } else {
            const double p0m = EG_RLD(F.ga, pri), p1m = EG_RLD(F.gb, pri), p0v = EG_RLD(F.va, pri), p1v = EG_RLD(F.vb, pri);
            const double n0m = EG_RLD(F.ga, nwi), n1m = EG_RLD(F.gb, nwi), n0v = EG_RLD(F.va, nwi), n1v = EG_RLD(F.vb, nwi);
            const double icn = (n0v * n1m - n1v * n0m) / (n1m - n0m);  // intercepts
            const double icp = (p0v * p1m - p1v * p0m) / (p1m - p0m);
            if (p1m == p0m) {  // previous max is vertical
                x = p0m;
                top = (x * (n1v - n0v) / (n1m - n0m)) + icn;
            } else if (n1m == n0m) {  // entering function is vertical
                x = n0m;
                top = (x * (p1v - p0v) / (p1m - p0m)) + icp;
            } else if (((n1v - n0v) / (n1m - n0m)) == ((p1v - p0v) / (p1m - p0m))) {  // identical slopes
                x = (p0m + p1m + n0m + n1m) / 4;
                top = (x * (n1v - n0v) / (n1m - n0m)) + icn;
            } else {
                x = (icp - icn) / (((n1v - n0v) / (n1m - n0m)) - ((p1v - p0v) / (p1m - p0m)));
                top = (x * (n1v - n0v) / (n1m - n0m)) + icn;
            }
        }
        const int best = env_lane_pick(e, F, lane, x, top, marked, mode == 0, &top);
        if (best != -1) {  // a third function is higher at the crossing: split (:1827-1845)
            if (2 * (sp + 2) > e.stackcap) {
                e.err = 2703;
                return;
            }
            if (mode != 0) {
                if (lane == 0) e.stack[2 * sp] = best, e.stack[2 * sp + 1] = nwi;
                sp++;
            }
            nwi = best;  // (pri, best) next
            continue;
        }
        const double pol0 = env_policy_of(e, F, pri, x), pol1 = env_policy_of(e, F, nwi, x);
        double gx = x;  // grid value of the row written last (kept for the duplicate test)
        if (lane == 0) {
            e.og[e.oi] = x;
            e.ov[e.oi] = top;
            e.oc[e.oi] = (pol0 + pol1) / 2;
            env_log_kink(e, x, pol0, pol1);
            e.oth[e.oj] = x;
            e.oix[e.oj] = nwi;
        }
        e.pm = nwi;
        e.oi += 1;
        e.oj += 1;
        if (e.oi >= e.ocap) {
            e.err = e.e13;
            return;
        }
        if (e.oj >= e.nthrhmax) {
            e.err = 20;
            return;
        }
        if (EG_RLD(F.evf, nwi) == -INFINITY && EG_RLI(F.cur, nwi) == -1) {  // :1892-1900
            gx = x - EG_TOL;
            if (lane == 0) {
                e.oc[e.oi - 1] = pol0;
                e.og[e.oi - 1] = gx;
            }
        } else if (EG_DPD > 0) {  // double point at the kink, :1902-1913
            gx = x + EG_DPD;
            if (lane == 0) {
                e.oc[e.oi - 1] = pol0;
                e.og[e.oi] = gx;
                e.ov[e.oi] = top;
                e.oc[e.oi] = pol1;
            }
            e.oi += 1;
            if (e.oi >= e.ocap) {
                e.err = e.e13;
                return;
            }
        }
        e.lastg = gx;
        if (sp == 0) return;
        EG_WSYNC();  // (lane 0's pushes are in place)
        sp--;
        pri = e.stack[2 * sp], nwi = e.stack[2 * sp + 1];
        EG_WSYNC();  // (read before the slot is pushed again)
    }
}

// OLDOK: env_step_wave may be called for a step whose lane state is not clean (false: such a step is error 2701, which the
// throughput path answers by leaving the cell to k_envelope)
template <int L, bool OLDOK> static __device__ __forceinline__ bool env_step_lanes(EnvCtxT<L> &e, int i)
{
    const int lane = threadIdx.x & (EG_WAVE - 1);
    const double a0 = e.E->a0, bound = e.bound;
    EG_WSYNC();  // cursors written by the previous step / the rebuild are in place
    // ---- the lane's function ----------------------------------------------------------------------------------------
    EnvLaneFn F;
    F.cur = -1, F.dim = 0, F.pa = F.pb = 0, F.ga = F.gb = F.va = F.vb = F.ca = F.cb = 0, F.evf = -INFINITY;
    bool clean = true;
    if (lane < e.nf) {
        F.dim = e.dims[lane];
        if (F.dim > 0) {
            F.cur = e.cur[lane];
            F.evf = env_evf(e, lane);
            const int o = e.fstart[lane] + F.cur;
            clean = F.cur >= -1 && F.cur + 1 < F.dim && o + 1 >= 0 && o + 1 < e.cap && (F.cur < 0 || o >= 0);
            if (clean) {
                F.pb = e.rank[o + 1];
                if (F.cur >= 0) F.pa = e.rank[o];
                clean = F.pb >= 0 && F.pb < e.npts && F.pa >= 0 && F.pa < e.npts;
                if (clean) {
                    F.gb = e.m[F.pb], F.vb = e.v[F.pb], F.cb = e.c[F.pb];
                    if (F.cur >= 0) F.ga = e.m[F.pa], F.va = e.v[F.pa], F.ca = e.c[F.pa];
                }
            }
        }
    }
    if (__ballot(!clean)) {  // (never expected: env_at's guards)
        if (OLDOK) return env_step_wave(e, i);
        e.err = 2701;
        return false;
    }
    const int f = e.f[i];
    const double x = e.m[i];
    if (f < 0 || f >= e.nf || EG_RLI(F.dim, f < 0 || f >= e.nf ? 0 : f) <= 0) {  // sorted stream inconsistent with the per-function lists
        if (e.dbg && atomicCAS(&e.dbg[0], 0, 2708) == 0)
            e.dbg[1] = f, e.dbg[2] = i, e.dbg[3] = e.npts, e.dbg[4] = e.nf, e.dbg[5] = e.sec_id, e.dbg[6] = e.ist;
        e.err = 2708;
        return false;
    }
    const int curf = EG_RLI(F.cur, f);
    if ((e.oi > 0 || e.later) && e.lastg == x) {  // duplicate grid point (:1290-1298)
        if (lane == 0) e.cur[f] = curf + 1;
        return true;
    }
    double fv = EG_RLD(F.vb, f);          // e.v[self], self = env_at(e, f, curf + 1)
    const double cself = EG_RLD(F.cb, f);  // consumption at that point
    const unsigned long long absent = __ballot(lane >= e.nf || F.dim <= 0);  // functions without points
    if (e.oj == 0 && !e.later) {  // first point of the common grid (:1303-1347)
        // the highest function at x, own value included, smallest index among equals
        double t = fv;
        {
            double tj = 0;
            bool cand = false;
            if (!((absent >> lane) & 1ull)) {
                tj = (lane == f) ? fv : env_fn_own(e, F, lane, x);
                cand = !(tj != tj);  // (a NaN never wins; a NaN own value never loses: handled below)
            }
            unsigned long long mk = __ballot(cand);
            int bj = -1;
            double bv = 0;
            while (mk) {
                const int j = __ffsll((long long)mk) - 1;
                mk &= mk - 1ull;
                const double v_ = EG_RLD(tj, j);
                if (bj < 0 || v_ > bv) bv = v_, bj = j;
            }
            if (fv != fv)
                e.ci = f;  // every comparison with NaN is false: nothing replaces the own point
            else
                t = bv, e.ci = bj;
        }
        if (lane == 0) {
            e.oth[e.oj] = a0;
            e.oix[e.oj] = e.ci;
        }
        e.pm = e.ci;
        e.oj++;
        if (e.oj >= e.nthrhmax) {
            e.err = 20;
            return false;
        }
        if (e.ci == f) {
            env_push_wave(e, x, t, cself, lane);
            if (e.oi >= e.ocap) {
                e.err = e.e13;
                return false;
            }
        }
    } else {
        int xa = -1, xb = -1, xmode = 0, post = 0;  // post: 0 nothing, 1 last row of ci, 2 push own point, 3 last row of cj
        int cj = -1;
        if (e.pm == f) {  // point of the current max function
            double t = 0;
            const int j = env_lane_pick(e, F, lane, x, fv, absent | (1ull << f), x != bound, &t);
            if (j < 0) {
                env_push_wave(e, x, fv, cself, lane);
                if (e.oi == e.ocap) {
                    e.err = e.e13;
                    return false;
                }
            } else if (x != bound) {
                xa = f, xb = j, xmode = 0, post = 0;
            } else {
                fv = t;
                e.ci = j;
                xa = f, xb = e.ci, xmode = 1, post = 1;
            }
        } else {  // point of another function
            e.ci = e.pm;
            const double t = env_fn_of(e, F, e.ci, x);
            if (t < fv) {
                double tv = 0;
                cj = env_lane_pick(e, F, lane, x, fv, absent | (1ull << f) | (1ull << e.ci), 0, &tv);
                if (cj == -1)
                    xa = e.ci, xb = f, xmode = 1, post = 2;
                else {
                    fv = tv;
                    xa = e.ci, xb = cj, xmode = 1, post = (x == bound) ? 3 : 0;
                }
            } else if (x == bound) {
                const double vv = env_fn_of(e, F, e.ci, x), pp = env_policy_of(e, F, e.ci, x);
                env_push_wave(e, x, vv, pp, lane);
            }
        }
        if (xa >= 0) {
            env_crossing_lanes(e, F, lane, absent, xa, xb, xmode);
            if (e.err) return false;
            if (post == 1) {
                e.lastg = x;
                const double vv = env_fn_of(e, F, e.ci, x);
                // (:1406-1408; the reference indexes evfa0 with the exhausted loop variable there)
                double pp = x - a0;
                if (EG_RLI(F.cur, e.ci) >= 0)
                    pp = env_seg_regs(EG_RLD(F.ga, e.ci), EG_RLD(F.gb, e.ci), EG_RLD(F.ca, e.ci), EG_RLD(F.cb, e.ci), x);
                if (lane == 0) {
                    e.og[e.oi] = x;
                    e.ov[e.oi] = vv;
                    e.oc[e.oi] = pp;
                }
                e.oi++;
                if (e.oi >= e.ocap) {
                    e.err = e.e13;
                    return false;
                }
            } else if (post == 2) {
                env_push_wave(e, x, fv, cself, lane);
                if (e.oi >= e.ocap) {
                    e.err = e.e13;
                    return false;
                }
            } else if (post == 3) {
                const double vv = env_fn_of(e, F, cj, x), pp = env_policy_of(e, F, cj, x);
                env_push_wave(e, x, vv, pp, lane);
            }
        }
    }
    const int dimf = EG_RLI(F.dim, f);  // (a collective: outside the branch)
    if (lane == 0) e.cur[f] = MS_MIN(curf + 1, dimf - 2);
    return true;
}

#ifdef EGDST_SEQ_WALK
// Plain sequential walk (diagnostic build only).
template <int L> static __device__ __forceinline__ void env_walk(EnvCtxT<L> &e, int npts)
{
    env_begin(e);
    for (int i = 0; i < npts && e.m[i] <= e.bound && !e.err; i++)
        if (!env_step(e, i)) return;
}
#endif

// ---- wave-cooperative walk ------------------------------------------------------------------------------
// The walk only does something irregular at regime changes (a crossing, the first and the last grid value);
// everywhere else a sorted point is either KEPT (it belongs to the current max function and nothing is above
// it), SKIPPED (another function, below the current max) or a DUPLICATE of the last kept grid value.  These
// three outcomes depend on the walk's state only through (a) the current max function and (b) how many points
// of each function precede the position -- and for grid values strictly below `bound` the latter is a pure
// count (cur[j] = count_j - 1: no function has run out of points yet, so the clamp of :1517 is inactive).  So
// the 64 lanes classify 64 consecutive positions at once, the leading run of regular positions is committed
// with ballot/popcount compaction in order, and the first irregular position is handed to env_step() with
// cur[] rebuilt from the counts.  Every lane executes env_step redundantly (uniform control flow, identical
// stores).  Output is identical to env_walk(); the CPU harness checks that bit for bit.
template <int L> static __device__ __forceinline__ int env_count_before(const EnvCtxT<L> &e, int j, int p)
{
    // number of points of function j at sorted positions < p (the position list of j is ascending)
    const typename EgMem<L>::S *lst = e.rank + e.fstart[j];
    int lo = 0, hi = e.dims[j];
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (lst[mid] < p)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

template <int L> static __device__ __forceinline__ double env_fn_at(EnvCtxT<L> &e, int j, int curj, double x)
{
    if (curj >= 0) return env_seg(e, j, curj, x, 0);
    if (env_evf(e, j) == -INFINITY) return -INFINITY;
    return env_analytic(e, j, x);
}

// value of function j at x given that cj of its points precede the position (wave walk; no bounds guards: below the
// bound every function still has a point ahead, so cj <= dims[j]-1)
template <int L> static __device__ __forceinline__ double env_fn_cnt(EnvCtxT<L> &e, int j, int cj, double x)
{
    if (cj >= 1) {
        const int base = e.fstart[j] + cj - 1;
        const int a = e.rank[base], b = e.rank[base + 1];
        const double ga = e.m[a], gb = e.m[b], fa = e.v[a], fb = e.v[b];
        if (x == ga) return fa;
        if (x < ga) return -INFINITY;
        if (x > gb) return -INFINITY;
        return fb * (x - ga) / (gb - ga) + fa * (gb - x) / (gb - ga);
    }
    if (env_evf(e, j) == -INFINITY) return -INFINITY;
    return env_analytic(e, j, x);
}

// Pre-classification of every sorted position by ALL threads of the workgroup (the floating-point part of the
// walk).  For a grid value below the bound the walk's decision depends on its state only through the current
// max function pm, so everything else is computed here, in parallel and state-free:
//   bit 0      some other function is above this point (f == pm: the point is irregular, else kept)
//   bit 1+j    function j is below this point          (f != pm: irregular iff bit 1+pm, else skipped)
//   bit 30     no per-function bits (more than 29 functions): the scan evaluates f_pm(x) itself for f != pm
//   sign bit   always irregular (at/after the bound, inconsistent stream)
#define ENV_CLS_FORCE ((int)0x80000000)
#define ENV_CLS_NOMASK (1 << 30)
template <int L>
static __device__ __forceinline__ void env_preclass(EnvCtxT<L> &e, int npts, typename EgMem<L>::I *cls, int tid, int nthreads)
{
    double bound = INFINITY;
    for (int f = 0; f < e.nf; f++)
        if (e.dims[f] > 0) {
            const double last = e.m[e.rank[e.fstart[f] + e.dims[f] - 1]];
            if (last < bound) bound = last;
        }
    for (int p = tid; p < npts; p += nthreads) {
        const int f = e.f[p];
        const double x = e.m[p];
        int w = 0;
        if (!(x < bound) || f < 0 || f >= e.nf || e.dims[f] <= 0)
            w = ENV_CLS_FORCE;
        else {
            const double fv = e.v[p];
            for (int j = 0; j < e.nf; j++) {
                if (e.dims[j] <= 0 || j == f) continue;
                const int cj = env_count_before(e, j, p);
                if (cj >= e.dims[j]) {  // cannot happen below the bound
                    w = ENV_CLS_FORCE;
                    break;
                }
                const double t = env_fn_cnt(e, j, cj, x);
                if (fv < t) w |= 1;
                if (t < fv && e.nf <= 29) w |= (2 << j);
            }
            if (e.nf > 29 && w >= 0) w |= ENV_CLS_NOMASK;
        }
        cls[p] = w;
    }
}

// Walks the sorted positions [p0, p1).  later == 0: from the start of the stream (p0 == 0) with the reference's initial
// state.  later == 1: a later segment (run_walk): starts in the regular phase with the current max function pm0 and the
// last output grid value lastg0 that the preceding segment is PREDICTED to end with; rows and thresholds are counted
// from 0 in this segment's own output region.  On return e.pm / e.lastg hold the state after position p1-1.
// OLDSTEP: env_step_wave is compiled in (not for the walks of the throughput path, L == 2, whose callers keep to a wave's worth
// of functions: the kernel is 167 VGPRs of mostly dead code otherwise)
#if defined(EGDST_EMU) || !ENV_LANE_STEP
#define ENV_OLDSTEP(L) true
#else
#define ENV_OLDSTEP(L) ((L) != 2)
#endif
template <int L, bool OLDSTEP = ENV_OLDSTEP(L)>
static __device__ __forceinline__ void env_walk_wave(EnvCtxT<L> &e, int npts, int p0 = 0, int p1 = -1, int later = 0,
                                                                       int pm0 = -1, double lastg0 = 0)
{
    const int lane = threadIdx.x & (EG_WAVE - 1);
    if (p1 < 0) p1 = npts;
    e.later = later;
#ifdef EGDST_SEQ_WALK  // diagnostic build: the plain sequential walk on lane 0
    if (lane == 0) env_walk(e, npts);
    __threadfence_block();
    e.oi = __shfl(e.oi, 0);
    e.oj = __shfl(e.oj, 0);
    e.err = __shfl(e.err, 0);
    return;
#else
    {   // env_begin: every lane needs `bound`; the cursor array is shared, lane 0 initialises it
        double bound = INFINITY;
        for (int f = 0; f < e.nf; f++)
            if (e.dims[f] > 0) {
                double last = e.m[env_at(e, f, e.dims[f] - 1)];
                if (last < bound) bound = last;
            }
        e.bound = bound;
        e.oi = e.oj = 0;
        e.ci = 0;
        e.lastg = later ? lastg0 : -INFINITY;
        e.pm = later ? pm0 : -1;
        if (lane == 0)
            for (int f = 0; f < e.nf; f++) e.cur[f] = -1;
        e.err = __shfl(e.err, 0);
    }
    // phase 0: the first point(s) until a current-max function exists; phase 1: batches of EG_WAVE positions;
    // phase 2: the grid values at the bound (last point of the shortest function).  One generic-step site.
    int i = p0, pm = e.pm, phase = later ? 1 : 0;
    double lastg = e.lastg;
    // L == 2, the consumption column in global memory: a batch only copies it to the rows it keeps -- but the wave cannot wait
    // for that load without waiting for the row stores of the batch before as well (vector memory operations retire in
    // order, the compiler waits for all of them at the top of the loop), a round trip of about a microsecond per batch of 64
    // positions and most of the walk's time (profiles/r03_*).  So the batches do not touch the column at all: a consumed
    // position's class word is replaced by the row it went to (-1: none), and when the walk is over all threads of the
    // workgroup copy the column to the rows in one parallel pass (run_walk).  A walk that has to be done again (a failed
    // prediction of the segmented walk) classifies the stream again first.
    constexpr bool CDEFER = ENV_CDEFER(L);
    typename EgMem<L>::I *const clsw = const_cast<typename EgMem<L>::I *>(e.cls);
#ifdef EGDST_STAMPS
    unsigned long long w_t0 = wall_clock64(), w_step = 0, w_batch = 0, w_nstep = 0, w_nbatch = 0;
#define WSTAMP(acc, cnt) do { unsigned long long n_ = wall_clock64(); acc += n_ - w_t0; w_t0 = n_; cnt++; } while (0)
#else
#define WSTAMP(acc, cnt)
#endif
    bool chain = false;  // the position after a generic step is irregular too (decided below): step again, the cursors are in place
    while (i < p1 && !e.err) {
        bool step_now = true;
        if (chain) {
            chain = false;
        } else
        if (phase != 1) {
            if (!(e.m[i] <= e.bound)) break;
            // A run of positions that repeat the grid value of the last output row (:1290-1298).  The generic step does
            // nothing for such a position but advance its function's cursor by one -- no row, no threshold, the grid value
            // stays -- so the next position with the same value is in the same case, and so on to the end of the run: the run
            // is consumed here, 64 positions at a time, with the same cursor increments.  (A guess stream that ended in the
            // generator's resend fixed point -- k_probe / k_fixup fast-forward it -- brings ~9000 copies of its last point, and
            // that point is the bound of the secondary envelope: one generic step per copy was 6 ms for such a cell.)
            if (ENV_DUPRUN && phase == 2 && (e.oi > 0 || e.later) && lastg == e.m[i]) {
                const double x0 = lastg;
                for (;;) {
                    const int p = i + lane;
                    int fp = -1;
                    bool rep = false;
                    if (p < p1) {
                        fp = e.f[p];
                        rep = e.m[p] == x0 && fp >= 0 && fp < e.nf;
                        if (rep) rep = e.dims[fp] > 0;  // (an inconsistent stream is the generic step's to report)
                    }
                    const unsigned long long notrep = ~__ballot(rep) & (EG_WAVE >= 64 ? ~0ull : ((1ull << (EG_WAVE & 63)) - 1ull));
                    const int run = notrep ? (__ffsll((long long)notrep) - 1) : EG_WAVE;  // leading positions of the chunk in the run
                    const bool mine = lane < run;
                    unsigned long long todo = __ballot(mine);
                    EG_WSYNC();
                    while (todo) {  // one turn per function with points in the chunk (normally one)
                        const int g = __shfl(fp, __ffsll((long long)todo) - 1);
                        const unsigned long long same = __ballot(mine && fp == g);
                        if (lane == 0) e.cur[g] = e.cur[g] + __popcll(same);
                        todo &= ~same;
                    }
                    EG_WSYNC();
                    if (CDEFER && mine) clsw[p] = -1;  // consumed, no row
                    i += run;
                    if (run < EG_WAVE || !(i < p1)) break;
                }
                continue;  // (the loop's own tests decide about position i: the end of the range, the bound)
            }
        } else {
            // The regular batches are a loop of their own, entered with no vector memory operation pending (EG_VM_DRAIN):
            // inside it the wave only issues row stores, which nothing waits for.
            bool tail = false;
            EG_VM_DRAIN();
            for (;;) {
            const int p = i + lane;
            const bool valid = p < p1 && e.m[p] < e.bound;
            const unsigned long long vmask = __ballot(valid);
            if (!(vmask & 1ull)) {  // position i is at (or beyond) the bound: sequential tail with rebuilt cursors
                tail = true;
                break;
            }
            int cls = 1, f = -1;  // 0 keep, 1 skip, 2 event
            double x = 0, fv = 0, cc = 0;
            const unsigned long long below = (1ull << lane) - 1ull;  // lanes < lane
            if (valid) {
                f = e.f[p];
                x = e.m[p];
                fv = e.v[p];
                if (!CDEFER) cc = e.c[p];
                const int w = e.cls[p];
                if (w < 0)
                    cls = 2;
                else if (f == pm)
                    cls = (w & 1) ? 2 : 0;
                else if (w & ENV_CLS_NOMASK) {
                    const int cj = env_count_before(e, pm, p);
                    cls = (cj >= e.dims[pm]) ? 2 : ((env_fn_cnt(e, pm, cj, x) < fv) ? 2 : 1);
                } else
                    cls = ((w >> (pm + 1)) & 1) ? 2 : 1;
            }
            // duplicates (:1290-1298): equal to the last kept grid value, carried in or kept earlier in this batch
            const double xprev = __shfl_up(x, 1);
            const bool newrun = valid && (lane == 0 || x != xprev);
            const unsigned long long rmask = __ballot(newrun), kmask = __ballot(valid && cls == 0);
            bool dup = false;
            if (valid) {
                const unsigned long long upto = below | (1ull << lane);
                const int s = 63 - __clzll((long long)(rmask & upto));  // first lane of this run of equal grid values
                const unsigned long long inrun = below & ~((1ull << s) - 1ull);
                dup = (lastg == x) || ((kmask & inrun) != 0ull);
            }
            const unsigned long long emask = __ballot(valid && !dup && cls == 2);
            const int nvalid = __popcll(vmask);
            const int stop = emask ? (__ffsll((long long)emask) - 1) : nvalid;  // lanes [0, stop) are regular
            const bool out = valid && lane < stop && !dup && cls == 0;
            const unsigned long long omask = __ballot(out);
            const int nout = __popcll(omask);
            if (nout) {
                if (e.oi + nout >= e.ocap) {  // the push that fills the grid is an error in the reference (:1378)
                    e.err = e.e13;
                    return;
                }
                if (out) {
                    const int d = e.oi + __popcll(omask & ((1ull << lane) - 1ull));
                    e.og[d] = x;
                    e.ov[d] = fv;
                    if (CDEFER)
                        clsw[p] = d;
                    else
                        e.oc[d] = cc;
                }
                e.oi += nout;
                lastg = __shfl(x, 63 - __clzll((long long)omask));  // grid value of the last point kept in this batch
            }
            if (CDEFER && valid && lane < stop && !out) clsw[p] = -1;  // consumed, no row
            i += stop;
            step_now = stop < nvalid;
            WSTAMP(w_batch, w_nbatch);
            if (step_now || !(i < p1)) break;
            }
            if (tail || step_now) {  // the sequential tail / an irregular position: the generic step needs the per-function cursors, a lane each
                for (int j = lane; j < e.nf; j += EG_WAVE) e.cur[j] = (e.dims[j] > 0) ? env_count_before(e, j, i) - 1 : -1;
                (void)__ballot(1);  // wave-wide: every cursor is written before lane 0 steps
            }
            if (tail) {
                phase = 2;
                continue;
            }
        }
        if (step_now) {
            e.lastg = lastg;  // rows committed by the batches since the last generic step
            bool ok_;
            if (ENV_LANES(L) && e.nf <= EG_WAVE)
                ok_ = env_step_lanes<L, OLDSTEP>(e, i);
            else if (OLDSTEP)
                ok_ = env_step_wave(e, i);
            else {  // (the caller promised at most a wave's worth of functions)
                e.err = 2701;
                ok_ = false;
            }
            lastg = e.lastg;
            pm = e.pm;
            if (!ok_) return;
            if (CDEFER && lane == 0) clsw[i] = -1;  // (whatever row the step kept, it wrote whole)
            i++;
            if (phase == 0 && e.oj > 0) phase = 1;
            WSTAMP(w_step, w_nstep);
            // Events come in clusters -- where a choice list folds, and along the whole stream when the pieces of a re-based guess
            // stream lie within ulps of each other (hundreds of crossings in a row: the slowest walk of nearly every launch,
            // profiles/r04_*).  Going back to the batches after every step costs a classification of 64 positions that commits
            // none and a rebuild of the cursors from the position lists (a bisection per function) that the step has just left in
            // place: a generic step keeps cur[] exactly as the reference's loop does (below the bound: count - 1, what the rebuild
            // gives).  So: if the next position is below the bound and irregular for the new current-max function -- the batch's
            // own test, for one position -- step again at once.
            if (ENV_STEP_CHAIN && phase == 1 && i < p1) {
                const double xn = e.m[i];
                if (xn < e.bound && xn != lastg) {
                    const int wn_ = e.cls[i], fn_ = (int)e.f[i];
                    bool ev_;
                    if (wn_ < 0)
                        ev_ = true;
                    else if (fn_ == pm)
                        ev_ = (wn_ & 1) != 0;
                    else if (wn_ & ENV_CLS_NOMASK)
                        ev_ = false;  // (needs a count: the batches decide)
                    else
                        ev_ = ((wn_ >> (pm + 1)) & 1) != 0;
                    chain = ev_;
                }
            }
        }
    }
    e.pm = pm;
    e.lastg = lastg;
    e.iend = i;
#if defined(EGDST_STAMPS) && !defined(EGDST_STAMPS2) && !defined(EGDST_STAMPS3)
    if (lane == 0 && e.dbg) {
        atomicAdd((unsigned long long *)e.dbg + 3, (w_nbatch << 32) | w_nstep);
        atomicAdd((unsigned long long *)e.dbg + 4, w_step);
        atomicAdd((unsigned long long *)e.dbg + 7, w_batch);
    }
#endif
#endif
}
