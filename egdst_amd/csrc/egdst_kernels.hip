// egdst_kernels.hip -- backward-induction DC-EGM solver and forward simulator for MI355X (gfx950).
//
// One backward period = four launches per draw group, each group on its own stream:
//   k_probe     one wave per (draw, state, choice): the sequential head of the guess generator
//               (adraw stage 0/1 and the zero-consumption resend, egdst_solver.c:955-1099) with
//               each full expectation evaluated cooperatively by the 64 lanes (one lane per
//               shock node) and accumulated in the reference's order;
//   k_grid      one lane per remaining end-of-period asset point: closed-form log grid
//               (egdst_solver.c:1110-1136) + the serial (next state, shock) loop of egmbellman
//               (egdst_solver.c:494-574) + Euler inversion (:628-650);
//   k_fixup     one workgroup per (draw, state, choice), idle unless a zero-consumption resend turned up inside the
//               grid stage (:1080-1099): then the stream is regenerated in the reference's order;
//   k_envelope  one workgroup per (draw, state): stop rule (:1100,1150), compaction, secondary
//               envelope (:776-913), rank-merge sort + primary envelope (:1165-1550), saveoutput
//               layout (:917-952);
// the terminal period uses k_terminal (:433-476) + k_envelope.  k_simulate runs one lane per
// simulated agent (egdst_simulator.c:204-383).  Parameter draws are independent: every kernel
// is batched over `draw`, which is also the unit sharded across GPUs (no data-path collective).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#ifndef WAVE  // the sanitizer harness (tests/cpu_emu) overrides these
#define WAVE 64
#ifndef GRID_BS
#define GRID_BS 256
#endif
#define ENV_MAXBS 512  // most threads of a k_envelope workgroup (what it is compiled for: <= 256 VGPRs)
#else
#define ENV_MAXBS ENV_BS_EMU
#endif
// k_envelope runs with 256 threads per workgroup in big batches and with 512 when the batch leaves CUs idle (its
// sort, compaction and classification phases scale with the threads, the walk is one wave either way): measured on
// C2, a single solve 18 -> 16 ms with 512 threads, a 4096-draw batch 332 -> 350 ms.  Device code reads the size.
#define ENV_BS ((int)blockDim.x)
#define EG_WAVE WAVE
#ifndef EG_FIX_LROWS
#define EG_FIX_LROWS 2048        // rows of the next-period table that k_fixup stages in LDS (48 KB)
#endif
#ifndef FIX_LPG_SHIFT  // k_fixup: log2 of the lanes that share a guess in the small batch after a resend (0: no small batches)
#ifdef EGDST_EMU
#define FIX_LPG_SHIFT 0  // (the harness cannot diverge inside a wave)
#else
#define FIX_LPG_SHIFT 4
#endif
#endif
#ifndef EG_SEQ_AFTER_RESEND  // k_fixup: guesses evaluated one at a time after a c1<=0 resend, before the next batch (with small batches: none
#define EG_SEQ_AFTER_RESEND (FIX_LPG_SHIFT > 0 ? 0 : 1)  // -- C2 a0=-5 x 4096: 172.6 ms with one, 170.2 without; without small batches 0 or 1 were equal)
#endif
#ifndef FIX_BS  // threads of a k_fixup workgroup = guesses evaluated per batch of the sequential stream
#define FIX_BS (4 * WAVE)  // one wave per SIMD: the sequential stretches run redundantly in every wave
#endif

#ifdef EGDST_EMU  // the CPU sanitizer harness has no dynamic LDS: a static buffer stands in for it
#define EG_DYN_LDS(name) static double name[20480 + 64]
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/asan_interface.h>
#define EG_EMU_POISON(p, n) __asan_poison_memory_region((const void *)(p), (n))
#define EG_EMU_UNPOISON_DYN(p) __asan_unpoison_memory_region((const void *)(p), sizeof(double) * (20480 + 64))  // (a phase of a fused kernel lays the LDS out anew)
#else
#define EG_EMU_POISON(p, n) ((void)0)
#define EG_EMU_UNPOISON_DYN(p) ((void)0)
#endif
#else
#define EG_DYN_LDS(name) extern __shared__ double name[]
#endif
#include "egdst_device.h"
#include "egdst_envelope.h"
#include "../../include/egdst.h"

// ---------------------------------------------------------------------------------------------
static __device__ __forceinline__ void eg_fail(BatchRef b, int draw, int it, int ist, int code)
{
    if (atomicCAS(&b.status[draw], 0, code) == 0) {
        b.where[2 * draw] = it;
        b.where[2 * draw + 1] = ist;
    }
}

static __device__ __forceinline__ ms_env eg_env(BatchRef b, int draw)
{
    ms_env E;
    E.t0 = b.g.t0;
    E.T = b.g.T;
    E.ngridm = b.g.ngridm;
    E.ngridmax = b.g.ngridmax;
    E.nthrhmax = b.g.nthrhmax;
    E.ny = b.g.ny;
    E.mmax = b.g.mmax;
    E.a0 = b.g.a0;
    E.par = b.par + (size_t)draw * MS_NPARAM;
    return E;
}

static __device__ __forceinline__ size_t eg_cand(BatchRef b, int draw, int ist, int id)
{
    return (((size_t)draw * MS_NST + ist) * MS_ND + id) * (size_t)b.g.Cp;
}

// A candidate of the grid stage is three doubles and one word: cM holds M -- or, when the point's status is not 0 (cM would be
// NaN and is never kept), the value the evaluation REPORTS BACK to the guess generator (egdst_solver.c:595-599,632: a0-1 for
// c1<=0, the cash at the break otherwise), which for a normal point is M itself; so the stop rule (:1100) reads cM whatever the
// status.  cSt packs the status (low 16 bits, signed: 0 normal, 1 c1<=0, 2 evf=-inf, < 0 a hard error code) and the evaluations
// done for the point (high 16 bits).  Until round 3 these were four arrays (cM, cR, cSt, cCnt: 24 B per point, 12 of them
// redundant); round 4 writes and reads 28 B per candidate instead of 40.
static __device__ __forceinline__ int eg_sc_pack(int status, int cnt) { return (int)(((unsigned)cnt << 16) | ((unsigned)status & 0xffffu)); }
static __device__ __forceinline__ int eg_sc_status(int w) { return (int)(short)(w & 0xffff); }
static __device__ __forceinline__ int eg_sc_count(int w) { return (int)((unsigned)w >> 16); }

#ifdef EGDST_CENSUS
// Diagnostic build (tests/diag/gpu_census.py): a log of what the slow paths of a batch were given to do -- 8 ints per record:
// kind, draw, it, a, b, c, d, ticks of 10 ns.  kinds: 1 a sort + walk job of k_envelope (a job, b points, c functions | path << 16,
// d why the throughput path left the cell), 2 a cell of k_envelope (a rows out, b thresholds, c error, d why), 3 a stream of
// k_fixup (a choice, b points kept, c calls), 4 a k_probe wave that took long (a choice, b calls, c fast-forwarded calls),
// 5 a walk of the throughput path that gave up (a stage, b error, c points, d rows), 6 every walk of the throughput path (a stage |
// second tier << 4, b points, c functions, d workgroup).
#define EG_CENSUS_CAP (1 << 20)
__shared__ int cz_sh_bad;  // the sort of the current job found a list out of comp1 order (1: network, 2: counted)
__shared__ int cz_sh_fb;   // the current walk: 0 one wave, 1 cut into segments and merged, 2 cut and fallen back to one wave, | generic steps << 8
__device__ int g_census[EG_CENSUS_CAP * 8];
__device__ unsigned g_census_n;
static __device__ __forceinline__ void eg_census(int kind, int draw, int it, int a, int b_, int c, int d, unsigned long long ticks)
{
    const unsigned k = atomicAdd(&g_census_n, 1u);
    if (k >= EG_CENSUS_CAP) return;
    int *o = g_census + 8 * (size_t)k;
    o[0] = kind, o[1] = draw, o[2] = it, o[3] = a, o[4] = b_, o[5] = c, o[6] = d, o[7] = (int)(ticks > 0x7fffffffull ? 0x7fffffffull : ticks);
}
#define EG_CENSUS(...) eg_census(__VA_ARGS__)
#define EG_CENSUS_BAD(v) do { if (threadIdx.x == 0) cz_sh_bad = (v); } while (0)
#else
#define EG_CENSUS_BAD(v) ((void)0)
#define EG_CENSUS(...) ((void)0)
#endif

// ---------------------------------------------------------------------------------------------
// terminal period (egdst_solver.c:433-476, END2): M_i = trinv(m1 + i (m2-m1)/(ngridm-1)), C = M, V = u(C)
__global__ void __launch_bounds__(GRID_BS) k_terminal(const Batch *bp_, int it)
{
    BatchRef b = EG_BATCH_REF(bp_);
    const int combo = blockIdx.y;
    const int id = combo % MS_ND, ist = (combo / MS_ND) % MS_NST, draw = b.order[b.draw0 + combo / (MS_ND * MS_NST)];
    const int i = blockIdx.x * GRID_BS + threadIdx.x;
    if (b.status[draw]) return;
    ms_env E = eg_env(b, draw);
    ms_pv cur;
    cur.it = it;
    cur.ist = ist;
    cur.id = id;
    cur.cash = cur.savings = cur.shock = 0;
    ProbeOut *P = &b.probe[((size_t)draw * MS_NST + ist) * MS_ND + id];
    const int act = ms_feasible(&E, &cur) == 1 && ms_inchoiceset(&E, &cur) == 1;
    if (i == 0) {
        P->active = act;
        P->seq = 0;
        P->np = 0;
        P->grid = 0;
        P->probe_evals = 0;
        P->evfa0 = -INFINITY;
    }
    if (!act || i >= b.g.ngridm) return;
    const double m1 = ms_tr(&E, &cur, EG_ZEROC - EG_A0T), m2 = ms_tr(&E, &cur, b.g.mmax - EG_A0T);
    const double m = ms_trinv(&E, &cur, m1 + i * (m2 - m1) / (b.g.ngridm - 1)) + EG_A0T;
    const double c = m - EG_A0T;
    const size_t o = eg_cand(b, draw, ist, id) + i;
    b.cM[o] = m;
    b.cC[o] = c;
    b.cV[o] = ms_utility(&E, &cur, c);
}

// ---------------------------------------------------------------------------------------------
// Closed-form log grid of end-of-period assets (egdst_solver.c:1110-1136) and the per-lane EGM evaluation of one
// point (the serial (next state, shock) loop of egmbellman, :494-574, and the Euler inversion, :628-650).
struct GridLims {
    double lim1, lim2, lim3, lim3p, k3;
    int ntogenerate;
};

static __device__ __forceinline__ double eg_grid_target(const ms_env *E, const ms_pv *cur, const GridLims &P, int n)
{
    // X_n of :1110-1119 (n is the value of `ngenerated` at the call)
    if (n < (int)P.k3 - 1)
        return -ms_trinv(E, cur, P.lim3 + (P.k3 - 1 - n) * (P.lim1 - P.lim3) / (P.k3 - 1)) + P.lim3p;
    return ms_trinv(E, cur, P.lim3 + (n - P.k3 + 1) * (P.lim2 - P.lim3) / (P.ntogenerate - P.k3)) + P.lim3p;
}

// A_n for n > nb, where A_nb = Ab is known exactly.  The reference advances by steps A_n = A_{n-1} + (X_n - A_{n-1})
// with a floor on negative steps (:1120-1136).  Whenever X_n and A_{n-1} are within a factor two of each other the
// step is exact and A_n == X_n, so the chain only needs to be followed back to the last such point.
static __device__ __forceinline__ double eg_grid_A(const ms_env *E, const ms_pv *cur, const GridLims &P, int nb, double Ab, int n)
{
    int k = n;
    while (k > nb + 1) {  // find a start whose predecessor makes the step exact
        const double xk = eg_grid_target(E, cur, P, k - 1), xk1 = eg_grid_target(E, cur, P, k);
        const bool same = (xk > 0 && xk1 > 0) || (xk < 0 && xk1 < 0);
        if ((same && fabs(xk1) <= 2 * fabs(xk) && fabs(xk) <= 2 * fabs(xk1)) || n - k >= 8) break;  // A_k == X_k
        k--;
    }
    double prev = (k == nb + 1) ? Ab : eg_grid_target(E, cur, P, k - 1);
    for (int j = k; j <= n; j++) {
        double step = eg_grid_target(E, cur, P, j) - prev;
        if (step < 0) step = MS_MAX(step, 1e-5);
        prev += step;
    }
    return prev;
}

struct LaneEval {
    int status, cnt, bist;  // 0 normal, 1 c1<=0, 2 evf=-inf, <0 hard error; evaluations; next state at the break
    double M, C, V, R;      // point (M NaN unless normal) and the M reported back to the generator (:595-599,632)
    double bshock, bcash;
};

// lt: nullptr, or (models with one state) the next-period table staged in LDS by the caller -- k_fixup's speculative batches:
// their bracket searches were 2 x 11 dependent L2 round trips per term and made a regeneration ~0.8 ms (profiles/r03_*)
template <class TT = Tab>
static __device__ __forceinline__ LaneEval eg_lane_eval(BatchRef b, const ms_env *E, const ms_pv *cur, int slot1, int draw,
                                                        double A, const TT *lt = nullptr, int lt_sorted = 0)
{
    LaneEval r;
    const int ny = b.g.ny;
    double rhs = 0, evf = 0, checksum = 0, c1 = 1.0;
    int status = 0, terr = 0, cnt = 0;
    r.bist = 0;
    r.bshock = r.bcash = 0;
    ms_pv nxt;
    nxt.it = cur->it + 1;
    nxt.id = 0;
    nxt.cash = 0;
    nxt.shock = 0;
    nxt.savings = A;
    for (nxt.ist = 0; nxt.ist < MS_NST; nxt.ist++) {
        if (ms_feasible(E, &nxt) != 1) continue;
        double pr1pre = 0;
        if (MS_OPTIM_TRPRNOSH) {
            pr1pre = ms_trpr(E, cur, &nxt, &terr);
            if (pr1pre == 0.0) continue;
        }
        const int niy = (ms_sigma(E, cur, &nxt) <= 0 || ny == 1) ? 1 : ny;
        TT t;
        if (lt)
            t = *lt;
        else {
            const Tab tg = eg_tab(b, slot1, draw, nxt.ist);
            t.M = (decltype(t.M))tg.M, t.C = (decltype(t.C))tg.C, t.V = (decltype(t.V))tg.V;  // (TT == Tab here)
            t.TH = tg.TH, t.D = tg.D, t.len = tg.len, t.thlen = tg.thlen;
        }
        if (t.len < 2) {
            status = -10;
            break;
        }
        if (t.len > b.g.Sp || t.thlen > b.g.nthrhmax || t.thlen < 1) {
            status = -2707;
            break;
        }
        // the M column is known to be in order: k_sortcheck ran this period, or the caller checked its staged copy
        const int tsorted = lt ? lt_sorted : eg_tab_sorted(b, cur->it, draw, nxt.ist);
        for (int iy = 0; iy < niy; iy++) {
            double pr1;
            if (niy == 1) {
                nxt.shock = eg_shock_mean(E, cur, &nxt);
                pr1 = MS_OPTIM_TRPRNOSH ? pr1pre : ms_trpr(E, cur, &nxt, &terr);
            } else {
                nxt.shock = eg_shock_node(E, cur, &nxt, b.qz[iy]);
                pr1 = MS_OPTIM_TRPRNOSH ? pr1pre : ms_trpr(E, cur, &nxt, &terr);
                pr1 *= b.qw[iy];
            }
            if (pr1 == 0.0) continue;
            checksum += pr1;
            cnt++;
            double t_rhs, t_evf;
            int verr = 0;
            c1 = eg_term(E, t, cur, &nxt, pr1, 1, &t_rhs, &t_evf, tsorted, &verr);
            if (c1 <= 0) break;
            rhs += t_rhs;
            if (verr) {  // valuefunc on a table with one row beside the a0 row (egdst_lib.c:183, egdst_solver.c:567-568)
                status = -10;
                break;
            }
            evf += t_evf;
            if (evf == -INFINITY) break;
        }
        if (status) break;
        if (c1 <= 0 || evf == -INFINITY) {
            r.bist = nxt.ist;
            r.bshock = nxt.shock;
            r.bcash = nxt.cash;
            break;
        }
    }
    if (terr) status = -25;
    if (status == 0) {
        if (c1 <= 0)
            status = 1;
        else if (evf == -INFINITY)
            status = 2;
        else if (fabs(checksum - 1) > EG_TOL)
            status = -11;
    }
    r.status = status;
    r.cnt = cnt;
    if (status == 0) {
        rhs *= ms_discount(E, cur);
        r.M = A + ms_utility_marginal_inverse(E, cur, rhs);
        r.C = r.M - A;
        r.V = ms_utility(E, cur, r.C) + ms_discount(E, cur) * evf;
        r.R = r.M;
    } else {
        r.M = NAN;
        r.C = r.V = 0;
        r.R = (status == 1) ? b.g.a0 - 1 : r.bcash;
    }
    return r;
}

// ---------------------------------------------------------------------------------------------
// Full expectation at one savings guess, evaluated by one wave: lane l handles shock node l of the
// current next-state; the weighted terms are then accumulated in (ist1 asc, iy asc) order by every
// lane redundantly, so that sums and early exits equal the serial loop (egdst_solver.c:494-574).
// Returns 0 normal, 1 c1<=0, 2 evf=-inf, <0 hard error (-code).  brk_* describe the break point.
// TT: table type; `lt` (TabL only): the next-period table of state 0 staged in LDS by the caller (MS_NST == 1).
// GW: lanes that share one guess (a whole wave for k_probe / k_fixup; 16 for k_grid_wide, where the groups of a wave
// work on different guesses and may leave the accumulation loop at different shock nodes -- every shuffle reads a
// lane of the own group, which is active whenever the reader is).
template <class TT, int GW = WAVE>
static __device__ __forceinline__ int eg_wave_expectation(BatchRef b, const ms_env *E, int slot1, int draw, const ms_pv *cur,
                                          double savings, int keep, double *rhs_o, double *evf_o, int *nev,
                                          int *brk_ist, double *brk_shock, double *brk_cash, const TT *lt, int lt_sorted = 0)
{
    const int lane = threadIdx.x & (GW - 1);                          // position in the group
    const int gbase = (threadIdx.x & (WAVE - 1)) - lane;            // first lane of the group within the wave
    const int ny = b.g.ny;
    double rhs = 0, evf = 0, checksum = 0, c1_last = 1.0;
    int status = 0, terr = 0, cnt = 0;
    ms_pv nxt;
    nxt.it = cur->it + 1;
    nxt.id = 0;
    nxt.cash = 0;
    nxt.shock = 0;
    nxt.savings = savings;
    for (int ist1 = 0; ist1 < MS_NST && status == 0; ist1++) {
        nxt.ist = ist1;
        if (ms_feasible(E, &nxt) != 1) continue;
        double pr1pre = 0;
        if (MS_OPTIM_TRPRNOSH) {
            pr1pre = ms_trpr(E, cur, &nxt, &terr);
            if (terr) return -25;
            if (pr1pre == 0.0) continue;
        }
        const int niy = (ms_sigma(E, cur, &nxt) <= 0 || ny == 1) ? 1 : ny;
        TT t;
        if (lt)
            t = *lt;
        else {
            const Tab tg = eg_tab(b, slot1, draw, ist1);
            t.M = (decltype(t.M))tg.M, t.C = (decltype(t.C))tg.C, t.V = (decltype(t.V))tg.V;  // (TT == Tab here)
            t.TH = tg.TH, t.D = tg.D, t.len = tg.len, t.thlen = tg.thlen;
        }
        if (t.len < 2) return -10;
        if (t.len > b.g.Sp || t.thlen > b.g.nthrhmax || t.thlen < 1) return -2707;
        const int tsorted = lt ? lt_sorted : eg_tab_sorted(b, cur->it, draw, ist1);  // (see eg_lane_eval)
        for (int base = 0; base < niy && status == 0; base += GW) {
            const int iy = base + lane;
            double pr1 = 0, c1 = 1.0, t_rhs = 0, t_evf = 0, shock = 0, cash = 0;
            int verr = 0;  // this lane's term: valuefunc on a table with one row beside the a0 row (eg_next_value)
            if (iy < niy) {
                ms_pv nl = nxt;
                if (niy == 1) {
                    nl.shock = eg_shock_mean(E, cur, &nl);
                    pr1 = MS_OPTIM_TRPRNOSH ? pr1pre : ms_trpr(E, cur, &nl, &terr);
                } else {
                    nl.shock = eg_shock_node(E, cur, &nl, b.qz[iy]);
                    pr1 = MS_OPTIM_TRPRNOSH ? pr1pre : ms_trpr(E, cur, &nl, &terr);
                    pr1 *= b.qw[iy];
                }
                if (pr1 != 0.0) c1 = eg_term(E, t, cur, &nl, pr1, keep, &t_rhs, &t_evf, tsorted, &verr);
                shock = nl.shock;
                cash = nl.cash;
            }
            if (GW == WAVE ? __any(terr) : (((__ballot(terr) >> gbase) & ((1ull << (GW & 63)) - 1ull)) != 0ull)) return -25;
            const int nl_ = min(GW, niy - base);
            for (int l = 0; l < nl_; l++) {  // ordered accumulation
                const double p = __shfl(pr1, gbase + l);
                if (p == 0.0) continue;
                checksum += p;
                cnt++;
                c1_last = __shfl(c1, gbase + l);
                if (c1_last <= 0) {
                    status = 1;
                    *brk_ist = ist1;
                    *brk_shock = __shfl(shock, gbase + l);
                    *brk_cash = __shfl(cash, gbase + l);
                    break;
                }
                rhs += __shfl(t_rhs, gbase + l);
                if (keep == 1) {
                    if (__shfl(verr, gbase + l)) {  // (egdst_lib.c:183, egdst_solver.c:567-568: the solver returns at once)
                        status = -10;
                        break;
                    }
                    evf += __shfl(t_evf, gbase + l);
                    if (evf == -INFINITY) {
                        status = 2;
                        *brk_ist = ist1;
                        *brk_shock = __shfl(shock, gbase + l);
                        *brk_cash = __shfl(cash, gbase + l);
                        break;
                    }
                }
            }
        }
    }
    *nev += cnt;
    *rhs_o = rhs;
    *evf_o = evf;
    if (status == 0 && fabs(checksum - 1) > EG_TOL) return -11;
    return status;
}

// The guess generator (adraw, egdst_solver.c:955-1159) run by one wave for one (draw, ist, id); all lanes carry
// the same state and every expectation is evaluated cooperatively (eg_wave_expectation).
//   full == 0 (k_probe): stop at the call that would emit the first point of the closed-form grid and hand the
//            limits over to k_grid, which evaluates the remaining points in parallel;
//   full == 1 (k_fixup): keep going point by point to the end of the stream.  This is the exact sequential
//            algorithm; it is used when a zero-consumption resend turns up INSIDE the grid stage
//            (egdst_solver.c:1080-1099), which re-bases every later guess and cannot be speculated in parallel.
template <int NW, int full>
static __device__ __forceinline__ void eg_adraw_cycle(BatchRef b, int it, int draw, int ist, int id)
{
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const bool lead = threadIdx.x == 0;  // all NW waves carry the same generator state; one thread writes
    ms_env E = eg_env(b, draw);
    ms_pv cur;
    cur.it = it;
    cur.ist = ist;
    cur.id = id;
    cur.cash = cur.savings = cur.shock = 0;
    ProbeOut *P = &b.probe[((size_t)draw * MS_NST + ist) * MS_ND + id];
    const int act = ms_feasible(&E, &cur) == 1 && ms_inchoiceset(&E, &cur) == 1;
    if (!act) {
        if (lead) P->active = 0;
        return;
    }
    const int slot1 = (b.g.nslots == 2) ? ((it + 1) & 1) : (it + 1);
    const double a0 = b.g.a0, mmax = b.g.mmax;
    const size_t co = eg_cand(b, draw, ist, id);
    // generator state (aspacestruct, :350-367)
    int ngenerated = 0, ncalls = 0, keep = 0, ntogenerate = b.g.ngridm, np = 0, nev = 0, grid = 0;
    double baseM = 0, baseA = 0, lim1 = 0, lim2 = 0, lim2p = 0, lim3 = 0, lim3p = 0, k3 = 0, last = 0, M = INFINITY;
    double evfa0 = 0.0;
    // k_fixup, one next state: the next-period table goes to LDS once, so that the two searches of every cooperative
    // expectation of a re-basing stream are LDS reads instead of dependent L2 loads (the row past the end is part of
    // the table: the reference reads it for one-row tables)
    bool staged = false;
    int staged_sorted = 0;
    TabL ltab;
    ltab.M = ltab.C = ltab.V = nullptr, ltab.TH = ltab.D = nullptr, ltab.len = ltab.thlen = 0;
    if (full && MS_NST == 1) {
        __shared__ double fx_tab[3 * EG_FIX_LROWS];
        const Tab tg = eg_tab(b, slot1, draw, 0);
        const int nrow = min(tg.len + 1, b.g.Sp);
        staged = tg.len >= 2 && nrow <= EG_FIX_LROWS;
        if (staged) {
            for (int i = threadIdx.x; i < nrow; i += NW * WAVE) {
                fx_tab[i] = tg.M[i];
                fx_tab[EG_FIX_LROWS + i] = tg.C[i];
                fx_tab[2 * EG_FIX_LROWS + i] = tg.V[i];
            }
            ltab.M = (const eg_ldsd *)fx_tab;
            ltab.C = ltab.M + EG_FIX_LROWS;
            ltab.V = ltab.M + 2 * EG_FIX_LROWS;
            ltab.TH = tg.TH, ltab.D = tg.D, ltab.len = tg.len, ltab.thlen = tg.thlen;
        }
        __syncthreads();
        if (staged) {   // is the staged M column in order?  then one bracket search serves both interpolations of a term (eg_term)
            __shared__ int fx_bad;
            if (threadIdx.x == 0) fx_bad = 0;
            __syncthreads();
            int bad = 0;
            for (int i = threadIdx.x; i + 1 < tg.len; i += NW * WAVE)
                if (!(fx_tab[i] <= fx_tab[i + 1])) bad = 1;
            if (bad) fx_bad = 1;  // (benign race: every writer stores 1)
            __syncthreads();
            staged_sorted = (tg.len >= 4 && !fx_bad) ? 1 : 0;
        }
    }
    double fp_last = NAN, fp_M = NAN, kp_M = 0, kp_C = 0, kp_V = 0;  // previous visit of the resend branch; the point kept last
    int fp_ncalls = -2, fp_ngen = -1, fp_np = 0, fp_nev = 0;
    // the visits before that one (cycles of the resend, below): [0] two visits ago, [1] three, [2] four.  In LDS, a copy per wave
    // (every wave carries the whole generator state; its lane 0 writes): in registers the history cost k_probe 44 VGPRs and a wave
    // per SIMD (168 -> 212)
    __shared__ double cy_d[NW][3][2];  // guess, returned M
    __shared__ int cy_i[NW][3][4];     // ncalls, ngenerated, np, nev at the visit
    if (lane == 0)
        for (int q = 0; q < 3; q++) cy_i[wave][q][0] = -9;
    EG_WSYNC();
    int last_cnt = 0;  // evaluations of the most recent expectation
    int skipped = 0;   // calls of the stage-0 fixed point that were accounted for without being executed
    int seq_left = 0;  // full mode: grid guesses to evaluate one at a time before batching again
    int small = 0;     // full mode: the next batch is a small one (several lanes per guess), see the batch below
#ifdef EGDST_FIXSTAT  // diagnostic: what a regeneration consists of (dbg ints 8..11, ticks in ints 12-13)
    int fs_batches = 0, fs_single = 0, fs_resend = 0;
    const unsigned long long fs_t0 = wall_clock64();
#endif
#ifdef EGDST_CENSUS
    const unsigned long long cz_t0_ = wall_clock64();
#endif
    for (;;) {
        // ---- next guess -------------------------------------------------------------------
        if (ncalls + 1 >= b.g.ngridmax) {  // runaway guard (:963-978): the stream simply ends
            ncalls += 1;
            break;
        }
        if (ngenerated == 0) {
            if (M == INFINITY && ncalls > 0) {
                // Stage 0 asked for mmax and got M=+inf back: the generator asks for mmax again, and the expectation
                // is a pure function of the guess, so every further call repeats this one until the runaway guard
                // (:963-978) ends the stream.  Account for those calls instead of executing them.
                const int k = b.g.ngridmax - 1 - ncalls;  // calls that would still evaluate
                nev += k * last_cnt;
                if (lead && k > 0) atomicAdd(&b.credited[draw], (unsigned long long)k * (unsigned long long)last_cnt);
                skipped = k + 1;  // (accounted for, not executed: no straggler work)
                ncalls = b.g.ngridmax;
                break;
            }
            ncalls += 1;
            keep = 0;
            if (M == INFINITY)
                last = mmax;
            else if (M <= mmax) {
                baseA = last;
                baseM = M;
                ngenerated = 1;
                keep = 1;
                last = a0;
                k3 = 0;
            } else {
                if (last - a0 < EG_TOL) {
                    if (lead) eg_fail(b, draw, it, ist, 19);
                    return;
                }
                last = (last + a0) / 2;
            }
        } else {
            double aa = (M - baseM) / (last - baseA);
            double bb = baseM - aa * baseA;
            if (ngenerated == 1 && k3 == 0) {  // limits after the first kept call (:1051-1078)
                ntogenerate = b.g.ngridm;
                lim2p = MS_MIN(mmax, (mmax - bb) / aa);
                lim3p = -bb / aa;
                if (a0 < 0 && a0 < lim3p)
                    k3 = MS_MAX(floor(ntogenerate * (lim3p - a0) / (lim2p - a0)), 2.0);
                else {
                    lim3p = a0;
                    k3 = 1.0;
                }
                lim1 = ms_tr(&E, &cur, lim3p - a0);
                lim2 = ms_tr(&E, &cur, lim2p - lim3p);
                lim3 = ms_tr(&E, &cur, 0);
            }
            if (M <= a0 - 1 + EG_TOL) {  // c1<=0 signal: resend the prepared point (:1080-1099)
                // Fixed point of the resend.  The generator signals "c1<=0" in band, by M = a0-1; when the re-sent guess
                // itself returns an M below a0-1 (a re-based guess far below the credit limit), that genuine M is taken
                // for the signal, the same guess is sent again, and -- the expectation being a pure function of the guess --
                // every call from here to the runaway guard (:963-978) repeats the previous one: same guess, same M, the
                // same kept point appended.  Seen on ~0.2 % of the C2 parameter draws, ~9000 sequential calls each.  Two
                // consecutive visits with identical (guess, returned M) prove it; the remaining calls are then accounted
                // for (points written, evaluations credited) instead of executed, exactly as the reference would end.
                if (last == fp_last && M == fp_M && ncalls == fp_ncalls + 1 && ngenerated == fp_ngen) {
                    const int dnp = np - fp_np, dnev = nev - fp_nev;
                    const int k = b.g.ngridmax - 1 - ncalls;  // calls that would still run before the guard ends the stream
                    if (k > 0 && dnp >= 0 && dnp <= 1) {
                        if (dnp == 1) {
                            if (np + k - 1 >= b.g.ngridmax - 1) {  // (:662) the repeats fill the grid first
                                if (lead) eg_fail(b, draw, it, ist, 13);
                                return;
                            }
                            if (np + k > b.g.Cp) {
                                if (lead) eg_fail(b, draw, it, ist, EGDST_E_CAPACITY);
                                return;
                            }
                            for (int q = threadIdx.x; q < k; q += NW * WAVE) {
                                b.cM[co + np + q] = kp_M;
                                b.cC[co + np + q] = kp_C;
                                b.cV[co + np + q] = kp_V;
                            }
                            np += k;
                        }
                        nev += k * dnev;
                        if (lead && dnev > 0) atomicAdd(&b.credited[draw], (unsigned long long)k * (unsigned long long)dnev);
                        skipped += k + 1;
                        ncalls = b.g.ngridmax;
                        break;
                    }
                }
                // A CYCLE of the resend: the re-sent guess signals c1<=0 again at ANOTHER shock node, the guess prepared for that node
                // signals it at the first one again, and so on -- the same (guess, M) as two, three or four visits ago with nothing
                // but resend visits in between and no point kept.  The generator's state at a visit is a function of (guess, M) and of
                // constants of the stream, so the visits repeat with that period up to the runaway guard (:963-978); seen on C2 with
                // a0 = -5 (one draw in ~10 000: 10 000 sequential calls of 9 us, a whole batch waiting 90 ms for one wave).  The
                // remaining calls are accounted for: evaluations credited phase by phase, nothing kept.
                for (int q = 0; q < 3; q++) {
                    const int per = q + 2;  // the period this entry would prove
                    if (!(last == cy_d[wave][q][0] && M == cy_d[wave][q][1] && ncalls == cy_i[wave][q][0] + per && ngenerated == cy_i[wave][q][1] &&
                          np == cy_i[wave][q][2]))
                        continue;
                    const int k = b.g.ngridmax - 1 - ncalls;  // calls that would still run before the guard ends the stream
                    if (k <= 0) break;
                    // evaluations of the remaining calls: whole periods (the count now minus the count `per` visits ago) and the first
                    // k mod per calls of one more (the count at the visit that many after the old one, minus the old one's)
                    const int m = k % per, at0 = cy_i[wave][q][3], atm = (m <= q) ? cy_i[wave][q - min(m, q)][3] : fp_nev;
                    const int add = (k / per) * (nev - at0) + (atm - at0);
                    nev += add;
                    if (lead && add > 0) atomicAdd(&b.credited[draw], (unsigned long long)add);
                    skipped += k + 1;
                    ncalls = b.g.ngridmax;
                    break;
                }
                if (ncalls >= b.g.ngridmax) break;
                EG_WSYNC();  // (every lane has read the history)
                if (lane == 0) {
                    for (int q = 2; q > 0; q--) {
                        cy_d[wave][q][0] = cy_d[wave][q - 1][0], cy_d[wave][q][1] = cy_d[wave][q - 1][1];
                        for (int j = 0; j < 4; j++) cy_i[wave][q][j] = cy_i[wave][q - 1][j];
                    }
                    cy_d[wave][0][0] = fp_last, cy_d[wave][0][1] = fp_M;
                    cy_i[wave][0][0] = fp_ncalls, cy_i[wave][0][1] = fp_ngen, cy_i[wave][0][2] = fp_np, cy_i[wave][0][3] = fp_nev;
                }
                EG_WSYNC();
                fp_last = last, fp_M = M, fp_ncalls = ncalls, fp_ngen = ngenerated, fp_np = np, fp_nev = nev;
                ncalls += 1;
                keep = 1;
                aa = (a0 - baseM) / (a0 - baseA);
                bb = baseM - aa * baseA;
                lim2p = MS_MIN(mmax, (mmax - bb) / aa);
                lim3p = last - EG_ZEROC;
                k3 = 1.0;
                lim1 = ms_tr(&E, &cur, lim3p - a0);
                lim2 = ms_tr(&E, &cur, lim2p - lim3p);
                lim3 = ms_tr(&E, &cur, 0);
            } else if (!full) {
                // the next call either starts the closed-form grid (handled by k_grid) or ends the stream
                grid = (M < mmax && ngenerated < ntogenerate) ? 1 : 0;
                break;
            } else if (full && seq_left > 0 && M < mmax && ngenerated < ntogenerate) {
                // Right after a resend the next guess often signals c1<=0 again (streams that re-base at every
                // point exist): take the next few grid guesses one at a time, evaluated cooperatively below,
                // before paying for a whole batch of speculative evaluations again.
                GridLims GL;
                GL.lim1 = lim1, GL.lim2 = lim2, GL.lim3 = lim3, GL.lim3p = lim3p, GL.k3 = k3, GL.ntogenerate = ntogenerate;
                last = eg_grid_A(&E, &cur, GL, ngenerated - 1, last, ngenerated);
                ngenerated += 1;
                ncalls += 1;
                keep = 1;
                seq_left -= 1;
            } else if (M < mmax && ngenerated < ntogenerate) {
                // Grid stage of the sequential stream (:1100-1149), WAVE guesses at a time: lane l evaluates the
                // guess the generator would emit l calls from now (serial shock loop inside the lane, as k_grid
                // does); the results are then consumed in stream order up to the first one that stops the stream
                // or signals c1<=0 -- exactly the calls the reference would have made.
                GridLims GL;
                GL.lim1 = lim1, GL.lim2 = lim2, GL.lim3 = lim3, GL.lim3p = lim3p, GL.k3 = k3, GL.ntogenerate = ntogenerate;
                __shared__ unsigned long long sx_can[NW], sx_hard[NW], sx_neg[NW], sx_stop[NW], sx_kept[NW], sx_inf[NW];
                __shared__ int sx_hst[NW], sx_cnt[NW], sx_bist;
                __shared__ double sx_take[2], sx_neg4[3];
#ifdef EGDST_FIXSTAT
                fs_batches++;
#endif
                // Two shapes of a batch.  The regular one: a guess per lane, its (next state, shock node) terms one after the other inside
                // the lane (as k_grid_lds does).  Right after a resend (small != 0) the next c1<=0 is usually a few points away -- a
                // stream re-bases five times on average, nearly always within its first dozens of points -- and a whole batch of
                // serial evaluations is 35 us of latency for a handful of consumed guesses: then FIX_LPG lanes share a guess, a
                // lane per shock node as in k_grid_wide (eg_wave_expectation with groups of FIX_LPG: the same terms accumulated in
                // the same order), WAVE / FIX_LPG guesses per wave -- a quarter of the latency.  A small batch that is consumed
                // whole without a signal hands back to the regular shape.  Only the lane (lane % lpg) == 0 of a guess votes below.
                const int lsh = small ? FIX_LPG_SHIFT : 0, lpg = 1 << lsh, gpw = WAVE >> lsh;  // lanes per guess, guesses per wave
                const bool rep = (lane & (lpg - 1)) == 0;
                const int gl = wave * gpw + (lane >> lsh);  // position of this thread's guess in the batch
                const int n = ngenerated + gl;              // value of `ngenerated` at this thread's call
                const bool can = n < ntogenerate && (ncalls + 1 + gl) < b.g.ngridmax;
                LaneEval r;
                r.status = 0, r.cnt = 0, r.bist = 0, r.M = NAN, r.C = r.V = r.R = 0, r.bshock = r.bcash = 0;
                double An = last;
                if (can) {
                    An = eg_grid_A(&E, &cur, GL, ngenerated - 1, last, n);
#if FIX_LPG_SHIFT > 0
                    if (small) {
                        double rhs_ = 0, evf_ = 0;
                        int cnt_ = 0, st_;
                        if (full && staged)
                            st_ = eg_wave_expectation<TabL, (1 << FIX_LPG_SHIFT)>(b, &E, slot1, draw, &cur, An, 1, &rhs_, &evf_, &cnt_, &r.bist, &r.bshock, &r.bcash, &ltab, staged_sorted);
                        else
                            st_ = eg_wave_expectation<Tab, (1 << FIX_LPG_SHIFT)>(b, &E, slot1, draw, &cur, An, 1, &rhs_, &evf_, &cnt_, &r.bist, &r.bshock, &r.bcash, (const Tab *)nullptr);
                        r.status = st_, r.cnt = cnt_;  // (what eg_lane_eval reports for the point, as in k_grid_wide)
                        if (st_ == 0) {
                            rhs_ *= ms_discount(&E, &cur);
                            r.M = An + ms_utility_marginal_inverse(&E, &cur, rhs_);
                            r.C = r.M - An;
                            r.V = ms_utility(&E, &cur, r.C) + ms_discount(&E, &cur) * evf_;
                            r.R = r.M;
                        } else {
                            r.M = NAN;
                            r.C = r.V = 0;
                            r.R = (st_ == 1) ? b.g.a0 - 1 : r.bcash;
                        }
                    } else
#endif
                    r = (full && staged) ? eg_lane_eval<TabL>(b, &E, &cur, slot1, draw, An, &ltab, staged_sorted) : eg_lane_eval<Tab>(b, &E, &cur, slot1, draw, An);
                }
                {   // phase 1: what every wave found, in batch order
                    const unsigned long long canm = __ballot(can && rep);
                    const unsigned long long hardm = __ballot(can && rep && r.status < 0);
                    const unsigned long long negm = __ballot(can && rep && r.status == 1);
                    const unsigned long long stopm = __ballot(can && rep && r.status != 1 && !(r.R < mmax));
                    const int hst = __shfl(r.status, hardm ? __ffsll((long long)hardm) - 1 : 0);
                    if (lane == 0) {
                        sx_can[wave] = canm;
                        sx_hard[wave] = hardm;
                        sx_neg[wave] = negm;
                        sx_stop[wave] = stopm;
                        sx_hst[wave] = hst;
                    }
                }
                __syncthreads();
                int ncan = 0, fneg = -1, fstop = -1, fhard = -1, hcode = 0;
                for (int w = 0; w < NW; w++) {  // threads [0, ncan) of the batch hold requested-if-reached guesses
                    ncan += __popcll(sx_can[w]);
                    if (fneg < 0 && sx_neg[w]) fneg = w * gpw + ((__ffsll((long long)sx_neg[w]) - 1) >> lsh);
                    if (fstop < 0 && sx_stop[w]) fstop = w * gpw + ((__ffsll((long long)sx_stop[w]) - 1) >> lsh);
                    if (fhard < 0 && sx_hard[w]) {
                        fhard = w * gpw + ((__ffsll((long long)sx_hard[w]) - 1) >> lsh);
                        hcode = -sx_hst[w];
                    }
                }
                if (fneg < 0) fneg = ncan;
                if (fstop < 0) fstop = ncan;
                // guesses [0, take) are consumed as ordinary calls; a stopping point is itself consumed
                const int take = (fneg <= fstop) ? fneg : min(fstop + 1, ncan);
                const bool negnext = fneg <= fstop && fneg < ncan;  // the call after them hits c1<=0
                if (fhard >= 0 && (fhard < take || (negnext && fhard == fneg))) {
                    if (lead) eg_fail(b, draw, it, ist, hcode);
                    return;
                }
                const bool kept = rep && gl < take && r.status == 0 && isfinite(r.M);
                {   // phase 2
                    const unsigned long long keptm = __ballot(kept);
                    const unsigned long long infm = __ballot(rep && gl < take && r.status == 2);
                    // evaluations of the consumed calls (and of the c1<=0 call, if it is next)
                    int c = (rep && (gl < take || (negnext && gl == fneg))) ? r.cnt : 0;
                    for (int o = WAVE / 2; o > 0; o >>= 1) c += __shfl_xor(c, o);
                    if (lane == 0) {
                        sx_kept[wave] = keptm;
                        sx_inf[wave] = infm;
                        sx_cnt[wave] = c;
                    }
                    if (rep && take > 0 && gl == take - 1) {
                        sx_take[0] = An;
                        sx_take[1] = r.R;
                    }
                    if (rep && negnext && gl == fneg) {
                        sx_neg4[0] = An;
                        sx_neg4[1] = r.bshock;
                        sx_neg4[2] = r.bcash;
                        sx_bist = r.bist;
                    }
                }
                __syncthreads();
                int nkept = 0, kbefore = 0, csum = 0;
                bool anyinf = false;
                for (int w = 0; w < NW; w++) {
                    if (w < wave) kbefore += __popcll(sx_kept[w]);
                    nkept += __popcll(sx_kept[w]);
                    csum += sx_cnt[w];
                    anyinf = anyinf || sx_inf[w] != 0;
                }
                if (np + nkept > b.g.ngridmax - 1) {  // (:662) the point that makes the count reach ngridmax
                    if (lead) eg_fail(b, draw, it, ist, 13);
                    return;
                }
                if (np + nkept > b.g.Cp) {
                    if (lead) eg_fail(b, draw, it, ist, EGDST_E_CAPACITY);
                    return;
                }
                if (kept) {
                    const size_t o = co + np + kbefore + __popcll(sx_kept[wave] & ((1ull << lane) - 1ull));
                    b.cM[o] = r.M;
                    b.cC[o] = r.C;
                    b.cV[o] = r.V;
                }
                np += nkept;
                if (anyinf) evfa0 = -INFINITY;
                nev += csum;
                if (take > 0) {
                    last = sx_take[0];
                    M = sx_take[1];
                    ngenerated += take;
                    ncalls += take;
                }
                keep = 1;
                if (!negnext && take == ncan) small = 0;  // (consumed whole, no signal: back to a guess per lane)
                if (negnext) {  // the next call hit c1<=0 (:583-621): prepare the resend
#ifdef EGDST_FIXSTAT
                    fs_resend++;
#endif
                    small = FIX_LPG_SHIFT > 0 ? 1 : 0;
                    seq_left = EG_SEQ_AFTER_RESEND;
                    ms_pv nb;
                    nb.it = it + 1;
                    nb.ist = sx_bist;
                    nb.id = 0;
                    nb.shock = sx_neg4[1];
                    nb.cash = sx_neg4[2];
                    nb.savings = sx_neg4[0];
                    const Tab tb = eg_tab(b, slot1, draw, nb.ist);
                    int ierr = 0;
                    ngenerated += 1;
                    ncalls += 1;
                    evfa0 = -INFINITY;
                    M = a0 - 1;
                    last = eg_invert_budget(&E, cur, nb, (tb.V[0] > -INFINITY) ? a0 : tb.M[1], &ierr) + EG_ZEROC;
                    if (ierr) {
                        if (lead) eg_fail(b, draw, it, ist, ierr);
                        return;
                    }
                } else if (take == 0)
                    break;  // nothing could be requested (runaway guard, :963-978)
                continue;   // (a consumed stopping point ends the stream at the next turn: M >= mmax)
            } else
                break;  // stream ends (:1150-1152)
        }
        // ---- evaluate the guess -----------------------------------------------------------
#ifdef EGDST_FIXSTAT
        fs_single++;
#endif
        double rhs, evf;
        int bist = 0;
        double bshock = 0, bcash = 0;
        const int nev0 = nev;
        int st;
        if (full && staged)
            st = eg_wave_expectation<TabL>(b, &E, slot1, draw, &cur, last, keep, &rhs, &evf, &nev, &bist, &bshock, &bcash, &ltab, staged_sorted);
        else
            st = eg_wave_expectation<Tab>(b, &E, slot1, draw, &cur, last, keep, &rhs, &evf, &nev, &bist, &bshock, &bcash,
                                          (const Tab *)nullptr);
        last_cnt = nev - nev0;
        if (st < 0) {
            if (lead) eg_fail(b, draw, it, ist, -st);
            return;
        }
        if (st > 0) {  // emergency (:583-627)
            if (ngenerated == 0) {
                if (lead) eg_fail(b, draw, it, ist, 12);
                return;
            }
            evfa0 = -INFINITY;
            M = bcash;
            if (st == 1) {
                small = (full && FIX_LPG_SHIFT > 0) ? 1 : 0;
                seq_left = EG_SEQ_AFTER_RESEND;
                ms_pv nb;
                nb.it = it + 1;
                nb.ist = bist;
                nb.id = 0;
                nb.shock = bshock;
                nb.cash = bcash;
                nb.savings = last;
                const Tab tb = eg_tab(b, slot1, draw, bist);
                int ierr = 0;
                M = a0 - 1;
                last = eg_invert_budget(&E, cur, nb, (tb.V[0] > -INFINITY) ? a0 : tb.M[1], &ierr) + EG_ZEROC;
                if (ierr) {
                    if (lead) eg_fail(b, draw, it, ist, ierr);
                    return;
                }
            }
            continue;
        }
        rhs *= ms_discount(&E, &cur);
        M = last + ms_utility_marginal_inverse(&E, &cur, rhs);
        if (keep == 1 && isfinite(M)) {
            if (fabs(last - a0) < EG_TOL && evfa0 > -INFINITY) evfa0 = evf;
            if (np >= b.g.ngridmax - 1) {  // (:662)
                if (lead) eg_fail(b, draw, it, ist, 13);
                return;
            }
            if (np >= b.g.Cp) {
                if (lead) eg_fail(b, draw, it, ist, EGDST_E_CAPACITY);
                return;
            }
            kp_M = M;
            kp_C = M - last;
            kp_V = ms_utility(&E, &cur, kp_C) + ms_discount(&E, &cur) * evf;
            if (lead) {
                b.cM[co + np] = kp_M;
                b.cC[co + np] = kp_C;
                b.cV[co + np] = kp_V;
            }
            np += 1;  // full == 0: normally at most one kept point precedes the grid stage
        }
    }
#ifdef EGDST_EMU
    if (lead && full && getenv("EGDST_TRACE_FIXUP"))
        fprintf(stderr, "fixup end it=%d id=%d ncalls=%d ngen=%d np=%d last=%g M=%g nev=%d\n", it, id, ncalls, ngenerated, np, last, M, nev);
#endif
#ifdef EGDST_FIXSTAT
    if (lead && full) {
        atomicAdd(&b.dbg[16 * draw + 8], fs_batches), atomicAdd(&b.dbg[16 * draw + 9], fs_single);
        atomicAdd(&b.dbg[16 * draw + 10], fs_resend), atomicAdd(&b.dbg[16 * draw + 11], 1);
        atomicAdd((unsigned long long *)(b.dbg + 16 * draw) + 6, wall_clock64() - fs_t0);
    }
#endif
#ifdef EGDST_CENSUS
    if (lead) {
        const unsigned long long cz_ = wall_clock64() - cz_t0_;
        if (full) EG_CENSUS(3, draw, it, id, np, ncalls, skipped, cz_);
        else if (cz_ > 5000ull) EG_CENSUS(4, draw, it, id, ncalls, skipped, np, cz_);
    }
#endif
    // re-basing calls of this stream beyond the regular ones: the host schedules draws with many of them apart
    if (lead && ncalls - skipped > (full ? ntogenerate : 0) + 64)
        atomicAdd(&b.work[draw], (unsigned)(ncalls - skipped - (full ? ntogenerate : 0)));
    if (lead) {
        P->active = 1;
        P->seq = full;
        P->np = np;
        P->grid = grid;
        P->ncalls = ncalls;
        P->ntogenerate = ntogenerate;
        P->probe_evals = nev;
        P->lim1 = lim1;
        P->lim2 = lim2;
        P->lim3 = lim3;
        P->lim3p = lim3p;
        P->k3 = k3;
        P->A0 = last;
        P->M0 = M;
        P->evfa0 = evfa0;
    }
}

#ifndef PROBE_WAVES
#define PROBE_WAVES 3  // waves per SIMD k_probe is compiled for (at most 168 VGPRs): a latency-bound kernel of one-wave workgroups lives on
#endif                 // how many of them share a SIMD (round 4: the cycle history of the resend took it to 181 VGPRs and 7.1 -> 9.4 ms per C2 solve)
#ifdef EGDST_EMU
#define PROBE_ATTR
#else
#define PROBE_ATTR __attribute__((amdgpu_waves_per_eu(PROBE_WAVES, PROBE_WAVES)))
#endif
__global__ void __launch_bounds__(WAVE) PROBE_ATTR k_probe(const Batch *bp_, int it)
{
    BatchRef b = EG_BATCH_REF(bp_);
    const int combo = blockIdx.x;
    const int id = combo % MS_ND, ist = (combo / MS_ND) % MS_NST, draw = b.order[b.draw0 + combo / (MS_ND * MS_NST)];
    if (b.status[draw]) return;
    if (threadIdx.x == 0) b.negflag[((size_t)draw * MS_NST + ist) * MS_ND + id] = 0;
    eg_adraw_cycle<1, 0>(b, it, draw, ist, id);
}

// After k_grid: does the stream of (draw, ist, id) contain a zero-consumption signal among the points the generator
// would actually request?  k_fixup_scan (one wave per stream, no LDS) lists those streams; k_fixup (a small grid of
// 4-wave workgroups with the LDS table buffer) redoes each listed stream sequentially, exactly as the reference does.
// `cnt` is this (group, period)'s counter, `list` the group's list (one entry per stream at most).
__global__ void __launch_bounds__(WAVE) k_fixup_scan(const Batch *bp_, int it, int *cnt, int *list)
{
    BatchRef b = EG_BATCH_REF(bp_);
    const int combo = blockIdx.x;
    const int id = combo % MS_ND, ist = (combo / MS_ND) % MS_NST, draw = b.order[b.draw0 + combo / (MS_ND * MS_NST)];
    if (b.status[draw]) return;
    const int lane = threadIdx.x & (WAVE - 1);
    if (!b.negflag[((size_t)draw * MS_NST + ist) * MS_ND + id]) return;  // no grid point of the stream signalled c1<=0
    const ProbeOut P = b.probe[((size_t)draw * MS_NST + ist) * MS_ND + id];
    if (!P.active || !P.grid) return;
    const size_t co = eg_cand(b, draw, ist, id);
    const int navail = min(b.g.ngridm - 1, b.g.ngridmax - 1 - P.ncalls);
    int resend = 0;
    for (int base = 1; base <= navail; base += WAVE) {  // in stream order, 64 points at a time
        const int n = base + lane;
        int stop = 0, neg = 0;
        if (n <= navail) {
            stop = !(b.cM[co + n] < b.g.mmax);
            neg = (eg_sc_status(b.cSt[co + n]) == 1);
        }
        const unsigned long long ms = __ballot(stop), mn = __ballot(neg);
        if (mn) {
            const int fneg = __ffsll((long long)mn) - 1, fstop = ms ? __ffsll((long long)ms) - 1 : WAVE;
            if (fneg <= fstop) resend = 1;  // a stopping point is itself still requested
        }
        if (resend || ms) break;
    }
    if (resend && lane == 0) list[atomicAdd((unsigned *)cnt, 1u)] = combo;
}

#ifndef FIX_MINW
#define FIX_MINW 1  // (3: at most 168 VGPRs, so that a k_fixup wave fits a SIMD beside two waves of the -DENV_MINW=3 k_envelope)
#endif
__global__ void __launch_bounds__(FIX_BS, FIX_MINW) k_fixup(const Batch *bp_, int it, const int *cnt, const int *list)
{
    BatchRef b = EG_BATCH_REF(bp_);
    const int n = *cnt;
    for (int k = blockIdx.x; k < n; k += gridDim.x) {
        const int combo = list[k];
        const int id = combo % MS_ND, ist = (combo / MS_ND) % MS_NST, draw = b.order[b.draw0 + combo / (MS_ND * MS_NST)];
#ifdef EGDST_EMU
        if (threadIdx.x == 0 && getenv("EGDST_TRACE_FIXUP")) fprintf(stderr, "fixup it=%d draw=%d ist=%d id=%d\n", it, draw, ist, id);
#endif
        if (threadIdx.x == 0) atomicAdd(&b.nregen[draw], 1u);
        eg_adraw_cycle<FIX_BS / WAVE, 1>(b, it, draw, ist, id);
        __syncthreads();  // the LDS buffers of the stream generator are reused by the next listed stream
    }
}

// ---------------------------------------------------------------------------------------------
// Closed-form grid point n (1 <= n <= ngridm-1) and its EGM evaluation; one lane per point.
__global__ void __launch_bounds__(GRID_BS) k_grid(const Batch *bp_, int it)
{
    BatchRef b = EG_BATCH_REF(bp_);
    const int combo = blockIdx.y;
    const int id = combo % MS_ND, ist = (combo / MS_ND) % MS_NST, draw = b.order[b.draw0 + combo / (MS_ND * MS_NST)];
    const int n = blockIdx.x * GRID_BS + threadIdx.x + 1;
    if (b.status[draw]) return;
    const ProbeOut P = b.probe[((size_t)draw * MS_NST + ist) * MS_ND + id];
    if (!P.active || !P.grid || n >= b.g.ngridm) return;
    ms_env E = eg_env(b, draw);
    ms_pv cur;
    cur.it = it;
    cur.ist = ist;
    cur.id = id;
    cur.cash = cur.savings = cur.shock = 0;
    GridLims L;
    L.lim1 = P.lim1, L.lim2 = P.lim2, L.lim3 = P.lim3, L.lim3p = P.lim3p, L.k3 = P.k3, L.ntogenerate = P.ntogenerate;
    const double A = eg_grid_A(&E, &cur, L, 0, P.A0, n);
    const int slot1 = (b.g.nslots == 2) ? ((it + 1) & 1) : (it + 1);
    const LaneEval r = eg_lane_eval(b, &E, &cur, slot1, draw, A);
    const size_t o = eg_cand(b, draw, ist, id) + n;
    if (r.status == 1) b.negflag[((size_t)draw * MS_NST + ist) * MS_ND + id] = 1;  // (rare; k_fixup_scan looks closer)
    b.cSt[o] = eg_sc_pack(r.status, r.cnt);
    b.cM[o] = r.R;  // (== r.M for a normal point, see eg_sc_pack)
    if (r.status == 0) {
        b.cC[o] = r.C;
        b.cV[o] = r.V;
    }
}

// ---------------------------------------------------------------------------------------------
// k_grid with the searched columns in LDS (batches that fill the GPU).  Counters of round 2 on C2 x 4096
// (profiles/r02_pmc_c2_4096.csv): k_grid executes ~360 vector instructions and 22 dependent global loads per
// evaluation and its waves wait 63 % of their cycles.  Three changes, none of which touches a floating-point operation:
//   * the M column of every next-period table a lane can reach is staged in LDS once per workgroup (8 B per row), so the
//     bracket search is 32-bit LDS addressing instead of dependent L2 round trips; C and V are read at the bracket only;
//   * ONE bracket search per evaluation: valuefunc searches the same column shifted by one row (egdst_solver.c:768,
//     egdst_lib.c:179-206), and on a non-decreasing column its bracket follows from the first one (eg_second_bracket);
//     the workgroup checks the order while staging and falls back to the two searches otherwise;
//   * the shock nodes exp(mu + z sigma) of a (state, decision, next state) do not depend on the asset point when mu and
//     sigma do not (MS_SHOCK_NODES_SHARED, found by the code generator): computed once per workgroup.
// Same arithmetic per evaluation as eg_lane_eval / eg_term / eg_next_value, same order of accumulation.
#ifndef EG_GRID_LROWS_MAX
#define EG_GRID_LROWS_MAX 4096  // rows of M columns (all next states together) a workgroup may stage: 32 KB
#endif
#define EG_GRID_NYMAX 32        // shock nodes kept per next state

// bracket of valuefunc's search over (M+1, len-1) given the bracket i of linter's search over (M, len), for a
// non-decreasing column with len >= 4 (both searches return "the last row <= x" clamped to their ranges)
static __device__ __forceinline__ int eg_second_bracket(double x, int i, double m2, double mlast2, int n1)
{
    return (x < m2) ? 0 : ((x >= mlast2) ? n1 - 3 : i - 1);
}

// eg_bracket(x, g, n, 0) on a NON-DECREASING column, starting from a hint: the bracket of a neighbouring asset point for the
// same next state and shock node.  The search returns the last row i in [1, n-3] with g[i] <= x (0 below g[1], n-2 from
// g[n-2] on), which on an ordered column is unique -- however it is found.  Cash-in-hand moves by less than a table row or two
// from one asset point to the next, so a step or two from the neighbour's bracket replaces log2(n) dependent reads; after
// EG_NEAR_STEPS steps (the hint was far off) or for a NaN the full search decides.
#ifndef EG_NEAR_STEPS
#define EG_NEAR_STEPS 6
#endif
template <class P> static __device__ __forceinline__ int eg_bracket_near(double x, P g, int n, int hint)
{
    if (x < g[1]) return 0;
    if (x >= g[n - 2]) return n - 2;
    int i = min(max(hint, 1), n - 3);
    if (g[i] <= x) {
        for (int k = 0; k < EG_NEAR_STEPS; k++) {
            if (!(g[i + 1] <= x)) return i;  // (i + 1 <= n - 2 and x < g[n-2]: the scan stays inside)
            i++;
        }
    } else if (g[i] > x) {
        for (int k = 0; k < EG_NEAR_STEPS; k++) {
            i--;
            if (g[i] <= x) return i;         // (g[1] <= x: the scan stops at row 1 at the latest)
        }
    }
    return eg_bracket(x, g, n, 0);
}

// eg_term + eg_next_value for keep == 1 with the M column in LDS and one search (see above).  ibr: nullptr, or in: the
// bracket of the neighbouring asset point (< 0: none), out: this point's.
// TT: Tab (C and V in global memory) or TabL (staged in LDS as well, k_grid_lds_cv)
// sorted: the staged column is known to be in order (the callers' fast path): the branch-free search
template <class TT>
static __device__ __forceinline__ double eg_term_lds(const ms_env *E, const eg_ldsd *M, const TT &t, const ms_pv *cur, ms_pv *nxt,
                                                     double pr1, double *t_rhs, double *t_evf, int *ibr = nullptr, int sorted = 0)
{
    nxt->cash = ms_cashinhand(E, cur, nxt);
    const double x = nxt->cash;
    const int n1 = t.len;
    const int i = (ibr && *ibr >= 0) ? eg_bracket_near(x, M, n1, *ibr) : (sorted ? eg_bracket_sorted(x, M, n1) : eg_bracket(x, M, n1, 0));
    if (ibr) *ibr = i;
    const double mlast = M[n1 - 1], mfirst = M[1];
    // rows i and i+1 of C and of V in ONE round of global reads: away from the table's ends valuefunc's bracket is the same
    // pair of rows (j+1 == i below), and a lane that waits twice per evaluation waits half as often
    const double Ci = t.C[i], Ci1 = t.C[i + 1], Vi = t.V[i], Vi1 = t.V[i + 1];
    double c1 = eg_lerp_fast(x, M[i], M[i + 1], Ci, Ci1);
    if (x > mlast) c1 = MS_MAX(c1, t.C[n1 - 1]);  // constant extrapolation, :554
    *t_rhs = 0;
    *t_evf = 0;
    if (c1 <= 0) return c1;
    if (!MS_OPTIM_MUNOD || (!MS_OPTIM_UNOD && x < mfirst))
        nxt->id = (int)t.D[eg_bracket(x, t.TH, t.thlen, 1)];  // optimd, egdst_lib.c:129-132
    else
        nxt->id = 0;
    *t_rhs = pr1 * ms_utility_marginal(E, nxt, c1) * ms_cashinhand_marginal(E, cur, nxt);
    const double evf1 = t.V[0], a0 = E->a0;
    double val;
    if (x < mfirst && evf1 > -INFINITY)
        val = ms_utility(E, nxt, x - a0) + ms_discount(E, nxt) * evf1;
    else {
        const int j = eg_second_bracket(x, i, M[2], M[n1 - 2], n1);  // == eg_bracket(x, M + 1, n1 - 1, 0)
        const double f0 = (j + 1 == i) ? Vi : t.V[j + 1], f1 = (j + 1 == i) ? Vi1 : t.V[j + 2];
        if (!isfinite(f0))
            val = f0;
        else if (!isfinite(f1))
            val = f1;
        else {
            const double g0 = M[j + 1], g1 = M[j + 2];
            if (x > a0 && (x > mlast || x < mfirst)) {
                const double tx = ms_tr(E, nxt, x - a0), t0 = ms_tr(E, nxt, g0 - a0), t1 = ms_tr(E, nxt, g1 - a0);
                val = f1 * (tx - t0) / (t1 - t0) + f0 * (t1 - tx) / (t1 - t0);
            } else
                val = eg_lerp_fast(x, g0, g1, f0, f1);
        }
    }
    *t_evf = pr1 * val;
    return c1;
}

// The same for tables that are too long for LDS (C4: 65 537 rows, C5: 4 x 32 769): every `stride`-th row of the M column
// is staged (a sampled index), the bracket search runs on the sample in LDS and is finished by log2(stride) steps in the
// window of the global column between two samples (one or two cache lines) -- 3-5 dependent global loads per evaluation
// instead of 2 x 16.  The column must be in order (k_sortcheck, once per table); `edge` holds M[1], M[2], M[len-2], M[len-1].
// Both searches in the branch-free form of eg_last_le: the same number of steps on every lane (ns and stride are uniform); the
// window's reads are the only global ones.
static __device__ __forceinline__ int eg_bracket_sampled(double x, const eg_ldsd *S, int ns, int stride, const double *M, int n1,
                                                         const eg_ldsd *edge)
{
    const int k = eg_last_le(x, S, ns);  // last sample not above x (S[0] = M[0] <= M[1] <= x on the lanes whose result is used)
    int base = k * stride;
    const int zl = max(min(base + stride, n1 - 2) - 1, 0);  // M[base] <= x < M[zl + 1]: the last row of [base, zl] that is not above x
    for (int len = stride; len > 1;) {
        const int half = len >> 1, c = min(base + half, zl);
        const double mj = M[c];
        base = (mj <= x) ? c : base;
        len -= half;
    }
    if (x != x) base = n1 - 3;  // (a NaN fails every comparison of the reference's bisection, which then runs up to row n-3)
    if (x < edge[0]) base = 0;
    if (x >= edge[2]) base = n1 - 2;
    return base;
}

static __device__ __forceinline__ double eg_term_sampled(const ms_env *E, const eg_ldsd *S, int ns, int stride, const eg_ldsd *edge,
                                                         const Tab &t, const ms_pv *cur, ms_pv *nxt, double pr1, double *t_rhs,
                                                         double *t_evf, int *ibr = nullptr)
{
    nxt->cash = ms_cashinhand(E, cur, nxt);
    const double x = nxt->cash;
    const int n1 = t.len;
    // (with a neighbour's bracket at hand: a step or two in the global column, the lines the neighbour just read)
    const int i = (ibr && *ibr >= 0) ? eg_bracket_near(x, t.M, n1, *ibr) : eg_bracket_sampled(x, S, ns, stride, t.M, n1, edge);
    if (ibr) *ibr = i;
    const double mlast = edge[3], mfirst = edge[0];
    // (rows i and i+1 of M, C and V in one round of global reads, see eg_term_lds)
    const double Mi = t.M[i], Mi1 = t.M[i + 1], Ci = t.C[i], Ci1 = t.C[i + 1], Vi = t.V[i], Vi1 = t.V[i + 1];
    double c1 = eg_lerp_fast(x, Mi, Mi1, Ci, Ci1);
    if (x > mlast) c1 = MS_MAX(c1, t.C[n1 - 1]);  // constant extrapolation, :554
    *t_rhs = 0;
    *t_evf = 0;
    if (c1 <= 0) return c1;
    if (!MS_OPTIM_MUNOD || (!MS_OPTIM_UNOD && x < mfirst))
        nxt->id = (int)t.D[eg_bracket(x, t.TH, t.thlen, 1)];  // optimd, egdst_lib.c:129-132
    else
        nxt->id = 0;
    *t_rhs = pr1 * ms_utility_marginal(E, nxt, c1) * ms_cashinhand_marginal(E, cur, nxt);
    const double evf1 = t.V[0], a0 = E->a0;
    double val;
    if (x < mfirst && evf1 > -INFINITY)
        val = ms_utility(E, nxt, x - a0) + ms_discount(E, nxt) * evf1;
    else {
        const int j = eg_second_bracket(x, i, edge[1], edge[2], n1);  // == eg_bracket(x, M + 1, n1 - 1, 0)
        const bool same = (j + 1 == i);
        const double f0 = same ? Vi : t.V[j + 1], f1 = same ? Vi1 : t.V[j + 2];
        if (!isfinite(f0))
            val = f0;
        else if (!isfinite(f1))
            val = f1;
        else {
            const double g0 = same ? Mi : t.M[j + 1], g1 = same ? Mi1 : t.M[j + 2];
            if (x > a0 && (x > mlast || x < mfirst)) {
                const double tx = ms_tr(E, nxt, x - a0), t0 = ms_tr(E, nxt, g0 - a0), t1 = ms_tr(E, nxt, g1 - a0);
                val = f1 * (tx - t0) / (t1 - t0) + f0 * (t1 - tx) / (t1 - t0);
            } else
                val = eg_lerp_fast(x, g0, g1, f0, f1);
        }
    }
    *t_evf = pr1 * val;
    return c1;
}

// Is the M column of every next-period table non-decreasing?  Once per period, for the handles whose tables do not fit
// k_grid_lds' LDS (the staged form checks the order while staging).  A table is spread over workgroups of SORTCHK_ROWS rows
// (one workgroup per table took 23 us per period on C4's 65 536 rows); a workgroup that finds a pair out of order -- or, the
// first one, a table too short or too long for the one-search form -- stamps the table with this period's mark; eg_tab_sorted
// reads "no stamp of this period" as "in order", so nothing is cleared between periods.
#define SORTCHK_ROWS (4 * GRID_BS)
__global__ void __launch_bounds__(GRID_BS) k_sortcheck(const Batch *bp_, int it)
{
    BatchRef b = EG_BATCH_REF(bp_);
    const int ist = blockIdx.y % MS_NST, draw = b.order[b.draw0 + blockIdx.y / MS_NST];
    const int slot1 = (b.g.nslots == 2) ? ((it + 1) & 1) : (it + 1);
    const Tab t = eg_tab(b, slot1, draw, ist);
    const int r0 = (int)blockIdx.x * SORTCHK_ROWS, r1 = min(t.len - 1, r0 + SORTCHK_ROWS);
    int bad = (blockIdx.x == 0 && threadIdx.x == 0 && !(t.len >= 4 && t.len <= b.g.Sp)) ? 1 : 0;
    if (t.len <= b.g.Sp)
        for (int r = r0 + (int)threadIdx.x; r < r1; r += GRID_BS)
            if (!(t.M[r] <= t.M[r + 1])) bad = 1;
    if (bad) b.tsorted[(size_t)draw * MS_NST + ist] = b.sorted_valid + it;
}

#ifndef GRID_MINW
#define GRID_MINW 1  // (experiments: waves per SIMD k_grid_lds is compiled for; 8 = at most 64 VGPRs)
#endif
// PPL: consecutive asset points per lane.  1: a lane per point.  More (batches with points to spare): the lane evaluates its
// points side by side -- (next state, shock node) outside, the points inside, every point with its own sums, so the order of
// accumulation per point is the reference's -- and a point starts its bracket search from its predecessor's bracket
// (eg_bracket_near): the verdict of round 2 on this kernel was "issue-bound on integer work", four of five issue slots
// address arithmetic and compares of searches that land a row or two from where the neighbouring point's search landed.
// CV: the C and V columns are staged too (whole columns only; the dynamic LDS then holds 3 x lrows doubles): no global read is
// left in the loop over (next state, shock node) -- k_grid_lds_cv, below
template <int PPL, bool CV = false> static __device__ __forceinline__ void eg_grid_lds_body(BatchRef b, int it, int lrows)
{
    EG_DYN_LDS(gl_dyn);                       // [lrows] staged M columns, consecutive by next state
    __shared__ int gl_off[MS_NST], gl_ok;     // first staged row of a next state (-1: not staged), staging succeeded
    __shared__ int gl_stride, gl_ns[MS_NST];  // every gl_stride-th row is staged (1: whole columns); staged entries per state
    __shared__ double gl_edge[MS_NST * 4];    // M[1], M[2], M[len-2], M[len-1] of every staged table
    __shared__ double gl_shock[MS_SHOCK_NODES_SHARED ? MS_NST * EG_GRID_NYMAX : 1];
    __shared__ int gl_niy[MS_NST];
    const int combo = blockIdx.y;
    const int id = combo % MS_ND, ist = (combo / MS_ND) % MS_NST, draw = b.order[b.draw0 + combo / (MS_ND * MS_NST)];
    const int n = (blockIdx.x * GRID_BS + threadIdx.x) * PPL + 1;  // the lane's first point
    if (b.status[draw]) return;
    const ProbeOut P = b.probe[((size_t)draw * MS_NST + ist) * MS_ND + id];
    if (!P.active || !P.grid) return;  // (uniform over the workgroup)
    ms_env E = eg_env(b, draw);
    ms_pv cur;
    cur.it = it;
    cur.ist = ist;
    cur.id = id;
    cur.cash = cur.savings = cur.shock = 0;
    const int slot1 = (b.g.nslots == 2) ? ((it + 1) & 1) : (it + 1);
    const int ny = b.g.ny;
    eg_ldsd *LM = (eg_ldsd *)gl_dyn;
    eg_ldsd *LC = LM + lrows, *LV = LM + 2 * (size_t)lrows;  // (CV only)
    // ---- stage: which next states, how many rows, are the columns in order -------------------------------------
    if (threadIdx.x == 0) {
        int tot = 0, ok = (ny <= EG_GRID_NYMAX || !MS_SHOCK_NODES_SHARED) ? 1 : 0;
        ms_pv nx;
        nx.it = it + 1, nx.id = 0, nx.cash = nx.shock = 0, nx.savings = 0;
        for (nx.ist = 0; nx.ist < MS_NST; nx.ist++) {
            gl_off[nx.ist] = -1;
            if (ms_feasible(&E, &nx) != 1) continue;
            const size_t k = ((size_t)slot1 * b.g.ndraw + draw) * MS_NST + nx.ist;
            const int len = b.tlen[k];
            if (len < 4 || len > b.g.Sp) {  // (short or missing tables: the general path reports them)
                ok = 0;
                break;
            }
            gl_off[nx.ist] = len;  // (rows for now; offsets below)
            tot += len;
        }
        int stride = 1;
        if (CV && ok && tot > lrows) ok = 0;  // (whole columns or the general path)
        if (ok && tot > lrows) {
            // too long: a sampled index of every column; needs the tables in order (k_sortcheck ran before this kernel)
            int nfe = 0;
            for (int s1 = 0; s1 < MS_NST; s1++) nfe += gl_off[s1] > 0;
            stride = (tot + lrows - nfe - 1) / max(lrows - nfe, 1);
            for (int s1 = 0; s1 < MS_NST && ok; s1++)
                if (gl_off[s1] > 0 && !eg_tab_sorted(b, it, draw, s1)) ok = 0;
        }
        int acc = 0;
        for (int s1 = 0; s1 < MS_NST; s1++) {
            const int len = gl_off[s1];
            if (len <= 0) continue;
            gl_ns[s1] = (len + stride - 1) / stride;
            gl_off[s1] = acc;
            acc += gl_ns[s1];
        }
        if (acc > lrows) ok = 0;
        gl_stride = stride;
        gl_ok = ok;
    }
    __syncthreads();
    bool fast = gl_ok != 0;
    const int stride = gl_stride;
    if (fast && stride > 1) {  // sampled index + the four edge values of every column (order checked by k_sortcheck)
        ms_pv nx;
        nx.it = it + 1, nx.id = 0, nx.cash = nx.shock = 0, nx.savings = 0;
        for (int s1 = 0; s1 < MS_NST; s1++) {
            const int off = gl_off[s1];
            if (off < 0) continue;
            const Tab t = eg_tab(b, slot1, draw, s1);
            for (int k = threadIdx.x; k < gl_ns[s1]; k += GRID_BS) LM[off + k] = t.M[(size_t)k * stride];
            if (threadIdx.x == 0) {
                gl_edge[4 * s1] = t.M[1], gl_edge[4 * s1 + 1] = t.M[2];
                gl_edge[4 * s1 + 2] = t.M[t.len - 2], gl_edge[4 * s1 + 3] = t.M[t.len - 1];
            }
            if (MS_SHOCK_NODES_SHARED) {
                nx.ist = s1;
                const int niy = (ms_sigma(&E, &cur, &nx) <= 0 || ny == 1) ? 1 : ny;
                if (threadIdx.x == 0) gl_niy[s1] = niy;
                for (int iy = threadIdx.x; iy < niy; iy += GRID_BS)
                    gl_shock[s1 * EG_GRID_NYMAX + iy] = (niy == 1) ? eg_shock_mean(&E, &cur, &nx) : eg_shock_node(&E, &cur, &nx, b.qz[iy]);
            }
        }
        __syncthreads();
    } else if (fast) {
        int bad = 0;
        ms_pv nx;
        nx.it = it + 1, nx.id = 0, nx.cash = nx.shock = 0, nx.savings = 0;
        for (int s1 = 0; s1 < MS_NST; s1++) {
            const int off = gl_off[s1];
            if (off < 0) continue;
            const Tab t = eg_tab(b, slot1, draw, s1);
            for (int r = threadIdx.x; r < t.len; r += GRID_BS) {
                const double m = t.M[r];
                LM[off + r] = m;
                if (CV) LC[off + r] = t.C[r], LV[off + r] = t.V[r];
                if (r + 1 < t.len && !(m <= t.M[r + 1])) bad = 1;  // (NaN counts as out of order)
            }
            if (MS_SHOCK_NODES_SHARED) {
                nx.ist = s1;
                const int niy = (ms_sigma(&E, &cur, &nx) <= 0 || ny == 1) ? 1 : ny;
                if (threadIdx.x == 0) gl_niy[s1] = niy;
                for (int iy = threadIdx.x; iy < niy; iy += GRID_BS)
                    gl_shock[s1 * EG_GRID_NYMAX + iy] = (niy == 1) ? eg_shock_mean(&E, &cur, &nx) : eg_shock_node(&E, &cur, &nx, b.qz[iy]);
            }
        }
        if (bad) gl_ok = 0;  // (benign race: every writer stores 0)
        __syncthreads();
        fast = gl_ok != 0;
    }
    if (n >= b.g.ngridm) return;
    GridLims L;
    L.lim1 = P.lim1, L.lim2 = P.lim2, L.lim3 = P.lim3, L.lim3p = P.lim3p, L.k3 = P.k3, L.ntogenerate = P.ntogenerate;
    const int np = min(PPL, b.g.ngridm - n);  // points of this lane
    double A[PPL];
    LaneEval r[PPL];
#pragma unroll
    for (int j = 0; j < PPL; j++) A[j] = (j < np) ? eg_grid_A(&E, &cur, L, 0, P.A0, n + j) : 0.0;
    if (!fast) {
#pragma unroll
        for (int j = 0; j < PPL; j++)
            if (j < np) r[j] = eg_lane_eval(b, &E, &cur, slot1, draw, A[j]);
    } else {  // eg_lane_eval with the staged columns, the lane's points side by side
        double rhs[PPL], evf[PPL], checksum[PPL], c1[PPL];
        int cnt[PPL], done[PPL];  // done: the point has left the loops (c1 <= 0, evf = -inf: :556,569,572; a hard error)
        int status[PPL], terr[PPL];
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            rhs[j] = evf[j] = checksum[j] = 0, c1[j] = 1.0, cnt[j] = 0, done[j] = (j < np) ? 0 : 1, status[j] = 0, terr[j] = 0;
            r[j].bist = 0, r[j].bshock = r[j].bcash = 0;
        }
        ms_pv nxt;
        nxt.it = it + 1;
        nxt.id = 0;
        nxt.cash = 0;
        nxt.shock = 0;
        nxt.savings = A[0];
        for (nxt.ist = 0; nxt.ist < MS_NST; nxt.ist++) {
            bool all_done = true;
#pragma unroll
            for (int j = 0; j < PPL; j++) all_done = all_done && done[j];
            if (all_done) break;
            if (ms_feasible(&E, &nxt) != 1) continue;
            // transition probability and number of nodes per point (they may depend on savings); a point whose probability
            // is zero skips the state (:501-508)
            double pr1pre[PPL];
            int niyj[PPL], niymax = 0;
#pragma unroll
            for (int j = 0; j < PPL; j++) {
                pr1pre[j] = 0, niyj[j] = 0;
                if (done[j]) continue;
                nxt.savings = A[j];
                if (MS_OPTIM_TRPRNOSH) {
                    pr1pre[j] = ms_trpr(&E, &cur, &nxt, &terr[j]);
                    if (pr1pre[j] == 0.0) continue;
                }
                niyj[j] = MS_SHOCK_NODES_SHARED ? gl_niy[nxt.ist] : ((ms_sigma(&E, &cur, &nxt) <= 0 || ny == 1) ? 1 : ny);
                niymax = max(niymax, niyj[j]);
            }
            if (niymax == 0) continue;
            const Tab t = eg_tab(b, slot1, draw, nxt.ist);
            if (t.thlen > b.g.nthrhmax || t.thlen < 1) {  // (the points that reach this state stop here)
#pragma unroll
                for (int j = 0; j < PPL; j++)
                    if (niyj[j] > 0) status[j] = -2707, done[j] = 1;
                continue;
            }
            const eg_ldsd *M = LM + gl_off[nxt.ist];
            const int ns = gl_ns[nxt.ist];
            const eg_ldsd *edge = (const eg_ldsd *)gl_edge + 4 * nxt.ist;
            for (int iy = 0; iy < niymax; iy++) {
                int hint = -1;  // the bracket of the previous point of this lane for this (state, node)
#pragma unroll
                for (int j = 0; j < PPL; j++) {
                    if (done[j] || iy >= niyj[j]) continue;
                    nxt.savings = A[j];
                    if (MS_SHOCK_NODES_SHARED)
                        nxt.shock = gl_shock[nxt.ist * EG_GRID_NYMAX + iy];
                    else
                        nxt.shock = (niyj[j] == 1) ? eg_shock_mean(&E, &cur, &nxt) : eg_shock_node(&E, &cur, &nxt, b.qz[iy]);
                    double pr1 = MS_OPTIM_TRPRNOSH ? pr1pre[j] : ms_trpr(&E, &cur, &nxt, &terr[j]);
                    if (niyj[j] != 1) pr1 *= b.qw[iy];
                    if (pr1 == 0.0) continue;
                    checksum[j] += pr1;
                    cnt[j]++;
                    double t_rhs, t_evf;
                    if (CV) {
                        TabL tl;
                        tl.M = M, tl.C = (const eg_ldsd *)LC + gl_off[nxt.ist], tl.V = (const eg_ldsd *)LV + gl_off[nxt.ist];
                        tl.TH = t.TH, tl.D = t.D, tl.len = t.len, tl.thlen = t.thlen;
                        c1[j] = eg_term_lds(&E, M, tl, &cur, &nxt, pr1, &t_rhs, &t_evf, PPL > 1 ? &hint : nullptr, 1);
                    } else
                    c1[j] = (stride > 1) ? eg_term_sampled(&E, M, ns, stride, edge, t, &cur, &nxt, pr1, &t_rhs, &t_evf, PPL > 1 ? &hint : nullptr)
                                         : eg_term_lds(&E, M, t, &cur, &nxt, pr1, &t_rhs, &t_evf, PPL > 1 ? &hint : nullptr, 1);
                    if (c1[j] > 0) {
                        rhs[j] += t_rhs;
                        evf[j] += t_evf;
                    }
                    if (c1[j] <= 0 || evf[j] == -INFINITY) {
                        done[j] = 1;
                        r[j].bist = nxt.ist;
                        r[j].bshock = nxt.shock;
                        r[j].bcash = nxt.cash;
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            if (j >= np) continue;
            int st = status[j];
            if (terr[j]) st = -25;
            if (st == 0) {
                if (c1[j] <= 0)
                    st = 1;
                else if (evf[j] == -INFINITY)
                    st = 2;
                else if (fabs(checksum[j] - 1) > EG_TOL)
                    st = -11;
            }
            r[j].status = st;
            r[j].cnt = cnt[j];
            if (st == 0) {
                const double rr = rhs[j] * ms_discount(&E, &cur);
                r[j].M = A[j] + ms_utility_marginal_inverse(&E, &cur, rr);
                r[j].C = r[j].M - A[j];
                r[j].V = ms_utility(&E, &cur, r[j].C) + ms_discount(&E, &cur) * evf[j];
                r[j].R = r[j].M;
            } else {
                r[j].M = NAN;
                r[j].C = r[j].V = 0;
                r[j].R = (st == 1) ? b.g.a0 - 1 : r[j].bcash;
            }
        }
    }
    const size_t o = eg_cand(b, draw, ist, id) + n;
#pragma unroll
    for (int j = 0; j < PPL; j++) {
        if (j >= np) continue;
        if (r[j].status == 1) b.negflag[((size_t)draw * MS_NST + ist) * MS_ND + id] = 1;  // (rare; k_fixup_scan looks closer)
        b.cSt[o + j] = eg_sc_pack(r[j].status, r[j].cnt);
        b.cM[o + j] = r[j].R;  // (== r[j].M for a normal point, see eg_sc_pack)
        if (r[j].status == 0) {
            b.cC[o + j] = r[j].C;
            b.cV[o + j] = r[j].V;
        }
    }
}

#ifndef GRID_PPL
#define GRID_PPL 4  // asset points per lane of k_grid_lds_n (the form for batches with points to spare)
#endif
__global__ void __launch_bounds__(GRID_BS, GRID_MINW) k_grid_lds(const Batch *bp_, int it, int lrows)
{
    eg_grid_lds_body<1>(EG_BATCH_REF(bp_), it, lrows);
}
// Whole tables in LDS -- M, C and V, 24 B per row -- for batches whose tables fit: the loop over (next state, shock node) then
// reads nothing from global memory (four scattered 8-byte reads per term in k_grid_lds, the C and V of the bracket's two rows).
__global__ void __launch_bounds__(GRID_BS, GRID_MINW) k_grid_lds_cv(const Batch *bp_, int it, int lrows)
{
    eg_grid_lds_body<1, true>(EG_BATCH_REF(bp_), it, lrows);
}
#ifdef EGDST_WITH_GRID_PPL  // diagnostic builds only (tests/diag/gpu_grid_ppl.py): measured not faster in round 3
#ifndef GRID_N_MINW
#define GRID_N_MINW 1  // (the lane's points side by side want registers: 133 VGPRs by default, three waves per SIMD)
#endif
__global__ void __launch_bounds__(GRID_BS, GRID_N_MINW) k_grid_lds_n(const Batch *bp_, int it, int lrows)
{
    eg_grid_lds_body<GRID_PPL>(EG_BATCH_REF(bp_), it, lrows);
}
#endif

// k_grid for small batches: 16 lanes per grid point, a lane per shock node (eg_wave_expectation with groups of 16),
// so that a solve that leaves the GPU mostly idle does not spend 20 serial shock terms per point: same arithmetic and
// the same order of accumulation as eg_lane_eval, 62 -> 20 us per period on a single C2 draw.
#ifdef EGDST_EMU
#define EG_GW WAVE        // the harness cannot diverge inside a wave: one group per wave
#define EG_GRIDW_BS WAVE
#else
#define EG_GW 16
#define EG_GRIDW_BS 256
#endif
__global__ void __launch_bounds__(EG_GRIDW_BS) k_grid_wide(const Batch *bp_, int it)
{
    BatchRef b = EG_BATCH_REF(bp_);
    const int combo = blockIdx.y;
    const int id = combo % MS_ND, ist = (combo / MS_ND) % MS_NST, draw = b.order[b.draw0 + combo / (MS_ND * MS_NST)];
    const int n = (blockIdx.x * EG_GRIDW_BS + (int)threadIdx.x) / EG_GW + 1;  // the group's grid point
    if (b.status[draw]) return;
    const ProbeOut P = b.probe[((size_t)draw * MS_NST + ist) * MS_ND + id];
    if (!P.active || !P.grid || n >= b.g.ngridm) return;  // (whole groups leave together)
    ms_env E = eg_env(b, draw);
    ms_pv cur;
    cur.it = it;
    cur.ist = ist;
    cur.id = id;
    cur.cash = cur.savings = cur.shock = 0;
    GridLims L;
    L.lim1 = P.lim1, L.lim2 = P.lim2, L.lim3 = P.lim3, L.lim3p = P.lim3p, L.k3 = P.k3, L.ntogenerate = P.ntogenerate;
    const double A = eg_grid_A(&E, &cur, L, 0, P.A0, n);
    const int slot1 = (b.g.nslots == 2) ? ((it + 1) & 1) : (it + 1);
    double rhs = 0, evf = 0, bshock = 0, bcash = 0;
    int cnt = 0, bist = 0;
    const int st = eg_wave_expectation<Tab, EG_GW>(b, &E, slot1, draw, &cur, A, 1, &rhs, &evf, &cnt, &bist, &bshock, &bcash,
                                                   (const Tab *)nullptr);
    if ((threadIdx.x & (EG_GW - 1)) != 0) return;
    LaneEval r;  // what eg_lane_eval reports for the point
    r.status = st;
    r.cnt = cnt;
    if (st == 0) {
        rhs *= ms_discount(&E, &cur);
        r.M = A + ms_utility_marginal_inverse(&E, &cur, rhs);
        r.C = r.M - A;
        r.V = ms_utility(&E, &cur, r.C) + ms_discount(&E, &cur) * evf;
        r.R = r.M;
    } else {
        r.M = NAN;
        r.C = r.V = 0;
        r.R = (st == 1) ? b.g.a0 - 1 : bcash;
    }
    const size_t o = eg_cand(b, draw, ist, id) + n;
    if (r.status == 1) b.negflag[((size_t)draw * MS_NST + ist) * MS_ND + id] = 1;
    b.cSt[o] = eg_sc_pack(r.status, r.cnt);
    b.cM[o] = r.R;  // (== r.M for a normal point, see eg_sc_pack)
    if (r.status == 0) {
        b.cC[o] = r.C;
        b.cV[o] = r.V;
    }
}

// ---------------------------------------------------------------------------------------------
// block-wide helpers (ENV_BS threads = ENV_BS/WAVE waves): wave-level ballot/shuffle first, one LDS exchange
// across waves, two barriers per call
#define ENV_NW (ENV_BS / WAVE)
static __device__ __forceinline__ int blk_min(int v, int *sh)
{
    for (int o = WAVE / 2; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & (WAVE - 1)) == 0) sh[threadIdx.x / WAVE] = v;
    __syncthreads();
    int r = sh[0];
    for (int w = 1; w < ENV_NW; w++) r = min(r, sh[w]);
    return r;
}
static __device__ __forceinline__ int blk_sum(int v, int *sh)
{
    for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & (WAVE - 1)) == 0) sh[threadIdx.x / WAVE] = v;
    __syncthreads();
    int r = 0;
    for (int w = 0; w < ENV_NW; w++) r += sh[w];
    return r;
}
// exclusive scan of one 0/1 flag per thread; returns the exclusive prefix, *total the block total
static __device__ __forceinline__ int blk_scan(int flag, int *sh, int *total)
{
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const unsigned long long mask = __ballot(flag != 0);
    const int ex = __popcll(mask & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) sh[wave] = __popcll(mask);
    __syncthreads();
    int off = 0, tot = 0;
    for (int w = 0; w < ENV_NW; w++) {
        if (w < wave) off += sh[w];
        tot += sh[w];
    }
    *total = tot;
    return off + ex;
}

// exclusive scan of one small count per thread; returns the exclusive prefix, *total the block total
static __device__ __forceinline__ int blk_scan_int(int v, int *sh, int *total)
{
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    int incl = v;
    for (int o = 1; o < WAVE; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    __syncthreads();
    if (lane == WAVE - 1) sh[wave] = incl;
    __syncthreads();
    int off = 0, tot = 0;
    for (int w = 0; w < ENV_NW; w++) {
        if (w < wave) off += sh[w];
        tot += sh[w];
    }
    *total = tot;
    return off + incl - v;
}
// three reductions with one exchange: max of *a, bitwise or of *b, sum of *c
static __device__ __forceinline__ void blk_reduce3(int *a, int *b, int *c, int *sh)
{
    int x = *a, y = *b, z = *c;
    for (int o = WAVE / 2; o > 0; o >>= 1) {
        x = max(x, __shfl_xor(x, o));
        y |= __shfl_xor(y, o);
        z += __shfl_xor(z, o);
    }
    __syncthreads();
    if ((threadIdx.x & (WAVE - 1)) == 0) {
        const int w = threadIdx.x / WAVE;
        sh[3 * w] = x, sh[3 * w + 1] = y, sh[3 * w + 2] = z;
    }
    __syncthreads();
    x = sh[0], y = sh[1], z = sh[2];
    for (int w = 1; w < ENV_NW; w++) x = max(x, sh[3 * w]), y |= sh[3 * w + 1], z += sh[3 * w + 2];
    *a = x, *b = y, *c = z;
}

#ifdef EGDST_RANKCHK
__device__ int g_rankchk;
#endif
// strict order of comp1 (egdst_solver.c:1570-1582) extended by the original index (qsort of glibc is a
// stable merge sort, so fully tied quadruples keep their input order)
static __device__ __forceinline__ bool pt_before(double am, double av, int af, int ai, double bm, double bv, int bf, int bi)
{
    if (am != bm) return am < bm;
    if (av != bv) return av > bv;
    if (af != bf) return af < bf;
    return ai < bi;
}

// Rank of point i = (m, v, f) among npts points of nf functions whose lists are in comp1 order: the sum over the
// functions of "how many of its points precede this one".  The same counts give the walk's class word of the point
// (env_preclass): evaluated right here when cls_on.  KP: pointer type of the keys (LDS-staged or global).
// A thread ranks ENV_RK CONSECUTIVE points (i0 .. i0+n-1).  Consecutive points of one list are neighbours in every other
// list as well -- the count for point i+1 is at least the count for point i, and in merged lists of similar density a
// position or two more -- so only a run's first point pays for a binary search over the whole of g; the others gallop on
// from their predecessor's count.  (The count itself is unique: the lists are in comp1 order, checked by the caller.)
// pt_before for a key that lives in two arrays (am: its M key, already read): the V key is read only when the M keys tie
template <class KPV>
static __device__ __forceinline__ bool eg_key_before_am(double am, KPV Kv, int k, int g, double m, double v, int f, int i)
{
    if (am != m) return am < m;
    const double av = Kv[k];
    if (av != v) return av > v;
    if (g != f) return g < f;
    return k < i;
}
#ifndef ENV_RK
#define ENV_RK 1  // (measured on C2, sort phase of one solve: 1: 1.94 ms, 2: 2.15 ms, 4: 2.65 ms -- see DESIGN.md)
#endif
template <class KP, class KPV, class ANA>
static __device__ __forceinline__ void eg_rank_classify_run(int i0, int n, int nf, KP Km, KPV Kv, const int *f, const double *m,
                                                           const double *v, const eg_ldsi *fstart, const eg_ldsi *dims, bool cls_on,
                                                           double kbound, ANA ana, int *r, int *w, const eg_ldsd *S = nullptr,
                                                           int sk = 0)
{
    // S, sk: a sampled index of the keys in LDS (S[j] = Km[j * sk]) for streams whose keys do not fit (C5 at full size:
    // 65 536 points and more): the search runs on the sample first and is finished in the window between two samples, two
    // or three dependent global reads instead of seventeen
    bool force[ENV_RK];
#pragma unroll
    for (int k = 0; k < ENV_RK; k++) r[k] = 0, w[k] = 0, force[k] = false;
    for (int g = 0; g < nf; g++) {
        const int dg = dims[g];
        if (dg <= 0) continue;
        const int s0 = fstart[g];
        int prev = -1;  // count of the previous point when it belongs to the same list (and that list is not g)
#pragma unroll
        for (int k = 0; k < ENV_RK; k++) {
            if (k >= n) continue;
            const int i = i0 + k, fk = f[k];
            const double mk = m[k], vk = v[k];
            if (fk == g) {
                r[k] += i - s0;
                prev = -1;
                continue;
            }
            int lo, hi;  // first position in [lo, hi] that does not precede the point (positions below lo do, hi does not or is dg)
            if (prev >= 0 && k > 0 && f[k > 0 ? k - 1 : 0] == fk) {
                lo = prev, hi = prev;
                int step = 1;
                for (;;) {
                    if (hi >= dg) {
                        hi = dg;
                        break;
                    }
                    if (!eg_key_before_am(Km[s0 + hi], Kv, s0 + hi, g, mk, vk, fk, i)) break;
                    lo = hi + 1;
                    hi += step;
                    step <<= 1;
                }
            } else {
                // most lists lie entirely on one side of the point (pieces of a folded choice list overlap only near the
                // kinks), which two or three key reads settle; the last point of a closed piece is its extrapolation point
                lo = 0, hi = 0;
                if (eg_key_before_am(Km[s0], Kv, s0, g, mk, vk, fk, i)) {
                    const bool b1 = eg_key_before_am(Km[s0 + dg - 1], Kv, s0 + dg - 1, g, mk, vk, fk, i);
                    const bool b2 = !b1 && dg >= 2 && eg_key_before_am(Km[s0 + dg - 2], Kv, s0 + dg - 2, g, mk, vk, fk, i);
                    // (selects on purpose: an if / else-if / else chain here was miscompiled by hipcc 7.2 for gfx950 in an
                    //  earlier form of this function -- the last branch set hi but not lo; -DEGDST_RANKCHK checks every rank)
                    lo = b1 ? dg : (b2 ? dg - 1 : 1);
                    hi = b1 ? dg : (b2 ? dg - 1 : (dg >= 2 ? dg - 2 : dg - 1));
                }
            }
            if (sk > 1 && hi - lo > sk) {
                // positions below a = s0+lo precede the point, position z = s0+hi does not (or is the end of g); the sampled
                // positions in [a, z) are ja*sk .. jz*sk: the first of them that does not precede the point bounds the window
                const int a = s0 + lo, z = s0 + hi, ja = (a + sk - 1) / sk, jz = (z - 1) / sk;
                int l = ja, h = jz + 1;
                while (l < h) {
                    const int mid = (l + h) >> 1;
                    if (eg_key_before_am(S[mid], Kv, mid * sk, g, mk, vk, fk, i))
                        l = mid + 1;
                    else
                        h = mid;
                }
                if (l > ja) lo = (l - 1) * sk + 1 - s0;
                if (l <= jz) hi = l * sk - s0;
            }
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (eg_key_before_am(Km[s0 + mid], Kv, s0 + mid, g, mk, vk, fk, i))
                    lo = mid + 1;
                else
                    hi = mid;
            }
            prev = lo;
            r[k] += lo;
            if (cls_on && !force[k]) {
                if (lo >= dg)
                    force[k] = true;  // g has no point ahead: cannot happen below the bound
                else {
                    double t;
                    if (lo >= 1) {  // env_fn_cnt on the keys: the segment between g's points lo-1 and lo
                        const double ga = Km[s0 + lo - 1], gb = Km[s0 + lo], fa = Kv[s0 + lo - 1], fb = Kv[s0 + lo];
                        if (mk == ga)
                            t = fa;
                        else if (mk < ga || mk > gb)
                            t = -INFINITY;
                        else
                            t = fb * (mk - ga) / (gb - ga) + fa * (gb - mk) / (gb - ga);
                    } else
                        t = ana(g, mk);
                    if (vk < t) w[k] |= 1;
                    if (t < vk && nf <= 29) w[k] |= (2 << g);
                }
            }
        }
    }
    if (cls_on) {
#pragma unroll
        for (int k = 0; k < ENV_RK; k++) {
            if (force[k] || !(m[k] < kbound))
                w[k] = ENV_CLS_FORCE;
            else if (nf > 29)
                w[k] |= ENV_CLS_NOMASK;
        }
    }
}

// Exactly two non-empty lists (the primary envelope of a two-choice model, a choice list with one fold): a MERGE PATH instead
// of one binary search per point.  Thread t owns the sorted positions [t*C, (t+1)*C): one binary search along its diagonal
// finds how many points of either list precede them, then it merges its C points one after the other, which reads each key
// once and in order (C2's 2000-point primary: 11 + 4 dependent LDS reads per thread instead of 4 x 14; C5's 65 536-point
// streams in global memory: 16 + 128 mostly sequential reads instead of 128 x 6 dependent round trips).  At every point the
// merge knows how many points of the OTHER list precede it and holds that list's neighbouring pair of keys, which is all the
// class word needs (same arithmetic as eg_rank_classify_run).  emit(r, i, f, m, v, w): sorted position, input index,
// function, keys, class word.
template <class KP, class KPV, class ANA, class EMIT>
static __device__ __forceinline__ void eg_merge2_classify(int npts, int nf, int ga, int sa, int na, int gb, int sb, int nb, KP Km, KPV Kv,
                                                         bool cls_on, double kbound, ANA ana, EMIT emit)
{
    const int C = (npts + ENV_BS - 1) / ENV_BS;
    const int d0 = min(npts, (int)threadIdx.x * C), d1 = min(npts, d0 + C);
    if (d0 >= d1) return;
    int lo = max(0, d0 - nb), hi = min(d0, na);
    while (lo < hi) {  // fewest points of A among the first d0 such that A[lo] does not precede B[d0-1-lo]
        const int mid = (lo + hi) >> 1, ia = sa + mid, ib = sb + d0 - 1 - mid;
        if (pt_before(Km[ia], Kv[ia], ga, ia, Km[ib], Kv[ib], gb, ib))
            lo = mid + 1;
        else
            hi = mid;
    }
    int a = lo, b = d0 - lo;
    double ma = 0, va = 0, mb = 0, vb = 0, pma = 0, pva = 0, pmb = 0, pvb = 0;  // current and previous keys of the lists
    if (a < na) ma = Km[sa + a], va = Kv[sa + a];
    if (b < nb) mb = Km[sb + b], vb = Kv[sb + b];
    if (a > 0) pma = Km[sa + a - 1], pva = Kv[sa + a - 1];
    if (b > 0) pmb = Km[sb + b - 1], pvb = Kv[sb + b - 1];
    for (int r = d0; r < d1; r++) {
        const bool takeA = (b >= nb) || (a < na && pt_before(ma, va, ga, sa + a, mb, vb, gb, sb + b));
        // the point, and the other list: its function, how many of its points precede the point, its length, its pair of keys
        const double m = takeA ? ma : mb, v = takeA ? va : vb;
        const int f = takeA ? ga : gb, i = takeA ? sa + a : sb + b, g = takeA ? gb : ga;
        const int cnt = takeA ? b : a, dg = takeA ? nb : na;
        const double k0m = takeA ? pmb : pma, k0v = takeA ? pvb : pva, k1m = takeA ? mb : ma, k1v = takeA ? vb : va;
        int w = 0;
        if (cls_on) {
            bool force = false;
            if (cnt >= dg)
                force = true;  // g has no point ahead: cannot happen below the bound
            else {
                double t;
                if (cnt >= 1) {  // env_fn_cnt on the keys: the segment between g's points cnt-1 and cnt
                    if (m == k0m)
                        t = k0v;
                    else if (m < k0m || m > k1m)
                        t = -INFINITY;
                    else
                        t = k1v * (m - k0m) / (k1m - k0m) + k0v * (k1m - m) / (k1m - k0m);
                } else
                    t = ana(g, m);
                if (v < t) w |= 1;
                if (t < v && nf <= 29) w |= (2 << g);
            }
            if (force || !(m < kbound))
                w = ENV_CLS_FORCE;
            else if (nf > 29)
                w |= ENV_CLS_NOMASK;
        }
        emit(r, i, f, m, v, w);
        if (takeA) {
            pma = ma, pva = va;
            a++;
            if (a < na) ma = Km[sa + a], va = Kv[sa + a];
        } else {
            pmb = mb, pvb = vb;
            b++;
            if (b < nb) mb = Km[sb + b], vb = Kv[sb + b];
        }
    }
}

// Sort npts points of nf functions (function f occupies [fstart[f], fstart[f]+dims[f]) of the input) into
// (om,oc,ov,of) and record rank[].  Each function's list is normally already ordered, so the rank of a point
// is a sum of binary searches (a merge); an unordered list falls back to counting.
template <class ANA>
static __device__ __forceinline__ void blk_rank_sort(int npts, int nf, const double *im, const double *ic, const double *iv, const int *ifn,
                                     const eg_ldsi *fstart, const eg_ldsi *dims, double *om, double *oc, double *ov, int *of,
                                     int *rank, int *sh, int *oob, int *dbg, int *cls, int *fused, ANA ana, eg_ldsd *lkeys,
                                     int lkeys_cap, eg_ldss *lperm = nullptr, eg_ldsi *lcls = nullptr, int lperm_cap = 0)
{
    // The sorted stream is written in TWO steps.  A point's rank is where it goes, and neighbouring threads hold neighbouring
    // points of ONE list, whose ranks are interleaved with the other lists' -- written straight to (om, oc, ov, of, cls)[rank]
    // every store instruction of a wave fills half of each cache line it touches, the other half arrives from another wave
    // much later, and with thousands of workgroups in flight the half-written lines are evicted in between: measured on
    // C2 x 4096 (profiles/r03_a_*), the throughput path's sort ran 3x SLOWER with seven workgroups per CU than with one.
    // So step one scatters only the inverse permutation perm[rank] = input index (and the class word) -- in LDS when the
    // caller has room (lperm, lcls: npts <= lperm_cap < 65536), else into `of` (one 4-byte array instead of six arrays) --
    // and step two walks the sorted positions in order: coalesced stores, gathered loads from lists that are read nearly
    // in sequence.  rank[i] itself is indexed by the input and always written coalesced.
    const bool perm_lds = lperm != nullptr && npts <= lperm_cap && npts <= lkeys_cap && npts < 65536;  // (lcls holds lkeys_cap words)
    int Pnet = 1;  // a bitonic network over the inverse permutation needs a power of two of entries
    while (Pnet < npts) Pnet <<= 1;
    const bool net_ok = perm_lds && Pnet <= lperm_cap;
    // lkeys: the workgroup's dynamic LDS, lkeys_cap doubles.  A stream that is too long to be sorted and walked in LDS
    // (32 B per point) often still fits with its M keys alone (8 B per point: C3's 12 000-point primary stream, 96 KB):
    // the binary searches of the rank merge then run on LDS, and the V keys are read from global memory only at M ties
    // and at the bracket of the classification.
#ifdef EGDST_TPSTAMPS  // diagnostic: where a global-memory sort spends its time (dbg as 8 x u64: 0 staging + order check, 1 presort
    unsigned long long tp_t_ = wall_clock64();  // of lists out of order, 2 ranks, 3 tail; 4 sorts, 5 sorts with a list out of order)
#define TPST(k) do { __syncthreads(); if (threadIdx.x == 0 && dbg) { const unsigned long long n_ = wall_clock64(); atomicAdd((unsigned long long *)dbg + (k), n_ - tp_t_); tp_t_ = n_; } } while (0)
#define TPCNT(k, v) do { if (threadIdx.x == 0 && dbg) atomicAdd((unsigned long long *)dbg + (k), (unsigned long long)(v)); } while (0)
#else
#define TPST(k)
#define TPCNT(k, v)
#endif
    const bool lds_keys = lkeys != nullptr && npts <= lkeys_cap;
    // (longer still: every sk-th key, a sampled index -- see eg_rank_classify_run)
    const int sk = (lkeys != nullptr && !lds_keys && lkeys_cap >= 64) ? (npts + lkeys_cap - 1) / lkeys_cap : 0;
    if (lds_keys) {
        for (int i = threadIdx.x; i < npts; i += ENV_BS) lkeys[i] = im[i];
        __syncthreads();
    } else if (sk > 1) {
        for (int j = threadIdx.x; j * sk < npts; j += ENV_BS) lkeys[j] = im[(size_t)j * sk];
        __syncthreads();
    }
    int bad = 0;
    for (int i = threadIdx.x + 1; i < npts; i += ENV_BS)
        if (ifn[i] == ifn[i - 1] && !pt_before(im[i - 1], iv[i - 1], ifn[i - 1], i - 1, im[i], iv[i], ifn[i], i)) bad = 1;
    bad = blk_sum(bad, sh);
    EG_CENSUS_BAD(bad ? 1 : 0);
    TPST(0);
    TPCNT(4, 1);
    TPCNT(5, bad ? 1 : 0);
    if (bad) {
        // A list out of comp1 order is out of order LOCALLY: the double point of a kink (x + 1e-10) has overtaken the next
        // grid point or two -- on fine grids (C5: 32 768 points, spacing below 1e-10 near a0) that happens in many periods,
        // and the counting fallback below is quadratic (65 536 points: 1.5 s per cell, measured).  A few rounds of odd-even
        // transposition WITHIN the lists put them in order first: neighbours of the same function are swapped only when
        // strictly out of order, so fully tied points keep their input order (what the reference's stable qsort gives), and
        // the final sorted stream is the same.  (The input arrays are the kernel's own work arrays.)
        double *wm = (double *)im, *wc = (double *)ic, *wv = (double *)iv;
        // (with the LDS network below at hand, a few rounds: what they do not repair is not a local disorder)
        const int max_rounds = net_ok ? 8 : 64;
        for (int round = 0; round < max_rounds && bad; round += 2) {
            for (int par = 0; par < 2; par++) {
                for (int i = 2 * (int)threadIdx.x + par; i + 1 < npts; i += 2 * ENV_BS) {
                    if (ifn[i] != ifn[i + 1]) continue;
                    const double am = wm[i], av = wv[i], bm = wm[i + 1], bv = wv[i + 1];
                    if (bm < am || (bm == am && bv > av)) {  // strictly before in (M asc, V desc): swap the two points
                        const double ac = wc[i];
                        wm[i] = bm, wv[i] = bv, wc[i] = wc[i + 1];
                        wm[i + 1] = am, wv[i + 1] = av, wc[i + 1] = ac;
                    }
                }
                __syncthreads();
            }
            int b2 = 0;
            for (int i = threadIdx.x + 1; i < npts; i += ENV_BS)
                if (ifn[i] == ifn[i - 1] && !pt_before(wm[i - 1], wv[i - 1], ifn[i - 1], i - 1, wm[i], wv[i], ifn[i], i)) b2 = 1;
            bad = blk_sum(b2, sh);
        }
#ifdef EGDST_EMU
        if (threadIdx.x == 0 && getenv("EGDST_TRACE_BAD")) {
            int nb = 0, fb = -1;
            for (int i = 1; i < npts; i++)
                if (ifn[i] == ifn[i - 1] && !pt_before(wm[i - 1], wv[i - 1], ifn[i - 1], i - 1, wm[i], wv[i], ifn[i], i)) {
                    nb++;
                    if (fb < 0) fb = i;
                }
            fprintf(stderr, "sort: lists out of order npts=%d nf=%d still_bad=%d (pairs %d)", npts, nf, bad, nb);
            if (fb >= 0)
                fprintf(stderr, " first at %d f=%d: (%.17g, %.17g) then (%.17g, %.17g)", fb, ifn[fb], wm[fb - 1], wv[fb - 1], wm[fb], wv[fb]);
            fprintf(stderr, "\n");
        }
#endif
        if (lds_keys) {  // (the staged M keys follow the lists)
            __syncthreads();
            for (int i = threadIdx.x; i < npts; i += ENV_BS) lkeys[i] = im[i];
            __syncthreads();
        } else if (sk > 1) {
            __syncthreads();
            for (int j = threadIdx.x; j * sk < npts; j += ENV_BS) lkeys[j] = im[(size_t)j * sk];
            __syncthreads();
        }
    }
    // Still out of order: a list that is not merely locally disordered -- a guess stream that re-based (k_fixup) runs up to
    // mmax, drops back and climbs again, and the reference hands such lists to qsort like any other (egdst_solver.c:1240).
    // Counting ranks (below) is quadratic: 2.8 ms for a 2000-point stream, which 0.4 % of the C2 cells need -- one in nearly
    // every launch of a few hundred cells, so that EVERY launch lasted 2.8 ms (measured, profiles/r03_*).  With the inverse
    // permutation in LDS the whole stream goes through a bitonic network on the input indices instead (keys looked up in
    // LDS, V keys in global memory only at M ties): a fraction of a millisecond.
    bool networked = false;
    if (bad && net_ok) {
        const int P = Pnet;
        {
            for (int i = threadIdx.x; i < P; i += ENV_BS) lperm[i] = (unsigned short)(i < npts ? i : 0xffff);  // padding sorts last
            __syncthreads();
            for (int k = 2; k <= P; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = threadIdx.x; i < P; i += ENV_BS) {
                        const int q = i ^ j;
                        if (q > i) {
                            const int a = lperm[i], z = lperm[q];
                            bool zfirst;  // does the entry at q precede the entry at i?
                            if (a == 0xffff)
                                zfirst = (z != 0xffff);
                            else if (z == 0xffff)
                                zfirst = false;
                            else {
                                // (comp1 by its keys in turn: the M keys sit in LDS, the rest -- V, the function -- is
                                //  fetched from global memory only when the M keys tie, which is rare)
                                const double mz = lds_keys ? (double)lkeys[z] : im[z], ma = lds_keys ? (double)lkeys[a] : im[a];
                                if (mz != ma)
                                    zfirst = mz < ma;
                                else
                                    zfirst = pt_before(mz, iv[z], ifn[z], z, ma, iv[a], ifn[a], a);
                            }
                            if (((i & k) == 0) == zfirst) lperm[i] = (unsigned short)z, lperm[q] = (unsigned short)a;
                        }
                    }
                    __syncthreads();
                }
            networked = true;
        }
    }
    TPST(1);
    double kbound = INFINITY;  // min over the functions of their last grid value (:1266-1271)
    *fused = 0;
    if (!bad && cls) {
        for (int g = 0; g < nf; g++)
            if (dims[g] > 0) {
                const double last = im[fstart[g] + dims[g] - 1];
                if (last < kbound) kbound = last;
            }
        *fused = 1;
    }
    // two non-empty lists: merge path (eg_merge2_classify); otherwise one search per point and foreign list
    int nact = 0, ga = -1, gb = -1;
    for (int g = 0; g < nf; g++)
        if (dims[g] > 0) {
            if (nact == 0) ga = g;
            if (nact == 1) gb = g;
            nact++;
        }
#ifndef EGDST_MERGE2_GLOBAL  // (measured on C5 x 128: 3.95 s with the merge path in global memory against 3.69 s without -- a
    nact = 0;                // thread's 128 dependent steps cost more there than its 128 searches; LDS-resident streams only)
#endif
    if (!bad && nact == 2) {
        auto emit = [&](int r, int i, int f, double m, double v, int w) {
            rank[i] = r;
            if (perm_lds) {
                lperm[r] = (unsigned short)i;
                if (cls) lcls[r] = w;
            } else {
                of[r] = i;
                if (cls) cls[r] = w;
            }
        };
        if (lds_keys)
            eg_merge2_classify(npts, nf, ga, (int)fstart[ga], (int)dims[ga], gb, (int)fstart[gb], (int)dims[gb], (const eg_ldsd *)lkeys, iv,
                               cls != nullptr, kbound, ana, emit);
        else
            eg_merge2_classify(npts, nf, ga, (int)fstart[ga], (int)dims[ga], gb, (int)fstart[gb], (int)dims[gb], im, iv, cls != nullptr,
                               kbound, ana, emit);
    }
    for (int i0 = ENV_RK * (int)threadIdx.x; i0 < npts && !bad && nact != 2; i0 += ENV_RK * ENV_BS) {  // a run of consecutive points per round
        double m[ENV_RK], v[ENV_RK];
        int f[ENV_RK], r[ENV_RK], w[ENV_RK];
        const int n = min(ENV_RK, npts - i0);
#pragma unroll
        for (int k = 0; k < ENV_RK; k++) {
            m[k] = v[k] = 0, f[k] = 0;
            if (k < n) m[k] = im[i0 + k], v[k] = iv[i0 + k], f[k] = ifn[i0 + k];
        }
        if (lds_keys)
            eg_rank_classify_run(i0, n, nf, (const eg_ldsd *)lkeys, iv, f, m, v, fstart, dims, cls != nullptr, kbound, ana, r, w);
        else
            eg_rank_classify_run(i0, n, nf, im, iv, f, m, v, fstart, dims, cls != nullptr, kbound, ana, r, w,
                                 (const eg_ldsd *)lkeys, sk);
#ifdef EGDST_RANKCHK  // diagnostic build: every rank against a plain count (as in blk_sort_lds)
        for (int k = 0; k < n; k++) {
            const int i = i0 + k;
            int rc_ = 0;
            for (int j = 0; j < npts; j++)
                if (pt_before(im[j], iv[j], ifn[j], j, im[i], iv[i], ifn[i], i)) rc_++;
            if (dbg) atomicAdd(&dbg[12], 1);
            if (rc_ != r[k] && dbg) atomicAdd(&dbg[13], 1);
        }
#endif
#pragma unroll
        for (int k = 0; k < ENV_RK; k++) {
            if (k >= n) continue;
            if (r[k] < 0 || r[k] >= npts) {
                *oob = 1;
                continue;
            }
            rank[i0 + k] = r[k];
            if (perm_lds) {
                lperm[r[k]] = (unsigned short)(i0 + k);
                if (cls) lcls[r[k]] = w[k];
            } else {
                of[r[k]] = i0 + k;
                if (cls) cls[r[k]] = w[k];
            }
        }
    }
    TPST(2);
    for (int i = threadIdx.x; i < npts && bad && !networked; i += ENV_BS) {  // (lists still out of order after the presort: counting)
        const double m = im[i], v = iv[i];
        const int f = ifn[i];
        int r = 0;
        for (int j = 0; j < npts; j++)
            if (pt_before(im[j], iv[j], ifn[j], j, m, v, f, i)) r++;
        if (r < 0 || r >= npts) {
            *oob = 1;
            continue;
        }
        rank[i] = r;
        if (perm_lds)
            lperm[r] = (unsigned short)i;
        else
            of[r] = i;
    }
    __syncthreads();
    // step two: the sorted stream, position by position (see the head of this function).  `of` may hold the permutation:
    // every position is read and rewritten by the same thread.
    if (!*oob) {
        // (four positions per thread and round: their gathered loads in flight together, then the stores -- position by position
        //  the loop was a chain of dependent global reads, a sixth of a sort on the throughput path)
        for (int r0 = threadIdx.x; r0 < npts; r0 += 4 * ENV_BS) {
            double gm[4], gc[4], gv[4];
            int gf[4], gi[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int r = r0 + u * ENV_BS;
                gi[u] = -2, gm[u] = gc[u] = gv[u] = 0, gf[u] = 0;
                if (r < npts) {
                    const int i = perm_lds ? (int)lperm[r] : of[r];
                    gi[u] = (i < 0 || i >= npts) ? -1 : i;  // (a rank taken twice leaves a hole: never expected, the walk would read garbage)
                    if (gi[u] >= 0) gm[u] = im[i], gc[u] = ic[i], gv[u] = iv[i], gf[u] = ifn[i];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int r = r0 + u * ENV_BS;
                if (gi[u] == -1) *oob = 1;
                if (gi[u] < 0) continue;
                om[r] = gm[u];
                oc[r] = gc[u];
                ov[r] = gv[u];
                of[r] = gf[u];
                if (perm_lds && cls && *fused) cls[r] = lcls[r];
            }
        }
    }
    __syncthreads();
    // rank[fstart[f]+k] must be the sorted position of the k-th point of f IN SORTED ORDER (the reference builds
    // its per-function lists from the sorted array, egdst_solver.c:1247-1255).  That is what the merge produced
    // when every list was already ordered; after the counting fallback the lists are rebuilt from the sorted
    // stream (sequential, rare: a list is only out of order when a kink's double point overtakes a grid point).
    if (bad) {
        // `rank` is not needed as a permutation any more: overwrite it with the per-function position lists (every thread
        // counts a function's points in its chunk of the sorted stream, a prefix over the threads places them)
        const int C = (npts + ENV_BS - 1) / ENV_BS, qa = min(npts, (int)threadIdx.x * C), qb = min(npts, qa + C);
        for (int g = 0; g < nf; g++) {
            if (dims[g] <= 0) continue;
            int c = 0;
            for (int r = qa; r < qb; r++) c += (of[r] == g);
            __syncthreads();
            sh[threadIdx.x] = c;
            __syncthreads();
            int before = 0;
            for (int t = 0; t < (int)threadIdx.x; t++) before += sh[t];
            for (int r = qa; r < qb; r++)
                if (of[r] == g) rank[fstart[g] + before++] = r;
        }
        __syncthreads();
    }
    TPST(3);
#ifdef EGDST_VERIFY_SORT
    // diagnostic build: the output must be a permutation of the input in comp1 order
    for (int i = threadIdx.x; i < npts && !bad; i += ENV_BS) {
        int r = rank[i];
        bool okp = (of[r] == ifn[i]) && (om[r] == im[i] || (om[r] != om[r] && im[i] != im[i]));
        bool oks = (i == 0) || pt_before(om[i - 1], ov[i - 1], of[i - 1], 0, om[i], ov[i], of[i], 1);
        if ((!okp || !oks) && dbg && atomicCAS(&dbg[0], 0, 2720 + (okp ? 1 : 0)) == 0) {
            dbg[1] = i, dbg[2] = r, dbg[3] = npts, dbg[4] = nf, dbg[5] = bad, dbg[6] = ifn[i], dbg[7] = of[r];
            dbg[8] = dims[0], dbg[9] = fstart[0], dbg[10] = nf > 1 ? dims[1] : -1, dbg[11] = nf > 1 ? fstart[1] : -1;
            dbg[12] = nf > 2 ? dims[2] : -1, dbg[13] = nf > 2 ? fstart[2] : -1;
            dbg[14] = (im[i] != im[i]) * 1 + (iv[i] != iv[i]) * 2 + (i > 0 ? (of[i - 1] * 10) : 0);
            dbg[15] = i > 0 ? of[i] : -1;
            *oob = 1;
        }
    }
    __syncthreads();
#endif
}

// ---------------------------------------------------------------------------------------------
// LDS-resident sort of the point stream (npts <= lcap).  Keys are staged in LDS first, so the binary searches
// of the merge are LDS reads instead of dependent global loads.  When a list is out of comp1 order the whole
// stream goes through a bitonic network instead (padded to a power of two); the per-function position lists
// are then rebuilt from the sorted stream with one block scan per function.
// Returns the position lists (rank[fstart[f]+k] = sorted position of the k-th point of f).
// LDS regions (lcap entries each): R1, R2, R3 doubles; Lf, Lp ints.  Keys are staged in R1 (M) and R2 (V); the
// sorted stream ends up as M in R2, V in R3 and C in R1 (a key region is recycled as soon as every thread has
// read it), so a point costs 32 B of LDS including its class word and the 16-bit function id and position.
// The rank of a point is the sum over the functions of "how many of its points precede this one", and that count is
// also all the envelope walk's pre-classification needs to evaluate the function at the point (env_preclass): when
// the lists are in order the class word is computed right here from the staged keys (cls != nullptr; *fused = 1),
// `ana(g, x)` being the value of g before its first point (env_analytic / -inf).
template <class ANA>
static __device__ __forceinline__ eg_ldss *blk_sort_lds(int npts, int nf, const double *im, const double *ic,
                                                        const double *iv, const int *ifn, const eg_ldsi *fstart,
                                                        const eg_ldsi *dims, eg_ldsd *R1, eg_ldsd *R2, eg_ldsd *R3,
                                                        eg_ldss *Lf, eg_ldss *Lp, int lcap, int *sh, int *oob,
                                                        eg_ldsi *cls, int *fused, ANA ana, int *dbg = nullptr)
{
    eg_ldsd *Km = R1, *Kv = R2, *Lm = R2, *Lv = R3, *Lc = R1;
    const int tid = threadIdx.x;
#ifdef EGDST_EXP_NOCLS  // experiment: ranks only, the walk classifies
    cls = nullptr;
#endif
#ifdef EGDST_STAMPS3  // diagnostic: staging + order check (slot 3), ranks (4), scatter (7)
    unsigned long long t3_ = wall_clock64();
#define ST3(k) do { __syncthreads(); if (tid == 0 && dbg) { const unsigned long long n_ = wall_clock64(); atomicAdd((unsigned long long *)dbg + (k), n_ - t3_); t3_ = n_; } } while (0)
#else
#define ST3(k)
#endif
    *fused = 0;
    for (int i = tid; i < npts; i += ENV_BS) {
        Km[i] = im[i];
        Kv[i] = iv[i];
    }
    __syncthreads();
    int bad = 0;
    for (int i = tid + 1; i < npts; i += ENV_BS)
        if (ifn[i] == ifn[i - 1] && !pt_before(Km[i - 1], Kv[i - 1], ifn[i - 1], i - 1, Km[i], Kv[i], ifn[i], i)) bad = 1;
    bad = blk_sum(bad, sh);
    int P = 1;
    while (P < npts) P <<= 1;
    if (bad && P > lcap) bad = 2;  // no room for the padded network: counting ranks instead
    EG_CENSUS_BAD(bad);
    ST3(3);
    if (bad != 1) {
        double kbound = INFINITY;  // min over the functions of their last grid value (:1266-1271)
        if (!bad && cls) {
            for (int g = 0; g < nf; g++)
                if (dims[g] > 0) {
                    const double last = Km[fstart[g] + dims[g] - 1];
                    if (last < kbound) kbound = last;
                }
            *fused = 1;
        }
        int nact = 0, ga = -1, gb = -1;  // two non-empty lists: merge path (eg_merge2_classify)
        for (int g = 0; g < nf; g++)
            if (dims[g] > 0) {
                if (nact == 0) ga = g;
                if (nact == 1) gb = g;
                nact++;
            }
#ifdef EGDST_NO_MERGE2
        nact = 0;
#endif
        if (!bad && nact == 2) {
            auto emit = [&](int r, int i, int f, double m, double v, int w) {
                if (cls) cls[r] = w;
                Lf[r] = f;
                Lp[i] = r;
            };
            eg_merge2_classify(npts, nf, ga, (int)fstart[ga], (int)dims[ga], gb, (int)fstart[gb], (int)dims[gb], Km, Kv, cls != nullptr,
                               kbound, ana, emit);
        }
        for (int i0 = ENV_RK * tid; i0 < npts && !bad && nact != 2; i0 += ENV_RK * ENV_BS) {  // a run of ENV_RK consecutive points per round
            double m[ENV_RK], v[ENV_RK];
            int f[ENV_RK], r[ENV_RK], w[ENV_RK];  // (w: class words)
            const int n = min(ENV_RK, npts - i0);
#pragma unroll
            for (int k = 0; k < ENV_RK; k++) {
                m[k] = v[k] = 0, f[k] = 0;
                if (k < n) m[k] = Km[i0 + k], v[k] = Kv[i0 + k], f[k] = ifn[i0 + k];
            }
            eg_rank_classify_run(i0, n, nf, Km, Kv, f, m, v, fstart, dims, cls != nullptr, kbound, ana, r, w);
#ifdef EGDST_RANKCHK  // diagnostic build: every rank against a plain count
            for (int k = 0; k < n; k++) {
                const int i = i0 + k;
                int rc_ = 0;
                for (int j = 0; j < npts; j++)
                    if (pt_before(Km[j], Kv[j], ifn[j], j, Km[i], Kv[i], ifn[i], i)) rc_++;
                if (dbg) atomicAdd(&dbg[12], 1);                     // ranks checked / ranks that differ: debug words 12 / 13 of the draw
                if (rc_ != r[k] && dbg) atomicAdd(&dbg[13], 1);
                if (rc_ != r[k] && atomicAdd(&g_rankchk, 1) < 6)
                    printf("rankchk i=%d k=%d n=%d r=%d count=%d f=%d npts=%d nf=%d tid=%d\n", i, k, n, r[k], rc_, f[k], npts, nf,
                           (int)threadIdx.x);
            }
#endif
#pragma unroll
            for (int k = 0; k < ENV_RK; k++) {
                if (k >= n) continue;
                if (r[k] < 0 || r[k] >= npts) {
                    *oob = 1;
                    r[k] = -1;
                } else {
                    if (cls) cls[r[k]] = w[k];
                    Lf[r[k]] = f[k];
                }
                Lp[i0 + k] = r[k];
            }
        }
        for (int i = tid; i < npts && bad; i += ENV_BS) {
            const double m = Km[i], v = Kv[i];
            const int f = ifn[i];
            int r = 0;
            for (int j = 0; j < npts; j++)
                if (pt_before(Km[j], Kv[j], ifn[j], j, m, v, f, i)) r++;
            if (r < 0 || r >= npts) {
                *oob = 1;
                r = -1;
            } else
                Lf[r] = f;
            Lp[i] = r;
        }
        __syncthreads();  // every rank is known
        ST3(4);
        if (*oob) return Lp;
        for (int i = tid; i < npts; i += ENV_BS) Lv[Lp[i]] = Kv[i];  // R2 -> R3
        __syncthreads();
        for (int i = tid; i < npts; i += ENV_BS) Lm[Lp[i]] = Km[i];  // R1 -> R2 (V keys are consumed)
        __syncthreads();
        for (int i = tid; i < npts; i += ENV_BS) Lc[Lp[i]] = ic[i];  // -> R1 (M keys are consumed)
        __syncthreads();
        ST3(7);
        if (!bad) return Lp;
    } else {
        // bitonic network on (Km, Kv, Lf = function, Lp = input index); padding sorts last
        for (int i = tid; i < P; i += ENV_BS) {
            if (i < npts) {
                Lf[i] = ifn[i];
            } else {
                Km[i] = INFINITY;
                Kv[i] = -INFINITY;
                Lf[i] = 0xffff;  // (no function has this id: ENV_SMALLF < 65535)
            }
            Lp[i] = i;
        }
        __syncthreads();
        for (int k = 2; k <= P; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < P; i += ENV_BS) {
                    const int q = i ^ j;
                    if (q > i) {
                        const bool asc = (i & k) == 0;
                        const bool qfirst = pt_before(Km[q], Kv[q], Lf[q], Lp[q], Km[i], Kv[i], Lf[i], Lp[i]);
                        if (asc == qfirst) {  // ascending: q must not precede i; descending: the opposite
                            const double tm = Km[i], tv = Kv[i];
                            const int tf = Lf[i], tp = Lp[i];
                            Km[i] = Km[q], Kv[i] = Kv[q], Lf[i] = Lf[q], Lp[i] = Lp[q];
                            Km[q] = tm, Kv[q] = tv, Lf[q] = tf, Lp[q] = tp;
                        }
                    }
                }
                __syncthreads();
            }
        for (int r = tid; r < npts; r += ENV_BS) {  // same index in every region: no cross-thread hazard
            Lv[r] = Kv[r];      // R2 -> R3
            Lm[r] = Km[r];      // R1 -> R2
            Lc[r] = ic[Lp[r]];  // -> R1
        }
        __syncthreads();
    }
    // position lists from the sorted stream (Lp held input indices or ranks; both are consumed)
    eg_ldss *posl = Lp;
    for (int g = 0; g < nf; g++) {
        if (dims[g] <= 0) continue;
        int carry = 0;
        for (int base = 0; base < npts; base += ENV_BS) {
            const int r = base + tid;
            const int flag = (r < npts && Lf[r] == g);
            int tot;
            const int ex = blk_scan(flag, sh, &tot);
            if (flag) posl[fstart[g] + carry + ex] = r;
            carry += tot;
        }
    }
    __syncthreads();
    return posl;
}

// ---------------------------------------------------------------------------------------------
// Stop rule, compaction, secondary and primary envelopes, output rows for one (draw, ist).
#ifdef EGDST_CENSUS
#define ENV_FAIL_CENSUS(code) EG_CENSUS(2, draw, it, -1, -1, (code), cz_why_, wall_clock64() - cz_cell0_)
#else
#define ENV_FAIL_CENSUS(code) ((void)0)
#endif
#define ENV_FAIL(code)                                                                    \
    do {                                                                                  \
        if (tid == 0) {                                                                   \
            ENV_FAIL_CENSUS(code);                                                        \
            if (part == 1) /* part 2 reports the first error of the cell in choice order */ \
                b.secerr[cell * MS_ND + pid] = (code);                                    \
            else {                                                                        \
                eg_fail(b, draw, it, ist, (code));                                        \
                b.tlen[tk] = b.tthlen[tk] = 0;                                            \
            }                                                                             \
        }                                                                                 \
        return;                                                                           \
    } while (0)
#ifndef ENV_MAXSEG
#define ENV_MAXSEG 8            // segments of a walk; when there are more than walking waves, the waves take them from a queue
                                // (measured on C2: 16 segments of >= 128 or 96 points do not shorten the longest one -- an event
                                //  cluster cannot be cut -- and cost 0.2-0.3 ms per solve more to plan and gather)
#endif
#define ENV_MAXWW 8             // waves that walk segments
#ifndef ENV_SEGNF_SLICE
#define ENV_SEGNF_SLICE 128
#endif
// ENV_SEGNF_SLICE: the per-function LDS arrays (cur, mark, stack) are cut into one slice per walking wave:
                                // 128 entries (2*(nf+2) stack entries) while nf <= 126, wider slices and fewer walking waves
                                // beyond (C5 at full size: lists of up to 292 pieces walk on 3 waves instead of 1)
#ifndef ENV_SEG_EVENT_COST
#define ENV_SEG_EVENT_COST 320  // what a crossing costs the walk, in regular positions (~5 batches of 64)
#endif
#ifndef ENV_SEG_MINPTS
#define ENV_SEG_MINPTS 192      // sorted points per segment below which cutting a walk is not worth it
#endif
#define ENV_CK 4         // candidates a thread compacts per chunk
#ifndef ENV_SMALLF
#define ENV_SMALLF 1024
#endif
// ENV_SMALLF: functions (choices, or monotone pieces of one choice) whose bookkeeping fits the LDS arrays
                         // (C5 at full size: 292 pieces in one list; the reference allows 10000, :832)
static_assert(MS_ND <= ENV_SMALLF, "too many discrete choices for the LDS bookkeeping arrays");

struct WalkJob {  // what one envelope walk needs besides the sorted stream
    int it, ist, nf, npts, sec_id, ocap, e13, nthrhmax, stackcap, cap;
    int curcap, slice0;  // entries of cur[] / mark[] (stack: twice as many + 4); slice of them a walking wave gets while nf + 2 fits it
    double sec_ev;
    eg_ldsi *fstart, *dims, *cur, *mark;
    const eg_ldsd *evfa0;
    eg_ldsi *stack;
    int *dbg;
    int noseg;     // 1: never cut the walk into segments (EGDST_NOSEG; tests compare both ways)
    unsigned *segstat;  // [2] of the draw: segmented walks merged / fallen back
    int *sh;            // the workgroup's LDS scratch for scans ([ENV_MAXBS] ints)
    double *wM, *wV, *wC;  // scratch rows for the segments of a cut walk ([wcap] each; work arrays that are free while walking)
    int wcap;
    double *klog;  // kink log of the cell (dbgout), or nullptr
    int *kcnt;
    int kcap;
    double *og, *ov, *oc, *oth, *oix;
};

// All threads pre-classify the sorted stream, then wave 0 walks it; results through *err, *n, *nth (valid in wave 0,
// every lane of it holds the same values)
// SEG = false: the walk is never cut into segments -- the throughput path (k_tp_walk), where a whole cell belongs to one
// wave and the batch, not the cell, provides the parallelism; the planning, checking and gathering code is compiled out.
template <int L, bool SEG = true>
static __device__ __forceinline__ void run_walk(const ms_env *E, const WalkJob &j, const typename EgMem<L>::D *m,
                                                const typename EgMem<L>::DC *c, const typename EgMem<L>::D *v,
                                                const typename EgMem<L>::S *f, const typename EgMem<L>::S *posl,
                                                typename EgMem<L>::I *cls, int *err, int *n, int *nth, int classified)
{
    EnvCtxT<L> e;
    e.E = E;
    e.it = j.it;
    e.ist = j.ist;
    e.nf = j.nf;
    e.m = m;
    e.c = c;
    e.v = v;
    e.f = f;
    e.rank = posl;
    e.cls = cls;
    e.fstart = j.fstart;
    e.dims = j.dims;
    e.cur = j.cur;
    e.mark = j.mark;
    e.stack = j.stack;
    e.stackcap = j.stackcap;
    e.evfa0 = j.evfa0;
    e.sec_id = j.sec_id;
    e.sec_ev = j.sec_ev;
    e.og = j.og;
    e.ov = j.ov;
    e.oc = j.oc;
    e.oth = j.oth;
    e.oix = j.oix;
    e.ocap = j.ocap;
    e.e13 = j.e13;
    e.nthrhmax = j.nthrhmax;
    e.cap = j.cap;
    e.npts = j.npts;
    e.dbg = j.dbg;
    e.klog = j.klog;
    e.kcnt = j.kcnt;
    e.kcap = j.kcap;
    e.err = 0;
    e.later = 0;
    e.bound = 0;
    e.ci = 0;
    e.oi = e.oj = 0;
    e.lastg = 0;
    e.pm = -1;
#ifdef EGDST_STAMPS
    const unsigned long long pc0_ = wall_clock64();
#endif
    if (!classified) env_preclass(e, j.npts, cls, (int)threadIdx.x, ENV_BS);  // (else done by the sort)
    __syncthreads();
#ifdef EGDST_EMU
    if (classified && getenv("EGDST_VERIFY_CLS")) {  // harness: the sort's class words must equal env_preclass's
        static int chk[8192];
        static int nbad;
        if (threadIdx.x == 0) nbad = 0;
        __syncthreads();
        if (j.npts <= 8192) {
            env_preclass(e, j.npts, (typename EgMem<L>::I *)chk, (int)threadIdx.x, ENV_BS);
            __syncthreads();
            if (threadIdx.x == 0) {
                for (int p = 0; p < j.npts; p++)
                    if (chk[p] != cls[p] && nbad++ < 5)
                        fprintf(stderr, "CLS MISMATCH it=%d ist=%d sec=%d p=%d/%d f=%d sort=%08x preclass=%08x\n", j.it, j.ist, j.sec_id, p,
                                j.npts, (int)f[p], (unsigned)cls[p], (unsigned)chk[p]);
                if (nbad) fprintf(stderr, "CLS MISMATCHES %d\n", nbad);
                static long nver = 0;
                if (++nver % 50 == 0) fprintf(stderr, "cls verified on %ld walks\n", nver);
            }
            __syncthreads();
        }
    }
#endif
#ifdef EGDST_STAMPS
    if (threadIdx.x == 0 && j.dbg) atomicAdd((unsigned long long *)j.dbg + 1, wall_clock64() - pc0_);
#endif
    // ---- segmented walk ----------------------------------------------------------------------------------------
    // The walk is a state machine over the sorted positions, but its state after a position is small -- the current max
    // function and the grid value of the last output row (the cursors are rebuilt from counts at every event) -- and
    // PREDICTABLE at most positions: after a point that no other function lies above (class bit 0 clear) the walk has
    // kept that point, so the current max is the point's function and the last grid value is its M.  The stream is
    // therefore cut at such points into as many segments as the workgroup has waves, every wave walks one segment from
    // its predicted start state into a region of its own in the output arrays, and afterwards the predictions are
    // CHECKED against the true end states of the preceding segments: if every one holds (and nothing overflowed), the
    // concatenation of the segments' outputs is exactly what the sequential walk writes (by induction from segment 0,
    // which starts from the true initial state) and is compacted in place; otherwise -- an exact tie at a cut, an error,
    // a full grid -- wave 0 simply does the whole walk again sequentially.  Same code per position either way.
    __shared__ int sg_p[ENV_MAXSEG + 1], sg_oi[ENV_MAXSEG], sg_oj[ENV_MAXSEG], sg_err[ENV_MAXSEG], sg_pm[ENV_MAXSEG], sg_n, sg_next;
    __shared__ int sg_end[ENV_MAXSEG];  // first position a segment's walk did not consume
    __shared__ double sg_lastg[ENV_MAXSEG];
    const int tid_ = (int)threadIdx.x, wave_ = tid_ / WAVE, lane_ = tid_ & (WAVE - 1);
    int nseg = 1, thstride = j.nthrhmax, thcap = j.nthrhmax;
    // slice of the cursor arrays per walking wave, and how many waves that leaves room for
    const int slice = (j.nf + 2 <= j.slice0) ? j.slice0 : ((j.nf + 2 + 31) / 32) * 32;
    const int ww = min(min(ENV_BS / WAVE, ENV_MAXWW), j.curcap / slice);
#ifdef EGDST_STAMPS2
    unsigned long long s2_ = wall_clock64();
#define STAMP2(k) do { __syncthreads(); if (tid_ == 0 && j.dbg) { const unsigned long long n_ = wall_clock64(); atomicAdd((unsigned long long *)j.dbg + (k), n_ - s2_); s2_ = n_; } } while (0)
#else
#define STAMP2(k)
#endif
#if !defined(EGDST_SEQ_WALK)
    if (SEG) {
        // more segments than waves: the walking waves (at most ENV_MAXWW: the slices of the cursor arrays) take them from
        // a queue, which evens out what the cost estimate below gets wrong
        nseg = (ENV_BS / WAVE >= 2) ? ENV_MAXSEG : 1;
        while (nseg > 1 && j.npts < ENV_SEG_MINPTS * nseg) nseg--;  // (a segment should be worth a few batches)
        // every segment writes its rows and thresholds into a region of its own in scratch arrays (rows: up to twice its
        // points plus 64; thresholds: the last 2*thcap entries, thstride per segment -- nthrhmax of them when there is
        // room, fewer otherwise: a segment that runs out of its share only sends the walk back to one wave)
        if (j.klog || !j.wM || ww < 2 || nseg < 2) nseg = 1;
        {
            const long long room = ((long long)j.wcap - 2 * (long long)j.npts - 64 * nseg - 2) / 2;
            thcap = (int)(room < (long long)j.nthrhmax ? (room > 0 ? room : 0) : (long long)j.nthrhmax);
        }
        while (nseg > 1 && thcap / nseg < 8) nseg--;
        thstride = thcap / nseg;
        if (j.noseg) nseg = 1, thstride = j.nthrhmax;
    }
#endif
    if (SEG && nseg > 1) {
        // Where to cut: segments of equal COST, not of equal length.  A regular stretch costs ~1/64 of a batch per position,
        // a crossing several batches, and crossings cluster (the folds of a choice list sit in a narrow range of M).  The
        // number of crossings in a stretch is estimated by how often the function of consecutive points that nothing lies
        // above changes: every thread counts that in a chunk of the stream, a block-wide prefix sum of
        // (positions + ENV_SEG_EVENT_COST * changes) follows, and cut k goes to the chunk where it passes k/nseg of the total.
        __shared__ int sg_w[ENV_MAXBS / WAVE + 1];
        int *const sg_a = j.sh;  // (the workgroup's scan scratch, [ENV_MAXBS])
        const int C = (j.npts + ENV_BS - 1) / ENV_BS;
        int ff = -1, lf = -1, sw = 0;
        {
            const int qa = tid_ * C, qb = min(j.npts, qa + C);
            for (int q = qa; q < qb; q++) {
                const int w = cls[q];
                if (w >= 0 && !(w & 1)) {
                    const int fq = (int)f[q];
                    if (lf >= 0 && fq != lf) sw++;
                    if (ff < 0) ff = fq;
                    lf = fq;
                }
            }
        }
        sg_a[tid_] = lf;
        if (tid_ <= ENV_MAXSEG) sg_p[tid_] = -1;
        __syncthreads();
        if (tid_ > 0 && ff >= 0 && sg_a[tid_ - 1] >= 0 && sg_a[tid_ - 1] != ff) sw++;  // (a change across the chunk border)
        int cost = min(C, max(0, j.npts - tid_ * C)) + ENV_SEG_EVENT_COST * sw, incl = cost;
        for (int o = 1; o < WAVE; o <<= 1) {  // inclusive prefix sum over the wave, then over the waves
            const int t = __shfl_up(incl, o);
            if (lane_ >= o) incl += t;
        }
        __syncthreads();
        if (lane_ == WAVE - 1) sg_w[wave_] = incl;
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < ENV_BS / WAVE; w++) {
            if (w < wave_) before += sg_w[w];
            total += sg_w[w];
        }
        incl += before;
        {   // the chunk in which the running cost passes k/nseg of the total (thresholds k * floor(total / nseg): where exactly a
            // cut falls is a matter of balance only, and one 32-bit division replaces seven 64-bit ones per thread)
            const int per = total / nseg;
            for (int k = 1; k < nseg; k++) {
                const int thr = per * k;
                if (incl - cost < thr && thr <= incl) sg_p[k] = min(tid_ * C, j.npts - 1);
            }
        }
        if (tid_ < nseg && total <= 0) sg_p[tid_] = -1;
        __syncthreads();
        // cuts: the first position at or after the target whose point nothing lies above (and which is below the bound:
        // such class words are never negative); one lane per cut, then thread 0 keeps the increasing ones
        if (tid_ >= 1 && tid_ < nseg) {
            int q = sg_p[tid_];
            if (q < 0) q = j.npts;
            const int lim = min(j.npts - 2, q + 6 * WAVE);
            while (q < lim && (cls[q] < 0 || (cls[q] & 1))) q++;
            sg_p[tid_] = (q < lim) ? q + 1 : -1;
        }
        __syncthreads();
        if (tid_ == 0) {
            int n = 1, prev = 0;
            sg_p[0] = 0;
            for (int k = 1; k < nseg; k++) {
                const int pk = sg_p[k];
                if (pk > prev + WAVE / 4) sg_p[n++] = pk, prev = pk;
            }
            sg_p[n] = j.npts;
            sg_n = n;
            sg_next = ww;  // (the first segments go to the waves in order)
        }
        __syncthreads();
        nseg = sg_n;
    }
    STAMP2(3);  // planning: cost scan, cuts
    if (SEG && nseg > 1) {
        double *const wTH = j.wM + (j.wcap - 2 * (size_t)thcap), *const wIX = wTH + thcap;
        // (every thread owns a copy of the context: a wave points its own at the segment it has taken)
        for (int sgi = wave_; wave_ < ww && sgi < nseg;) {
            const int p0 = sg_p[sgi], p1 = sg_p[sgi + 1];
            const size_t row0 = 2 * (size_t)p0 + 64 * (size_t)sgi;
            e.cur = j.cur + wave_ * slice;
            e.mark = j.mark + wave_ * slice;
            e.stack = j.stack + wave_ * 2 * slice;
            e.stackcap = 2 * slice;
            e.og = j.wM + row0, e.ov = j.wV + row0, e.oc = j.wC + row0;
            e.ocap = 2 * (p1 - p0) + 64;
            e.oth = wTH + (size_t)sgi * thstride, e.oix = wIX + (size_t)sgi * thstride;
            e.nthrhmax = thstride;
            e.err = 0;
            int pm0 = -1;
            double lastg0 = 0;
            if (sgi > 0) pm0 = (int)f[p0 - 1], lastg0 = m[p0 - 1];
#ifdef EGDST_STAMPS
            const unsigned long long sgt0_ = wall_clock64();
#endif
            env_walk_wave(e, j.npts, p0, p1, sgi > 0 ? 1 : 0, pm0, lastg0);
#ifdef EGDST_STAMPS
            if (lane_ == 0 && j.dbg) {  // diagnostic: longest segment (slot 0) and sum over segments (slot 1, was: classification)
                const unsigned long long d_ = wall_clock64() - sgt0_;
#if !defined(EGDST_STAMPS2) && !defined(EGDST_STAMPS5)
                atomicMax((unsigned long long *)j.dbg + 0, d_);
#endif
#ifndef EGDST_STAMPS2_WHY
                atomicAdd((unsigned long long *)j.dbg + 1, d_);
#endif
            }
#endif
            int nx = 0;
            if (lane_ == 0) {
                sg_oi[sgi] = e.oi, sg_oj[sgi] = e.oj, sg_err[sgi] = e.err, sg_pm[sgi] = e.pm;
                sg_lastg[sgi] = e.lastg;
                sg_end[sgi] = e.iend;
                nx = atomicAdd(&sg_next, 1);
            }
            sgi = __shfl(nx, 0);
        }
        STAMP2(4);  // the segments (wall: the longest one)
        __syncthreads();
        if (tid_ == 0) {  // do the predictions hold?
            int ok = 1, toi = 0, toj = 0;
            for (int k = 0; k < nseg; k++) {
                if (sg_err[k]) ok = 0;
                toi += sg_oi[k], toj += sg_oj[k];
                if (k + 1 < nseg) {
                    const int q = sg_p[k + 1] - 1;
                    if (sg_pm[k] != (int)f[q] || !(sg_lastg[k] == m[q])) ok = 0;
                }
            }
#ifdef EGDST_EMU
            if (getenv("EGDST_EMU_FAIL_SEG")) ok = 0;  // harness: every segmented walk is done again by one wave
#endif
            if (sg_oj[0] <= 0) ok = 0;                             // (segment 0 must have left the first-point phase)
            if (toi >= j.ocap || toj >= j.nthrhmax) ok = 0;      // a full grid or threshold list: the sequential walk reports it
            sg_n = ok ? nseg : 0;
            atomicAdd(&j.segstat[ok ? 0 : 1], 1u);
#ifdef EGDST_CENSUS
            cz_sh_fb = ok ? 1 : 2;
#endif
        }
        __syncthreads();
        if (sg_n) {
            // gather: the segments' rows and thresholds, in order, to the front of the output (one pass over all of them:
            // every thread finds the segment of its output row from the running totals)
            int off = 0, offj = 0;
            for (int k = 0; k < nseg; k++) off += sg_oi[k], offj += sg_oj[k];
            // (four rows per thread and round: all their loads in flight, then the stores -- a row at a time the loop was a chain
            //  of global round trips, 8 us of a 37-us walk on the throughput path, profiles/r04_stamps_seg2_tp.txt)
            for (int r0 = tid_; r0 < off; r0 += 4 * ENV_BS) {
                double gm[4], gv[4], gc[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int r = r0 + u * ENV_BS;
                    gm[u] = gv[u] = gc[u] = 0;
                    if (r < off) {
                        int k = 0, q = r;
                        while (q >= sg_oi[k]) q -= sg_oi[k], k++;
                        const size_t src = 2 * (size_t)sg_p[k] + 64 * (size_t)k + q;
                        gm[u] = j.wM[src], gv[u] = j.wV[src], gc[u] = j.wC[src];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int r = r0 + u * ENV_BS;
                    if (r < off) j.og[r] = gm[u], j.ov[r] = gv[u], j.oc[r] = gc[u];
                }
            }
            for (int r = tid_; r < offj; r += ENV_BS) {
                int k = 0, q = r;
                while (q >= sg_oj[k]) q -= sg_oj[k], k++;
                j.oth[r] = wTH[(size_t)k * thstride + q];
                j.oix[r] = wIX[(size_t)k * thstride + q];
            }
            STAMP2(7);  // check + gather
            if (ENV_CDEFER(L)) {  // the consumption of the rows the batches kept (env_walk_wave), over the gathered rows
                __syncthreads();
                int k = 0, offk = 0;
                for (int p0 = tid_; p0 < j.npts; p0 += 4 * ENV_BS) {  // (four positions per thread and round, as above)
                    int dst[4];
                    double cc[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int p = p0 + u * ENV_BS;
                        dst[u] = -1, cc[u] = 0;
                        if (p < j.npts) {
                            while (p >= sg_p[k + 1]) offk += sg_oi[k], k++;
                            if (p < sg_end[k]) {
                                const int d = cls[p];
                                if (d >= 0) dst[u] = offk + d, cc[u] = c[p];
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (dst[u] >= 0) j.oc[dst[u]] = cc[u];
                }
            }
            if (tid_ < WAVE) *err = 0, *n = off, *nth = offj;
            return;
        }
        // fall through: the plain walk by wave 0, from the job's own arrays again
        e.cur = j.cur, e.mark = j.mark, e.stack = j.stack, e.stackcap = j.stackcap;
        e.og = j.og, e.ov = j.ov, e.oc = j.oc, e.oth = j.oth, e.oix = j.oix;
        e.ocap = j.ocap, e.nthrhmax = j.nthrhmax, e.err = 0;
        if (ENV_CDEFER(L)) {  // (the segments left row numbers where the class words were)
            env_preclass(e, j.npts, cls, (int)threadIdx.x, ENV_BS);
            __syncthreads();
        }
    }
    if ((int)threadIdx.x < WAVE) {
        env_walk_wave(e, j.npts);
        *err = e.err;
        *n = e.oi;
        *nth = e.oj;
        if (ENV_CDEFER(L) && lane_ == 0) sg_end[0] = e.err ? 0 : e.iend;
    }
    if (ENV_CDEFER(L)) {  // the consumption of the rows the batches kept (env_walk_wave)
        __syncthreads();
        const int pe = sg_end[0];
        for (int p0 = tid_; p0 < pe; p0 += 4 * ENV_BS) {
            int dst[4];
            double cc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int p = p0 + u * ENV_BS;
                dst[u] = -1, cc[u] = 0;
                if (p < pe) {
                    const int d = cls[p];
                    if (d >= 0) dst[u] = d, cc[u] = c[p];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (dst[u] >= 0) j.oc[dst[u]] = cc[u];
        }
    }
#ifdef EGDST_STAMPS2
    __syncthreads();
    if (tid_ == 0 && j.dbg) {  // diagnostic: walks by one wave -- ticks, (<<40) how many, (<<52) how many of them not for being short
        const int why = j.klog ? 1 : !j.wM ? 2 : ww < 2 ? 3 : j.npts < 2 * ENV_SEG_MINPTS ? 4 :
                        (2 * (long long)j.npts + 128 + 2 * (long long)j.nthrhmax + 2 > (long long)j.wcap) ? 5 : 6;
        atomicAdd((unsigned long long *)j.dbg + 0, (wall_clock64() - s2_) + (1ull << 40) + ((unsigned long long)(why != 4) << 52));
#ifdef EGDST_STAMPS2_WHY  // (slot 1 then holds the reasons, 10 bits each, instead of the segments' sum)
        atomicAdd((unsigned long long *)j.dbg + 1, 1ull << (10 * (why - 1)));
#endif
    }
#endif
}

// lcap: sorted points that fit the dynamic LDS (32 B each: sorted M/C/V, class word, 16-bit function id and position).
// pass 0: launched with a small lcap so that two workgroups share a CU; a cell whose stream does not fit is left
//         untouched and flagged in b.defer;  pass 1: launched with the large lcap, works on the flagged cells only
//         (streams beyond that sort and walk in global memory);  pass 2: every cell, one launch (no deferral).
#ifndef ENV_MINW
#define ENV_MINW 1
#endif
// part: 0 the whole cell in one workgroup (jobs one after another);  1 / 2 (models with several choices): the choice lists
//       and their secondary envelopes as workgroups of their own, one per (cell, choice) -- they are independent of each
//       other (egdst_solver.c:668: envelope2 per id) -- and then the primary envelope per cell.  The lists of part 1 live in
//       per-choice slices of the cell's work arrays; what part 2 needs besides them is handed over in Batch.sec*.
#ifdef ENV_VGPR  // experiment: fewer registers, so that waves of the grid kernels fit next to an envelope workgroup
#define ENV_VGPR_ATTR __attribute__((amdgpu_waves_per_eu(ENV_VGPR, ENV_VGPR)))
#else
#define ENV_VGPR_ATTR
#endif
// bxi: the workgroup's job -- cell slot of the group (part 0, 2) or cell slot * MS_ND + choice (part 1)
static __device__ __forceinline__ void eg_envelope_cell(BatchRef b, int it, int terminal, int lcap, int pass, int part, int bxi)
{
    EG_DYN_LDS(dynlds);
#ifdef EGDST_STAMPS5  // diagnostic: residency of the whole workgroup (slot 0), completed cells only
    const unsigned long long wg_t0_ = wall_clock64();
#endif
    __shared__ int sh[ENV_MAXBS + 2];  // (+2: blk_reduce3 of a one-thread harness build)
    __shared__ int s_fstart[ENV_SMALLF], s_fdims[ENV_SMALLF], s_fcur[ENV_SMALLF], s_fmark[ENV_SMALLF];
    __shared__ int s_stack[2 * (ENV_SMALLF + 2)];
    __shared__ double s_evfa0[MS_ND];
    __shared__ int s_cnt[MS_ND], s_start[MS_ND];
    __shared__ int s_err, s_n, s_m, s_oob;
    const int bx_ = (part == 1) ? bxi / MS_ND : bxi, pid = (part == 1) ? bxi % MS_ND : 0;
    const int ist = bx_ % MS_NST, draw = b.order[b.draw0 + bx_ / MS_NST];
    const int tid = threadIdx.x;
    const int slot = (b.g.nslots == 2) ? (it & 1) : it;
    const size_t tk = ((size_t)slot * b.g.ndraw + draw) * MS_NST + ist;
    const size_t cell = (size_t)draw * MS_NST + ist;
    if (part == 1 && tid == 0) {  // (whatever happens below, part 2 finds a record)
        b.secn[cell * MS_ND + pid] = 0;
        b.secact[cell * MS_ND + pid] = 0;
        b.secerr[cell * MS_ND + pid] = 0;
        b.secev[cell * MS_ND + pid] = 0.0;
        b.secevals[cell * MS_ND + pid] = 0ull;
    }
#ifdef EGDST_CENSUS
    const unsigned long long cz_cell0_ = wall_clock64();
    const int cz_why_ = (pass == 1) ? b.defer[cell] : 0;
#endif
    if (pass == 1) {
        if (!b.defer[cell]) return;  // done in pass 0
        __syncthreads();
        if (tid == 0) b.defer[cell] = 0;
    }
    if (b.status[draw]) {
        if (tid == 0 && part != 1) b.tlen[tk] = b.tthlen[tk] = 0;
        return;
    }
    ms_env E = eg_env(b, draw);
    ms_pv cur;
    cur.it = it;
    cur.ist = ist;
    cur.id = 0;
    cur.cash = cur.savings = cur.shock = 0;
    if (ms_feasible(&E, &cur) != 1) {
        if (tid == 0 && part != 1) b.tlen[tk] = b.tthlen[tk] = 0;
        return;
    }
    // Rows of this cell that may hold non-zero leftovers of an earlier period (ping-pong slots) or of an earlier
    // solve.  The reference reads one row past a table's end when the table has a single row (linter on n=1,
    // egdst_lib.c:123-206) and finds zeros there, because every period owns a freshly zeroed matrix
    // (egdst_solver.c:198-217): the rows past the new length are zeroed below to keep exactly that.
    const int hw_rows = b.thw[tk], hw_th = b.thhw[tk];
    __syncthreads();
    if (tid == 0 && part != 1) b.thw[tk] = b.g.Sp, b.thhw[tk] = b.g.nthrhmax;  // until this cell is complete: unknown
    const int ngridmax = b.g.ngridmax, ngridm = b.g.ngridm;
    const double mmax = b.g.mmax;
    const int compact = b.g.Cp < ngridmax;  // physical capacity below the logical one: overflow = EGDST_E_CAPACITY
    // part 1 works in the slice [pid*Cp, (pid+1)*Cp) of the cell's work arrays (W then means that slice)
    const size_t Wcell = (size_t)(MS_ND + 1) * b.g.Cp, W = (part == 1) ? (size_t)b.g.Cp : Wcell;
    const size_t wo = ((size_t)draw * MS_NST + ist) * Wcell + (size_t)pid * b.g.Cp;
    double *pM = b.pM + wo, *pC = b.pC + wo, *pV = b.pV + wo;
    int *pF = b.pF + wo;
    double *sM = b.sM + wo, *sC = b.sC + wo, *sV = b.sV + wo;
    int *sF = b.sF + wo;
    double *qM = b.qM + wo, *qC = b.qC + wo, *qV = b.qV + wo;
    int *qF = b.qF + wo, *rank = b.rank + wo;
    const size_t eo = ((size_t)draw * MS_NST + ist) * (size_t)b.g.Cp;
    const size_t eto = (((size_t)draw * MS_NST + ist) * MS_ND + pid) * (size_t)b.g.nthrhmax;
    // (part 1: the secondary envelope writes its result over the choice list it came from, which is dead by then)
    double *eM = (part == 1) ? pM : b.eM + eo, *eV = (part == 1) ? pV : b.eV + eo, *eC = (part == 1) ? pC : b.eC + eo;
    double *eTH = b.eTH + eto, *eIX = b.eIX + eto;
    double *oM = b.tM + tk * b.g.Sp, *oC = b.tC + tk * b.g.Sp, *oV = b.tV + tk * b.g.Sp;
    double *oTH = b.tTH + tk * b.g.nthrhmax, *oD = b.tD + tk * b.g.nthrhmax;
    // typed LDS views
#ifdef EGDST_EMU
    // sanitizer harness: 64 poisoned bytes between the LDS regions, so that an overrun of one region into the next
    // traps under AddressSanitizer instead of landing in a neighbour silently
    eg_ldsd *R1 = (eg_ldsd *)dynlds, *R2 = R1 + lcap + 8, *R3 = R2 + lcap + 8;
    eg_ldsi *Lq = (eg_ldsi *)(R3 + lcap + 8);
    eg_ldss *Lf = (eg_ldss *)(Lq + lcap + 16), *Lr = Lf + lcap + 32;
    EG_EMU_POISON(R1 + lcap, 64), EG_EMU_POISON(R2 + lcap, 64), EG_EMU_POISON(R3 + lcap, 64), EG_EMU_POISON(Lq + lcap, 64),
        EG_EMU_POISON(Lf + lcap, 64), EG_EMU_POISON(Lr + lcap, 64);
#else
    eg_ldsd *R1 = (eg_ldsd *)dynlds, *R2 = R1 + lcap, *R3 = R2 + lcap;
    eg_ldsi *Lq = (eg_ldsi *)(R3 + lcap);            // class words
    eg_ldss *Lf = (eg_ldss *)(Lq + lcap), *Lr = Lf + lcap;  // function ids, position lists
#endif
    eg_ldsi *fstart = (eg_ldsi *)s_fstart, *fdims = (eg_ldsi *)s_fdims;

    WalkJob job;
    job.it = it;
    job.ist = ist;
    job.ocap = b.g.Cp;  // rows of a secondary output (eM..) and of a table without its a0 row
    job.e13 = compact ? EGDST_E_CAPACITY : 13;
    job.nthrhmax = b.g.nthrhmax;
    job.stackcap = 2 * (ENV_SMALLF + 2);
    job.curcap = ENV_SMALLF, job.slice0 = ENV_SEGNF_SLICE;
    job.cap = (int)W;
    job.fstart = fstart;
    job.dims = fdims;
    job.cur = (eg_ldsi *)s_fcur;
    job.mark = (eg_ldsi *)s_fmark;
    job.evfa0 = (const eg_ldsd *)s_evfa0;
    job.stack = (eg_ldsi *)s_stack;
    job.dbg = b.dbg + 16 * draw;
    job.noseg = b.noseg;
    job.sh = sh;
    job.segstat = b.segstat + 2 * (size_t)draw;
    job.klog = nullptr, job.kcnt = nullptr, job.kcap = b.kcap;
    if (b.klog) {  // third output of the solver gateway requested (egdst_set_dbgout): this cell's slice of the log
        const size_t kc = ((size_t)it * b.g.ndraw + draw) * MS_NST + ist;   // (the host launches part 0 when the log is on)
        job.klog = b.klog + kc * 4 * (size_t)b.kcap;
        job.kcnt = b.kcnt + kc;
        if (tid == 0) *job.kcnt = 0;
    }

#ifdef EGDST_STAMPS  // diagnostic build: where does a workgroup spend its time (wall_clock64 ticks of 10 ns)
#define STAMP(k)                                                                                   \
    do {                                                                                           \
        if (tid == 0) {                                                                            \
            const unsigned long long now_ = wall_clock64();                                        \
            atomicAdd((unsigned long long *)(b.dbg + 16 * draw) + 2 + (k), now_ - stamp_);        \
            stamp_ = now_;                                                                         \
        }                                                                                          \
    } while (0)
    unsigned long long stamp_ = wall_clock64();
#else
#define STAMP(k)
#endif
    if (tid == 0) s_err = 0, s_oob = 0;
    __syncthreads();
    int any = 0, nall = 0, outn = 0, outm = 0, done = 0;
    unsigned long long evals = 0;
    // jobs 0..MS_ND-1: the list of one choice (stop rule, compaction, secondary envelope when it folds back);
    // job MS_ND: the primary envelope across choices.  Sorting and walking is ONE code site for all of them.
    int part_eff = part;
    if (part == 2) {
        // the choice lists of part 1: first error in choice order, then the lists packed one after another at the front
        // of the cell's arrays (choice 0's slice starts there already; a later list moves down, never past its own start)
        // (a list or its pieces did not fit the slice of the cell's arrays -- a degenerate guess stream with ~ngridmax
        // points: this workgroup then does the whole cell again, one job after the other, in the full arrays)
        int err2 = 0;
        for (int k = 0; k < MS_ND; k++)
            if (b.secerr[cell * MS_ND + k] == -1) part_eff = 0;
        for (int k = 0; k < MS_ND && !err2 && part_eff; k++) err2 = b.secerr[cell * MS_ND + k];
        if (err2) ENV_FAIL(err2);
        for (int k = 0; k < MS_ND && part_eff; k++) {
            evals += b.secevals[cell * MS_ND + k];
            const int cntk = b.secn[cell * MS_ND + k];
            if (b.secact[cell * MS_ND + k]) any = 1;
            if (tid == 0) {
                s_cnt[k] = cntk;
                s_start[k] = nall;
                s_evfa0[k] = b.secev[cell * MS_ND + k];
            }
            const size_t src = (size_t)k * b.g.Cp;
            if (k > 0 && cntk > 0) {
                for (int base = 0; base < cntk; base += ENV_BS) {  // (chunks: read, barrier, write -- the ranges may overlap)
                    const int r = base + tid;
                    double a_ = 0, b2_ = 0, c_ = 0;
                    if (r < cntk) a_ = pM[src + r], b2_ = pC[src + r], c_ = pV[src + r];
                    __syncthreads();
                    if (r < cntk) pM[nall + r] = a_, pC[nall + r] = b2_, pV[nall + r] = c_;
                    __syncthreads();
                }
            }
            for (int r = tid; r < cntk; r += ENV_BS) pF[nall + r] = k;
            nall += cntk;
        }
        __syncthreads();
    }
    for (int jb = (part_eff == 2 ? MS_ND : (part_eff == 1 ? pid : 0)); jb <= (part_eff == 1 ? pid : MS_ND) && !done; jb++) {
        const bool primary = (jb == MS_ND);
        const int id = jb;
        const double *iM, *iC, *iV;
        const int *iF;
        int cnt = 0;
        double evfa0 = 0;
        if (!primary) {
            __syncthreads();
            if (tid == 0) {
                s_cnt[id] = 0;
                s_start[id] = nall;
                s_evfa0[id] = 0.0;
            }
            const ProbeOut P = b.probe[((size_t)draw * MS_NST + ist) * MS_ND + id];
            if (!P.active) continue;
            any = 1;
            const size_t co = eg_cand(b, draw, ist, id);
            int nreq = 0, fusedstats = 0;
            evfa0 = P.evfa0;
            if (terminal) {
                nreq = ngridm - 1;  // candidate indices 0..ngridm-1, all kept
            } else if (P.seq) {
                nreq = P.np - 1;    // k_fixup stored the kept points of the whole stream in order
                evals += (unsigned long long)P.probe_evals;
            } else {
                const int navail = P.grid ? min(ngridm - 1, ngridmax - 1 - P.ncalls) : 0;
                // first requested point whose returned M stops the stream (:1100): the point itself is kept
                int first = navail + 1;
                for (int n = 1 + tid; n <= navail; n += ENV_BS)  // (no early exit: the loads of a thread go out together)
                    if (!(b.cM[co + n] < mmax)) first = min(first, n);
                first = blk_min(first, sh);
                nreq = min(first, navail);
                fusedstats = 1;
            }
            // ---- compaction of kept points into the choice's list ------------------------------
            // Every thread takes ENV_CK consecutive candidates of a chunk of ENV_CK*ENV_BS, so a list of a few thousand is
            // one chunk: one round of loads, one block scan.  The statistics of the requested points (hard errors, statuses
            // 1 and 2, evaluation counts) ride along and are reduced once at the end.
            int hard = 0, n12 = 0, ev = 0;
            {
                int carry = 0;
                for (int base = 0; base <= nreq; base += ENV_CK * ENV_BS) {
                    const int n0 = base + ENV_CK * tid;
                    double vM[ENV_CK], vC[ENV_CK], vV[ENV_CK];
                    int keep[ENV_CK], mine = 0;
#pragma unroll
                    for (int k = 0; k < ENV_CK; k++) {
                        const int n = n0 + k;
                        keep[k] = 0;
                        vM[k] = vC[k] = vV[k] = 0;
                        if (n <= nreq) {
                            vM[k] = b.cM[co + n], vC[k] = b.cC[co + n], vV[k] = b.cV[co + n];
                            if (terminal || P.seq)
                                keep[k] = 1;
                            else if (n == 0)
                                keep[k] = P.np;
                            else {
                                const int sc = b.cSt[co + n], st = eg_sc_status(sc);
                                if (st < 0) hard = max(hard, -st);
                                n12 |= (st == 1) | ((st == 2) << 1);
                                ev += eg_sc_count(sc);
                                keep[k] = (st == 0 && isfinite(vM[k]));
                            }
                        }
                        mine += keep[k];
                    }
                    int tot;
                    int d = nall + carry + blk_scan_int(mine, sh, &tot);
#pragma unroll
                    for (int k = 0; k < ENV_CK; k++)
                        if (keep[k]) {
                            if (d < 0 || (size_t)d >= W)
                                s_oob = 1;
                            else {
                                pM[d] = vM[k];
                                pC[d] = vC[k];
                                pV[d] = vV[k];
                                pF[d] = id;
                            }
                            d++;
                        }
                    carry += tot;
                }
                cnt = carry;
            }
            if (fusedstats) {
                blk_reduce3(&hard, &n12, &ev, sh);
                evals += (unsigned long long)ev + (unsigned long long)P.probe_evals;
                if (hard) ENV_FAIL(hard);
                if (n12 & 1) ENV_FAIL(26);  // k_fixup should have taken this stream over
                if (n12 & 2) evfa0 = -INFINITY;
            }
            __syncthreads();
            STAMP(0);  // stop rule + compaction
            if (s_oob) ENV_FAIL(part == 1 ? -1 : (compact ? EGDST_E_CAPACITY : 2705));  // (-1: see part 2)
            // ---- does the list fold back?  then it needs a secondary envelope (:776-913) -----------
            int nfold = 0;
            if (!terminal && cnt > 1) {
                for (int i = 1 + tid; i < cnt; i += ENV_BS)
                    if (pM[nall + i - 1] > pM[nall + i] || pV[nall + i - 1] > pV[nall + i]) nfold++;
                nfold = blk_sum(nfold, sh);
            }
            if (nfold == 0) {
                if (tid == 0) {
                    s_cnt[id] = cnt;
                    s_evfa0[id] = evfa0;
                }
                nall += cnt;
                continue;
            }
            if (id + nfold + 1 > ENV_SMALLF) ENV_FAIL(2709);  // more monotone pieces than this build keeps in LDS
            // input with one constant-extrapolation point appended to every closed piece (:822-835)
            int carry = 0, lastfold = 0;
            for (int base = 0; base < cnt; base += ENV_BS) {
                const int i = base + tid;
                int fold = 0;
                if (i >= 1 && i < cnt) fold = (pM[nall + i - 1] > pM[nall + i] || pV[nall + i - 1] > pV[nall + i]);
                int tot;
                const int ex = blk_scan(fold, sh, &tot);
                if (i < cnt) {
                    const int sidx = carry + ex + fold;  // pieces closed before this point
                    const int d = i + sidx;
                    if ((size_t)d >= W || id + sidx >= ENV_SMALLF)
                        s_oob = 1;
                    else {
                        sM[d] = pM[nall + i];
                        sC[d] = pC[nall + i];
                        sV[d] = pV[nall + i];
                        sF[d] = id + sidx;
                        if (fold) {
                            sM[d - 1] = 1.5 * mmax;
                            sC[d - 1] = pC[nall + i - 1];
                            sV[d - 1] = pV[nall + i - 1];
                            sF[d - 1] = id + sidx - 1;
                            fstart[id + sidx] = d;
                            if (sidx == nfold) lastfold = i;
                        }
                    }
                    if (i == 0) fstart[id] = 0;
                }
                carry += tot;
            }
            lastfold = blk_sum(lastfold, sh);
            if (s_oob) ENV_FAIL(part == 1 ? -1 : (compact ? EGDST_E_CAPACITY : 2706));
            if (lastfold + (nfold - 1) >= ngridmax) ENV_FAIL(17);
            if (id + nfold >= 10000) ENV_FAIL(18);
            job.npts = cnt + nfold;
            job.nf = id + nfold + 1;
            __syncthreads();
            for (int f = tid; f < job.nf; f += ENV_BS) {
                if (f < id)
                    fdims[f] = 0, fstart[f] = 0;
                else
                    fdims[f] = ((f + 1 < job.nf) ? fstart[f + 1] : job.npts) - fstart[f];
            }
            __syncthreads();
            job.sec_id = id;
            job.sec_ev = evfa0;
            job.og = eM, job.ov = eV, job.oc = eC, job.oth = eTH, job.oix = eIX;
            iM = sM, iC = sC, iV = sV, iF = sF;
        } else {
            __syncthreads();
            if (!any) ENV_FAIL(14);
            if (nall == 0) ENV_FAIL(15);
            int nact = 0, fsingle = -1;
            for (int k = 0; k < MS_ND; k++)
                if (s_cnt[k] > 0) nact++, fsingle = k;
            if (nact == 1) {
                // a single tabulated function: the walk keeps every point except repeats of a grid value,
                // provided the list is already in comp1 order (it is unless M ties carry increasing V)
                int bad = 0;
                for (int i = 1 + tid; i < nall; i += ENV_BS)
                    if (pM[i - 1] > pM[i] || (pM[i - 1] == pM[i] && pV[i - 1] < pV[i])) bad = 1;
                bad = blk_sum(bad, sh);
                if (!bad) {
                    if (nall > b.g.Cp) ENV_FAIL(compact ? EGDST_E_CAPACITY : 13);
                    int carry = 0;
                    for (int base = 0; base < nall; base += ENV_BS) {
                        const int i = base + tid;
                        const int keepit = (i < nall) && (i == 0 || pM[i] != pM[i - 1]);
                        int tot;
                        const int ex = blk_scan(keepit, sh, &tot);
                        if (keepit) {
                            const int d = 1 + carry + ex;
                            oM[d] = pM[i];
                            oC[d] = pC[i];
                            oV[d] = pV[i];
                        }
                        carry += tot;
                    }
                    outn = carry;
                    outm = 1;
                    if (tid == 0) {
                        oTH[0] = b.g.a0;
                        oD[0] = fsingle;
                    }
                    done = 1;
                    if (outn >= ngridmax || 1 >= b.g.nthrhmax) ENV_FAIL(outn >= ngridmax ? 13 : 20);
                    break;
                }
            }
            for (int f = tid; f < MS_ND; f += ENV_BS) {
                fstart[f] = s_start[f];
                fdims[f] = s_cnt[f];
            }
            __syncthreads();
            job.npts = nall;
            job.nf = MS_ND;
            job.sec_id = -1;
            job.sec_ev = 0;
            job.og = oM + 1, job.ov = oV + 1, job.oc = oC + 1, job.oth = oTH, job.oix = oD;
            iM = pM, iC = pC, iV = pV, iF = pF;
        }
        // ---- common: sort the stream (comp1 order) and walk it ----------------------------------------
        // value of function g before its first point (env_analytic / env_evf of the walk)
        const int sec_id_ = job.sec_id;
        const double sec_ev_ = job.sec_ev;
        auto ana = [&](int g, double x) -> double {
            const double ev = (sec_id_ >= 0) ? (g == sec_id_ ? sec_ev_ : -INFINITY) : s_evfa0[g];
            if (ev == -INFINITY) return -INFINITY;
            ms_pv cv;
            cv.it = it;
            cv.ist = ist;
            cv.id = g;
            cv.cash = cv.savings = cv.shock = 0;
            return ms_utility(&E, &cv, x - E.a0) + ms_discount(&E, &cv) * ev;
        };
#ifdef EGDST_CENSUS
        const unsigned long long cz_job0_ = wall_clock64();
        unsigned long long cz_sort1_ = cz_job0_;
#endif
        int fused = 0;
        if (job.npts <= lcap) {
            const eg_ldss *posl = blk_sort_lds(job.npts, job.nf, iM, iC, iV, iF, fstart, fdims, R1, R2, R3, Lf, Lr, lcap,
                                               sh, &s_oob, Lq, &fused, ana, job.dbg);
            STAMP(3);  // LDS sort
#ifdef EGDST_CENSUS
            cz_sort1_ = wall_clock64();
#endif
            if (s_oob) ENV_FAIL(2704);
            {
                int we = 0, wn = 0, wm = 0;
                job.wM = qM, job.wV = qV, job.wC = qC, job.wcap = (int)W;   // (the global sort arrays are idle on this path)
                run_walk<1>(&E, job, R2, R1, R3, Lf, posl, Lq, &we, &wn, &wm, fused);  // sorted M, C, V
                if (tid < WAVE) s_err = we, s_n = wn, s_m = wm;
            }
        } else if (pass == 0) {
            // does not fit the small LDS of this pass: leave the cell exactly as it was for pass 1
            if (tid == 0) {
                b.defer[cell] = 1;
                b.dbg[16 * draw + 15] += 1;  // diagnostic: deferrals of this draw
                b.thw[tk] = hw_rows;
                b.thhw[tk] = hw_th;
            }
            return;
        } else {
            int *gcls = b.gcls + wo;  // class words of the sorted stream
            blk_rank_sort(job.npts, job.nf, iM, iC, iV, iF, fstart, fdims, qM, qC, qV, qF, rank, sh, &s_oob, job.dbg, gcls, &fused,
                          ana, R1,
#ifdef EGDST_EMU
                          lcap);      // (the harness keeps poisoned gaps between the LDS regions: the keys stay in the first)
#else
                          4 * lcap);  // (all of the dynamic LDS, as doubles)
#endif
#ifdef EGDST_CENSUS
            cz_sort1_ = wall_clock64();
#endif
            if (s_oob) ENV_FAIL(2714);
            {
                int we = 0, wn = 0, wm = 0;
                // (the unsorted input of a secondary envelope is dead once it is sorted; the primary's input lives in p*)
                job.wM = sM, job.wV = sV, job.wC = sC, job.wcap = (int)W;
                run_walk<0>(&E, job, qM, qC, qV, qF, rank, gcls, &we, &wn, &wm, fused);
                if (tid < WAVE) s_err = we, s_n = wn, s_m = wm;
            }
        }
        __syncthreads();
        STAMP(4);  // walk
#ifdef EGDST_CENSUS
        // (d: why << 24 | bad << 20 | sort time in us)
        if (tid == 0) EG_CENSUS(1, draw, it, jb, job.npts, job.nf | ((job.npts <= lcap ? 0 : 1) << 16),
                                (cz_why_ << 24) | (cz_sh_bad << 20) | (int)min((unsigned long long)0xfffff, (cz_sort1_ - cz_job0_) / 100), wall_clock64() - cz_job0_);
#endif
        if (s_err) ENV_FAIL(s_err);
        if (!primary) {
            if (s_n >= ngridmax) ENV_FAIL(17);  // (:884)
            cnt = s_n;
            if ((size_t)(nall + cnt) > W) ENV_FAIL(part == 1 ? -1 : (compact ? EGDST_E_CAPACITY : 2705));
            for (int i = tid; i < cnt; i += ENV_BS) {
                pM[nall + i] = eM[i];
                pC[nall + i] = eC[i];
                pV[nall + i] = eV[i];
                pF[nall + i] = id;
            }
            if (tid == 0) {
                s_cnt[id] = cnt;
                s_evfa0[id] = evfa0;
            }
            nall += cnt;
            __syncthreads();
        } else {
            if (s_n == 0) ENV_FAIL(16);
            outn = s_n;
            outm = s_m;
        }
    }
    if (part == 1) {  // hand the choice's list over to part 2
        __syncthreads();
        if (tid == 0) {
            b.secn[cell * MS_ND + pid] = s_cnt[pid];
            b.secev[cell * MS_ND + pid] = s_evfa0[pid];
            b.secact[cell * MS_ND + pid] = any;
            b.secevals[cell * MS_ND + pid] = evals;
        }
        return;
    }
    for (int i = outn + 1 + tid; i < hw_rows; i += ENV_BS) oM[i] = oC[i] = oV[i] = 0.0;
    for (int i = outm + tid; i < hw_th; i += ENV_BS) oTH[i] = oD[i] = 0.0;
    // ---- row 0 and lengths (saveoutput :917-952; evf(a0) :730) ------------------------------------
    if (tid == 0) {
        b.thw[tk] = outn + 1;
        b.thhw[tk] = outm;
        oM[0] = b.g.a0;
        oC[0] = 0;
        oV[0] = s_evfa0[(int)oD[0]];
        b.tlen[tk] = outn + 1;
        b.tthlen[tk] = outm;
#ifdef EGDST_EMU
        if (getenv("EGDST_TRACE_ENV"))
            fprintf(stderr, "env it=%d draw=%d outn=%d outm=%d V0=%.17g M1=%.17g C1=%.17g V1=%.17g Mn=%.17g Vn=%.17g\n", it, draw, outn, outm, oV[0],
                    oM[1], oC[1], oV[1], oM[outn], oV[outn]);
#endif
        if (evals) atomicAdd(&b.evals[draw], evals);
#ifdef EGDST_CENSUS
        EG_CENSUS(2, draw, it, outn, outm, 0, cz_why_, wall_clock64() - cz_cell0_);
#endif
#ifdef EGDST_STAMPS5
        atomicAdd((unsigned long long *)(b.dbg + 16 * draw) + 0, wall_clock64() - wg_t0_);
#endif
        {   // algorithmic bytes of this cell: its rows written once (M, C, V; A = M - C is derived on export) and,
            // for the EGM periods, the next-period table of the same state index read once
            unsigned long long by = 24ull * (unsigned long long)(outn + 1) + 16ull * (unsigned long long)outm;
            if (!terminal) {
                const int slot1 = (b.g.nslots == 2) ? ((it + 1) & 1) : (it + 1);
                const size_t k1 = ((size_t)slot1 * b.g.ndraw + draw) * MS_NST + ist;
                by += 24ull * (unsigned long long)b.tlen[k1] + 16ull * (unsigned long long)b.tthlen[k1];
            }
            atomicAdd(&b.algbytes[draw], by);
        }
    }
}

// pass 0 / 1 / 2: one workgroup per job of the launch.  pass 3: the cells the throughput path (k_tp_*, below) left to this
// kernel -- `list` holds their slots, *cnt how many; a small grid loops over them, so that a period with no such cell costs
// a handful of workgroups that return at once instead of one 125-KB-LDS workgroup per cell of the group.
__global__ void __launch_bounds__(ENV_MAXBS, ENV_MINW) ENV_VGPR_ATTR k_envelope(const Batch *bp_, int it, int terminal, int lcap, int pass, int part,
                                                                                const int *list, const int *cnt)
{
    BatchRef b = EG_BATCH_REF(bp_);
    // (ONE inlined copy of the cell's code: the other passes are the loop with a single turn)
    const int n = (pass == 3) ? *cnt : (int)gridDim.x;
    for (int k = blockIdx.x; k < n; k += gridDim.x) {
        eg_envelope_cell(b, it, terminal, lcap, pass == 3 ? 1 : pass, part, pass == 3 ? list[k] : k);
        __syncthreads();  // (the LDS of the cell is reused by the next one)
    }
}

// ---------------------------------------------------------------------------------------------
// THROUGHPUT PATH of the envelope step (models with several choices, batches with many cells per period).
//
// k_envelope is built for latency: one workgroup of 512 threads per cell, the cell's stream resident in ~125 KB of LDS,
// 256 VGPRs -- one workgroup per CU, two waves per SIMD, and the counters of round 2 say what that costs when there are
// thousands of cells per period: 73 % of the wave cycles waiting, 4 % VALU busy, a whole CU slower than one CPU core on this
// step (profiles/r02_f_pmc_C2_ndraw4096.csv).  A batch does not need a fast cell, it needs many cells in flight.  The same
// step as FIVE lean kernels, each with the registers and the LDS of its own phase only, streams in global memory (L2):
//   k_tp_prep  (cell, choice), 256 threads: stop rule (:1100,1150), keep rule (:634), compaction into the choice's list, fold
//              detection and the pieces of a folded list with their extrapolation points (envelope2, :776-835);
//   k_tp_sort  stage 0 (cell, choice with a folded list) / stage 1 (cell), 256 threads: comp1 order (:1240) as ranks from
//              binary searches over the lists' M keys staged in LDS (8 B per point), class words of the walk alongside
//              (blk_rank_sort, the code k_envelope uses for streams that do not fit its LDS);
//   k_tp_walk  stage 0 / stage 1, ONE WAVE per job: the envelope walk (:1285-1532, thresholds :1596-1915) over the sorted
//              stream, never cut into segments -- the batch provides the parallelism, 12-16 such waves share a CU; stage 1
//              writes the period's table (saveoutput, :917-952).
// Nothing new is computed: every phase calls the device functions k_envelope calls, on the same numbers in the same order.
// The path produces a cell only when everything about it is regular; ANY condition that k_envelope would report as an
// error, and every limit of this path (more than TP_NF pieces, a failed draw, an infeasible state), sets b.defer[cell] and the
// cell is done afterwards, from the untouched candidates, by k_envelope itself (pass 1) -- so error codes, their order and
// the partially written tables of failing draws stay exactly k_envelope's.
#if MS_ND > 1
#ifdef EGDST_EMU
#define TP_BS ENV_BS_EMU  // (the harness sizes the scan scratch of the block-wide helpers by its envelope workgroup)
#else
#define TP_BS 256
#endif
static_assert(MS_ND <= TP_NF, "TP_NF must hold one function per choice");
// (the value written is the reason -- any non-zero value defers the cell; -DEGDST_CENSUS logs it)
#define TP_DEFER(why)                         \
    do {                                      \
        if (threadIdx.x == 0) b.defer[cell] = (why); \
        return;                               \
    } while (0)

#ifndef TP_PREP_MINW
#define TP_PREP_MINW 1
#endif
#ifndef TP_SORT_MINW
#define TP_SORT_MINW 8  // (at most 64 VGPRs: eight waves per SIMD; the default build came out at 65)
#endif
// b.defer[cell] may be set by ANOTHER workgroup of the same launch at any moment (the other choice of the cell): the waves
// of a workgroup must not read it each for themselves -- a return that not every wave takes leaves the others at the next
// barrier for ever.  One thread reads, everybody uses what it saw.
#define TP_DEFERRED_UNIFORM(S, cell) \
    ([&]() -> int {                     \
        __syncthreads();                \
        if (threadIdx.x == 0) (S)->res[0] = b.defer[cell]; \
        __syncthreads();                \
        return (S)->res[0];             \
    }())

// static LDS of a workgroup of the path, shared by its phases (the fused kernels run them one after the other)
#ifdef EGDST_EMU
#define TP_WALK_BS ENV_BS_EMU
#define TP_SORT_BS ENV_BS_EMU
#else
#define TP_WALK_BS 256   // threads of a k_tp_walk workgroup: all load the stream, up to four waves walk segments of it
#define TP_SORT_BS 512
#endif
#define TP_SEG_SLICE 32  // cursor entries per walking wave while nf + 2 <= 32 (run_walk: WalkJob.slice0)
#define TP_CURCAP 128    // cursor entries in all: four walking waves of 32, two of 64
static_assert(TP_NF + 2 <= TP_CURCAP, "the cursor arrays must hold one walk of TP_NF functions");
struct TpShared {
    int sh[ENV_MAXBS + 2];                                   // scan scratch of the block-wide helpers
    int fstart[TP_NF], fdims[TP_NF], fcur[TP_CURCAP], fmark[TP_CURCAP];
    int stack[2 * (TP_CURCAP + 2)];
    double evfa0[MS_ND];
    int oob, res[3];
};

// the last kept candidate of every thread of k_tp_prep's one-pass form, for the threads after it
struct TpPrepShared {
    int has[TP_BS];
    double m[TP_BS], c[TP_BS], v[TP_BS];
};
// bxi: cell slot of the group * MS_ND + choice
static __device__ __forceinline__ void tp_prep(BatchRef b, int it, int bxi, TpShared *S, TpPrepShared *X)
{
    int *const sh = S->sh, *const s_fstart = S->fstart;
    int &s_oob = S->oob;
    const int bx_ = bxi / MS_ND, id = bxi % MS_ND;
    const int ist = bx_ % MS_NST, draw = b.order[b.draw0 + bx_ / MS_NST];
    const int tid = threadIdx.x;
    const size_t cell = (size_t)draw * MS_NST + ist;
    TpRec *R = b.tprec + cell * MS_ND + id;
    if (tid == 0) R->active = 0, R->cnt = 0, R->nfold = 0, R->fused = 0, R->evfa0 = 0.0, R->evals = 0ull, s_oob = 0;
    if (b.status[draw]) TP_DEFER(1);  // (k_envelope clears the lengths of a failed draw's cells)
    ms_env E = eg_env(b, draw);
    ms_pv cur;
    cur.it = it, cur.ist = ist, cur.id = 0, cur.cash = cur.savings = cur.shock = 0;
    if (ms_feasible(&E, &cur) != 1) TP_DEFER(2);
    const ProbeOut P = b.probe[cell * MS_ND + id];
    if (!P.active) return;
    __syncthreads();
    const int ngridmax = b.g.ngridmax, ngridm = b.g.ngridm;
    const double mmax = b.g.mmax;
    const size_t W = (size_t)b.g.Cp, wo = cell * (size_t)(MS_ND + 1) * b.g.Cp + (size_t)id * b.g.Cp;
    double *pM = b.pM + wo, *pC = b.pC + wo, *pV = b.pV + wo;
    double *sM = b.sM + wo, *sC = b.sC + wo, *sV = b.sV + wo;
    int *sF = b.sF + wo;
    const size_t co = eg_cand(b, draw, ist, id);
    int nfold = 0, cnt = 0;
    unsigned long long evals = 0;
    double evfa0 = P.evfa0;
    const int navail_ = P.seq ? P.np - 1 : (P.grid ? min(ngridm - 1, ngridmax - 1 - P.ncalls) : 0);
#ifndef TP_PREP_ONEPASS
#define TP_PREP_ONEPASS 1
#endif
    if (TP_PREP_ONEPASS && navail_ + 1 <= ENV_CK * TP_BS) {
        // ---- ONE pass (round 4): a list of at most ENV_CK * TP_BS candidates is read once, ENV_CK consecutive candidates per
        // thread, and everything below works on registers: stop rule (:1100), keep rule (:634) with the statistics of the requested
        // points, fold detection and the pieces of a folded list with their extrapolation points (:776-835).  The earlier form
        // (kept below for longer lists) read the candidates twice and the compacted list twice more; the numbers and the order of
        // every write are the same.
        const int n0 = ENV_CK * tid;
        double vM[ENV_CK], vC[ENV_CK], vV[ENV_CK];
        int sc[ENV_CK], keep[ENV_CK];
#pragma unroll
        for (int k = 0; k < ENV_CK; k++) {
            const int n = n0 + k;
            vM[k] = vC[k] = vV[k] = 0, sc[k] = 0, keep[k] = 0;
            if (n <= navail_) {
                vM[k] = b.cM[co + n], vC[k] = b.cC[co + n], vV[k] = b.cV[co + n];
                if (!P.seq && n >= 1) sc[k] = b.cSt[co + n];
            }
        }
        int nreq = navail_;
        if (!P.seq) {
            int first = navail_ + 1;
#pragma unroll
            for (int k = ENV_CK - 1; k >= 0; k--)
                if (n0 + k >= 1 && n0 + k <= navail_ && !(vM[k] < mmax)) first = n0 + k;
            first = blk_min(first, sh);
            nreq = min(first, navail_);
        } else
            evals = (unsigned long long)P.probe_evals;
        int hard = 0, n12 = 0, ev = 0, mine = 0;
        double lm = 0, lc = 0, lv = 0;
#pragma unroll
        for (int k = 0; k < ENV_CK; k++) {
            const int n = n0 + k;
            if (n > nreq) continue;
            if (P.seq)
                keep[k] = 1;
            else if (n == 0)
                keep[k] = P.np;
            else {
                const int st = eg_sc_status(sc[k]);
                if (st < 0) hard = max(hard, -st);
                n12 |= (st == 1) | ((st == 2) << 1);
                ev += eg_sc_count(sc[k]);
                keep[k] = (st == 0 && isfinite(vM[k]));
            }
            if (keep[k]) mine++, lm = vM[k], lc = vC[k], lv = vV[k];
        }
        X->has[tid] = mine, X->m[tid] = lm, X->c[tid] = lc, X->v[tid] = lv;
        __syncthreads();
        // the kept point before this thread's first one (nearly always the previous thread's last)
        int have = 0;
        double pm = 0, pc = 0, pv = 0;
        if (mine)
            for (int t = tid - 1; t >= 0 && !have; t--)
                if (X->has[t]) pm = X->m[t], pc = X->c[t], pv = X->v[t], have = 1;
        int fold[ENV_CK], myfolds = 0;
        double qc[ENV_CK], qv[ENV_CK];  // C and V of the kept point before a fold (its extrapolation point)
#pragma unroll
        for (int k = 0; k < ENV_CK; k++) {
            fold[k] = 0, qc[k] = qv[k] = 0;
            if (!keep[k]) continue;
            if (have && (pm > vM[k] || pv > vV[k])) fold[k] = 1, qc[k] = pc, qv[k] = pv, myfolds++;
            pm = vM[k], pc = vC[k], pv = vV[k], have = 1;
        }
        int tot;
        const int ex = blk_scan_int(mine | (myfolds << 16), sh, &tot);
        cnt = tot & 0xffff, nfold = (cnt > 1) ? (tot >> 16) : 0;
        int d = ex & 0xffff;
        // (a folded list goes on as its pieces in the s slice; its p slice is where the secondary envelope will be written, and
        //  nothing reads it before that: k_tp_sort stage 0 sorts the pieces, k_envelope works from the candidates)
#pragma unroll
        for (int k = 0; k < ENV_CK; k++)
            if (keep[k]) {
                if ((size_t)d >= W)
                    s_oob = 1;
                else if (nfold == 0)
                    pM[d] = vM[k], pC[d] = vC[k], pV[d] = vV[k];
                d++;
            }
        if (!P.seq) {
            blk_reduce3(&hard, &n12, &ev, sh);
            evals += (unsigned long long)ev + (unsigned long long)P.probe_evals;
            if (hard || (n12 & 1)) TP_DEFER(3);  // a hard error, a zero-consumption signal left over: k_envelope reports them
            if (n12 & 2) evfa0 = -INFINITY;
        }
        __syncthreads();
        if (s_oob) TP_DEFER(4);
        if (nfold > 0) {
            if (id + nfold + 1 > TP_NF) TP_DEFER(5);
            int i = ex & 0xffff, sidx = ex >> 16, lastfold = 0;
#pragma unroll
            for (int k = 0; k < ENV_CK; k++)
                if (keep[k]) {
                    sidx += fold[k];  // pieces closed before this point
                    const int dd = i + sidx;
                    if ((size_t)dd >= W)
                        s_oob = 1;
                    else {
                        sM[dd] = vM[k], sC[dd] = vC[k], sV[dd] = vV[k], sF[dd] = id + sidx;
                        if (fold[k]) {
                            sM[dd - 1] = 1.5 * mmax, sC[dd - 1] = qc[k], sV[dd - 1] = qv[k], sF[dd - 1] = id + sidx - 1;
                            s_fstart[id + sidx] = dd;
                            if (sidx == nfold) lastfold = i;
                        }
                    }
                    if (i == 0) s_fstart[id] = 0;
                    i++;
                }
            lastfold = blk_sum(lastfold, sh);
            if (s_oob || lastfold + (nfold - 1) >= ngridmax) TP_DEFER(6);  // (:823 not enough space: k_envelope reports it)
            __syncthreads();
            for (int f = tid; f < id + nfold + 1; f += TP_BS) R->fstart[f] = (f < id) ? 0 : s_fstart[f];
        }
    } else {
    // ---- stop rule: the first requested point whose returned M stops the stream is itself kept (:1100) -----------------------
    int nreq = 0, fusedstats = 0;
    if (P.seq) {
        nreq = P.np - 1;  // k_fixup stored the kept points of the whole stream in order
        evals = (unsigned long long)P.probe_evals;
    } else {
        const int navail = P.grid ? min(ngridm - 1, ngridmax - 1 - P.ncalls) : 0;
        int first = navail + 1;
        for (int n = 1 + tid; n <= navail; n += TP_BS)
            if (!(b.cM[co + n] < mmax)) first = min(first, n);
        first = blk_min(first, sh);
        nreq = min(first, navail);
        fusedstats = 1;
    }
    // ---- compaction of the kept points (:634) with the statistics of the requested ones -----------------------------------
    int hard = 0, n12 = 0, ev = 0;
    {
        int carry = 0;
        for (int base = 0; base <= nreq; base += ENV_CK * TP_BS) {
            const int n0 = base + ENV_CK * tid;
            double vM[ENV_CK], vC[ENV_CK], vV[ENV_CK];
            int keep[ENV_CK], mine = 0;
#pragma unroll
            for (int k = 0; k < ENV_CK; k++) {
                const int n = n0 + k;
                keep[k] = 0;
                vM[k] = vC[k] = vV[k] = 0;
                if (n <= nreq) {
                    vM[k] = b.cM[co + n], vC[k] = b.cC[co + n], vV[k] = b.cV[co + n];
                    if (P.seq)
                        keep[k] = 1;
                    else if (n == 0)
                        keep[k] = P.np;
                    else {
                        const int sc = b.cSt[co + n], st = eg_sc_status(sc);
                        if (st < 0) hard = max(hard, -st);
                        n12 |= (st == 1) | ((st == 2) << 1);
                        ev += eg_sc_count(sc);
                        keep[k] = (st == 0 && isfinite(vM[k]));
                    }
                }
                mine += keep[k];
            }
            int tot;
            int d = carry + blk_scan_int(mine, sh, &tot);
#pragma unroll
            for (int k = 0; k < ENV_CK; k++)
                if (keep[k]) {
                    if (d < 0 || (size_t)d >= W)
                        s_oob = 1;
                    else
                        pM[d] = vM[k], pC[d] = vC[k], pV[d] = vV[k];
                    d++;
                }
            carry += tot;
        }
        cnt = carry;
    }
    if (fusedstats) {
        blk_reduce3(&hard, &n12, &ev, sh);
        evals += (unsigned long long)ev + (unsigned long long)P.probe_evals;
        if (hard || (n12 & 1)) TP_DEFER(3);  // a hard error, a zero-consumption signal left over: k_envelope reports them
        if (n12 & 2) evfa0 = -INFINITY;
    }
    __syncthreads();
    if (s_oob) TP_DEFER(4);
    // ---- does the list fold back?  then it needs a secondary envelope (:776-913) -------------------------------------------
    if (cnt > 1) {
        for (int i = 1 + tid; i < cnt; i += TP_BS)
            if (pM[i - 1] > pM[i] || pV[i - 1] > pV[i]) nfold++;
        nfold = blk_sum(nfold, sh);
    }
    if (nfold > 0) {
        if (id + nfold + 1 > TP_NF) TP_DEFER(5);
        // the pieces, one constant-extrapolation point appended to every closed one (:822-835)
        int carry = 0, lastfold = 0;
        for (int base = 0; base < cnt; base += TP_BS) {
            const int i = base + tid;
            int fold = 0;
            if (i >= 1 && i < cnt) fold = (pM[i - 1] > pM[i] || pV[i - 1] > pV[i]);
            int tot;
            const int ex = blk_scan(fold, sh, &tot);
            if (i < cnt) {
                const int sidx = carry + ex + fold;  // pieces closed before this point
                const int d = i + sidx;
                if ((size_t)d >= W)
                    s_oob = 1;
                else {
                    sM[d] = pM[i], sC[d] = pC[i], sV[d] = pV[i], sF[d] = id + sidx;
                    if (fold) {
                        sM[d - 1] = 1.5 * mmax, sC[d - 1] = pC[i - 1], sV[d - 1] = pV[i - 1], sF[d - 1] = id + sidx - 1;
                        s_fstart[id + sidx] = d;
                        if (sidx == nfold) lastfold = i;
                    }
                }
                if (i == 0) s_fstart[id] = 0;
            }
            carry += tot;
        }
        lastfold = blk_sum(lastfold, sh);
        if (s_oob || lastfold + (nfold - 1) >= ngridmax) TP_DEFER(6);  // (:823 not enough space: k_envelope reports it)
        __syncthreads();
        for (int f = tid; f < id + nfold + 1; f += TP_BS) R->fstart[f] = (f < id) ? 0 : s_fstart[f];
    }
    }
    if (tid == 0) R->active = 1, R->cnt = cnt, R->nfold = nfold, R->evfa0 = evfa0, R->evals = evals;
}
__global__ void __launch_bounds__(TP_BS, TP_PREP_MINW) k_tp_prep(const Batch *bp_, int it)
{
    __shared__ TpShared S;
    __shared__ TpPrepShared X;
    tp_prep(EG_BATCH_REF(bp_), it, (int)blockIdx.x, &S, &X);
}

// stage 0: the pieces of one folded choice list (secondary envelope); stage 1: the choice lists of a cell (primary).
// lkcap: M keys that fit the dynamic LDS.
// bxi: stage 0: cell slot * MS_ND + choice; stage 1: cell slot
// wcap: points the walk of this stage keeps in LDS -- a longer stream (a degenerate guess stream: thousands of repeated
// points) is k_envelope's, and is not sorted here either; 0: no limit (the walks run on global memory, k_tp_walk_g)
// big (stage 1 only): the second tier of the stage -- the cells whose lists exceed the stream budget of the regular launch (wcap:
// what lets three walks share a CU) but fit the largest one (bigcap) are flagged TP_BIG and listed (biglist, bigcnt) by the
// regular launch and done by a second, small launch with the large budget (big != 0: bxi comes from that list).  On C2 with the
// surveyed credit limit a0 = -5 one cell in sixteen has such lists -- the secondary envelope of a regenerated guess stream adds a
// few hundred rows -- and left to k_envelope they were 55 % of its time (profiles/r04_*).
#define TP_BIG (-2)
static __device__ __forceinline__ void tp_sort(BatchRef b, int it, int stage, int lkcap, int wcap, int bxi, TpShared *S, double *dynlds,
                                               int big = 0, int bigcap = 0, int *biglist = nullptr, int *bigcnt = nullptr)
{
    int *const sh = S->sh, *const s_fstart = S->fstart, *const s_fdims = S->fdims;
    double *const s_evfa0 = S->evfa0;
    int &s_oob = S->oob;
    const int bx_ = stage ? bxi : bxi / MS_ND, id = stage ? 0 : bxi % MS_ND;
    const int ist = bx_ % MS_NST, draw = b.order[b.draw0 + bx_ / MS_NST];
    const int tid = threadIdx.x, TPB = (int)blockDim.x;
    const size_t cell = (size_t)draw * MS_NST + ist;
    {
        const int df = TP_DEFERRED_UNIFORM(S, cell);  // (set by k_tp_prep or an earlier stage of this period; a racing setter is caught by the next stage)
        if (big ? df != TP_BIG : df != 0) return;
        if (big) {
            __syncthreads();  // (every thread has the flag: it is this launch's to clear)
            if (threadIdx.x == 0) b.defer[cell] = 0;
        }
    }
#ifdef EGDST_TPSTAMPS
    const unsigned long long tpk_t0_ = wall_clock64();
#endif
    TpRec *R = b.tprec + cell * MS_ND + id;
    ms_env E = eg_env(b, draw);
    const size_t Wcell = (size_t)(MS_ND + 1) * b.g.Cp;
    const size_t wo = cell * Wcell + (size_t)id * b.g.Cp;
    double *qM = b.qM + wo, *qC = b.qC + wo, *qV = b.qV + wo;
    int *qF = b.qF + wo, *rank = b.rank + wo, *gcls = b.gcls + wo;
    eg_ldsi *fstart = (eg_ldsi *)s_fstart, *fdims = (eg_ldsi *)s_fdims;
    if (tid == 0) s_oob = 0;
    int npts = 0, nf = 0, sec_id = -1;
    double sec_ev = 0;
    const double *iM, *iC, *iV;
    const int *iF;
    if (stage == 0) {
        if (!R->active || R->nfold <= 0) return;
        npts = R->cnt + R->nfold;
        if (wcap > 0 && npts > wcap) TP_DEFER(7);
        nf = id + R->nfold + 1;
        for (int f = tid; f < nf; f += TPB) {
            const int a = R->fstart[f], z = (f + 1 < nf) ? R->fstart[f + 1] : npts;
            fstart[f] = (f < id) ? 0 : a;
            fdims[f] = (f < id) ? 0 : z - a;
        }
        sec_id = id, sec_ev = R->evfa0;
        iM = b.sM + wo, iC = b.sC + wo, iV = b.sV + wo, iF = b.sF + wo;
    } else {
        // the choice lists packed one after another at the front of the cell's p arrays (choice 0's slice starts there already;
        // a later list moves down, never past its own start), as part 2 of k_envelope does
        double *pM = b.pM + wo, *pC = b.pC + wo, *pV = b.pV + wo;
        int *pF = b.pF + wo;
        int any = 0, nall = 0;
        for (int k = 0; k < MS_ND; k++) nall += (R + k)->cnt;
        if (wcap > 0 && nall > wcap) {
            if (!big && biglist && nall <= bigcap) {  // the second tier's (nothing of the cell has been touched yet)
                if (threadIdx.x == 0) {
                    b.defer[cell] = TP_BIG;
                    biglist[atomicAdd((unsigned *)bigcnt, 1u)] = bx_;
                }
                return;
            }
            TP_DEFER(8);
        }
        nall = 0;
        for (int k = 0; k < MS_ND; k++) {
            const TpRec *Rk = R + k;
            const int cntk = Rk->cnt;
            if (Rk->active) any = 1;
            if (tid == 0) s_fstart[k] = nall, s_fdims[k] = cntk, s_evfa0[k] = Rk->evfa0;
            const size_t src = (size_t)k * b.g.Cp;
            if (k > 0 && cntk > 0 && src != (size_t)nall) {
                for (int base = 0; base < cntk; base += TPB) {  // (chunks: read, barrier, write -- the ranges may overlap)
                    const int r = base + tid;
                    double a_ = 0, b2_ = 0, c_ = 0;
                    if (r < cntk) a_ = pM[src + r], b2_ = pC[src + r], c_ = pV[src + r];
                    __syncthreads();
                    if (r < cntk) pM[nall + r] = a_, pC[nall + r] = b2_, pV[nall + r] = c_;
                    __syncthreads();
                }
            }
            for (int r = tid; r < cntk; r += TPB) pF[nall + r] = k;
            nall += cntk;
        }
        if (!any || nall == 0) TP_DEFER(9);  // (errors 14 and 15 of k_envelope)
        npts = nall, nf = MS_ND;
        iM = pM, iC = pC, iV = pV, iF = pF;
    }
    __syncthreads();
#ifdef EGDST_TPSTAMPS
    if (tid == 0) atomicAdd((unsigned long long *)(b.dbg + 16 * draw) + 6, wall_clock64() - tpk_t0_);
#endif
    const int sec_id_ = sec_id;
    const double sec_ev_ = sec_ev;
    auto ana = [&](int g, double x) -> double {  // value of function g before its first point (env_analytic / env_evf)
        const double ev = (sec_id_ >= 0) ? (g == sec_id_ ? sec_ev_ : -INFINITY) : s_evfa0[g];
        if (ev == -INFINITY) return -INFINITY;
        ms_pv cv;
        cv.it = it, cv.ist = ist, cv.id = g, cv.cash = cv.savings = cv.shock = 0;
        return ms_utility(&E, &cv, x - E.a0) + ms_discount(&E, &cv) * ev;
    };
    int fused = 0;
    // dynamic LDS: lkcap M keys (8 B) and class words (4 B), and the inverse permutation of the sort (2 B per entry, the next
    // power of two of lkcap entries: room for the bitonic network of a stream of any length up to lkcap)
    eg_ldsd *lkeys = (eg_ldsd *)dynlds;
    int lpcap = 1;
    while (lpcap < lkcap) lpcap <<= 1;
#ifdef EGDST_EMU
    eg_ldsi *lcls = (eg_ldsi *)(lkeys + lkcap + 8);   // (poisoned gaps between the regions, see k_envelope)
    eg_ldss *lperm = (eg_ldss *)(lcls + lkcap + 16);
    __syncthreads();
    if (threadIdx.x == 0) {
        EG_EMU_UNPOISON_DYN(dynlds);
        EG_EMU_POISON(lkeys + lkcap, 64), EG_EMU_POISON(lcls + lkcap, 64);
    }
    __syncthreads();
#else
    eg_ldsi *lcls = (eg_ldsi *)(lkeys + lkcap);
    eg_ldss *lperm = (eg_ldss *)(lcls + lkcap);
#endif
    blk_rank_sort(npts, nf, iM, iC, iV, iF, fstart, fdims, qM, qC, qV, qF, rank, sh, &s_oob, b.dbg + 16 * draw, gcls, &fused, ana,
                  lkeys, lkcap, lperm, lcls, lpcap);
    __syncthreads();
#ifdef EGDST_TPSTAMPS
    if (tid == 0) atomicAdd((unsigned long long *)(b.dbg + 16 * draw) + 7, wall_clock64() - tpk_t0_);
#endif
    if (s_oob) TP_DEFER(10);
    if (tid == 0) {
        if (stage == 0)
            R->fused = fused;
        else
            b.tpcell[cell].npts = npts, b.tpcell[cell].fused = fused;
    }
}
__global__ void __launch_bounds__(TP_SORT_BS, TP_SORT_MINW) k_tp_sort(const Batch *bp_, int it, int stage, int lkcap, int wcap, int bigcap,
                                                                      int *biglist, int *bigcnt)
{
    EG_DYN_LDS(dynlds);
    __shared__ TpShared S;
    tp_sort(EG_BATCH_REF(bp_), it, stage, lkcap, wcap, (int)blockIdx.x, &S, (double *)dynlds, 0, bigcap, biglist, bigcnt);
}
// One wave per job.  stage 0: secondary envelope of a folded choice list, result written over the list it came from (dead by
// then: its pieces live in the s slice, the sorted stream in the q slice); stage 1: primary envelope into the period's table.
#ifndef TP_WALK_MINW
#define TP_WALK_MINW 3  // (at most 168 VGPRs: three 4-wave workgroups per CU, what the LDS of the stream allows anyway; 172 without)
#endif
// stage 1 also lists the cells that are left to k_envelope (`list`, `cnt`: this (group, period)'s; every cell's stage-1
// workgroup runs exactly once, so a cell is listed exactly once whichever kernel flagged it).
#define TP_DEFER_LISTED(why)                                                \
    do {                                                                    \
        if (threadIdx.x == 0) {                                             \
            b.defer[cell] = (why);                                           \
            list[atomicAdd((unsigned *)cnt, 1u)] = bx_;                     \
            atomicAdd(&b.tpstat[2 * draw + 1], 1u);                         \
        }                                                                   \
        return;                                                             \
    } while (0)
// dynlds: the sorted stream: lcap entries of M, V (8 B), class word (4 B), function id and position list (2 B).  Any number
// of waves: all of them load the stream and classify where the sort did not, wave 0 walks.
// GLOBAL: the stream stays in global memory (long streams: C5's 65 536 points, C3's 12 000): the walk k_envelope runs on such
// streams, but as a kernel of its own -- a few KB of LDS and 168 VGPRs, so that several cells share a CU where k_envelope, sized
// for its LDS-resident streams, takes a CU per cell.
template <bool GLOBAL>
static __device__ __forceinline__ void tp_walk(BatchRef b, int it, int stage, int *list, int *cnt, int lcap, int bxi, TpShared *S, double *dynlds,
                                               int big = 0)
{
    int *const sh = S->sh, *const s_fstart = S->fstart, *const s_fdims = S->fdims, *const s_fcur = S->fcur, *const s_fmark = S->fmark;
    int *const s_stack = S->stack;
    double *const s_evfa0 = S->evfa0;
    const int TPW = (int)blockDim.x;
    const int bx_ = stage ? bxi : bxi / MS_ND, id = stage ? 0 : bxi % MS_ND;
    const int ist = bx_ % MS_NST, draw = b.order[b.draw0 + bx_ / MS_NST];
    const int tid = threadIdx.x;
    const size_t cell = (size_t)draw * MS_NST + ist;
    {
        const int df = TP_DEFERRED_UNIFORM(S, cell);
        if (df == TP_BIG && !big) return;  // the second tier of the stage decides about this cell (and lists it if it gives up)
        if (df) {
            if (stage == 1 && tid == 0) list[atomicAdd((unsigned *)cnt, 1u)] = bx_, atomicAdd(&b.tpstat[2 * draw + 1], 1u);
            return;
        }
    }
    TpRec *R = b.tprec + cell * MS_ND + id;
    if (stage == 0 && (!R->active || R->nfold <= 0)) return;
#ifdef EGDST_TPSTAMPS_WALK  // diagnostic: dbg as 8 x u64 per draw: stage*3 + {0 set-up and load, 1 the walk, 2 the rest}; 6, 7: walks per stage
    unsigned long long tw_t_ = wall_clock64();
#define TWST(k) do { if (tid == 0) { const unsigned long long n_ = wall_clock64(); atomicAdd((unsigned long long *)(b.dbg + 16 * draw) + stage * 3 + (k), n_ - tw_t_); tw_t_ = n_; } } while (0)
    if (tid == 0) atomicAdd((unsigned long long *)(b.dbg + 16 * draw) + 6 + stage, 1ull);
#else
#define TWST(k)
#endif
    ms_env E = eg_env(b, draw);
    const int slot = (b.g.nslots == 2) ? (it & 1) : it;
    const size_t tk = ((size_t)slot * b.g.ndraw + draw) * MS_NST + ist;
    const size_t Wcell = (size_t)(MS_ND + 1) * b.g.Cp;
    const size_t wo = cell * Wcell + (size_t)id * b.g.Cp;
    const int compact = b.g.Cp < b.g.ngridmax;
    eg_ldsi *fstart = (eg_ldsi *)s_fstart, *fdims = (eg_ldsi *)s_fdims;
    WalkJob job;
    job.it = it, job.ist = ist;
    job.ocap = b.g.Cp;
    job.e13 = compact ? EGDST_E_CAPACITY : 13;
    job.nthrhmax = b.g.nthrhmax;
    job.stackcap = 2 * (TP_CURCAP + 2);
    job.curcap = TP_CURCAP, job.slice0 = TP_SEG_SLICE;
    job.fstart = fstart, job.dims = fdims;
    job.cur = (eg_ldsi *)s_fcur, job.mark = (eg_ldsi *)s_fmark;
    job.evfa0 = (const eg_ldsd *)s_evfa0;
    job.stack = (eg_ldsi *)s_stack;
    job.dbg = b.dbg + 16 * draw;
    job.noseg = b.noseg;
    job.sh = sh;
    job.segstat = b.segstat + 2 * (size_t)draw;
    job.klog = nullptr, job.kcnt = nullptr, job.kcap = 0;
    // scratch rows for the segments of a cut walk: the s arrays -- the unsorted pieces of stage 0, dead once they are sorted
    job.wM = b.sM + wo, job.wV = b.sV + wo, job.wC = b.sC + wo, job.wcap = stage ? (int)Wcell : b.g.Cp;
    int classified = 0, hw_rows = 0, hw_th = 0;
    unsigned long long evals = 0;
    double *oM = b.tM + tk * b.g.Sp, *oC = b.tC + tk * b.g.Sp, *oV = b.tV + tk * b.g.Sp;
    double *oTH = b.tTH + tk * b.g.nthrhmax, *oD = b.tD + tk * b.g.nthrhmax;
    if (stage == 0) {
        job.npts = R->cnt + R->nfold;
        job.nf = id + R->nfold + 1;
        for (int f = tid; f < job.nf; f += TPW) {
            const int a = R->fstart[f], z = (f + 1 < job.nf) ? R->fstart[f + 1] : job.npts;
            fstart[f] = (f < id) ? 0 : a;
            fdims[f] = (f < id) ? 0 : z - a;
        }
        job.sec_id = id, job.sec_ev = R->evfa0;
        job.cap = b.g.Cp;
        job.og = b.pM + wo, job.ov = b.pV + wo, job.oc = b.pC + wo;
        const size_t eto = (cell * MS_ND + id) * (size_t)b.g.nthrhmax;
        job.oth = b.eTH + eto, job.oix = b.eIX + eto;
        classified = R->fused;
    } else {
        int nall = 0;
        for (int k = 0; k < MS_ND; k++) {
            const TpRec *Rk = R + k;
            if (tid == 0) s_fstart[k] = nall, s_fdims[k] = Rk->cnt, s_evfa0[k] = Rk->evfa0;
            nall += Rk->cnt;
            evals += Rk->evals;
        }
        job.npts = b.tpcell[cell].npts;
        if (job.npts != nall) TP_DEFER_LISTED(11);  // (never expected: the sort packed exactly these lists)
        job.nf = MS_ND;
        job.sec_id = -1, job.sec_ev = 0;
        job.cap = (int)Wcell;
        job.og = oM + 1, job.ov = oV + 1, job.oc = oC + 1, job.oth = oTH, job.oix = oD;
        classified = b.tpcell[cell].fused;
        // rows of this cell that may hold leftovers of an earlier period or solve (see k_envelope); until the cell is complete
        // the marks say "unknown", so that k_envelope, should the cell still be deferred, clears everything past its own end
        hw_rows = b.thw[tk], hw_th = b.thhw[tk];
        __syncthreads();  // (every lane has read the marks)
        if (tid == 0) b.thw[tk] = b.g.Sp, b.thhw[tk] = b.g.nthrhmax;
    }
    // The stream goes to LDS first: at a crossing the walk follows chains of dependent reads (position list -> point ->
    // neighbour of the other function ...), a few microseconds each from global memory under load -- the one-wave walks
    // over global memory took 230-440 us per launch (profiles/r03_*), nearly all of it in four or five such events.  The
    // consumption column stays where it is: the walk copies it to the rows it keeps and reads it at a kink only.
    // A stream that does not fit (a degenerate guess stream with thousands of points) is k_envelope's.
    int we = 0, wn = 0, wm = 0;
#ifdef EGDST_CENSUS
    const unsigned long long cz_w0_ = wall_clock64();
    if (tid == 0) cz_sh_fb = 0;
    __syncthreads();
#endif
    if (GLOBAL) {
        TWST(0);
        run_walk<0, true>(&E, job, b.qM + wo, b.qC + wo, b.qV + wo, b.qF + wo, b.rank + wo, b.gcls + wo, &we, &wn, &wm, classified);
    } else {
    if (job.npts > lcap || job.npts >= 65536) {
        if (stage == 1) TP_DEFER_LISTED(12);
        TP_DEFER(12);
    }
#ifdef EGDST_EMU
    eg_ldsd *Lm = (eg_ldsd *)dynlds, *Lv = Lm + lcap + 8;
    eg_ldsi *Lc = (eg_ldsi *)(Lv + lcap + 8);
    eg_ldss *Lf = (eg_ldss *)(Lc + lcap + 16), *Lp = Lf + lcap + 32;
    __syncthreads();
    if (threadIdx.x == 0) {
        EG_EMU_UNPOISON_DYN(dynlds);
        EG_EMU_POISON(Lm + lcap, 64), EG_EMU_POISON(Lv + lcap, 64), EG_EMU_POISON(Lc + lcap, 64), EG_EMU_POISON(Lf + lcap, 64);
    }
    __syncthreads();
#else
    eg_ldsd *Lm = (eg_ldsd *)dynlds, *Lv = Lm + lcap;
    eg_ldsi *Lc = (eg_ldsi *)(Lv + lcap);
    eg_ldss *Lf = (eg_ldss *)(Lc + lcap), *Lp = Lf + lcap;
#endif
    {
        const double *gM = b.qM + wo, *gV = b.qV + wo;
        const int *gF = b.qF + wo, *gR = b.rank + wo, *gC = b.gcls + wo;
        for (int i0 = tid; i0 < job.npts; i0 += 4 * TPW) {  // (four rounds of loads in flight, then the stores)
            double tm[4], tv[4];
            int tc[4], tf[4], tr[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = i0 + k * TPW;
                tm[k] = tv[k] = 0, tc[k] = tf[k] = tr[k] = 0;
                if (i < job.npts) tm[k] = gM[i], tv[k] = gV[i], tc[k] = gC[i], tf[k] = gF[i], tr[k] = gR[i];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = i0 + k * TPW;
                if (i < job.npts) Lm[i] = tm[k], Lv[i] = tv[k], Lc[i] = tc[k], Lf[i] = (unsigned short)tf[k], Lp[i] = (unsigned short)tr[k];
            }
        }
    }
    __syncthreads();
    TWST(0);
    run_walk<2, true>(&E, job, Lm, b.qC + wo, Lv, Lf, Lp, Lc, &we, &wn, &wm, classified);
    }
    TWST(1);
#ifdef EGDST_CENSUS
    if (tid == 0) EG_CENSUS(6, draw, it, stage | (big << 4), job.npts, job.nf | (cz_sh_fb << 16), wn, wall_clock64() - cz_w0_);
#endif
    __syncthreads();  // (the results are wave 0's: hand them to the other waves of a fused kernel)
    if (tid == 0) S->res[0] = we, S->res[1] = wn, S->res[2] = wm;
    __syncthreads();
    we = S->res[0], wn = S->res[1], wm = S->res[2];
    if (stage == 0) {
#ifdef EGDST_CENSUS
        if ((we || wn >= b.g.ngridmax) && tid == 0) EG_CENSUS(5, draw, it, 0, we, job.npts, wn, 0);
#endif
        if (we || wn >= b.g.ngridmax) TP_DEFER(13);  // (:884)
        if (tid == 0) R->cnt = wn;
        TWST(2);
        return;
    }
#ifdef EGDST_CENSUS
    if ((we || wn == 0) && tid == 0) EG_CENSUS(5, draw, it, 1, we, job.npts, wn, 0);
#endif
    if (we || wn == 0) TP_DEFER_LISTED(14);  // (wn == 0: error 16 of k_envelope)
    const int outn = wn, outm = wm;
    for (int i = outn + 1 + tid; i < hw_rows; i += TPW) oM[i] = oC[i] = oV[i] = 0.0;
    for (int i = outm + tid; i < hw_th; i += TPW) oTH[i] = oD[i] = 0.0;
    // ---- row 0 and lengths (saveoutput :917-952; evf(a0) :730) ------------------------------------
    if (tid == 0) {
        b.thw[tk] = outn + 1;
        b.thhw[tk] = outm;
        oM[0] = b.g.a0;
        oC[0] = 0;
        oV[0] = s_evfa0[(int)oD[0]];
        b.tlen[tk] = outn + 1;
        b.tthlen[tk] = outm;
        if (evals) atomicAdd(&b.evals[draw], evals);
        atomicAdd(&b.tpstat[2 * draw], 1u);
        TWST(2);
        unsigned long long by = 24ull * (unsigned long long)(outn + 1) + 16ull * (unsigned long long)outm;  // (see k_envelope)
        const int slot1 = (b.g.nslots == 2) ? ((it + 1) & 1) : (it + 1);
        const size_t k1 = ((size_t)slot1 * b.g.ndraw + draw) * MS_NST + ist;
        by += 24ull * (unsigned long long)b.tlen[k1] + 16ull * (unsigned long long)b.tthlen[k1];
        atomicAdd(&b.algbytes[draw], by);
    }
}
__global__ void __launch_bounds__(TP_WALK_BS, TP_WALK_MINW) k_tp_walk(const Batch *bp_, int it, int stage, int *list, int *cnt, int lcap)
{
    EG_DYN_LDS(dynlds);
    __shared__ TpShared S;
    tp_walk<false>(EG_BATCH_REF(bp_), it, stage, list, cnt, lcap, (int)blockIdx.x, &S, (double *)dynlds);
}
// The second tier of stage 1 in ONE launch: a workgroup sorts a listed cell's lists and walks them (the sort's 12 B per point,
// then the walk's 24 B per point, in the same dynamic LDS).  Two launches -- k_tp_sort_big, then a walk kernel -- put two more
// kernel latencies and launch gaps, ~110 us, on every group's chain every period for a handful of cells (tests/diag/gpu_group_finish.py).
// TP_SORT_BS threads: the sort's size; the walk loads with all of them and walks with up to four waves as it does elsewhere.
__global__ void __launch_bounds__(TP_SORT_BS, TP_WALK_MINW) k_tp_big(const Batch *bp_, int it, int *list, int *cnt, int cap, const int *biglist,
                                                                     const int *bigcnt)
{
    EG_DYN_LDS(dynlds);
    __shared__ TpShared S;
    const int n = *bigcnt;
    for (int k = blockIdx.x; k < n; k += gridDim.x) {
        tp_sort(EG_BATCH_REF(bp_), it, 1, cap, cap, biglist[k], &S, (double *)dynlds, 1);
        __threadfence_block();  // (the sorted stream goes through global memory from one phase to the next)
        __syncthreads();
        tp_walk<false>(EG_BATCH_REF(bp_), it, 1, list, cnt, cap, biglist[k], &S, (double *)dynlds, 1);
        __syncthreads();  // (the LDS of the cell is reused by the next one)
    }
}
#ifdef EGDST_WITH_TP_LONG  // diagnostic builds only (tests/diag/gpu_tp_long.py): measured not faster than k_envelope's global walk in round 3
#ifdef EGDST_EMU
#define TP_WALKG_BS ENV_BS_EMU
#else
#define TP_WALKG_BS 512  // eight walking waves for the long streams
#endif
__global__ void __launch_bounds__(TP_WALKG_BS, TP_WALK_MINW) k_tp_walk_g(const Batch *bp_, int it, int stage, int *list, int *cnt)
{
    __shared__ TpShared S;
    tp_walk<true>(EG_BATCH_REF(bp_), it, stage, list, cnt, 0, (int)blockIdx.x, &S, nullptr);
}
#endif

// (Measured and not kept: the same phases FUSED into two kernels -- per (cell, choice) the list, its sort and its secondary
//  envelope; per cell the sort of the lists and the primary envelope -- so that a launch costs the slowest chain of phases
//  instead of the sum of the slowest workgroups of five launches.  C2 x 4096: 202 ms against 184 ms: a fused workgroup holds
//  256 threads' registers and the walk's LDS while one of its waves walks, and the period's chain is no shorter, 512 + 348 us
//  against 40 + 113 + 375 + 231 + 142 us.)
#endif  // MS_ND > 1

// ---------------------------------------------------------------------------------------------
// Single-choice models (MS_ND == 1, e.g. the Deaton family): in the regular case the "envelope" of a cell is the stop
// rule (:1100,1150), the keep rule (:634) and the removal of repeated grid values (:1290-1298) -- a stream compaction with
// nothing sequential in it, which one workgroup per cell (k_envelope) does at the speed of one CU (C4: 65 536 points,
// 0.58 ms per period, 77 % of a solve).  k_env1 spreads a cell over TILES of E1_TILE candidates, one workgroup each, and does
// the whole step in ONE pass over the candidates (rounds 2-3 had four kernels that read them three times: 146 us per period
// of C4 x 32 against the 27 us its bytes take):
//   * a tile finds its first candidate whose returned M stops the stream, counts the rows it will write (kept, and not a
//     repeat of the previous kept grid value), its evaluations, and whether anything irregular turns up (a fold, a list out
//     of order, a hard error, a zero-consumption signal left over);
//   * it publishes (rows, stop seen, irregular) in its descriptor and reads the descriptors of the tiles before it (decoupled
//     look-back: those have smaller block indices, so they are resident or done; the wait is bounded all the same, and a
//     tile that gives up marks the cell irregular): no stop before it -> its rows go to the table at the prefix of the counts;
//   * the LAST tile of a cell to finish (a counter) sums up: an irregular cell, an empty one (error 15), a full grid (error
//     13), no room for the threshold (error 20) or the compact capacity are k_envelope's (pass 1, from the untouched candidates:
//     b.defer; the rows already written are covered by the high-water mark); otherwise row 0, the lengths, the counters and the
//     rows past the new end.
// Descriptors carry the period's tag (it + 1), so nothing is cleared between periods; the host clears them once per solve.
#if MS_ND == 1
#ifdef EGDST_EMU
#define E1_BS ENV_BS_EMU
#else
#define E1_BS 256
#endif
#define E1_PT 4                    // consecutive candidates of a thread
#define E1_TILE (E1_BS * E1_PT)    // candidates of a workgroup
#define E1_D_STOP (1ull << 24)
#define E1_D_IRR (1ull << 25)
#define E1_SPIN_MAX (1 << 22)      // polls of one descriptor before a tile gives up (never expected: seconds)
struct Env1Tile {  // per (schedule slot, state, tile)
    unsigned long long desc;   // [63:32] tag of the period, [25] irregular, [24] stop seen, [23:12] zero-consumption signals, [11:0] rows
    unsigned long long evals;  // evaluations of the tile's candidates (written before desc)
};
// Everything one workgroup tells another goes through agent-scope atomics (they are performed where all XCDs see them), ordered by
// a wait for the issuing wave's own memory operations.  NOT through __threadfence(): on this part a release fence writes back the
// XCD's whole L2 and an acquire fence invalidates it, and with 2048 workgroups streaming rows through those L2s three fences per
// workgroup made the kernel 497 us per launch on C4 x 32 (measured, gpurun_out/trace_r04b) -- nothing but the atomics is shared here.
#ifdef EGDST_EMU
#define E1_LOAD(p) __atomic_load_n((p), __ATOMIC_ACQUIRE)
#define E1_STORE(p, v) __atomic_store_n((p), (v), __ATOMIC_RELEASE)
#define E1_WAIT() __sync_synchronize()
#else
#define E1_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define E1_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define E1_WAIT()                          \
    do {                                   \
        __asm__ volatile("" ::: "memory"); \
        __builtin_amdgcn_s_waitcnt(0);     \
        __asm__ volatile("" ::: "memory"); \
    } while (0)
#endif
static __device__ __forceinline__ bool e1_kept(BatchRef b, const ProbeOut &P, size_t co, int n, int terminal)
{
    if (terminal) return true;
    if (n == 0) return P.np != 0;
    return eg_sc_status(b.cSt[co + n]) == 0 && isfinite(b.cM[co + n]);
}
// exclusive scan of one int per thread over the workgroup; *total = sum
static __device__ __forceinline__ int e1_scan(int v, int *sh, int *total)
{
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int o = 1; o < E1_BS; o <<= 1) {
        const int t = (tid >= o) ? sh[tid - o] : 0;
        __syncthreads();
        sh[tid] += t;
        __syncthreads();
    }
    *total = sh[E1_BS - 1];
    const int ex = sh[tid] - v;
    __syncthreads();
    return ex;
}

__global__ void __launch_bounds__(E1_BS) k_env1(const Batch *bp_, int it, int terminal, Env1Tile *tiles, unsigned *done, int nb, unsigned tag,
                                                int defer_all /* tests: 1 treat every cell as irregular, 2 find it irregular late */)
{
    BatchRef b = EG_BATCH_REF(bp_);
    __shared__ int sh[E1_BS], s_has[E1_BS];
    __shared__ double s_lm[E1_BS], s_lv[E1_BS];
    __shared__ int s_first, s_flags, s_n2, s_pre, s_dead, s_last, s_stoptile;
    __shared__ unsigned long long s_ev;
    const int cellslot = blockIdx.y, ist = cellslot % MS_NST, draw = b.order[b.draw0 + cellslot / MS_NST];
    const size_t cell = (size_t)draw * MS_NST + ist, sslot = (size_t)b.draw0 * MS_NST + cellslot;
    const int tid = threadIdx.x, blk = blockIdx.x;
    const int slot = (b.g.nslots == 2) ? (it & 1) : it;
    const size_t tk = ((size_t)slot * b.g.ndraw + draw) * MS_NST + ist;
    // (the three early exits are the same for every tile of a cell: no tile of such a cell reaches the counter)
    if (b.status[draw]) {
        if (blk == 0 && tid == 0) b.tlen[tk] = b.tthlen[tk] = 0;
        return;
    }
    ms_env E = eg_env(b, draw);
    ms_pv cur;
    cur.it = it, cur.ist = ist, cur.id = 0, cur.cash = cur.savings = cur.shock = 0;
    if (ms_feasible(&E, &cur) != 1) {
        if (blk == 0 && tid == 0) b.tlen[tk] = b.tthlen[tk] = 0;
        return;
    }
    const ProbeOut P = b.probe[cell * MS_ND];
    if (!P.active || P.seq || defer_all == 1) {  // a regenerated stream, an inactive choice (error 14): k_envelope's
        if (blk == 0 && tid == 0) b.defer[cell] = 1;
        return;
    }
    Env1Tile *T = tiles + sslot * (size_t)nb;
    const unsigned long long dtag = (unsigned long long)tag << 32;
    const size_t co = eg_cand(b, draw, ist, 0);
    // candidates 0 .. navail may have been requested; those after the first whose M stops the stream were not
    const int navail = terminal ? b.g.ngridm - 1 : (P.grid ? min(b.g.ngridm - 1, b.g.ngridmax - 1 - P.ncalls) : 0);
    const int lo = blk * E1_TILE, hi = min(navail + 1, lo + E1_TILE);
    if (tid == 0) s_first = 0x7fffffff, s_flags = 0, s_n2 = 0, s_ev = 0, s_pre = 0, s_dead = 0, s_last = 0, s_stoptile = 0x7fffffff;
    const int a = lo + tid * E1_PT;
    double m[E1_PT], v[E1_PT], c[E1_PT];
    int st[E1_PT];
#pragma unroll
    for (int k = 0; k < E1_PT; k++) {
        const int n = a + k;
        m[k] = v[k] = c[k] = 0, st[k] = 0;
        if (n < hi) {
            m[k] = b.cM[co + n], v[k] = b.cV[co + n], c[k] = b.cC[co + n];
            if (n >= 1 && !terminal) st[k] = b.cSt[co + n];
        }
    }
    __syncthreads();
    if (!terminal) {
        int f = 0x7fffffff;
#pragma unroll
        for (int k = E1_PT - 1; k >= 0; k--)
            if (a + k < hi && a + k >= 1 && !(m[k] < b.g.mmax)) f = a + k;
        if (f != 0x7fffffff) atomicMin(&s_first, f);
    }
    __syncthreads();
    const int first = s_first, nreq = min(first, navail);  // (as if no tile before this one stopped the stream: see below)
    // ---- this thread's candidates: kept?  the last kept one for the threads after it --------------------
    bool kept[E1_PT];
    int has = 0;
    double lm = 0, lv = 0;
#pragma unroll
    for (int k = 0; k < E1_PT; k++) {
        const int n = a + k;
        kept[k] = false;
        if (n >= hi || n > nreq) continue;
        kept[k] = terminal ? true : (n == 0 ? P.np != 0 : (eg_sc_status(st[k]) == 0 && isfinite(m[k])));
        if (kept[k]) has = 1, lm = m[k], lv = v[k];
    }
    s_has[tid] = has, s_lm[tid] = lm, s_lv[tid] = lv;
    __syncthreads();
    // the previous kept candidate of the cell, if any: in this tile (nearly always the thread before), else further back
    bool have = false;
    double pm = 0, pv = 0;
    if (has) {
        for (int t = tid - 1; t >= 0 && !have; t--)
            if (s_has[t]) pm = s_lm[t], pv = s_lv[t], have = true;
        for (int q = lo - 1; q >= 0 && !have; q--)
            if (e1_kept(b, P, co, q, terminal)) pm = b.cM[co + q], pv = b.cV[co + q], have = true;
    }
    int cnt = 0, flags = 0, n2 = 0, rowmask = 0;
    unsigned long long ev = 0;
#pragma unroll
    for (int k = 0; k < E1_PT; k++) {
        const int n = a + k;
        if (n >= hi || n > nreq) continue;
        if (n >= 1 && !terminal) {
            const int s_ = eg_sc_status(st[k]);
            if (s_ < 0 || s_ == 1) flags |= 1;  // hard error / c1<=0 left over: k_envelope reports it
            if (s_ == 2) n2 += 1;
            ev += (unsigned long long)eg_sc_count(st[k]);
        }
        if (!kept[k]) continue;
        if (!have) {  // first kept point of the cell
            rowmask |= 1 << k, cnt++;
            pm = m[k], pv = v[k], have = true;
            continue;
        }
        if (!terminal && (pm > m[k] || pv > v[k])) flags |= 1;            // the list folds back: secondary envelope
        if (pm > m[k] || (pm == m[k] && pv < v[k])) flags |= 1;           // not in comp1 order: general sort
        if (m[k] != pm) rowmask |= 1 << k, cnt++;
        pm = m[k], pv = v[k];
    }
    if (defer_all == 2) flags |= 1;  // (tests: every cell is found irregular AFTER its tiles have written their rows)
    int total;
    const int ex = e1_scan(cnt, sh, &total);
    if (flags) atomicOr(&s_flags, flags);
    if (n2) atomicAdd(&s_n2, n2);
    if (ev) atomicAdd(&s_ev, ev);
    __syncthreads();
    // ---- publish, then look back -----------------------------------------------------------------
    unsigned long long mine = dtag | (unsigned long long)total | ((unsigned long long)s_n2 << 12) | (first != 0x7fffffff ? E1_D_STOP : 0ull) |
                              (s_flags ? E1_D_IRR : 0ull);
    if (tid == 0) {
        E1_STORE(&T[blk].evals, s_ev);
        E1_WAIT();
        E1_STORE(&T[blk].desc, mine);
    }
    if (tid < WAVE) {  // the first wave: lane j reads the descriptors j, j + WAVE, ... of the tiles before this one
        int pre = 0, dead = 0, gaveup = 0;
        for (int j0 = 0; j0 < blk && !dead && !gaveup; j0 += WAVE) {
            const int j = j0 + tid;
            unsigned long long d = dtag;
            if (j < blk) {
                int spins = 0;
                d = E1_LOAD(&T[j].desc);
                while ((d >> 32) != (unsigned long long)tag && ++spins < E1_SPIN_MAX) {
#ifndef EGDST_EMU
                    __builtin_amdgcn_s_sleep(2);
#endif
                    d = E1_LOAD(&T[j].desc);
                }
                if ((d >> 32) != (unsigned long long)tag) gaveup = 1, d = dtag;
            }
            int cj = (int)(d & 0xfffull);
            for (int o = WAVE / 2; o > 0; o >>= 1) cj += __shfl_xor(cj, o);
            pre += cj;
            dead = __any((d & E1_D_STOP) != 0);
            gaveup = __any(gaveup);
        }
        if (tid == 0) {
            s_pre = pre, s_dead = dead;
            if (gaveup) {  // (the tiles after this one used its rows and its stop flag only; the last tile reads the rest)
                mine |= E1_D_IRR;
                E1_STORE(&T[blk].desc, mine);
                s_dead = 1;
            }
        }
    }
    __syncthreads();
    double *oM = b.tM + tk * b.g.Sp, *oC = b.tC + tk * b.g.Sp, *oV = b.tV + tk * b.g.Sp;
    if (!s_dead) {  // no tile before this one stopped the stream: its rows, in place
        int row = 1 + s_pre + ex;
#pragma unroll
        for (int k = 0; k < E1_PT; k++)
            if (rowmask & (1 << k)) {
                if (row < b.g.Sp) oM[row] = m[k], oC[row] = c[k], oV[row] = v[k];
                row++;
            }
    }
    // ---- the last tile of the cell to get here finishes the cell ------------------------------------
    __syncthreads();
    if (tid == 0) {
        E1_WAIT();  // (this tile's descriptor is where the last tile will look for it)
        s_last = ((atomicAdd(&done[sslot], 1u) + 1u) % (unsigned)nb == 0u);
    }
    __syncthreads();
    if (!s_last) return;
    // the tiles up to and including the first that saw a stop count; the others wrote nothing
    for (int j = tid; j < nb; j += E1_BS)
        if (E1_LOAD(&T[j].desc) & E1_D_STOP) atomicMin(&s_stoptile, j);
    if (tid == 0) s_flags = 0, s_n2 = 0, s_ev = 0;
    __syncthreads();
    int outn = 0;
    {
        const int lastt = min(nb - 1, s_stoptile);
        int rows = 0, fl = 0, z2 = 0;
        unsigned long long e_ = 0;
        for (int j = tid; j <= lastt; j += E1_BS) {
            const unsigned long long d = E1_LOAD(&T[j].desc);
            rows += (int)(d & 0xfffull), z2 += (int)((d >> 12) & 0xfffull), fl |= (d & E1_D_IRR) ? 1 : 0;
            if ((d >> 32) != (unsigned long long)tag) fl |= 1;  // (never expected)
            e_ += E1_LOAD(&T[j].evals);
        }
        int tot;
        (void)e1_scan(rows, sh, &tot);
        outn = tot;
        if (fl) atomicOr(&s_flags, 1);
        if (z2) atomicAdd(&s_n2, z2);
        if (e_) atomicAdd(&s_ev, e_);
        __syncthreads();
    }
    // anything k_envelope would treat differently goes to k_envelope: irregular lists, no point at all (error 15), a
    // full grid (error 13), no room for the threshold (error 20), the compact capacity.  Rows of the table may have been
    // written: the marks say "unknown", so that k_envelope clears everything past its own end.
    if (s_flags || outn == 0 || outn >= b.g.ngridmax || outn > b.g.Cp || 1 >= b.g.nthrhmax) {
        if (tid == 0) b.defer[cell] = 1, b.thw[tk] = b.g.Sp;
        return;
    }
    double *oTH = b.tTH + tk * b.g.nthrhmax, *oD = b.tD + tk * b.g.nthrhmax;
    // rows past the new end of the table are zero (see k_envelope)
    const int hw_rows = b.thw[tk], hw_th = b.thhw[tk];
    for (int i = outn + 1 + tid; i < hw_rows; i += E1_BS) oM[i] = oC[i] = oV[i] = 0.0;
    for (int i = 1 + tid; i < hw_th; i += E1_BS) oTH[i] = oD[i] = 0.0;
    __syncthreads();  // (every thread has read the old marks)
    if (tid == 0) {
        oTH[0] = b.g.a0;
        oD[0] = 0;
        oM[0] = b.g.a0;
        oC[0] = 0;
        oV[0] = (s_n2 > 0) ? -INFINITY : P.evfa0;
        b.tlen[tk] = outn + 1;
        b.tthlen[tk] = 1;
        b.thw[tk] = outn + 1, b.thhw[tk] = 1;
        const unsigned long long evals = s_ev + (terminal ? 0ull : (unsigned long long)P.probe_evals);
        if (evals) atomicAdd(&b.evals[draw], evals);
        unsigned long long by = 24ull * (unsigned long long)(outn + 1) + 16ull;
        if (!terminal) {
            const int slot1 = (b.g.nslots == 2) ? ((it + 1) & 1) : (it + 1);
            const size_t k1 = ((size_t)slot1 * b.g.ndraw + draw) * MS_NST + ist;
            by += 24ull * (unsigned long long)b.tlen[k1] + 16ull * (unsigned long long)b.tthlen[k1];
        }
        atomicAdd(&b.algbytes[draw], by);
    }
}
#endif

// ---------------------------------------------------------------------------------------------
// Forward simulation, one lane per agent (egdst_simulator.c:204-383, policy :145-199, output :122-143).
struct SimArgs {
    int draw, nsim, rndtype, nout;   // first draw of the launch (blockIdx.y counts on from it)
    int batch;                 // 1: several draws per launch, failed draws are skipped (their moments are NaN)
    const double *init;        // [nsim x 2] column-major
    const double *randstream;  // uniforms, or nullptr: generated from `seed` (eg_uniform)
    unsigned long long seed;
    double *sims;              // [draws of the launch][nout x nt x nsim]
    int *err;
};

// The counter-based uniform generator of the batched simulation (include/egdst.h: egdst_uniform is the same on the
// host): number k of stream `seed` is the splitmix64 output for the state seed + (k+1)*golden, top 53 bits.
static __host__ __device__ __forceinline__ double eg_uniform(unsigned long long seed, unsigned long long k)
{
    unsigned long long z = seed + (k + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * 0x1p-53;
}

__global__ void __launch_bounds__(GRID_BS) k_simulate(const Batch *bp_, SimArgs a)
{
    BatchRef b = EG_BATCH_REF(bp_);
    const int isim = blockIdx.x * GRID_BS + threadIdx.x;
    if (isim >= a.nsim) return;
    const int nt = b.g.nt, draw = a.draw + (int)blockIdx.y;
    if (a.batch && b.status[draw]) return;  // (a failed draw has no complete solution: its paths stay NaN)
    a.sims += (size_t)blockIdx.y * a.nout * nt * a.nsim;
    ms_env E = eg_env(b, draw);
    const unsigned long long rbase = a.rndtype == 1 ? 0ull : 4ull * (unsigned long long)nt * (unsigned long long)isim;
    const double *rs = a.randstream ? a.randstream + rbase : nullptr;
#define EG_RS(k) (rs ? rs[k] : eg_uniform(a.seed, rbase + (unsigned long long)(k)))
    const int ist0 = (int)a.init[isim] - 1;
    const double m0 = a.init[a.nsim + isim];
    if (ist0 < 0 || ist0 >= MS_NST) return;
    if (m0 < b.g.a0 || m0 > b.g.mmax) return;
    ms_pv cp, np_;
    double mu = NAN, sigma = NAN, eqs[MS_NEQ + 1];
    int irnd = 0, terr = 0;
    cp.it = 0;
    cp.ist = ist0;
    cp.id = 0;
    cp.cash = m0;
    cp.savings = 0;
    cp.shock = NAN;
#if MS_NCONT > 0
    // continuous states (egdst_simulator.c:91-92,237-246): the model functions read states and decisions BY VALUE
    cp.byval = np_.byval = 1;
    for (int k = 0; k < MS_NNST; k++) cp.st[k] = ms_states[ist0 + k * MS_NST], np_.st[k] = 0;
    for (int k = 0; k < MS_NND; k++) cp.dc[k] = np_.dc[k] = 0;
#endif
    for (int it = 0; it < nt; it++) {
        if (it == 0) {
            if (!ms_feasible(&E, &cp)) return;
            ms_eqs_sim(&E, &cp, &cp, 0, eqs);
        } else {
            np_.it = it;
            np_.id = 0;
            np_.cash = 0;
            np_.shock = 0;
            np_.savings = cp.savings;
            double r0 = EG_RS(irnd);
            const double r1 = EG_RS(irnd + 1), r2 = EG_RS(irnd + 2);
            irnd += 3;
            if (r2 > ms_survival(&E, &cp)) return;  // death: remaining periods stay NaN
            double pr = 0;
            for (np_.ist = 0; np_.ist < MS_NST; np_.ist++) {
#if MS_NCONT > 0
                {   // only the first grid point of every continuous state; their values travel in st[] (:274-280)
                    int k = 0;
                    for (; k < MS_NNST; k++)
                        if (ms_stcont[k] && (np_.ist / ms_ststride[k]) % ms_stsize[k] != 0) break;
                    if (k < MS_NNST) continue;
                    for (k = 0; k < MS_NNST; k++)
                        if (!ms_stcont[k]) np_.st[k] = ms_states[np_.ist + k * MS_NST];
                    ms_trpr_cont(&E, &cp, &np_);
                }
#endif
                if (!ms_feasible(&E, &np_)) continue;
                if (MS_OPTIM_TRPRNOSH)
                    pr = ms_trpr_discrete(&E, &cp, &np_, &terr);
                else {
                    mu = ms_mu(&E, &cp, &np_);
                    sigma = ms_sigma(&E, &cp, &np_);
                    np_.shock = (sigma <= 0) ? eg_shock_mean(&E, &cp, &np_) : eg_shock_uniform(r1, mu, sigma);
                    pr = ms_trpr_discrete(&E, &cp, &np_, &terr);
                }
                r0 -= pr;
                if (r0 <= 0) break;
            }
            if (np_.ist >= MS_NST) {
                atomicCAS(a.err, 0, EGDST_E_SIM_STATE);
                return;
            }
            if (MS_OPTIM_TRPRNOSH) {
                mu = ms_mu(&E, &cp, &np_);
                sigma = ms_sigma(&E, &cp, &np_);
                np_.shock = (sigma <= 0) ? eg_shock_mean(&E, &cp, &np_) : eg_shock_uniform(r1, mu, sigma);
            }
            np_.cash = ms_cashinhand(&E, &cp, &np_);
            ms_eqs_sim(&E, &cp, &np_, 1, eqs);
            cp = np_;
        }
#if MS_NCONT > 0
        // Consumption interpolated over the 2^k grid corners around the continuous states; the (state, decision) pair is
        // drawn among the corners by their weights with the fixed number .5 (egdst_simulator.c:318-372).  The reference
        // calls policy() without the value function there and adds up its uninitialised period buffer: the value
        // function column is 0 here (what a zeroed buffer gives).
        double c = 0, vf = 0.0;
        {
            const int nw = 1 << MS_NCONT, ist_base = cp.ist;
            double wts[1 << MS_NCONT], wc = 0, rr = .5;
            int wist[1 << MS_NCONT], ist1 = -1, id1 = 0;
            for (int ii = 0; ii < nw; ii++) wts[ii] = 1, wist[ii] = ist_base;
            for (int q = 0, k = 0; k < MS_NNST; k++) {
                if (!ms_stcont[k]) continue;
                const double *g = ms_stgrid(k);
                const int j1 = ms_bxsearch(cp.st[k], g, ms_stsize[k]);
                for (int ii = 0; ii < nw; ii++) {
                    if ((ii >> q) % 2 == 0) {
                        wts[ii] *= (g[j1 + 1] - cp.st[k]) / (g[j1 + 1] - g[j1]);
                        wist[ii] += ms_ststride[k] * j1;
                    } else {
                        wts[ii] *= (cp.st[k] - g[j1]) / (g[j1 + 1] - g[j1]);
                        wist[ii] += ms_ststride[k] * (j1 + 1);
                    }
                }
                q++;
            }
            for (int ii = 0; ii < nw; ii++) {
                if (!(wts[ii] > 0)) continue;
                const int is_ = wist[ii];
                if (is_ < 0 || is_ >= MS_NST) {
                    atomicCAS(a.err, 0, EGDST_E_NOT_SOLVED);
                    return;
                }
                const Tab t = eg_tab(b, it, draw, is_);
                if (t.len < 2) {
                    atomicCAS(a.err, 0, EGDST_E_NOT_SOLVED);
                    return;
                }
                const int i = eg_bracket(cp.cash, t.M, t.len, 0);
                const double cc = eg_lerp(cp.cash, t.M[i], t.M[i + 1], t.C[i], t.C[i + 1]);
                int ith = 0;
                while (ith < t.thlen && cp.cash >= t.TH[ith]) ith++;
                wc += cc * wts[ii];
                rr -= wts[ii];
                if (rr < 0 && ist1 == -1) ist1 = is_, id1 = (int)t.D[max(ith - 1, 0)];
            }
            if (ist1 < 0) {
                atomicCAS(a.err, 0, EGDST_E_NOT_SOLVED);
                return;
            }
            c = MS_MIN(wc, cp.cash - b.g.a0);
            cp.savings = cp.cash - c;
            cp.ist = ist1;
            cp.id = id1;
            for (int k = 0; k < MS_NND; k++) cp.dc[k] = ms_decisions[cp.id + k * MS_ND];
        }
#else
        // policy at (it, ist)
        const Tab t = eg_tab(b, it, draw, cp.ist);
        if (t.len < 2) {
            atomicCAS(a.err, 0, EGDST_E_NOT_SOLVED);
            return;
        }
        const int i = eg_bracket(cp.cash, t.M, t.len, 0);
        const double c = eg_lerp(cp.cash, t.M[i], t.M[i + 1], t.C[i], t.C[i + 1]);
        cp.savings = cp.cash - c;
        int ith = 0;
        while (ith < t.thlen && cp.cash >= t.TH[ith]) ith++;
        cp.id = (int)t.D[max(ith - 1, 0)];
        double vf;
        if (cp.cash < t.M[1] && t.V[0] > -INFINITY)
            vf = ms_utility(&E, &cp, c) + ms_discount(&E, &cp) * t.V[0];
        else
            vf = eg_lerp(cp.cash, t.M[i], t.M[i + 1], t.V[i], t.V[i + 1]);
#endif
        double *o = a.sims + ((size_t)isim * nt + it) * a.nout;
        o[0] = cp.cash;
        o[1] = c;
        o[2] = cp.savings;
        o[3] = vf;
        o[4] = (double)cp.id;
        o[5] = (double)cp.ist;
        o[6] = mu;
        o[7] = sigma;
        o[8] = cp.shock;
        o[9] = ms_utility(&E, &cp, c);
        o[10] = ms_discount(&E, &cp);
#if MS_NCONT > 0
        for (int k = 0; k < MS_NNST; k++) o[11 + k] = cp.st[k];  // exact values of the continuous states (:138)
#else
        for (int k = 0; k < MS_NNST; k++) o[11 + k] = ms_states[cp.ist + k * MS_NST];
#endif
        for (int k = 0; k < MS_NND; k++) o[11 + MS_NNST + k] = ms_decisions[cp.id + k * MS_ND];
        for (int k = 0; k < MS_NEQ; k++) o[11 + MS_NNST + MS_NND + k] = eqs[k];
    }
    if (terr) atomicCAS(a.err, 0, EGDST_E_TRPR_CASES);
}

// ---------------------------------------------------------------------------------------------
// egdst_call.c:17-164, the accessor behind egdstmodel.call: one lane per row of arguments.  The host has already found
// the first row whose index arguments are out of range (`first_bad`; rows from there on are NaN, egdst_host.inc), so the
// rows are independent.  sw: 1 utility, 2 marginal utility, 3 discount, 4 budget, 5 marginal budget, 6 value function.
struct CallArgs {
    int draw, sw, narg, ncol, first_bad, bad_keeps_zero;
    const double *args;  // [narg x ncol] column-major
    double *res;         // [narg]
};

__global__ void __launch_bounds__(GRID_BS) k_call(const Batch *bp_, CallArgs a)
{
    BatchRef b = EG_BATCH_REF(bp_);
    const int i = blockIdx.x * GRID_BS + threadIdx.x;
    if (i >= a.narg) return;
    if (i >= a.first_bad) {
        a.res[i] = (i == a.first_bad && a.bad_keeps_zero) ? 0.0 : NAN;
        return;
    }
    const int nt = b.g.nt, draw = a.draw;
    const size_t n = (size_t)a.narg;
    const double *arg = a.args + i;
    ms_env E = eg_env(b, draw);
    ms_pv cur, nxt;
    cur.it = (int)arg[0] - b.g.t0;
    cur.ist = (int)arg[n] - 1;
    cur.id = (a.ncol > 2 && a.sw != 6) ? (int)arg[2 * n] - 1 : 0;
    cur.cash = cur.savings = cur.shock = 0;
    nxt = cur;
    double r = 0.0;
    switch (a.sw) {
    case 1:
    case 2:
        if (arg[3 * n] > b.g.mmax - b.g.a0)
            r = NAN;
        else
            r = a.sw == 1 ? ms_utility(&E, &cur, arg[3 * n]) : ms_utility_marginal(&E, &cur, arg[3 * n]);
        break;
    case 3:
        r = ms_discount(&E, &cur);
        break;
    case 4:
    case 5:
        nxt.it = cur.it + 1;
        nxt.id = 0;
        nxt.savings = arg[3 * n];
        nxt.ist = (int)arg[4 * n] - 1;
        nxt.shock = arg[5 * n];
        if (nxt.it < 0 || nxt.it > nt - 1 || nxt.savings < b.g.a0)
            r = NAN;
        else
            r = a.sw == 4 ? ms_cashinhand(&E, &cur, &nxt) : ms_cashinhand_marginal(&E, &cur, &nxt);
        break;
    case 6:
        cur.cash = arg[2 * n];
        if (cur.cash > b.g.mmax)
            r = NAN;
        else if (cur.it == nt - 1)
            r = ms_utility(&E, &cur, MS_MAX(0, cur.cash));
        else {
            const Tab t = eg_tab(b, cur.it, draw, cur.ist);  // (history kept: slot = period)
            if (t.len <= 0)
                r = NAN;  // "Solution missing for given it,ist.."
            else if (t.len < 2)
                r = -1.0;  // linter's error return (egdst_lib.c:166)
            else {
                const double ma0 = t.M[1], evf = t.V[0];
                int k = eg_bracket(cur.cash, t.M, t.len, 0);
                const double c = eg_lerp(cur.cash, t.M[k], t.M[k + 1], t.C[k], t.C[k + 1]);
                cur.savings = cur.cash - c;
                int ith = 0;
                while (ith < t.thlen && cur.cash >= t.TH[ith]) ith++;
                cur.id = (int)t.D[ith > 0 ? ith - 1 : 0];
                if (cur.cash < ma0 && evf > -INFINITY)
                    r = ms_utility(&E, &cur, c) + ms_discount(&E, &cur) * evf;
                else if (cur.cash < ma0 && evf == -INFINITY)
                    r = -INFINITY;
                else
                    r = eg_lerp(cur.cash, t.M[k], t.M[k + 1], t.V[k], t.V[k + 1]);
            }
        }
        break;
    default:
        r = NAN;
    }
    a.res[i] = r;
}

// Mean of one (column, period) cell of the simulated paths over the agents that have a value there (the others are NaN:
// died, or started outside the admissible range).  One workgroup per cell; each thread adds its agents in index order,
// then a fixed tree over the threads: the result does not depend on scheduling.
#ifdef EGDST_EMU
#define MOM_BS 1
#else
#define MOM_BS 256
#endif
__global__ void __launch_bounds__(MOM_BS) k_moments(const double *sims, int nsim, int ncell, double *means, int *counts)
{
    __shared__ double ssum[MOM_BS];
    __shared__ int scnt[MOM_BS];
    const int cell = blockIdx.x, tid = threadIdx.x;
    sims += (size_t)blockIdx.y * ncell * nsim;   // (one set of paths, means and counts per draw of the launch)
    means += (size_t)blockIdx.y * ncell;
    counts += (size_t)blockIdx.y * ncell;
    double acc = 0;
    int cnt = 0;
    for (int i = tid; i < nsim; i += MOM_BS) {
        const double v = sims[(size_t)cell + (size_t)ncell * i];
        if (v == v) acc += v, cnt++;
    }
    ssum[tid] = acc;
    scnt[tid] = cnt;
    __syncthreads();
    for (int o = MOM_BS / 2; o > 0; o >>= 1) {
        if (tid < o) ssum[tid] += ssum[tid + o], scnt[tid] += scnt[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        counts[cell] = scnt[0];
        means[cell] = scnt[0] ? ssum[0] / scnt[0] : NAN;
    }
}

// Distance of one draw's simulated moments to the targets: sum over the cells with a non-zero weight of
// weight * (mean - target)^2, added in cell order by one thread per draw (deterministic); NaN if a weighted cell is empty.
__global__ void k_moment_objective(const double *means, const int *counts, int ncell, const double *target, const double *weight,
                                   int ndraw, double *obj)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= ndraw) return;
    const double *m = means + (size_t)d * ncell;
    const int *c = counts + (size_t)d * ncell;
    double acc = 0;
    for (int k = 0; k < ncell; k++) {
        if (weight[k] == 0.0) continue;
        if (c[k] == 0) {
            acc = NAN;
            break;
        }
        const double e = m[k] - target[k];
        acc += weight[k] * e * e;
    }
    obj[d] = acc;
}

// zero_first: the gateway's NaN fill starts at element 1 of its output array (egdst_simulator.c:105: element 0 is written by
// agent 0 anyway, or stays 0.0); the batched estimation step (egdst_simulate_batch_moments) promises NaN moments for a draw
// that failed or whose agent 0 has no value, so there every element is NaN
__global__ void k_fill_nan(double *p, size_t n, size_t per_draw, int zero_first)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (zero_first && i % per_draw == 0) ? 0.0 : NAN;
}

// Per-draw objective contribution for an estimation loop (new surface): the value and the
// consumption at the first endogenous grid point of (it=0, ist=0); NaN when the draw failed.
__global__ void k_objective(const Batch *bp_, double *out)
{
    BatchRef b = EG_BATCH_REF(bp_);
    const int draw = blockIdx.x * blockDim.x + threadIdx.x;
    if (draw >= b.g.ndraw) return;
    const int slot = (b.g.nslots == 2) ? 0 : 0;
    const size_t k = ((size_t)slot * b.g.ndraw + draw) * MS_NST;
    const bool ok = b.status[draw] == 0 && b.tlen[k] >= 2;
    out[2 * draw] = ok ? b.tV[k * b.g.Sp + 1] : NAN;
    out[2 * draw + 1] = ok ? b.tC[k * b.g.Sp + 1] : NAN;
}

// ---------------------------------------------------------------------------------------------
// Checksum of every cell of one draw: the wrapping 64-bit sums over the M, C, V rows i = 0..len-1 and over the TH, D
// entries i = 0..thlen-1 of  bits(x_i) * (2 i + 1);  out[(it*MS_NST+ist)*5 + {0..4}].  The odd weight ties every value to
// its row: swapped or shifted rows change the sum, which a plain sum of bit patterns would not notice.  Equal sums <=>
// bit-identical tables (up to 2^-64 luck), so a full-size solution is compared with a committed fixture without moving
// gigabytes (tests/golden/big_*.npz; tests/golden/make_golden_big.py: cell_sums is the same formula in numpy).
#define CHK_BS 256
__global__ void __launch_bounds__(CHK_BS) k_checksum(const Batch *bp_, int draw, unsigned long long *out)
{
    BatchRef b = EG_BATCH_REF(bp_);
    __shared__ unsigned long long sh[CHK_BS];
    const int it = blockIdx.x / MS_NST, ist = blockIdx.x % MS_NST, tid = threadIdx.x;
    const Tab t = eg_tab(b, it, draw, ist);  // (history kept: slot = period)
    const double *cols[5] = {t.M, t.C, t.V, t.TH, t.D};
    for (int c = 0; c < 5; c++) {
        const int n = c < 3 ? t.len : t.thlen;
        unsigned long long acc = 0;
        for (int i = tid; i < n; i += CHK_BS) acc += egm_bits(cols[c][i]) * (2ull * (unsigned long long)i + 1ull);
        sh[tid] = acc;
        __syncthreads();
        for (int o = CHK_BS / 2; o > 0; o >>= 1) {
            if (tid < o) sh[tid] += sh[tid + o];
            __syncthreads();
        }
        if (tid == 0) out[(size_t)blockIdx.x * 5 + c] = sh[0];
        __syncthreads();
    }
}

// The device's exp / log / pow on caller-supplied arguments (diagnostics: they must equal the host libm bit for bit,
// include/egdst_math.h).  fn: 0 exp(x), 1 log(x), 2 pow(x, y).
__global__ void k_math_eval(int fn, int n, const double *x, const double *y, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (fn >= 3) {  // the interpolation f0 (g1 - x) / (g1 - g0) + f1 (x - g0) / (g1 - g0): 3 the grid kernels' shared-reciprocal form, 4 the plain one
        const double xq = x[i], g0 = x[n + i], g1 = x[2 * (size_t)n + i], f0 = y[i], f1 = y[n + i];
        out[i] = fn == 3 ? eg_lerp_fast(xq, g0, g1, f0, f1) : eg_lerp(xq, g0, g1, f0, f1);
        return;
    }
    out[i] = fn == 0 ? MS_EXP(x[i]) : (fn == 1 ? MS_LOG(x[i]) : MS_POW(x[i], y[i]));
}

#include "egdst_host.inc"
