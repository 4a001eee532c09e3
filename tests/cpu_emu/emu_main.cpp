// Stand-alone driver of the sanitizer build (TSan does not work preloaded into python):
//   emu_main <case file>     case file: t0 T ngridm ngridmax nthrhmax ny mmax a0 nparam  then 2*ny quadrature, then params
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/egdst.h"
int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "r");
    if (!f) return 2;
    egdst_desc d;
    int np;
    if (fscanf(f, "%d %d %d %d %d %d %lf %lf %d", &d.t0, &d.T, &d.ngridm, &d.ngridmax, &d.nthrhmax, &d.ny, &d.mmax, &d.a0, &np) != 9) return 2;
    std::vector<double> q(2 * d.ny), par(np > 0 ? np : 1);
    for (auto &x : q) if (fscanf(f, "%lf", &x) != 1) return 2;
    for (int i = 0; i < np; i++) if (fscanf(f, "%lf", &par[i]) != 1) return 2;
    d.quadrature = q.data();
    egdst_handle *h = nullptr;
    int rc = egdst_create(&d, 1, 1, nullptr, &h);
    if (rc) { printf("create rc=%d %s\n", rc, egdst_last_error()); return 1; }
    egdst_set_params(h, par.data(), 1);
    rc = egdst_solve(h);
    long long ev = 0;
    egdst_get_evals(h, &ev, nullptr);
    int dbg[16];
    egdst_get_debug(h, 0, dbg);
    printf("solve rc=%d evals=%lld dbg0=%d\n", rc, ev, dbg[0]);
    egdst_destroy(h);
    return rc != 0;
}
