// Host check of path_planner_amd/csrc/pp_cr.h (tests/test_cr_trig.py): the double-double sin / cos / atan2 / acos against glibc and
// against the x87 long-double functions rounded to double.  Prints, per function, how many of N arguments differ from each.
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "../path_planner_amd/csrc/pp_cr.h"
static double cr_sin(double x) { PPdd s, c; pp_cr_sincos_dd(x, &s, &c); return s.h + s.l; }
static double cr_cos(double x) { PPdd s, c; pp_cr_sincos_dd(x, &s, &c); return c.h + c.l; }
static double cr_atan2(double y, double x) {
    const double t0 = atan2(y, x);
    if (x == 0.0 || y == 0.0 || !(fabs(t0) > 1e-300) || !std::isfinite(x) || !std::isfinite(y)) return t0;
    PPdd s, c; pp_cr_sincos_dd(t0, &s, &c); return pp_cr_atan2_step(y, x, t0, s, c);
}
static double cr_acos(double v) {
    const double t0 = acos(v);
    if (!(fabs(v) < 1.0) || t0 == 0.0) return t0;
    PPdd s, c; pp_cr_sincos_dd(t0, &s, &c); return pp_cr_acos_step(v, t0, s, c);
}
static uint64_t st = 88172645463325252ull;
static double rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) * (1.0 / 9007199254740992.0); }
// is the long-double value v within 2^-9 ulp(double) of the midpoint between two doubles?  Then rounding it says nothing.
static bool near_midpoint(long double v) {
    const double d = (double)v;
    const long double e = fabsl(v - (long double)d);
    const double u = fabs(nextafter(d, INFINITY) - d);
    return fabsl(e - 0.5L * u) < u * (1.0L / 512);
}
int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 2000000;
    long bad[4] = {0, 0, 0, 0}, hard[4] = {0, 0, 0, 0}, glibc[4] = {0, 0, 0, 0};
    for (long i = 0; i < n; i++) {
        double x = (rnd() * 2 - 1) * 20.0;
        if (i % 4 == 0) x = rnd() * 1e-3;
        if (i % 7 == 0) x = (rnd() * 2 - 1) * 9e4;
        long double r = sinl((long double)x); double a = cr_sin(x);
        if (a != (double)r) { if (near_midpoint(r)) hard[0]++; else bad[0]++; } if (a != sin(x)) glibc[0]++;
        r = cosl((long double)x); a = cr_cos(x);
        if (a != (double)r) { if (near_midpoint(r)) hard[1]++; else bad[1]++; } if (a != cos(x)) glibc[1]++;
        double y = (rnd() * 2 - 1) * 100, xx = (rnd() * 2 - 1) * 100;
        if (i % 5 == 0) y *= 1e-9;
        if (i % 11 == 0) xx *= 1e-12;
        r = atan2l((long double)y, (long double)xx); a = cr_atan2(y, xx);
        if (a != (double)r) { if (near_midpoint(r)) hard[2]++; else bad[2]++; } if (a != atan2(y, xx)) glibc[2]++;
        double v = rnd() * 2 - 1;
        if (i % 9 == 0) v = 1.0 - rnd() * 1e-9;
        r = acosl((long double)v); a = cr_acos(v);
        if (a != (double)r) { if (near_midpoint(r)) hard[3]++; else bad[3]++; } if (a != acos(v)) glibc[3]++;
    }
    const char* name[4] = {"sin", "cos", "atan2", "acos"};
    for (int f = 0; f < 4; f++) std::printf("%s n %ld wrong %ld undecided %ld differs_from_glibc %ld\n", name[f], n, bad[f], hard[f], glibc[f]);
    return 0;
}
