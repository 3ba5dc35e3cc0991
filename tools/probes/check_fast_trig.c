/* CPU check of the lean sincos / tanh used by the model evaluations (ihm2_amd/csrc/model.hpp: fast_sincos, tanh_e): the same constants and
 * operation order in plain C against libm over the argument ranges the models use (and far beyond).  tests/test_cabi.py builds and
 * runs it; exit code 0 = every deviation below 2.5e-16 absolute. */
#include <math.h>
#include <stdio.h>
static void fast_sincos(double x, double *sp, double *cp)
{
    const double n = rint(x * 6.36619772367581382433e-01);
    double r = fma(-n, 1.57079632679489655800e+00, x);
    r = fma(-n, 6.12323399573676603587e-17, r);
    r = fma(-n, -1.49738490485916983693e-33, r);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double sv = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double cv = 1.0 - fma(0.5, z, -(z * z) * pc);
    const int q = (int)n;
    double so = (q & 1) ? cv : sv, co = (q & 1) ? sv : cv;
    so = (q & 2) ? -so : so;
    co = ((q + 1) & 2) ? -co : co;
    *sp = so; *cp = co;
}
static double tanh_e(double y) { const double t = exp(-2.0 * fabs(y)); return copysign((1.0 - t) / (1.0 + t), y); }
int main(void)
{
    double es = 0, ec = 0, et = 0;
    for (int i = -2000000; i <= 2000000; i++) {
        double x = i * 1e-5 * (1 + (i % 7) * 0.37);      // up to ~ +-64
        double s, c; fast_sincos(x, &s, &c);
        double a = fabs(s - sin(x)), b = fabs(c - cos(x));
        if (a > es) es = a; if (b > ec) ec = b;
        double y = i * 1e-4; double d = fabs(tanh_e(y) - tanh(y)); if (d > et) et = d;
    }
    double s, c; fast_sincos(99999.5, &s, &c);
    printf("max abs err sin %.3e cos %.3e tanh %.3e ; at 99999.5: %.3e %.3e\n", es, ec, et, fabs(s - sin(99999.5)), fabs(c - cos(99999.5)));
    int bad = 0;
    for (double x = -8; x <= 8; x += 0.785) { fast_sincos(x, &s, &c); if (fabs(s - sin(x)) > 1e-15 || fabs(c - cos(x)) > 1e-15) { printf("bad %g\n", x); bad = 1; } }
    return (bad || es > 2.5e-16 || ec > 2.5e-16 || et > 2.5e-16) ? 1 : 0;
}
