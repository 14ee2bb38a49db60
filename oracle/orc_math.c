/* oracle/orc_math.c -- scalar primitives of the CPU oracle (test infrastructure, see orc.h).
 * Restates OpenCV / libstdc++ primitives that the reference calls but does not ship
 * (SURVEY.md Appendix A.1, A.5).  Compiled with -ffp-contract=off: no FMA anywhere. */
#include "orc.h"
#include <math.h>
#include <string.h>

/* cvRound(double): lrint under round-to-nearest-even (ORBextractor.cpp:81,115,119; frame.cpp:78 convertTo) */
int orc_cvround(double v) { return (int)lrint(v); }
int orc_cvroundf(float v) { return (int)lrintf(v); }
int orc_cvfloorf(float v) { int i = (int)v; return i - (v < (float)i); }

/* Deterministic sin/cos: Cody-Waite reduction by pi/2 + fdlibm kernel polynomials, Horner form,
 * every operation an IEEE-754 double add or multiply (no FMA), so the HIP kernels reproduce it bit
 * for bit.  Stands in for libm cos/sin at frame.cpp:141-149 and (through a cast) ORBextractor.cpp:113. */
void orc_sincos(double x, double* s, double* c)
{
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1  = 1.57079632673412561417e+00;
    const double pio2_1t = 6.07710050650619224932e-11;
    double k = rint(x * invpio2);
    double r = (x - k * pio2_1) - k * pio2_1t;
    double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = ps * z + -2.50507602534068634195e-08;
    ps = ps * z + 2.75573137070700676789e-06;
    ps = ps * z + -1.98412698298579493134e-04;
    ps = ps * z + 8.33333333332248946124e-03;
    ps = ps * z + -1.66666666666666324348e-01;
    double sr = r + (r * z) * ps;
    double pc = -1.13596475577881948265e-11;
    pc = pc * z + 2.08757232129817482790e-09;
    pc = pc * z + -2.75573143513906633035e-07;
    pc = pc * z + 2.48015872894767294178e-05;
    pc = pc * z + -1.38888888888741095749e-03;
    pc = pc * z + 4.16666666666666019037e-02;
    double cr = (1.0 - 0.5 * z) + (z * z) * pc;
    long q = ((long)k) & 3;
    switch (q) {
    case 0: *s = sr;  *c = cr;  break;
    case 1: *s = cr;  *c = -sr; break;
    case 2: *s = -sr; *c = -cr; break;
    default:*s = -cr; *c = sr;  break;
    }
}

/* cv::fastAtan2 (scalar path), called at ORBextractor.cpp:103 */
float orc_fast_atan2(float y, float x)
{
    const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* cv::RNG: multiply-with-carry, default state 0xffffffff (FEAmatcher.cpp:59,201) */
uint32_t orc_cvrng_next(uint64_t* state)
{
    *state = (uint64_t)(uint32_t)(*state) * 4164903690U + (uint32_t)(*state >> 32);
    return (uint32_t)(*state);
}
int orc_cvrng_uniform(uint64_t* state, int a, int b)
{
    return a == b ? a : (int)(orc_cvrng_next(state) % (uint32_t)(b - a) + a);
}

/* std::default_random_engine (minstd_rand0, seed 1) + std::normal_distribution<double>(0,1) as
 * implemented by libstdc++ (Marsaglia polar, second value cached): optimizer.cpp:30-31,154-158 */
void orc_normal_fill(double* out, int n)
{
    uint64_t x = 1;
    int have = 0; double saved = 0;
    const double R = 2147483646.0;
    for (int i = 0; i < n; ++i) {
        if (have) { have = 0; out[i] = saved; continue; }
        double u, v, r2;
        do {
            double c[2];
            for (int k = 0; k < 2; ++k) {
                x = (x * 16807ULL) % 2147483647ULL; double e1 = (double)(x - 1);
                x = (x * 16807ULL) % 2147483647ULL; double e2 = (double)(x - 1);
                double sum = e1 + e2 * R;
                double can = sum / (R * R);
                if (can >= 1.0) can = nextafter(1.0, 0.0);
                c[k] = can;
            }
            u = 2.0 * c[0] - 1.0;
            v = 2.0 * c[1] - 1.0;
            r2 = u * u + v * v;
        } while (r2 > 1.0 || r2 == 0.0);
        double mult = sqrt(-2 * log(r2) / r2);
        saved = u * mult; have = 1;
        out[i] = v * mult;
    }
}

/* FEAmatcher::DescriptorDistance (FEAmatcher.cpp:442-458): SWAR popcount over 8 x u32 */
int orc_hamming256(const uint8_t* a, const uint8_t* b)
{
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4); memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

/* cv::norm(a,b,NORM_L2) on 32 bytes is sqrt of this integer (FEAmatcher.cpp:113) */
int orc_l2sq32(const uint8_t* a, const uint8_t* b)
{
    int s = 0;
    for (int i = 0; i < 32; ++i) { int d = (int)a[i] - (int)b[i]; s += d * d; }
    return s;
}
int orc_l2sq(const uint8_t* a, const uint8_t* b, int n)
{
    int s = 0;
    for (int i = 0; i < n; ++i) { int d = (int)a[i] - (int)b[i]; s += d * d; }
    return s;
}
