// nb_libm.h -- atan2f / sinf / cosf as the HOST's libm computes them, restated so that the device produces the same bits.
//
// Row a7 of the path (src/main.rs:437-439, rotation_of :141-143): M = T(p) * Rz(atan2(v.y, v.x)).  The reference's
// `f32::atan2`, `sin` and `cos` bind to the C library's atan2f / sinf / cosf (or sincosf, which LLVM forms from the pair and
// which evaluates the same two polynomials); on the Linux hosts the reference runs on that is glibc.  The device's own
// atan2f / sinf / cosf differ from it in the last place on a few inputs in a thousand, so until round 3 the model matrices were
// the one row graded by tolerance.  These are glibc 2.35's algorithms (the version of this image and of the GPU boxes), written
// out operation by operation:
//   atan2f, atanf   sysdeps/ieee754/flt-32/e_atan2f.c, s_atanf.c: the fdlibm forms, binary32 arithmetic throughout, no
//                   contraction (the x86-64 baseline they are built for has no FMA; this unit is compiled -ffp-contract=off)
//   sinf, cosf      sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h (ARM optimized routines): binary64 polynomials
//                   of the reduced argument, result rounded once to binary32.  glibc selects at load time between a build for
//                   the x86-64 baseline and one built with -mfma (sysdeps/x86_64/fpu/multiarch/s_sinf.c: every CPU with AVX2 +
//                   FMA, i.e. every host of the last decade); in the latter the compiler has contracted EVERY multiply-add of
//                   the routine -- read from the disassembly of libm.so.6's __sinf_fma / __cosf_fma -- which is what NB_LIBM_FMA
//                   restates with explicit fma(); NB_LIBM_FMA=0 gives the baseline build's unfused arithmetic.
// Written from the published algorithms and constants (the constants were checked against the tables in this image's libm.so.6);
// plain C, so that the same text compiles for the device (nb_aux.inc) and for the host, where tests/test_libm_restatement.py
// holds it to the host libm: sinf, cosf and atanf on all 2^32 arguments (tools/libm_exhaustive.c), atan2f on its special cases
// and on billions of drawn pairs.  Not a general libm: errno and the floating-point exception flags are not kept.
#ifndef NB_LIBM_H
#define NB_LIBM_H

#include <stdint.h>

#ifndef NB_LIBM_FN
#define NB_LIBM_FN static inline
#endif
#ifndef NB_LIBM_FMA
#define NB_LIBM_FMA 1
#endif

NB_LIBM_FN uint32_t nbm_bits(float x)
{
    union {
        float f;
        uint32_t u;
    } c;
    c.f = x;
    return c.u;
}
NB_LIBM_FN float nbm_float(uint32_t u)
{
    union {
        float f;
        uint32_t u;
    } c;
    c.u = u;
    return c.f;
}
NB_LIBM_FN float nbm_fabsf(float x) { return nbm_float(nbm_bits(x) & 0x7fffffffu); }
// a * b + c as the routine's build evaluates it
NB_LIBM_FN double nbm_madd(double a, double b, double c)
{
#if NB_LIBM_FMA
    return __builtin_fma(a, b, c);
#else
    return a * b + c;
#endif
}

// ---- atanf: glibc 2.35 sysdeps/ieee754/flt-32/s_atanf.c --------------------------------------------------------------------
NB_LIBM_FN float nbm_atanf(float x)
{
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                          6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    const float one = 1.0f;
    const int32_t hx = (int32_t)nbm_bits(x);
    const int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {  // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;  // NaN
        if (hx > 0) return atanhi[3] + atanlo[3];
        return -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {              // |x| < 0.4375
        if (ix < 0x31000000) return x;  // |x| < 2^-29
        id = -1;
    } else {
        x = nbm_fabsf(x);
        if (ix < 0x3f980000) {      // |x| < 1.1875
            if (ix < 0x3f300000) {  // 7/16 <= |x| < 11/16
                id = 0;
                x = (2.0f * x - one) / (2.0f + x);
            } else {  // 11/16 <= |x| < 19/16
                id = 1;
                x = (x - one) / (x + one);
            }
        } else {
            if (ix < 0x401c0000) {  // |x| < 2.4375
                id = 2;
                x = (x - 1.5f) / (one + 1.5f * x);
            } else {  // 2.4375 <= |x| < 2^25
                id = 3;
                x = -1.0f / x;
            }
        }
    }
    float z = x * x;
    const float w = z * z;
    // the sum of aT[i] z^(i+1), i = 0..10, as an odd and an even polynomial
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return hx < 0 ? -z : z;
}

// ---- atan2f: glibc 2.35 sysdeps/ieee754/flt-32/e_atan2f.c ------------------------------------------------------------------
NB_LIBM_FN float nbm_atan2f(float y, float x)
{
    const float tiny = 1.0e-30f, zero = 0.0f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
                pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)nbm_bits(x), hy = (int32_t)nbm_bits(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;        // x or y is NaN
    if (hx == 0x3f800000) return nbm_atanf(y);                     // x = 1.0
    const int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);         // 2 * sign(x) + sign(y)
    if (iy == 0) {                                                 // y = 0
        switch (m) {
        case 0:
        case 1: return y;         // atan(+-0, +anything) = +-0
        case 2: return pi + tiny;  // atan(+0, -anything) = pi
        default: return -pi - tiny;
        }
    }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;  // x = 0
    if (ix == 0x7f800000) {                                        // x is INF
        if (iy == 0x7f800000) {
            switch (m) {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0f * pi_o_4 + tiny;
            default: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
            case 0: return zero;
            case 1: return -zero;
            case 2: return pi + tiny;
            default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;  // y is INF
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60)
        z = pi_o_2 + 0.5f * pi_lo;  // |y / x| > 2^60
    else if (hx < 0 && k < -60)
        z = 0.0f;  // |y| / x < -2^60
    else
        z = nbm_atanf(nbm_fabsf(y / x));
    switch (m) {
    case 0: return z;                                   // atan(+, +)
    case 1: return nbm_float(nbm_bits(z) ^ 0x80000000u);  // atan(-, +)
    case 2: return pi - (z - pi_lo);                    // atan(+, -)
    default: return (z - pi_lo) - pi;                   // atan(-, -)
    }
}

// ---- sinf / cosf: glibc 2.35 sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h, s_sincosf_data.c -----------------------
// the polynomial of the quadrant: its table is that of n & 2 == 0 (c0 = 1 ...) or its negative; sine for even n, cosine for odd
NB_LIBM_FN float nbm_sinf_poly(double x, double x2, int negated, int n)
{
    const double sg = negated ? -1.0 : 1.0;
    const double c0 = sg * 0x1p0, c1 = sg * -0x1.ffffffd0c621cp-2, c2 = sg * 0x1.55553e1068f19p-5, c3 = sg * -0x1.6c087e89a359dp-10,
                 c4 = sg * 0x1.99343027bf8c3p-16;
    const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;  // (the same in both tables)
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = nbm_madd(x2, s3c, s2c);
        const double x7 = x3 * x2;
        const double s = nbm_madd(x3, s1c, x);
        return (float)nbm_madd(x7, s1, s);
    }
    const double x4 = x2 * x2;
    const double cc2 = nbm_madd(x2, c4, c3);
    const double cc1 = nbm_madd(x2, c1, c0);
    const double x6 = x4 * x2;
    const double c = nbm_madd(x4, c2, cc1);
    return (float)nbm_madd(x6, cc2, c);
}

// |y| in [pi/4, 120): x - n * (pi / 2), n the nearest integer to x * 2 / pi (taken from bits 24.. of a product scaled by 2^24)
NB_LIBM_FN double nbm_reduce_fast(double x, int *np)
{
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double r = x * hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return nbm_madd(-(double)n, hpi, x);
}

// |y| >= 120: the fraction of x * 4 / pi from 192 bits of 4 / pi, three 32 x 32-bit products
NB_LIBM_FN double nbm_reduce_large(uint32_t xi, int *np)
{
    const uint32_t inv_pio4[24] = {0xa2,       0xa2f9,     0xa2f983,   0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
                                   0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0,
                                   0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
    const uint32_t *arr = &inv_pio4[(xi >> 26) & 15];
    const int shift = (xi >> 23) & 7;
    uint64_t n, res0, res1, res2;
    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;
    res0 = (uint32_t)(xi * arr[0]);
    res1 = (uint64_t)xi * arr[4];
    res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    const double x = (double)(int64_t)res0;
    *np = (int)n;
    return x * 0x1.921FB54442D18p-62;
}

NB_LIBM_FN uint32_t nbm_abstop12(float x) { return (nbm_bits(x) >> 20) & 0x7ff; }

// cosine = 0: sinf(y); 1: cosf(y)
NB_LIBM_FN float nbm_sincosf(float y, int cosine)
{
    double x = y;
    int n;
    if (nbm_abstop12(y) < nbm_abstop12(0x1.921FB6p-1f)) {  // |y| < pi / 4
        const double x2 = x * x;
        if (nbm_abstop12(y) < nbm_abstop12(0x1p-12f)) return cosine ? 1.0f : y;
        return nbm_sinf_poly(x, x2, 0, cosine);
    }
    if (nbm_abstop12(y) < nbm_abstop12(120.0f)) {
        x = nbm_reduce_fast(x, &n);
        const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;  // the sign of the sine in quadrants 0..3: + - - +
        return nbm_sinf_poly(x * s, x * x, (n & 2) != 0, n ^ cosine);
    }
    if (nbm_abstop12(y) < nbm_abstop12(nbm_float(0x7f800000u))) {
        const uint32_t xi = nbm_bits(y);
        const int sign = (int)(xi >> 31);
        x = nbm_reduce_large(xi, &n);
        const int q = (n + sign) & 3;
        const double s = (q == 1 || q == 2) ? -1.0 : 1.0;
        return nbm_sinf_poly(x * s, x * x, ((n + sign) & 2) != 0, n ^ cosine);
    }
    return (y - y) / (y - y);  // inf, NaN: __math_invalidf
}
NB_LIBM_FN float nbm_sinf(float y) { return nbm_sincosf(y, 0); }
NB_LIBM_FN float nbm_cosf(float y) { return nbm_sincosf(y, 1); }

#endif /* NB_LIBM_H */
