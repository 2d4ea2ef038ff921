/* libm_exhaustive.c -- nenbody_amd/csrc/nb_libm.h against the host's libm, on the CPU (no GPU involved):
 *     gcc -O2 -mfma -ffp-contract=off -fopenmp -I nenbody_amd/csrc tools/libm_exhaustive.c -lm -o /tmp/libm_exhaustive
 *     /tmp/libm_exhaustive [stride [pairs]]
 * sinf, cosf and atanf on every `stride`-th binary32 bit pattern (stride 1: all 2^32, about a minute per function on 8 cores),
 * atan2f on its special cases, on a grid of exponents x signs x structured significands and on `pairs` drawn pairs.
 * Prints the mismatch counts (NaN results compare equal whatever their payload) and exits non-zero if there is one.
 * tests/test_libm_restatement.py runs it with a stride in the CPU suite; profiles/r04/libm_exhaustive.log holds the full run. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nb_libm.h"

static int same(float a, float b)
{
    if (a != a && b != b) return 1;
    return nbm_bits(a) == nbm_bits(b);
}
static uint64_t splitmix(uint64_t *s)
{
    uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv)
{
    const uint64_t stride = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
    const uint64_t pairs = argc > 2 ? strtoull(argv[2], 0, 10) : (1ull << 32);
    uint64_t bad_sin = 0, bad_cos = 0, bad_atan = 0, bad_atan2 = 0, n1 = 0, n2 = 0;
    uint32_t first_sin = 0, first_cos = 0, first_atan = 0;
#pragma omp parallel for reduction(+ : bad_sin, bad_cos, bad_atan, n1) schedule(static)
    for (int64_t i = 0; i < (int64_t)((0x100000000ull + stride - 1) / stride); ++i) {
        const uint32_t u = (uint32_t)((uint64_t)i * stride);
        const float x = nbm_float(u);
        ++n1;
        if (!same(nbm_sinf(x), sinf(x))) {
            if (!bad_sin) first_sin = u;
            ++bad_sin;
        }
        if (!same(nbm_cosf(x), cosf(x))) {
            if (!bad_cos) first_cos = u;
            ++bad_cos;
        }
        if (!same(nbm_atanf(x), atanf(x))) {
            if (!bad_atan) first_atan = u;
            ++bad_atan;
        }
    }
    printf("sinf  : %llu arguments, %llu mismatches (first at 0x%08x)\n", (unsigned long long)n1, (unsigned long long)bad_sin, first_sin);
    printf("cosf  : %llu arguments, %llu mismatches (first at 0x%08x)\n", (unsigned long long)n1, (unsigned long long)bad_cos, first_cos);
    printf("atanf : %llu arguments, %llu mismatches (first at 0x%08x)\n", (unsigned long long)n1, (unsigned long long)bad_atan, first_atan);
    /* atan2f: special values x special values, then exponents x signs x significands, then drawn pairs */
    static const uint32_t sp[] = {0x00000000u, 0x80000000u, 0x00000001u, 0x80000001u, 0x007fffffu, 0x00800000u, 0x3f800000u, 0xbf800000u,
                                  0x3f7fffffu, 0x3f800001u, 0x7f7fffffu, 0xff7fffffu, 0x7f800000u, 0xff800000u, 0x7fc00000u, 0xffc00000u,
                                  0x7f800001u, 0x40490fdbu, 0x3fc90fdbu, 0x3f490fdbu, 0x3ee00000u, 0x3f300000u, 0x3f980000u, 0x401c0000u,
                                  0x4c000000u, 0x31000000u, 0x3edfffffu, 0x3f2fffffu, 0x3f97ffffu, 0x401bffffu, 0x4bffffffu, 0x30ffffffu};
    const int nsp = (int)(sizeof(sp) / sizeof(sp[0]));
    for (int a = 0; a < nsp; ++a)
        for (int b = 0; b < nsp; ++b) {
            const float y = nbm_float(sp[a]), x = nbm_float(sp[b]);
            ++n2;
            if (!same(nbm_atan2f(y, x), atan2f(y, x))) {
                if (!bad_atan2) printf("atan2f(0x%08x, 0x%08x): 0x%08x, libm 0x%08x\n", sp[a], sp[b], nbm_bits(nbm_atan2f(y, x)), nbm_bits(atan2f(y, x)));
                ++bad_atan2;
            }
        }
    static const uint32_t mant[] = {0x000000u, 0x000001u, 0x7fffffu, 0x400000u, 0x3fffffu, 0x555555u, 0x2aaaaau, 0x123456u};
#pragma omp parallel for reduction(+ : bad_atan2, n2) schedule(dynamic, 4)
    for (int ey = 0; ey < 255; ++ey)
        for (int ex = 0; ex < 255; ++ex)
            for (int my = 0; my < 8; ++my)
                for (int mx = 0; mx < 8; ++mx)
                    for (int sg = 0; sg < 4; ++sg) {
                        const float y = nbm_float(((uint32_t)(sg & 1) << 31) | ((uint32_t)ey << 23) | mant[my]);
                        const float x = nbm_float(((uint32_t)(sg >> 1) << 31) | ((uint32_t)ex << 23) | mant[mx]);
                        ++n2;
                        if (!same(nbm_atan2f(y, x), atan2f(y, x))) ++bad_atan2;
                    }
#pragma omp parallel for reduction(+ : bad_atan2, n2) schedule(static)
    for (int64_t i = 0; i < (int64_t)pairs; ++i) {
        uint64_t s = 0x1234567ull + (uint64_t)i * 0x9e3779b97f4a7c15ull;
        const uint64_t r = splitmix(&s), r2 = splitmix(&s);
        float y, x;
        switch (i & 3) {
        case 0:  /* any two bit patterns */
            y = nbm_float((uint32_t)r), x = nbm_float((uint32_t)(r >> 32));
            break;
        case 1:  /* the same binade: ratios near 1 (every reduction interval of atanf) */
            y = nbm_float(((uint32_t)r & 0x807fffffu) | 0x3f800000u), x = nbm_float(((uint32_t)(r >> 32) & 0x80ffffffu) | 0x3f000000u);
            break;
        case 2:  /* velocities as the step produces them: magnitudes 1e-4 .. 10 */
            y = (float)((double)(int64_t)(r >> 11) * 0x1p-53 * 20.0 - 10.0) * ((r2 & 1) ? 1e-3f : 1.0f);
            x = (float)((double)(int64_t)(r2 >> 11) * 0x1p-53 * 20.0 - 10.0);
            break;
        default: /* exponents up to 2^+-70 apart: both saturation branches */
            y = nbm_float(((uint32_t)r & 0x807fffffu) | ((64u + (uint32_t)(r2 % 128u)) << 23));
            x = nbm_float(((uint32_t)(r >> 32) & 0x807fffffu) | ((64u + (uint32_t)((r2 >> 8) % 128u)) << 23));
        }
        ++n2;
        if (!same(nbm_atan2f(y, x), atan2f(y, x))) ++bad_atan2;
    }
    printf("atan2f: %llu pairs, %llu mismatches\n", (unsigned long long)n2, (unsigned long long)bad_atan2);
    return (bad_sin | bad_cos | bad_atan | bad_atan2) ? 1 : 0;
}
