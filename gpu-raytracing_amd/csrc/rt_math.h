/* rt_math.h -- the three transcendental functions of the shading code (log2f in ComputeLOD, Tracer.cu:237-254;
 * powf(2, lod) in Bump2Normal, :157-185; the double pow() of the specular term, :412-416) as plain IEEE arithmetic.
 *
 * Why: the reference calls the CUDA math library, whose results are specified to 1-2 ulp, not bit for bit, and neither
 * the device library of this platform nor the host libm reproduces it -- or each other: the last bit of log2f next to an
 * integer flips `(int)lod`, i.e. the mip level (a 20-grey-level jump in kLODs, a different texel in kTextureLit).  These
 * versions use only +, -, *, /, floor and integer bit operations in double precision, evaluated in one fixed order, so
 * the HIP kernels (compiled -ffp-contract=off) and the C oracle (gcc -ffp-contract=off) produce IDENTICAL bits; all nine
 * render types are then byte-comparable.  Accuracy: < 1 ulp of the float result for log2f / exp2f, relative error
 * < 1e-13 for pow (the reference narrows it to float at once); tests/test_oracle_cpu.py checks them against libm.
 * C99 and HIP; included by trace_kernel.hip and by oracle/rt_oracle.c (deliberately the same text on both sides --
 * its agreement with libm is what the CPU test pins). */
#ifndef RT_MATH_H
#define RT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RT_MATH_FN __host__ __device__ static inline
#else
#define RT_MATH_FN static inline
#endif

RT_MATH_FN double rt_from_bits64(uint64_t u) { double d; __builtin_memcpy(&d, &u, 8); return d; }
RT_MATH_FN uint64_t rt_to_bits64(double d) { uint64_t u; __builtin_memcpy(&u, &d, 8); return u; }
RT_MATH_FN float rt_from_bits32(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

/* log2 of a positive finite double: x = m * 2^e with m in (sqrt(1/2), sqrt(2)], log(m) = 2 atanh((m-1)/(m+1)) as a
 * series in s^2 (|s| <= 0.1716, 11 terms: remainder < 1e-18) */
RT_MATH_FN double rt_log2_pos(double x)
{
    uint64_t u = rt_to_bits64(x);
    int e = (int)((u >> 52) & 0x7FFu);
    if (e == 0) {                               /* subnormal: scale by 2^54 */
        u = rt_to_bits64(x * 18014398509481984.0);
        e = (int)((u >> 52) & 0x7FFu) - 54;
    }
    e -= 1023;
    double m = rt_from_bits64((u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);   /* [1, 2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 0.047619047619047616;            /* 1/21 */
    p = p * z + 0.052631578947368418;           /* 1/19 */
    p = p * z + 0.058823529411764705;           /* 1/17 */
    p = p * z + 0.066666666666666666;           /* 1/15 */
    p = p * z + 0.076923076923076927;           /* 1/13 */
    p = p * z + 0.090909090909090912;           /* 1/11 */
    p = p * z + 0.1111111111111111;             /* 1/9 */
    p = p * z + 0.14285714285714285;            /* 1/7 */
    p = p * z + 0.2;                            /* 1/5 */
    p = p * z + 0.33333333333333331;            /* 1/3 */
    p = p * z + 1.0;
    return (double)e + ((2.0 * s) * p) * 1.4426950408889634;   /* 1 / ln 2 */
}

/* 2^x in double: x = k + f, |f| <= 1/2, exp(f ln 2) by its Taylor series to degree 13 (remainder < 5e-18) */
RT_MATH_FN double rt_exp2_d(double x)
{
    if (x != x) return x;
    if (x >= 1024.0) return rt_from_bits64(0x7FF0000000000000ull);   /* +inf */
    if (x < -1100.0) return 0.0;
    const double k = __builtin_floor(x + 0.5);
    const double t = (x - k) * 0.69314718055994529;                   /* ln 2 */
    double p = 1.6059043836821613e-10;          /* 1/13! */
    p = p * t + 2.08767569878681e-09;           /* 1/12! */
    p = p * t + 2.505210838544172e-08;          /* 1/11! */
    p = p * t + 2.7557319223985888e-07;         /* 1/10! */
    p = p * t + 2.7557319223985893e-06;         /* 1/9! */
    p = p * t + 2.4801587301587302e-05;         /* 1/8! */
    p = p * t + 0.00019841269841269841;         /* 1/7! */
    p = p * t + 0.0013888888888888889;          /* 1/6! */
    p = p * t + 0.0083333333333333332;          /* 1/5! */
    p = p * t + 0.041666666666666664;           /* 1/4! */
    p = p * t + 0.16666666666666666;            /* 1/3! */
    p = p * t + 0.5;
    p = p * t + 1.0;
    p = p * t + 1.0;
    int ki = (int)k;
    if (ki < -1022) {                           /* result in the subnormal range: two exact scalings */
        p = p * rt_from_bits64(0x0010000000000000ull);   /* 2^-1022 */
        ki = ki + 1022;
        if (ki < -1022) return 0.0;
    }
    return p * rt_from_bits64((uint64_t)(ki + 1023) << 52);
}

/* log2f: the float nearest to rt_log2_pos (double rounding cannot be excluded: < 1 ulp, almost always correctly rounded) */
RT_MATH_FN float rt_log2f(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return rt_from_bits32(0x7FC00000u);
    if (x == 0.0f) return rt_from_bits32(0xFF800000u);                /* -inf */
    if (x == rt_from_bits32(0x7F800000u)) return x;
    return (float)rt_log2_pos((double)x);
}

RT_MATH_FN float rt_exp2f(float x) { return (float)rt_exp2_d((double)x); }

/* pow(x, y) for x >= 0 (the specular term: x = max(dot, 0.0), y = Ns) */
RT_MATH_FN double rt_pow_d(double x, double y)
{
    if (y == 0.0) return 1.0;
    if (x != x || y != y) return rt_from_bits64(0x7FF8000000000000ull);
    if (x == 1.0) return 1.0;
    if (x < 0.0) return rt_from_bits64(0x7FF8000000000000ull);
    if (x == 0.0) return y > 0.0 ? 0.0 : rt_from_bits64(0x7FF0000000000000ull);
    if (x == rt_from_bits64(0x7FF0000000000000ull)) return y > 0.0 ? x : 0.0;
    return rt_exp2_d(y * rt_log2_pos(x));
}

#endif /* RT_MATH_H */
