/* Check on the CPU that five fused operations with the correctly rounded reciprocal give the
 * correctly rounded quotient, bit for bit, signed zeros included:
 *     r  = RN(1/d)
 *     q0 = RN(x r);  e0 = -RN(q0 d - x);  q1 = RN(q0 + e0 r);  e1 = -RN(q1 d - x);  q = RN(q1 + e1 r)
 * for |x| in [2^-900, 2^900] or zero and d in [2^-100, 2^100] (no intermediate leaves the normal
 * range there).  q1 is a faithful rounding of x/d whatever the error of q0, so the last step is
 * Markstein's correction.  Random operands plus quotients placed next to rounding boundaries.
 *   gcc -O2 -ffp-contract=off -mfma -o fma_division_check fma_division_check.c -lm && ./fma_division_check [percent of the cases] */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t s[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
static uint64_t rnd(void) {
  uint64_t a = s[0], b = s[1];
  s[0] = b;
  a ^= a << 23;
  s[1] = a ^ b ^ (a >> 17) ^ (b >> 26);
  return s[1] + b;
}
static double from_bits(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static uint64_t bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static double rnd_double(int emin, int emax, int sign) {
  const uint64_t m = rnd() & 0xFFFFFFFFFFFFFull;
  const int e = emin + (int)(rnd() % (uint64_t)(emax - emin + 1));
  return from_bits(((uint64_t)(sign && (rnd() & 1)) << 63) | ((uint64_t)(e + 1023) << 52) | m);
}
static double fdiv(double x, double d, double r) {
  /* volatile: the negation must be applied to the rounded result (a fused negate-multiply-subtract
   * instruction returns the other zero when the exact result is zero) */
  const double q0 = x * r;
  const volatile double t0 = fma(q0, d, -x);
  const double q1 = fma(-t0, r, q0);
  const volatile double t1 = fma(q1, d, -x);
  return fma(-t1, r, q1);
}
static long bad = 0, n = 0;
static void check(double x, double d) {
  const volatile double qv = x / d;
  const double q = fdiv(x, d, 1.0 / d);
  n++;
  if (bits(q) != bits(qv)) {
    if (bad < 10) printf("MISMATCH x=%a d=%a: %a, IEEE %a\n", x, d, q, (double)qv);
    bad++;
  }
}
int main(int argc, char **argv) {
  const long pc = argc > 1 ? atol(argv[1]) : 100;
  for (long k = 0; k < 1000000L * pc; k++) check(rnd_double(-900, 900, 1), rnd_double(-100, 100, 0));
  /* narrow exponent ranges: more ties and near-ties per case */
  for (long k = 0; k < 1000000L * pc; k++) check(rnd_double(-2, 2, 1), rnd_double(-2, 2, 0));
  /* quotients next to a rounding boundary: x = RN((q + half an ulp) d) and its neighbours */
  for (long k = 0; k < 300000L * pc; k++) {
    const double d = rnd_double(-3, 3, 0), q = rnd_double(-3, 3, 1);
    const double x0 = fma(q, d, 0.5 * (nextafter(q, INFINITY) - q) * d);
    double x = x0;
    for (int j = -2; j <= 2; j++) {
      x = x0;
      for (int t = 0; t < (j < 0 ? -j : j); t++) x = nextafter(x, j < 0 ? -INFINITY : INFINITY);
      check(x, d);
    }
  }
  /* divisors with few bits, with all bits, powers of two; zeros of either sign */
  for (long k = 0; k < 200000L * pc; k++) {
    const double x = rnd_double(-900, 900, 1);
    check(x, (double)(1 + rnd() % 1000) * 0x1p-7);
    check(x, from_bits(bits(rnd_double(-100, 100, 0)) | 0xFFFFFFFFFFFFFull));
    check(x, ldexp(1.0, (int)(rnd() % 200) - 100));
  }
  check(0.0, 3.0);
  check(-0.0, 3.0);
  check(0.0, 0x1.fffffffffffffp-3);
  check(-0.0, 0x1.fffffffffffffp-3);
  printf("%ld cases, %ld mismatches\n", n, bad);
  return bad != 0;
}
