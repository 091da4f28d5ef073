// Is RN(q0 + (ps - q0 res) inv) with inv = RN(1 / res), q0 = RN(ps inv) the IEEE quotient RN(ps / res)?  (fdm_device.hpp:
// div_by_res)  gcc -O2 -ffp-contract=off div_markstein.c -lm
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
static uint64_t s = 88172645463325252ull;
static uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static double nextn(double x, int n) { for (int i = 0; i < abs(n); ++i) x = nextafter(x, n > 0 ? INFINITY : -INFINITY); return x; }
int main() {
  const double ress[] = {0.1, 0.05, 0.02, 0.25, 0.3, 1.0 / 3.0, 0.07, (double)0.1f, (double)0.05f, (double)0.02f, 0.5, 1.0, 0.013};
  long bad = 0, tot = 0;
  for (unsigned ri = 0; ri < sizeof(ress) / sizeof(ress[0]); ++ri) {
    const volatile double res = ress[ri];
    const double inv = 1.0 / res;
    // adversarial: around (k + 0.5) * res and k * res, +- a few ulps
    for (long it = 0; it < 6000000; ++it) {
      const long k = (long)(rnd() % 2000001) - 1000000;
      const double half = (rnd() & 1) ? 0.5 : 0.0;
      double ps = ((double)k + half) * res;
      ps = nextn(ps, (int)(rnd() % 17) - 8);
      const double q = ps / res;
      const double q0 = ps * inv;
      const double r = fma(-q0, res, ps);
      const double q1 = fma(r, inv, q0);
      ++tot;
      if (memcmp(&q, &q1, 8) != 0 && !(q == 0.0 && q1 == 0.0)) { if (bad < 5) printf("res %.17g ps %.17g q %.17g q1 %.17g\n", res, ps, q, q1); ++bad; }
    }
    // random magnitudes
    for (long it = 0; it < 6000000; ++it) {
      const double m = (double)(rnd() >> 11) / 9007199254740992.0;
      const int e = (int)(rnd() % 60) - 30;
      double ps = ldexp(m + 0.5, e) * ((rnd() & 1) ? 1 : -1);
      const double q = ps / res, q0 = ps * inv, r = fma(-q0, res, ps), q1 = fma(r, inv, q0);
      ++tot;
      if (memcmp(&q, &q1, 8) != 0) { if (bad < 5) printf("res %.17g ps %.17g q %.17g q1 %.17g\n", res, ps, q, q1); ++bad; }
    }
  }
  printf("checked %ld, mismatches %ld\n", tot, bad);
  return 0;
}
