/* dropin_roundtrip.c -- TEST: a plain-C caller using the drop-in library exactly the
 * way the reference's dctz-test.c does (allocation sizes dctz-test.c:135-162, calls
 * :181 and :250, un-scale for PSNR :188-210, metric :276).  Links against
 * libdctz-ec.so or libdctz-qt.so; prints one line "N outSize CR PSNR maxerr". */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "dctz.h"

int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 100000;
  const double eb = argc > 2 ? atof(argv[2]) : 1e-3;
  const int use_float = argc > 3 ? atoi(argv[3]) : 0;
  const size_t ts = use_float ? sizeof(float) : sizeof(double);
  t_var *var = malloc(sizeof(t_var)), *var_z = malloc(sizeof(t_var)), *var_r = malloc(sizeof(t_var));
  var->datatype = var_z->datatype = var_r->datatype = use_float ? FLOAT : DOUBLE;
  var->buf.d = malloc((size_t)N * ts);
  var_z->buf.d = malloc((size_t)N * ts + 4096);
  var_r->buf.d = malloc((size_t)N * ts);
  unsigned s = 12345u;
  for (int i = 0; i < N; i++) {
    s = s * 1664525u + 1013904223u;
    const double v = 37.0 * (sin(i / 97.0) + 0.3 * cos(5.1 * i / 97.0)) + 2.0 * ((double)(s >> 8) / (1 << 24) - 0.5);
    if (use_float) var->buf.f[i] = (float)v; else var->buf.d[i] = v;
  }
  size_t outSize = 0;
  if (dctz_compress(var, N, &outSize, var_z, eb) != 1) return 2;
  struct header h;
  memcpy(&h, var_z->buf.d, sizeof h);
  if (use_float) { if (h.scaling_factor.f != 1.0f) for (int i = 0; i < N; i++) var->buf.f[i] *= h.scaling_factor.f; }
  else { if (h.scaling_factor.d != 1.0) for (int i = 0; i < N; i++) var->buf.d[i] *= h.scaling_factor.d; }
  if (dctz_decompress(var_z, var_r) != 1) return 3;
  double maxerr = 0.0;
  for (int i = 0; i < N; i++) {
    const double e = use_float ? fabs((double)var->buf.f[i] - (double)var_r->buf.f[i]) : fabs(var->buf.d[i] - var_r->buf.d[i]);
    if (e > maxerr) maxerr = e;
  }
  const double psnr = calc_psnr(var, var_r, N, eb);
  printf("RESULT %d %zu %.4f %.6f %.9e %u\n", N, outSize, (double)N * ts / (double)outSize, psnr, maxerr, h.tot_AC_exact_count);
  return 0;
}
