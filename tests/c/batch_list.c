/* batch_list.c -- TEST / EXAMPLE: a LIST of small arrays through the batch entry points of the C ABI
 * (include/dctz_hip.h: dctzhip_compress_batch / dctzhip_decompress_batch), INTEGRATION.md section E.
 * The reference's own workloads are such lists, one dctz_compress call -- one process -- per array
 * (tests/test-dctz.sh:13-56 over tests/list-msst19.txt:1-6).  This program takes the list's lengths on the command
 * line (fp64 arrays, every one at the four bounds 1e-3 .. 1e-6 of the reference's sweep), fills them with a seeded
 * smooth-plus-noise signal, runs the list (a) one call per array and (b) as one batch, and checks that (b)'s streams
 * and reconstructions are byte for byte (a)'s.
 * usage: batch_list [length ...]      (default: the six list-msst19 lengths)
 * Prints "BATCH arrays=<k> identical looped_us=<t> batch_us=<t>". */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include "dctz_hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != DCTZHIP_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, dctzhip_last_error(ctx)); return 1; } } while (0)

static double now_us(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }

int main(int argc, char **argv) {
  static const size_t dflt[6] = {31040, 32768, 12960, 12960, 16384, 37024};      /* tests/list-msst19.txt:1-6 */
  static const double bounds[4] = {1e-3, 1e-4, 1e-5, 1e-6};
  const int nlen = argc > 1 ? argc - 1 : 6;
  const int k = nlen * 4;
  dctzhip_ctx *ctx = NULL;
  if (dctzhip_device_count() < 1) { fprintf(stderr, "no GPU\n"); return 1; }
  CHECK(dctzhip_ctx_create(&ctx, 0));
  dctzhip_batch_citem *ci = calloc((size_t)k, sizeof *ci);
  dctzhip_batch_ditem *di = calloc((size_t)k, sizeof *di);
  dctzhip_cinfo *info_b = calloc((size_t)k, sizeof *info_b), *info_s = calloc((size_t)k, sizeof *info_s);
  void **bin_s = calloc((size_t)k, sizeof(void *)), **out_s = calloc((size_t)k, sizeof(void *));
  float **dc_s = calloc((size_t)k, sizeof(float *)), **ac_s = calloc((size_t)k, sizeof(float *));
  for (int i = 0; i < k; i++) {
    const size_t n = argc > 1 ? (size_t)atoll(argv[1 + i / 4]) : dflt[i / 4];
    const size_t nblk = (n + 63) / 64;
    double *x = malloc(n * sizeof(double));
    unsigned s = 12345u + (unsigned)(i / 4);
    for (size_t j = 0; j < n; j++) {
      s = s * 1664525u + 1013904223u;
      const double t = (double)j / (double)n;
      x[j] = 41.0 * (3.0 * sin(14.0 * M_PI * t) + cos(90.0 * M_PI * t * t) + 0.02 * ((double)(s >> 8) / 8388608.0 - 1.0));
    }
    void *d_x;
    CHECK(dctzhip_malloc(ctx, &d_x, n * sizeof(double)));
    CHECK(dctzhip_memcpy_h2d(ctx, d_x, x, n * sizeof(double)));
    free(x);
    ci[i].d_in = d_x; ci[i].n = n; ci[i].dtype = DCTZHIP_F64; ci[i].error_bound = bounds[i % 4];
    CHECK(dctzhip_malloc(ctx, &ci[i].d_bin_index, n));
    CHECK(dctzhip_malloc(ctx, (void **)&ci[i].d_dc, nblk * 4));
    CHECK(dctzhip_malloc(ctx, (void **)&ci[i].d_ac_exact, n * 4));
    CHECK(dctzhip_malloc(ctx, &bin_s[i], n));
    CHECK(dctzhip_malloc(ctx, (void **)&dc_s[i], nblk * 4));
    CHECK(dctzhip_malloc(ctx, (void **)&ac_s[i], n * 4));
    CHECK(dctzhip_malloc(ctx, &di[i].d_out, n * sizeof(double)));
    CHECK(dctzhip_malloc(ctx, &out_s[i], n * sizeof(double)));
  }
  double t_loop = 0, t_batch = 0;
  for (int rep = 0; rep < 20; rep++) {
    /* (a) the reference's way: one call per array */
    double t0 = now_us();
    for (int i = 0; i < k; i++) {
      CHECK(dctzhip_compress(ctx, ci[i].d_in, ci[i].n, DCTZHIP_F64, ci[i].error_bound, DCTZHIP_EC, bin_s[i], dc_s[i], ac_s[i], NULL, NULL, &info_s[i]));
      CHECK(dctzhip_decompress(ctx, bin_s[i], dc_s[i], ac_s[i], info_s[i].cnt, NULL, ci[i].n, DCTZHIP_F64, ci[i].error_bound, info_s[i].sf, DCTZHIP_EC, out_s[i]));
    }
    CHECK(dctzhip_sync(ctx));
    double t1 = now_us();
    /* (b) the list as one batch */
    CHECK(dctzhip_compress_batch(ctx, k, ci, DCTZHIP_EC, info_b));
    for (int i = 0; i < k; i++) {
      di[i].d_bin_index = ci[i].d_bin_index; di[i].d_dc = ci[i].d_dc; di[i].d_ac_exact = ci[i].d_ac_exact;
      di[i].ac_count = info_b[i].cnt; di[i].qtable_host = NULL; di[i].n = ci[i].n; di[i].dtype = DCTZHIP_F64;
      di[i].error_bound = ci[i].error_bound; di[i].sf = info_b[i].sf;
    }
    CHECK(dctzhip_decompress_batch(ctx, k, di, DCTZHIP_EC, NULL));
    CHECK(dctzhip_sync(ctx));
    double t2 = now_us();
    if (rep >= 5) { t_loop += t1 - t0; t_batch += t2 - t1; }
  }
  /* byte for byte */
  for (int i = 0; i < k; i++) {
    const size_t n = ci[i].n, nblk = (n + 63) / 64;
    if (info_b[i].cnt != info_s[i].cnt || info_b[i].sf != info_s[i].sf) { fprintf(stderr, "array %d: header scalars differ\n", i); return 2; }
    const size_t sizes[4] = {n, nblk * 4, (size_t)info_b[i].cnt * 4, n * sizeof(double)};
    const void *a[4] = {ci[i].d_bin_index, ci[i].d_dc, ci[i].d_ac_exact, di[i].d_out};
    const void *b[4] = {bin_s[i], dc_s[i], ac_s[i], out_s[i]};
    for (int s = 0; s < 4; s++) {
      if (!sizes[s]) continue;
      unsigned char *ha = malloc(sizes[s]), *hb = malloc(sizes[s]);
      CHECK(dctzhip_memcpy_d2h(ctx, ha, a[s], sizes[s]));
      CHECK(dctzhip_memcpy_d2h(ctx, hb, b[s], sizes[s]));
      if (memcmp(ha, hb, sizes[s])) { fprintf(stderr, "array %d, stream %d differs\n", i, s); return 2; }
      free(ha); free(hb);
    }
  }
  printf("BATCH arrays=%d identical looped_us=%.1f batch_us=%.1f\n", k, t_loop / 15.0, t_batch / 15.0);
  dctzhip_ctx_destroy(ctx);
  return 0;
}
