/* entropy_stage.c -- TEST / EXAMPLE: the code INTEGRATION.md section B shows for a maintainer who moves the zlib tail of
 * dctz_compress() (dctz-comp-lib.c:620-732) to the device: compress through the C ABI, deflate the three sections where
 * the kernels left them, copy only the compressed bytes, and -- the reader's side, dctz-decomp-lib.c:244-322 unchanged --
 * inflate them with zlib and feed dctzhip_decompress.
 * usage: entropy_stage <elements> <error bound>
 * Prints "ENTROPY n cnt raw_bytes stream_bytes max_abs_err". */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include "dctz_hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != DCTZHIP_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, dctzhip_last_error(g)); return 1; } } while (0)

int main(int argc, char **argv) {
  const size_t N = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)1 << 20;
  const double error_bound = argc > 2 ? atof(argv[2]) : 1e-3;
  const size_t nblk = (N + 63) / 64;
  dctzhip_ctx *g = NULL;
  if (dctzhip_device_count() < 1) { fprintf(stderr, "no GPU\n"); return 1; }
  CHECK(dctzhip_ctx_create(&g, 0));

  double *x = (double *)malloc(N * sizeof(double)), *r = (double *)malloc(N * sizeof(double));
  for (size_t i = 0; i < N; i++) x[i] = 3.0 * sin((double)i / 37.0) + 0.4 * cos((double)i / 5.1) + 1e-3 * (double)((i * 2654435761u) % 1000);
  void *d_in, *d_bin, *d_dc, *d_ac, *d_out;
  CHECK(dctzhip_malloc(g, &d_in, N * 8)); CHECK(dctzhip_malloc(g, &d_bin, N)); CHECK(dctzhip_malloc(g, &d_dc, nblk * 4));
  CHECK(dctzhip_malloc(g, &d_ac, N * 4)); CHECK(dctzhip_malloc(g, &d_out, N * 8));
  CHECK(dctzhip_memcpy_h2d(g, d_in, x, N * 8));
  dctzhip_cinfo info;
  CHECK(dctzhip_compress(g, d_in, N, DCTZHIP_F64, error_bound, DCTZHIP_EC, d_bin, (float *)d_dc, (float *)d_ac, NULL, NULL, &info));

  /* --- the tail on the device: three zlib streams in HBM, compressed bytes to the host --- */
  const void *src[3] = {d_bin, d_dc, d_ac};
  size_t raw[3] = {N, nblk * sizeof(float), (size_t)info.cnt * sizeof(float)}, cap[3], zlen[3];
  void *dz[3];
  unsigned char *z[3];
  for (int i = 0; i < 3; i++) { cap[i] = dctzhip_deflate_bound(raw[i]); CHECK(dctzhip_malloc(g, &dz[i], cap[i])); }
  CHECK(dctzhip_deflate(g, 3, src, raw, dz, cap, zlen, NULL));
  size_t raw_total = 0, z_total = 0;
  for (int i = 0; i < 3; i++) {
    z[i] = (unsigned char *)malloc(zlen[i]);
    CHECK(dctzhip_memcpy_d2h(g, z[i], dz[i], zlen[i]));
    raw_total += raw[i]; z_total += zlen[i];
  }

  /* --- the reader: plain zlib, as the reference --- */
  void *h[3];
  for (int i = 0; i < 3; i++) {
    h[i] = malloc(raw[i] ? raw[i] : 1);
    uLongf got = (uLongf)raw[i];
    if (uncompress((Bytef *)h[i], &got, z[i], (uLong)zlen[i]) != Z_OK || got != raw[i]) { fprintf(stderr, "section %d does not inflate\n", i); return 1; }
  }
  CHECK(dctzhip_memcpy_h2d(g, d_bin, h[0], raw[0])); CHECK(dctzhip_memcpy_h2d(g, d_dc, h[1], raw[1]));
  if (raw[2]) CHECK(dctzhip_memcpy_h2d(g, d_ac, h[2], raw[2]));
  CHECK(dctzhip_decompress(g, d_bin, (const float *)d_dc, (const float *)d_ac, info.cnt, NULL, N, DCTZHIP_F64, error_bound, info.sf, DCTZHIP_EC, d_out));
  CHECK(dctzhip_memcpy_d2h(g, r, d_out, N * 8));
  double maxerr = 0.0;
  for (size_t i = 0; i < N; i++) { const double e = fabs(r[i] * info.sf - x[i]); if (e > maxerr) maxerr = e; }
  printf("ENTROPY %zu %u %zu %zu %.6g\n", N, info.cnt, raw_total, z_total, maxerr / info.sf);
  dctzhip_ctx_destroy(g);
  return 0;
}
