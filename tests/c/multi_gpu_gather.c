/* multi_gpu_gather.c -- TEST / EXAMPLE: one process per GPU drives the hot path through the C ABI (include/dctz_hip.h)
 * and gathers the pre-zlib streams of every shard on rank 0 over RCCL.
 *   DCTZ_RANK / DCTZ_WORLD   this process's rank and the number of processes (default 0 / 1)
 *   DCTZ_COMM_ID_FILE        where rank 0 leaves the 128-byte communicator id for the others (world > 1)
 * usage: multi_gpu_gather <elements per shard> <error bound>
 * Prints "GATHER rank world n_total cnt_total checksum" on rank 0. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <unistd.h>
#include "dctz_hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != DCTZHIP_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, dctzhip_last_error(ctx)); return 1; } } while (0)

int main(int argc, char **argv) {
  const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)1 << 20;
  const double eb = argc > 2 ? atof(argv[2]) : 1e-3;
  const int rank = getenv("DCTZ_RANK") ? atoi(getenv("DCTZ_RANK")) : 0;
  const int world = getenv("DCTZ_WORLD") ? atoi(getenv("DCTZ_WORLD")) : 1;
  const size_t nblk = (n + 63) / 64;
  dctzhip_ctx *ctx = NULL;
  const int ndev = dctzhip_device_count();
  if (ndev < 1) { fprintf(stderr, "no GPU\n"); return 1; }
  CHECK(dctzhip_ctx_create(&ctx, rank % ndev));

  /* the communicator: rank 0 makes the id, the others read it (any side channel will do) */
  unsigned char id[DCTZHIP_COMM_ID_BYTES];
  const char *idfile = getenv("DCTZ_COMM_ID_FILE");
  if (rank == 0) {
    CHECK(dctzhip_comm_unique_id(id));
    if (world > 1) {
      char tmp[4096];
      snprintf(tmp, sizeof tmp, "%s.tmp", idfile);
      FILE *f = fopen(tmp, "wb");
      if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) return 1;
      fclose(f);
      rename(tmp, idfile);
    }
  } else {
    FILE *f = NULL;
    for (int tries = 0; tries < 600 && !(f = fopen(idfile, "rb")); tries++) usleep(100000);
    if (!f || fread(id, 1, sizeof id, f) != sizeof id) return 1;
    fclose(f);
  }
  CHECK(dctzhip_comm_create(ctx, rank, world, id));

  /* this rank's shard (seeded by the rank), resident on its GPU */
  double *x = malloc(n * sizeof(double));
  for (size_t i = 0; i < n; i++) x[i] = 37.0 * sin((double)i / 97.0 + rank) + 0.5 * cos((double)i * 0.37);
  void *d_x, *d_bin, *d_dc, *d_ac;
  CHECK(dctzhip_malloc(ctx, &d_x, n * sizeof(double)));
  CHECK(dctzhip_malloc(ctx, &d_bin, n));
  CHECK(dctzhip_malloc(ctx, &d_dc, nblk * sizeof(float)));
  CHECK(dctzhip_malloc(ctx, &d_ac, n * sizeof(float)));
  CHECK(dctzhip_memcpy_h2d(ctx, d_x, x, n * sizeof(double)));
  dctzhip_cinfo info;
  CHECK(dctzhip_compress(ctx, d_x, n, DCTZHIP_F64, eb, DCTZHIP_EC, d_bin, d_dc, d_ac, NULL, NULL, &info));

  /* sizes of every shard, then the streams to rank 0 */
  uint64_t *sizes = malloc(3 * (size_t)world * sizeof(uint64_t));
  CHECK(dctzhip_comm_sizes(ctx, n, info.cnt, sizes));
  void *d_bin_all = NULL, *d_dc_all = NULL, *d_ac_all = NULL;
  uint64_t tn = 0, tb = 0, tc = 0;
  for (int r = 0; r < world; r++) { tn += sizes[3 * r]; tb += sizes[3 * r + 1]; tc += sizes[3 * r + 2]; }
  if (rank == 0) {
    CHECK(dctzhip_malloc(ctx, &d_bin_all, tn));
    CHECK(dctzhip_malloc(ctx, &d_dc_all, tb * sizeof(float)));
    CHECK(dctzhip_malloc(ctx, &d_ac_all, (tc ? tc : 1) * sizeof(float)));
  }
  CHECK(dctzhip_comm_gather(ctx, 0, d_bin, d_dc, d_ac, sizes, d_bin_all, d_dc_all, d_ac_all));
  if (rank == 0) {                       /* the host zlib tail of all shards would start here (dctz-comp-lib.c:620-760) */
    unsigned char *bins = malloc(tn);
    CHECK(dctzhip_memcpy_d2h(ctx, bins, d_bin_all, tn));
    unsigned long long sum = 0;
    for (uint64_t i = 0; i < tn; i++) sum += bins[i];
    printf("GATHER %d %d %llu %llu %llu\n", rank, world, (unsigned long long)tn, (unsigned long long)tc, sum);
    free(bins);
  }
  CHECK(dctzhip_comm_destroy(ctx));
  dctzhip_ctx_destroy(ctx);
  free(x); free(sizes);
  return 0;
}
