/* dctz_cli.c -- command-line harness with the reference's argv / stdout / file-name
 * contract, on top of the drop-in library:
 *
 *   dctz-ec-test | dctz-qt-test  -d|-f  <err bound>  <var name>  <src file>  <dim1> [dim2 [dim3 [dim4]]]
 *
 * Behaviour followed (file:line in the reference tree):
 *   - usage text + exit(0) when fewer than 6 arguments          dctz-test.c:40-50
 *   - N = product of the 1..4 dimension sizes                   dctz-test.c:77-91
 *   - prints "total number of elements = %d"                    dctz-test.c:94
 *   - <src>.{ec|qt}.<err bound as typed>.z   (compressed)       dctz-test.c:99-103, 222-236
 *     <src>.{ec|qt}.<err bound as typed>.z.r (reconstruction)   dctz-test.c:238-243, 258-266
 *   - the compressed buffer is N*type_size bytes                dctz-test.c:143, 158
 *   - after dctz_compress() the input (scaled in place by the library) is multiplied
 *     back by the header's scaling factor before PSNR           dctz-test.c:188-210
 *   - prints "oriFilePath = ..., outputFilePath = ..., datatype = ..., error = ..., dim1.."
 *     then "outsize = %zu", finally "CR = %.2f, PSNR = %.2f" and "done"
 *                                                               dctz-test.c:183-184, 274-283
 * Built twice like the reference (Makefile:12-17): -DUSE_QTABLE selects the .qt. names and
 * links libdctz-qt.so.  tests/test-dctz.sh and test-dctz-f.sh of the reference run unchanged
 * against these binaries.  The Z-checker hooks (WITH_Z_CHECKER) are out of scope.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dctz.h"

#ifdef USE_QTABLE
#define VARIANT "qt"
#else
#define VARIANT "ec"
#endif

static void usage(const char *prog) {
  printf("Test case: %s -d|-f [err bound] [var name] [srcFilePath] [dimension sizes...] \n", prog);
  printf("Example: %s -d 1E-3 sedov testdata/x86/testfloat_8_8_128.dat 8 8 128 \n", prog);
}

static void *xmalloc(size_t bytes, const char *what) {
  void *p = malloc(bytes ? bytes : 1);
  if (!p) {
    fprintf(stderr, "Out of memory: %s\n", what);
    exit(1);
  }
  return p;
}

static void set_buf(t_var *v, t_datatype dt, void *p) {
  v->datatype = dt;
  if (dt == DOUBLE) v->buf.d = (double *)p;
  else v->buf.f = (float *)p;
}

static int write_file(const char *path, const void *p, size_t bytes) {
  FILE *fp = fopen(path, "wb");
  if (!fp) return 0;
  const size_t ok = bytes ? fwrite(p, bytes, 1, fp) : 1;
  fclose(fp);
  return ok == 1;
}

int main(int argc, char *argv[]) {
  if (argc < 6) {
    usage(argv[0]);
    exit(0);
  }
  const t_datatype dt = strcmp(argv[1], "-d") == 0 ? DOUBLE : FLOAT;
  const size_t ts = dt == DOUBLE ? sizeof(double) : sizeof(float);
  const char *eb_text = argv[2];
  const double eb = atof(eb_text);
  const char *src = argv[4];

  size_t dim[4] = {0, 0, 0, 0};
  int N = 1;
  for (int i = 0; i < 4 && 5 + i < argc; i++) {
    dim[i] = (size_t)atoi(argv[5 + i]);
    N *= (int)dim[i];
  }
  printf("total number of elements = %d\n", N);

  char zpath[640], rpath[660];
  snprintf(zpath, sizeof(zpath), "%s." VARIANT ".%s.z", src, eb_text);
  snprintf(rpath, sizeof(rpath), "%s.r", zpath);

  FILE *fp = fopen(src, "rb");
  if (!fp) {
    perror("Failed: ");
    printf("File Not Found\n");
    return 1;
  }
  const size_t bytes = (size_t)N * ts;
  void *orig = xmalloc(bytes, "org_buf");
  void *recon = xmalloc(bytes, "reconst_buf");
  void *comp = xmalloc(bytes, "comp_buf");
  if (fread(orig, ts, (size_t)N, fp) != (size_t)N) {
    perror("Error reading file");
    exit(EXIT_FAILURE);
  }
  fclose(fp);

  t_var var, var_r, var_z;
  memset(&var, 0, sizeof(var)); memset(&var_r, 0, sizeof(var_r)); memset(&var_z, 0, sizeof(var_z));
  set_buf(&var, dt, orig);
  set_buf(&var_r, dt, recon);
  set_buf(&var_z, dt, comp);

  /* extension (not in the reference, whose library ignores the extents, dctz-test.c:77-91): with DCTZ_ND_BLOCKS set
   * and 2 or 3 extents on the command line the blocks are 8 x 8 / 4 x 4 x 4 tiles (dctz.h: dctz_set_block_dims).  The
   * command line lists the FASTEST extent first, as the reference's data lists do (tests/list-CESM-ATM-tylor.txt:1:
   * a 1800 x 3600 field is "3600 1800"); the library wants row-major order. */
  if (getenv("DCTZ_ND_BLOCKS") && dim[1] && !dim[3]) {
    size_t rm[3];
    const int nd = dim[2] ? 3 : 2;
    for (int i = 0; i < nd; i++) rm[i] = dim[nd - 1 - i];
    if (dctz_set_block_dims(nd, rm) != 0) { printf("bad dimension sizes for multi-dimensional blocks\n"); exit(1); }
    printf("multi-dimensional blocks: %s tiles\n", nd == 2 ? "8 x 8" : "4 x 4 x 4");
  }
  size_t out_size = 0;
  dctz_compress(&var, N, &out_size, &var_z, eb);
  printf("oriFilePath = %s, outputFilePath = %s, datatype = %s, error = %s, dim1 = %zu, dim2 = %zu, dim3 = %zu, dim4 = %zu\n",
         src, zpath, dt == FLOAT ? "float" : "double", eb_text, dim[0], dim[1], dim[2], dim[3]);
  printf("outsize = %zu\n", out_size);

  /* the library scaled the caller's array in place: undo it with the header's factor */
  struct header h;
  memcpy(&h, comp, sizeof(h));
  if (dt == DOUBLE) {
    if (h.scaling_factor.d != 1.0)
      for (int i = 0; i < N; i++) var.buf.d[i] *= h.scaling_factor.d;
  } else {
    if (h.scaling_factor.f != 1.0)
      for (int i = 0; i < N; i++) var.buf.f[i] *= h.scaling_factor.f;
  }

  if (!write_file(zpath, comp, out_size)) {
    printf("Write qtz file failed: %lu != %d!\n", (unsigned long)out_size, 0);
    exit(1);
  }
  dctz_decompress(&var_z, &var_r);
  if (!write_file(rpath, recon, bytes)) {
    printf("Write qtz.r file failed:  != %d!\n", 0);
    exit(1);
  }

  const double cr = (double)bytes / (double)out_size;
  const double psnr = calc_psnr(&var, &var_r, N, eb);
  printf("CR = %.2f, PSNR = %.2f\n", cr, psnr);
  free(comp); free(recon); free(orig);
  printf("done\n");
  return 0;
}
