/* dctz_dump.c -- inspect a DCTZ container.
 *
 *   dctz-dump <file.z>            the reference tool's six lines (tools/dctz-dump.c:41-50)
 *   dctz-dump -v <file.z>         + section sizes, offsets, a bounds check of the whole layout
 *                                   against the file size, and (QT files) the table's first entries
 *
 * The header is `struct header` of dctz.h:96-119: 56 bytes, native little-endian, then
 * deflate(bin_index[N]) | deflate(DC[nblk] as float) | deflate(AC_exact[cnt] as float)
 * [| qtable[64] in the data type]  (dctz-comp-lib.c:775-820).  EC and QT headers have the
 * same size (the QT-only bindex_count sits in what is padding in the EC layout), so one
 * binary reads both; -v tells them apart by the file size.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define USE_QTABLE 1 /* the larger view of the header: bindex_count is readable for both variants */
#include "dctz.h"

int main(int argc, char *argv[]) {
  int verbose = 0;
  const char *path = NULL;
  if (argc == 2) path = argv[1];
  else if (argc == 3 && strcmp(argv[1], "-v") == 0) { verbose = 1; path = argv[2]; }
  if (!path) {
    printf("Usage: %s filename\n", argv[0]);
    exit(0);
  }
  FILE *fp = fopen(path, "rb");
  if (!fp) {
    perror("Failed: ");
    printf("File Not Found\n");
    return 0;
  }
  struct header h;
  if (fread(&h, sizeof(h), 1, fp) != 1) {
    printf("%s: shorter than a DCTZ header (%zu bytes)\n", path, sizeof(h));
    fclose(fp);
    return 1;
  }
  printf("File Name=%s\n", path);
  const unsigned geom = DCTZ_GEOM_OF(h.datatype);          /* 0: the reference's flat blocks; 2, 3: tiles (dctz.h) */
  h.datatype = DCTZ_TYPE_OF(h.datatype);
  printf("data type=%s\n", (h.datatype == DOUBLE) ? "double" : "float");
  printf("N=%d\n", h.num_elements);
  printf("error_bound=%f\n", h.error_bound);
  printf("total # of AC_exact=%d\n", h.tot_AC_exact_count);
  printf("SF=%f\n", h.datatype == DOUBLE ? h.scaling_factor.d : (double)h.scaling_factor.f);

  int rc = 0;
  if (verbose) {
    fseek(fp, 0, SEEK_END);
    const long fsz_all = ftell(fp);
    const size_t ts = h.datatype == DOUBLE ? sizeof(double) : sizeof(float);
    /* "DZIX" chunk index behind everything else (written when the entropy stage ran on the GPU, dctz.h): look where an
     * ec and where a qt container would have it */
    long ix_bytes = 0;
    unsigned int ixh[5] = {0, 0, 0, 0, 0};
    for (int qt = 0; qt < 2 && !ix_bytes; qt++) {
      const long pos = (long)(sizeof(h) + (size_t)h.bindex_sz_compressed + h.DC_sz_compressed + h.AC_exact_sz_compressed +
                              (qt ? BLK_SZ * ts : 0) + (geom ? 16 : 0));
      if (pos + 20 <= fsz_all && fseek(fp, pos, SEEK_SET) == 0 && fread(ixh, sizeof(ixh), 1, fp) == 1 && ixh[0] == DCTZ_IX_MAGIC) {
        const long want = (long)((20 + 2 * ((size_t)ixh[2] + ixh[3] + ixh[4]) + 3) & ~(size_t)3);
        if (pos + want == fsz_all) ix_bytes = want;
      }
    }
    const long fsz = fsz_all - ix_bytes;
    size_t nblk = ((size_t)h.num_elements + BLK_SZ - 1) / BLK_SZ;
    const size_t o0 = sizeof(h), o1 = o0 + h.bindex_sz_compressed, o2 = o1 + h.DC_sz_compressed;
    size_t end = o2 + h.AC_exact_sz_compressed;
    size_t npos = h.num_elements;
    if (geom) {                                             /* "DZND" + three extents close the file */
      unsigned int tr[4] = {0, 0, 0, 0};
      if (fsz >= 16 && fseek(fp, fsz - 16, SEEK_SET) == 0 && fread(tr, sizeof(tr), 1, fp) == 1 && tr[0] == DCTZ_ND_MAGIC) {
        const size_t e = geom == 2 ? 8 : 4;
        nblk = 1;
        for (unsigned i = 0; i < geom; i++) nblk *= (tr[1 + i] + e - 1) / e;
        npos = nblk * BLK_SZ;
        if (geom == 2) printf("multi-dimensional blocks: %u x %u array, 8 x 8 tiles\n", tr[1], tr[2]);
        else printf("multi-dimensional blocks: %u x %u x %u array, 4 x 4 x 4 tiles\n", tr[1], tr[2], tr[3]);
      } else {
        printf("LAYOUT MISMATCH: geometry %u in the header but no extents at the end of the file\n", geom);
        rc = 2;
      }
    }
    printf("mean=%.17g\n", h.datatype == DOUBLE ? h.mean.d : (double)h.mean.f);
    printf("blocks=%zu (last one %zu elements)\n", nblk, (!geom && h.num_elements % BLK_SZ) ? (size_t)(h.num_elements % BLK_SZ) : (size_t)BLK_SZ);
    printf("bin_index: offset %zu, %u bytes deflated (%zu raw)\n", o0, h.bindex_sz_compressed, npos);
    printf("DC:        offset %zu, %u bytes deflated (%zu raw)\n", o1, h.DC_sz_compressed, nblk * sizeof(float));
    printf("AC_exact:  offset %zu, %u bytes deflated (%zu raw)\n", o2, h.AC_exact_sz_compressed,
           (size_t)h.tot_AC_exact_count * sizeof(float));
    const size_t trailer = geom ? 16 : 0;
    end += trailer;
    if ((size_t)fsz == end) {
      printf("variant=ec (no table), file size %ld = layout\n", fsz);
    } else if ((size_t)fsz == end + BLK_SZ * ts) {
      printf("variant=qt, bindex_count=%u, table at offset %zu, file size %ld = layout\n", h.bindex_count, end - trailer, fsz);
      unsigned char q[BLK_SZ * sizeof(double)];
      fseek(fp, (long)(end - trailer), SEEK_SET);
      if (fread(q, ts, BLK_SZ, fp) == BLK_SZ) {
        printf("qtable[1..4]=");
        for (int j = 1; j <= 4; j++) {
          double v;
          if (ts == 8) memcpy(&v, q + 8 * j, 8);
          else { float f; memcpy(&f, q + 4 * j, 4); v = f; }
          printf("%s%.9g", j > 1 ? ", " : "", v);
        }
        printf("\n");
      }
    } else {
      printf("LAYOUT MISMATCH: header describes %zu bytes (ec) or %zu (qt), file has %ld\n", end, end + BLK_SZ * ts, fsz);
      rc = 2;
    }
    if (ix_bytes) {
      printf("chunk index: %ld bytes at offset %ld, chunks of %u bytes: %u + %u + %u (sections made by the GPU entropy stage)\n", ix_bytes, fsz,
             ixh[1], ixh[2], ixh[3], ixh[4]);
      /* the sizes of a section's chunks + 2 (zlib header) + 6 (03 00 + adler32) must be the section's size */
      const unsigned int zs[3] = {h.bindex_sz_compressed, h.DC_sz_compressed, h.AC_exact_sz_compressed};
      int tiles = 1;
      if (fseek(fp, fsz + 20, SEEK_SET) == 0) {
        for (int i = 0; i < 3; i++) {
          unsigned long long sum = 8;
          for (unsigned int k = 0; k < ixh[2 + i]; k++) { unsigned short e = 0; if (fread(&e, 2, 1, fp) != 1) { tiles = 0; break; } sum += e; }
          if (sum != zs[i]) tiles = 0;
        }
      } else tiles = 0;
      printf("chunk index %s\n", tiles ? "tiles the three streams" : "does NOT tile the streams");
      if (!tiles) rc = 2;
    }
    printf("compression ratio=%.2f\n", (double)h.num_elements * ts / (double)fsz_all);
  }
  fclose(fp);
  return rc;
}
