/* pdeflate.c -- chunked, multi-threaded deflate that still yields ONE zlib stream
 * per section, so the reference's reader (dctz-decomp-lib.c:244-322: inflateInit +
 * one inflate() per section) decodes it unchanged.
 *
 * the reference deflates bin_index / DC / AC_exact
 * on three threads, one stream each (dctz-comp-lib.c:620-732), which is 85-90 %
 * of its compress wall time; with the block-DCT stage on the GPU it is all that
 * is left.  Method (the pigz construction): cut the input into chunks; every
 * chunk is deflated independently as a RAW deflate stream (windowBits -15), primed
 * with the last 32 KiB of the previous chunk as dictionary, and ended with
 * Z_SYNC_FLUSH (byte-aligned, not final) -- the last chunk with Z_FINISH; the
 * section is   0x78 0x9C | chunk outputs back to back | adler32(input) big-endian.
 * Deflate parameters are the reference's (dctz-comp-lib.c:642-643: default level,
 * 32 KiB window, memLevel 8, default strategy).
 *
 * Compressed BYTES differ from a single-shot deflate (so do they between zlib
 * versions); inflated CONTENT is identical.  The drop-in library keeps
 * the reference's one-stream-per-thread behaviour by default and uses this only
 * when DCTZ_ZLIB_THREADS > 3 (libdctz.c).
 */
#include "pdeflate.h"

#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#ifndef DEF_MEM_LEVEL
#define DEF_MEM_LEVEL 8
#endif

#define PD_DICT 32768u

typedef struct {
  const unsigned char *src;   /* chunk start */
  size_t len;
  const unsigned char *dict;  /* up to 32 KiB before the chunk (NULL for the first) */
  size_t dict_len;
  int last;                   /* Z_FINISH instead of Z_SYNC_FLUSH */
  unsigned char *dst;
  size_t cap, out_len;
  uLong adler;                /* adler32 of the chunk alone */
  int err;
} pd_chunk;

typedef struct {
  pd_chunk *chunks;
  size_t nchunks;
  size_t next;                /* work counter */
  pthread_mutex_t mu;
} pd_queue;

/* deflate level of the chunk jobs: the reference's Z_DEFAULT_COMPRESSION (dctz-comp-lib.c:642)
 * unless dctz_pdeflate_set_level() chose another (DCTZ_ZLIB_LEVEL in the drop-in library) */
static int g_level = Z_DEFAULT_COMPRESSION;
void dctz_pdeflate_set_level(int level) { g_level = (level >= 1 && level <= 9) ? level : Z_DEFAULT_COMPRESSION; }

static void pd_do_chunk(pd_chunk *c) {
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (deflateInit2(&zs, g_level, Z_DEFLATED, -15, DEF_MEM_LEVEL, Z_DEFAULT_STRATEGY) != Z_OK) {
    c->err = 1;
    return;
  }
  if (c->dict_len) deflateSetDictionary(&zs, c->dict, (uInt)c->dict_len);
  zs.next_in = (Bytef *)c->src;
  zs.avail_in = (uInt)c->len;
  zs.next_out = c->dst;
  zs.avail_out = (uInt)c->cap;
  const int rc = deflate(&zs, c->last ? Z_FINISH : Z_SYNC_FLUSH);
  if ((c->last && rc != Z_STREAM_END) || (!c->last && (rc != Z_OK || zs.avail_in != 0 || zs.avail_out == 0))) c->err = 1;
  c->out_len = zs.total_out;
  deflateEnd(&zs);
  c->adler = adler32(adler32(0L, Z_NULL, 0), c->src, (uInt)c->len);
}

static void *pd_worker(void *arg) {
  pd_queue *q = (pd_queue *)arg;
  for (;;) {
    pthread_mutex_lock(&q->mu);
    const size_t i = q->next < q->nchunks ? q->next++ : (size_t)-1;
    pthread_mutex_unlock(&q->mu);
    if (i == (size_t)-1) break;
    pd_do_chunk(&q->chunks[i]);
  }
  return NULL;
}

size_t dctz_pdeflate_bound(size_t n, size_t chunk) {
  if (chunk < PD_DICT) chunk = PD_DICT;
  const size_t nchunks = n ? (n + chunk - 1) / chunk : 1;
  /* per chunk: deflate's own bound + the 5-byte empty stored block of a sync flush */
  return 2 + 4 + nchunks * (compressBound((uLong)chunk) + 16);
}

int dctz_pdeflate_many(const dctz_pd_section *sec, int nsec, int threads, size_t chunk) {
  if (chunk < PD_DICT) chunk = PD_DICT;
  if (chunk > ((size_t)1 << 30)) chunk = (size_t)1 << 30;
  if (threads < 1) threads = 1;
  size_t total = 0;
  for (int s = 0; s < nsec; s++) total += sec[s].n ? (sec[s].n + chunk - 1) / chunk : 1;
  pd_chunk *chunks = (pd_chunk *)calloc(total, sizeof(pd_chunk));
  if (!chunks) return -1;
  const size_t per = compressBound((uLong)chunk) + 16;
  unsigned char *scratch = (unsigned char *)malloc(total * per);
  if (!scratch) { free(chunks); return -1; }
  size_t k = 0;
  for (int s = 0; s < nsec; s++) {
    const unsigned char *p = (const unsigned char *)sec[s].src;
    const size_t n = sec[s].n, nch = n ? (n + chunk - 1) / chunk : 1;
    for (size_t i = 0; i < nch; i++, k++) {
      pd_chunk *c = &chunks[k];
      const size_t off = i * chunk;
      c->src = p + off;
      c->len = n - off < chunk ? n - off : chunk;
      c->dict_len = off < PD_DICT ? off : PD_DICT;
      c->dict = c->dict_len ? p + off - c->dict_len : NULL;
      c->last = (i + 1 == nch);
      c->dst = scratch + k * per;
      c->cap = per;
    }
  }
  pd_queue q;
  q.chunks = chunks; q.nchunks = total; q.next = 0;
  pthread_mutex_init(&q.mu, NULL);
  if ((size_t)threads > total) threads = (int)total;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
  int started = 0;
  if (th)
    for (int t = 0; t < threads - 1; t++) {
      if (pthread_create(&th[started], NULL, pd_worker, &q)) break;
      started++;
    }
  pd_worker(&q);                                   /* the caller's thread works too */
  for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
  free(th);
  pthread_mutex_destroy(&q.mu);

  int rc = 0;
  k = 0;
  for (int s = 0; s < nsec && rc == 0; s++) {
    const size_t n = sec[s].n, nch = n ? (n + chunk - 1) / chunk : 1;
    unsigned char *out = (unsigned char *)sec[s].dst;
    size_t pos = 0;
    uLong adler = adler32(0L, Z_NULL, 0);
    if (sec[s].cap < 6) { rc = -2; break; }
    out[pos++] = 0x78;                             /* CMF: deflate, 32 KiB window            */
    out[pos++] = 0x9C;                             /* FLG: "default level" hint (informative only), no preset dictionary */
    for (size_t i = 0; i < nch; i++, k++) {
      const pd_chunk *c = &chunks[k];
      if (c->err) { rc = -3; break; }
      if (pos + c->out_len + 4 > sec[s].cap) { rc = -2; break; }
      memcpy(out + pos, c->dst, c->out_len);
      pos += c->out_len;
      adler = adler32_combine(adler, c->adler, (z_off_t)c->len);
    }
    if (rc) break;
    out[pos++] = (unsigned char)(adler >> 24);
    out[pos++] = (unsigned char)(adler >> 16);
    out[pos++] = (unsigned char)(adler >> 8);
    out[pos++] = (unsigned char)adler;
    *sec[s].out_len = pos;
  }
  free(scratch);
  free(chunks);
  return rc;
}

int dctz_pdeflate(const void *src, size_t n, void *dst, size_t cap, size_t *out_len, int threads, size_t chunk) {
  dctz_pd_section s;
  s.src = src; s.n = n; s.dst = dst; s.cap = cap; s.out_len = out_len;
  return dctz_pdeflate_many(&s, 1, threads, chunk);
}
