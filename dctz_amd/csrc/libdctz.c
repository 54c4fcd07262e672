/*
 * libdctz.c -- drop-in host library: the reference's public API (include/dctz.h;
 * upstream dctz.h:121-128 and dct.h:17-27) over the MI355X hot path
 * (include/dctz_hip.h).  Plain C, built twice like the reference's Makefile:12-17:
 *   libdctz-ec.so   -DUSE_TRUNCATE
 *   libdctz-qt.so   -DUSE_TRUNCATE -DUSE_QTABLE
 *
 * What runs where
 *   GPU  : calc_data_stat, scaling, block DCT-II/III, binning, QT table, ordered
 *          AC_exact compaction, de-quantisation, de-scaling (dctz-comp-lib.c:186-544,
 *          dctz-decomp-lib.c:358-511) -- every arithmetic step of the hot path.
 *   host : PCIe copies, the zlib tail on 3 pthreads (dctz-comp-lib.c:620-732,
 *          kept as the reference has it) and the 56-byte container header
 *          (dctz.h:96-119; writer dctz-comp-lib.c:775-820, reader
 *          dctz-decomp-lib.c:84-100,186-199).
 * There is no CPU implementation of the GPU stages here: if the HIP library
 * cannot create a context the process exits(1), the reference's own error
 * convention (dctz-comp-lib.c:123-126).
 *
 * Observable behaviour kept from the reference: both entry points return 1;
 * error_bound < 1e-6 prints "ERROR BOUND is not acceptable" and exits(1)
 * (dctz-comp-lib.c:135-138); var->buf is scaled IN PLACE by 1/sf (:193-216);
 * "outSize = ..." / "uncompressed bin_index size is: ..." go to stdout (:841-843,
 * dctz-decomp-lib.c:260-262) unless DCTZ_QUIET is set.  The unconditional dump
 * files ./bin_index.bin and ./AC_exact.bin (:583-595; ./qtable.bin :443-448) are
 * written only when DCTZ_DUMP_STREAMS is set.
 */
#define _GNU_SOURCE
#include "dctz.h"

#include <pthread.h>
#include <sched.h>
#include <unistd.h>
#include "pdeflate.h"
#include <stdint.h>
#include <sys/time.h>

#include "dctz_hip.h"

#ifndef USE_TRUNCATE
#error "build with -DUSE_TRUNCATE (every reference target does, Makefile:13-24)"
#endif

#define DEF_MEM_LEVEL 8 /* dctz-comp-lib.c:25 */

#ifdef USE_QTABLE
#define DCTZ_MODE DCTZHIP_QT
#else
#define DCTZ_MODE DCTZHIP_EC
#endif

/* ------------------------------------------------------------------ state -- */
static dctzhip_ctx *g_ctx = NULL;
static struct {
  void *in, *bin, *dc, *ac, *out, *z[3];
  size_t in_cap, bin_cap, dc_cap, ac_cap, out_cap, z_cap[3];
} g_dev;
/* Host copies of the raw streams (what the zlib tails read / the inflates write), kept between calls like the device
 * buffers above: a fresh 170 MB allocation per call costs its page faults on the way in and an munmap on the way out
 * (about 25 ms per GiB shard, more than the indexed inflate itself). */
static struct {
  void *p[3];
  size_t cap[3];
} g_host;
static void *host_buf(int i, size_t need) {
  if (need == 0) need = 1;
  if (need > g_host.cap[i]) {
    free(g_host.p[i]);
    g_host.p[i] = malloc(need);
    g_host.cap[i] = g_host.p[i] ? need : 0;
    if (!g_host.p[i]) { fprintf(stderr, "Out of memory: streams\n"); exit(1); }
  }
  return g_host.p[i];
}
static dctz_stage_times g_times;
/* multi-dimensional blocks requested for the next dctz_compress call (dctz.h: dctz_set_block_dims) */
static int g_nd = 0;
static size_t g_dims[3] = {0, 0, 0};

static double now_s(void) {
  struct timeval tv;
  gettimeofday(&tv, NULL);
  return (double)tv.tv_sec + 1e-6 * (double)tv.tv_usec;
}

static int quiet(void) { return getenv("DCTZ_QUIET") != NULL; }
/* DCTZ_ZLIB_THREADS: unset / <= 3 = the reference's tail (three threads, one single-shot
 * deflate each, dctz-comp-lib.c:620-732).  > 3: chunked deflate on that many threads (pdeflate.c;
 * still one zlib stream per section).  The reader inflates the three sections side by side unless
 * DCTZ_ZLIB_THREADS is 1..3 (then one after the other, dctz-decomp-lib.c:244-322: same bytes out either way).
 * DCTZ_ZLIB_CHUNK: bytes per deflate job (default 256 KiB).  DCTZ_ZLIB_LEVEL: 1..9 for the chunked tail
 * only (default: zlib's default level, as the reference; a lower level trades ratio for host time). */
static int zlib_threads(void) { const char *e = getenv("DCTZ_ZLIB_THREADS"); return e ? atoi(e) : 0; }
/* DCTZ_ZLIB_GPU=1: the entropy stage runs on the device too (include/dctz_hip.h: dctzhip_deflate):
 * the three sections are deflated where k_compress left them and only compressed bytes come back over PCIe.  One zlib
 * stream per section as before (any inflate reads it, dctz-decomp-lib.c:244-322); the bytes -- and the sizes, by a per
 * cent or so -- differ from zlib's, which is why the reference's tail stays the default. */
static int zlib_gpu(void) { const char *e = getenv("DCTZ_ZLIB_GPU"); return e && atoi(e) != 0; }
/* DCTZ_FAST_MEAN=1: header.mean = the tree-order sum / N that the compress kernels produce anyway, instead of the
 * reference's serial-order sum (util.c:18-28).  The field is written but never read back (dctz-decomp-lib.c:499 is
 * commented out); the serial chain of N dependent additions is what bounds a call once the entropy stage is on the GPU
 * (50 ms per GiB on one host core).  Off by default: the header is then bit-identical to the reference's. */
static int fast_mean(void) { const char *e = getenv("DCTZ_FAST_MEAN"); return e && atoi(e) != 0; }
/* DCTZ_SCALE_HOST: 1 = x / sf is written into the caller's buffer by host threads, 0 = by the GPU + a D2H copy.
 * Default: host threads exactly when the entropy stage runs on the GPU (the host cores are idle then). */
static int scale_on_host(int gpu_tail) { const char *e = getenv("DCTZ_SCALE_HOST"); return e ? atoi(e) != 0 : gpu_tail; }
static size_t zlib_chunk(void) {
  const char *e = getenv("DCTZ_ZLIB_CHUNK");
  long long v = e ? atoll(e) : 0;
  return v >= 32768 ? (size_t)v : (size_t)262144;
}

static void die(const char *what) {
  fprintf(stderr, "libdctz: %s: %s\n", what, dctzhip_last_error(g_ctx));
  exit(1);
}

/* Which GPU: DCTZ_DEVICE (an index), else the process's rank on its node -- DCTZ_RANK, or what the usual launchers export
 * (LOCAL_RANK, OMPI_COMM_WORLD_LOCAL_RANK, SLURM_LOCALID) -- modulo the GPUs visible: one process per GPU, every process
 * compresses its own shards (they are independent dctz_compress calls), see INTEGRATION.md section D. */
static int pick_device(void) {
  const char *d = getenv("DCTZ_DEVICE");
  if (d) return atoi(d);
  const char *names[] = {"DCTZ_RANK", "LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID"};
  for (int i = 0; i < 4; i++) {
    const char *r = getenv(names[i]);
    if (r && *r) { const int n = dctzhip_device_count(); return n > 0 ? atoi(r) % n : -1; }
  }
  return -1;                                   /* the current HIP device */
}

static dctzhip_ctx *ctx(void) {
  if (!g_ctx) {
    if (dctzhip_ctx_create(&g_ctx, pick_device()) != DCTZHIP_OK) {
      fprintf(stderr, "libdctz: no usable MI355X context: %s\n", dctzhip_last_error(NULL));
      exit(1);
    }
  }
  return g_ctx;
}

static void grow(void **p, size_t *cap, size_t need) {
  if (need <= *cap) return;
  if (*p && dctzhip_free(ctx(), *p) != DCTZHIP_OK) die("dctzhip_free");
  *p = NULL;
  *cap = 0;
  if (dctzhip_malloc(ctx(), p, need) != DCTZHIP_OK) die("Out of memory (device)");
  *cap = need;
}

void dctz_last_stage_times(dctz_stage_times *t) { if (t) *t = g_times; }

/* ------------------------------------------------------------- zlib tail --- */
/* One deflate stream per thread, finished in one call; the compressed size is
 * the thread's exit value (dctz-comp-lib.c:75-88). */
void *compress_thread(void *arg) {
  z_stream *zs = (z_stream *)arg;
  deflate(zs, Z_FINISH);
  uLong produced = zs->total_out;
  deflateEnd(zs);
  pthread_exit((void *)produced);
}

/* Our own section jobs feed zlib in pieces of at most 1 GiB: avail_in / avail_out are 32-bit (uInt), and a section can
 * be larger (AC_exact: 4 bytes x up to 2^31 - 1 exceptions).  Same stream as one deflate(Z_FINISH) call of the whole. */
static size_t z_piece(void) {                /* DCTZ_ZLIB_PIECE: test hook (tiny pieces exercise the refill loops) */
  static size_t v = 0;
  if (!v) { const char *e = getenv("DCTZ_ZLIB_PIECE"); long long x = e ? atoll(e) : 0; v = (x >= 64 && x <= (1ll << 30)) ? (size_t)x : ((size_t)1 << 30); }
  return v;
}
#define Z_PIECE z_piece()
typedef struct {
  z_stream zs;
  pthread_t th;
  Bytef *dst;
  uLong bound;
  const Bytef *src;
  size_t left;
  int rc;
} zjob;

static void *zjob_main(void *arg) {
  zjob *j = (zjob *)arg;
  z_stream *zs = &j->zs;
  size_t out_left = j->bound;
  zs->next_in = (Bytef *)j->src; zs->avail_in = 0;
  zs->next_out = j->dst; zs->avail_out = 0;
  for (;;) {
    if (zs->avail_in == 0 && j->left) {
      const size_t piece = j->left < Z_PIECE ? j->left : Z_PIECE;
      zs->avail_in = (uInt)piece; j->left -= piece;                /* next_in already points behind the previous piece */
    }
    if (zs->avail_out == 0 && out_left) {
      const size_t piece = out_left < Z_PIECE ? out_left : Z_PIECE;
      zs->avail_out = (uInt)piece; out_left -= piece;
    }
    const int rc = deflate(zs, j->left ? Z_NO_FLUSH : Z_FINISH);
    if (rc == Z_STREAM_END) break;
    if (rc != Z_OK && !(rc == Z_BUF_ERROR && (zs->avail_in == 0 || zs->avail_out == 0) && (j->left || out_left))) { j->rc = rc ? rc : -1; break; }
  }
  uLong produced = zs->total_out;
  deflateEnd(zs);
  return (void *)produced;
}

static void zjob_start(zjob *j, const void *src, size_t nbytes, pthread_attr_t *attr) {
  j->bound = (nbytes <= 0xffffffffu) ? compressBound((uLong)nbytes) : (uLong)(nbytes + nbytes / 1000 + (nbytes >> 25) * 16 + 4096);
  j->dst = (Bytef *)malloc(j->bound ? j->bound : 1);
  if (!j->dst) { fprintf(stderr, "Out of memory: zlib buffer\n"); exit(1); }
  memset(&j->zs, 0, sizeof(j->zs));
  j->zs.zalloc = Z_NULL; j->zs.zfree = Z_NULL; j->zs.opaque = Z_NULL;
  /* dctz-comp-lib.c:642-643: default level, 32K window, memLevel 8, default strategy */
  deflateInit2(&j->zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15, DEF_MEM_LEVEL, Z_DEFAULT_STRATEGY);
  j->zs.data_type = Z_UNKNOWN;
  j->src = (const Bytef *)src; j->left = nbytes; j->rc = 0;
  if (pthread_create(&j->th, attr, zjob_main, j)) {
    fprintf(stderr, "Error creating thread\n");
    exit(0); /* dctz-comp-lib.c:651-654 */
  }
}

static uLong zjob_join(zjob *j) {
  void *ret = NULL;
  pthread_join(j->th, &ret);
  if (j->rc) { fprintf(stderr, "libdctz: deflate failed (%d)\n", j->rc); exit(1); }
  return (uLong)ret;
}

static uLong inflate_into(const Bytef *src, uLong src_len, void *dst, size_t dst_len) {
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  zs.zalloc = Z_NULL; zs.zfree = Z_NULL; zs.opaque = Z_NULL;
  inflateInit(&zs); /* dctz-decomp-lib.c:250 */
  zs.avail_in = (uInt)src_len;                                       /* a section's compressed size is a uint32 of the header */
  zs.next_in = (Bytef *)src;
  zs.next_out = (Bytef *)dst;
  size_t out_left = dst_len;
  for (;;) {                                                         /* the inflated size may exceed 4 GiB */
    if (zs.avail_out == 0 && out_left) {
      const size_t piece = out_left < Z_PIECE ? out_left : Z_PIECE;
      zs.avail_out = (uInt)piece; out_left -= piece;
    }
    const int rc = inflate(&zs, Z_NO_FLUSH);
    if (rc != Z_OK || (zs.avail_out != 0) || !out_left) break;
  }
  uLong produced = zs.total_out;
  inflateEnd(&zs);
  return produced;
}

/* The header's `mean` in the reference's own summation order (util.c:18-28 / :31-41: x[1] + x[2] + ...
 * one after the other, in the data type).  That chain of roundings is inherently sequential:
 * one GPU lane needs ~30 ns per dependent add (k_serial_sum: 4 s per GiB, fine underneath the
 * reference's 7 s zlib tail), a host core ~1 ns.  With the chunked tail the zlib stage is too
 * short to hide the GPU version, so the sum runs on one host thread over the caller's array,
 * which is where the reference computes it too; it must finish before x/sf is copied back. */
#define MEAN_BLOCK ((size_t)1 << 20)          /* elements between two progress reports of the serial sum */
typedef struct {
  const void *x;
  size_t n;
  int is_d;
  double mean;
  size_t progress;              /* elements the sum has consumed (released block by block: the in-place scaling follows behind) */
} host_mean_job;
static void *host_mean_main(void *arg) {
  host_mean_job *j = (host_mean_job *)arg;
  if (j->is_d) {
    const double *x = (const double *)j->x;
    double sum = 0.0;
    for (size_t b = 1; b < j->n; b += MEAN_BLOCK) {
      const size_t e = b + MEAN_BLOCK < j->n ? b + MEAN_BLOCK : j->n;
      for (size_t i = b; i < e; i++) sum += x[i];           /* util.c:22: starts at i = 1 */
      __atomic_store_n(&j->progress, e, __ATOMIC_RELEASE);
    }
    j->mean = sum / (double)(int)j->n;
  } else {
    const float *x = (const float *)j->x;
    float sum = 0.0f;
    for (size_t b = 1; b < j->n; b += MEAN_BLOCK) {
      const size_t e = b + MEAN_BLOCK < j->n ? b + MEAN_BLOCK : j->n;
      for (size_t i = b; i < e; i++) sum += x[i];           /* util.c:35 */
      __atomic_store_n(&j->progress, e, __ATOMIC_RELEASE);
    }
    j->mean = (double)(sum / (float)(int)j->n);
  }
  __atomic_store_n(&j->progress, j->n, __ATOMIC_RELEASE);
  return NULL;
}

/* The reference's in-place "/= sf" (dctz-comp-lib.c:193-216) on host threads, for the device entropy stage: there the
 * tail is too short to hide a 1 GiB write-back over PCIe, and the host cores are idle.  Same operation as the
 * reference's loop (IEEE division in the data type), so the caller's buffer ends up bit-identical to x / sf.  An
 * element may only change once the serial-order mean has read it: the workers take 1 Mi-element blocks in order and
 * follow the sum's progress counter, so the scaling ends a block after the sum does instead of starting there. */
typedef struct {
  void *x;
  size_t n;
  int is_d, threads;
  double sf;
  pthread_t *wait_for;          /* the serial-mean thread (joined here), or NULL */
  host_mean_job *mean;          /* its progress: elements it has read for good, or NULL = all */
  size_t next;                  /* next block to scale (workers take blocks in order) */
} scale_mgr;
static void *scale_worker(void *arg) {
  scale_mgr *m = (scale_mgr *)arg;
  for (;;) {
    const size_t lo = __atomic_fetch_add(&m->next, MEAN_BLOCK, __ATOMIC_RELAXED);
    if (lo >= m->n) break;
    const size_t hi = lo + MEAN_BLOCK < m->n ? lo + MEAN_BLOCK : m->n;
    while (m->mean && __atomic_load_n(&m->mean->progress, __ATOMIC_ACQUIRE) < hi) sched_yield();   /* the sum still needs the originals */
    if (m->is_d) { double *x = (double *)m->x; const double sf = m->sf; for (size_t i = lo; i < hi; i++) x[i] /= sf; }
    else { float *x = (float *)m->x; const float sf = (float)m->sf; for (size_t i = lo; i < hi; i++) x[i] /= sf; }
  }
  return NULL;
}
static void *scale_mgr_main(void *arg) {
  scale_mgr *m = (scale_mgr *)arg;
  int T = m->threads < 1 ? 1 : (m->threads > 64 ? 64 : m->threads);
  if ((size_t)T > m->n / MEAN_BLOCK + 1) T = (int)(m->n / MEAN_BLOCK + 1);
  if (m->mean && T > 6) T = 6;              /* behind the serial sum (one block per ~0.5 ms) a few workers keep up; more would only spin */
  pthread_t th[64];
  int started = 0;
  for (int t = 1; t < T; t++) { if (pthread_create(&th[started], NULL, scale_worker, m) == 0) started++; }
  scale_worker(m);
  for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
  if (m->wait_for) pthread_join(*m->wait_for, NULL);
  return NULL;
}
static int host_threads(void) {
  int t = zlib_threads();
  if (t <= 0) { long nc = sysconf(_SC_NPROCESSORS_ONLN); t = nc > 32 ? 32 : (nc < 1 ? 1 : (int)nc); }
  return t;
}

typedef struct {
  dctz_pd_section sec[3];
  int threads, rc;
  size_t chunk;
} pd_args;
static void *pd_main(void *arg) {
  pd_args *a = (pd_args *)arg;
  a->rc = dctz_pdeflate_many(a->sec, 3, a->threads, a->chunk);
  return NULL;
}

typedef struct {
  const Bytef *src;
  uLong src_len, dst_len, produced;
  void *dst;
} inflate_job;
static void *inflate_main(void *arg) {
  inflate_job *j = (inflate_job *)arg;
  j->produced = inflate_into(j->src, j->src_len, j->dst, j->dst_len);
  return NULL;
}

/* ---- sections whose deflate blocks are independent chunks ("DZIX" trailer, dctz.h): inflated side by side ---- */
typedef struct {
  const unsigned char *src;   /* chunk bytes (raw deflate, byte aligned) */
  unsigned int zlen;
  unsigned char *dst;
  unsigned int len;           /* bytes the chunk must inflate to */
  uLong adler;
  int err;
} ix_chunk;
typedef struct {
  ix_chunk *chunks;
  size_t nchunks, next;
  pthread_mutex_t mu;
} ix_queue;
static void *ix_worker(void *arg) {
  ix_queue *q = (ix_queue *)arg;
  for (;;) {
    pthread_mutex_lock(&q->mu);
    size_t lo = q->next, hi = lo + 16 < q->nchunks ? lo + 16 : q->nchunks;   /* a few chunks per visit to the counter */
    q->next = hi;
    pthread_mutex_unlock(&q->mu);
    if (lo >= hi) break;
    for (size_t i = lo; i < hi; i++) {
      ix_chunk *c = &q->chunks[i];
      z_stream zs;
      memset(&zs, 0, sizeof(zs));
      if (inflateInit2(&zs, -15) != Z_OK) { c->err = 1; continue; }
      zs.next_in = (Bytef *)c->src; zs.avail_in = c->zlen;
      zs.next_out = c->dst; zs.avail_out = c->len;
      const int rc = inflate(&zs, Z_SYNC_FLUSH);
      if ((rc != Z_OK && rc != Z_BUF_ERROR) || zs.avail_in != 0 || zs.avail_out != 0) c->err = 1;
      inflateEnd(&zs);
      c->adler = adler32(adler32(0L, Z_NULL, 0), c->dst, c->len);
    }
  }
  return NULL;
}
/* The chunk index of a container ("DZIX", dctz.h) as three arrays of compressed sizes.  Returns 1 when the sections
 * carry the mark (78 5E), the trailer is there and describes them exactly -- chunk counts that follow from the raw sizes,
 * sizes that tile every stream up to its 03 00 + adler32 --, else 0 (nothing allocated): the caller then takes the
 * ordinary inflate. */
static int read_index(const unsigned char *const sec[3], const unsigned int zlen[3], const size_t raw[3], const unsigned char *trailer,
                      size_t *chunk_out, uint32_t *sizes[3]) {
  for (int i = 0; i < 3; i++) { sizes[i] = NULL; if (zlen[i] < 8 || sec[i][0] != 0x78 || sec[i][1] != 0x5E) return 0; }
  unsigned int hd[5];
  memcpy(hd, trailer, sizeof(hd));
  if (hd[0] != DCTZ_IX_MAGIC || hd[1] < 1024 || hd[1] > 65535) return 0;
  const size_t chunk = hd[1];
  for (int i = 0; i < 3; i++) if (hd[2 + i] != (raw[i] + chunk - 1) / chunk) return 0;
  const unsigned char *e = trailer + sizeof(hd);
  int ok = 1;
  for (int i = 0; i < 3 && ok; i++) {
    sizes[i] = (uint32_t *)malloc((hd[2 + i] ? hd[2 + i] : 1) * sizeof(uint32_t));
    if (!sizes[i]) { ok = 0; break; }
    size_t off = 2;
    for (size_t j = 0; j < hd[2 + i]; j++, e += 2) { unsigned short z; memcpy(&z, e, 2); sizes[i][j] = z; off += z; }
    if (off + 6 != zlen[i] || sec[i][off] != 0x03 || sec[i][off + 1] != 0x00) ok = 0;      /* the sizes must tile the stream */
  }
  if (!ok) { for (int i = 0; i < 3; i++) { free(sizes[i]); sizes[i] = NULL; } return 0; }
  *chunk_out = chunk;
  return 1;
}

/* The three sections of an indexed container inflated chunk by chunk on host threads (sizes / chunk from read_index).
 * Returns 1, or 0 when a chunk does not inflate or the content's adler32 is not the stream's: the caller then hands the
 * sections to the ordinary inflate, which treats damage the way the reference's reader does (dctz-decomp-lib.c:244-322
 * ignores inflate's return code). */
static int inflate_indexed(const unsigned char *const sec[3], const unsigned int zlen[3], unsigned char *const dst[3], const size_t raw[3],
                            size_t chunk, uint32_t *const sizes[3]) {
  size_t nch[3], total = 0;
  for (int i = 0; i < 3; i++) { nch[i] = (raw[i] + chunk - 1) / chunk; total += nch[i]; }
  ix_chunk *chunks = (ix_chunk *)calloc(total ? total : 1, sizeof(ix_chunk));
  if (!chunks) { fprintf(stderr, "Out of memory: chunk list\n"); exit(1); }
  size_t k = 0;
  for (int i = 0; i < 3; i++) {
    size_t off = 2;
    for (size_t j = 0; j < nch[i]; j++, k++) {
      chunks[k].src = sec[i] + off; chunks[k].zlen = sizes[i][j];
      chunks[k].dst = dst[i] + j * chunk;
      chunks[k].len = (unsigned int)(raw[i] - j * chunk < chunk ? raw[i] - j * chunk : chunk);
      off += sizes[i][j];
    }
  }
  int threads = zlib_threads();
  if (threads <= 0) { long nc = sysconf(_SC_NPROCESSORS_ONLN); threads = nc > 32 ? 32 : (nc < 1 ? 1 : (int)nc); }
  if ((size_t)threads > total) threads = total ? (int)total : 1;
  ix_queue q;
  q.chunks = chunks; q.nchunks = total; q.next = 0;
  pthread_mutex_init(&q.mu, NULL);
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
  int started = 0;
  if (th) for (int t = 0; t < threads - 1; t++) { if (pthread_create(&th[started], NULL, ix_worker, &q)) break; started++; }
  ix_worker(&q);
  for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
  free(th);
  pthread_mutex_destroy(&q.mu);
  int ok = 1;
  k = 0;
  for (int i = 0; i < 3 && ok; i++) {                   /* what inflate() checks at the end of a stream: the adler32 of the content */
    uLong a = adler32(0L, Z_NULL, 0);
    for (size_t j = 0; j < nch[i]; j++, k++) { if (chunks[k].err) ok = 0; a = adler32_combine(a, chunks[k].adler, (z_off_t)chunks[k].len); }
    const unsigned char *t = sec[i] + zlen[i] - 4;
    const uLong want = ((uLong)t[0] << 24) | ((uLong)t[1] << 16) | ((uLong)t[2] << 8) | (uLong)t[3];
    if (a != want) ok = 0;
  }
  free(chunks);
  if (!ok) fprintf(stderr, "libdctz: a chunk of an indexed section does not inflate; falling back to the one-stream inflate\n");
  return ok;
}

/* DCTZ_INFLATE_GPU=1: indexed sections are inflated on the device (one lane per chunk) instead of by host threads.
 * Off by default: a lane decodes its 16 KiB alone, which takes 18 - 30 ms per section however few chunks there are
 * (38 ms for the three sections of a 1 GiB shard side by side), where sixteen host threads need 19 ms; it pays on hosts with few
 * cores (the work is 0.5 core-seconds per GiB) and takes the raw streams off PCIe. */
static int inflate_gpu(void) { const char *e = getenv("DCTZ_INFLATE_GPU"); return e && atoi(e) != 0; }

static void dump_file(const char *name, const void *p, size_t bytes) {
  FILE *fp = fopen(name, "wb");
  if (!fp) return;
  fwrite(p, bytes, 1, fp);
  fclose(fp);
}

/* ------------------------------------------------- multi-dimensional blocks --- */
int dctz_set_block_dims(int ndims, const size_t *dims) {
  g_nd = 0;
  if (ndims <= 1) return 0;
  if (ndims > 3 || !dims) return -1;
  for (int i = 0; i < ndims; i++) if (dims[i] == 0 || dims[i] > 0x7FFFFFFFu) return -1;
  for (int i = 0; i < 3; i++) g_dims[i] = i < ndims ? dims[i] : 1;
  g_nd = ndims;
  return 0;
}

/* geometry of this call: the explicit request (one call only), else DCTZ_BLOCK_DIMS when its product is N */
static int take_block_dims(size_t n, size_t dims[3]) {
  int nd = g_nd;
  if (nd) { for (int i = 0; i < 3; i++) dims[i] = g_dims[i]; g_nd = 0; }
  else {
    const char *e = getenv("DCTZ_BLOCK_DIMS");
    if (!e || !*e) return 0;
    char *end = NULL;
    while (nd < 3) {
      const unsigned long long v = strtoull(e, &end, 10);
      if (end == e || v == 0 || v > 0x7FFFFFFFull) return 0;
      dims[nd++] = (size_t)v;
      if (*end != 'x' && *end != 'X') break;
      e = end + 1;
    }
    if (*end != 0 || nd < 2) return 0;
  }
  size_t prod = 1;
  for (int i = 0; i < nd; i++) prod *= dims[i];
  if (prod != n) {
    if (nd && !getenv("DCTZ_BLOCK_DIMS")) { fprintf(stderr, "libdctz: block dims do not multiply to N\n"); exit(1); }
    return 0;                                    /* the environment describes some other array of the run */
  }
  return nd;
}

/* ---- dctz_compress of a large array, pipelined (round 4) --------------------------------------------------------------
 * With the entropy stage on the device a 1 GiB call was H2D 19.4 ms -> kernels 0.3 -> deflate 2.6 -> D2H of the
 * compressed sections 4.5, one after the other, the host cores scaling the caller's array meanwhile: 27 ms, of which the
 * GPU worked for three.  The blocks of an array are independent once its scaling factor is known (util.c:29: a function of
 * max|x| alone), so:
 *   * the array crosses PCIe in GROUPS of DCTZ_PIPE_GROUP elements (default 8 Mi; dctzhip_h2d_pipe_*);
 *   * host threads take max|x| / min|x| of the caller's array while the first groups are under way (3 ms);
 *   * every group that has landed is compressed with the ARRAY's statistics (dctzhip_compress_part: same streams as the
 *     one call, AC_exact appended behind the groups in front), its bin_index and DC are deflated on the device -- the
 *     entropy stage codes 16 KiB chunks that reference nothing outside themselves, and a group is a whole number of chunks
 *     of both sections, so the groups' pieces concatenate to exactly the stream of the one call (adler32_combine for the
 *     check value) -- and the compressed bytes go back while the next groups are still arriving;
 *   * the in-place x /= sf of the caller's array (dctz-comp-lib.c:193-216) follows the copy: a block is divided once the
 *     group it lies in is on the device;
 *   * behind the last group only AC_exact (known in full only then) is left to deflate and bring back.
 * Same container as the serial path of this mode except for the header's `mean` (a tree-order sum either way; this path
 * adds the groups' sums).  DCTZ_PIPELINE=0: the serial path. */
static int pipeline_on(void) { const char *e = getenv("DCTZ_PIPELINE"); return e ? atoi(e) != 0 : 1; }
static size_t pipe_group(void) {
  const char *e = getenv("DCTZ_PIPE_GROUP");
  long long v = e ? atoll(e) : 0;
  return v >= (1 << 18) ? ((size_t)v & ~(size_t)((1 << 18) - 1)) : ((size_t)1 << 23);   /* a multiple of 256 Ki elements */
}
typedef struct { const void *x; size_t lo, hi; int is_d; double mx, mn; } mm_job;
/* util.c:18-25: max|x| and min|x|.  A NaN never wins a comparison there; vmaxpd / vminpd return their SECOND operand when
 * the comparison is unordered, so with the running value second a NaN element is passed over the same way. */
__attribute__((target("avx2"))) static void mm_range_avx2(mm_job *j) {
  if (j->is_d) {
    typedef double v4 __attribute__((vector_size(32), aligned(8)));
    typedef long long m4 __attribute__((vector_size(32)));
    const double *x = (const double *)j->x;
    const m4 absm = {0x7fffffffffffffffLL, 0x7fffffffffffffffLL, 0x7fffffffffffffffLL, 0x7fffffffffffffffLL};
    v4 mx[4], mn[4];
    for (int u = 0; u < 4; u++) { mx[u] = (v4){-1.0, -1.0, -1.0, -1.0}; mn[u] = (v4){INFINITY, INFINITY, INFINITY, INFINITY}; }
    size_t i = j->lo;
    for (; i + 16 <= j->hi; i += 16)
      for (int u = 0; u < 4; u++) {
        const v4 a = (v4)((m4)(*(const v4 *)(x + i + 4 * u)) & absm);
        mx[u] = __builtin_ia32_maxpd256(a, mx[u]);
        mn[u] = __builtin_ia32_minpd256(a, mn[u]);
      }
    double m = -1.0, l = INFINITY;
    for (int u = 0; u < 4; u++) for (int k = 0; k < 4; k++) { if (mx[u][k] > m) m = mx[u][k]; if (mn[u][k] < l) l = mn[u][k]; }
    for (; i < j->hi; i++) { const double a = fabs(x[i]); if (a > m) m = a; if (a < l) l = a; }
    j->mx = m; j->mn = l;
  } else {
    typedef float v8 __attribute__((vector_size(32), aligned(4)));
    typedef int m8 __attribute__((vector_size(32)));
    const float *x = (const float *)j->x;
    const m8 absm = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    v8 mx[4], mn[4];
    for (int u = 0; u < 4; u++) for (int k = 0; k < 8; k++) { mx[u][k] = -1.0f; mn[u][k] = INFINITY; }
    size_t i = j->lo;
    for (; i + 32 <= j->hi; i += 32)
      for (int u = 0; u < 4; u++) {
        const v8 a = (v8)((m8)(*(const v8 *)(x + i + 8 * u)) & absm);
        mx[u] = __builtin_ia32_maxps256(a, mx[u]);
        mn[u] = __builtin_ia32_minps256(a, mn[u]);
      }
    float m = -1.0f, l = INFINITY;
    for (int u = 0; u < 4; u++) for (int k = 0; k < 8; k++) { if (mx[u][k] > m) m = mx[u][k]; if (mn[u][k] < l) l = mn[u][k]; }
    for (; i < j->hi; i++) { const float a = fabsf(x[i]); if (a > m) m = a; if (a < l) l = a; }
    j->mx = (double)m; j->mn = (double)l;
  }
}
static void *mm_main(void *arg) {
  mm_job *j = (mm_job *)arg;
  if (__builtin_cpu_supports("avx2")) { mm_range_avx2(j); return NULL; }
  enum { U = 8 };                               /* independent chains: a single running maximum is one element per 4 cycles */
  if (j->is_d) {
    const double *x = (const double *)j->x;
    double mx[U], mn[U];
    for (int u = 0; u < U; u++) { mx[u] = -1.0; mn[u] = INFINITY; }
    size_t i = j->lo;
    for (; i + U <= j->hi; i += U)
      for (int u = 0; u < U; u++) { const double a = fabs(x[i + u]); if (a > mx[u]) mx[u] = a; if (a < mn[u]) mn[u] = a; }
    for (; i < j->hi; i++) { const double a = fabs(x[i]); if (a > mx[0]) mx[0] = a; if (a < mn[0]) mn[0] = a; }
    for (int u = 1; u < U; u++) { if (mx[u] > mx[0]) mx[0] = mx[u]; if (mn[u] < mn[0]) mn[0] = mn[u]; }
    j->mx = mx[0]; j->mn = mn[0];
  } else {
    const float *x = (const float *)j->x;
    float mx[U], mn[U];
    for (int u = 0; u < U; u++) { mx[u] = -1.0f; mn[u] = INFINITY; }
    size_t i = j->lo;
    for (; i + U <= j->hi; i += U)
      for (int u = 0; u < U; u++) { const float a = fabsf(x[i + u]); if (a > mx[u]) mx[u] = a; if (a < mn[u]) mn[u] = a; }
    for (; i < j->hi; i++) { const float a = fabsf(x[i]); if (a > mx[0]) mx[0] = a; if (a < mn[0]) mn[0] = a; }
    for (int u = 1; u < U; u++) { if (mx[u] > mx[0]) mx[0] = mx[u]; if (mn[u] < mn[0]) mn[0] = mn[u]; }
    j->mx = (double)mx[0]; j->mn = (double)mn[0];
  }
  return NULL;
}
/* x[lo, hi) /= sf, the reference's loop (IEEE division in the data type); four / eight lanes at a time where the CPU has
 * AVX2 (vdivpd / vdivps round like divsd / divss) */
__attribute__((target("avx2"))) static void scale_range_avx2(void *xv, size_t lo, size_t hi, int is_d, double sf) {
  if (is_d) {
    double *x = (double *)xv;
    typedef double v4 __attribute__((vector_size(32), aligned(8)));
    const v4 d = {sf, sf, sf, sf};
    size_t i = lo;
    for (; i + 4 <= hi; i += 4) { v4 v = *(v4 *)(x + i); v = v / d; *(v4 *)(x + i) = v; }
    for (; i < hi; i++) x[i] /= sf;
  } else {
    float *x = (float *)xv;
    const float f = (float)sf;
    typedef float v8 __attribute__((vector_size(32), aligned(4)));
    const v8 d = {f, f, f, f, f, f, f, f};
    size_t i = lo;
    for (; i + 8 <= hi; i += 8) { v8 v = *(v8 *)(x + i); v = v / d; *(v8 *)(x + i) = v; }
    for (; i < hi; i++) x[i] /= f;
  }
}
static void scale_range(void *xv, size_t lo, size_t hi, int is_d, double sf) {
  if (__builtin_cpu_supports("avx2")) { scale_range_avx2(xv, lo, hi, is_d, sf); return; }
  if (is_d) { double *x = (double *)xv; for (size_t i = lo; i < hi; i++) x[i] /= sf; }
  else { float *x = (float *)xv; const float f = (float)sf; for (size_t i = lo; i < hi; i++) x[i] /= f; }
}
#define FOLLOW_BLOCK ((size_t)1 << 17)        /* elements a worker of the follower divides at a time */
typedef struct {
  void *x; size_t n; int is_d; double sf;
  size_t landed;                /* elements of the caller's array the copy has finished with (released group by group) */
  size_t next;
  int failed;
} follow_job;
static void *follow_worker(void *arg) {
  follow_job *f = (follow_job *)arg;
  for (;;) {
    const size_t lo = __atomic_fetch_add(&f->next, FOLLOW_BLOCK, __ATOMIC_RELAXED);
    if (lo >= f->n) break;
    const size_t hi = lo + FOLLOW_BLOCK < f->n ? lo + FOLLOW_BLOCK : f->n;
    while (__atomic_load_n(&f->landed, __ATOMIC_ACQUIRE) < hi) {
      if (__atomic_load_n(&f->failed, __ATOMIC_ACQUIRE)) return NULL;
      usleep(20);
    }
    scale_range(f->x, lo, hi, f->is_d, f->sf);
  }
  return NULL;
}
typedef struct { dctzhip_ctx *c; follow_job *f; size_t gel, ts; } track_job;
static void *track_main(void *arg) {            /* releases the groups to the follower as their copies complete */
  track_job *t = (track_job *)arg;
  for (size_t e = 0; e < t->f->n;) {
    e = e + t->gel < t->f->n ? e + t->gel : t->f->n;
    if (dctzhip_h2d_pipe_landed(t->c, e * t->ts) != DCTZHIP_OK) { __atomic_store_n(&t->f->failed, 1, __ATOMIC_RELEASE); return NULL; }
    __atomic_store_n(&t->f->landed, e, __ATOMIC_RELEASE);
  }
  return NULL;
}
static void put_be32(unsigned char *p, uLong v) { p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v; }
static uLong get_be32(const unsigned char *p) { return ((uLong)p[0] << 24) | ((uLong)p[1] << 16) | ((uLong)p[2] << 8) | (uLong)p[3]; }
/* the drainer: compressed pieces of finished groups, device -> host, in order, beside the calling thread */
typedef struct {
  const unsigned char *src;     /* device: a piece's stream (78 5E | chunks | 03 00 | adler32) */
  size_t len, raw;              /* its length; the bytes it inflates to */
  int sec;
} drain_piece;
typedef struct {
  dctzhip_ctx *c;
  drain_piece *q;
  size_t posted, cap;           /* pieces posted so far (released by the calling thread) */
  size_t done;                  /* pieces brought back so far (released by the drainer) */
  int closed, failed;
  unsigned char *base[3];       /* where a section's stream is assembled on the host */
  size_t off[3];                /* its bytes so far, trailer excluded */
  uLong adler[3];
  int started[3];
} drain_job;
static void *drain_main(void *arg) {
  drain_job *d = (drain_job *)arg;
  for (size_t k = 0;; k++) {
    while (__atomic_load_n(&d->posted, __ATOMIC_ACQUIRE) <= k) {
      if (__atomic_load_n(&d->closed, __ATOMIC_ACQUIRE) && __atomic_load_n(&d->posted, __ATOMIC_ACQUIRE) <= k) return NULL;
      usleep(10);
    }
    const drain_piece *p = &d->q[k];
    /* the first piece of a section keeps its two header bytes; every piece's trailer is overwritten by the next piece's chunks */
    const size_t skip = d->started[p->sec] ? 2 : 0;
    unsigned char *at = d->base[p->sec] + d->off[p->sec];
    if (p->len < 8 || dctzhip_memcpy_d2h_side(d->c, at, p->src + skip, p->len - skip) != DCTZHIP_OK) { __atomic_store_n(&d->failed, 1, __ATOMIC_RELEASE); return NULL; }
    d->off[p->sec] += p->len - skip - 6;
    const uLong a = get_be32(d->base[p->sec] + d->off[p->sec] + 2);
    d->adler[p->sec] = d->started[p->sec] ? adler32_combine(d->adler[p->sec], a, (z_off_t)p->raw) : a;
    d->started[p->sec] = 1;
    __atomic_store_n(&d->done, k + 1, __ATOMIC_RELEASE);
  }
}
typedef struct { unsigned char *dst; const unsigned char *src; size_t n; } pcopy_job;
static void *pcopy_main(void *arg) { pcopy_job *j = (pcopy_job *)arg; memcpy(j->dst, j->src, j->n); return NULL; }
static void par_memcpy(void *dst, const void *src, size_t n, int T) {
  if (T > 16) T = 16;
  if (n < ((size_t)1 << 20) || T < 2) { memcpy(dst, src, n); return; }
  pcopy_job j[16];
  pthread_t th[16];
  int started = 0;
  const size_t per = (n / (size_t)T + 4095) & ~(size_t)4095;
  for (int t = 0; t < T; t++) {
    const size_t lo = per * (size_t)t;
    if (lo >= n) break;
    j[t].dst = (unsigned char *)dst + lo; j[t].src = (const unsigned char *)src + lo; j[t].n = n - lo < per ? n - lo : per;
    if (t && pthread_create(&th[started], NULL, pcopy_main, &j[t]) == 0) started++; else pcopy_main(&j[t]);
  }
  for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
}
/* returns 1: the container is written; 0: not applicable, the caller takes the serial path */
static int compress_pipelined(dctzhip_ctx *c, t_var *var, void *host_in, size_t n, int is_d, double error_bound, t_var *var_z, size_t *outSize,
                              double t_begin) {
  if (DCTZ_MODE != DCTZHIP_EC) return 0;          /* QT: the table is a property of the whole array (dctz-comp-lib.c:435-476) */
  const size_t ts = is_d ? sizeof(double) : sizeof(float);
  const int dtype = is_d ? DCTZHIP_F64 : DCTZHIP_F32;
  const size_t chunk = dctzhip_deflate_chunk_bytes();
  /* groups of 16 Mi elements unless DCTZ_PIPE_GROUP says otherwise: a group's entropy stage is a millisecond whatever the
   * group's size (four latency-bound kernels per section), a 128 MiB group takes 2.4 ms to arrive */
  const size_t gel = getenv("DCTZ_PIPE_GROUP") ? pipe_group() : ((size_t)1 << 24);
  if (gel % (16 * chunk) != 0 || n < 4 * gel) return 0;
  if (is_d ? isnan(((const double *)host_in)[0]) : isnan(((const float *)host_in)[0])) return 0;   /* util.c:18-19: max = NaN from the start */
  const size_t G = (n + gel - 1) / gel, nblk = CEIL(n, BLK_SZ);
  if (G > DCTZHIP_H2D_PIPE_MAX_GROUPS) return 0;         /* (a small DCTZ_PIPE_GROUP on a large array: more groups than a pipe holds -> the serial path) */
  const int dbg = getenv("DCTZ_PIPE_DEBUG") != NULL;
  /* the first element of every group, as the caller gave it (the follower divides the array while the groups' sums come in) */
  double *firsts = (double *)malloc(G * sizeof(double));
  if (!firsts) { fprintf(stderr, "Out of memory\n"); exit(1); }
  for (size_t g = 0; g < G; g++) firsts[g] = is_d ? ((const double *)host_in)[g * gel] : (double)((const float *)host_in)[g * gel];

  grow(&g_dev.in, &g_dev.in_cap, n * ts);
  grow(&g_dev.bin, &g_dev.bin_cap, n);
  grow(&g_dev.dc, &g_dev.dc_cap, nblk * sizeof(float));
  grow(&g_dev.ac, &g_dev.ac_cap, n * sizeof(float));
  /* every group's pieces have slots of their own in the device buffers of the compressed sections (the drainer is still
   * reading a group's while the next group's are written); AC_exact: what a group can add, plus the chunk carried over */
  const size_t bound[3] = {(dctzhip_deflate_bound(gel) + 255) & ~(size_t)255, (dctzhip_deflate_bound(gel / 16) + 255) & ~(size_t)255,
                           (dctzhip_deflate_bound(gel * sizeof(float) + chunk) + 255) & ~(size_t)255};
  for (int i = 0; i < 3; i++) grow(&g_dev.z[i], &g_dev.z_cap[i], G * bound[i]);
  const size_t bpg = gel / chunk, dpg = gel / 16 / chunk;       /* chunks of bin_index / DC per group */
  size_t ix_n[3] = {(n + chunk - 1) / chunk, (nblk * sizeof(float) + chunk - 1) / chunk, 0};
  uint32_t *ix[3] = {(uint32_t *)malloc((ix_n[0] + 1) * sizeof(uint32_t)), (uint32_t *)malloc((ix_n[1] + 1) * sizeof(uint32_t)),
                     (uint32_t *)malloc((n * sizeof(float) / chunk + G + 2) * sizeof(uint32_t))};
  /* DC's and AC_exact's compressed pieces wait on the host until bin_index's length is known */
  unsigned char *dcz = (unsigned char *)host_buf(1, G * bound[1]), *acz = (unsigned char *)host_buf(2, G * bound[2]);
  drain_piece *pieces = (drain_piece *)malloc((3 * G + 3) * sizeof(drain_piece));
  if (!ix[0] || !ix[1] || !ix[2] || !pieces) { fprintf(stderr, "Out of memory: chunk index\n"); exit(1); }
  unsigned char *zc = (is_d ? (unsigned char *)var_z->buf.d : (unsigned char *)var_z->buf.f) + sizeof(struct header);

  const double t0 = now_s();
  if (dctzhip_h2d_pipe_begin(c, g_dev.in, host_in, n * ts, gel * ts) != DCTZHIP_OK) die("H2D pipe");
  /* max|x|, min|x| of the whole array (util.c:18-25) */
  int T = host_threads();
  if (T > 64) T = 64;
  /* threads of the max|x| pass / of the follower: the call is bound by the HOST's memory traffic (the copy's staging, the
   * max|x| pass and the division together move six times the array: 250 GB/s on the measured host), more threads only take
   * bandwidth from the copy (tools/cpipe_sweep.py) */
  int TM = T < 16 ? T : 16, TF = T < 12 ? T : 12;
  { const char *e = getenv("DCTZ_PIPE_MM_THREADS"); if (e && atoi(e) > 0) TM = atoi(e) > 64 ? 64 : atoi(e); }
  { const char *e = getenv("DCTZ_PIPE_FOLLOW_THREADS"); if (e && atoi(e) > 0) TF = atoi(e) > 64 ? 64 : atoi(e); }
  mm_job mj[64];
  pthread_t mth[64];
  int mstarted = 0;
  for (int t = 0; t < TM; t++) {
    mj[t].x = host_in; mj[t].is_d = is_d; mj[t].lo = n / (size_t)TM * (size_t)t; mj[t].hi = t + 1 == TM ? n : n / (size_t)TM * (size_t)(t + 1);
    mj[t].mx = -1.0; mj[t].mn = INFINITY;
    if (t + 1 < TM && pthread_create(&mth[mstarted], NULL, mm_main, &mj[t]) == 0) mstarted++; else mm_main(&mj[t]);
  }
  for (int t = 0; t < mstarted; t++) pthread_join(mth[t], NULL);
  double mx = -1.0, mn = INFINITY;
  for (int t = 0; t < TM; t++) { if (mj[t].mx > mx) mx = mj[t].mx; if (mj[t].mn < mn) mn = mj[t].mn; }
  if (!(mx >= 0.0)) { mx = 0.0; mn = 0.0; }             /* (nothing but NaNs behind a finite first element: as the serial loop, max = |x[0]|) */
  { const double a0 = fabs(firsts[0]); if (a0 > mx) mx = a0; if (a0 < mn) mn = a0; }
  const double t_mm = now_s();

  drain_job dj;
  memset(&dj, 0, sizeof(dj));
  dj.c = c; dj.q = pieces; dj.cap = 3 * G + 3;
  dj.base[0] = zc; dj.base[1] = dcz; dj.base[2] = acz;
  pthread_t dth;
  if (pthread_create(&dth, NULL, drain_main, &dj)) { fprintf(stderr, "Error creating thread\n"); exit(0); }
  size_t npieces = 0, S = 0, ac_done = 0;                 /* pieces posted; exact coefficients so far; bytes of AC_exact deflated so far */
  double sum = 0.0, sf = 1.0, t_gpu = 0.0, t_tail = 0.0;
  follow_job fj = {host_in, n, is_d, 1.0, 0, 0, 0};
  track_job tj = {c, &fj, gel, ts};
  pthread_t fth[64], tth;
  int fstarted = 0, tracking = 0;
  for (size_t g = 0; g < G; g++) {
    const size_t e0 = g * gel, ne = n - e0 < gel ? n - e0 : gel, b0 = e0 / BLK_SZ, nb = CEIL(ne, BLK_SZ);
    const double tg0 = now_s();
    if (dctzhip_h2d_pipe_wait(c, (e0 + ne) * ts) != DCTZHIP_OK) die("H2D pipe");
    uint32_t cnt_g = 0;
    double pst[3];
    if (dctzhip_compress_part(c, (const char *)g_dev.in + e0 * ts, ne, dtype, error_bound, mx, mn, (unsigned char *)g_dev.bin + e0,
                              (float *)g_dev.dc + b0, (float *)g_dev.ac + S, &cnt_g, pst, &sf) != DCTZHIP_OK) die("dctzhip_compress_part");
    sum += pst[2] + (g ? firsts[g] : 0.0);
    S += cnt_g;
    const double tg1 = now_s();
    if (g == 0 && sf != 1.0 && scale_on_host(1)) {        /* the in-place x /= sf follows the copy from here on */
      fj.sf = sf;
      if (pthread_create(&tth, NULL, track_main, &tj) == 0) tracking = 1;
      for (int t = 0; tracking && t < TF; t++) if (pthread_create(&fth[fstarted], NULL, follow_worker, &fj) == 0) fstarted++;
    }
    /* the group's bin_index and DC, and the chunks of AC_exact that are complete by now (all of the rest behind the last group) */
    size_t ac_upto = S * sizeof(float);
    if (g + 1 < G) ac_upto -= ac_upto % chunk;
    const int nsec = ac_upto > ac_done ? 3 : 2;
    const void *gsrc[3] = {(const unsigned char *)g_dev.bin + e0, (const float *)g_dev.dc + b0, (const unsigned char *)g_dev.ac + ac_done};
    const size_t gn[3] = {ne, nb * sizeof(float), ac_upto - ac_done};
    void *gdst[3] = {(unsigned char *)g_dev.z[0] + g * bound[0], (unsigned char *)g_dev.z[1] + g * bound[1], (unsigned char *)g_dev.z[2] + g * bound[2]};
    uint32_t *gix[3] = {ix[0] + g * bpg, ix[1] + g * dpg, ix[2] + ac_done / chunk};
    const unsigned gflags[3] = {0u, DCTZHIP_DEFLATE_LITERALS, DCTZHIP_DEFLATE_LITERALS};   /* DC and AC_exact are bytes of floats */
    size_t glen[3] = {0, 0, 0};
    if (dctzhip_deflate_ex(c, nsec, gsrc, gn, gdst, bound, glen, gix, gflags) != DCTZHIP_OK) die("dctzhip_deflate");
    const size_t first_new = npieces;
    for (int i = 0; i < nsec; i++) {
      pieces[npieces].src = (const unsigned char *)gdst[i]; pieces[npieces].len = glen[i]; pieces[npieces].raw = gn[i]; pieces[npieces].sec = i;
      npieces++;
    }
    if (g + 1 == G && S == 0) {                            /* no exact coefficient at all: the empty stream, as the one call writes it */
      const size_t zero = 0;
      const void *asrc[1] = {g_dev.ac};
      void *adst[1] = {g_dev.z[2]};
      uint32_t *aix[1] = {ix[2]};
      size_t alen = 0;
      if (dctzhip_deflate_ex(c, 1, asrc, &zero, adst, &bound[2], &alen, aix, NULL) != DCTZHIP_OK) die("dctzhip_deflate");
      pieces[npieces].src = (const unsigned char *)g_dev.z[2]; pieces[npieces].len = alen; pieces[npieces].raw = 0; pieces[npieces].sec = 2;
      npieces++;
    }
    if (g + 1 == G) {
      /* The lengths of all three streams are known now -- what the drainer has brought back plus these last pieces --, so
       * DC's and AC_exact's last pieces go straight to their final places behind bin_index, and what is staged of them is
       * moved there by this thread while the drainer is busy with the last group. */
      while (__atomic_load_n(&dj.done, __ATOMIC_ACQUIRE) < first_new) { if (__atomic_load_n(&dj.failed, __ATOMIC_ACQUIRE)) die("D2H compressed pieces"); usleep(10); }
      size_t fin[3] = {dj.off[0], dj.off[1], dj.off[2]};
      int st[3] = {dj.started[0], dj.started[1], dj.started[2]};
      for (size_t k = first_new; k < npieces; k++) {
        if (pieces[k].len < 8) die("deflate: short stream");
        fin[pieces[k].sec] += pieces[k].len - (st[pieces[k].sec] ? 2 : 0) - 6;
        st[pieces[k].sec] = 1;
      }
      const size_t staged1 = dj.off[1], staged2 = dj.off[2];
      dj.base[1] = zc + fin[0] + 6;
      dj.base[2] = zc + fin[0] + 6 + fin[1] + 6;
      __atomic_store_n(&dj.posted, npieces, __ATOMIC_RELEASE);
      par_memcpy(dj.base[1], dcz, staged1, 4);
      par_memcpy(dj.base[2], acz, staged2, 8);
    } else
      __atomic_store_n(&dj.posted, npieces, __ATOMIC_RELEASE);
    ac_done = ac_upto;
    const double tg2 = now_s();
    t_gpu += tg1 - tg0; t_tail += tg2 - tg1;
    if (dbg) fprintf(stderr, "[cpipe] group %zu: start %.2f ms, wait + kernels %.2f, deflate %.2f (cnt %u), landed %zu\n", g, (tg0 - t0) * 1e3, (tg1 - tg0) * 1e3, (tg2 - tg1) * 1e3, cnt_g,
                     __atomic_load_n(&fj.landed, __ATOMIC_RELAXED) / gel);
  }
  ix_n[2] = (S * sizeof(float) + chunk - 1) / chunk;
  const double t_loop = now_s();
  __atomic_store_n(&dj.closed, 1, __ATOMIC_RELEASE);
  pthread_join(dth, NULL);
  if (dj.failed) die("D2H compressed pieces");
  if (dctzhip_h2d_pipe_end(c, 0) != DCTZHIP_OK) die("H2D pipe");
  const double t_drain = now_s();
  /* close the three streams (03 00 | adler32 of the whole section) and put DC's and AC_exact's behind bin_index's */
  size_t zlen[3];
  for (int i = 0; i < 3; i++) {
    unsigned char *t = dj.base[i] + dj.off[i];
    t[0] = 0x03; t[1] = 0x00; put_be32(t + 2, dj.adler[i]);
    zlen[i] = dj.off[i] + 6;
  }
  const size_t z0 = zlen[0], z1 = zlen[1], z2 = zlen[2];
  if (z0 > 0xffffffffu || z1 > 0xffffffffu || z2 > 0xffffffffu) { fprintf(stderr, "libdctz: a compressed section exceeds the header's 32-bit sizes (dctz.h:104-113): shard the array\n"); exit(1); }
  if (dj.base[1] != zc + z0 || dj.base[2] != zc + z0 + z1) die("pipelined compress: section lengths do not add up");
  const double t_asm = now_s();
  for (int t = 0; t < fstarted; t++) pthread_join(fth[t], NULL);
  if (tracking) pthread_join(tth, NULL);
  if (fj.failed) die("H2D pipe (follower)");
  if (sf != 1.0 && !fstarted) {                            /* (no follower thread could be started, or DCTZ_SCALE_HOST=0: the device's copy) */
    if (dctzhip_scale_inplace(c, g_dev.in, n, dtype, sf) != DCTZHIP_OK) die("scale");
    if (dctzhip_memcpy_d2h(c, host_in, g_dev.in, n * ts) != DCTZHIP_OK) die("D2H scaled input");
  }

  /* container: header | bin_indexz | DCz | AC_exactz | "DZIX"  (:775-820) */
  struct header h;
  memset(&h, 0, sizeof(h));
  h.datatype = var->datatype;
  h.num_elements = (unsigned int)n;
  h.error_bound = error_bound;
  h.tot_AC_exact_count = (unsigned int)S;
  if (is_d) { h.scaling_factor.d = sf; h.mean.d = sum / (double)(int)n; }           /* util.c:28 / :41, on a tree-order sum (DCTZ_FAST_MEAN) */
  else { h.scaling_factor.f = (float)sf; h.mean.f = (float)sum / (float)(int)n; }
  h.bindex_sz_compressed = (unsigned int)z0;
  h.DC_sz_compressed = (unsigned int)z1;
  h.AC_exact_sz_compressed = (unsigned int)z2;
  memcpy(zc - sizeof(h), &h, sizeof(h));
  const size_t ix_bytes = (20 + 2 * (ix_n[0] + ix_n[1] + ix_n[2]) + 3) & ~(size_t)3;
  unsigned char *cur = zc + z0 + z1 + z2;
  const unsigned int hd[5] = {DCTZ_IX_MAGIC, (unsigned int)chunk, (unsigned int)ix_n[0], (unsigned int)ix_n[1], (unsigned int)ix_n[2]};
  memset(cur, 0, ix_bytes);
  memcpy(cur, hd, sizeof(hd));
  unsigned short *e = (unsigned short *)(cur + sizeof(hd));
  for (int i = 0; i < 3; i++) {
    for (size_t k = 0; k < ix_n[i]; k++) *e++ = (unsigned short)ix[i][k];
    free(ix[i]);
  }
  free(firsts); free(pieces);
  *outSize = sizeof(struct header) + z0 + z1 + z2 + ix_bytes;
  const double t_end = now_s();
  /* the stages overlap: h2d_s is the span up to the last group's kernels, gpu_s / zlib_s what the calling thread spent in the
   * groups' kernels / entropy stage, d2h_s what was left behind the last group (drain, assembly, the follower) */
  g_times.h2d_s = t_loop - t0; g_times.gpu_s = t_gpu; g_times.zlib_s = t_tail; g_times.d2h_s = t_end - t_loop;
  g_times.total_s = t_end - t_begin;
  if (dbg) fprintf(stderr, "[cpipe] max|x| at %.2f ms, last group queued %.2f, pieces drained %.2f, assembled %.2f, follower joined %.2f\n", (t_mm - t0) * 1e3, (t_loop - t0) * 1e3,
                   (t_drain - t0) * 1e3, (t_asm - t0) * 1e3, (t_end - t0) * 1e3);
  if (!quiet()) printf("outSize = %zu\n", *outSize); /* :841-843 */
  return 1;
}

/* -------------------------------------------------------------- compress --- */
int dctz_compress(t_var *var, int N, size_t *outSize, t_var *var_z, double error_bound) {
  const double t_begin = now_s();
  const int is_d = (var->datatype == DOUBLE);
  const size_t ts = is_d ? sizeof(double) : sizeof(float);
  const int dtype = is_d ? DCTZHIP_F64 : DCTZHIP_F32;
  void *host_in = is_d ? (void *)var->buf.d : (void *)var->buf.f;

  if (error_bound < 1E-6) { /* dctz-comp-lib.c:135-138 */
    printf("ERROR BOUND is not acceptable");
    exit(1);
  }
  if (N <= 0) { fprintf(stderr, "libdctz: N must be positive\n"); exit(1); }
  const size_t n = (size_t)N;
  size_t dims[3] = {0, 0, 0};
  const int nd = take_block_dims(n, dims);      /* 0: the reference's flat blocks */
  size_t nblk = CEIL(n, BLK_SZ);
  if (nd) {
    nblk = dctzhip_nd_blocks(nd, dims);
    if (!nblk) { fprintf(stderr, "libdctz: array too large for multi-dimensional blocks\n"); exit(1); }
  }
  const size_t npos = nd ? nblk * BLK_SZ : n;   /* positions the streams cover (edge tiles are padded) */

  dctzhip_ctx *c = ctx();
  if (!nd && zlib_gpu() && fast_mean() && pipeline_on() && !getenv("DCTZ_DUMP_STREAMS") &&
      compress_pipelined(c, var, host_in, n, is_d, error_bound, var_z, outSize, t_begin)) return 1;
  grow(&g_dev.in, &g_dev.in_cap, n * ts);
  grow(&g_dev.bin, &g_dev.bin_cap, npos);
  grow(&g_dev.dc, &g_dev.dc_cap, nblk * sizeof(float));
  grow(&g_dev.ac, &g_dev.ac_cap, npos * sizeof(float));

  double t0 = now_s();
  if (dctzhip_memcpy_h2d(c, g_dev.in, host_in, n * ts) != DCTZHIP_OK) die("H2D");
  double t1 = now_s();

  /* a2..a9 on the GPU; the scaled array is produced in place on the device and
   * copied back over the caller's buffer (the reference's in-place "/= sf") */
  dctzhip_cinfo info;
  const int gpu_tail = zlib_gpu();
  const int fast_tail = gpu_tail || zlib_threads() > 3;
  pthread_t mean_thread;
  host_mean_job mj = {host_in, n, is_d, 0.0, 0};
  int mean_on_host = 0;
  const int tree_mean = fast_mean();
  if (!tree_mean && fast_tail && pthread_create(&mean_thread, NULL, host_mean_main, &mj) == 0) mean_on_host = 1;
  if (!tree_mean && !mean_on_host && dctzhip_serial_mean_begin(c, g_dev.in, n, dtype) != DCTZHIP_OK) die("serial mean");
  int rc = nd ? dctzhip_compress_nd(c, g_dev.in, nd, dims, dtype, error_bound, DCTZ_MODE, g_dev.bin, (float *)g_dev.dc,
                                    (float *)g_dev.ac, NULL, &info)
              : dctzhip_compress(c, g_dev.in, n, dtype, error_bound, DCTZ_MODE, g_dev.bin, (float *)g_dev.dc,
                                 (float *)g_dev.ac, NULL, NULL, &info);
  if (rc != DCTZHIP_OK) die("dctzhip_compress");
  if (dctzhip_sync(c) != DCTZHIP_OK) die("sync");   /* (the call returns while its last kernels drain: keep the stage timers honest) */
  double t2 = now_s();

  /* raw streams on the host: what the host zlib tails read, and what the dump taps write */
  const int want_raw = !gpu_tail || getenv("DCTZ_DUMP_STREAMS") != NULL;
  t_bin_id *bin_index = NULL;
  float *DC = NULL, *AC_exact = NULL;
  if (want_raw) {
    bin_index = (t_bin_id *)host_buf(0, npos);
    DC = (float *)host_buf(1, nblk * sizeof(float));
    AC_exact = (float *)host_buf(2, (size_t)info.cnt * sizeof(float));
    if (dctzhip_memcpy_d2h(c, bin_index, g_dev.bin, npos) != DCTZHIP_OK) die("D2H bin_index");
    if (dctzhip_memcpy_d2h(c, DC, g_dev.dc, nblk * sizeof(float)) != DCTZHIP_OK) die("D2H DC");
    if (info.cnt && dctzhip_memcpy_d2h(c, AC_exact, g_dev.ac, (size_t)info.cnt * sizeof(float)) != DCTZHIP_OK)
      die("D2H AC_exact");
  }
  double t3 = now_s();

  if (getenv("DCTZ_DUMP_STREAMS")) { /* dctz-comp-lib.c:583-595, :443-448 */
    dump_file("bin_index.bin", bin_index, npos);
    dump_file("AC_exact.bin", AC_exact, (size_t)info.cnt * sizeof(float));
#ifdef USE_QTABLE
    if (is_d) dump_file("qtable.bin", info.qtable_raw, BLK_SZ * sizeof(double));
    else { float q[BLK_SZ]; for (int j = 0; j < BLK_SZ; j++) q[j] = (float)info.qtable_raw[j]; dump_file("qtable.bin", q, sizeof(q)); }
#endif
  }

  /* zlib tail: three streams on three threads (dctz-comp-lib.c:620-732) */
  const int zthreads = zlib_threads();
  const size_t sec_bytes[3] = {npos * sizeof(t_bin_id), nblk * sizeof(float), (size_t)info.cnt * sizeof(float)};
  const void *sec_src[3] = {bin_index, DC, AC_exact};
  pthread_attr_t attr;
  pthread_attr_init(&attr);
  pthread_attr_setdetachstate(&attr, PTHREAD_CREATE_JOINABLE);
  zjob jb[3];
  memset(jb, 0, sizeof(jb));
  pthread_t pd_thread;
  pd_args pda;
  size_t pd_len[3] = {0, 0, 0};
  size_t gz_len[3] = {0, 0, 0};
  double t_gz = 0.0;                        /* end of the device entropy stage (the write-back of x/sf follows it) */
  uint32_t *ix[3] = {NULL, NULL, NULL};     /* compressed bytes per chunk, for the "DZIX" trailer */
  size_t ix_n[3] = {0, 0, 0};
  const int host_scale = scale_on_host(gpu_tail) && info.sf != 1.0;
  pthread_t scale_thread;
  scale_mgr sm = {host_in, n, is_d, host_threads(), info.sf, mean_on_host ? &mean_thread : NULL, mean_on_host ? &mj : NULL, 0};
  int scale_started = 0;
  if (host_scale && pthread_create(&scale_thread, NULL, scale_mgr_main, &sm) == 0) scale_started = 1;
  if (gpu_tail) {                           /* deflate on the device, compressed bytes only over PCIe */
    const void *gsrc[3] = {g_dev.bin, g_dev.dc, g_dev.ac};
    size_t gcap[3];
    for (int i = 0; i < 3; i++) {
      gcap[i] = dctzhip_deflate_bound(sec_bytes[i]);
      grow(&g_dev.z[i], &g_dev.z_cap[i], gcap[i]);
      ix_n[i] = (sec_bytes[i] + dctzhip_deflate_chunk_bytes() - 1) / dctzhip_deflate_chunk_bytes();
      ix[i] = (uint32_t *)malloc((ix_n[i] ? ix_n[i] : 1) * sizeof(uint32_t));
      if (!ix[i]) { fprintf(stderr, "Out of memory: chunk index\n"); exit(1); }
    }
    const unsigned gflags[3] = {0u, DCTZHIP_DEFLATE_LITERALS, DCTZHIP_DEFLATE_LITERALS};   /* DC and AC_exact are bytes of floats */
    if (dctzhip_deflate_ex(c, 3, gsrc, sec_bytes, (void *const *)g_dev.z, gcap, gz_len, (uint32_t *const *)ix, gflags) != DCTZHIP_OK) die("dctzhip_deflate");
    /* the sections go straight from the device into the caller's container, behind the header (:775-820) */
    unsigned char *zc = (is_d ? (unsigned char *)var_z->buf.d : (unsigned char *)var_z->buf.f) + sizeof(struct header);
    for (int i = 0; i < 3; i++) {
      if (dctzhip_memcpy_d2h(c, zc, g_dev.z[i], gz_len[i]) != DCTZHIP_OK) die("D2H compressed section");
      zc += gz_len[i];
    }
    t_gz = now_s();
  } else if (zthreads > 3) {                /* chunked deflate, one pool for all three sections */
    for (int i = 0; i < 3; i++) {
      jb[i].bound = (uLong)dctz_pdeflate_bound(sec_bytes[i], zlib_chunk());
      jb[i].dst = (Bytef *)malloc(jb[i].bound);
      if (!jb[i].dst) { fprintf(stderr, "Out of memory: zlib buffer\n"); exit(1); }
      pda.sec[i].src = sec_src[i]; pda.sec[i].n = sec_bytes[i];
      pda.sec[i].dst = jb[i].dst; pda.sec[i].cap = jb[i].bound; pda.sec[i].out_len = &pd_len[i];
    }
    pda.threads = zthreads; pda.chunk = zlib_chunk(); pda.rc = 0;
    { const char *e = getenv("DCTZ_ZLIB_LEVEL"); dctz_pdeflate_set_level(e ? atoi(e) : -1); }
    if (pthread_create(&pd_thread, &attr, pd_main, &pda)) { fprintf(stderr, "Error creating thread\n"); exit(0); }
  } else {
    for (int i = 0; i < 3; i++) zjob_start(&jb[i], sec_src[i], sec_bytes[i], &attr);
  }

  /* while zlib runs: write x/sf back over the caller's buffer (:193-216) and
   * fetch the serial-order mean for the header */
  double mean_serial = 0.0;
  if (scale_started) pthread_join(scale_thread, NULL);                            /* (it has joined the mean thread itself) */
  else if (mean_on_host) pthread_join(mean_thread, NULL);                         /* before host_in is overwritten */
  if (tree_mean) mean_serial = info.mean;
  else if (mean_on_host) mean_serial = mj.mean;
  else if (dctzhip_serial_mean_end(c, &mean_serial) != DCTZHIP_OK) die("serial mean");
  if (info.sf != 1.0 && !scale_started) {   /* only now may the device copy of the input change */
    if (dctzhip_scale_inplace(c, g_dev.in, n, dtype, info.sf) != DCTZHIP_OK) die("scale");
    if (dctzhip_memcpy_d2h(c, host_in, g_dev.in, n * ts) != DCTZHIP_OK) die("D2H scaled input");
  }

  uLong zsz[3];
  if (gpu_tail) {
    for (int i = 0; i < 3; i++) zsz[i] = (uLong)gz_len[i];
  } else if (zthreads > 3) {
    pthread_join(pd_thread, NULL);
    if (pda.rc) { fprintf(stderr, "libdctz: parallel deflate failed (%d)\n", pda.rc); exit(1); }
    for (int i = 0; i < 3; i++) zsz[i] = (uLong)pd_len[i];
  } else {
    for (int i = 0; i < 3; i++) zsz[i] = zjob_join(&jb[i]);
  }
  for (int i = 0; i < 3; i++)
    if (zsz[i] > 0xffffffffu) { fprintf(stderr, "libdctz: compressed section %d is %lu bytes; the header holds 32-bit sizes (dctz.h:104-113): shard the array\n", i, zsz[i]); exit(1); }
  pthread_attr_destroy(&attr);
  double t4 = now_s();

  /* container: header | bin_indexz | DCz | AC_exactz | [qtable]  (:775-820) */
  struct header h;
  memset(&h, 0, sizeof(h));
  h.datatype = (t_datatype)((unsigned)var->datatype | ((unsigned)nd << DCTZ_GEOM_SHIFT));
  h.num_elements = (unsigned int)N;
  h.error_bound = error_bound;
  h.tot_AC_exact_count = info.cnt;
  if (is_d) { h.scaling_factor.d = info.sf; h.mean.d = mean_serial; }
  else { h.scaling_factor.f = (float)info.sf; h.mean.f = (float)mean_serial; }
  h.bindex_sz_compressed = (unsigned int)zsz[0];
  h.DC_sz_compressed = (unsigned int)zsz[1];
  h.AC_exact_sz_compressed = (unsigned int)zsz[2];
#ifdef USE_QTABLE
  h.bindex_count = (unsigned int)npos;
#endif
  *outSize = sizeof(struct header) + zsz[0] + zsz[1] + zsz[2];
#ifdef USE_QTABLE
  *outSize += BLK_SZ * ts;
#endif
  if (nd) *outSize += 16;                      /* "DZND" + the extents */
  const size_t ix_bytes = gpu_tail ? ((20 + 2 * (ix_n[0] + ix_n[1] + ix_n[2]) + 3) & ~(size_t)3) : 0;
  *outSize += ix_bytes;                        /* "DZIX" chunk index (dctz.h) */
  unsigned char *cur = is_d ? (unsigned char *)var_z->buf.d : (unsigned char *)var_z->buf.f;
  memcpy(cur, &h, sizeof(h)); cur += sizeof(h);
  if (gpu_tail) cur += zsz[0] + zsz[1] + zsz[2];         /* already in place */
  else {
    memcpy(cur, jb[0].dst, zsz[0]); cur += zsz[0];
    memcpy(cur, jb[1].dst, zsz[1]); cur += zsz[1];
    memcpy(cur, jb[2].dst, zsz[2]); cur += zsz[2];
  }
#ifdef USE_QTABLE
  if (is_d) memcpy(cur, info.qtable, BLK_SZ * sizeof(double));
  else { float q[BLK_SZ]; for (int j = 0; j < BLK_SZ; j++) q[j] = (float)info.qtable[j]; memcpy(cur, q, sizeof(q)); }
  cur += BLK_SZ * ts;
#endif
  if (nd) {
    const unsigned int tr[4] = {DCTZ_ND_MAGIC, (unsigned int)dims[0], (unsigned int)dims[1], (unsigned int)(nd == 3 ? dims[2] : 0)};
    memcpy(cur, tr, sizeof(tr));
    cur += sizeof(tr);
  }
  if (gpu_tail) {
    const unsigned int hd[5] = {DCTZ_IX_MAGIC, (unsigned int)dctzhip_deflate_chunk_bytes(), (unsigned int)ix_n[0], (unsigned int)ix_n[1], (unsigned int)ix_n[2]};
    memset(cur, 0, ix_bytes);
    memcpy(cur, hd, sizeof(hd));
    unsigned short *e = (unsigned short *)(cur + sizeof(hd));
    for (int i = 0; i < 3; i++) {
      for (size_t k = 0; k < ix_n[i]; k++) *e++ = (unsigned short)ix[i][k];
      free(ix[i]);
    }
  }
  for (int i = 0; i < 3; i++) free(jb[i].dst);

  g_times.h2d_s = t1 - t0; g_times.gpu_s = t2 - t1; g_times.d2h_s = t3 - t2; g_times.zlib_s = t4 - t3;
  if (gpu_tail) { g_times.zlib_s = t_gz - t3; g_times.d2h_s += t4 - t_gz; }   /* nothing overlaps the write-back of x/sf here: count it as the copy it is */
  g_times.total_s = now_s() - t_begin;
  if (!quiet()) printf("outSize = %zu\n", *outSize); /* :841-843 */
  return 1;
}

/* ---- lists of arrays in host memory (round 4; ADDITIONS to the reference's API, include/dctz.h) ---------------------------
 * The reference's own workloads are lists of small arrays, one dctz_compress() call -- one process -- per array
 * (tests/test-dctz.sh:13-56 over tests/list-msst19.txt:1-6).  Through this drop-in one such call is an H2D copy, one kernel,
 * a D2H copy and three zlib threads around a few microseconds of GPU work; k of them in a loop are k times that.
 * dctz_compress_batch() takes the k arrays at once: ONE staged H2D copy, ONE batch launch (dctzhip_compress_batch: every
 * array with its own statistics, scaling factor, bin ranges and tot_AC_exact_count, as in its own call), ONE copy back, and
 * the 3 k single-shot deflates of the reference's tail (dctz-comp-lib.c:620-732, same parameters) dealt to a pool of host
 * threads together with the k in-place scalings (:193-216) and serial-order means (util.c:18-28).  Every container is byte
 * for byte the one dctz_compress() writes for that array (reference tail), every caller's array ends up divided by its sf. */
typedef struct {
  int kind;                     /* 0: deflate one section, 1: scale + mean of one array, 2: inflate one section */
  int arr, sec;
  const void *src; size_t n;    /* deflate / inflate input */
  Bytef *dst; uLong cap, out;   /* ... output */
  void *x; size_t nx; int is_d; double sf, mean;
  int rc;
} bjob;
typedef struct { bjob *jobs; size_t njobs, next; } bpool;
static void bjob_run(bjob *j) {
  if (j->kind == 0) {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    /* dctz-comp-lib.c:642-643: default level, 32K window, memLevel 8, default strategy; one deflate(Z_FINISH) (:75-88) */
    if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15, DEF_MEM_LEVEL, Z_DEFAULT_STRATEGY) != Z_OK) { j->rc = 1; return; }
    zs.data_type = Z_UNKNOWN;
    zs.next_in = (Bytef *)j->src; zs.avail_in = (uInt)j->n;
    zs.next_out = j->dst; zs.avail_out = (uInt)j->cap;
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) j->rc = 1;
    j->out = zs.total_out;
    deflateEnd(&zs);
  } else if (j->kind == 2) {
    j->out = inflate_into((const Bytef *)j->src, (uLong)j->n, j->dst, (size_t)j->cap);
  } else {
    /* the header's mean in the reference's order (util.c:18-28 / :31-41), then the in-place x /= sf (:193-216) */
    if (j->is_d) {
      double *x = (double *)j->x, sum = 0.0;
      for (size_t i = 1; i < j->nx; i++) sum += x[i];
      j->mean = sum / (double)(int)j->nx;
      if (j->sf != 1.0) for (size_t i = 0; i < j->nx; i++) x[i] /= j->sf;
    } else {
      float *x = (float *)j->x, sum = 0.0f;
      const float sf = (float)j->sf;
      for (size_t i = 1; i < j->nx; i++) sum += x[i];
      j->mean = (double)(sum / (float)(int)j->nx);
      if (sf != 1.0f) for (size_t i = 0; i < j->nx; i++) x[i] /= sf;
    }
  }
}
static void *bpool_worker(void *arg) {
  bpool *p = (bpool *)arg;
  for (;;) {
    const size_t i = __atomic_fetch_add(&p->next, 1, __ATOMIC_RELAXED);
    if (i >= p->njobs) break;
    bjob_run(&p->jobs[i]);
  }
  return NULL;
}
static void bpool_run(bjob *jobs, size_t njobs) {
  bpool p = {jobs, njobs, 0};
  int T = host_threads();
  if ((size_t)T > njobs) T = (int)njobs;
  pthread_t th[64];
  if (T > 64) T = 64;
  int started = 0;
  for (int t = 1; t < T; t++) { if (pthread_create(&th[started], NULL, bpool_worker, &p) == 0) started++; }
  bpool_worker(&p);
  for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
}
static size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
static struct { void *pin; size_t pin_cap; void *dev; size_t dev_cap; } g_batch;
static void batch_buffers(dctzhip_ctx *c, size_t host_bytes, size_t dev_bytes) {
  if (host_bytes > g_batch.pin_cap) {
    if (g_batch.pin) dctzhip_host_unregister(c, g_batch.pin), free(g_batch.pin);
    g_batch.pin = NULL; g_batch.pin_cap = 0;
    if (posix_memalign(&g_batch.pin, 4096, host_bytes)) { fprintf(stderr, "Out of memory: batch staging\n"); exit(1); }
    memset(g_batch.pin, 0, host_bytes);
    (void)dctzhip_host_register(c, g_batch.pin, host_bytes);      /* (pinned: one fast copy each way; unpinned it still works) */
    g_batch.pin_cap = host_bytes;
  }
  grow(&g_batch.dev, &g_batch.dev_cap, dev_bytes);
}

int dctz_compress_batch(int k, t_var *const *vars, const int *N, size_t *outSizes, t_var *const *vars_z, const double *error_bounds) {
  if (k <= 0) return 1;
  dctzhip_ctx *c = ctx();
  /* layout of the staging area (host, pinned) and of its device twin: inputs | bin_index | DC | AC_exact, array after array */
  size_t *off_in = (size_t *)malloc(4 * (size_t)k * sizeof(size_t));
  dctzhip_batch_citem *items = (dctzhip_batch_citem *)calloc((size_t)k, sizeof(*items));
  dctzhip_cinfo *infos = (dctzhip_cinfo *)calloc((size_t)k, sizeof(*infos));
  if (!off_in || !items || !infos) { fprintf(stderr, "Out of memory: batch\n"); exit(1); }
  size_t *off_bin = off_in + k, *off_dc = off_in + 2 * k, *off_ac = off_in + 3 * k;
  size_t in_bytes = 0, total = 0;
  for (int i = 0; i < k; i++) {
    if (error_bounds[i] < 1E-6) { printf("ERROR BOUND is not acceptable"); exit(1); }     /* dctz-comp-lib.c:135-138 */
    if (N[i] <= 0) { fprintf(stderr, "libdctz: N must be positive\n"); exit(1); }
    const size_t ts = vars[i]->datatype == DOUBLE ? 8 : 4;
    off_in[i] = in_bytes; in_bytes += up256((size_t)N[i] * ts);
  }
  total = in_bytes;
  for (int i = 0; i < k; i++) { off_bin[i] = total; total += up256((size_t)N[i]); }
  for (int i = 0; i < k; i++) { off_dc[i] = total; total += up256(CEIL((size_t)N[i], BLK_SZ) * sizeof(float)); }
  for (int i = 0; i < k; i++) { off_ac[i] = total; total += up256((size_t)N[i] * sizeof(float)); }
  batch_buffers(c, total, total);
  unsigned char *hp = (unsigned char *)g_batch.pin, *dp = (unsigned char *)g_batch.dev;
  for (int i = 0; i < k; i++) {
    const int is_d = vars[i]->datatype == DOUBLE;
    const size_t ts = is_d ? 8 : 4;
    memcpy(hp + off_in[i], is_d ? (void *)vars[i]->buf.d : (void *)vars[i]->buf.f, (size_t)N[i] * ts);
    items[i].d_in = dp + off_in[i]; items[i].n = (size_t)N[i]; items[i].dtype = is_d ? DCTZHIP_F64 : DCTZHIP_F32;
    items[i].error_bound = error_bounds[i];
    items[i].d_bin_index = dp + off_bin[i]; items[i].d_dc = (float *)(dp + off_dc[i]); items[i].d_ac_exact = (float *)(dp + off_ac[i]);
    items[i].d_scaled = NULL;
  }
  if (dctzhip_memcpy_h2d(c, dp, hp, in_bytes) != DCTZHIP_OK) die("H2D (batch)");
  if (dctzhip_compress_batch(c, k, items, DCTZ_MODE, infos) != DCTZHIP_OK) die("dctzhip_compress_batch");
  if (dctzhip_memcpy_d2h(c, hp + in_bytes, dp + in_bytes, total - in_bytes) != DCTZHIP_OK) die("D2H (batch)");   /* (waits for the kernels first) */
  /* the tails: 3 k deflates + k (mean, scaling) jobs on the pool, the longest first */
  bjob *jobs = (bjob *)calloc(4 * (size_t)k, sizeof(bjob));
  if (!jobs) { fprintf(stderr, "Out of memory: batch\n"); exit(1); }
  size_t nj = 0;
  for (int pass = 0; pass < 4; pass++)                /* AC_exact sections first (bytes of floats: the slow ones), then bin_index, DC, scalings */
    for (int i = 0; i < k; i++) {
      const size_t n = (size_t)N[i], nblk = CEIL(n, BLK_SZ);
      bjob *j = &jobs[nj];
      j->arr = i;
      if (pass == 3) {
        const int is_d = vars[i]->datatype == DOUBLE;
        j->kind = 1; j->x = is_d ? (void *)vars[i]->buf.d : (void *)vars[i]->buf.f; j->nx = n; j->is_d = is_d; j->sf = infos[i].sf;
      } else {
        const int sec = pass == 0 ? 2 : pass - 1;
        const size_t bytes = sec == 0 ? n : (sec == 1 ? nblk * sizeof(float) : (size_t)infos[i].cnt * sizeof(float));
        j->kind = 0; j->sec = sec; j->n = bytes;
        j->src = hp + (sec == 0 ? off_bin[i] : (sec == 1 ? off_dc[i] : off_ac[i]));
        j->cap = compressBound((uLong)bytes);
        j->dst = (Bytef *)malloc(j->cap ? j->cap : 1);
        if (!j->dst) { fprintf(stderr, "Out of memory: zlib buffer\n"); exit(1); }
      }
      nj++;
    }
  if (getenv("DCTZ_DUMP_STREAMS")) fprintf(stderr, "libdctz: DCTZ_DUMP_STREAMS is a tap of single calls; dctz_compress_batch writes no dump files\n");
  bpool_run(jobs, nj);
  /* containers: header | bin_indexz | DCz | AC_exactz | [qtable]  (:775-820) */
  for (int i = 0; i < k; i++) {
    const int is_d = vars[i]->datatype == DOUBLE;
    const size_t ts = is_d ? 8 : 4;
    const bjob *jz[3] = {NULL, NULL, NULL}, *js = NULL;
    for (size_t q = 0; q < nj; q++) if (jobs[q].arr == i) { if (jobs[q].kind == 1) js = &jobs[q]; else jz[jobs[q].sec] = &jobs[q]; }
    for (int s3 = 0; s3 < 3; s3++) if (jz[s3]->rc) { fprintf(stderr, "libdctz: deflate failed\n"); exit(1); }
    struct header h;
    memset(&h, 0, sizeof(h));
    h.datatype = vars[i]->datatype;
    h.num_elements = (unsigned int)N[i];
    h.error_bound = error_bounds[i];
    h.tot_AC_exact_count = infos[i].cnt;
    if (is_d) { h.scaling_factor.d = infos[i].sf; h.mean.d = js->mean; }
    else { h.scaling_factor.f = (float)infos[i].sf; h.mean.f = (float)js->mean; }
    h.bindex_sz_compressed = (unsigned int)jz[0]->out;
    h.DC_sz_compressed = (unsigned int)jz[1]->out;
    h.AC_exact_sz_compressed = (unsigned int)jz[2]->out;
#ifdef USE_QTABLE
    h.bindex_count = (unsigned int)N[i];
#endif
    unsigned char *cur = is_d ? (unsigned char *)vars_z[i]->buf.d : (unsigned char *)vars_z[i]->buf.f;
    size_t out = sizeof(h) + jz[0]->out + jz[1]->out + jz[2]->out;
    memcpy(cur, &h, sizeof(h)); cur += sizeof(h);
    for (int s3 = 0; s3 < 3; s3++) { memcpy(cur, jz[s3]->dst, jz[s3]->out); cur += jz[s3]->out; }
#ifdef USE_QTABLE
    if (is_d) memcpy(cur, infos[i].qtable, BLK_SZ * sizeof(double));
    else { float qf[BLK_SZ]; for (int j = 0; j < BLK_SZ; j++) qf[j] = (float)infos[i].qtable[j]; memcpy(cur, qf, sizeof(qf)); }
    out += BLK_SZ * ts;
#endif
    (void)ts;
    outSizes[i] = out;
    if (!quiet()) printf("outSize = %zu\n", out); /* :841-843 */
  }
  for (size_t q = 0; q < nj; q++) free(jobs[q].dst);
  free(jobs); free(off_in); free(items); free(infos);
  return 1;
}

int dctz_decompress_batch(int k, t_var *const *vars_z, t_var *const *vars_r) {
  if (k <= 0) return 1;
  dctzhip_ctx *c = ctx();
  struct header *hs = (struct header *)malloc((size_t)k * sizeof(struct header));
  size_t *off_bin = (size_t *)malloc(4 * (size_t)k * sizeof(size_t));
  dctzhip_batch_ditem *items = (dctzhip_batch_ditem *)calloc((size_t)k, sizeof(*items));
  bjob *jobs = (bjob *)calloc(3 * (size_t)k, sizeof(bjob));
  double *qtabs = (double *)malloc((size_t)k * BLK_SZ * sizeof(double));
  if (!hs || !off_bin || !items || !jobs || !qtabs) { fprintf(stderr, "Out of memory: batch\n"); exit(1); }
  size_t *off_dc = off_bin + k, *off_ac = off_bin + 2 * k, *off_out = off_bin + 3 * k;
  size_t total = 0, in_bytes;
  for (int i = 0; i < k; i++) {
    const unsigned char *cur = vars_z[i]->datatype == DOUBLE ? (const unsigned char *)vars_z[i]->buf.d : (const unsigned char *)vars_z[i]->buf.f;
    memcpy(&hs[i], cur, sizeof(struct header));                       /* dctz-decomp-lib.c:84-94 */
    if (hs[i].num_elements == 0) { fprintf(stderr, "libdctz: empty stream\n"); exit(1); }
    if (DCTZ_GEOM_OF(hs[i].datatype)) { fprintf(stderr, "libdctz: dctz_decompress_batch takes flat containers\n"); exit(1); }
  }
  for (int i = 0; i < k; i++) { off_bin[i] = total; total += up256(hs[i].num_elements); }
  for (int i = 0; i < k; i++) { off_dc[i] = total; total += up256(CEIL((size_t)hs[i].num_elements, BLK_SZ) * sizeof(float)); }
  for (int i = 0; i < k; i++) { off_ac[i] = total; total += up256(((size_t)hs[i].tot_AC_exact_count + 4) * sizeof(float)); }
  in_bytes = total;
  for (int i = 0; i < k; i++) { off_out[i] = total; total += up256((size_t)hs[i].num_elements * (vars_z[i]->datatype == DOUBLE ? 8 : 4)); }
  batch_buffers(c, total, total);
  unsigned char *hp = (unsigned char *)g_batch.pin, *dp = (unsigned char *)g_batch.dev;
  size_t nj = 0;
  for (int i = 0; i < k; i++) {
    const int is_d = vars_z[i]->datatype == DOUBLE;
    const size_t ts = is_d ? 8 : 4, n = hs[i].num_elements, nblk = CEIL(n, BLK_SZ);
    const unsigned char *cur = (is_d ? (const unsigned char *)vars_z[i]->buf.d : (const unsigned char *)vars_z[i]->buf.f) + sizeof(struct header);
    const unsigned int zl[3] = {hs[i].bindex_sz_compressed, hs[i].DC_sz_compressed, hs[i].AC_exact_sz_compressed};
    const size_t raw[3] = {n, nblk * sizeof(float), (size_t)hs[i].tot_AC_exact_count * sizeof(float)};
    const size_t offs[3] = {off_bin[i], off_dc[i], off_ac[i]};
    for (int s3 = 0; s3 < 3; s3++) {                                   /* three inflates per array (dctz-decomp-lib.c:244-322) */
      bjob *j = &jobs[nj++];
      j->kind = 2; j->arr = i; j->sec = s3; j->src = cur; j->n = zl[s3]; j->dst = hp + offs[s3]; j->cap = (uLong)raw[s3];
      cur += zl[s3];
    }
    const void *qt = NULL;
#ifdef USE_QTABLE
    memcpy(qtabs + (size_t)i * BLK_SZ, cur, BLK_SZ * ts);              /* :193-199 */
    qt = qtabs + (size_t)i * BLK_SZ;
#endif
    (void)ts;
    items[i].d_bin_index = dp + off_bin[i]; items[i].d_dc = (const float *)(dp + off_dc[i]); items[i].d_ac_exact = (const float *)(dp + off_ac[i]);
    items[i].ac_count = hs[i].tot_AC_exact_count; items[i].qtable_host = qt;
    items[i].n = n; items[i].dtype = is_d ? DCTZHIP_F64 : DCTZHIP_F32; items[i].error_bound = hs[i].error_bound;
    items[i].sf = is_d ? hs[i].scaling_factor.d : (double)hs[i].scaling_factor.f;
    items[i].d_out = dp + off_out[i];
  }
  bpool_run(jobs, nj);
  if (!quiet()) for (int i = 0; i < k; i++) printf("uncompressed bin_index size is: %lu\n", (unsigned long)jobs[3 * (size_t)i].out); /* :260-262 */
  if (dctzhip_memcpy_h2d(c, dp, hp, in_bytes) != DCTZHIP_OK) die("H2D (batch)");
  if (dctzhip_decompress_batch(c, k, items, DCTZ_MODE, NULL) != DCTZHIP_OK) die("dctzhip_decompress_batch");
  if (dctzhip_memcpy_d2h(c, hp + in_bytes, dp + in_bytes, total - in_bytes) != DCTZHIP_OK) die("D2H (batch)");
  for (int i = 0; i < k; i++) {
    const int is_d = vars_z[i]->datatype == DOUBLE;
    memcpy(is_d ? (void *)vars_r[i]->buf.d : (void *)vars_r[i]->buf.f, hp + off_out[i], (size_t)hs[i].num_elements * (is_d ? 8 : 4));
  }
  free(hs); free(off_bin); free(items); free(jobs); free(qtabs);
  return 1;
}

/* ------------------------------------------------------- container check --- */
/* dctz_decompress() trusts the header the way the reference does (dctz-decomp-lib.c:84-100:
 * no size is checked against the buffer, a corrupt file reads out of bounds).  A caller that
 * knows how many bytes it holds runs this first: it checks the layout of dctz-comp-lib.c:775-820
 * against `zbytes`, the plausibility of the header fields, and -- deep != 0 -- that the three
 * sections really inflate to N, 4*nblk and 4*cnt bytes (no GPU involved). */
int dctz_check_container(const void *z, size_t zbytes, int max_elements, int deep) {
  struct header h;
  if (!z || zbytes < sizeof(h)) return DCTZ_CHECK_TRUNCATED;
  memcpy(&h, z, sizeof(h));
  const t_datatype base = DCTZ_TYPE_OF(h.datatype);
  const unsigned geom = DCTZ_GEOM_OF(h.datatype);
  if ((base != FLOAT && base != DOUBLE) || ((unsigned)h.datatype >> 16) || (geom != 0 && geom != 2 && geom != 3)) return DCTZ_CHECK_BAD_HEADER;
  if (h.num_elements == 0 || h.num_elements > 0x7FFFFFFFu) return DCTZ_CHECK_BAD_HEADER;     /* N is an int, dctz.h:126 */
  if (max_elements > 0 && h.num_elements > (unsigned int)max_elements) return DCTZ_CHECK_TOO_LARGE;
  if (!(h.error_bound >= 1E-6) || h.error_bound != h.error_bound) return DCTZ_CHECK_BAD_HEADER;   /* :135-138 */
  const size_t n = h.num_elements, ts = base == DOUBLE ? sizeof(double) : sizeof(float);
  const size_t body = (size_t)h.bindex_sz_compressed + h.DC_sz_compressed + h.AC_exact_sz_compressed;
  size_t want = sizeof(h) + body;
#ifdef USE_QTABLE
  want += BLK_SZ * ts;
#else
  (void)ts;
#endif
  size_t nblk = CEIL(n, BLK_SZ), npos = n;
  if (geom) {                                           /* multi-dimensional blocks: the extents follow the last section */
    if (zbytes < want + 16) return DCTZ_CHECK_TRUNCATED;
    unsigned int tr[4];
    memcpy(tr, (const unsigned char *)z + want, sizeof(tr));
    want += 16;
    size_t dims[3] = {tr[1], tr[2], tr[3]};
    if (tr[0] != DCTZ_ND_MAGIC || (geom == 2 && tr[3] != 0)) return DCTZ_CHECK_BAD_HEADER;
    nblk = dctzhip_nd_blocks((int)geom, dims);
    size_t prod = 1;
    for (unsigned i = 0; i < geom; i++) prod *= dims[i];
    if (!nblk || prod != n) return DCTZ_CHECK_BAD_HEADER;
    npos = nblk * BLK_SZ;
  }
  if ((size_t)h.tot_AC_exact_count > npos - nblk) return DCTZ_CHECK_BAD_HEADER;                /* at most 63 per block */
#ifdef USE_QTABLE
  if (h.bindex_count != npos) return DCTZ_CHECK_BAD_HEADER;                                     /* :798 */
#endif
  if (zbytes < want) return DCTZ_CHECK_TRUNCATED;
  const unsigned char *cur = (const unsigned char *)z + sizeof(h);
  const size_t raw[3] = {npos, nblk * sizeof(float), (size_t)h.tot_AC_exact_count * sizeof(float)};
  const unsigned int zs[3] = {h.bindex_sz_compressed, h.DC_sz_compressed, h.AC_exact_sz_compressed};
  {
    /* Sections that carry the mark of the GPU entropy stage (78 5E): dctz_decompress will look for the "DZIX" chunk index
     * BEHIND the container -- it has no size to check that against, this function has.  The index must be there in full
     * and describe the sections exactly (the same test as the reader's), or the container is refused: a foreign stream
     * that merely starts 78 5E (zlib at levels 2 .. 5 writes those bytes too) would make the reader look past the buffer. */
    const unsigned char *sp[3] = {cur, cur + zs[0], cur + (size_t)zs[0] + zs[1]};
    int marked = 1;
    for (int i = 0; i < 3; i++) if (zs[i] < 8 || sp[i][0] != 0x78 || sp[i][1] != 0x5E) marked = 0;
    if (marked) {
      if (zbytes < want + 20) return DCTZ_CHECK_TRUNCATED;
      unsigned int hd[5];
      memcpy(hd, (const unsigned char *)z + want, sizeof(hd));
      if (hd[0] != DCTZ_IX_MAGIC || hd[1] < 1024 || hd[1] > 65535) return DCTZ_CHECK_BAD_STREAM;
      size_t entries = 0;
      for (int i = 0; i < 3; i++) {
        if (hd[2 + i] != (raw[i] + hd[1] - 1) / hd[1]) return DCTZ_CHECK_BAD_STREAM;
        entries += hd[2 + i];
      }
      if (zbytes < want + 20 + 2 * entries) return DCTZ_CHECK_TRUNCATED;
      uint32_t *sizes[3];
      size_t chunk = 0;
      if (!read_index(sp, zs, raw, (const unsigned char *)z + want, &chunk, sizes)) return DCTZ_CHECK_BAD_STREAM;
      for (int i = 0; i < 3; i++) free(sizes[i]);
    }
  }
  if (!deep) return DCTZ_CHECK_OK;
  for (int i = 0; i < 3; i++) {
    /* inflate into a small window, counting: the section must end exactly at `raw[i]` bytes */
    z_stream st;
    unsigned char win[65536];
    memset(&st, 0, sizeof(st));
    if (inflateInit(&st) != Z_OK) return DCTZ_CHECK_BAD_STREAM;
    st.next_in = (Bytef *)cur; st.avail_in = zs[i];
    size_t produced = 0;
    int rc;
    do {
      st.next_out = win; st.avail_out = sizeof(win);
      rc = inflate(&st, Z_NO_FLUSH);
      produced += sizeof(win) - st.avail_out;
    } while (rc == Z_OK && produced <= raw[i]);
    inflateEnd(&st);
    if (rc != Z_STREAM_END || produced != raw[i]) return DCTZ_CHECK_BAD_STREAM;
    cur += zs[i];
  }
  return DCTZ_CHECK_OK;
}

/* ------------------------------------------------------------ decompress --- */
/* ---- dctz_decompress, pipelined (round 4) ------------------------------------------------------------------------------
 * A container whose sections are independent chunks ("DZIX") is rebuilt GROUP by group of DCTZ_PIPE_GROUP elements
 * (default 8 Mi): while the host threads inflate the chunks of the groups ahead, the streams of the groups that are
 * complete go to the device, their blocks are rebuilt there (dctzhip_decompress on the group's slices: a group is a
 * range of whole blocks, and its first "stored exactly" coefficient is the number of flags in front of it -- the running
 * `pos` of dctz-decomp-lib.c:402-412 -- which the inflating threads count on the way), and a copier thread brings finished
 * groups back into the caller's array.  Round 3 ran inflate (17 ms per GiB), H2D (3), kernels (0.3) and D2H (21) one after
 * the other; the stages now overlap and the call takes about as long as its longest one.  The reconstruction is the same
 * bytes: the same kernels on the same inputs.  DCTZ_PIPELINE=0: the serial path. */
typedef struct {
  const unsigned char *src;
  unsigned int zlen, len;
  unsigned char *dst;
  uLong adler;
  unsigned int n255;          /* bytes == 255 in the chunk (bin_index chunks: flags + block heads) */
  int count255;
  volatile int done;          /* 1: inflated, 2: does not inflate */
} pp_chunk;
typedef struct {
  pp_chunk *chunks;
  size_t nchunks;
  size_t next;
} pp_queue;
static void pp_do(pp_chunk *c) {
  z_stream zs;
  int bad = 0;
  memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, -15) != Z_OK) bad = 1;
  else {
    zs.next_in = (Bytef *)c->src; zs.avail_in = c->zlen;
    zs.next_out = c->dst; zs.avail_out = c->len;
    const int rc = inflate(&zs, Z_SYNC_FLUSH);
    if ((rc != Z_OK && rc != Z_BUF_ERROR) || zs.avail_in != 0 || zs.avail_out != 0) bad = 1;
    inflateEnd(&zs);
  }
  if (!bad) {
    c->adler = adler32(adler32(0L, Z_NULL, 0), c->dst, c->len);
    if (c->count255) {
      unsigned int k = 0;
      for (unsigned int i = 0; i < c->len; i++) k += (c->dst[i] == 255);
      c->n255 = k;
    }
  }
  __atomic_store_n(&c->done, bad ? 2 : 1, __ATOMIC_RELEASE);
}
static int pp_step(pp_queue *q) {                       /* one chunk, if any is left; 0: the list is exhausted */
  const size_t i = __atomic_fetch_add(&q->next, 1, __ATOMIC_RELAXED);
  if (i >= q->nchunks) return 0;
  pp_do(&q->chunks[i]);
  return 1;
}
static void *pp_worker(void *arg) {
  while (pp_step((pp_queue *)arg)) {}
  return NULL;
}
/* returns 1: done; 0: not applicable / something does not inflate -- the caller takes the serial path (and its way of
 * reporting damage) */
static int decompress_pipelined(dctzhip_ctx *c, const struct header *h, const unsigned char *const sec[3], const unsigned int zl[3],
                                size_t chunk, uint32_t *const sizes[3], const void *qtable, t_var *var_r, size_t *got_out) {
  const int is_d = ((h->datatype & 0xff) == DOUBLE);
  const size_t ts = is_d ? sizeof(double) : sizeof(float);
  const int dtype = is_d ? DCTZHIP_F64 : DCTZHIP_F32;
  const size_t n = h->num_elements, nblk = CEIL(n, BLK_SZ), cnt = h->tot_AC_exact_count;
  const size_t gel = pipe_group();
  if (gel % (16 * chunk) != 0 || n < 2 * gel) return 0;
  const size_t G = (n + gel - 1) / gel;
  if (G > DCTZHIP_D2H_PIPE_MAX_MARKS) return 0;           /* (more groups than the D2H pipe has marks -> the serial path) */
  const size_t raw[3] = {n, nblk * sizeof(float), cnt * sizeof(float)};
  size_t nch[3], total = 0;
  for (int i = 0; i < 3; i++) { nch[i] = (raw[i] + chunk - 1) / chunk; total += nch[i]; }
  unsigned char *const dst[3] = {(unsigned char *)host_buf(0, raw[0]), (unsigned char *)host_buf(1, raw[1]), (unsigned char *)host_buf(2, raw[2])};
  pp_chunk *chunks = (pp_chunk *)calloc(total ? total : 1, sizeof(pp_chunk));
  size_t *first[3];                                     /* position of a section's chunk j in the (priority-ordered) list */
  for (int i = 0; i < 3; i++) first[i] = (size_t *)malloc((nch[i] ? nch[i] : 1) * sizeof(size_t));
  size_t *goff = (size_t *)malloc(G * sizeof(size_t)), *gbytes = (size_t *)malloc(G * sizeof(size_t));
  if (!chunks || !first[0] || !first[1] || !first[2] || !goff || !gbytes) { fprintf(stderr, "Out of memory: chunk list\n"); exit(1); }
  /* the list in the order the groups need it: group g's bin_index and DC chunks, and the g-th share of AC_exact's */
  size_t zoff[3] = {2, 2, 2}, nextc[3] = {0, 0, 0}, k = 0;
  const size_t bpg = gel / chunk, dpg = gel / 16 / chunk;
  for (size_t g = 0; g < G; g++) {
    const size_t upto[3] = {(g + 1) * bpg < nch[0] ? (g + 1) * bpg : nch[0], (g + 1) * dpg < nch[1] ? (g + 1) * dpg : nch[1],
                            g + 1 == G ? nch[2] : nch[2] * (g + 1) / G};
    for (int i = 0; i < 3; i++)
      for (; nextc[i] < upto[i]; nextc[i]++, k++) {
        const size_t j = nextc[i];
        chunks[k].src = sec[i] + zoff[i]; chunks[k].zlen = sizes[i][j];
        chunks[k].dst = dst[i] + j * chunk;
        chunks[k].len = (unsigned int)(raw[i] - j * chunk < chunk ? raw[i] - j * chunk : chunk);
        chunks[k].count255 = (i == 0);
        zoff[i] += sizes[i][j];
        first[i][j] = k;
      }
  }
  pp_queue q = {chunks, total, 0};
  int threads = host_threads();
  if ((size_t)threads > total) threads = total ? (int)total : 1;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
  int started = 0;
  if (th) for (int t = 0; t < threads; t++) { if (pthread_create(&th[started], NULL, pp_worker, &q)) break; started++; }

  grow(&g_dev.out, &g_dev.out_cap, n * ts);
  unsigned char *host_out = is_d ? (unsigned char *)var_r->buf.d : (unsigned char *)var_r->buf.f;
  if (dctzhip_d2h_pipe_begin(c, host_out, g_dev.out, n * ts) != DCTZHIP_OK) die("D2H pipe");
  const double sf = is_d ? h->scaling_factor.d : (double)h->scaling_factor.f;
  int ok = 1;
  size_t S = 0, ac_up = 0;                               /* exact coefficients consumed so far; bytes of AC_exact on the device */
  const int dbg = getenv("DCTZ_PIPE_DEBUG") != NULL;
  const double tp0 = now_s();
  for (size_t g = 0; g < G && ok; g++) {
    const double tg0 = now_s();
    const size_t e0 = g * gel, ne = n - e0 < gel ? n - e0 : gel;
    const size_t b0 = e0 / BLK_SZ, nb = CEIL(ne, BLK_SZ);
    /* wait for the group's bin_index and DC chunks (helping with whatever chunk is next meanwhile) */
    size_t flags = 0;
    const size_t c_lo[2] = {e0 / chunk, b0 * sizeof(float) / chunk};
    const size_t c_hi[2] = {(e0 + ne + chunk - 1) / chunk, ((b0 + nb) * sizeof(float) + chunk - 1) / chunk};
    for (int i = 0; i < 2 && ok; i++)
      for (size_t j = c_lo[i]; j < c_hi[i] && ok; j++) {
        pp_chunk *pc = &chunks[first[i][j]];
        int d;
        while ((d = __atomic_load_n(&pc->done, __ATOMIC_ACQUIRE)) == 0) { if (!pp_step(&q)) sched_yield(); }
        if (d != 1) ok = 0;
        if (i == 0) flags += pc->n255;
      }
    if (!ok) break;
    /* every block's first byte is the 255 of dctz-comp-lib.c:361, not a flag -- in a container this library or the reference
     * wrote.  One whose block heads are something else (adler-valid all the same) would make the count wrap and the groups'
     * places in AC_exact go backwards while the kernels, which never look at byte 0, rebuild something else than the serial
     * path does: such a container takes the serial path. */
    if (flags < nb) { ok = 0; break; }
    flags -= nb;
    size_t S1 = S + flags;
    if (S1 > cnt) S1 = cnt;                                /* (a stream that flags more than it brings: the kernels report it) */
    for (size_t j = ac_up / chunk; j < (S1 * sizeof(float) + chunk - 1) / chunk && ok; j++) {
      pp_chunk *pc = &chunks[first[2][j]];
      int d;
      while ((d = __atomic_load_n(&pc->done, __ATOMIC_ACQUIRE)) == 0) { if (!pp_step(&q)) sched_yield(); }
      if (d != 1) ok = 0;
    }
    if (!ok) break;
    const double tg1 = now_s();
    /* the group's streams -> device */
    if (dctzhip_memcpy_h2d(c, (unsigned char *)g_dev.bin + e0, dst[0] + e0, ne) != DCTZHIP_OK) die("H2D bin_index");
    if (dctzhip_memcpy_h2d(c, (unsigned char *)g_dev.dc + b0 * sizeof(float), dst[1] + b0 * sizeof(float), nb * sizeof(float)) != DCTZHIP_OK) die("H2D DC");
    const size_t ac_to = S1 * sizeof(float);
    if (ac_to > ac_up) {
      if (dctzhip_memcpy_h2d(c, (unsigned char *)g_dev.ac + ac_up, dst[2] + ac_up, ac_to - ac_up) != DCTZHIP_OK) die("H2D AC_exact");
      ac_up = ac_to;
    }
    const double tg2 = now_s();
    if (dctzhip_decompress(c, (unsigned char *)g_dev.bin + e0, (const float *)g_dev.dc + b0, (const float *)g_dev.ac + S, (uint32_t)(cnt - S), qtable, ne, dtype,
                           h->error_bound, sf, DCTZ_MODE, (unsigned char *)g_dev.out + e0 * ts) != DCTZHIP_OK) die("dctzhip_decompress");
    if (dbg) fprintf(stderr, "[pipe] group %zu: start %.2f ms, waited %.2f for its chunks, H2D %.2f, call %.2f\n", g, (tg0 - tp0) * 1e3, (tg1 - tg0) * 1e3, (tg2 - tg1) * 1e3, (now_s() - tg2) * 1e3);
    S = S1;
    if (dctzhip_d2h_pipe_advance(c, (e0 + ne) * ts) != DCTZHIP_OK) die("D2H pipe");     /* the group's bytes: complete behind its kernels */
  }
  while (pp_step(&q)) {}                                   /* (what is left of the list: chunks no group waited for) */
  for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
  if (dbg) fprintf(stderr, "[pipe] all groups queued at %.2f ms\n", (now_s() - tp0) * 1e3);
  if (dctzhip_d2h_pipe_end(c, !ok) != DCTZHIP_OK) die("D2H output");
  if (dbg) fprintf(stderr, "[pipe] copies done at %.2f ms\n", (now_s() - tp0) * 1e3);
  /* what inflate() checks at the end of a stream: the adler32 of the content */
  for (int i = 0; i < 3 && ok; i++) {
    uLong a = adler32(0L, Z_NULL, 0);
    for (size_t j = 0; j < nch[i]; j++) { const pp_chunk *pc = &chunks[first[i][j]]; if (pc->done != 1) ok = 0; a = adler32_combine(a, pc->adler, (z_off_t)pc->len); }
    const unsigned char *t = sec[i] + zl[i] - 4;
    const uLong want = ((uLong)t[0] << 24) | ((uLong)t[1] << 16) | ((uLong)t[2] << 8) | (uLong)t[3];
    if (a != want) ok = 0;
  }
  free(th); free(chunks); free(goff); free(gbytes);
  for (int i = 0; i < 3; i++) free(first[i]);
  if (!ok) fprintf(stderr, "libdctz: a chunk of an indexed section does not inflate; falling back to the one-stream inflate\n");
  *got_out = n;
  return ok;
}

int dctz_decompress(t_var *var_z, t_var *var_r) {
  const double t_begin = now_s();
  const int is_d = (var_z->datatype == DOUBLE);
  const size_t ts = is_d ? sizeof(double) : sizeof(float);
  const int dtype = is_d ? DCTZHIP_F64 : DCTZHIP_F32;
  const unsigned char *cur = is_d ? (const unsigned char *)var_z->buf.d : (const unsigned char *)var_z->buf.f;
  struct header h;
  memcpy(&h, cur, sizeof(h)); /* dctz-decomp-lib.c:84-94 */
  cur += sizeof(h);
  const size_t n = h.num_elements;
  const unsigned int cnt = h.tot_AC_exact_count;
  if (n == 0) { fprintf(stderr, "libdctz: empty stream\n"); exit(1); }
  /* multi-dimensional blocks (dctz.h: dctz_set_block_dims): the extents sit behind the last section */
  const int nd = (int)DCTZ_GEOM_OF(h.datatype);
  size_t dims[3] = {0, 0, 0};
  size_t nblk = CEIL(n, BLK_SZ);
  if (nd) {
    size_t off = (size_t)h.bindex_sz_compressed + h.DC_sz_compressed + h.AC_exact_sz_compressed;
#ifdef USE_QTABLE
    off += BLK_SZ * ts;
#endif
    unsigned int tr[4];
    memcpy(tr, cur + off, sizeof(tr));
    dims[0] = tr[1]; dims[1] = tr[2]; dims[2] = tr[3];
    nblk = (tr[0] == DCTZ_ND_MAGIC && (nd == 2 || nd == 3)) ? dctzhip_nd_blocks(nd, dims) : 0;
    if (!nblk || dims[0] * dims[1] * (nd == 3 ? dims[2] : 1) != n) { fprintf(stderr, "libdctz: bad multi-dimensional container\n"); exit(1); }
  }
  const size_t npos = nd ? nblk * BLK_SZ : n;

  double t0 = now_s();
  dctzhip_ctx *c = ctx();
  const unsigned int zl[3] = {h.bindex_sz_compressed, h.DC_sz_compressed, h.AC_exact_sz_compressed};
  const unsigned char *const secp[3] = {cur, cur + zl[0], cur + zl[0] + zl[1]};
  const size_t rawn[3] = {npos, nblk * sizeof(float), (size_t)cnt * sizeof(float)};
  size_t ix_off = (size_t)zl[0] + zl[1] + zl[2] + (nd ? 16 : 0);
#ifdef USE_QTABLE
  ix_off += BLK_SZ * ts;
#endif
  /* sections written by the GPU entropy stage start 78 5E and bring a chunk index: only then are the bytes behind the
   * container looked at.  On request (DCTZ_INFLATE_GPU=1) they are inflated on the device, one lane per chunk -- the
   * compressed sections go over PCIe instead of the raw streams, no host core inflates (include/dctz_hip.h:
   * dctzhip_inflate); otherwise by host threads, chunks side by side. */
  uint32_t *ix_sizes[3] = {NULL, NULL, NULL};
  size_t ix_chunk_bytes = 0;
  const int indexed = zl[0] >= 8 && zl[1] >= 8 && zl[2] >= 8 && secp[0][1] == 0x5E && secp[1][1] == 0x5E && secp[2][1] == 0x5E &&
                      read_index(secp, zl, rawn, cur + ix_off, &ix_chunk_bytes, ix_sizes);
  int on_device = 0;
  double t_h2d_z = 0.0;
  grow(&g_dev.bin, &g_dev.bin_cap, npos);
  grow(&g_dev.dc, &g_dev.dc_cap, nblk * sizeof(float));
  grow(&g_dev.ac, &g_dev.ac_cap, (cnt ? cnt : 4) * sizeof(float));
  if (indexed && !nd && !inflate_gpu() && pipeline_on()) {
    const void *qt_p = NULL;
#ifdef USE_QTABLE
    double qd_p[BLK_SZ];
    float qf_p[BLK_SZ];
    if (is_d) { memcpy(qd_p, cur + zl[0] + zl[1] + zl[2], sizeof(qd_p)); qt_p = qd_p; } /* :193-199 */
    else { memcpy(qf_p, cur + zl[0] + zl[1] + zl[2], sizeof(qf_p)); qt_p = qf_p; }
#endif
    size_t got_p = 0;
    if (decompress_pipelined(c, &h, secp, zl, ix_chunk_bytes, ix_sizes, qt_p, var_r, &got_p)) {
      for (int i = 0; i < 3; i++) free(ix_sizes[i]);
      if (!quiet()) printf("uncompressed bin_index size is: %lu\n", (unsigned long)got_p); /* :260-262 */
      g_times.zlib_s = 0.0; g_times.h2d_s = 0.0; g_times.gpu_s = 0.0; g_times.d2h_s = 0.0;   /* (the stages overlap: only the total means something) */
      g_times.total_s = now_s() - t_begin;
      return 1;
    }
  }
  if (indexed && inflate_gpu() && ix_chunk_bytes == dctzhip_deflate_chunk_bytes()) {
    size_t zlen[3];
    for (int i = 0; i < 3; i++) {
      zlen[i] = zl[i];
      grow(&g_dev.z[i], &g_dev.z_cap[i], zl[i]);
      if (dctzhip_memcpy_h2d(c, g_dev.z[i], secp[i], zl[i]) != DCTZHIP_OK) die("H2D compressed section");
    }
    t_h2d_z = now_s() - t0;
    void *const ddst[3] = {g_dev.bin, g_dev.dc, g_dev.ac};
    int ok = 0;
    if (dctzhip_inflate(c, 3, (const void *const *)g_dev.z, zlen, (const uint32_t *const *)ix_sizes, rawn, ddst, &ok) != DCTZHIP_OK) die("dctzhip_inflate");
    on_device = ok;            /* 0: inconsistent -- the host path below decides, and reports damage like the reference */
  }

  t_bin_id *bin_index = NULL;
  float *DC = NULL, *AC_exact = NULL;
  uLong got = (uLong)npos;
  if (!on_device) {
  bin_index = (t_bin_id *)host_buf(0, npos);
  DC = (float *)host_buf(1, nblk * sizeof(float));
  AC_exact = (float *)host_buf(2, (size_t)cnt * sizeof(float));
  /* three inflates, in order (dctz-decomp-lib.c:244-322) */
  unsigned char *const rawp[3] = {(unsigned char *)bin_index, (unsigned char *)DC, (unsigned char *)AC_exact};
  if (indexed && inflate_indexed(secp, zl, rawp, rawn, ix_chunk_bytes, ix_sizes)) {   /* chunks side by side on host threads */
    got = (uLong)npos;
  } else if (zlib_threads() == 0 || zlib_threads() > 3) {   /* the sections are independent streams: inflate them side by side
                                                               (DCTZ_ZLIB_THREADS=1..3 keeps the reference's one-after-the-other) */
    inflate_job ij[3] = {{secp[0], zl[0], (uLong)npos, 0, bin_index},
                         {secp[1], zl[1], (uLong)(nblk * sizeof(float)), 0, DC},
                         {secp[2], zl[2], (uLong)((size_t)cnt * sizeof(float)), 0, AC_exact}};
    pthread_t th[2];
    int started = 0;
    for (int i = 1; i < 3; i++) { if (pthread_create(&th[started], NULL, inflate_main, &ij[i])) inflate_main(&ij[i]); else started++; }
    inflate_main(&ij[0]);
    for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
    got = ij[0].produced;
  } else {
    got = inflate_into(secp[0], zl[0], bin_index, npos);
    inflate_into(secp[1], zl[1], DC, nblk * sizeof(float));
    inflate_into(secp[2], zl[2], AC_exact, (size_t)cnt * sizeof(float));
  }
  }
  cur += (size_t)zl[0] + zl[1] + zl[2];
  for (int i = 0; i < 3; i++) free(ix_sizes[i]);
  if (!quiet()) printf("uncompressed bin_index size is: %lu\n", got); /* :260-262 */
  const void *qtable = NULL;
#ifdef USE_QTABLE
  double qd[BLK_SZ];
  float qf[BLK_SZ];
  if (is_d) { memcpy(qd, cur, sizeof(qd)); qtable = qd; } /* :193-199 */
  else { memcpy(qf, cur, sizeof(qf)); qtable = qf; }
#endif
  double t1 = now_s();

  grow(&g_dev.out, &g_dev.out_cap, n * ts);
  if (!on_device) {
    if (dctzhip_memcpy_h2d(c, g_dev.bin, bin_index, npos) != DCTZHIP_OK) die("H2D bin_index");
    if (dctzhip_memcpy_h2d(c, g_dev.dc, DC, nblk * sizeof(float)) != DCTZHIP_OK) die("H2D DC");
    if (cnt && dctzhip_memcpy_h2d(c, g_dev.ac, AC_exact, (size_t)cnt * sizeof(float)) != DCTZHIP_OK) die("H2D AC_exact");
  }
  double t2 = now_s();

  const double sf = is_d ? h.scaling_factor.d : (double)h.scaling_factor.f;
  int rc = nd ? dctzhip_decompress_nd(c, g_dev.bin, (const float *)g_dev.dc, (const float *)g_dev.ac, cnt, qtable, nd, dims,
                                      dtype, h.error_bound, sf, DCTZ_MODE, g_dev.out)
              : dctzhip_decompress(c, g_dev.bin, (const float *)g_dev.dc, (const float *)g_dev.ac, cnt, qtable, n, dtype,
                                   h.error_bound, sf, DCTZ_MODE, g_dev.out);
  if (rc != DCTZHIP_OK) die("dctzhip_decompress");
  if (dctzhip_sync(c) != DCTZHIP_OK) die("sync");   /* (stage timers: the call returns while the reconstruction is being written) */
  double t3 = now_s();
  void *host_out = is_d ? (void *)var_r->buf.d : (void *)var_r->buf.f;
  if (dctzhip_memcpy_d2h(c, host_out, g_dev.out, n * ts) != DCTZHIP_OK) die("D2H output");
  double t4 = now_s();

  g_times.zlib_s = t1 - t0; g_times.h2d_s = t2 - t1; g_times.gpu_s = t3 - t2; g_times.d2h_s = t4 - t3;
  if (on_device) { g_times.zlib_s -= t_h2d_z; g_times.h2d_s += t_h2d_z; }      /* the compressed sections' way to the device is a copy, not inflate */
  g_times.total_s = now_s() - t_begin;
  return 1;
}

/* ------------------------------------------------------ calc_data_stat ----- */
/* util.c:12-44 on the GPU: max/min by tree reduction (order-independent),
 * the sum by the serial-order kernel so that mean is bit-identical. */
void calc_data_stat(t_var *in, t_bstat *bs, int N) {
  const int is_d = (in->datatype == DOUBLE);
  const size_t ts = is_d ? sizeof(double) : sizeof(float);
  const size_t n = (size_t)N;
  dctzhip_ctx *c = ctx();
  grow(&g_dev.in, &g_dev.in_cap, n * ts);
  if (dctzhip_memcpy_h2d(c, g_dev.in, is_d ? (void *)in->buf.d : (void *)in->buf.f, n * ts) != DCTZHIP_OK) die("H2D");
  dctzhip_cinfo info;
  double mean = 0.0;
  if (dctzhip_serial_mean_begin(c, g_dev.in, n, is_d ? DCTZHIP_F64 : DCTZHIP_F32) != DCTZHIP_OK) die("serial mean");
  if (dctzhip_stats(c, g_dev.in, n, is_d ? DCTZHIP_F64 : DCTZHIP_F32, &info) != DCTZHIP_OK) die("dctzhip_stats");
  if (dctzhip_serial_mean_end(c, &mean) != DCTZHIP_OK) die("serial mean");
  if (is_d) { bs->max.d = info.max_abs; bs->min.d = info.min_abs; bs->mean.d = mean; bs->sf.d = info.sf; }
  else { bs->max.f = (float)info.max_abs; bs->min.f = (float)info.min_abs; bs->mean.f = (float)mean; bs->sf.f = (float)info.sf; }
}

/* ------------------------------------------------------------- gen_bins ---- */
/* binning.c:12-50: centre of bin b is (+1,-1,+2,-2,...) times the bin width. */
void gen_bins(double min, double max, double *bin_center, int nbins, double error_bound) {
  (void)min; (void)max;
  const double bin_width = error_bound * 2 * BRSF;
  bin_center[0] = 0.0;
  for (int b = 1; b < nbins; b++) {
    const int signed_step = (b & 1) ? (b / 2) + 1 : -(b / 2);
    bin_center[b] = signed_step * bin_width;
  }
}

void gen_bins_f(float min, float max, float *bin_center, int nbins, float error_bound) {
  (void)min; (void)max;
  const float bin_width = error_bound * 2 * BRSF;
  bin_center[0] = 0.0;
  for (int b = 1; b < nbins; b++) {
    const int signed_step = (b & 1) ? (b / 2) + 1 : -(b / 2);
    bin_center[b] = signed_step * bin_width;
  }
}

/* ------------------------------------------------------------- calc_psnr --- */
/* util.c:54-104 (harness metric; not part of the codec). */
/* Arrays of at least this many elements take the GPU reductions (dctzhip_psnr_terms: min, max and max |e| exact,
 * the sum of squares in tree order, 1e-15 relative to the serial loop); DCTZ_PSNR_HOST=1 keeps the loop below. */
#define PSNR_GPU_MIN (1 << 16)
double calc_psnr(t_var *var, t_var *var_r, int N, double error_bound) {
  (void)error_bound;
  double lo, hi, worst = 0.0, sq = 0.0;
  if (N >= PSNR_GPU_MIN && !getenv("DCTZ_PSNR_HOST")) {
    const int is_d = var->datatype == DOUBLE;
    const size_t bytes = (size_t)N * (is_d ? sizeof(double) : sizeof(float));
    dctzhip_ctx *c = ctx();
    double t[4];
    grow(&g_dev.in, &g_dev.in_cap, bytes);
    grow(&g_dev.out, &g_dev.out_cap, bytes);
    if (dctzhip_memcpy_h2d(c, g_dev.in, is_d ? (void *)var->buf.d : (void *)var->buf.f, bytes) != DCTZHIP_OK ||
        dctzhip_memcpy_h2d(c, g_dev.out, is_d ? (void *)var_r->buf.d : (void *)var_r->buf.f, bytes) != DCTZHIP_OK)
      die("H2D");
    if (dctzhip_psnr_terms(c, g_dev.in, g_dev.out, (size_t)N, is_d ? DCTZHIP_F64 : DCTZHIP_F32, t) != DCTZHIP_OK)
      die("dctzhip_psnr_terms");
    lo = t[0]; hi = t[1]; worst = t[2]; sq = t[3];
  } else if (var->datatype == DOUBLE) {
    const double *x = var->buf.d, *r = var_r->buf.d;
    lo = hi = x[0];
    for (int i = 1; i < N; i++) { if (x[i] > hi) hi = x[i]; if (x[i] < lo) lo = x[i]; }
    for (int i = 0; i < N; i++) {
      const double e = x[i] - r[i];
      if (fabs(e) > worst) worst = fabs(e);
      sq += e * e;
    }
  } else {
    const float *x = var->buf.f, *r = var_r->buf.f;
    lo = hi = x[0];
    for (int i = 1; i < N; i++) { if (x[i] > hi) hi = x[i]; if (x[i] < lo) lo = x[i]; }
    for (int i = 0; i < N; i++) {
      const float e = x[i] - r[i];
      if (fabs(e) > worst) worst = fabs(e);
      sq += (e * e);
    }
  }
  const double rmse = sqrt(sq / N), range = hi - lo;
  printf("Max relative error = %.6f\n", worst / range); /* util.c:95 */
  return 20 * log10(range / rmse);
}

/* ------------------------------------------------- dct.h transform layer --- */
/* dct_init / dct_finish keep no state here (the GPU context owns the tables,
 * rebuilt per length on demand); they exist so dct-test.c links and runs. */
void dct_init(int dn) { (void)dn; (void)ctx(); }
void dct_init_f(int dn) { (void)dn; (void)ctx(); }
void dct_finish(void) {}
void dct_finish_f(void) {}
void idct_finish(void) {}
void idct_finish_f(void) {}

static void blocks(void *a, void *b, size_t n, int is_d, int inverse) {
  const size_t ts = is_d ? sizeof(double) : sizeof(float);
  dctzhip_ctx *c = ctx();
  grow(&g_dev.in, &g_dev.in_cap, n * ts);
  grow(&g_dev.out, &g_dev.out_cap, n * ts);
  if (dctzhip_memcpy_h2d(c, g_dev.in, a, n * ts) != DCTZHIP_OK) die("H2D");
  if (dctzhip_dct_blocks(c, g_dev.in, g_dev.out, n, is_d ? DCTZHIP_F64 : DCTZHIP_F32, inverse) != DCTZHIP_OK)
    die("dctzhip_dct_blocks");
  if (dctzhip_memcpy_d2h(c, b, g_dev.out, n * ts) != DCTZHIP_OK) die("D2H");
}

void dctz_dct_blocks(double *a, double *b, size_t n, int inverse) { blocks(a, b, n, 1, inverse); }
void dctz_dct_blocks_f(float *a, float *b, size_t n, int inverse) { blocks(a, b, n, 0, inverse); }

/* one block of length dn per call, like the reference (dct.c:55, :115).  The reference transforms ANY length with one
 * length-dn plan; the codec only ever uses dn <= BLK_SZ = 64 (dctz.h:28), which is what the GPU tables cover: a longer
 * block is refused loudly instead of being cut into 64-element blocks (a different transform). */
/* ... on the HOST (dct_host.cpp): the caller's block is in host memory and is read back at once; the product's own lane
 * flow (dct64_block.h, the code one GPU lane runs) compiled for the CPU gives bit for bit what dctz_dct_blocks() returns
 * from the device, without an H2D copy, a launch and a D2H copy per 512 bytes.  (DCTZ_BLOCK_ON_GPU=1: through the GPU.) */
void dctz_host_block_f64(const double *a, double *b, int dn, int inverse);
void dctz_host_block_f32(const float *a, float *b, int dn, int inverse);
static void one_block(void *a, void *b, int dn, int is_d, int inverse) {
  static int on_gpu = -1;
  if (dn < 1 || dn > BLK_SZ) {
    fprintf(stderr, "libdctz: %s with dn = %d: only block lengths 1..%d are supported\n", inverse ? "ifft_idct" : "dct_fftw", dn, BLK_SZ);
    exit(1);
  }
  if (on_gpu < 0) { const char *e = getenv("DCTZ_BLOCK_ON_GPU"); on_gpu = e && atoi(e) ? 1 : 0; }
  if (on_gpu) { blocks(a, b, (size_t)dn, is_d, inverse); return; }
  if (is_d) dctz_host_block_f64((const double *)a, (double *)b, dn, inverse);
  else dctz_host_block_f32((const float *)a, (float *)b, dn, inverse);
}
void dct_fftw(double *a, double *b, int dn, int nblk) { (void)nblk; one_block(a, b, dn, 1, 0); }
void dct_fftw_f(float *a, float *b, int dn, int nblk) { (void)nblk; one_block(a, b, dn, 0, 0); }
void ifft_idct(int dn, double *a, double *data) { one_block(a, data, dn, 1, 1); }
void ifft_idct_f(int dn, float *a, float *data) { one_block(a, data, dn, 0, 1); }
