/* pdeflate.h -- chunked multi-threaded deflate producing one standard zlib stream per
 * section (see pdeflate.c).  Host-only; part of the drop-in libraries. */
#ifndef DCTZ_PDEFLATE_H
#define DCTZ_PDEFLATE_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  const void *src;     /* section input                           */
  size_t n;            /* bytes                                   */
  void *dst;           /* output buffer                           */
  size_t cap;          /* its size, >= dctz_pdeflate_bound(n, chunk) */
  size_t *out_len;     /* bytes written                           */
} dctz_pd_section;

/* Worst-case output size of one section. */
size_t dctz_pdeflate_bound(size_t n, size_t chunk);
/* Deflate all sections with one shared pool of `threads` workers (the calling thread is
 * one of them); chunk = bytes per job (>= 32 KiB).  0 on success. */
int dctz_pdeflate_many(const dctz_pd_section *sec, int nsec, int threads, size_t chunk);
/* deflate level of subsequent calls: 1..9, anything else = zlib's default (what the reference uses) */
void dctz_pdeflate_set_level(int level);
int dctz_pdeflate(const void *src, size_t n, void *dst, size_t cap, size_t *out_len, int threads, size_t chunk);

#ifdef __cplusplus
}
#endif
#endif
