// Host twin of dctz_amd/csrc/dctz_deflate.hip: the same per-chunk steps (deflate_chunk.h routines), the lanes of a
// workgroup replaced by loops.  Test infrastructure: lets the CPU suite check the stream format against zlib's inflate
// without a GPU, and the GPU suite compare the device's bytes with these.
// Build: g++ -O2 -shared -fPIC -I dctz_amd/csrc tests/emu/emu_deflate.cpp -o tests/emu/emu_deflate.so
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "deflate_chunk.h"

using namespace dctz::dfl;

namespace {
struct Words {
  std::vector<uint32_t> w;
  void operator()(uint32_t i, uint32_t v) { if (i >= w.size()) w.resize(i + 1, 0); w[i] |= v; }
};
struct OrRef { Words* o; void operator()(uint32_t i, uint32_t v) const { (*o)(i, v); } };

void sort_used(const uint32_t* freq, int n, std::vector<int>& sorted) {
  sorted.clear();
  for (int i = 0; i < n; i++) if (freq[i]) sorted.push_back(i);
  std::stable_sort(sorted.begin(), sorted.end(), [&](int a, int b) { return freq[a] < freq[b]; });
}
void lengths_codes(const uint32_t* freq, int n, int maxbits, uint8_t* len, uint16_t* code) {
  std::vector<int> sorted;
  sort_used(freq, n, sorted);
  const int k = (int)sorted.size();
  std::vector<uint32_t> w(k);
  std::vector<uint16_t> ch(2 * k), dep(k);
  uint16_t bl[16], nc[16];
  memset(len, 0, n);
  huff_lengths([&](int s) { return freq[s]; }, [&](int i) { return sorted[i]; }, k, maxbits, [&](int s, int b) { len[s] = (uint8_t)b; },
               w.data(), ch.data(), dep.data(), bl);
  first_codes(bl, maxbits, nc);
  for (int i = 0; i < n; i++) {
    code[i] = 0;
    if (!len[i]) continue;
    int before = 0;
    for (int j = 0; j < i; j++) before += len[j] == len[i];
    code[i] = (uint16_t)bit_reverse((uint32_t)nc[len[i]] + before, len[i]);
  }
}
}  // namespace

extern "C" size_t emu_deflate_bound(size_t n, int nthr) {
  const size_t chunk = (size_t)nthr * SEG, nch = (n + chunk - 1) / chunk;
  return n + 5 * nch + 8;
}

// returns the stream length (0 if cap is too small)
static size_t emu_deflate_ix(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, int nthr, uint32_t* sizes, int literals = 0);
extern "C" size_t emu_deflate(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, int nthr) { return emu_deflate_ix(src, n, dst, cap, nthr, nullptr); }
// the same, and the compressed bytes of every chunk (the container's chunk index)
extern "C" size_t emu_deflate_index(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, int nthr, uint32_t* sizes) { return emu_deflate_ix(src, n, dst, cap, nthr, sizes); }
// the same without the match search (DCTZHIP_DEFLATE_LITERALS)
extern "C" size_t emu_deflate_literals(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, int nthr, uint32_t* sizes) { return emu_deflate_ix(src, n, dst, cap, nthr, sizes, 1); }
static size_t emu_deflate_ix(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, int nthr, uint32_t* sizes, int literals) {
  const size_t chunk = (size_t)nthr * SEG;
  if (cap < emu_deflate_bound(n, nthr)) return 0;
  size_t pos = 0;
  dst[pos++] = 0x78; dst[pos++] = 0x5E;
  uint32_t s1 = 1, s2 = 0;
  for (size_t i = 0; i < n; i++) { s1 = (s1 + src[i]) % 65521u; s2 = (s2 + s1) % 65521u; }
  std::vector<uint8_t> tok(chunk + 2);
  for (size_t off = 0; off < n; off += chunk) {
    const int len = (int)std::min(chunk, n - off);
    const int avail = 0;
    const uint8_t* base = src + off;
    auto in = [&](int i) -> int { return base[i]; };
    uint32_t fl[NLIT] = {0}, fd[NDIST] = {0}, fc[NCL] = {0};
    if (literals) for (int p = 0; p < len; p++) { tok[p] = 0; fl[base[p]]++; }
    else {
      const int nseg = (len + SEG - 1) / SEG;
      for (int t = 0; t < nseg; t++)
        parse_segment(in, [&](int p, int v) { tok[p] = (uint8_t)v; }, t * SEG, std::min((t + 1) * SEG, len), avail, [](int) {}, [](int) {});
      // merge across segment boundaries (deflate_chunk.h: merge_allowed), every boundary decided from the ORIGINAL tokens
      std::vector<int> last_p(nseg), last_len(nseg), ntok(nseg), first_len(nseg), first_c(nseg), last_c(nseg);
      for (int t = 0; t < nseg; t++) {
        const int p0 = t * SEG, p1 = std::min((t + 1) * SEG, len);
        int k = 0, lp = p0, ll = 0, lc = -1;
        first_len[t] = 0; first_c[t] = -1;
        for (int p = p0; p < p1;) {
          int l = 1, c = -1;
          if (tok[p]) { l = tok[p + 1] + 3; c = tok[p] - 1; }
          if (k == 0) { first_len[t] = l; first_c[t] = c; }
          lp = p; ll = l; lc = c; k++;
          p += l;
        }
        last_p[t] = lp; last_len[t] = ll; last_c[t] = lc; ntok[t] = k;
      }
      for (int t = 0; t + 1 < nseg; t++) {
        const bool reaches = last_c[t] >= 0 && last_p[t] + last_len[t] == (t + 1) * SEG;
        if (!reaches || first_c[t + 1] != last_c[t] || last_len[t] + first_len[t + 1] > MAXMATCH) continue;
        if (!merge_allowed(t, ntok[t] == 1, ntok[t + 1] == 1)) continue;
        tok[last_p[t] + 1] = (uint8_t)(last_len[t] + first_len[t + 1] - 3);
        tok[(t + 1) * SEG] = TOK_ABSORBED;
      }
      for (int p = 0; p < len;) {                           // counts of the merged tokens
        if (tok[p] == 0) { fl[base[p]]++; p++; }
        else {
          const int l = tok[p + 1] + 3, c = tok[p] - 1;
          int sym, eb, ev;
          len_code(l, sym, eb, ev);
          fl[sym]++; fd[cand_dsym_rt(c)]++;
          p += l;
        }
      }
    }
    fl[256] = 1;
    int used = 0;
    for (int i = 0; i < NDIST; i++) used += fd[i] ? 1 : 0;
    for (int i = 0; used < 2; i++) if (!fd[i]) { fd[i] = 1; used++; }
    uint8_t ll[NLIT], dl[NDIST], cll[NCL];
    uint16_t lc[NLIT], dc[NDIST], clc[NCL];
    lengths_codes(fl, NLIT, MAXBITS, ll, lc);
    lengths_codes(fd, NDIST, MAXBITS, dl, dc);
    uint64_t tokbits = 0;
    for (int p = 0; p < len;) {
      if (tok[p] == 0) { tokbits += ll[base[p]]; p++; }
      else {
        const int l = tok[p + 1] + 3, c = tok[p] - 1;
        int sym, eb, ev;
        len_code(l, sym, eb, ev);
        tokbits += ll[sym] + eb + dl[cand_dsym_rt(c)] + cand_deb_rt(c);
        p += l;
      }
    }
    int hlit = NLIT, hdist = NDIST;
    while (hlit > 257 && ll[hlit - 1] == 0) hlit--;
    while (hdist > 1 && dl[hdist - 1] == 0) hdist--;
    std::vector<uint16_t> cl;
    auto outcl = [&](int sym, int, int ev) { cl.push_back((uint16_t)(sym | (ev << 8))); fc[sym]++; };
    rle_lengths([&](int i) { return (int)ll[i]; }, hlit, outcl);
    rle_lengths([&](int i) { return (int)dl[i]; }, hdist, outcl);
    int k = 0;
    for (int i = 0; i < NCL; i++) k += fc[i] ? 1 : 0;
    for (int i = 0; k < 2; i++) if (!fc[i]) { fc[i] = 1; k++; }
    lengths_codes(fc, NCL, MAXBITS_CL, cll, clc);
    int hclen = NCL;
    while (hclen > 4 && cll[cl_order(hclen - 1)] == 0) hclen--;
    uint32_t hb = 3 + 5 + 5 + 4 + 3 * hclen;
    for (uint16_t e : cl) { const int sym = e & 31; hb += cll[sym] + (sym == 16 ? 2 : sym == 17 ? 3 : sym == 18 ? 7 : 0); }
    const uint32_t body_bits = hb + (uint32_t)tokbits + ll[256];
    const uint32_t dyn_bytes = (body_bits + 3 + 7) / 8 + 4, stored_bytes = (uint32_t)len + 5;
    if (dyn_bytes >= stored_bytes) {
      dst[pos++] = 0;
      dst[pos++] = (uint8_t)(len & 255); dst[pos++] = (uint8_t)(len >> 8);
      dst[pos++] = (uint8_t)(~len & 255); dst[pos++] = (uint8_t)((~len >> 8) & 255);
      memcpy(dst + pos, base, len);
      pos += len;
      if (sizes) sizes[off / chunk] = stored_bytes;
      continue;
    }
    Words out;
    out.w.assign(dyn_bytes / 4 + 2, 0);
    OrRef orw{&out};
    BitW<OrRef> bw(orw, 0);
    bw.put(0u | (2u << 1), 3);
    bw.put((uint32_t)(hlit - 257), 5);
    bw.put((uint32_t)(hdist - 1), 5);
    bw.put((uint32_t)(hclen - 4), 4);
    for (int i = 0; i < hclen; i++) bw.put(cll[cl_order(i)], 3);
    for (uint16_t e : cl) {
      const int sym = e & 31, ev = e >> 8;
      bw.put(clc[sym], cll[sym]);
      if (sym == 16) bw.put((uint32_t)ev, 2);
      else if (sym == 17) bw.put((uint32_t)ev, 3);
      else if (sym == 18) bw.put((uint32_t)ev, 7);
    }
    for (int p = 0; p < len;) {
      if (tok[p] == 0) { bw.put(lc[base[p]], ll[base[p]]); p++; }
      else {
        const int l = tok[p + 1] + 3, c = tok[p] - 1;
        int sym, eb, ev;
        len_code(l, sym, eb, ev);
        bw.put(lc[sym], ll[sym]);
        if (eb) bw.put((uint32_t)ev, eb);
        const int ds = cand_dsym_rt(c), de = cand_deb_rt(c);
        bw.put(dc[ds], dl[ds]);
        if (de) bw.put((uint32_t)cand_dev_rt(c), de);
        p += l;
      }
    }
    bw.put(lc[256], ll[256]);
    bw.flush();
    const uint32_t end_byte = (body_bits + 3 + 7) / 8;
    BitW<OrRef> tail(orw, (uint64_t)(end_byte + 2) * 8);
    tail.put(0xFFFFu, 16);
    tail.flush();
    out.w.resize(dyn_bytes / 4 + 2, 0);
    memcpy(dst + pos, out.w.data(), dyn_bytes);
    pos += dyn_bytes;
    if (sizes) sizes[off / chunk] = dyn_bytes;
  }
  dst[pos++] = 0x03; dst[pos++] = 0x00;
  dst[pos++] = (uint8_t)(s2 >> 8); dst[pos++] = (uint8_t)s2; dst[pos++] = (uint8_t)(s1 >> 8); dst[pos++] = (uint8_t)s1;
  return pos;
}
