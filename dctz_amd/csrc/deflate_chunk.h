// deflate_chunk.h -- the sequential pieces of the GPU entropy stage, written once for the device
// (dctz_deflate.hip: one workgroup per chunk, one lane per 128-byte segment) and for the host twin under tests/emu
// (same routines driven by plain loops), so that the two can be compared byte for byte.
//
// What it replaces: the reference's zlib tail -- compress-side deflateInit/deflate of bin_index, DC and AC_exact, one
// zlib stream per section (dctz-comp-lib.c:620-732) -- which is 85-90 % of the reference's compress wall time and all
// that is left once the block-DCT stage runs on the GPU.  The stream stays what the reference's reader inflates
// (dctz-decomp-lib.c:244-322: inflateInit + inflate of one zlib stream per section): RFC 1950 framing, RFC 1951 blocks.
//
// Format of one section:   78 5E | chunk 0 | chunk 1 | ... | 03 00 | adler32 (big endian)
// Every chunk (CHUNK input bytes) is one deflate block, not final, ending on a byte boundary:
//   - dynamic Huffman block (BTYPE 10) followed by an empty stored block (the "sync flush" marker 00 00 FF FF), or
//   - one stored block (BTYPE 00) when that is smaller.
// 03 00 is the final, empty fixed-Huffman block.  Matches are searched at a fixed set of distances only (runs, the
// previous block's pattern 64 positions back, ...: DFL_CANDS) inside the 128-byte segment of one lane, so that every lane
// parses its segment without waiting for a neighbour (a match that reaches its segment's end is afterwards joined with a
// same-distance match at the start of the next one: "merge" below); distances reach back across segment boundaries but
// never across the start of the chunk: every chunk is a deflate block that can be inflated on its own, given where it
// starts (the container's chunk index, include/dctz.h: "DZIX", lets a reader inflate the chunks of a section in parallel).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DFL_HD __host__ __device__ __forceinline__
#else
#define DFL_HD inline
#endif

namespace dctz {
namespace dfl {

enum : int {
  SEG = 128,             // bytes one lane tokenises
  SEG_SHIFT = 7,
  HIST = 0,              // bytes of the previous chunk a match may reach into: none (chunks inflate independently)
  NLIT = 286,            // literal / length alphabet in use (0..255, 256 = end of block, 257..285)
  NDIST = 30,
  NCL = 19,
  MAXBITS = 15,
  MAXBITS_CL = 7,
  MINMATCH = 3,
  MAXMATCH = 258,
};

// Candidate distances of the match search.  On DCTZ streams two of them carry the ratio -- 1 (runs of the centre bin)
// and 64 (the same position of the previous block) -- and every further candidate costs the parse kernel a dozen vector
// instructions per position; measured on the test workloads (flat, 8 x 8 and 4 x 4 x 4 tiles, smooth and noisy), the sets
// {1, 64}, {1, 2, 64, 128} and {1, 2, 4, ..., 128} give the same section sizes to 0.4 %.
#ifndef DFL_CANDS
#define DFL_NCAND 4
#define DFL_CANDS {1, 2, 64, 128}
#endif
enum : int { NCAND = DFL_NCAND };

DFL_HD int ilog2(uint32_t x) { return 31 - __builtin_clz(x); }

// length 3..258 -> (symbol 257.., extra bit count, extra value)     RFC 1951 3.2.5
DFL_HD void len_code(int len, int& sym, int& eb, int& ev) {
  const int x = len - 3;
  if (x < 8) { sym = 257 + x; eb = 0; ev = 0; return; }
  if (x == 255) { sym = 285; eb = 0; ev = 0; return; }
  eb = ilog2((uint32_t)x) - 2;
  sym = 257 + 4 * (eb + 1) + ((x >> eb) & 3);
  ev = x & ((1 << eb) - 1);
}
// distance 1..32768 -> (symbol 0..29, extra bit count, extra value)
DFL_HD constexpr void dist_code(int dist, int& sym, int& eb, int& ev) {
  const int x = dist - 1;
  if (x < 4) { sym = x; eb = 0; ev = 0; return; }
  int l = 0;
  for (int t = x; t > 1; t >>= 1) l++;
  eb = l - 1;
  sym = 2 * eb + 2 + ((x >> eb) & 1);
  ev = x & ((1 << eb) - 1);
}
struct CandTab { int dist[NCAND], dsym[NCAND], deb[NCAND], dev[NCAND], maxdist; };
DFL_HD constexpr CandTab make_cand_tab() {
  CandTab t{};
  constexpr int d[NCAND] = DFL_CANDS;
  t.maxdist = 0;
  for (int c = 0; c < NCAND; c++) {
    t.dist[c] = d[c];
    int s = 0, e = 0, v = 0;
    dist_code(d[c], s, e, v);
    t.dsym[c] = s; t.deb[c] = e; t.dev[c] = v;
    if (d[c] > t.maxdist) t.maxdist = d[c];
  }
  return t;
}
DFL_HD constexpr int cand_dist(int c) { return make_cand_tab().dist[c]; }
DFL_HD constexpr int cand_dsym(int c) { return make_cand_tab().dsym[c]; }
DFL_HD constexpr int cand_maxdist() { return make_cand_tab().maxdist; }
// the same by a run-time index (small constant tables)
DFL_HD int cand_dist_rt(int c) { constexpr CandTab t = make_cand_tab(); return t.dist[c]; }
DFL_HD int cand_dsym_rt(int c) { constexpr CandTab t = make_cand_tab(); return t.dsym[c]; }
DFL_HD int cand_deb_rt(int c) { constexpr CandTab t = make_cand_tab(); return t.deb[c]; }
DFL_HD int cand_dev_rt(int c) { constexpr CandTab t = make_cand_tab(); return t.dev[c]; }

// ------------------------------------------------------------------ parse --
// Tokens of the segment [p0, p1) of a chunk.  in(i): input byte at chunk offset i, i in [-HIST, len); avail = bytes that
// exist in front of the chunk (HIST, or less at the head of the section).  Token record, indexed by position:
// tok(p) = 0: literal in(p);  tok(p) = 1 + c: match at distance cand_dist(c), its length - 3 in tok(p + 1).
// lit(sym) / dst(sym) count one use of a literal-length / distance symbol.
// info(p, length, c): called once per token in order (c = candidate index of a match, -1 for a literal).
struct NoTokInfo { DFL_HD void operator()(int, int, int) const {} };
template <class In, class TokW, class CountL, class CountD, class Info = NoTokInfo>
DFL_HD void parse_segment(In in, TokW tokw, int p0, int p1, int avail, CountL lit, CountD dst, Info info = Info()) {
  int p = p0;
  while (p < p1) {
    int best = 0, bc = 0;
    if (p + MINMATCH <= p1) {
      const int b0 = in(p), b1 = in(p + 1), b2 = in(p + 2);
      // first bytes of all candidates in one batch of loads (a literal -- the common case -- costs one round trip)
      unsigned m = 0;
#pragma unroll
      for (int c = 0; c < NCAND; c++) {
        const int d = cand_dist(c);
        const int q = (p + avail >= d) ? p - d : p;       // (an unavailable candidate reads p itself and is masked out)
        m |= (unsigned)((in(q) == b0) & (p + avail >= d)) << c;
      }
      while (m) {
        const int c = __builtin_ctz(m);
        m &= m - 1;
        const int d = cand_dist_rt(c);
        if (in(p + 1 - d) != b1 || in(p + 2 - d) != b2) continue;
        int l = 3;
        for (;;) {                                        // four positions per round trip
          const int room = p1 - (p + l) < MAXMATCH - l ? p1 - (p + l) : MAXMATCH - l;
          if (room >= 4) {
            const int e0 = in(p + l) == in(p + l - d), e1 = in(p + l + 1) == in(p + l + 1 - d), e2 = in(p + l + 2) == in(p + l + 2 - d),
                      e3 = in(p + l + 3) == in(p + l + 3 - d);
            const int run = e0 ? (e1 ? (e2 ? (e3 ? 4 : 3) : 2) : 1) : 0;
            l += run;
            if (run < 4) break;
          } else {
            int r = 0;
            while (r < room && in(p + l + r) == in(p + l + r - d)) r++;
            l += r;
            break;
          }
        }
        if (l > best) { best = l; bc = c; }
      }
    }
    if (best) {
      tokw(p, 1 + bc);
      tokw(p + 1, best - 3);
      int s, e, v;
      len_code(best, s, e, v);
      lit(s);
      dst(cand_dsym_rt(bc));
      info(p, best, bc);
      p += best;
    } else {
      tokw(p, 0);
      lit(in(p));
      info(p, 1, -1);
      p++;
    }
  }
}

// The same tokens, computed the way the GPU likes it (k_dfl_parse; the loop above stays the definition, and the host twin
// under tests/emu runs it: the two must agree byte for byte).  The loop above is a walk with data-dependent steps, a
// handful of byte loads and ~170 vector instructions per position.  Here:
//   pass A, backwards over the segment in aligned dwords, every lane of a wave at the same offset of its own segment:
//     for every candidate distance the bytes d positions back (aligned dwords for multiples of four, two dwords and a
//     byte shift otherwise) are compared four at a time, and a running count per candidate -- "this position and how many
//     behind it repeat at distance d" = the length the walk above would find from here -- is kept; the longest (first
//     candidate on a tie, as above) goes into a 16-bit entry per position: (length << 2) | (NCAND - 1 - candidate);
//   pass B, forwards: the greedy walk itself, one entry per TOKEN.
// in4(i): the four input bytes at chunk offset i (i a multiple of four, >= 0; bytes beyond the chunk's end read as anything:
// they are masked); tmpw(i, lo, hi): entries of positions i .. i + 3 (two per word); tmpr(p): entry of position p.
template <class In, class In4, class TmpW, class TmpR, class TokW, class CountL, class CountD, class Info = NoTokInfo>
DFL_HD void parse_segment_dwords(In in, In4 in4, TmpW tmpw, TmpR tmpr, TokW tokw, int p0, int p1, CountL lit, CountD dst, Info info = Info()) {
  static_assert(HIST == 0, "candidates never reach in front of the chunk");
  static_assert(NCAND <= 4 && SEG <= 128, "an entry is (length << 2) | (NCAND - 1 - candidate)");
  // R[c] = (running length << 2) | (NCAND - 1 - c): the maximum over the candidates is the longest run, the FIRST
  // candidate on a tie -- and is the position's entry as it stands
  uint32_t R[NCAND];
#pragma unroll
  for (int c = 0; c < NCAND; c++) R[c] = (uint32_t)(NCAND - 1 - c);
  const int pe = (p1 + 3) & ~3;
  // positions of the segment's last dword that lie inside it (all four but for the section's last segment)
  uint32_t inside = (p1 & 3) ? (0x80808080u >> (8 * (4 - (p1 & 3)))) : 0x80808080u;
  uint32_t cur = in4(pe - 4);
  for (int w = pe - 4; w >= p0; w -= 4) {
    // every load of the trip up front, from clamped offsets; what lies in front of the chunk is replaced by bytes that
    // cannot compare equal (the complement of what they are compared with)
    const uint32_t prev = in4(w >= 4 ? w - 4 : 0);
    uint32_t hi[NCAND], lo[NCAND];
#pragma unroll
    for (int c = 0; c < NCAND; c++) {
      const int d = cand_dist(c), a = d >> 2, r = d & 3;
      hi[c] = a == 0 ? cur : in4(w >= 4 * a ? w - 4 * a : 0);
      lo[c] = r == 0 ? 0u : (a == 0 ? prev : in4(w >= 4 * a + 4 ? w - 4 * a - 4 : 0));
    }
    uint32_t z[NCAND];
#pragma unroll
    for (int c = 0; c < NCAND; c++) {
      const int d = cand_dist(c), a = d >> 2, r = d & 3;
      uint32_t ref;
      if (r == 0) ref = w >= d ? hi[c] : ~cur;
      else {
        const uint32_t h = w >= 4 * a ? hi[c] : ~(cur >> (8 * r));                 // (bytes r .. 3 of ref)
        const uint32_t l = w >= 4 * a + 4 ? lo[c] : ~(cur << (32 - 8 * r));        // (bytes 0 .. r - 1 of ref)
        ref = (h << (8 * r)) | (l >> (32 - 8 * r));
      }
      const uint32_t x = cur ^ ref;
      z[c] = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & inside;                      // bit 7 of byte k: position w + k repeats at distance d
    }
    inside = 0x80808080u;
    uint32_t e[4];
#pragma unroll
    for (int k = 3; k >= 0; k--) {
      uint32_t key = 0;
#pragma unroll
      for (int c = 0; c < NCAND; c++) {
        R[c] = ((z[c] >> (8 * k + 7)) & 1u) ? R[c] + 4u : (uint32_t)(NCAND - 1 - c);
        key = R[c] > key ? R[c] : key;
      }
      e[k] = key;
    }
    tmpw(w, e[0] | (e[1] << 16), e[2] | (e[3] << 16));
    cur = prev;
  }
  // (one sequence of instructions for both kinds of token: lanes of a wave are at different places of their streams, and
  // a loop with a branch per kind runs both branches nearly every trip)
  for (int p = p0; p < p1;) {
    const uint32_t en = tmpr(p);
    const int byte = in(p);
    const int best = (int)(en >> 2), bc = NCAND - 1 - (int)(en & 3u);
    const bool m = best >= MINMATCH;
    int s, eb, v;
    len_code(m ? best : MINMATCH, s, eb, v);
    tokw(p, m ? 1 + bc : 0);
    if (m) tokw(p + 1, best - 3);
    lit(m ? s : byte);
    if (m) dst(cand_dsym_rt(bc));
    info(p, m ? best : 1, m ? bc : -1);
    p += m ? best : 1;
  }
}

// ------------------------------------------------------------------ merge --
// A match that ends with its segment may go on into the next one: when segment t's LAST token is a match that reaches the
// segment's end and segment t + 1 BEGINS with a match at the same distance, the two are one match of the summed length
// (always <= 258: both are <= 128 ... 130), which halves the tokens of long runs and of repeated blocks.  Decided per
// boundary from the two tokens alone, so that every lane can do it for its own end: tok(start of segment t + 1) becomes
// ABSORBED (its length byte stays), the last token of segment t gets the summed length.  A segment that is ONE token
// from end to end could be both absorbed and absorbing; it is absorbed when t is odd and absorbs when t is even.
enum : int { TOK_ABSORBED = 255 };
// last_p: start of segment t's last token (a match), last_len its length, single: it is the segment's only token.
// Returns the length to add to it (0: no merge); the caller of segment t + 1 tests absorbed_by_previous() the same way.
DFL_HD bool merge_allowed(int t, bool left_single, bool right_single) {
  // left = segment t, right = segment t + 1
  if (left_single && (t & 1)) return false;             // an odd single-token segment is itself absorbed (or stays), never absorbs
  if (right_single && !((t + 1) & 1)) return false;     // an even single-token segment absorbs (or stays), is never absorbed
  return true;
}

// --------------------------------------------------------------- Huffman --
// Code lengths, limited to maxbits, of the k symbols sorted[0..k) (ascending frequency, ties by symbol; every one with
// freq > 0; k >= 2).  Two-queue construction, leaf counts per depth, the overflow repair of zlib's trees.c gen_bitlen, then
// the longest codes go to the rarest symbols.  Scratch: w[k] (node weights), ch[2k] (children), dep[k].
template <class F, class S, class L>
DFL_HD void huff_lengths(F freq, S sorted, int k, int maxbits, L len, uint32_t* w, uint16_t* ch, uint16_t* dep, uint16_t* bl_count) {
  for (int b = 0; b <= MAXBITS; b++) bl_count[b] = 0;
  // children: id < k = leaf sorted[id]; id >= k = internal node id - k
  int li = 0, ii = 0, ni = 0;
  for (; ni < k - 1; ni++) {
    uint32_t wsum = 0;
    for (int t = 0; t < 2; t++) {
      const bool take_leaf = li < k && (ii >= ni || freq(sorted(li)) <= w[ii]);
      if (take_leaf) { wsum += freq(sorted(li)); ch[2 * ni + t] = (uint16_t)li; li++; }
      else { wsum += w[ii]; ch[2 * ni + t] = (uint16_t)(k + ii); ii++; }
    }
    w[ni] = wsum;
  }
  dep[k - 2] = 0;
  for (int i = k - 2; i >= 0; i--) {
    const int d = dep[i] + 1;
    for (int t = 0; t < 2; t++) {
      const int c = ch[2 * i + t];
      if (c >= k) dep[c - k] = (uint16_t)d;
      else bl_count[d > maxbits ? maxbits : d]++;
    }
  }
  // leaves clamped to maxbits over-subscribe the code: every repair step (one leaf one level down, one leaf of the
  // deepest level up as its sibling) takes 2^-maxbits off the Kraft sum
  uint32_t kraft = 0;
  for (int b = 1; b <= maxbits; b++) kraft += (uint32_t)bl_count[b] << (maxbits - b);
  for (uint32_t over = kraft - (1u << maxbits); over > 0; over--) {
    int bits = maxbits - 1;
    while (bl_count[bits] == 0) bits--;
    bl_count[bits]--;
    bl_count[bits + 1] += 2;
    bl_count[maxbits]--;
  }
  int h = 0;
  for (int bits = maxbits; bits >= 1; bits--)
    for (int n = bl_count[bits]; n > 0; n--) len(sorted(h++), bits);
}

// first canonical code of every length (RFC 1951 3.2.2)
DFL_HD void first_codes(const uint16_t* bl_count, int maxbits, uint16_t* next_code) {
  uint32_t code = 0;
  next_code[0] = 0;
  for (int bits = 1; bits <= maxbits; bits++) {
    code = (code + (bits > 1 ? bl_count[bits - 1] : 0)) << 1;
    next_code[bits] = (uint16_t)code;
  }
}
DFL_HD uint32_t bit_reverse(uint32_t code, int len) {
  uint32_t r = 0;
  for (int i = 0; i < len; i++) { r = (r << 1) | (code & 1); code >>= 1; }
  return r;
}

// ------------------------------------------------------------ bit writer --
// LSB-first bit stream into 32-bit words that several writers share: every word is OR-ed in (orw(word index, value)),
// so a writer may start and end in the middle of a word.
template <class OrW>
struct BitW {
  OrW orw;
  uint64_t acc;
  int nbits;
  uint32_t wpos;
  DFL_HD BitW(OrW o, uint64_t bit_offset) : orw(o), acc(0), nbits((int)(bit_offset & 31)), wpos((uint32_t)(bit_offset >> 5)) {}
  DFL_HD void put(uint32_t v, int n) {          // n <= 32 (v has no bits above n; n = 0 writes nothing)
    acc |= (uint64_t)v << nbits;
    nbits += n;
    if (nbits >= 32) { orw(wpos++, (uint32_t)acc); acc >>= 32; nbits -= 32; }
  }
  DFL_HD void flush() { if (nbits > 0) orw(wpos, (uint32_t)acc); }
  DFL_HD uint64_t bit_pos() const { return (uint64_t)wpos * 32 + nbits; }
};

// --------------------------------------------------------- block header --
// Run-length form of the code lengths (RFC 1951 3.2.7; the scan of zlib's trees.c scan_tree, literal and distance
// lengths scanned separately).  Emits (symbol, extra value) pairs through out(sym, extra_bits, extra_value).
template <class L, class Out>
DFL_HD void rle_lengths(L len, int n, Out out) {
  int i = 0;
  while (i < n) {
    const int v = len(i);
    int run = 1;
    while (i + run < n && len(i + run) == v) run++;
    i += run;
    if (v == 0) {
      while (run >= 11) { const int r = run > 138 ? 138 : run; out(18, 7, r - 11); run -= r; }
      if (run >= 3) { out(17, 3, run - 3); run = 0; }
      while (run-- > 0) out(0, 0, 0);
    } else {
      out(v, 0, 0); run--;
      while (run >= 3) { const int r = run > 6 ? 6 : run; out(16, 2, r - 3); run -= r; }
      while (run-- > 0) out(v, 0, 0);
    }
  }
}
DFL_HD constexpr int cl_order(int i) {
  constexpr int o[NCL] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  return o[i];
}

}  // namespace dfl
}  // namespace dctz
