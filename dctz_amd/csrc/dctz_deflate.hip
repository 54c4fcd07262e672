// dctz_deflate.hip -- GPU entropy stage: the three sections of a DCTZ container (bin_index, DC, AC_exact) deflated on
// the device into standard zlib streams, so that only compressed bytes cross PCIe and no host core runs deflate.
// SURVEY 8(f) rank 1; replaces the reference's zlib tail, dctz-comp-lib.c:620-732 (three threads, one single-shot
// deflate each); the output is what dctz-decomp-lib.c:244-322 inflates.  Format and method: deflate_chunk.h.
//
// Three kernels per section, so that the work that only one lane can do (building a Huffman tree) never holds the
// LDS image of a chunk hostage:
// k_dfl_parse   one workgroup per CHUNK input bytes, one lane per 128-byte segment: load (LDS, padded so that lanes
//               walking their segments hit different banks), adler32 pieces, tokens + symbol counts -> HBM.
// k_dfl_codes   one WAVE per chunk, 7 KiB of LDS (a CU keeps ~20 chunks in flight, which is what hides the serial
//               stretches): symbol counts -> code lengths (rank sort in parallel, two-queue tree by one lane) ->
//               canonical codes -> block header bits -> stored or dynamic, bytes of the chunk.
// k_deflate_scan  byte offsets of the chunks inside the section (one workgroup).
// k_dfl_emit    one workgroup per chunk: tokens + codes -> bits (bit counts per lane, workgroup scan, every lane writes
//               at its own bit offset) -> the chunk's place in the stream; the first workgroup adds the frame
//               78 5E ... 03 00 adler32 and hands the section length to the host box.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "deflate_chunk.h"

#ifndef DCTZ_DFL_THREADS
#define DCTZ_DFL_THREADS 128                           // 16 KiB chunks; the workgroup's LDS image stays under 64 KiB
#endif

namespace dctz {
namespace dfl {

constexpr int NTHR = DCTZ_DFL_THREADS;               // lanes per workgroup = segments per chunk
constexpr int CHUNK = NTHR * SEG;
constexpr int NSYM = NLIT + NDIST + NCL;               // the three alphabets side by side: [0,286) [286,316) [316,335)
constexpr uint32_t ADLER_M = 65521u;

__device__ __forceinline__ int pad(int i) { return i + ((i >> SEG_SHIFT) << 2); }   // 4 bytes of padding per segment
__device__ __forceinline__ int pad16(int i) { return i + ((i >> SEG_SHIFT) << 1); } // the same for 16-bit entries: lanes at the same offset of their segments hit different banks

constexpr int META_SYMS = 320;                         // NLIT + NDIST rounded up
constexpr int HDR_WORDS = 160;                         // >= (17 + 3 * 19 + 316 * 14) / 32 + 1
struct ChunkMeta {                                     // what k_dfl_codes hands to k_dfl_emit
  uint16_t code[META_SYMS];                            // bit-reversed canonical codes: [0, 286) literal/length, [286, 316) distance
  uint8_t len[META_SYMS];
  uint32_t hdr[HDR_WORDS];                             // the block header, bit 0 first
  uint32_t hbits;                                      // its length in bits
  uint32_t dyn;                                        // 1: dynamic block, 0: stored
  uint32_t pad[2];
};

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(v, d, 64);
    if ((int)(threadIdx.x & 63) >= d) v += o;
  }
  return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// ------------------------------------------------------------------ parse --
struct LdsParse {
  uint8_t in[CHUNK + (CHUNK >> SEG_SHIFT) * 4 + 16];
  // per position: during the parse its entry of parse_segment_dwords (length and candidate of the best match from there), afterwards -- low
  // byte -- the token record (deflate_chunk.h: tok)
  uint16_t tok[CHUNK + (CHUNK >> SEG_SHIFT) * 2 + 8];
  uint32_t freq[META_SYMS];
  unsigned long long adler_b;
  uint32_t adler_a;
  // first and last token of every segment, for the merge across segment boundaries (deflate_chunk.h)
  int16_t first_c[NTHR], last_c[NTHR];                 // candidate index, -1: a literal (or no token)
  uint16_t first_len[NTHR], last_len[NTHR], last_p[NTHR], ntok[NTHR];
};

// chunk bytes -> padded LDS image; dwords when the source allows it
__device__ __forceinline__ void load_chunk(uint8_t* lds, const uint8_t* g, int len, bool aligned) {
  if (aligned) {
    for (int i = threadIdx.x * 4; i < CHUNK; i += NTHR * 4) {
      uint32_t v = 0;
      if (i + 4 <= len) v = *(const uint32_t*)(g + i);
      else
        for (int b = 0; b < 4; b++) if (i + b < len) v |= (uint32_t)g[i + b] << (8 * b);
      *(uint32_t*)&lds[pad(i)] = v;
    }
  } else {
    for (int i = threadIdx.x; i < CHUNK; i += NTHR) lds[pad(i)] = i < len ? g[i] : (uint8_t)0;
  }
}

// MATCH = false: the section is known not to repeat itself (float bytes: DC, AC_exact -- zlib finds next to nothing
// there either): every byte is a literal, the kernel is a byte histogram.
template <bool MATCH>
__global__ __launch_bounds__(NTHR) void k_dfl_parse(const uint8_t* __restrict__ src, unsigned long long n, uint8_t* __restrict__ tok_g,
                                                    uint32_t* __restrict__ freq_g, unsigned long long* __restrict__ adler_acc) {
  __shared__ LdsParse s;
  const int tid = threadIdx.x;
  const unsigned long long off = (unsigned long long)blockIdx.x * CHUNK;
  const int len = (int)((n - off) < (unsigned long long)CHUNK ? (n - off) : (unsigned long long)CHUNK);
  load_chunk(s.in, src + off, len, (((uintptr_t)src) & 3) == 0);
  for (int i = tid; i < META_SYMS; i += NTHR) s.freq[i] = 0;
  if (tid == 0) { s.adler_a = 0; s.adler_b = 0; }
  __syncthreads();

  auto in = [&](int i) -> int { return s.in[pad(i)]; };
  const int p0 = tid * SEG, p1 = (p0 + SEG < len) ? p0 + SEG : len;
  auto in4 = [&](int i) -> uint32_t { return *(const uint32_t*)&s.in[pad(i)]; };     // i a multiple of four
  if (p0 < len) {
    uint32_t a = 0, b = 0;                                // adler32 pieces of this segment: sum d, sum (seglen - j) d_j
    if (p1 - p0 == SEG) {
      for (int w = p0; w < p1; w += 4) {
        const uint32_t x = in4(w);
        a += x & 255u; b += a; a += (x >> 8) & 255u; b += a; a += (x >> 16) & 255u; b += a; a += x >> 24; b += a;
      }
    } else
      for (int p = p0; p < p1; p++) { a += (uint32_t)in(p); b += a; }
    atomicAdd(&s.adler_a, a);
    atomicAdd(&s.adler_b, (unsigned long long)b + (unsigned long long)a * (unsigned long long)(len - p1));
    if (MATCH) {
      int fc = -1, fl = 0, lc = -1, ll = 0, lp = p0, nt = 0;
      parse_segment_dwords(in, in4,
                           [&](int i, uint32_t lo, uint32_t hi) { uint32_t* t = (uint32_t*)&s.tok[pad16(i)]; t[0] = lo; t[1] = hi; },
                           [&](int p) -> uint32_t { return s.tok[pad16(p)]; },
                           [&](int p, int v) { s.tok[pad16(p)] = (uint16_t)v; }, p0, p1,
                           [&](int sym) { atomicAdd(&s.freq[sym], 1u); }, [&](int sym) { atomicAdd(&s.freq[NLIT + sym], 1u); },
                           [&](int p, int l, int c) { if (nt == 0) { fc = c; fl = l; } lc = c; ll = l; lp = p; nt++; });
      s.first_c[tid] = (int16_t)fc; s.first_len[tid] = (uint16_t)fl; s.last_c[tid] = (int16_t)lc; s.last_len[tid] = (uint16_t)ll;
      s.last_p[tid] = (uint16_t)lp; s.ntok[tid] = (uint16_t)nt;
    } else
      for (int p = p0; p < p1; p++) { s.tok[pad16(p)] = 0; atomicAdd(&s.freq[in(p)], 1u); }
  } else if (MATCH) { s.first_c[tid] = -1; s.last_c[tid] = -1; s.ntok[tid] = 0; s.first_len[tid] = 0; s.last_len[tid] = 0; s.last_p[tid] = 0; }
  __syncthreads();
  if (MATCH && p0 < len) {
    // merge with the next segment: this lane's last token takes that segment's first one in (same distance, contiguous),
    // decided from the two tokens as the parse left them; the counts follow (two length symbols and a distance out,
    // one length symbol in)
    if (tid + 1 < NTHR && p0 + SEG < len) {
      const int lc = s.last_c[tid], ll = s.last_len[tid], lp = s.last_p[tid];
      const int fc = s.first_c[tid + 1], fl = s.first_len[tid + 1];
      if (lc >= 0 && lp + ll == p0 + SEG && fc == lc && ll + fl <= MAXMATCH && merge_allowed(tid, s.ntok[tid] == 1, s.ntok[tid + 1] == 1)) {
        s.tok[pad16(lp + 1)] = (uint16_t)(ll + fl - 3);
        int sa, sb, sc, eb, ev;
        len_code(ll, sa, eb, ev); len_code(fl, sb, eb, ev); len_code(ll + fl, sc, eb, ev);
        atomicSub(&s.freq[sa], 1u); atomicSub(&s.freq[sb], 1u); atomicAdd(&s.freq[sc], 1u);
        atomicSub(&s.freq[NLIT + cand_dsym_rt(lc)], 1u);
      }
    }
    if (tid > 0) {                                        // the mirror image: is this segment's first token taken in by the previous one?
      const int lc = s.last_c[tid - 1], ll = s.last_len[tid - 1], lp = s.last_p[tid - 1];
      const int fc = s.first_c[tid], fl = s.first_len[tid];
      if (lc >= 0 && lp + ll == p0 && fc == lc && ll + fl <= MAXMATCH && merge_allowed(tid - 1, s.ntok[tid - 1] == 1, s.ntok[tid] == 1))
        s.tok[pad16(p0)] = (uint16_t)TOK_ABSORBED;
    }
  }
  __syncthreads();
  for (int i = tid * 4; i < CHUNK; i += NTHR * 4) {       // the low bytes of four entries = four token bytes (scratch is a whole number of chunks)
    const uint32_t* t = (const uint32_t*)&s.tok[pad16(i)];
    const uint32_t d0 = t[0], d1 = t[1];
    *(uint32_t*)(tok_g + off + i) = (d0 & 255u) | ((d0 >> 8) & 0xFF00u) | ((d1 & 255u) << 16) | ((d1 << 8) & 0xFF000000u);
  }
  for (int i = tid; i < META_SYMS; i += NTHR) freq_g[(size_t)blockIdx.x * META_SYMS + i] = s.freq[i];
  if (tid == 0) {
    // this chunk's share of the section's adler32: S1 = sum d, S2 = sum (n - i) d_i  (mod 65521)
    const unsigned long long after = n - (off + (unsigned long long)len);
    const unsigned long long a = s.adler_a, b = s.adler_b;
    atomicAdd(&adler_acc[0], a % ADLER_M);
    atomicAdd(&adler_acc[1], (b % ADLER_M + (a % ADLER_M) * (after % ADLER_M)) % ADLER_M);
  }
}

// ------------------------------------------------------------------ codes --
constexpr int CW = 64;                                 // one wave per chunk
struct LdsCodes {
  uint32_t freq[NSYM + 1];
  uint16_t code[NSYM + 1];
  uint8_t len[NSYM + 1];
  uint16_t sorted[NSYM + 1];
  uint32_t w[NLIT + NDIST];
  uint16_t ch[2 * (NDIST + NCL) + 8];                  // children / depths of the two small trees (distance, code lengths), built one after the other
  uint16_t dep[NDIST + NCL + 4];
  uint16_t bl_count[3][16];
  uint16_t next_code[3][16];
  uint16_t cl[NLIT + NDIST + 4];                       // run-length form of the code lengths: sym | extra value << 8
  uint32_t hdr[HDR_WORDS];
  uint32_t lw[NLIT];                                   // literal/length tree: leaf weights in sorted order
  uint16_t par[NLIT];                                  //   parent of every internal node (pointer jumping: ancestor)
  uint16_t pl[NLIT];                                   //   parent of every leaf
  uint16_t up[NLIT];                                   //   distance to par[]
  uint32_t blc[16];                                    //   leaves per depth
  uint16_t used[NLIT];                                 // literal/length symbols in use, ascending (k[0] of them)
  int k[3];                                            // symbols in use per alphabet
  int hlit, hdist, hclen, ncl;
  uint32_t hbits;
  uint32_t forced;                                     // distance symbols counted only to complete the code
};

// symbols of one alphabet with freq > 0, ascending (freq, symbol): every lane ranks the symbols it owns
__device__ __forceinline__ void rank_sort(LdsCodes& s, int base, int n, int which) {
  for (int i = threadIdx.x; i < n; i += CW) {
    const uint32_t f = s.freq[base + i];
    if (!f) continue;
    int r = 0;
    for (int j = 0; j < n; j++) {
      const uint32_t g = s.freq[base + j];
      r += (g != 0 && (g < f || (g == f && j < i))) ? 1 : 0;
    }
    s.sorted[base + r] = (uint16_t)i;
    atomicAdd(&s.k[which], 1);
  }
}
// The same for the literal/length alphabet, through the list of symbols in use: a bin_index chunk uses a few dozen of the
// 286 symbols, and both this ranking and the numbering of equal-length codes are quadratic in what they walk over.
__device__ __forceinline__ void rank_sort_used(LdsCodes& s) {
  const int lane = threadIdx.x;
  int k = 0;
  for (int r = 0; r < (NLIT + CW - 1) / CW; r++) {       // compaction, in symbol order
    const int i = r * CW + lane;
    const bool on = i < NLIT && s.freq[i] != 0;
    const unsigned long long m = __ballot(on);
    if (on) s.used[k + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)i;
    k += __popcll(m);
  }
  if (lane == 0) s.k[0] = k;
  __syncthreads();
  for (int a = lane; a < k; a += CW) {
    const int i = s.used[a];
    const uint32_t f = s.freq[i];
    int r = 0;
    for (int b = 0; b < k; b++) {
      const int j = s.used[b];
      const uint32_t g = s.freq[j];
      r += (g < f || (g == f && j < i)) ? 1 : 0;
    }
    s.sorted[r] = (uint16_t)i;
  }
}
__device__ __forceinline__ void assign_codes_used(LdsCodes& s) {
  const int k = s.k[0];
  for (int a = threadIdx.x; a < k; a += CW) {
    const int i = s.used[a], l = s.len[i];
    int before = 0;
    for (int b = 0; b < a; b++) before += (s.len[s.used[b]] == l) ? 1 : 0;
    s.code[i] = (uint16_t)bit_reverse((uint32_t)s.next_code[0][l] + before, l);
  }
}
__device__ __forceinline__ void build_lengths(LdsCodes& s, int base, int which, int maxbits) {
  huff_lengths([&](int sym) { return s.freq[base + sym]; }, [&](int i) { return (int)s.sorted[base + i]; }, s.k[which], maxbits,
               [&](int sym, int bits) { s.len[base + sym] = (uint8_t)bits; }, s.w + (which == 1 ? NLIT : 0), s.ch, s.dep, s.bl_count[which]);
  first_codes(s.bl_count[which], maxbits, s.next_code[which]);
}
// canonical code of every symbol in use, bit-reversed (deflate sends Huffman codes most significant bit first)
__device__ __forceinline__ void assign_codes(LdsCodes& s, int base, int n, int which) {
  for (int i = threadIdx.x; i < n; i += CW) {
    const int l = s.len[base + i];
    if (!l) continue;
    int before = 0;
    for (int j = 0; j < i; j++) before += (s.len[base + j] == l) ? 1 : 0;
    s.code[base + i] = (uint16_t)bit_reverse((uint32_t)s.next_code[which][l] + before, l);
  }
}
// The literal/length tree by the whole wave.  Same tree as huff_lengths() (same tie rule, hence the same leaf count per
// depth and the same lengths): the two-queue merge stays with one lane, but reads only leaf weights laid out in sorted
// order and node weights, four values fetched together per step; depths come from pointer jumping over the parent
// links (log2 rounds, all lanes), leaf counts per depth from LDS atomics, lengths from the rank of the leaf.
__device__ __forceinline__ void build_lengths_wave(LdsCodes& s, int maxbits) {
  const int tid = threadIdx.x, k = s.k[0];
  for (int r = tid; r < k; r += CW) s.lw[r] = s.freq[s.sorted[r]];
  if (tid < 16) s.blc[tid] = 0;
  __syncthreads();
  if (tid == 0) {
    const uint32_t INF = 0xFFFFFFFFu;
    int li = 0, ii = 0;
    for (int ni = 0; ni < k - 1; ni++) {
      uint32_t a0 = li < k ? s.lw[li] : INF, a1 = li + 1 < k ? s.lw[li + 1] : INF;
      uint32_t b0 = ii < ni ? s.w[ii] : INF, b1 = ii + 1 < ni ? s.w[ii + 1] : INF;
      uint32_t wsum = 0;
#pragma unroll
      for (int t = 0; t < 2; t++) {
        if (a0 <= b0 && a0 != INF) { wsum += a0; s.pl[li] = (uint16_t)ni; li++; a0 = a1; }
        else { wsum += b0; s.par[ii] = (uint16_t)ni; ii++; b0 = b1; }
      }
      s.w[ni] = wsum;
    }
    s.par[k - 2] = (uint16_t)(k - 2);                    // the root
  }
  __syncthreads();
  const int nint = k - 1;
  for (int i = tid; i < nint; i += CW) s.up[i] = (i == nint - 1) ? 0 : 1;
  __syncthreads();
  for (int span = 1; span < nint; span <<= 1) {          // after the round, up[i] = min(depth, 2 * span) hops towards the root
    uint16_t np[(NLIT + CW - 1) / CW], nu[(NLIT + CW - 1) / CW];
    int c = 0;
    for (int i = tid; i < nint; i += CW, c++) { const int p = s.par[i]; nu[c] = (uint16_t)(s.up[i] + s.up[p]); np[c] = s.par[p]; }
    __syncthreads();
    c = 0;
    for (int i = tid; i < nint; i += CW, c++) { s.up[i] = nu[c]; s.par[i] = np[c]; }
    __syncthreads();
  }
  for (int r = tid; r < k; r += CW) {
    const int d = s.up[s.pl[r]] + 1;
    atomicAdd(&s.blc[d > maxbits ? maxbits : d], 1u);
  }
  __syncthreads();
  if (tid == 0) {
    // leaves clamped to maxbits over-subscribe the code: every repair step takes 2^-maxbits off the Kraft sum (deflate_chunk.h)
    uint32_t kraft = 0;
    for (int b = 1; b <= maxbits; b++) kraft += s.blc[b] << (maxbits - b);
    for (uint32_t over = kraft - (1u << maxbits); over > 0; over--) {
      int bits = maxbits - 1;
      while (s.blc[bits] == 0) bits--;
      s.blc[bits]--;
      s.blc[bits + 1] += 2;
      s.blc[maxbits]--;
    }
    for (int b = 0; b <= MAXBITS; b++) s.bl_count[0][b] = (uint16_t)(b <= maxbits ? s.blc[b] : 0);
    s.bl_count[0][0] = 0;
    first_codes(s.bl_count[0], maxbits, s.next_code[0]);
  }
  __syncthreads();
  for (int r = tid; r < k; r += CW) {                    // the longest codes go to the rarest symbols
    int bits = maxbits, upto = (int)s.blc[maxbits];
    while (r >= upto) { bits--; upto += (int)s.blc[bits]; }
    s.len[s.sorted[r]] = (uint8_t)bits;
  }
}

__device__ __forceinline__ int len_extra_bits(int sym) { return (sym < 265 || sym == 285) ? 0 : (sym - 261) >> 2; }
__device__ __forceinline__ int dist_extra_bits(int sym) { return sym < 4 ? 0 : (sym >> 1) - 1; }

__global__ __launch_bounds__(CW) void k_dfl_codes(const uint32_t* __restrict__ freq_g, unsigned long long n, ChunkMeta* __restrict__ meta,
                                                  uint32_t* __restrict__ sizes) {
  __shared__ LdsCodes s;
  const int tid = threadIdx.x;
  const unsigned long long off = (unsigned long long)blockIdx.x * CHUNK;
  const int len = (int)((n - off) < (unsigned long long)CHUNK ? (n - off) : (unsigned long long)CHUNK);
  for (int i = tid; i < NSYM + 1; i += CW) { s.freq[i] = i < NLIT + NDIST ? freq_g[(size_t)blockIdx.x * META_SYMS + i] : 0u; s.len[i] = 0; s.code[i] = 0; }
  for (int i = tid; i < HDR_WORDS; i += CW) s.hdr[i] = 0;
  if (tid == 0) { s.k[0] = s.k[1] = s.k[2] = 0; s.forced = 0; }
  __syncthreads();
  if (tid == 0) {
    s.freq[256] = 1;                                     // end of block
    // at least two distance codes in use, as zlib does it (trees.c build_tree): a complete code for any decoder
    int used = 0;
    uint32_t forced = 0;
    for (int i = 0; i < NDIST; i++) used += s.freq[NLIT + i] ? 1 : 0;
    for (int i = 0; used < 2; i++) if (!s.freq[NLIT + i]) { s.freq[NLIT + i] = 1; forced |= 1u << i; used++; }
    s.forced = forced;
  }
  __syncthreads();
  rank_sort_used(s);
  rank_sort(s, NLIT, NDIST, 1);
  __syncthreads();
  build_lengths_wave(s, MAXBITS);
  if (tid == 0) build_lengths(s, NLIT, 1, MAXBITS);
  __syncthreads();
  assign_codes_used(s);
  assign_codes(s, NLIT, NDIST, 1);
  // bits of all tokens, from the counts: code length + extra bits per use (end of block included)
  uint32_t bits = 0;
  for (int i = tid; i < NLIT; i += CW) bits += s.freq[i] * (uint32_t)(s.len[i] + (i > 256 ? len_extra_bits(i) : 0));
  for (int i = tid; i < NDIST; i += CW) if (!((s.forced >> i) & 1u)) bits += s.freq[NLIT + i] * (uint32_t)(s.len[NLIT + i] + dist_extra_bits(i));
  const uint32_t tokbits = wave_sum_u32(bits);
  __syncthreads();

  // header: run-length form of the lengths, its own Huffman code, the bits (one lane)
  if (tid == 0) {
    int hlit = NLIT, hdist = NDIST;
    while (hlit > 257 && s.len[hlit - 1] == 0) hlit--;
    while (hdist > 1 && s.len[NLIT + hdist - 1] == 0) hdist--;
    int ncl = 0;
    auto outcl = [&](int sym, int eb, int ev) { (void)eb; s.cl[ncl++] = (uint16_t)(sym | (ev << 8)); s.freq[NLIT + NDIST + sym]++; };
    rle_lengths([&](int i) { return (int)s.len[i]; }, hlit, outcl);
    rle_lengths([&](int i) { return (int)s.len[NLIT + i]; }, hdist, outcl);
    const int base = NLIT + NDIST;
    int k = 0;
    for (int i = 0; i < NCL; i++) if (s.freq[base + i]) k++;
    for (int i = 0; k < 2; i++) if (!s.freq[base + i]) { s.freq[base + i] = 1; k++; }
    k = 0;
    for (int i = 0; i < NCL; i++) {                      // insertion sort, ascending (freq, symbol)
      const uint32_t f = s.freq[base + i];
      if (!f) continue;
      int j = k++;
      while (j > 0 && s.freq[base + s.sorted[base + j - 1]] > f) { s.sorted[base + j] = s.sorted[base + j - 1]; j--; }
      s.sorted[base + j] = (uint16_t)i;
    }
    s.k[2] = k;
    huff_lengths([&](int sym) { return s.freq[base + sym]; }, [&](int i) { return (int)s.sorted[base + i]; }, k, MAXBITS_CL,
                 [&](int sym, int b) { s.len[base + sym] = (uint8_t)b; }, s.w, s.ch, s.dep, s.bl_count[2]);
    first_codes(s.bl_count[2], MAXBITS_CL, s.next_code[2]);
    for (int i = 0; i < NCL; i++) {
      const int l = s.len[base + i];
      if (!l) continue;
      int before = 0;
      for (int j = 0; j < i; j++) before += (s.len[base + j] == l) ? 1 : 0;
      s.code[base + i] = (uint16_t)bit_reverse((uint32_t)s.next_code[2][l] + before, l);
    }
    int hclen = NCL;
    while (hclen > 4 && s.len[base + cl_order(hclen - 1)] == 0) hclen--;
    auto orw = [&](uint32_t w, uint32_t v) { s.hdr[w] |= v; };
    BitW<decltype(orw)> bw(orw, 0);
    bw.put(0u | (2u << 1), 3);                           // BFINAL 0, BTYPE 10
    bw.put((uint32_t)(hlit - 257), 5);
    bw.put((uint32_t)(hdist - 1), 5);
    bw.put((uint32_t)(hclen - 4), 4);
    for (int i = 0; i < hclen; i++) bw.put(s.len[base + cl_order(i)], 3);
    bw.flush();
    s.hlit = hlit; s.hdist = hdist; s.hclen = hclen; s.ncl = ncl;
  }
  __syncthreads();
  {
    // the coded lengths: every lane takes a run of entries, bit offsets from a wave scan
    const int base = NLIT + NDIST, ncl = s.ncl, per = (ncl + CW - 1) / CW;
    const int e0 = tid * per, e1 = (e0 + per < ncl) ? e0 + per : ncl;
    auto ebits = [&](int sym) { return (int)s.len[base + sym] + (sym == 16 ? 2 : sym == 17 ? 3 : sym == 18 ? 7 : 0); };
    uint32_t mine = 0;
    for (int i = e0; i < e1; i++) mine += (uint32_t)ebits(s.cl[i] & 31);
    const uint32_t incl = wave_incl_scan_u32(mine);
    const uint32_t fixed = 17u + 3u * (uint32_t)s.hclen;
    auto orw = [&](uint32_t w, uint32_t v) { atomicOr(&s.hdr[w], v); };
    BitW<decltype(orw)> bw(orw, (uint64_t)fixed + incl - mine);
    for (int i = e0; i < e1; i++) {
      const int sym = s.cl[i] & 31, ev = s.cl[i] >> 8;
      bw.put(s.code[base + sym], s.len[base + sym]);
      if (sym == 16) bw.put((uint32_t)ev, 2);
      else if (sym == 17) bw.put((uint32_t)ev, 3);
      else if (sym == 18) bw.put((uint32_t)ev, 7);
    }
    bw.flush();
    if (tid == CW - 1) s.hbits = fixed + incl;
  }
  __syncthreads();
  const uint32_t body_bits = s.hbits + tokbits;
  const uint32_t dyn_bytes = (body_bits + 3 + 7) / 8 + 4;       // + the empty stored block that ends on a byte boundary
  const uint32_t stored_bytes = (uint32_t)len + 5;
  const bool dyn = dyn_bytes < stored_bytes;
  ChunkMeta* m = meta + blockIdx.x;
  for (int i = tid; i < META_SYMS; i += CW) { m->code[i] = i < NLIT + NDIST ? s.code[i] : (uint16_t)0; m->len[i] = i < NLIT + NDIST ? s.len[i] : (uint8_t)0; }
  for (int i = tid; i < HDR_WORDS; i += CW) m->hdr[i] = s.hdr[i];
  if (tid == 0) { m->hbits = s.hbits; m->dyn = dyn ? 1u : 0u; sizes[blockIdx.x] = dyn ? dyn_bytes : stored_bytes; }
}

// ------------------------------------------------------------------- emit --
struct LdsEmit {
  uint8_t in[CHUNK + (CHUNK >> SEG_SHIFT) * 4 + 16];
  uint8_t tok[CHUNK + (CHUNK >> SEG_SHIFT) * 4 + 16];
  uint32_t out[CHUNK / 4 + 8];
  uint32_t cl[META_SYMS];                              // code | length << 16: one look-up per symbol
  uint32_t scan[NTHR / 64 + 1];
};

__global__ __launch_bounds__(NTHR) void k_dfl_emit(const uint8_t* __restrict__ src, unsigned long long n, const uint8_t* __restrict__ tok_g,
                                                   const ChunkMeta* __restrict__ meta, const uint32_t* __restrict__ sizes,
                                                   const unsigned long long* __restrict__ offs, uint32_t nchunks,
                                                   const unsigned long long* __restrict__ adler_acc, uint8_t* __restrict__ dst,
                                                   unsigned long long* __restrict__ box_len) {
  __shared__ LdsEmit s;
  const int tid = threadIdx.x;
  if (blockIdx.x == 0 && tid == 0) {                      // the frame of the section
    const unsigned long long total = offs[nchunks];
    dst[0] = 0x78; dst[1] = 0x5E;                        // FLG: "fastest algorithm" hint, no preset dictionary -- and this library's mark for
                                                         // "the deflate blocks are independent chunks" (include/dctz.h: DCTZ_IX_MAGIC)
    uint8_t* t = dst + 2 + total;
    t[0] = 0x03; t[1] = 0x00;
    const uint32_t s1 = (uint32_t)((1 + adler_acc[0]) % ADLER_M), s2 = (uint32_t)((n % ADLER_M + adler_acc[1]) % ADLER_M);
    t[2] = (uint8_t)(s2 >> 8); t[3] = (uint8_t)s2; t[4] = (uint8_t)(s1 >> 8); t[5] = (uint8_t)s1;
    *box_len = total + 8;
  }
  if (blockIdx.x >= nchunks) return;
  const unsigned long long off = (unsigned long long)blockIdx.x * CHUNK;
  const int len = (int)((n - off) < (unsigned long long)CHUNK ? (n - off) : (unsigned long long)CHUNK);
  const ChunkMeta* m = meta + blockIdx.x;
  uint8_t* d = dst + 2 + offs[blockIdx.x];
  if (!m->dyn) {
    // ---- stored block: 00 | LEN | ~LEN | bytes
    if (tid == 0) {
      d[0] = 0;
      d[1] = (uint8_t)(len & 255); d[2] = (uint8_t)(len >> 8);
      d[3] = (uint8_t)(~len & 255); d[4] = (uint8_t)((~len >> 8) & 255);
    }
    for (int i = tid; i < len; i += NTHR) d[5 + i] = src[off + i];
    return;
  }
  load_chunk(s.in, src + off, len, (((uintptr_t)src) & 3) == 0);
  load_chunk(s.tok, tok_g + off, CHUNK, true);
  for (int i = tid; i < META_SYMS; i += NTHR) s.cl[i] = (uint32_t)m->code[i] | ((uint32_t)m->len[i] << 16);
  const uint32_t hbits = m->hbits, hwords = (hbits + 31) / 32;
  for (int i = tid; i < CHUNK / 4 + 8; i += NTHR) s.out[i] = i < (int)hwords ? m->hdr[i] : 0u;
  __syncthreads();

  auto in = [&](int i) -> int { return s.in[pad(i)]; };
  const int p0 = tid * SEG, p1 = (p0 + SEG < len) ? p0 + SEG : len;
  uint32_t mybits = 0;
  // (a segment whose first token was merged into the previous segment's last one starts behind it: deflate_chunk.h)
  const int pstart = (p0 < len && s.tok[pad(p0)] == TOK_ABSORBED) ? p0 + s.tok[pad(p0 + 1)] + 3 : p0;
  // (a token's three bytes -- record, length, literal -- are asked for together, then its code(s): two round trips through
  // LDS per token instead of three or four; the walk is a chain of them)
  // Literals and matches go through ONE sequence of instructions (a literal is a token without length-extra, distance and
  // distance-extra bits): with 64 lanes at 64 different places of their token streams nearly every trip of a two-branch
  // loop ran both branches.
  auto token = [&](int p, uint32_t& e1, int& eb, int& ev, uint32_t& e2, int& de, int& dv) -> int {
    const int t = s.tok[pad(p)], t1 = s.tok[pad(p + 1)], b = in(p);
    const bool m = t != 0;
    const int l = m ? t1 + 3 : 3;
    int sym;
    len_code(l, sym, eb, ev);
    const int c = m ? t - 1 : 0;
    e1 = s.cl[m ? sym : b];
    e2 = m ? s.cl[NLIT + cand_dsym_rt(c)] : 0u;
    eb = m ? eb : 0;
    de = m ? cand_deb_rt(c) : 0;
    dv = m ? cand_dev_rt(c) : 0;
    return m ? l : 1;
  };
  if (p0 < len) {
    for (int p = pstart; p < p1;) {
      uint32_t e1, e2;
      int eb, ev, de, dv;
      const int adv = token(p, e1, eb, ev, e2, de, dv);
      mybits += (e1 >> 16) + (uint32_t)eb + (e2 >> 16) + (uint32_t)de;
      p += adv;
    }
  }
  const uint32_t incl = wave_incl_scan_u32(mybits);
  if ((tid & 63) == 63) s.scan[tid >> 6] = incl;
  __syncthreads();
  uint32_t wave_base = 0, total = 0;
  for (int w = 0; w < NTHR / 64; w++) { if (w < (tid >> 6)) wave_base += s.scan[w]; total += s.scan[w]; }
  const uint32_t excl = wave_base + incl - mybits;
  const uint32_t body_bits = hbits + total + (s.cl[256] >> 16);
  const uint32_t dyn_bytes = (body_bits + 3 + 7) / 8 + 4;       // == sizes[chunk]

  auto orw = [&](uint32_t w, uint32_t v) { atomicOr(&s.out[w], v); };
  if (p0 < len) {
    BitW<decltype(orw)> bw(orw, (uint64_t)hbits + excl);
    for (int p = pstart; p < p1;) {
      uint32_t e1, e2;
      int eb, ev, de, dv;
      const int adv = token(p, e1, eb, ev, e2, de, dv);
      const int n1 = (int)(e1 >> 16), n2 = (int)(e2 >> 16);
      // code + its extra bits (<= 15 + 5), distance code + its extra bits (<= 15 + 13; nothing for a literal)
      bw.put((e1 & 0xFFFFu) | ((uint32_t)ev << n1), n1 + eb);
      bw.put((e2 & 0xFFFFu) | ((uint32_t)dv << n2), n2 + de);
      p += adv;
    }
    bw.flush();
  }
  const uint32_t end_byte = (body_bits + 3 + 7) / 8;     // where LEN = 0 of the empty stored block starts
  if (tid == 0) {
    BitW<decltype(orw)> bw(orw, (uint64_t)hbits + total);
    bw.put(s.cl[256] & 0xFFFFu, (int)(s.cl[256] >> 16));   // end of block; the 3 header bits 000 and the padding are zeros already
    bw.flush();
    BitW<decltype(orw)> tail(orw, (uint64_t)(end_byte + 2) * 8);
    tail.put(0xFFFFu, 16);
    tail.flush();
  }
  __syncthreads();
  // the chunk's bytes to its place in the stream: whole dwords of the destination (the bytes in front of the first and
  // behind the last one singly -- neighbouring chunks own the rest of those dwords)
  const uint8_t* ob = (const uint8_t*)s.out;
  uint32_t head = (uint32_t)((4 - ((uintptr_t)d & 3)) & 3);
  if (head > dyn_bytes) head = dyn_bytes;
  const uint32_t nd = (dyn_bytes - head) / 4;
  for (uint32_t i = tid; i < head; i += NTHR) d[i] = ob[i];
  uint32_t* dw = (uint32_t*)(d + head);
  const uint32_t sh = (head & 3) * 8;
  for (uint32_t k = tid; k < nd; k += NTHR) {
    const uint32_t w0 = s.out[(head >> 2) + k], w1 = s.out[(head >> 2) + k + 1];
    dw[k] = sh ? ((w0 >> sh) | (w1 << (32 - sh))) : w0;
  }
  for (uint32_t i = head + 4 * nd + tid; i < dyn_bytes; i += NTHR) d[i] = ob[i];
}

// byte offsets of the chunks (exclusive scan of their sizes); one workgroup
__global__ __launch_bounds__(1024) void k_deflate_scan(const uint32_t* __restrict__ sizes, uint32_t nchunks, unsigned long long* __restrict__ offs) {
  __shared__ unsigned long long part[1024];
  const uint32_t per = (nchunks + 1023) / 1024;
  const uint32_t lo = threadIdx.x * per, hi = (lo + per < nchunks) ? lo + per : nchunks;
  unsigned long long sum = 0;
  for (uint32_t i = lo; i < hi; i++) sum += sizes[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long run = 0;
    for (int i = 0; i < 1024; i++) { const unsigned long long v = part[i]; part[i] = run; run += v; }
    offs[nchunks] = run;
  }
  __syncthreads();
  unsigned long long run = part[threadIdx.x];
  for (uint32_t i = lo; i < hi; i++) { offs[i] = run; run += sizes[i]; }
}

}  // namespace dfl

// ------------------------------------------------------------------ host side --
size_t deflate_chunk_bytes() { return (size_t)dfl::CHUNK; }

// scratch of one section: tokens (whole chunks) | counts | ChunkMeta | sizes | offsets | adler
struct DflScratch {
  uint8_t* tok;
  uint32_t* freq;
  dfl::ChunkMeta* meta;
  uint32_t* sizes;
  unsigned long long* offs;
  unsigned long long* adler;
  size_t bytes;
};
static DflScratch carve(void* base, size_t n) {
  const size_t nch = (n + dfl::CHUNK - 1) / dfl::CHUNK;
  DflScratch r;
  uint8_t* p = (uint8_t*)base;
  r.tok = p; p += nch * dfl::CHUNK;
  r.freq = (uint32_t*)p; p += nch * dfl::META_SYMS * sizeof(uint32_t);
  r.meta = (dfl::ChunkMeta*)p; p += nch * sizeof(dfl::ChunkMeta);
  r.sizes = (uint32_t*)p; p += (nch * 4 + 15) & ~(size_t)15;
  r.offs = (unsigned long long*)p; p += (nch + 1) * 8;
  r.adler = (unsigned long long*)p; p += 16;
  r.bytes = (size_t)(p - (uint8_t*)base) + 64;
  return r;
}
size_t deflate_scratch_bytes(size_t n) { return carve(nullptr, n).bytes; }
size_t deflate_bound(size_t n) {
  const size_t nch = (n + dfl::CHUNK - 1) / dfl::CHUNK;
  return n + 5 * nch + 8;
}

hipError_t launch_deflate(const void* src, size_t n, void* dst, void* scratch, unsigned long long* box_len, uint32_t* host_sizes, bool literals_only,
                          hipStream_t st) {
  static_assert(sizeof(dfl::ChunkMeta) % 16 == 0, "ChunkMeta is laid out in an array");
  const size_t nch = (n + dfl::CHUNK - 1) / dfl::CHUNK;
  const DflScratch sc = carve(scratch, n);
  hipError_t e = hipMemsetAsync(sc.adler, 0, 16, st);
  if (e != hipSuccess) return e;
  if (nch) {
    if (literals_only)
      hipLaunchKernelGGL(dfl::k_dfl_parse<false>, dim3((unsigned)nch), dim3(dfl::NTHR), 0, st, (const uint8_t*)src, (unsigned long long)n, sc.tok, sc.freq, sc.adler);
    else
      hipLaunchKernelGGL(dfl::k_dfl_parse<true>, dim3((unsigned)nch), dim3(dfl::NTHR), 0, st, (const uint8_t*)src, (unsigned long long)n, sc.tok, sc.freq, sc.adler);
    hipLaunchKernelGGL(dfl::k_dfl_codes, dim3((unsigned)nch), dim3(dfl::CW), 0, st, sc.freq, (unsigned long long)n, sc.meta, sc.sizes);
    if (host_sizes) {                                  // the chunk index of the section (bytes per chunk), for whoever writes a container
      e = hipMemcpyAsync(host_sizes, sc.sizes, nch * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
      if (e != hipSuccess) return e;
    }
  }
  hipLaunchKernelGGL(dfl::k_deflate_scan, dim3(1), dim3(1024), 0, st, sc.sizes, (uint32_t)nch, sc.offs);
  hipLaunchKernelGGL(dfl::k_dfl_emit, dim3(nch ? (unsigned)nch : 1u), dim3(dfl::NTHR), 0, st, (const uint8_t*)src, (unsigned long long)n, sc.tok, sc.meta,
                     sc.sizes, sc.offs, (uint32_t)nch, sc.adler, (uint8_t*)dst, box_len);
  return hipGetLastError();
}

}  // namespace dctz

// =====================================================================================================================
// Inflate of sections made by the kernels above (and only of those): every chunk is one stored or one dynamic block
// that starts on a byte boundary, references at most 128 bytes back and never anything in front of the chunk, and the
// container's "DZIX" index says where it starts.  One LANE per chunk: 64 chunks per wave decode side by side; a
// section of 8192 chunks is 128 waves.  Replaces, for such sections, the reader's inflate() calls
// (dctz-decomp-lib.c:244-322) and the H2D copy of the raw streams.
//
// A damaged stream must never turn into a wild access: every read of the chunk is bounded by its size, every write by
// the chunk's output length, every table index by construction of the canonical tables, every loop by a counter that
// moves towards its end; what does not fit sets the section's error flag and the lane stops (the host then takes the
// zlib path, which reports the damage the way the reference does).
namespace dctz {
namespace dfl {

constexpr int IW = 64;                                 // lanes (= chunks) per workgroup
constexpr int RING = 128;                              // bytes of history per lane >= the largest distance the encoder uses
static_assert(RING >= cand_maxdist() && (RING & (RING - 1)) == 0, "the decoder's ring covers every candidate distance");

struct LdsInflate {                                    // per-lane arrays, element i of lane t at [i][t]
  union {
    uint8_t len[NLIT + NDIST][IW];                     // code lengths as read from the header (until the tables stand)
    uint8_t ring[RING][IW];                            // then: the last RING bytes written
  };
  uint16_t lsym[NLIT][IW];                             // literal/length symbols sorted by (length, symbol)
  uint8_t dsym[NDIST][IW];
  uint8_t csym[NCL][IW];
};

// LSB-first bit reader over the chunk's bytes.  Memory is read as aligned 32-bit words, one word AHEAD of the one being
// consumed (the load issued when a word is taken is not needed before the next 32 bits are: its latency hides behind
// the decoding of the symbols in between); words that lie entirely behind the chunk are not read, the few bytes of a
// word that straddle its ends are whatever the section holds there (never used: `over` is set when more bits are asked
// for than the chunk has).
struct BitR {
  const uint32_t* w;                                   // aligned word that holds the chunk's first byte
  uint32_t nwords;                                     // words that hold bytes of the chunk
  uint32_t wi;                                         // next word to load
  uint32_t nextw;                                      // the word loaded ahead
  uint64_t acc;
  int nbits;
  long long left;                                      // bits of the chunk not yet handed out
  bool over;
  __device__ __forceinline__ BitR(const uint8_t* z, uint32_t zlen) {
    const uintptr_t p = (uintptr_t)z;
    const uint32_t skip = (uint32_t)(p & 3);
    w = (const uint32_t*)(p - skip);
    nwords = (skip + zlen + 3) / 4;
    left = (long long)zlen * 8;
    over = false;
    const uint32_t w0 = nwords > 0 ? w[0] : 0u;
    nextw = nwords > 1 ? w[1] : 0u;
    wi = 2;
    acc = (uint64_t)(w0 >> (8 * skip));
    nbits = 32 - 8 * (int)skip;
  }
  __device__ __forceinline__ void need(int n) {        // n <= 32
    if (nbits < n) {
      acc |= (uint64_t)nextw << nbits;
      nbits += 32;
      nextw = wi < nwords ? w[wi] : 0u;
      wi++;
    }
  }
  __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)(acc & ((1ull << n) - 1)); }
  __device__ __forceinline__ void drop(int n) {
    left -= n;
    if (left < 0) over = true;
    acc >>= n; nbits -= n;
  }
  __device__ __forceinline__ uint32_t get(int n) {
    need(n);
    const uint32_t v = peek(n);
    drop(n);
    return v;
  }
  __device__ __forceinline__ void align_byte() { const int r = (int)(left & 7); if (r) drop(r); }   // (the chunk starts on a byte boundary)
};

// canonical decode (the counting walk of zlib's contrib/puff): cnt[l] = codes of length l, symbols sorted by (length, symbol)
template <int MAXL, class Sym>
__device__ __forceinline__ int decode_sym(BitR& br, const uint16_t (&cnt)[MAXL + 1], Sym sym, bool& bad) {
  br.need(MAXL);
  uint32_t bits = br.peek(MAXL);
  int code = 0, first = 0, index = 0;
#pragma unroll
  for (int l = 1; l <= MAXL; l++) {
    code |= (int)(bits & 1u);
    bits >>= 1;
    const int c = cnt[l];
    if (code - c < first) { br.drop(l); return sym(index + (code - first)); }
    index += c; first += c;
    first <<= 1; code <<= 1;
  }
  bad = true;
  return 0;
}

__global__ __launch_bounds__(IW) void k_dfl_inflate(const uint8_t* __restrict__ sec, const uint32_t* __restrict__ offs, uint32_t nchunks,
                                                    unsigned long long n, uint8_t* __restrict__ dst, unsigned long long* __restrict__ adler_acc,
                                                    uint32_t* __restrict__ status) {
  __shared__ LdsInflate s;
  const int t = threadIdx.x;
  const uint32_t c = blockIdx.x * IW + t;
  if (c >= nchunks) return;
  const unsigned long long off = (unsigned long long)c * CHUNK;
  const uint32_t olen = (uint32_t)((n - off) < (unsigned long long)CHUNK ? (n - off) : (unsigned long long)CHUNK);
  uint8_t* out = dst + off;
  BitR br(sec + 2 + offs[c], offs[c + 1] - offs[c]);
  bool bad = false;
  uint32_t op = 0, a = 0;                                 // output position; adler32 pieces of the chunk: a = sum d,
  unsigned long long b = 0;                               // b = sum (olen - j) d_j (accumulated as b += a after every byte)

  const uint32_t hdr = br.get(3);
  if (hdr == 0) {                                         // stored, not final: LEN, ~LEN, bytes
    br.align_byte();
    const uint32_t len = br.get(16), nlen = br.get(16);
    if (len != olen || (len ^ nlen) != 0xFFFFu || br.over) bad = true;
    else {
      // the bit reader may hold bytes already: they come first
      for (; op < olen; op++) { const uint32_t d = br.get(8); out[op] = (uint8_t)d; a += d; b += a; }
      if (br.over) bad = true;
    }
  } else if (hdr == 4) {                                  // BFINAL 0, BTYPE 10
    const int hlit = 257 + (int)br.get(5), hdist = 1 + (int)br.get(5), hclen = 4 + (int)br.get(4);
    if (hlit > NLIT || hdist > NDIST) bad = true;
    uint16_t ccnt[MAXBITS_CL + 1] = {}, lcnt[MAXBITS + 1] = {}, dcnt[MAXBITS + 1] = {};
    if (!bad) {
      // code-length code
      uint8_t cl[NCL];
#pragma unroll
      for (int i = 0; i < NCL; i++) cl[i] = 0;
#pragma unroll
      for (int i = 0; i < NCL; i++) if (i < hclen) { const uint32_t v = br.get(3); cl[cl_order(i)] = (uint8_t)v; }
#pragma unroll
      for (int i = 0; i < NCL; i++) ccnt[cl[i]]++;
      ccnt[0] = 0;
      {
        int offs_l[MAXBITS_CL + 2];
        offs_l[1] = 0;
#pragma unroll
        for (int l = 1; l <= MAXBITS_CL; l++) offs_l[l + 1] = offs_l[l] + ccnt[l];
#pragma unroll
        for (int i = 0; i < NCL; i++) if (cl[i]) { s.csym[offs_l[cl[i]] < NCL ? offs_l[cl[i]] : 0][t] = (uint8_t)i; offs_l[cl[i]]++; }
      }
      // the literal/length and distance code lengths
      int i = 0, prev = 0;
      const int total = hlit + hdist;
      while (i < total && !bad && !br.over) {
        const int sym = decode_sym<MAXBITS_CL>(br, ccnt, [&](int k) { return (int)s.csym[k < NCL ? k : 0][t]; }, bad);
        if (bad) break;
        if (sym < 16) { s.len[i++][t] = (uint8_t)sym; prev = sym; }
        else {
          int rep, val;
          if (sym == 16) { if (i == 0) { bad = true; break; } rep = 3 + (int)br.get(2); val = prev; }
          else if (sym == 17) { rep = 3 + (int)br.get(3); val = 0; prev = 0; }
          else { rep = 11 + (int)br.get(7); val = 0; prev = 0; }
          if (i + rep > total) { bad = true; break; }
          for (int r = 0; r < rep; r++) s.len[i++][t] = (uint8_t)val;
        }
      }
      if (br.over) bad = true;
    }
    if (!bad) {
      // counts per length, then the symbols in canonical order
      for (int i = 0; i < hlit; i++) { const int l = s.len[i][t]; if (l) lcnt[l]++; }
      for (int i = 0; i < hdist; i++) { const int l = s.len[hlit + i][t]; if (l) dcnt[l]++; }
      int lo[MAXBITS + 2], dofs[MAXBITS + 2];
      lo[1] = 0; dofs[1] = 0;
#pragma unroll
      for (int l = 1; l <= MAXBITS; l++) { lo[l + 1] = lo[l] + lcnt[l]; dofs[l + 1] = dofs[l] + dcnt[l]; }
      for (int i = 0; i < hlit; i++) {
        const int l = s.len[i][t];
        if (!l) continue;
        int slot = 0;
#pragma unroll
        for (int q = 1; q <= MAXBITS; q++) if (q == l) { slot = lo[q]; lo[q]++; }
        s.lsym[slot < NLIT ? slot : 0][t] = (uint16_t)i;
      }
      for (int i = 0; i < hdist; i++) {
        const int l = s.len[hlit + i][t];
        if (!l) continue;
        int slot = 0;
#pragma unroll
        for (int q = 1; q <= MAXBITS; q++) if (q == l) { slot = dofs[q]; dofs[q]++; }
        s.dsym[slot < NDIST ? slot : 0][t] = (uint8_t)i;
      }
      // the tokens.  One byte per lane per turn: a lane inside a long match copies one byte while its neighbours decode
      // their next symbol (a loop that finished a whole match per turn would make every lane of the wave wait for the
      // longest match of the turn: 128 copy steps per literal of a neighbour).
      int mlen = 0, mdist = 0;
      bool done = false;
      for (uint32_t guard = 0; guard < 2 * olen + 2 && !bad && !done; guard++) {   // a turn writes a byte, opens a match or ends the block
        if (mlen > 0) {
          const uint32_t d = s.ring[(op - (uint32_t)mdist) & (RING - 1)][t];
          out[op] = (uint8_t)d; s.ring[op & (RING - 1)][t] = (uint8_t)d;
          a += d; b += a; op++; mlen--;
          continue;
        }
        const int sym = decode_sym<MAXBITS>(br, lcnt, [&](int k) { return (int)s.lsym[k < NLIT ? k : 0][t]; }, bad);
        if (bad || br.over) { bad = true; break; }
        if (sym < 256) {
          if (op >= olen) { bad = true; break; }
          out[op] = (uint8_t)sym; s.ring[op & (RING - 1)][t] = (uint8_t)sym;
          a += (uint32_t)sym; b += a; op++;
        } else if (sym == 256) done = true;
        else {
          const int ls = sym - 257;
          if (ls >= 29) { bad = true; break; }
          int len;
          if (ls < 8) len = 3 + ls;
          else if (ls == 28) len = 258;
          else { const int eb = (ls >> 2) - 1; len = 3 + ((4 + (ls & 3)) << eb) + (int)br.get(eb); }
          const int ds = decode_sym<MAXBITS>(br, dcnt, [&](int k) { return (int)s.dsym[k < NDIST ? k : 0][t]; }, bad);
          if (bad) break;
          int dist;
          if (ds < 4) dist = 1 + ds;
          else { const int eb = (ds >> 1) - 1; dist = 1 + ((2 + (ds & 1)) << eb) + (int)br.get(eb); }
          if (dist > RING || (uint32_t)dist > op || op + (uint32_t)len > olen || br.over) { bad = true; break; }
          mlen = len; mdist = dist;
        }
      }
      if (!done) bad = true;
    }
  } else bad = true;
  if (!bad && op != olen) bad = true;
  if (bad) { atomicOr(status, 1u); return; }
  // this chunk's share of the section's adler32 (as in k_dfl_parse)
  const unsigned long long after = n - (off + olen);
  atomicAdd(&adler_acc[0], (unsigned long long)(a % ADLER_M));
  atomicAdd(&adler_acc[1], (unsigned long long)((b % ADLER_M + (unsigned long long)(a % ADLER_M) * (after % ADLER_M)) % ADLER_M));
}

}  // namespace dfl

// One section: compressed stream `sec` (device, zlen bytes), chunk offsets offs[nch + 1] (device, relative to sec + 2),
// raw length n -> dst.  status: one word, OR-ed with 1 on any inconsistency; adler: two u64 accumulators (zeroed here).
hipError_t launch_inflate(const void* sec, const uint32_t* offs, size_t nch, size_t n, void* dst, unsigned long long* adler, uint32_t* status,
                          hipStream_t st) {
  hipError_t e = hipMemsetAsync(adler, 0, 16, st);
  if (e != hipSuccess) return e;
  if (nch) hipLaunchKernelGGL(dfl::k_dfl_inflate, dim3((unsigned)((nch + dfl::IW - 1) / dfl::IW)), dim3(dfl::IW), 0, st, (const uint8_t*)sec, offs,
                              (uint32_t)nch, (unsigned long long)n, (uint8_t*)dst, adler, status);
  return hipGetLastError();
}

}  // namespace dctz
