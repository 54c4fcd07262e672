// dctz_deflate.hip -- GPU entropy stage: the three sections of a DCTZ container (bin_index, DC, AC_exact) deflated on
// the device into standard zlib streams, so that only compressed bytes cross PCIe and no host core runs deflate.
// SURVEY 8(f) rank 1; replaces the reference's zlib tail, dctz-comp-lib.c:620-732 (three threads, one single-shot
// deflate each); the output is what dctz-decomp-lib.c:244-322 inflates.  Format and method: deflate_chunk.h.
//
// k_deflate_chunks   one workgroup per CHUNK input bytes, one lane per 128-byte segment:
//                      load (LDS, padded so that lanes walking their segments hit different banks) + adler32 pieces
//                      -> parse (tokens, symbol counts) -> code lengths (rank sort in parallel, tree by one lane per
//                      alphabet) -> canonical codes -> bit counts + workgroup scan -> stored or dynamic -> emit
//                      -> slot in scratch (HBM), byte count per chunk.
// k_deflate_scan     byte offsets of the chunks inside the section (one workgroup).
// k_deflate_gather   slots -> one contiguous stream: 78 5E | chunks | 03 00 | adler32; section length to the host box.
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
constexpr int SLOT = CHUNK + 64;                       // bytes of scratch per chunk (a chunk never grows by more than 5)
constexpr int NSYM = NLIT + NDIST + NCL;               // the three alphabets side by side: [0,286) [286,316) [316,335)
constexpr uint32_t ADLER_M = 65521u;

__device__ __forceinline__ int pad(int i) { return i + ((i >> SEG_SHIFT) << 2); }   // 4 bytes of padding per segment

struct Lds {
  uint8_t in[HIST + CHUNK + ((HIST + CHUNK) >> SEG_SHIFT) * 4 + 16];
  uint8_t tok[CHUNK + (CHUNK >> SEG_SHIFT) * 4 + 16];
  uint32_t out[CHUNK / 4 + 8];
  uint32_t freq[NSYM + 1];
  uint16_t code[NSYM + 1];
  uint8_t len[NSYM + 1];
  uint16_t sorted[NLIT + NDIST + NCL + 1];
  uint32_t w[NLIT + NDIST];
  uint16_t ch[2 * (NLIT + NDIST)];
  uint16_t dep[NLIT + NDIST];
  uint16_t bl_count[3][16];
  uint16_t next_code[3][16];
  uint16_t cl[NLIT + NDIST + 4];                       // run-length form of the code lengths: sym | extra value << 8
  uint32_t scan[NTHR / 64 + 1];
  unsigned long long adler_b;
  uint32_t adler_a;
  int k[3];                                            // symbols in use per alphabet
  int hlit, hdist, hclen, ncl;
  uint32_t hbits;                                      // bits of the block header
  uint32_t tokbits;                                    // bits of all tokens
};

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(v, d, 64);
    if ((int)(threadIdx.x & 63) >= d) v += o;
  }
  return v;
}

// symbols of one alphabet with freq > 0, ascending (freq, symbol): every lane ranks the symbols it owns
__device__ __forceinline__ void rank_sort(Lds& s, int base, int n, int which) {
  for (int i = threadIdx.x; i < n; i += NTHR) {
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

__device__ __forceinline__ void build_lengths(Lds& s, int base, int which, int maxbits) {
  huff_lengths([&](int sym) { return s.freq[base + sym]; }, [&](int i) { return (int)s.sorted[base + i]; }, s.k[which], maxbits,
               [&](int sym, int bits) { s.len[base + sym] = (uint8_t)bits; }, s.w + (which == 1 ? NLIT : 0), s.ch + (which == 1 ? 2 * NLIT : 0),
               s.dep + (which == 1 ? NLIT : 0), s.bl_count[which]);
  first_codes(s.bl_count[which], maxbits, s.next_code[which]);
}

// canonical code of every symbol in use, bit-reversed (deflate sends Huffman codes most significant bit first)
__device__ __forceinline__ void assign_codes(Lds& s, int base, int n, int which) {
  for (int i = threadIdx.x; i < n; i += NTHR) {
    const int l = s.len[base + i];
    if (!l) continue;
    int before = 0;
    for (int j = 0; j < i; j++) before += (s.len[base + j] == l) ? 1 : 0;
    s.code[base + i] = (uint16_t)bit_reverse((uint32_t)s.next_code[which][l] + before, l);
  }
}

__global__ __launch_bounds__(NTHR) void k_deflate_chunks(const uint8_t* __restrict__ src, unsigned long long n, uint8_t* __restrict__ slots,
                                                         uint32_t* __restrict__ sizes, unsigned long long* __restrict__ adler_acc) {
  __shared__ Lds s;
  const int tid = threadIdx.x;
  const unsigned long long off = (unsigned long long)blockIdx.x * CHUNK;
  const int len = (int)((n - off) < (unsigned long long)CHUNK ? (n - off) : (unsigned long long)CHUNK);
  const int avail = 0;                                   // nothing in front of the chunk is referenced (deflate_chunk.h)
  uint8_t* slot = slots + (size_t)blockIdx.x * SLOT;

  // ---- load: history + chunk, dwords when the source allows it
  {
    const uint8_t* g = src + off - HIST;                 // logical index 0 of the padded image
    const int first = HIST - avail, last = HIST + len;   // valid logical range
    if ((((uintptr_t)src) & 3) == 0) {
      for (int i = tid * 4; i < HIST + CHUNK; i += NTHR * 4) {
        uint32_t v = 0;
        if (i >= first && i + 4 <= last) v = *(const uint32_t*)(g + i);
        else
          for (int b = 0; b < 4; b++) if (i + b >= first && i + b < last) v |= (uint32_t)g[i + b] << (8 * b);
        *(uint32_t*)&s.in[pad(i)] = v;
      }
    } else {
      for (int i = tid; i < HIST + CHUNK; i += NTHR) s.in[pad(i)] = (i >= first && i < last) ? g[i] : (uint8_t)0;
    }
  }
  for (int i = tid; i < NSYM + 1; i += NTHR) { s.freq[i] = 0; s.len[i] = 0; s.code[i] = 0; }
  for (int i = tid; i < CHUNK / 4 + 8; i += NTHR) s.out[i] = 0;
  if (tid == 0) { s.adler_a = 0; s.adler_b = 0; s.k[0] = s.k[1] = s.k[2] = 0; }
  __syncthreads();

  auto in = [&](int i) -> int { return s.in[pad(i + HIST)]; };
  const int p0 = tid * SEG, p1 = (p0 + SEG < len) ? p0 + SEG : len;

  // ---- adler32 pieces of this segment, then parse
  if (p0 < len) {
    uint32_t a = 0, b = 0;
    for (int p = p0; p < p1; p++) { a += (uint32_t)in(p); b += a; }
    atomicAdd(&s.adler_a, a);
    atomicAdd(&s.adler_b, (unsigned long long)b + (unsigned long long)a * (unsigned long long)(len - p1));
    parse_segment(in, [&](int p, int v) { s.tok[pad(p)] = (uint8_t)v; }, p0, p1, avail,
                  [&](int sym) { atomicAdd(&s.freq[sym], 1u); }, [&](int sym) { atomicAdd(&s.freq[NLIT + sym], 1u); });
  }
  __syncthreads();
  if (tid == 0) {
    s.freq[256] = 1;                                     // end of block
    // at least two distance codes in use, as zlib does it (trees.c build_tree): a complete code for any decoder
    int used = 0;
    for (int i = 0; i < NDIST; i++) used += s.freq[NLIT + i] ? 1 : 0;
    for (int i = 0; used < 2; i++) if (!s.freq[NLIT + i]) { s.freq[NLIT + i] = 1; used++; }
    // this chunk's share of the section's adler32: S1 = sum d, S2 = sum (n - i) d_i  (mod 65521)
    const unsigned long long after = n - (off + (unsigned long long)len);
    const unsigned long long a = s.adler_a, b = s.adler_b;
    atomicAdd(&adler_acc[0], a % ADLER_M);
    atomicAdd(&adler_acc[1], (b % ADLER_M + (a % ADLER_M) * (after % ADLER_M)) % ADLER_M);
  }
  __syncthreads();

  // ---- code lengths and codes of the literal/length and distance alphabets
  rank_sort(s, 0, NLIT, 0);
  rank_sort(s, NLIT, NDIST, 1);
  __syncthreads();
  if (tid == 0) build_lengths(s, 0, 0, MAXBITS);
  if (tid == (NTHR > 64 ? 64 : 1)) build_lengths(s, NLIT, 1, MAXBITS);
  __syncthreads();
  assign_codes(s, 0, NLIT, 0);
  assign_codes(s, NLIT, NDIST, 1);

  // ---- bits of this lane's tokens
  uint32_t mybits = 0;
  if (p0 < len) {
    for (int p = p0; p < p1;) {
      const int t = s.tok[pad(p)];
      if (t == 0) { mybits += s.len[in(p)]; p++; }
      else {
        const int l = s.tok[pad(p + 1)] + 3;
        int sym, eb, ev;
        len_code(l, sym, eb, ev);
        mybits += s.len[sym] + eb + s.len[NLIT + cand_dsym_rt(t - 1)] + cand_deb_rt(t - 1);
        p += l;
      }
    }
  }
  uint32_t incl = wave_incl_scan_u32(mybits);
  if ((tid & 63) == 63) s.scan[tid >> 6] = incl;

  // ---- header: run-length form of the lengths, its own Huffman code (one lane)
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
                 [&](int sym, int bits) { s.len[base + sym] = (uint8_t)bits; }, s.w, s.ch, s.dep, s.bl_count[2]);
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
    uint32_t hb = 3 + 5 + 5 + 4 + 3 * hclen;
    for (int i = 0; i < ncl; i++) {
      const int sym = s.cl[i] & 31;
      hb += s.len[base + sym] + (sym == 16 ? 2 : sym == 17 ? 3 : sym == 18 ? 7 : 0);
    }
    s.hlit = hlit; s.hdist = hdist; s.hclen = hclen; s.ncl = ncl; s.hbits = hb;
  }
  __syncthreads();
  uint32_t wave_base = 0, total = 0;
  for (int w = 0; w < NTHR / 64; w++) { if (w < (tid >> 6)) wave_base += s.scan[w]; total += s.scan[w]; }
  const uint32_t excl = wave_base + incl - mybits;
  const uint32_t body_bits = s.hbits + total + s.len[256];
  const uint32_t dyn_bytes = (body_bits + 3 + 7) / 8 + 4;       // + the empty stored block that ends on a byte boundary
  const uint32_t stored_bytes = (uint32_t)len + 5;

  if (dyn_bytes >= stored_bytes) {
    // ---- stored block: 00 | LEN | ~LEN | bytes
    if (tid == 0) {
      slot[0] = 0;
      slot[1] = (uint8_t)(len & 255); slot[2] = (uint8_t)(len >> 8);
      slot[3] = (uint8_t)(~len & 255); slot[4] = (uint8_t)((~len >> 8) & 255);
      sizes[blockIdx.x] = stored_bytes;
    }
    for (int i = tid; i < len; i += NTHR) slot[5 + i] = (uint8_t)in(i);
    return;
  }

  // ---- dynamic block
  auto orw = [&](uint32_t w, uint32_t v) { atomicOr(&s.out[w], v); };
  if (tid == 0) {
    BitW<decltype(orw)> bw(orw, 0);
    const int base = NLIT + NDIST;
    bw.put(0u | (2u << 1), 3);                           // BFINAL 0, BTYPE 10
    bw.put((uint32_t)(s.hlit - 257), 5);
    bw.put((uint32_t)(s.hdist - 1), 5);
    bw.put((uint32_t)(s.hclen - 4), 4);
    for (int i = 0; i < s.hclen; i++) bw.put(s.len[base + cl_order(i)], 3);
    for (int i = 0; i < s.ncl; i++) {
      const int sym = s.cl[i] & 31, ev = s.cl[i] >> 8;
      bw.put(s.code[base + sym], s.len[base + sym]);
      if (sym == 16) bw.put((uint32_t)ev, 2);
      else if (sym == 17) bw.put((uint32_t)ev, 3);
      else if (sym == 18) bw.put((uint32_t)ev, 7);
    }
    bw.flush();
  }
  if (p0 < len) {
    BitW<decltype(orw)> bw(orw, (uint64_t)s.hbits + excl);
    for (int p = p0; p < p1;) {
      const int t = s.tok[pad(p)];
      if (t == 0) { const int b = in(p); bw.put(s.code[b], s.len[b]); p++; }
      else {
        const int l = s.tok[pad(p + 1)] + 3;
        int sym, eb, ev;
        len_code(l, sym, eb, ev);
        bw.put(s.code[sym], s.len[sym]);
        if (eb) bw.put((uint32_t)ev, eb);
        const int c = t - 1, ds = cand_dsym_rt(c), de = cand_deb_rt(c);
        bw.put(s.code[NLIT + ds], s.len[NLIT + ds]);
        if (de) bw.put((uint32_t)cand_dev_rt(c), de);
        p += l;
      }
    }
    bw.flush();
  }
  const uint32_t end_byte = (body_bits + 3 + 7) / 8;     // where LEN = 0 of the empty stored block starts
  if (tid == 0) {
    BitW<decltype(orw)> bw(orw, (uint64_t)s.hbits + total);
    bw.put(s.code[256], s.len[256]);                     // end of block; the 3 header bits 000 and the padding are zeros already
    bw.flush();
    BitW<decltype(orw)> tail(orw, (uint64_t)(end_byte + 2) * 8);
    tail.put(0xFFFFu, 16);
    tail.flush();
    sizes[blockIdx.x] = dyn_bytes;
  }
  __syncthreads();
  for (int i = tid; i < (int)(dyn_bytes + 3) / 4; i += NTHR) ((uint32_t*)slot)[i] = s.out[i];
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

// slots -> the section's zlib stream; the first workgroup adds the frame
__global__ __launch_bounds__(256) void k_deflate_gather(const uint8_t* __restrict__ slots, const uint32_t* __restrict__ sizes,
                                                        const unsigned long long* __restrict__ offs, uint32_t nchunks, unsigned long long n,
                                                        const unsigned long long* __restrict__ adler_acc, uint8_t* __restrict__ dst,
                                                        unsigned long long* __restrict__ box_len) {
  const unsigned long long total = offs[nchunks];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    dst[0] = 0x78; dst[1] = 0x5E;                        // FLG: "fastest algorithm" hint, no preset dictionary -- and this library's mark for
                                                         // "the deflate blocks are independent chunks" (include/dctz.h: DCTZ_IX_MAGIC)
    uint8_t* t = dst + 2 + total;
    t[0] = 0x03; t[1] = 0x00;
    const uint32_t s1 = (uint32_t)((1 + adler_acc[0]) % ADLER_M), s2 = (uint32_t)((n % ADLER_M + adler_acc[1]) % ADLER_M);
    t[2] = (uint8_t)(s2 >> 8); t[3] = (uint8_t)s2; t[4] = (uint8_t)(s1 >> 8); t[5] = (uint8_t)s1;
    *box_len = total + 8;
  }
  for (uint32_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const uint8_t* sl = slots + (size_t)c * SLOT;
    uint8_t* d = dst + 2 + offs[c];
    const uint32_t sz = sizes[c];
    for (uint32_t i = threadIdx.x; i < sz; i += 256) d[i] = sl[i];
  }
}

}  // namespace dfl

// ------------------------------------------------------------------ host side --
size_t deflate_chunk_bytes() { return (size_t)dfl::CHUNK; }
size_t deflate_slot_bytes() { return (size_t)dfl::SLOT; }

// One section.  scratch: slots (nchunks * SLOT) | sizes (u32 x nchunks, 8-byte aligned) | offs (u64 x (nchunks + 1)) | adler (2 x u64);
// layout computed by deflate_scratch_bytes().  box_len: where the section's stream length goes (host-visible).
size_t deflate_scratch_bytes(size_t n) {
  const size_t nch = (n + dfl::CHUNK - 1) / dfl::CHUNK;
  return nch * dfl::SLOT + ((nch * 4 + 7) & ~(size_t)7) + (nch + 1) * 8 + 16 + 64;
}
size_t deflate_bound(size_t n) {
  const size_t nch = (n + dfl::CHUNK - 1) / dfl::CHUNK;
  return n + 5 * nch + 8;
}

hipError_t launch_deflate(const void* src, size_t n, void* dst, void* scratch, unsigned long long* box_len, uint32_t* host_sizes, hipStream_t st) {
  const size_t nch = (n + dfl::CHUNK - 1) / dfl::CHUNK;
  uint8_t* slots = (uint8_t*)scratch;
  uint32_t* sizes = (uint32_t*)(slots + nch * dfl::SLOT);
  unsigned long long* offs = (unsigned long long*)((uint8_t*)sizes + ((nch * 4 + 7) & ~(size_t)7));
  unsigned long long* adler = offs + nch + 1;
  hipError_t e = hipMemsetAsync(adler, 0, 16, st);
  if (e != hipSuccess) return e;
  if (nch) hipLaunchKernelGGL(dfl::k_deflate_chunks, dim3((unsigned)nch), dim3(dfl::NTHR), 0, st, (const uint8_t*)src, (unsigned long long)n, slots, sizes, adler);
  if (nch && host_sizes) {                             // the chunk index of the section (bytes per chunk), for whoever writes a container
    e = hipMemcpyAsync(host_sizes, sizes, nch * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(dfl::k_deflate_scan, dim3(1), dim3(1024), 0, st, sizes, (uint32_t)nch, offs);
  const unsigned g = nch ? (unsigned)(nch < 4096 ? nch : 4096) : 1u;
  hipLaunchKernelGGL(dfl::k_deflate_gather, dim3(g), dim3(256), 0, st, slots, sizes, offs, (uint32_t)nch, (unsigned long long)n, adler, (uint8_t*)dst, box_len);
  return hipGetLastError();
}

}  // namespace dctz
