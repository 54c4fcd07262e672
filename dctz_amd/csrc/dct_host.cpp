// dct_host.cpp -- the per-block entry points of dct.h (dct_fftw / ifft_idct, dct.h:17-27) on HOST buffers.
//
// The reference's callers loop these per 64-element block (dct-test.c:81-89, :145-152), on arrays in host memory, and read
// the result of every call at once.  Round 3 sent every such call through the GPU: an H2D copy, a launch and a D2H copy
// per 512 bytes -- about 50 us per block against the reference's ~1 us.  One block is no work for a
// GPU; what it needs is the same ARITHMETIC.  So this file compiles the product's own lane flow -- dct64_block.h, the code
// one GPU lane executes, with the product's host-built tables (dctz_tables.h) -- for the host: operation for operation
// what k_dct_blocks / k_dct_rem compute (fused multiply-adds exactly where the header spells them, everything else one
// rounding per operation: built with -ffp-contract=off), so a block transformed here is bit for bit the block
// dctz_dct_blocks() returns from the device (tests/test_libdctz_gpu.py compares the two).  The BATCHED entry points
// (dctz_dct_blocks, dctz_compress, ...) stay on the GPU; nothing here is a fall-back for them.
#include <cstring>

#include "dct64_block.h"
#include "dctz_tables.h"

using namespace dctz;

namespace {

template <typename T>
const T* block_table() {
  static T tab[TBP_TOTAL];
  static bool ready = false;
  if (!ready) { fill_tab_block<T>(tab); ready = true; }      // (callers are single-threaded, like the reference's dct.c:18-22)
  return tab;
}
template <typename T>
const T* rem_table(int l) {
  static T tab[RTAB_SIZE];
  static int have = -1;
  if (have != l) { fill_rem_tab<T>(l, tab); have = l; }
  return tab;
}

// a full block: the lane flow of dct64_block.h
template <typename T, bool INVERSE>
inline void full_block(const T* a, T* b) {
  T x[64];
  for (int i = 0; i < 64; i++) x[i] = a[i];
  if (INVERSE) dct64_inv<T, const T*>(x, block_table<T>()); else dct64_fwd<T, const T*>(x, block_table<T>());
  for (int i = 0; i < 64; i++) b[i] = x[i];
}
// the same with the CPU's fused multiply-add instruction where it has one (else fma() of libm: the same result, slower)
#if defined(__x86_64__) && defined(__GNUC__)
template <typename T, bool INVERSE>
__attribute__((target("fma"))) void full_block_fma(const T* a, T* b) { full_block<T, INVERSE>(a, b); }
#endif
template <typename T, bool INVERSE>
void full_block_any(const T* a, T* b) {
#if defined(__x86_64__) && defined(__GNUC__)
  static const bool has_fma = __builtin_cpu_supports("fma");
  if (has_fma) { full_block_fma<T, INVERSE>(a, b); return; }
#endif
  full_block<T, INVERSE>(a, b);
}

// a short block (length l < 64): the definition-order DFT of k_dct_rem (dctz_kernels_aux.hip), lane k = output k
// (dct.c:59-72 / :144-199: the reference re-plans a length-l or 2l FFT)
template <typename T, bool INVERSE>
void short_block(const T* x, T* out, int l) {
  const T* rt = rem_table<T>(l);
  const int N = (l & 1) ? 2 * l : l;
  T v[128], w[128];
  for (int i = 0; i < 128; i++) { v[i] = T(0); w[i] = T(0); }
  if (!INVERSE) {
    for (int k = 0; k < l; k++) {
      const T a = x[k];
      if (l & 1) { v[k] = a; v[l + (l - 1 - k)] = a; }            // dct.c:61-64
      else if (k & 1) v[l - 1 - (k >> 1)] = a;                    // dct.c:75-83
      else v[k >> 1] = a;
    }
    for (int k = 0; k < l; k++) {
      T sr = T(0), si = T(0);
      for (int j = 0; j < N; j++) {
        const int tt = (j * k) % N;
        sr = sr + v[j] * rt[RTAB_WR + tt];
        si = si + v[j] * rt[RTAB_WI + tt];
      }
      out[k] = rt[RTAB_AS + k] * sr + rt[RTAB_AX + k] * si;       // dct.c:100-102
    }
  } else {
    for (int k = 0; k < l; k++) {
      v[k] = rt[RTAB_IAS + k] * x[k];                             // dct.c:146-151 / :166-172
      w[k] = rt[RTAB_IAX + k] * x[k];
      if ((l & 1) && k >= 1) {                                    // dct.c:152-153
        v[l + k] = rt[RTAB_IAX + k] * x[l - k];
        w[l + k] = -(rt[RTAB_IAS + k] * x[l - k]);
      }
    }
    T res[64];
    for (int k = 0; k < l; k++) {
      const int s = (l & 1) ? k : ((k & 1) ? l - 1 - (k >> 1) : (k >> 1));   // dct.c:189-199
      T acc = T(0);
      for (int j = 0; j < N; j++) {
        const int tt = (s * j) % N;
        acc = acc + (v[j] * rt[RTAB_WR + tt] - w[j] * rt[RTAB_WI + tt]);
      }
      res[k] = (l & 1) ? (acc / (T)l) / T(2) : acc / (T)l;        // dct.c:163 / :185
    }
    for (int k = 0; k < l; k++) out[k] = res[k];
  }
}

template <typename T>
void one(const T* a, T* b, int dn, int inverse) {
  T tmp[64];
  const T* src = a;
  if (a == b) { std::memcpy(tmp, a, sizeof(T) * (size_t)dn); src = tmp; }   // (in place is allowed, as in the reference)
  if (dn == 64) { if (inverse) full_block_any<T, true>(src, b); else full_block_any<T, false>(src, b); }
  else if (inverse) short_block<T, true>(src, b, dn);
  else short_block<T, false>(src, b, dn);
}

}  // namespace

extern "C" {
// dn in 1 .. 64 (checked by the caller, libdctz.c: one_block)
void dctz_host_block_f64(const double* a, double* b, int dn, int inverse) { one<double>(a, b, dn, inverse); }
void dctz_host_block_f32(const float* a, float* b, int dn, int inverse) { one<float>(a, b, dn, inverse); }
}
