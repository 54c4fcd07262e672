// dctz_kernels_aux.hip -- the kernels AROUND the two big ones: calc_data_stat as a pass of its own (and the sample of the
// speculative path) with the final reduction that also chooses the scaling factor on the device, the serial-order sum for
// the header's mean, the in-place scaling of the caller's copy, the batched transform entry points (dct.h), the gather /
// scatter passes of ragged multi-dimensional shapes, calc_psnr's reductions, and a division self-test.
// Reference code replaced: see include/dctz_hip.h (per entry point) and the comment on each kernel.
#include "dctz_kernel_common.h"

namespace dctz {

// =============================================================== statistics ==
// calc_data_stat (util.c:12-44): max|x|, min|x| and sum (x[0] is never added,
// util.c:22 starts at i = 1).  Tree order: `sum` is NOT the reference's serial
// order (it is never used by the codec; the host wrapper recomputes it
// serially for the header).
template <typename T>
__device__ __forceinline__ void stats_body(const T* __restrict__ x, size_t n, double* __restrict__ part, const unsigned wg, const unsigned nwg) {
  using Vec = typename Traits<T>::Vec;
  constexpr int EPV = Traits<T>::EPV;
  const size_t nvec = n / EPV;
  const Vec* src = reinterpret_cast<const Vec*>(x);
  T mx = T(0), mn = Traits<T>::huge();
  double sum = 0.0;
  constexpr int UN = 4;                            // each workgroup streams 16 KiB contiguous per trip
  for (size_t i0 = (size_t)wg * SWG * UN + threadIdx.x; i0 < nvec; i0 += (size_t)nwg * SWG * UN) {
    Vec v[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const size_t i = i0 + (size_t)u * SWG;
      v[u] = load_stream((i < nvec) ? &src[i] : &src[i0]);   // a repeated vector changes neither max nor min
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const size_t i = i0 + (size_t)u * SWG;
      T e[EPV];
      Traits<T>::unpack(v[u], e);
#pragma unroll
      for (int k = 0; k < EPV; k++) {
        const T a = fabs(e[k]);
        mx = a > mx ? a : mx;
        mn = a < mn ? a : mn;
        if (i < nvec && (i != 0 || k != 0)) sum += (double)e[k];
      }
    }
  }
  if (wg == 0 && threadIdx.x == 0)
    for (size_t i = nvec * EPV; i < n; i++) {
      const T a = fabs(x[i]);
      mx = a > mx ? a : mx;
      mn = a < mn ? a : mn;
      if (i != 0) sum += (double)x[i];
    }
  double dmx = (double)mx, dmn = (double)mn;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    dmx = fmax(dmx, __shfl_down(dmx, d));
    dmn = fmin(dmn, __shfl_down(dmn, d));
    sum += __shfl_down(sum, d);
  }
  __shared__ double s[3][SWG / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s[0][wave] = dmx; s[1][wave] = dmn; s[2][wave] = sum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < SWG / 64; w++) { dmx = fmax(dmx, s[0][w]); dmn = fmin(dmn, s[1][w]); sum += s[2][w]; }
    part[3 * wg + 0] = dmx;
    part[3 * wg + 1] = dmn;
    part[3 * wg + 2] = sum;
  }
}
template <typename T>
__global__ __launch_bounds__(SWG) void k_stats(const T* __restrict__ x, size_t n, double* __restrict__ part) {
  stats_body<T>(x, n, part, blockIdx.x, gridDim.x);
}

// final reduction of the statistics partials -> device words + (optionally) the host mailbox; one workgroup
__device__ __forceinline__ void stats_final_body(const double* part, int nparts, double* out, HostBox* box, unsigned long long seq) {
  double dmx, dmn, sum;
  reduce_parts(part, nparts, dmx, dmn, sum);
  if (threadIdx.x == 0) {
    out[0] = dmx; out[1] = dmn; out[2] = sum;
    if (box != nullptr) {                            // hand the three numbers straight to the polling host thread
      box->stats[0] = dmx; box->stats[1] = dmn; box->stats[2] = sum;
      box_publish(&box->seq_stats, seq);
    }
  }
}

// zero != NULL: also the first kernel-side act of a compress call -- the control block back to all-zero (the
// per-position maxima of the QT table accumulate with atomicMax, :371-372)
__global__ __launch_bounds__(SWG) void k_stats_final(const double* part, int nparts, double* out,
                                                    HostBox* box, unsigned long long seq, Ctl* zero) {
  if (zero != nullptr) {
    unsigned long long* w = reinterpret_cast<unsigned long long*>(zero);
    for (int i = threadIdx.x; i < (int)(sizeof(Ctl) / 8); i += SWG) w[i] = 0ull;
  }
  stats_final_body(part, nparts, out, box, seq);
}

// The same, for a speculative compress call: the scaling factor of util.c:29 / :43 for the SAMPLED max|x| is chosen
// here, with the host's own decade tables (SfTable), and left in device memory for k_compress -- no host round trip
// between the sample and the main launch (the host verifies the choice against the true statistics afterwards).
__device__ __forceinline__ void stats_final_sf_body(const double* part, int nparts, double* out, Ctl* zero, const SfTable& tab,
                                                    SfGuess* guess, HostBox* box) {
  if (zero != nullptr) {
    unsigned long long* w = reinterpret_cast<unsigned long long*>(zero);
    for (int i = threadIdx.x; i < (int)(sizeof(Ctl) / 8); i += SWG) w[i] = 0ull;
  }
  __shared__ double smax, smin;
  __shared__ unsigned cnt_s[SWG / 64];
  double dmx, dmn, sum;
  reduce_parts(part, nparts, dmx, dmn, sum);
  if (threadIdx.x == 0) {
    out[0] = dmx; out[1] = dmn; out[2] = sum; smax = dmx; smin = dmn;
    // (no sequence number here: the host reads these after the call's hand-off, which is a later kernel's)
    if (box != nullptr) { box->stats[0] = dmx; box->stats[1] = dmn; box->stats[2] = sum; }
  }
  __syncthreads();
  const double mx = smax, mn = smin;
  unsigned below = 0;                                  // decades whose upper end lies below max|x|
  for (int i = threadIdx.x; i < tab.nk; i += SWG) below += (tab.thr[i] < mx) ? 1u : 0u;
  const unsigned k = block_sum(below, cnt_s);
  if (threadIdx.x == 0) {
    double sf = (mx == 0.0) ? 1.0 : tab.pw[k < (unsigned)tab.nk ? k : (unsigned)tab.nk];   // all-zero input: sf = 1
    const bool f64 = tab.dtype == DCTZHIP_F64;
    unsigned fast = (tab.fastdiv && (f64 ? exp_in(sf, -250, 250) : exp_in(sf, -30, 30))) ? 1u : 0u;
    if (fast && tab.fastdiv >= 2 && (f64 ? (exp_in(mn, -500, 500) && exp_in(mx, -500, 500)) : (exp_in(mn, -63, 63) && exp_in(mx, -63, 63)))) fast = 2u;
    guess->sf = sf;
    guess->fast_sf = fast;
  }
}
__global__ __launch_bounds__(SWG) void k_stats_final_sf(const double* part, int nparts, double* out, Ctl* zero, SfTable tab,
                                                       SfGuess* guess, HostBox* box) {
  stats_final_sf_body(part, nparts, out, zero, tab, guess, box);
}

// Sampled statistics for the speculative path: one 4 KiB chunk out of every group of
// `group` chunks, at a hashed position inside the group (a fixed stride would alias with
// the row structure of power-of-two volumes).  Same partials layout as k_stats.
template <typename T>
__device__ __forceinline__ void stats_sample_body(const T* __restrict__ x, size_t n, unsigned group, double* __restrict__ part,
                                                  const unsigned wg, const unsigned nwg) {
  using Vec = typename Traits<T>::Vec;
  constexpr int EPV = Traits<T>::EPV;
  const size_t nchunks = n / ((size_t)SWG * EPV);                // whole chunks only; the tail is never sampled
  const size_t ngroups = nchunks / group;
  const Vec* src = reinterpret_cast<const Vec*>(x);
  StatAcc<T> acc;
  acc.init();
  // up to four chunks of a workgroup in flight at once (the kernel is a handful of dependent round trips otherwise)
  for (size_t g0 = wg; g0 < ngroups; g0 += (size_t)nwg * 4) {
    Vec v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const size_t g = g0 + (size_t)u * nwg;
      const size_t gg = g < ngroups ? g : g0;                    // (a repeated chunk changes neither max nor min; its sum is skipped)
      const unsigned h = ((unsigned)gg * 2654435761u) >> 8;
      const size_t chunk = gg * group + h % group;
      v[u] = load_stream(&src[chunk * SWG + threadIdx.x]);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const bool real = g0 + (size_t)u * nwg < ngroups;
      T e[EPV];
      Traits<T>::unpack(v[u], e);
#pragma unroll
      for (int k = 0; k < EPV; k++) acc.add(e[k], real);
    }
  }
  __shared__ double ss[3 * (SWG / 64)];
  acc.flush(part, wg, ss, SWG / 64);
}

// ---- batches (dctz_device.h: BatchFwd): calc_data_stat of k arrays in one launch, the scaling factor of every array
// chosen on the device (one workgroup per array), and the scaled copies.  k_stats_batch is the FIRST kernel of a batch
// sequence: it reads the item table from the host's pinned copy and leaves it in device memory for the kernels behind it.
__device__ __forceinline__ unsigned aux_item_of(const unsigned* __restrict__ first, const unsigned k, const unsigned b) {
  const unsigned lane = threadIdx.x & 63u;
  unsigned cnt = 0;
  for (unsigned base = 0; base < k; base += 64u) {
    const unsigned i = base + lane;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(i < k && first[i] <= b);
    cnt += (unsigned)__popcll(m);
    if (m != ~0ull) break;
  }
  return cnt - 1u;
}
template <typename T>
__global__ __launch_bounds__(SWG) void k_stats_batch(const BatchFwd<T>* items, const unsigned* __restrict__ first, unsigned k,
                                                     const uint4* __restrict__ blob_src, uint4* __restrict__ blob_dst, unsigned blob_vecs,
                                                     double* __restrict__ part) {
  for (unsigned v = blockIdx.x * SWG + threadIdx.x; v < blob_vecs; v += gridDim.x * SWG) blob_dst[v] = blob_src[v];
  const unsigned i = aux_item_of(first, k, blockIdx.x);
  const BatchFwd<T>& it = items[i];
  // (it.sample: a speculative item -- the pass reads one chunk out of every `sample`, k_compress_batch<STATS> takes the true
  // statistics while it streams the array, the host verifies the choice of sf afterwards: DESIGN section 3.4)
  if (it.sample) stats_sample_body<T>(it.p.x, (size_t)it.n, it.sample, part + 3 * (size_t)it.part_base, blockIdx.x - first[i], it.nparts);
  else stats_body<T>(it.p.x, (size_t)it.n, part + 3 * (size_t)it.part_base, blockIdx.x - first[i], it.nparts);
}
// (k_sf_batch as the LAST workgroup of every array inside k_stats_batch -- a ticket per array -- was measured in round 3 and
// dropped: the two agent-scope fences every statistics workgroup then needs cost more than the kernel boundary they save,
// 24 small arrays: 70 -> 101 us per compress batch; with the C2 field's 163 statistics workgroups: 97 -> 187 us.)
template <typename T>
__global__ __launch_bounds__(SWG) void k_sf_batch(const BatchFwd<T>* items, const double* part, double* bstats, SfTable tab) {
  const BatchFwd<T>& it = items[blockIdx.x];
  stats_final_sf_body(part + 3 * (size_t)it.part_base, (int)it.nparts, bstats + 3 * (size_t)blockIdx.x, it.p.ctl, tab,
                      const_cast<SfGuess*>(it.p.guess), nullptr);
}
// x / sf into the caller's copy (dctz-comp-lib.c:193-216), sf = what k_sf_batch chose for the array
template <typename T>
__global__ __launch_bounds__(SWG) void k_scale_batch(const BatchFwd<T>* items, const unsigned* __restrict__ first, unsigned k) {
  using Vec = typename Traits<T>::Vec;
  constexpr int EPV = Traits<T>::EPV;
  const unsigned i = aux_item_of(first, k, blockIdx.x);
  const BatchFwd<T>& it = items[i];
  const unsigned wg = blockIdx.x - first[i], nwg = first[i + 1] - first[i];
  const T sf = (T)it.p.guess->sf;
  const T* x = it.p.x;
  T* out = it.scaled;
  if (sf == T(1) && out == x) return;                      // :193 / :208: nothing to do
  const size_t n = it.n, nvec = n / EPV;
  const Vec* v = reinterpret_cast<const Vec*>(x);
  Vec* o = reinterpret_cast<Vec*>(out);
  for (size_t j = (size_t)wg * SWG + threadIdx.x; j < nvec; j += (size_t)nwg * SWG) {
    Vec a = v[j];
    if (sf != T(1)) Traits<T>::div(a, sf);
    o[j] = a;
  }
  if (wg == 0 && threadIdx.x == 0)
    for (size_t j = nvec * EPV; j < n; j++) out[j] = (sf != T(1)) ? x[j] / sf : x[j];
}
template <typename T>
void launch_stats_batch(const BatchFwd<T>* items_src, const unsigned* first_src, unsigned k, unsigned grid, const void* blob_src, void* blob_dst,
                        size_t blob_bytes, double* part, hipStream_t s) {
  hipLaunchKernelGGL(k_stats_batch<T>, dim3(grid), dim3(SWG), 0, s, items_src, first_src, k, (const uint4*)blob_src, (uint4*)blob_dst,
                     (unsigned)(blob_bytes / 16), part);
}
template <typename T>
void launch_sf_batch(const BatchFwd<T>* items, unsigned k, const double* part, double* bstats, SfTable tab, hipStream_t s) {
  hipLaunchKernelGGL(k_sf_batch<T>, dim3(k), dim3(SWG), 0, s, items, part, bstats, tab);
}
template <typename T>
void launch_scale_batch(const BatchFwd<T>* items, const unsigned* first, unsigned k, unsigned grid, hipStream_t s) {
  hipLaunchKernelGGL(k_scale_batch<T>, dim3(grid), dim3(SWG), 0, s, items, first, k);
}

template <typename T>
__global__ __launch_bounds__(SWG) void k_stats_sample(const T* __restrict__ x, size_t n, unsigned group,
                                                       double* __restrict__ part) {
  stats_sample_body<T>(x, n, group, part, blockIdx.x, gridDim.x);
}

// Serial-order sum for the header's `mean` (util.c:18-28 / :31-41): the reference
// adds x[1..N-1] one after the other in the data type, and a tree reduction
// cannot reproduce those roundings.  One wavefront: all lanes stage a chunk in
// LDS with coalesced loads, lane 0 adds it up in index order.  Slow by design
// (one dependent add per element) and OFF the critical path: the host wrapper
// runs it on a side stream underneath the zlib tail.
template <typename T>
__global__ __launch_bounds__(64) void k_serial_sum(const T* __restrict__ x, size_t n, double* __restrict__ out) {
  constexpr int CH = 4096;
  __shared__ T buf[CH];
  const int lane = threadIdx.x;
  T sum = T(0);
  for (size_t base = 0; base < n; base += CH) {
    const size_t m = (n - base < (size_t)CH) ? n - base : (size_t)CH;
    for (size_t i = lane; i < m; i += 64) buf[i] = x[base + i];
    __syncthreads();
    if (lane == 0) {
      size_t i = (base == 0) ? 1 : 0;          // util.c:22: the loop starts at i = 1
      for (; i < m; i++) sum += buf[i];
    }
    __syncthreads();
  }
  if (lane == 0) out[0] = (double)sum;
}

// out[i] = x[i] / sf (dctz-comp-lib.c:193-216), in place (out == x) or into the caller's copy:
// the reference's in-place side effect on the host buffer.
template <typename T>
__global__ __launch_bounds__(SWG) void k_scale(const T* __restrict__ x, T* __restrict__ out, size_t n, T sf) {
  using Vec = typename Traits<T>::Vec;
  constexpr int EPV = Traits<T>::EPV;
  const size_t nvec = n / EPV;
  const Vec* v = reinterpret_cast<const Vec*>(x);
  Vec* o = reinterpret_cast<Vec*>(out);
  for (size_t i = (size_t)blockIdx.x * SWG + threadIdx.x; i < nvec; i += (size_t)gridDim.x * SWG) {
    Vec a = v[i];
    Traits<T>::div(a, sf);
    o[i] = a;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t i = nvec * EPV; i < n; i++) out[i] = x[i] / sf;
}

// Diagnostics: FastDiv against the compiler's own division, element by element.
template <typename T>
__global__ __launch_bounds__(SWG) void k_debug_divide(const T* __restrict__ x, size_t n, T d, int ok,
                                                     T* __restrict__ fast, T* __restrict__ ref) {
  FastDiv<T> fd;
  fd.init(d, ok != 0);
  for (size_t i = (size_t)blockIdx.x * SWG + threadIdx.x; i < n; i += (size_t)gridDim.x * SWG) {
    fast[i] = fd.div(x[i]);
    ref[i] = x[i] / d;
  }
}

// ============================================================ transform only ==
// Batched dct_fftw / ifft_idct over all full blocks (dct.h:17-27; dct-test.c:81-89,144-152): one block per
// thread, straight from / to HBM (a utility entry point, not on the codec's path).
template <typename T, bool INVERSE>
__global__ __launch_bounds__(WG) void k_dct_blocks(const T* __restrict__ x, T* __restrict__ out, const T* __restrict__ tab, unsigned nfull) {
  using Vec = typename Traits<T>::Vec;
  constexpr int EPV = Traits<T>::EPV;
  for (unsigned blk = blockIdx.x * WG + threadIdx.x; blk < nfull; blk += gridDim.x * WG) {
    T v[64];
    const Vec* src = reinterpret_cast<const Vec*>(x + (size_t)blk * 64);
#pragma unroll
    for (int c = 0; c < 64 / EPV; c++) Traits<T>::unpack(src[c], &v[c * EPV]);
    if (INVERSE) dct64_inv<T, CTab<T>>(v, as_ctab<T>(tab)); else dct64_fwd<T, CTab<T>>(v, as_ctab<T>(tab));
    Vec* dst = reinterpret_cast<Vec*>(out + (size_t)blk * 64);
#pragma unroll
    for (int c = 0; c < 64 / EPV; c++) dst[c] = Traits<T>::pack(&v[c * EPV]);
  }
}

template <typename T, bool INVERSE>
__global__ __launch_bounds__(64) void k_dct_rem(const T* __restrict__ x, T* __restrict__ out, const T* __restrict__ rt, int l) {
  __shared__ T v[128];
  __shared__ T w[128];
  const int k = threadIdx.x;
  const int N = (l & 1) ? 2 * l : l;
  v[k] = T(0); v[k + 64] = T(0); w[k] = T(0); w[k + 64] = T(0);
  __syncthreads();
  if (!INVERSE) {
    if (k < l) {
      const T a = x[k];
      if (l & 1) { v[k] = a; v[l + (l - 1 - k)] = a; }
      else if (k & 1) v[l - 1 - (k >> 1)] = a;
      else v[k >> 1] = a;
    }
    __syncthreads();
    if (k < l) {
      T sr = T(0), si = T(0);
      for (int j = 0; j < N; j++) {
        const int tt = (j * k) % N;
        sr = sr + v[j] * rt[RTAB_WR + tt];
        si = si + v[j] * rt[RTAB_WI + tt];
      }
      out[k] = rt[RTAB_AS + k] * sr + rt[RTAB_AX + k] * si;
    }
  } else {
    if (k < l) {
      v[k] = rt[RTAB_IAS + k] * x[k];
      w[k] = rt[RTAB_IAX + k] * x[k];
      if ((l & 1) && k >= 1) {
        v[l + k] = rt[RTAB_IAX + k] * x[l - k];
        w[l + k] = -(rt[RTAB_IAS + k] * x[l - k]);
      }
    }
    __syncthreads();
    if (k < l) {
      const int s = (l & 1) ? k : ((k & 1) ? l - 1 - (k >> 1) : (k >> 1));
      T acc = T(0);
      for (int j = 0; j < N; j++) {
        const int tt = (s * j) % N;
        acc = acc + (v[j] * rt[RTAB_WR + tt] - w[j] * rt[RTAB_WI + tt]);
      }
      out[k] = (l & 1) ? (acc / (T)l) / T(2) : acc / (T)l;
    }
  }
}

// ==================================================== multi-dimensional blocks ==
// SURVEY section 8 f4 (not in the reference's library; the hint is dct-fftw-test.c:74-97).  A 2-D array is cut into
// 8 x 8 tiles, a 3-D array into 4 x 4 x 4 tiles (last axis fastest; edge tiles repeat the last sample), and
// k_gather_nd lays the tiles out block after block -- row-major over the tile grid, row-major inside a tile -- so that
// the 1-D pipeline above runs on them unchanged with the separable block transform (GEOM).  One thread moves one
// 16-byte piece of the block-linear side: fully coalesced there, whole rows of a tile (32 / 64 bytes) on the array side.
// The same pass takes calc_data_stat's reductions over the ORIGINAL elements (util.c:12-44; a repeated edge sample
// changes neither max nor min and stays out of the sum; x[0] never enters the sum, util.c:22).
template <typename T>
__device__ __forceinline__ void nd_locate(const NdShape& sh, size_t q, size_t (&src)[Traits<T>::EPV], bool (&real)[Traits<T>::EPV]) {
  constexpr int EPV = Traits<T>::EPV;
  const size_t blk = q / (64 / EPV);
  const int j0 = (int)(q % (64 / EPV)) * EPV;                       // first element of the piece inside its block
  if (sh.nd == 2) {
    const size_t b0 = blk / sh.nb[1], b1 = blk % sh.nb[1];
    const size_t r = b0 * 8 + (size_t)(j0 >> 3);
    const bool rin = r < sh.d[0];
    const size_t rr = rin ? r : sh.d[0] - 1;
#pragma unroll
    for (int k = 0; k < EPV; k++) {
      const size_t c = b1 * 8 + (size_t)((j0 & 7) + k);
      const bool cin = c < sh.d[1];
      src[k] = rr * sh.d[1] + (cin ? c : sh.d[1] - 1);
      real[k] = rin && cin;
    }
  } else {
    const size_t b2 = blk % sh.nb[2], t = blk / sh.nb[2], b1 = t % sh.nb[1], b0 = t / sh.nb[1];
    const size_t z = b0 * 4 + (size_t)(j0 >> 4), y = b1 * 4 + (size_t)((j0 >> 2) & 3);
    const bool zin = z < sh.d[0], yin = y < sh.d[1];
    const size_t base = ((zin ? z : sh.d[0] - 1) * sh.d[1] + (yin ? y : sh.d[1] - 1)) * sh.d[2];
#pragma unroll
    for (int k = 0; k < EPV; k++) {
      const size_t xx = b2 * 4 + (size_t)((j0 & 3) + k);
      const bool xin = xx < sh.d[2];
      src[k] = base + (xin ? xx : sh.d[2] - 1);
      real[k] = zin && yin && xin;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(SWG) void k_gather_nd(const T* __restrict__ x, T* __restrict__ lin, NdShape sh, double* __restrict__ part) {
  using Vec = typename Traits<T>::Vec;
  constexpr int EPV = Traits<T>::EPV;
  const size_t nq = sh.nblk * (64 / EPV);
  StatAcc<T> acc;
  acc.init();
  for (size_t q = (size_t)blockIdx.x * SWG + threadIdx.x; q < nq; q += (size_t)gridDim.x * SWG) {
    size_t src[EPV];
    bool real[EPV];
    nd_locate<T>(sh, q, src, real);
    T e[EPV];
#pragma unroll
    for (int k = 0; k < EPV; k++) {
      e[k] = x[src[k]];
      acc.minmax(e[k]);
      if (real[k] && src[k] != 0) acc.sum += (double)e[k];
    }
    reinterpret_cast<Vec*>(lin)[q] = Traits<T>::pack(e);
  }
  __shared__ double ss[3 * (SWG / 64)];
  acc.flush(part, blockIdx.x, ss, SWG / 64);
}

// block-linear reconstruction -> the array (the padding of edge tiles is dropped)
template <typename T>
__global__ __launch_bounds__(SWG) void k_scatter_nd(const T* __restrict__ lin, T* __restrict__ out, NdShape sh) {
  using Vec = typename Traits<T>::Vec;
  constexpr int EPV = Traits<T>::EPV;
  const size_t nq = sh.nblk * (64 / EPV);
  for (size_t q = (size_t)blockIdx.x * SWG + threadIdx.x; q < nq; q += (size_t)gridDim.x * SWG) {
    size_t dst[EPV];
    bool real[EPV];
    nd_locate<T>(sh, q, dst, real);
    T e[EPV];
    Traits<T>::unpack(reinterpret_cast<const Vec*>(lin)[q], e);
#pragma unroll
    for (int k = 0; k < EPV; k++)
      if (real[k]) out[dst[k]] = e[k];
  }
}

template <typename T>
void launch_gather_nd(const T* x, T* lin, const NdShape& sh, double* part, int nparts, hipStream_t s) {
  hipLaunchKernelGGL(k_gather_nd<T>, dim3(nparts), dim3(SWG), 0, s, x, lin, sh, part);
}
template <typename T>
void launch_scatter_nd(const T* lin, T* out, const NdShape& sh, int grid, hipStream_t s) {
  hipLaunchKernelGGL(k_scatter_nd<T>, dim3(grid), dim3(SWG), 0, s, lin, out, sh);
}

// ===================================================================== PSNR ==
// calc_psnr's reductions (util.c:54-104): min / max of the original, max |x - r|, sum of (x - r)^2 with the
// difference and its square taken in the data type (util.c:72-73 / :88-89), summed in double -- in tree order,
// so the last digits of the sum differ from the reference's serial loop (relative 1e-15).
template <typename T>
__global__ __launch_bounds__(SWG) void k_psnr(const T* __restrict__ x, const T* __restrict__ r, size_t n, double* __restrict__ part) {
  double mn = 1.79769313486231570815e308, mx = -1.79769313486231570815e308, worst = 0.0, sq = 0.0;
  for (size_t i = (size_t)blockIdx.x * SWG + threadIdx.x; i < n; i += (size_t)gridDim.x * SWG) {
    const T a = x[i];
    const T e = a - r[i];
    mn = fmin(mn, (double)a); mx = fmax(mx, (double)a);
    worst = fmax(worst, (double)fabs(e));
    sq += (double)(e * e);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    mn = fmin(mn, __shfl_down(mn, d)); mx = fmax(mx, __shfl_down(mx, d));
    worst = fmax(worst, __shfl_down(worst, d)); sq += __shfl_down(sq, d);
  }
  __shared__ double s[4][SWG / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s[0][wave] = mn; s[1][wave] = mx; s[2][wave] = worst; s[3][wave] = sq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < SWG / 64; w++) { mn = fmin(mn, s[0][w]); mx = fmax(mx, s[1][w]); worst = fmax(worst, s[2][w]); sq += s[3][w]; }
    part[4 * blockIdx.x + 0] = mn; part[4 * blockIdx.x + 1] = mx; part[4 * blockIdx.x + 2] = worst; part[4 * blockIdx.x + 3] = sq;
  }
}
__global__ __launch_bounds__(SWG) void k_psnr_final(const double* __restrict__ part, int nparts, double* __restrict__ out) {
  double mn = 1.79769313486231570815e308, mx = -1.79769313486231570815e308, worst = 0.0, sq = 0.0;
  for (int i = threadIdx.x; i < nparts; i += SWG) {
    mn = fmin(mn, part[4 * i]); mx = fmax(mx, part[4 * i + 1]); worst = fmax(worst, part[4 * i + 2]); sq += part[4 * i + 3];
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    mn = fmin(mn, __shfl_down(mn, d)); mx = fmax(mx, __shfl_down(mx, d));
    worst = fmax(worst, __shfl_down(worst, d)); sq += __shfl_down(sq, d);
  }
  __shared__ double s[4][SWG / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s[0][wave] = mn; s[1][wave] = mx; s[2][wave] = worst; s[3][wave] = sq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < SWG / 64; w++) { mn = fmin(mn, s[0][w]); mx = fmax(mx, s[1][w]); worst = fmax(worst, s[2][w]); sq += s[3][w]; }
    out[0] = mn; out[1] = mx; out[2] = worst; out[3] = sq;
  }
}

// ================================================================= launchers ==
template <typename T>
void launch_stats(const T* x, size_t n, double* part, int nparts, double* out, hipStream_t s, HostBox* box, unsigned long long seq, Ctl* zero,
                  const SfTable* tab, SfGuess* guess) {
  hipLaunchKernelGGL(k_stats<T>, dim3(nparts), dim3(SWG), 0, s, x, n, part);
  if (tab != nullptr) hipLaunchKernelGGL(k_stats_final_sf, dim3(1), dim3(SWG), 0, s, (const double*)part, nparts, out, zero, *tab, guess, box);
  else hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(SWG), 0, s, (const double*)part, nparts, out, box, seq, zero);
}

template <typename T>
void launch_stats_sample(const T* x, size_t n, unsigned group, double* part, int nparts, double* out, hipStream_t s,
                         HostBox* box, unsigned long long seq, Ctl* zero, const SfTable* tab, SfGuess* guess) {
  hipLaunchKernelGGL(k_stats_sample<T>, dim3(nparts), dim3(SWG), 0, s, x, n, group, part);
  if (tab != nullptr) hipLaunchKernelGGL(k_stats_final_sf, dim3(1), dim3(SWG), 0, s, (const double*)part, nparts, out, zero, *tab, guess, box);
  else hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(SWG), 0, s, (const double*)part, nparts, out, box, seq, zero);
}
void launch_stats_final(const double* part, int nparts, double* out, hipStream_t s, HostBox* box, unsigned long long seq, Ctl* zero,
                        const SfTable* tab, SfGuess* guess) {
  if (tab != nullptr) hipLaunchKernelGGL(k_stats_final_sf, dim3(1), dim3(SWG), 0, s, part, nparts, out, zero, *tab, guess, box);
  else hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(SWG), 0, s, part, nparts, out, box, seq, zero);
}

template <typename T>
void launch_debug_divide(const T* x, size_t n, T d, int ok, T* fast, T* ref, hipStream_t s) {
  hipLaunchKernelGGL(k_debug_divide<T>, dim3(1024), dim3(SWG), 0, s, x, n, d, ok, fast, ref);
}

template <typename T>
void launch_serial_sum(const T* x, size_t n, double* out, hipStream_t s) {
  hipLaunchKernelGGL(k_serial_sum<T>, dim3(1), dim3(64), 0, s, x, n, out);
}

template <typename T>
void launch_scale(const T* x, T* out, size_t n, T sf, int grid, hipStream_t s) {
  hipLaunchKernelGGL(k_scale<T>, dim3(grid), dim3(SWG), 0, s, x, out, n, sf);
}

template <typename T>
void launch_dct_blocks(const T* x, T* out, const T* gtab, const T* rtab, size_t n, bool inverse, int grid,
                       hipStream_t s) {
  const unsigned nfull = (unsigned)(n / 64);
  const int l = (int)(n % 64);
  if (nfull) {
    const int g = (int)min((unsigned)grid, (nfull + WG - 1) / WG);
    if (inverse) hipLaunchKernelGGL((k_dct_blocks<T, true>), dim3(g), dim3(WG), 0, s, x, out, gtab, nfull);
    else hipLaunchKernelGGL((k_dct_blocks<T, false>), dim3(g), dim3(WG), 0, s, x, out, gtab, nfull);
  }
  if (l) {
    const T* xr = x + (size_t)nfull * 64;
    T* orr = out + (size_t)nfull * 64;
    if (inverse) hipLaunchKernelGGL((k_dct_rem<T, true>), dim3(1), dim3(64), 0, s, xr, orr, rtab, l);
    else hipLaunchKernelGGL((k_dct_rem<T, false>), dim3(1), dim3(64), 0, s, xr, orr, rtab, l);
  }
}

template <typename T>
void launch_psnr(const T* x, const T* r, size_t n, double* part, int nparts, double* out, hipStream_t s) {
  hipLaunchKernelGGL(k_psnr<T>, dim3(nparts), dim3(SWG), 0, s, x, r, n, part);
  hipLaunchKernelGGL(k_psnr_final, dim3(1), dim3(SWG), 0, s, (const double*)part, nparts, out);
}

// explicit instantiations used by dctz_shim.hip
#define INST_AUX(T) \
  template void launch_stats<T>(const T*, size_t, double*, int, double*, hipStream_t, HostBox*, unsigned long long, Ctl*, const SfTable*, SfGuess*); \
  template void launch_stats_sample<T>(const T*, size_t, unsigned, double*, int, double*, hipStream_t, HostBox*, unsigned long long, Ctl*, const SfTable*, SfGuess*); \
  template void launch_debug_divide<T>(const T*, size_t, T, int, T*, T*, hipStream_t);                  \
  template void launch_serial_sum<T>(const T*, size_t, double*, hipStream_t);                           \
  template void launch_scale<T>(const T*, T*, size_t, T, int, hipStream_t);                             \
  template void launch_gather_nd<T>(const T*, T*, const NdShape&, double*, int, hipStream_t);           \
  template void launch_scatter_nd<T>(const T*, T*, const NdShape&, int, hipStream_t);                   \
  template void launch_dct_blocks<T>(const T*, T*, const T*, const T*, size_t, bool, int, hipStream_t); \
  template void launch_psnr<T>(const T*, const T*, size_t, double*, int, double*, hipStream_t);         \
  template void launch_stats_batch<T>(const BatchFwd<T>*, const unsigned*, unsigned, unsigned, const void*, void*, size_t, double*, hipStream_t); \
  template void launch_sf_batch<T>(const BatchFwd<T>*, unsigned, const double*, double*, SfTable, hipStream_t);                           \
  template void launch_scale_batch<T>(const BatchFwd<T>*, const unsigned*, unsigned, unsigned, hipStream_t);
INST_AUX(double)
INST_AUX(float)

}  // namespace dctz
