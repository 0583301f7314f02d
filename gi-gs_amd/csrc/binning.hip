// binning.hip -- instance expansion and tile sort.
//
// Reference behaviour restated (R/cuda_rasterizer/rasterizer_impl.cu):
//   cub::DeviceScan::InclusiveSum          :585      -> rocprim::inclusive_scan
//   duplicateWithKeys                      :70-112
//   cub::DeviceRadixSort::SortPairs        :612-617  -> rocprim::radix_sort_pairs (stable LSD
//                                                      sort over bits [0, 32+bit) -> identical
//                                                      point_list)
//   cudaMemset(ranges) + identifyTileRanges :621, :117-138
//
// All stages are integer, HBM-bound work (12 B written per instance by the
// expansion, 24 B moved per instance per sort pass).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string.h>
#include <rocprim/block/block_radix_sort.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "gigs_common.h"

namespace gigs {

size_t scan_temp_bytes(int P) {
  size_t bytes = 0;
  uint32_t* dummy = nullptr;
  (void)rocprim::inclusive_scan(nullptr, bytes, dummy, dummy, (size_t)P, rocprim::plus<uint32_t>());
  return bytes;
}

hipError_t scan_tiles(const GeomState& g, int P, hipStream_t s) {
  size_t bytes = g.scan_size;
  return rocprim::inclusive_scan(g.scan_space, bytes, g.tiles_touched, g.point_offsets, (size_t)P,
                                 rocprim::plus<uint32_t>(), s);
}

size_t sort_temp_bytes(int R) {
  size_t bytes = 0;
  uint64_t* k = nullptr;
  uint32_t* v = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)R, 0, 64);
  return bytes;
}

hipError_t sort_pairs(const BinningState& b, int R, int end_bit, hipStream_t s) {
  size_t bytes = b.sort_size;
  return rocprim::radix_sort_pairs(b.sort_space, bytes, b.keys_unsorted, b.keys, b.values_unsorted,
                                   b.point_list, (size_t)R, 0, (unsigned)end_bit, s);
}

// One lane per Gaussian; emission order y-major, x-minor, as the reference.
__global__ void __launch_bounds__(256)
duplicate_kernel(int P, const int* __restrict__ radii, unsigned gx, unsigned gy,
                 const float* __restrict__ means2D, const float* __restrict__ depths,
                 const uint32_t* __restrict__ offsets, uint64_t* __restrict__ keys,
                 uint32_t* __restrict__ values) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  const int r = radii[idx];
  if (r > 0) {
    uint32_t off = (idx == 0) ? 0 : offsets[idx - 1];
    const float2 xy = reinterpret_cast<const float2*>(means2D)[idx];
    unsigned minx, miny, maxx, maxy;
    tile_rect(xy.x, xy.y, r, gx, gy, minx, miny, maxx, maxy);
    const uint32_t dbits = __float_as_uint(depths[idx]);
    for (int y = (int)miny; y < (int)maxy; y++) {
      for (int x = (int)minx; x < (int)maxx; x++) {
        uint64_t key = (uint64_t)(y * gx + x);
        key <<= 32;
        key |= dbits;
        keys[off] = key;
        values[off] = (uint32_t)idx;
        off++;
      }
    }
  }
}

void launch_duplicate(int P, const int* radii, unsigned gx, unsigned gy, const GeomState& g,
                      const BinningState& b, hipStream_t s) {
  hipLaunchKernelGGL(duplicate_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, radii, gx, gy,
                     g.means2D, g.depths, g.point_offsets, b.keys_unsorted, b.values_unsorted);
}

__global__ void __launch_bounds__(256)
tile_ranges_kernel(int L, const uint64_t* __restrict__ keys, uint2* __restrict__ ranges) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= L) return;
  const uint32_t currtile = (uint32_t)(keys[idx] >> 32);
  if (idx == 0) {
    ranges[currtile].x = 0;
  } else {
    const uint32_t prevtile = (uint32_t)(keys[idx - 1] >> 32);
    if (currtile != prevtile) {
      ranges[prevtile].y = idx;
      ranges[currtile].x = idx;
    }
  }
  if (idx == L - 1) ranges[currtile].y = L;
}

// Launch order of the blend workgroups.  A blend kernel lasts as long as its longest tile list (lists are very
// skewed: median ~200, maximum ~6000 instances at 800x800), so the long tiles must start first instead of
// wherever raster order puts them.  rank(t) = number of tiles with a longer list (ties by index): an O(T^2)
// count is cheapest at T = 2500 tiles (one wave per tile, a few microseconds); above kMaxOrderedTiles the order
// is the identity.
constexpr int kMaxOrderedTiles = 16384;
__global__ void __launch_bounds__(256)
tile_order_kernel(int T, const uint2* __restrict__ ranges, uint32_t* __restrict__ order, int identity) {
  // one wave per tile: its 64 lanes stride over all tiles and count, by ballot, those that rank before it
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= T) return;
  if (identity) {
    if (lane == 0) order[t] = (uint32_t)t;
    return;
  }
  const uint32_t mine = ranges[t].y - ranges[t].x;
  uint32_t rank = 0;
  for (int base = 0; base < T; base += 64) {
    const int u = base + lane;
    const uint32_t other = u < T ? ranges[u].y - ranges[u].x : 0u;
    const bool before = u < T && (other > mine || (other == mine && u < t));
    rank += (uint32_t)__popcll(__ballot(before));
  }
  if (lane == 0) order[rank] = (uint32_t)t;
}

void launch_tile_order(int T, const uint2* ranges, uint32_t* tile_order, hipStream_t s) {
  hipLaunchKernelGGL(tile_order_kernel, dim3((T + 3) / 4), dim3(256), 0, s, T, ranges, tile_order,
                     T > kMaxOrderedTiles ? 1 : 0);
}

void launch_tile_ranges(int R, const BinningState& b, uint2* ranges, hipStream_t s) {
  if (R > 0)
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((R + 255) / 256), dim3(256), 0, s, R, b.keys, ranges);
}


// ------------------------------------------------------------------------------------------
// Tile-bucketed binning (the default; GIGS_BINNING=legacy selects the reference-shaped path above)
// ------------------------------------------------------------------------------------------
// The reference expands every Gaussian into (tile | depth, index) pairs in Gaussian order and sorts ALL of them
// with a global 32+bit-bit radix sort (rasterizer_impl.cu:585-621): six passes over 24 B per instance, plus a host
// read of the instance count in the middle of the forward.  What the blend needs is only, per tile, the list of its
// Gaussians ordered by depth with ties in Gaussian-index order (a stable sort of index-ordered input).  Here:
//   1. bin_count:   <= kBinGroups workgroups, each over a contiguous chunk of Gaussians, histogram their tiles in
//                   LDS (ds_add; large footprints are expanded by the whole wave) -> bin_hist[g][t];
//   2. bin_prefix:  per tile, exclusive prefix of the counts over the chunks (in place) and the tile's total; then an
//                   exclusive scan of the totals = `ranges` (identifyTileRanges' result) and R -- all on the device;
//   3. bin_scatter: the same walk again; a slot inside the tile's range comes from an LDS cursor, the 64-bit key
//                   (depth bits << idx_bits | Gaussian index, idx_bits = bits of P - 1) goes there.  The order inside a tile is arbitrary here...
//   4. bin_sort:    ...because each tile's keys are then sorted (one workgroup per tile, bitonic network in LDS, longest
//                   lists first).  The keys are unique and (depth, index)-ordered = the stable depth sort of
//                   index-ordered input, so point_list -- and `keys` in the reference's tile|depth format, which is
//                   written too -- equal the reference's bit for bit.
// No host read-back anywhere: with a caller-given capacity (gigs_set_async_binning) the forward never synchronises and
// can be captured into a hipGraph; R and an overflow flag stay in device memory.  Instances beyond the capacity are
// dropped (ranges are clamped, the flag is raised): memory-safe, and the caller replays the step with a larger capacity.
// [row0, row1): the tile rows this pass of the walk emits (bin_scatter's bands; everything for the other walks)
__device__ __forceinline__ bool bin_rect(int idx, int end, const int* __restrict__ radii, const float* __restrict__ means2D,
                                         unsigned gx, unsigned gy, unsigned& minx, unsigned& miny, unsigned& w, unsigned& n,
                                         unsigned row0 = 0u, unsigned row1 = 0xffffffffu) {
  minx = miny = w = n = 0;
  if (idx >= end) return false;
  const int r = radii[idx];
  if (r <= 0) return false;
  const float2 xy = reinterpret_cast<const float2*>(means2D)[idx];
  unsigned maxx, maxy;
  tile_rect(xy.x, xy.y, r, gx, gy, minx, miny, maxx, maxy);
  miny = max(miny, row0);
  maxy = min(maxy, row1);
  if (maxy <= miny) return false;
  w = maxx - minx;
  n = w * (maxy - miny);
  return n > 0;
}

constexpr unsigned kBinWaveExpand = 12;  // footprints with more tiles than this are expanded by the whole wave

// Walks the tiles of the chunk's Gaussians; f(idx, tile, payload) is called once per (Gaussian, tile) pair, with
// payload = pay(idx) evaluated once per Gaussian (the split path's depth bucket; the plain path passes nothing).
template <typename Pay, typename F>
__device__ __forceinline__ void bin_walk(int P, const int* __restrict__ radii, const float* __restrict__ means2D, unsigned gx,
                                         unsigned gy, Pay pay, F f, unsigned row0 = 0u, unsigned row1 = 0xffffffffu) {
  const int groups = min((int)kBinGroups, (P + 255) / 256);
  const int chunk = (P + groups - 1) / groups;
  const int begin = blockIdx.x * chunk, end = min(P, begin + chunk);
  const int lane = threadIdx.x & 63;
  for (int base = begin; base < end; base += blockDim.x) {
    const int idx = base + threadIdx.x;
    unsigned minx, miny, w, n;
    const bool any = bin_rect(idx, end, radii, means2D, gx, gy, minx, miny, w, n, row0, row1);
    const unsigned payload = any ? pay(idx) : 0u;
    const bool big = any && n > kBinWaveExpand;
    if (any && !big) {
      unsigned x = 0, y = 0;
      for (unsigned k = 0; k < n; k++) {
        f(idx, (miny + y) * gx + minx + x, payload);
        if (++x == w) { x = 0; y++; }
      }
    }
    unsigned long long m = __builtin_amdgcn_ballot_w64(big);
    while (m) {
      const int l = __builtin_ctzll(m);
      m &= m - 1;
      const unsigned bminx = __shfl(minx, l), bminy = __shfl(miny, l), bw = __shfl(w, l), bn = __shfl(n, l);
      const unsigned bpay = __shfl(payload, l);
      const int bidx = __shfl(idx, l);
      const unsigned bh = bn / max(bw, 1u);  // once per large footprint
      for (unsigned y = 0; y < bh; y++)
        for (unsigned x = lane; x < bw; x += 64) f(bidx, (bminy + y) * gx + bminx + x, bpay);
    }
  }
}
struct NoPay { __device__ __forceinline__ unsigned operator()(int) const { return 0u; } };

// kLanes = 256, or 1024 for large clouds (3 M Gaussians: 0.44 -> 0.15 ms; sixteen waves per CU instead of four cover the
// latency of the walk's loads)
template <int kLanes>
__global__ void __launch_bounds__(kLanes)
bin_count_kernel(int P, int T, const int* __restrict__ radii, const float* __restrict__ means2D, unsigned gx, unsigned gy,
                 uint32_t* __restrict__ bin_hist) {
  extern __shared__ uint32_t s_hist[];
  for (int t = threadIdx.x; t < T; t += kLanes) s_hist[t] = 0;
  __syncthreads();
  bin_walk(P, radii, means2D, gx, gy, NoPay(), [&](int, unsigned tile, unsigned) { atomicAdd(&s_hist[tile], 1u); });
  __syncthreads();
  uint32_t* row = bin_hist + (size_t)blockIdx.x * T;
  for (int t = threadIdx.x; t < T; t += kLanes) row[t] = s_hist[t];
}

// per tile: exclusive prefix over the chunk counts (in place) and the total
__global__ void __launch_bounds__(1024)
bin_prefix_groups_kernel(int groups, int T, uint32_t* __restrict__ bin_hist, uint32_t* __restrict__ totals) {
  // 64 tiles x 16 chunk-parts per workgroup: coalesced across tiles, sixteen independent short serial scans per tile
  __shared__ uint32_t s_q[16][64];
  const int tl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + tl;
  const int per = (groups + 15) / 16, g0 = min(groups, q * per), g1 = min(groups, g0 + per);
  constexpr int kPer = (kBinGroups + 15) / 16;  // chunks per part (per <= kPer)
  uint32_t c[kPer];
  uint32_t run = 0;
  if (t < T) {
#pragma unroll
    for (int k = 0; k < kPer; k++) c[k] = (g0 + k < g1) ? bin_hist[(size_t)(g0 + k) * T + t] : 0u;
#pragma unroll
    for (int k = 0; k < kPer; k++) { const uint32_t v = c[k]; c[k] = run; run += v; }
  }
  s_q[q][tl] = run;
  __syncthreads();
  uint32_t add = 0;
  for (int k = 0; k < q; k++) add += s_q[k][tl];
  if (t < T) {
#pragma unroll
    for (int k = 0; k < kPer; k++)
      if (g0 + k < g1) bin_hist[(size_t)(g0 + k) * T + t] = c[k] + add;
    if (q == 15) totals[t] = add + run;
  }
}

// exclusive scan of the tile totals -> ranges (clamped to the capacity), R and the overflow flag.  One workgroup.
__global__ void __launch_bounds__(1024)
bin_prefix_tiles_kernel(int T, unsigned capacity, unsigned P, const uint32_t* __restrict__ totals, uint2* __restrict__ ranges,
                        uint32_t* __restrict__ tile_start, uint32_t* __restrict__ counters,
                        uint32_t* __restrict__ user_counters) {
  __shared__ uint32_t s_sum[1024];
  const int per = (T + 1023) / 1024;
  const int t0 = threadIdx.x * per, t1 = min(T, t0 + per);
  uint32_t local = 0;
  for (int t = t0; t < t1; t++) local += totals[t];
  s_sum[threadIdx.x] = local;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
    const uint32_t v = threadIdx.x >= (unsigned)off ? s_sum[threadIdx.x - off] : 0u;
    __syncthreads();
    s_sum[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = s_sum[threadIdx.x] - local;
  for (int t = t0; t < t1; t++) {
    const uint32_t c = totals[t];
    // an empty tile keeps (0, 0), what the reference's memset leaves there (rasterizer_impl.cu:621)
    ranges[t] = c ? make_uint2(min(run, capacity), min(run + c, capacity)) : make_uint2(0u, 0u);
    tile_start[t] = run;  // unclamped: the scatter's cursors start here even when an overflow clamped the ranges
    run += c;
  }
  if (threadIdx.x == 1023) {
    const uint32_t R = s_sum[1023];
    counters[0] = R;
    counters[1] = R > capacity ? R : 0u;
    // what these lists were made for: a later forward that REUSES them (gigs_ctx_set_reuse_binning) blends them only if its
    // own Gaussian count and capacity are these (blend.hip: otherwise every tile is treated as empty -- no index can be out of range)
    counters[2] = P;
    counters[3] = capacity;
    if (user_counters) { user_counters[0] = R; user_counters[1] = R > capacity ? R : 0u; }
  }
}

template <int kLanes>
__global__ void __launch_bounds__(kLanes)
bin_scatter_kernel(int P, int T, const int* __restrict__ radii, const float* __restrict__ means2D,
                   const float* __restrict__ depths, unsigned gx, unsigned gy, unsigned capacity, unsigned idx_bits,
                   const uint32_t* __restrict__ bin_hist, const uint32_t* __restrict__ tile_start,
                   uint64_t* __restrict__ keys, unsigned bands) {
  extern __shared__ uint32_t s_cur[];
  const uint32_t* row = bin_hist + (size_t)blockIdx.x * T;
  for (int t = threadIdx.x; t < T; t += kLanes) s_cur[t] = tile_start[t] + row[t];
  __syncthreads();
  // bands > 1 (gigs_options.bin_bands): the chunk is walked once per band of tile rows and emits only that band's instances,
  // so that at any time the grid's 8-byte key stores go to 1 / bands of the (chunk, tile) sub-ranges -- the lines they fill
  // stay cache-resident until they are full (a dense scene's 512 chunks x 4 056 tiles x 128-byte lines are 266 MB of open
  // lines against 256 MB of Infinity Cache: 3.65 x write amplification, profiles/r03/pmc_summary_r03_c4).  Same slots, same
  // keys: the order inside a (chunk, tile) sub-range is arbitrary anyway (the per-tile sort follows).
  const unsigned rows = (gy + bands - 1) / bands;
  for (unsigned b = 0; b < bands; b++)
    bin_walk(P, radii, means2D, gx, gy, NoPay(), [&](int idx, unsigned tile, unsigned) {
      const uint32_t slot = atomicAdd(&s_cur[tile], 1u);
      if (slot < capacity) keys[slot] = ((uint64_t)__float_as_uint(depths[idx]) << idx_bits) | (uint32_t)idx;
    }, b * rows, min(gy, (b + 1) * rows));
}

// Bitonic network with every compare-exchange ascending (first substage of a merge mirrors the upper half), so an
// input padded with +inf at the END stays sorted-to-the-front: indices >= n never need to hold data.
template <typename Get, typename Put>
__device__ __forceinline__ void bitonic_sort_asc(unsigned n, unsigned npad_log2, unsigned nthreads, Get get, Put put) {
  const unsigned half = (1u << npad_log2) >> 1;
  for (unsigned kb = 1; kb <= npad_log2; kb++) {
    const unsigned k = 1u << kb;
    for (unsigned i = threadIdx.x; i < half; i += nthreads) {
      const unsigned blk = i >> (kb - 1), off = i & ((k >> 1) - 1);
      const unsigned a = (blk << kb) + off, b = (blk << kb) + (k - 1 - off);
      if (b < n) {
        const uint64_t x = get(a), y = get(b);
        if (x > y) { put(a, y); put(b, x); }
      }
    }
    __syncthreads();
    for (int jb = (int)kb - 2; jb >= 0; jb--) {
      const unsigned j = 1u << jb;
      for (unsigned i = threadIdx.x; i < half; i += nthreads) {
        const unsigned a = ((i >> jb) << (jb + 1)) + (i & (j - 1)), b = a + j;
        if (b < n) {
          const uint64_t x = get(a), y = get(b);
          if (x > y) { put(a, y); put(b, x); }
        }
      }
      __syncthreads();
    }
  }
}

// One workgroup per tile.  A list of up to kBlock * kItems keys is sorted by rocPRIM's block radix sort (keys in registers,
// 8 bits per pass over the 32 + log2(P) significant bits; measured on the box, tools/microbench/bitonic_lds.hip: 8192 keys
// 55 us, 4096 keys 34 us, 1024 keys 12 us per 1024-lane workgroup -- a bitonic network in LDS, the first implementation, needs
// 70 / 40 / 20: sorting thousands of 64-bit keys on ONE CU costs tens of microseconds either way, so what matters is how many
// lists are in flight).  tile_order lists the tiles longest first, so a size class is one contiguous stretch of it: small
// persistent grids stride over the order and stop at the first list that belongs to another kernel (a grid of T workgroups
// that mostly return at once costs more in dispatch than the sorting when each reserves its LDS).
// one tile of up to 1024 * kItems keys: rocPRIM block radix sort, keys in registers (blocked arrangement)
template <int kItems, typename Storage, int kBlock = 1024>
__device__ __forceinline__ void sort_tile_radix(Storage& storage, const uint64_t* __restrict__ src, unsigned n, unsigned idx_bits,
                                                uint64_t tile_hi, uint32_t base, uint64_t* __restrict__ keys_out,
                                                uint32_t* __restrict__ point_list) {
  using sorter = rocprim::block_radix_sort<uint64_t, kBlock, kItems, rocprim::empty_type, 1, 1, 8>;
  static_assert(sizeof(Storage) >= sizeof(uint64_t) * kBlock * kItems, "the sorted keys are staged in the sorter's storage");
  const uint64_t idx_mask = (1ull << idx_bits) - 1;
  uint64_t k[kItems];
#pragma unroll
  for (int i = 0; i < kItems; i++) {
    const unsigned idx = threadIdx.x * kItems + i;
    k[i] = idx < n ? src[idx] : ~0ull;  // padding sorts to the end (all ones in every sorted bit)
  }
  // The key is depth bits << idx_bits | index.  Radix passes are spent on the DEPTH bits only (4 passes instead of the 7
  // that all 32 + idx_bits bits take); the sort is stable, so keys of equal depth come out adjacent, in their (arbitrary)
  // input order, and are put into index order by the fix-up below -- which finds nothing to do unless two Gaussians of a tile
  // have bit-identical depths.  Of the depth bits, only those that differ anywhere in THIS list are sorted: the depths of
  // one tile -- let alone of one bucket of it -- share their sign, most of their exponent and often some mantissa bits, which
  // usually saves the fourth pass.
  __shared__ unsigned long long s_diff;
  if (threadIdx.x == 0) s_diff = 0ull;
  __syncthreads();
  {
    const uint64_t ref = src[0];
    uint64_t diff = 0;
#pragma unroll
    for (int i = 0; i < kItems; i++)
      if (threadIdx.x * kItems + i < n) diff |= k[i] ^ ref;
    diff >>= idx_bits;
    if (diff) atomicOr(&s_diff, (unsigned long long)diff);
  }
  __syncthreads();
  const unsigned dd = (unsigned)s_diff;
  const unsigned hi = dd ? 32u - (unsigned)__builtin_clz(dd) : 1u;
  sorter().sort(k, storage, idx_bits, idx_bits + hi);
  __syncthreads();
  uint64_t* s_k = reinterpret_cast<uint64_t*>(&storage);
#pragma unroll
  for (int i = 0; i < kItems; i++) s_k[threadIdx.x * kItems + i] = k[i];
  __syncthreads();
  bool tie = false;
#pragma unroll
  for (int i = 0; i < kItems; i++) {
    const unsigned idx = threadIdx.x * kItems + i;
    if (idx > 0 && idx < n) {
      const uint64_t prev = i > 0 ? k[i - 1] : s_k[idx - 1];
      tie |= (prev >> idx_bits) == (k[i] >> idx_bits) && prev > k[i];
    }
  }
  if (__syncthreads_or(tie)) {
    // odd-even transposition restricted to neighbours of equal depth: runs of equal depth are sorted by index, nothing else
    // moves; as many rounds as the longest run is long (bit-identical depths inside one tile: duplicated Gaussians).  The
    // rounds are bounded: a run longer than kFixRounds keys (a cloned cloud, a plane seen head-on: thousands of identical
    // depths would mean thousands of rounds of three barriers each) is sorted over the COMPLETE key instead -- the radix
    // passes the depth-only sort saved, spent once, on this list only.
    constexpr unsigned kPairs = (unsigned)kBlock * kItems / 2;
    constexpr int kFixRounds = 64;
    bool more = true;
    for (int round = 0; round < kFixRounds && more; round++) {
      bool swapped = false;
#pragma unroll
      for (int phase = 0; phase < 2; phase++) {
        for (unsigned p = threadIdx.x; p < kPairs; p += kBlock) {
          const unsigned a = 2 * p + phase, c = a + 1;
          if (c < n) {
            const uint64_t x = s_k[a], y = s_k[c];
            if ((x >> idx_bits) == (y >> idx_bits) && x > y) { s_k[a] = y; s_k[c] = x; swapped = true; }
          }
        }
        __syncthreads();
      }
      more = __syncthreads_or(swapped);  // uniform
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < kItems; i++) k[i] = s_k[threadIdx.x * kItems + i];  // padding (all ones) included
      __syncthreads();  // s_k aliases the sorter's storage
      sorter().sort(k, storage, 0, idx_bits + hi);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < kItems; i++) s_k[threadIdx.x * kItems + i] = k[i];
      __syncthreads();
    }
  }
  for (unsigned idx = threadIdx.x; idx < n; idx += kBlock) {  // coalesced
    const uint64_t kk = s_k[idx];
    point_list[base + idx] = (uint32_t)(kk & idx_mask);
    keys_out[base + idx] = tile_hi | (kk >> idx_bits);
  }
  __syncthreads();  // the storage is reused by the next tile of this workgroup
}

// bin_sort_kernel: one 1024-lane workgroup per tile, longest lists first (tile_order), on a small persistent grid, 4 or 8 keys
// per lane (kernel <false>: lists of 2049..8192 keys, 64 KB of LDS, two workgroups per CU) or 16 (kernel <true>: up to 16384,
// 132 KB; usually empty-handed).  One launch per size class, the first layout, ran them one after the other (190 us at C2);
// lists beyond 16384 keys are sorted in place in global memory with the all-ascending bitonic network (L2-resident; rare:
// dense scenes take the global radix sort).
using SortS4 = rocprim::block_radix_sort<uint64_t, 1024, 4, rocprim::empty_type, 1, 1, 8>::storage_type;
using SortS8 = rocprim::block_radix_sort<uint64_t, 1024, 8, rocprim::empty_type, 1, 1, 8>::storage_type;
using SortS16 = rocprim::block_radix_sort<uint64_t, 1024, 16, rocprim::empty_type, 1, 1, 8>::storage_type;

// The short lists -- two thirds of the tiles of a frame -- are sorted by 256-lane workgroups (8 keys per lane, 18 KB of LDS:
// eight of them per CU instead of two 1024-lane ones, whose sixteen waves mostly wait at the passes' barriers).  The grid
// walks tile_order from its short end and stops at the first list that belongs to bin_sort_kernel.
constexpr unsigned kSmallList = 2048;
using SortSmall = rocprim::block_radix_sort<uint64_t, 256, 8, rocprim::empty_type, 1, 1, 8>::storage_type;
__global__ void __launch_bounds__(256)
bin_sort_small_kernel(int T, unsigned idx_bits, const uint32_t* __restrict__ tile_order, const uint2* __restrict__ ranges,
                      const uint64_t* __restrict__ keys_unsorted, uint64_t* __restrict__ keys_out,
                      uint32_t* __restrict__ point_list) {
  __shared__ SortSmall storage;
  for (int ob = T - 1 - (int)blockIdx.x; ob >= 0; ob -= (int)gridDim.x) {
    const uint32_t tile = tile_order[ob];
    const uint2 rg = ranges[tile];
    const unsigned n = rg.y - rg.x;
    if (n > kSmallList) break;  // ascending from this end: the rest is bin_sort_kernel's
    if (n == 0) continue;
    sort_tile_radix<8, SortSmall, 256>(storage, keys_unsorted + rg.x, n, idx_bits, (uint64_t)tile << 32, rg.x, keys_out,
                                       point_list);
  }
}

template <bool kBig>
__global__ void __launch_bounds__(1024)
bin_sort_kernel(int T, unsigned idx_bits, const uint32_t* __restrict__ tile_order, const uint2* __restrict__ ranges,
                uint64_t* __restrict__ keys_unsorted, uint64_t* __restrict__ keys_out, uint32_t* __restrict__ point_list) {
  __shared__ union SortStorage {
    SortS4 s4; SortS8 s8;
    char big[kBig ? sizeof(SortS16) : 8];
    __device__ SortStorage() {}
  } storage;
  constexpr unsigned kSplit = 8192;  // kernel A: lists of 1 .. kSplit keys; kernel B: longer ones
  for (int ob = blockIdx.x; ob < T; ob += gridDim.x) {
    const uint32_t tile = tile_order[ob];
    const uint2 rg = ranges[tile];
    const unsigned n = rg.y - rg.x;
    if (kBig ? n <= kSplit : n <= kSmallList) break;  // tile_order is descending: nothing further for this kernel
    if (!kBig && n > kSplit) continue;
    uint64_t* src = keys_unsorted + rg.x;
    const uint64_t tile_hi = (uint64_t)tile << 32;
    if constexpr (kBig) {
      if (n <= 16384) {
        sort_tile_radix<16>(*reinterpret_cast<SortS16*>(storage.big), src, n, idx_bits, tile_hi, rg.x, keys_out, point_list);
      } else {
        const uint64_t idx_mask = (1ull << idx_bits) - 1;
        unsigned npad = 0;  // log2 of the padded length
        while ((1u << npad) < n) npad++;
        bitonic_sort_asc(n, npad, 1024u, [&](unsigned i) { return __builtin_nontemporal_load(src + i); },
                         [&](unsigned i, uint64_t v) { __builtin_nontemporal_store(v, src + i); });
        for (unsigned i = threadIdx.x; i < n; i += 1024) {
          const uint64_t kk = __builtin_nontemporal_load(src + i);
          point_list[rg.x + i] = (uint32_t)(kk & idx_mask);
          keys_out[rg.x + i] = tile_hi | (kk >> idx_bits);
        }
      }
    } else {
      if (n <= 4096) sort_tile_radix<4>(storage.s4, src, n, idx_bits, tile_hi, rg.x, keys_out, point_list);
      else sort_tile_radix<8>(storage.s8, src, n, idx_bits, tile_hi, rg.x, keys_out, point_list);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Dense scenes: long tile lists are partitioned by sampled splitters before the LDS sorts
// ------------------------------------------------------------------------------------------
// A scene that averages thousands of instances per tile (Mip-NeRF360 at images_4 with 3 M Gaussians: mean 7 000, median
// 1 900, 1 % of the tiles above 30 000) overflows what ONE workgroup sorts in LDS (8 192 keys at full speed, 16 384 at
// half), and the reference's answer -- one global 44-bit radix sort of all R pairs, six passes over 24 B, after reading R
// back -- costs 2.4 ms of a 9 ms step there and keeps the host in the loop.  Here the scatter by tile already IS the
// most significant radix pass; a tile whose list is long (> kLongList keys) gets ONE more most-significant pass, with
// splitters drawn from its own keys (a sample sort): after the scatter a tile's stretch of `keys_unsorted` is in
// arbitrary order, so every (n / 512)-th key is a uniform sample; its sorted sample yields B - 1 = ceil(n / 1536) - 1
// splitters, the tile's keys are partitioned into B buckets (histogram per 4096-key chunk, prefix, scatter into the
// tile's stretch of the output array -- runs of hundreds of keys: coalesced), and every bucket is sorted in place by the
// same LDS kernels as a short list.  Buckets are key ranges, keys are unique and totally ordered by (depth, index), so
// the concatenation is the tile's list exactly as the reference's stable sort produces it, wherever the splitters fall;
// they only balance the work (a bucket that comes out long takes the 8- / 16-keys-per-lane kernels or, beyond 16 384
// keys, the global-memory network: slower, never wrong).  No host anywhere: work lists and counts stay on the device.
constexpr unsigned kLongList = 8192;     // lists above this are partitioned (below: sorted directly, as in sparse scenes)
constexpr unsigned kBucketTarget = 1536;  // keys per bucket aimed at (the 256-lane LDS sort takes up to 2048)
constexpr unsigned kMaxBuckets = 32;
constexpr unsigned kPartChunk = 4096;    // keys per workgroup pass of the histogram / partition kernels
// splitter sample = 256 lanes x this many keys: 1024 / 512 / 256 samples measured at C4 -- plan 0.083 / 0.063 / 0.054 ms, but with
// 256 the buckets balance worse (the 1024-lane bucket sort grows from 0.02 to 0.07 ms)
#ifndef GIGS_SAMPLE_PER_LANE
#define GIGS_SAMPLE_PER_LANE 2
#endif
constexpr int kSamplePerLane = GIGS_SAMPLE_PER_LANE;
constexpr unsigned kSampleMax = 256 * kSamplePerLane;

struct LongTile { uint32_t tile, B, first_chunk, nchunks; };
struct LongState {
  uint32_t* counts;        // [0] long tiles, [1] chunks, [4..7] buckets by length class
  LongTile* tiles;         // [T]
  uint64_t* splitters;     // [T][kMaxBuckets] (B - 1 used)
  uint32_t* bucket_start;  // [T][kMaxBuckets] offsets inside the tile's stretch
  uint2* chunk_desc;       // [max_chunks] (slot in `tiles`, chunk index inside the tile)
  uint32_t* chunk_hist;    // [max_chunks][kMaxBuckets] keys of the chunk per bucket, then their exclusive prefix over the tile's chunks
  uint4* cls_list;         // [4][max_buckets] (start, n, tile, 0) by length class: <= 2048, <= 8192, <= 16384, longer
  size_t max_chunks, max_buckets;
  static LongState fromChunk(char*& chunk, size_t T, size_t R) {
    LongState st;
    st.max_chunks = R / kPartChunk + T + 1;
    st.max_buckets = (R / kLongList + 1) * kMaxBuckets;  // at most R / kLongList long tiles, kMaxBuckets buckets each
    carve(chunk, st.counts, 8);
    carve(chunk, st.tiles, T);
    carve(chunk, st.splitters, T * kMaxBuckets);
    carve(chunk, st.bucket_start, T * kMaxBuckets);
    carve(chunk, st.chunk_desc, st.max_chunks);
    carve(chunk, st.chunk_hist, st.max_chunks * kMaxBuckets);
    carve(chunk, st.cls_list, 4 * st.max_buckets);
    return st;
  }
};
size_t long_space_bytes(size_t T, size_t R) {
  char* p = nullptr;
  (void)LongState::fromChunk(p, T, R);
  return reinterpret_cast<size_t>(p) + kAlign;
}

__device__ __forceinline__ unsigned buckets_for(unsigned n, unsigned target) { return min(kMaxBuckets, (n + target - 1) / target); }

// one workgroup per tile (most return at once): sample, sort the sample, publish splitters and the chunk descriptors
using SampleSort = rocprim::block_radix_sort<uint64_t, 256, kSamplePerLane, rocprim::empty_type, 1, 1, 8>;
__global__ void __launch_bounds__(256)
long_plan_kernel(int T, unsigned idx_bits, unsigned target, const uint2* __restrict__ ranges,
                 const uint64_t* __restrict__ keys_unsorted, LongState st) {
  __shared__ SampleSort::storage_type storage;
  __shared__ uint64_t s_sorted[kSampleMax];
  __shared__ uint32_t s_slot, s_first;
  for (int tile = blockIdx.x; tile < T; tile += gridDim.x) {
    const uint2 rg = ranges[tile];
    const unsigned n = rg.y - rg.x;
    if (n <= kLongList) continue;
    const unsigned B = buckets_for(n, target), nch = (n + kPartChunk - 1) / kPartChunk;
    if (threadIdx.x == 0) {
      s_slot = atomicAdd(&st.counts[0], 1u);
      s_first = atomicAdd(&st.counts[1], nch);
    }
    // a uniform sample of the tile's keys (their order after the scatter is arbitrary): every (n / S)-th one
    const unsigned S = kSampleMax;  // n > kLongList > S
    uint64_t k[kSamplePerLane];
#pragma unroll
    for (int i = 0; i < kSamplePerLane; i++) {
      const unsigned j = threadIdx.x * kSamplePerLane + i;
      k[i] = keys_unsorted[rg.x + (unsigned)(((unsigned long long)j * n) / S)];
    }
    SampleSort().sort(k, storage, 0, 32 + idx_bits);
#pragma unroll
    for (int i = 0; i < kSamplePerLane; i++) s_sorted[threadIdx.x * kSamplePerLane + i] = k[i];
    __syncthreads();
    const unsigned slot = s_slot, first = s_first;
    if (threadIdx.x < kMaxBuckets) {
      const unsigned j = threadIdx.x;  // splitter j closes bucket j: bucket(x) = #{j < B - 1 : splitter[j] <= x}
      st.splitters[(size_t)slot * kMaxBuckets + j] = (j + 1 < B) ? s_sorted[((j + 1) * S) / B] : ~0ull;
    }
    if (threadIdx.x == 0) st.tiles[slot] = LongTile{(uint32_t)tile, B, first, nch};
    for (unsigned c = threadIdx.x; c < nch; c += 256) st.chunk_desc[first + c] = make_uint2(slot, c);
    __syncthreads();
  }
}

__device__ __forceinline__ unsigned bucket_of(uint64_t x, const uint64_t* __restrict__ s_split) {
  // upper bound over the 31 splitter slots (unused ones are ~0 > every key): 5 steps
  unsigned lo = 0;
#pragma unroll
  for (unsigned step = 16; step; step >>= 1)
    if (s_split[lo + step - 1] <= x) lo += step;
  return lo;
}

// histogram (kScatter = false) and partition (kScatter = true) over the 4096-key chunks of the long tiles
template <bool kScatter>
__global__ void __launch_bounds__(256)
long_chunks_kernel(const uint2* __restrict__ ranges, const uint64_t* __restrict__ keys_unsorted, uint64_t* __restrict__ keys_tmp,
                   LongState st) {
  __shared__ uint64_t s_split[kMaxBuckets];
  __shared__ uint32_t s_cnt[kMaxBuckets];
  const unsigned nchunks = st.counts[1];
  for (unsigned ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const uint2 d = st.chunk_desc[ch];
    const LongTile lt = st.tiles[d.x];
    const uint2 rg = ranges[lt.tile];
    const unsigned n = rg.y - rg.x, c0 = d.y * kPartChunk, c1 = min(n, c0 + kPartChunk);
    if (threadIdx.x < kMaxBuckets) {
      s_split[threadIdx.x] = st.splitters[(size_t)d.x * kMaxBuckets + threadIdx.x];
      // partition: the bucket's start inside the tile + what the tile's earlier chunks put into it
      s_cnt[threadIdx.x] = kScatter ? st.bucket_start[(size_t)d.x * kMaxBuckets + threadIdx.x] +
                                          st.chunk_hist[(size_t)ch * kMaxBuckets + threadIdx.x]
                                    : 0u;
    }
    __syncthreads();
    const uint64_t* src = keys_unsorted + rg.x;
    for (unsigned i = c0 + threadIdx.x; i < c1; i += 256) {
      const uint64_t x = src[i];
      const unsigned b = bucket_of(x, s_split);
      const unsigned pos = atomicAdd(&s_cnt[b], 1u);
      if (kScatter) keys_tmp[rg.x + pos] = x;
    }
    __syncthreads();
    if (!kScatter && threadIdx.x < kMaxBuckets) st.chunk_hist[(size_t)ch * kMaxBuckets + threadIdx.x] = s_cnt[threadIdx.x];
    __syncthreads();
  }
}

// per long tile (one wave): prefix of the chunk histograms over the tile's chunks, bucket starts, and the buckets listed by
// length class for the sort kernels
__global__ void __launch_bounds__(64)
long_prefix_kernel(const uint2* __restrict__ ranges, LongState st) {
  const unsigned nlong = st.counts[0];
  for (unsigned slot = blockIdx.x; slot < nlong; slot += gridDim.x) {
    const LongTile lt = st.tiles[slot];
    const uint2 rg = ranges[lt.tile];
    const unsigned b = threadIdx.x;
    uint32_t run = 0;
    if (b < kMaxBuckets)
      for (unsigned c = 0; c < lt.nchunks; c++) {
        uint32_t* h = st.chunk_hist + (size_t)(lt.first_chunk + c) * kMaxBuckets + b;
        const uint32_t v = *h;
        *h = run;
        run += v;
      }
    // exclusive scan of the bucket totals over the 32 bucket lanes
    uint32_t incl = run;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      const uint32_t v = __shfl_up(incl, off);
      if ((int)(b & 31) >= off) incl += v;
    }
    const uint32_t start = incl - run;
    if (b < kMaxBuckets) st.bucket_start[(size_t)slot * kMaxBuckets + b] = start;
    // append the non-empty buckets to their length class's list: one atomic per (tile, class), not per bucket
    const int cls = (b < kMaxBuckets && run) ? (run <= 2048u ? 0 : run <= 8192u ? 1 : run <= 16384u ? 2 : 3) : -1;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const unsigned long long m = __builtin_amdgcn_ballot_w64(cls == c);
      if (m == 0) continue;
      uint32_t base = 0;
      if (threadIdx.x == 0) base = atomicAdd(&st.counts[4 + c], (uint32_t)__popcll(m));
      base = __shfl(base, 0);
      if (cls == c)
        st.cls_list[(size_t)c * st.max_buckets + base + (uint32_t)__popcll(m & ((1ull << threadIdx.x) - 1))] =
            make_uint4(rg.x + start, run, lt.tile, 0u);
    }
  }
}

// sorts of the buckets (in place in keys_out, where the partition left them)
__global__ void __launch_bounds__(256)
long_sort_small_kernel(unsigned idx_bits, LongState st, uint64_t* __restrict__ keys_out, uint32_t* __restrict__ point_list) {
  __shared__ SortSmall storage;
  const unsigned count = st.counts[4];
  for (unsigned i = blockIdx.x; i < count; i += gridDim.x) {
    const uint4 d = st.cls_list[i];
    sort_tile_radix<8, SortSmall, 256>(storage, keys_out + d.x, d.y, idx_bits, (uint64_t)d.z << 32, d.x, keys_out, point_list);
  }
}

// kHuge = false: buckets of 2049 .. 8192 keys (64 KB of LDS, two workgroups per CU); true: the longer ones (132 KB)
template <bool kHuge>
__global__ void __launch_bounds__(1024)
long_sort_kernel(unsigned idx_bits, LongState st, uint64_t* __restrict__ keys_out, uint32_t* __restrict__ point_list) {
  __shared__ union SortStorage {
    SortS4 s4; SortS8 s8;
    char big[kHuge ? sizeof(SortS16) : 8];
    __device__ SortStorage() {}
  } storage;
  const unsigned n2 = kHuge ? st.counts[6] : 0u, count = kHuge ? n2 + st.counts[7] : st.counts[5];
  for (unsigned i = blockIdx.x; i < count; i += gridDim.x) {
    const uint4 d = !kHuge ? st.cls_list[st.max_buckets + i]
                           : i < n2 ? st.cls_list[2 * st.max_buckets + i] : st.cls_list[3 * st.max_buckets + (i - n2)];
    const unsigned n = d.y;
    uint64_t* src = keys_out + d.x;
    const uint64_t tile_hi = (uint64_t)d.z << 32;
    if constexpr (!kHuge) {
      if (n <= 4096) sort_tile_radix<4>(storage.s4, src, n, idx_bits, tile_hi, d.x, keys_out, point_list);
      else sort_tile_radix<8>(storage.s8, src, n, idx_bits, tile_hi, d.x, keys_out, point_list);
    } else if (n <= 16384) {
      sort_tile_radix<16>(*reinterpret_cast<SortS16*>(storage.big), src, n, idx_bits, tile_hi, d.x, keys_out, point_list);
    } else {
      const uint64_t idx_mask = (1ull << idx_bits) - 1;
      unsigned npad = 0;
      while ((1u << npad) < n) npad++;
      bitonic_sort_asc(n, npad, 1024u, [&](unsigned k) { return __builtin_nontemporal_load(src + k); },
                       [&](unsigned k, uint64_t v) { __builtin_nontemporal_store(v, src + k); });
      for (unsigned k = threadIdx.x; k < n; k += 1024) {
        const uint64_t kk = __builtin_nontemporal_load(src + k);
        point_list[d.x + k] = (uint32_t)(kk & idx_mask);
        keys_out[d.x + k] = tile_hi | (kk >> idx_bits);
      }
      __syncthreads();
    }
  }
}

__global__ void __launch_bounds__(256) zero_words_kernel(uint32_t* __restrict__ p, size_t n) {
  // n is a multiple of 4 for every caller's layout except a short tail
  const size_t n4 = n / 4, stride = (size_t)gridDim.x * 256;
  uint4* p4 = reinterpret_cast<uint4*>(p);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) p4[i] = make_uint4(0, 0, 0, 0);
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[4 * n4 + threadIdx.x] = 0;
}
void launch_zero_words(uint32_t* p, size_t n, hipStream_t s) {
  if (n == 0) return;
  const unsigned blocks = (unsigned)std::min<size_t>((n / 4 + 255) / 256 + 1, 2048);
  hipLaunchKernelGGL(zero_words_kernel, dim3(blocks), dim3(256), 0, s, p, n);
}

static unsigned bin_index_bits(int P) {  // bits that hold every Gaussian index
  unsigned ib = 1;
  while (ib < 32 && (1ull << ib) < (unsigned long long)P) ib++;
  return ib;
}

static int bin_groups(int P) { return std::min((int)kBinGroups, (P + 255) / 256); }
constexpr int kWideWalk = 1 << 20;  // clouds from this size on are walked by 1024-lane workgroups

void launch_bin_count(int P, const int* radii, unsigned gx, unsigned gy, const GeomState& g, const ImageState& img, hipStream_t s) {
  const int T = (int)(gx * gy);
  if (P >= kWideWalk)
    hipLaunchKernelGGL(bin_count_kernel<1024>, dim3(bin_groups(P)), dim3(1024), (size_t)T * sizeof(uint32_t), s, P, T, radii,
                       g.means2D, gx, gy, img.bin_hist);
  else
    hipLaunchKernelGGL(bin_count_kernel<256>, dim3(bin_groups(P)), dim3(256), (size_t)T * sizeof(uint32_t), s, P, T, radii,
                       g.means2D, gx, gy, img.bin_hist);
}

// the tile totals and the unclamped tile starts live in the two extra rows of bin_hist (ImageState carves kBinGroups + 2)
void launch_bin_prefix(int P, int T, unsigned capacity, const ImageState& img, unsigned* user_counters, hipStream_t s) {
  uint32_t* totals = img.bin_hist + (size_t)(kBinGroups) * T;       // carved with two extra rows (ImageState)
  uint32_t* tile_start = img.bin_hist + (size_t)(kBinGroups + 1) * T;
  hipLaunchKernelGGL(bin_prefix_groups_kernel, dim3((T + 63) / 64), dim3(1024), 0, s, bin_groups(P), T, img.bin_hist, totals);
  hipLaunchKernelGGL(bin_prefix_tiles_kernel, dim3(1), dim3(1024), 0, s, T, capacity, (unsigned)P, totals, img.ranges, tile_start,
                     img.bin_counters, user_counters);
}

void launch_bin_scatter(int P, const int* radii, unsigned gx, unsigned gy, unsigned capacity, unsigned bands, const GeomState& g,
                        const BinningState& b, const ImageState& img, hipStream_t s) {
  const int T = (int)(gx * gy);
  const uint32_t* tile_start = img.bin_hist + (size_t)(kBinGroups + 1) * T;
  bands = std::max(1u, std::min(bands, gy));
  if (P >= kWideWalk)
    hipLaunchKernelGGL(bin_scatter_kernel<1024>, dim3(bin_groups(P)), dim3(1024), (size_t)T * sizeof(uint32_t), s, P, T, radii,
                       g.means2D, g.depths, gx, gy, capacity, bin_index_bits(P), img.bin_hist, tile_start, b.keys_unsorted, bands);
  else
    hipLaunchKernelGGL(bin_scatter_kernel<256>, dim3(bin_groups(P)), dim3(256), (size_t)T * sizeof(uint32_t), s, P, T, radii,
                       g.means2D, g.depths, gx, gy, capacity, bin_index_bits(P), img.bin_hist, tile_start, b.keys_unsorted, bands);
}

// ---- dense scenes: the long lists' partition + bucket sorts (tables in the binning chunk's sort_space, which the
// bucketed path does not otherwise use; sort_size_cached() covers both uses)
void launch_long_lists(int T, int P, unsigned target, unsigned capacity, const BinningState& b, const ImageState& img, hipStream_t s) {
  char* space = b.sort_space;
  LongState st = LongState::fromChunk(space, (size_t)T, (size_t)capacity);
  const unsigned ib = bin_index_bits(P);
  launch_zero_words(st.counts, 8, s);
  if (target < 256u) target = kBucketTarget;  // gigs_options.bucket_target
  hipLaunchKernelGGL(long_plan_kernel, dim3(std::min(T, 1024)), dim3(256), 0, s, T, ib, target, img.ranges, b.keys_unsorted, st);
  hipLaunchKernelGGL(long_chunks_kernel<false>, dim3(2048), dim3(256), 0, s, img.ranges, b.keys_unsorted, b.keys, st);
  hipLaunchKernelGGL(long_prefix_kernel, dim3(std::min(T, 1024)), dim3(64), 0, s, img.ranges, st);
  hipLaunchKernelGGL(long_chunks_kernel<true>, dim3(2048), dim3(256), 0, s, img.ranges, b.keys_unsorted, b.keys, st);
  hipLaunchKernelGGL(long_sort_kernel<true>, dim3(256), dim3(1024), 0, s, ib, st, b.keys, b.point_list);
  hipLaunchKernelGGL(long_sort_kernel<false>, dim3(512), dim3(1024), 0, s, ib, st, b.keys, b.point_list);
  hipLaunchKernelGGL(long_sort_small_kernel, dim3(2048), dim3(256), 0, s, ib, st, b.keys, b.point_list);
}

int launch_bin_sort(int T, int P, bool long_lists, unsigned bucket_target, unsigned capacity, const BinningState& b, const ImageState& img,
                    hipStream_t s) {
  // (Running size classes concurrently on forked streams was tried: with the light's side stream and two sort streams
  // the runtime ran out of hardware queues and folded the light filter onto the main queue -- 25 % slower.  Hence one
  // kernel that holds lists of every length up to 8192 keys, and a second for the longer ones: in a sparse scene it
  // sorts them whole (usually there are none), in a dense one they are partitioned first (launch_long_lists).)
  const unsigned ib = bin_index_bits(P);
  if (long_lists)
    launch_long_lists(T, P, bucket_target, capacity, b, img, s);
  else
    hipLaunchKernelGGL(bin_sort_kernel<true>, dim3(std::min(T, 256)), dim3(1024), 0, s, T, ib, img.tile_order, img.ranges,
                       b.keys_unsorted, b.keys, b.point_list);
  hipLaunchKernelGGL(bin_sort_kernel<false>, dim3(std::min(T, 512)), dim3(1024), 0, s, T, ib, img.tile_order, img.ranges,
                     b.keys_unsorted, b.keys, b.point_list);
  hipLaunchKernelGGL(bin_sort_small_kernel, dim3(std::min(T, 2048)), dim3(256), 0, s, T, ib, img.tile_order, img.ranges,
                     b.keys_unsorted, b.keys, b.point_list);
  return 0;
}

}  // namespace gigs
