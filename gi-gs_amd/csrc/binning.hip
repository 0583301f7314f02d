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
__device__ __forceinline__ bool bin_rect(int idx, int end, const int* __restrict__ radii, const float* __restrict__ means2D,
                                         unsigned gx, unsigned gy, unsigned& minx, unsigned& miny, unsigned& w, unsigned& n) {
  minx = miny = w = n = 0;
  if (idx >= end) return false;
  const int r = radii[idx];
  if (r <= 0) return false;
  const float2 xy = reinterpret_cast<const float2*>(means2D)[idx];
  unsigned maxx, maxy;
  tile_rect(xy.x, xy.y, r, gx, gy, minx, miny, maxx, maxy);
  w = maxx - minx;
  n = w * (maxy - miny);
  return n > 0;
}

constexpr unsigned kBinWaveExpand = 12;  // footprints with more tiles than this are expanded by the whole wave

// Walks the tiles of the chunk's Gaussians; f(idx, tile, payload) is called once per (Gaussian, tile) pair, with
// payload = pay(idx) evaluated once per Gaussian (the split path's depth bucket; the plain path passes nothing).
template <typename Pay, typename F>
__device__ __forceinline__ void bin_walk(int P, const int* __restrict__ radii, const float* __restrict__ means2D, unsigned gx,
                                         unsigned gy, Pay pay, F f) {
  const int groups = min((int)kBinGroups, (P + 255) / 256);
  const int chunk = (P + groups - 1) / groups;
  const int begin = blockIdx.x * chunk, end = min(P, begin + chunk);
  const int lane = threadIdx.x & 63;
  for (int base = begin; base < end; base += blockDim.x) {
    const int idx = base + threadIdx.x;
    unsigned minx, miny, w, n;
    const bool any = bin_rect(idx, end, radii, means2D, gx, gy, minx, miny, w, n);
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

__global__ void __launch_bounds__(256)
bin_count_kernel(int P, int T, const int* __restrict__ radii, const float* __restrict__ means2D, unsigned gx, unsigned gy,
                 uint32_t* __restrict__ bin_hist) {
  extern __shared__ uint32_t s_hist[];
  for (int t = threadIdx.x; t < T; t += 256) s_hist[t] = 0;
  __syncthreads();
  bin_walk(P, radii, means2D, gx, gy, NoPay(), [&](int, unsigned tile, unsigned) { atomicAdd(&s_hist[tile], 1u); });
  __syncthreads();
  uint32_t* row = bin_hist + (size_t)blockIdx.x * T;
  for (int t = threadIdx.x; t < T; t += 256) row[t] = s_hist[t];
}

// per tile: exclusive prefix over the chunk counts (in place) and the total
__global__ void __launch_bounds__(1024)
bin_prefix_groups_kernel(int groups, int T, uint32_t* __restrict__ bin_hist, uint32_t* __restrict__ totals) {
  // 64 tiles x 16 chunk-parts per workgroup: coalesced across tiles, sixteen independent short serial scans per tile
  __shared__ uint32_t s_q[16][64];
  const int tl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + tl;
  const int per = (groups + 15) / 16, g0 = min(groups, q * per), g1 = min(groups, g0 + per);
  uint32_t c[16];
  uint32_t run = 0;
  if (t < T) {
#pragma unroll
    for (int k = 0; k < 16; k++) c[k] = (g0 + k < g1) ? bin_hist[(size_t)(g0 + k) * T + t] : 0u;  // per <= 16 (kBinGroups = 256)
#pragma unroll
    for (int k = 0; k < 16; k++) { const uint32_t v = c[k]; c[k] = run; run += v; }
  }
  s_q[q][tl] = run;
  __syncthreads();
  uint32_t add = 0;
  for (int k = 0; k < q; k++) add += s_q[k][tl];
  if (t < T) {
#pragma unroll
    for (int k = 0; k < 16; k++)
      if (g0 + k < g1) bin_hist[(size_t)(g0 + k) * T + t] = c[k] + add;
    if (q == 15) totals[t] = add + run;
  }
}

// exclusive scan of the tile totals -> ranges (clamped to the capacity), R and the overflow flag.  One workgroup.
__global__ void __launch_bounds__(1024)
bin_prefix_tiles_kernel(int T, unsigned capacity, const uint32_t* __restrict__ totals, uint2* __restrict__ ranges,
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
    if (user_counters) { user_counters[0] = R; user_counters[1] = R > capacity ? R : 0u; }
  }
}

__global__ void __launch_bounds__(256)
bin_scatter_kernel(int P, int T, const int* __restrict__ radii, const float* __restrict__ means2D,
                   const float* __restrict__ depths, unsigned gx, unsigned gy, unsigned capacity, unsigned idx_bits,
                   const uint32_t* __restrict__ bin_hist, const uint32_t* __restrict__ tile_start,
                   uint64_t* __restrict__ keys) {
  extern __shared__ uint32_t s_cur[];
  const uint32_t* row = bin_hist + (size_t)blockIdx.x * T;
  for (int t = threadIdx.x; t < T; t += 256) s_cur[t] = tile_start[t] + row[t];
  __syncthreads();
  bin_walk(P, radii, means2D, gx, gy, NoPay(), [&](int idx, unsigned tile, unsigned) {
    const uint32_t slot = atomicAdd(&s_cur[tile], 1u);
    if (slot < capacity) keys[slot] = ((uint64_t)__float_as_uint(depths[idx]) << idx_bits) | (uint32_t)idx;
  });
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
  const uint64_t idx_mask = (1ull << idx_bits) - 1;
  uint64_t k[kItems];
#pragma unroll
  for (int i = 0; i < kItems; i++) {
    const unsigned idx = threadIdx.x * kItems + i;
    k[i] = idx < n ? src[idx] : ~0ull;  // padding sorts to the end (all ones in every sorted bit)
  }
  sorter().sort(k, storage, 0, 32 + idx_bits);  // keys are depth bits << idx_bits | index: one contiguous field
#pragma unroll
  for (int i = 0; i < kItems; i++) {
    const unsigned idx = threadIdx.x * kItems + i;
    if (idx < n) {
      point_list[base + idx] = (uint32_t)(k[i] & idx_mask);
      keys_out[base + idx] = tile_hi | (k[i] >> idx_bits);
    }
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
// Dense scenes: depth-split bins (the tile-bucketed binning with B depth buckets per tile)
// ------------------------------------------------------------------------------------------
// A scene that averages thousands of instances per tile (Mip-NeRF360 at images_4 with 3 M Gaussians: 7 000, single tiles
// beyond 30 000) overflows what ONE workgroup sorts in LDS (8 192 keys at full speed, 16 384 at half), and the reference's
// answer -- one global 44-bit radix sort of all R pairs, six passes over 24 B, after reading R back -- costs 2.4 ms of a
// 9 ms step there and keeps the host in the loop.  Here the first scatter already IS the most significant radix pass
// (by tile); it is widened to (tile, depth bucket): B <= 8 buckets per tile, bounded by B - 1 depth thresholds that are
// the instance-weighted B-quantiles of the frame's depth distribution (a 8192-bin histogram over the Gaussians -- 12 B
// per Gaussian, not per instance -- and a one-workgroup scan; no host).  Bin (tile * B + b) is a contiguous stretch of
// the tile's range, buckets are in depth order, so sorting every bin's keys by (depth, index) IN PLACE yields the
// tile's list exactly as the reference's stable sort does: keys are unique, the order is total, and it does not matter
// where the thresholds fall -- they only balance the work.  Sub-lists are a B-th of the tile's list on average, so
// nearly all of them take the 256-lane LDS sort; a sub-list that is still long (a tile whose depths cluster) takes the
// 8- or 16-keys-per-lane kernels or, beyond 16 384 keys, the global-memory network: slower, never wrong.
constexpr int kDepthBins = 8192;      // 16 octaves [2^-3, 2^13) x 512 bins: the top 18 bits of the depth's float bits
constexpr int kDepthShift = 14;
constexpr int kDepthRows = 64;        // workgroups (and rows of partial histograms) of the depth histogram
constexpr unsigned kDepthBase = 0x3E000000u >> kDepthShift;  // float bits of 0.125 (the cull keeps z > 0.2)
__device__ __forceinline__ unsigned depth_bin(float d) {
  const int k = (int)(__float_as_uint(d) >> kDepthShift) - (int)kDepthBase;
  return (unsigned)min(max(k, 0), kDepthBins - 1);
}

struct SplitState {
  uint32_t* hist;        // [kBinGroups][NB] instances of chunk g in bin i, then (in place) their exclusive prefix over g
  uint32_t* totals;      // [NB]
  uint32_t* bin_start;   // [NB + 1] exclusive prefix of the totals (unclamped)
  uint2* sub_ranges;     // [NB] a bin's stretch of the instance arrays, clamped to the capacity
  uint32_t* cls_list;    // [4][NB] bins by length class: <= 2048, <= 8192, <= 16384, longer
  uint32_t* cls_count;   // [4]
  uint32_t* splits;      // [8] depth-bin thresholds: bucket = #{k : depth_bin >= splits[k]}
  uint32_t* depth_hist;  // [kDepthRows][kDepthBins]
  static SplitState fromChunk(char* chunk, size_t NB) {
    SplitState st;
    carve(chunk, st.hist, (size_t)kBinGroups * NB);
    carve(chunk, st.totals, NB);
    carve(chunk, st.bin_start, NB + 1);
    carve(chunk, st.sub_ranges, NB);
    carve(chunk, st.cls_list, 4 * NB);
    carve(chunk, st.cls_count, 4);
    carve(chunk, st.splits, 8);
    carve(chunk, st.depth_hist, (size_t)kDepthRows * kDepthBins);
    return st;
  }
};
size_t split_space_bytes(size_t NB) {
  return ((size_t)kBinGroups + 8) * NB * sizeof(uint32_t) + (size_t)kDepthRows * kDepthBins * sizeof(uint32_t) + 16 * kAlign;
}

__global__ void __launch_bounds__(1024)
split_depth_hist_kernel(int P, const int* __restrict__ radii, const float* __restrict__ depths,
                        const uint32_t* __restrict__ tiles_touched, uint32_t* __restrict__ depth_hist) {
  __shared__ uint32_t s_h[kDepthBins];
  for (int i = threadIdx.x; i < kDepthBins; i += 1024) s_h[i] = 0;
  __syncthreads();
  const int chunk = (P + kDepthRows - 1) / kDepthRows;
  const int begin = blockIdx.x * chunk, end = min(P, begin + chunk);
  for (int idx = begin + threadIdx.x; idx < end; idx += 1024)
    if (radii[idx] > 0) atomicAdd(&s_h[depth_bin(depths[idx])], tiles_touched[idx]);
  __syncthreads();
  uint32_t* row = depth_hist + (size_t)blockIdx.x * kDepthBins;
  for (int i = threadIdx.x; i < kDepthBins; i += 1024) row[i] = s_h[i];
}

// instance-weighted B-quantiles of the depth histogram -> splits[0 .. B-2]; the rest never matches.  One workgroup.
__global__ void __launch_bounds__(1024)
split_pick_kernel(unsigned B, const uint32_t* __restrict__ depth_hist, uint32_t* __restrict__ splits) {
  __shared__ unsigned long long s_sum[1024];
  constexpr int kPer = kDepthBins / 1024;
  unsigned long long c[kPer], local = 0;
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    unsigned long long v = 0;
    for (int r = 0; r < kDepthRows; r++) v += depth_hist[(size_t)r * kDepthBins + threadIdx.x * kPer + k];
    c[k] = v;
    local += v;
  }
  s_sum[threadIdx.x] = local;
  if (threadIdx.x < 8) splits[threadIdx.x] = 0xFFFFFFFFu;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const unsigned long long v = threadIdx.x >= (unsigned)off ? s_sum[threadIdx.x - off] : 0ull;
    __syncthreads();
    s_sum[threadIdx.x] += v;
    __syncthreads();
  }
  const unsigned long long W = s_sum[1023];
  unsigned long long run = s_sum[threadIdx.x] - local;
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    const unsigned long long incl = run + c[k];
    for (unsigned q = 1; q < B; q++) {
      const unsigned long long target = (W * q) / B;
      // the first bin whose inclusive count reaches the q-th quantile closes bucket q - 1
      if (run < target && target <= incl) splits[q - 1] = (uint32_t)(threadIdx.x * kPer + k + 1);
    }
    run = incl;
  }
}

struct DepthBucket {
  const float* __restrict__ depths;
  uint32_t sp[7];
  __device__ __forceinline__ unsigned operator()(int idx) const {
    const unsigned db = depth_bin(depths[idx]);
    unsigned b = 0;
#pragma unroll
    for (int k = 0; k < 7; k++) b += db >= sp[k] ? 1u : 0u;
    return b;
  }
};
__device__ __forceinline__ DepthBucket load_buckets(const float* depths, const uint32_t* __restrict__ splits) {
  DepthBucket d;
  d.depths = depths;
#pragma unroll
  for (int k = 0; k < 7; k++) d.sp[k] = splits[k];
  return d;
}

__global__ void __launch_bounds__(1024)
split_count_kernel(int P, int NB, unsigned B, const int* __restrict__ radii, const float* __restrict__ means2D,
                   const float* __restrict__ depths, unsigned gx, unsigned gy, const uint32_t* __restrict__ splits,
                   uint32_t* __restrict__ hist) {
  extern __shared__ uint32_t s_hist[];
  for (int i = threadIdx.x; i < NB; i += 1024) s_hist[i] = 0;
  __syncthreads();
  bin_walk(P, radii, means2D, gx, gy, load_buckets(depths, splits),
           [&](int, unsigned tile, unsigned b) { atomicAdd(&s_hist[tile * B + b], 1u); });
  __syncthreads();
  uint32_t* row = hist + (size_t)blockIdx.x * NB;
  for (int i = threadIdx.x; i < NB; i += 1024) row[i] = s_hist[i];
}

// exclusive scan of the bin totals -> bin starts, clamped sub-ranges, the tiles' ranges, R / overflow, and the bins
// listed by length class for the sort kernels.  One workgroup.
__global__ void __launch_bounds__(1024)
split_prefix_bins_kernel(int NB, unsigned B, int T, unsigned capacity, const uint32_t* __restrict__ totals,
                         uint32_t* __restrict__ bin_start, uint2* __restrict__ sub_ranges, uint2* __restrict__ ranges,
                         uint32_t* __restrict__ cls_list, uint32_t* __restrict__ cls_count,
                         uint32_t* __restrict__ counters, uint32_t* __restrict__ user_counters) {
  __shared__ uint32_t s_sum[1024];
  __shared__ uint32_t s_cnt[4];
  const int per = (NB + 1023) / 1024;
  const int i0 = min(NB, (int)threadIdx.x * per), i1 = min(NB, i0 + per);
  uint32_t local = 0;
  for (int i = i0; i < i1; i++) local += totals[i];
  s_sum[threadIdx.x] = local;
  if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const uint32_t v = threadIdx.x >= (unsigned)off ? s_sum[threadIdx.x - off] : 0u;
    __syncthreads();
    s_sum[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = s_sum[threadIdx.x] - local;
  for (int i = i0; i < i1; i++) {
    const uint32_t c = totals[i];
    bin_start[i] = run;
    const uint32_t a = min(run, capacity), e = min(run + c, capacity);
    sub_ranges[i] = make_uint2(a, e);
    const uint32_t n = e - a;
    if (n) {
      const int cls = n <= 2048u ? 0 : n <= 8192u ? 1 : n <= 16384u ? 2 : 3;
      cls_list[(size_t)cls * NB + atomicAdd(&s_cnt[cls], 1u)] = (uint32_t)i;
    }
    run += c;
  }
  const uint32_t R = s_sum[1023];
  if (threadIdx.x == 1023) {
    bin_start[NB] = R;
    counters[0] = R;
    counters[1] = R > capacity ? R : 0u;
    if (user_counters) { user_counters[0] = R; user_counters[1] = R > capacity ? R : 0u; }
  }
  __syncthreads();  // bin_start is complete (and visible to this workgroup)
  if (threadIdx.x < 4) cls_count[threadIdx.x] = s_cnt[threadIdx.x];
  for (int t = threadIdx.x; t < T; t += 1024) {
    const uint32_t a = bin_start[(size_t)t * B], e = bin_start[(size_t)(t + 1) * B];
    // an empty tile keeps (0, 0), what the reference's memset leaves there (rasterizer_impl.cu:621)
    ranges[t] = e > a ? make_uint2(min(a, capacity), min(e, capacity)) : make_uint2(0u, 0u);
  }
}

__global__ void __launch_bounds__(1024)
split_scatter_kernel(int P, int NB, unsigned B, const int* __restrict__ radii, const float* __restrict__ means2D,
                     const float* __restrict__ depths, unsigned gx, unsigned gy, unsigned capacity, unsigned idx_bits,
                     const uint32_t* __restrict__ splits, const uint32_t* __restrict__ hist,
                     const uint32_t* __restrict__ bin_start, uint64_t* __restrict__ keys) {
  extern __shared__ uint32_t s_cur[];
  const uint32_t* row = hist + (size_t)blockIdx.x * NB;
  for (int i = threadIdx.x; i < NB; i += 1024) s_cur[i] = bin_start[i] + row[i];
  __syncthreads();
  bin_walk(P, radii, means2D, gx, gy, load_buckets(depths, splits), [&](int idx, unsigned tile, unsigned b) {
    const uint32_t slot = atomicAdd(&s_cur[tile * B + b], 1u);
    if (slot < capacity) keys[slot] = ((uint64_t)__float_as_uint(depths[idx]) << idx_bits) | (uint32_t)idx;
  });
}

// the sort kernels of the split path: persistent grids over one length class's list of bins
__global__ void __launch_bounds__(256)
split_sort_small_kernel(int NB, unsigned B, unsigned idx_bits, const uint32_t* __restrict__ cls_list,
                        const uint32_t* __restrict__ cls_count, const uint2* __restrict__ sub_ranges,
                        const uint64_t* __restrict__ keys_unsorted, uint64_t* __restrict__ keys_out,
                        uint32_t* __restrict__ point_list) {
  __shared__ SortSmall storage;
  const unsigned count = cls_count[0];
  for (unsigned i = blockIdx.x; i < count; i += gridDim.x) {
    const uint32_t bin = cls_list[i];
    const uint2 rg = sub_ranges[bin];
    sort_tile_radix<8, SortSmall, 256>(storage, keys_unsorted + rg.x, rg.y - rg.x, idx_bits, (uint64_t)(bin / B) << 32, rg.x,
                                       keys_out, point_list);
  }
}

template <bool kBig>
__global__ void __launch_bounds__(1024)
split_sort_kernel(int NB, unsigned B, unsigned idx_bits, const uint32_t* __restrict__ cls_list,
                  const uint32_t* __restrict__ cls_count, const uint2* __restrict__ sub_ranges,
                  uint64_t* __restrict__ keys_unsorted, uint64_t* __restrict__ keys_out, uint32_t* __restrict__ point_list) {
  __shared__ union SortStorage {
    SortS4 s4; SortS8 s8;
    char big[kBig ? sizeof(SortS16) : 8];
    __device__ SortStorage() {}
  } storage;
  // kBig: classes 2 (<= 16384 keys) and 3 (longer) in one list walk; else class 1
  const unsigned n2 = kBig ? cls_count[2] : 0u, count = kBig ? n2 + cls_count[3] : cls_count[1];
  for (unsigned i = blockIdx.x; i < count; i += gridDim.x) {
    const uint32_t bin = kBig ? (i < n2 ? cls_list[(size_t)2 * NB + i] : cls_list[(size_t)3 * NB + (i - n2)])
                              : cls_list[(size_t)NB + i];
    const uint2 rg = sub_ranges[bin];
    const unsigned n = rg.y - rg.x;
    uint64_t* src = keys_unsorted + rg.x;
    const uint64_t tile_hi = (uint64_t)(bin / B) << 32;
    if constexpr (kBig) {
      if (n <= 16384) {
        sort_tile_radix<16>(*reinterpret_cast<SortS16*>(storage.big), src, n, idx_bits, tile_hi, rg.x, keys_out, point_list);
      } else {
        const uint64_t idx_mask = (1ull << idx_bits) - 1;
        unsigned npad = 0;
        while ((1u << npad) < n) npad++;
        bitonic_sort_asc(n, npad, 1024u, [&](unsigned k) { return __builtin_nontemporal_load(src + k); },
                         [&](unsigned k, uint64_t v) { __builtin_nontemporal_store(v, src + k); });
        for (unsigned k = threadIdx.x; k < n; k += 1024) {
          const uint64_t kk = __builtin_nontemporal_load(src + k);
          point_list[rg.x + k] = (uint32_t)(kk & idx_mask);
          keys_out[rg.x + k] = tile_hi | (kk >> idx_bits);
        }
        __syncthreads();
      }
    } else {
      if (n <= 4096) sort_tile_radix<4>(storage.s4, src, n, idx_bits, tile_hi, rg.x, keys_out, point_list);
      else sort_tile_radix<8>(storage.s8, src, n, idx_bits, tile_hi, rg.x, keys_out, point_list);
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

void launch_bin_count(int P, const int* radii, unsigned gx, unsigned gy, const GeomState& g, const ImageState& img, hipStream_t s) {
  const int T = (int)(gx * gy);
  hipLaunchKernelGGL(bin_count_kernel, dim3(bin_groups(P)), dim3(256), (size_t)T * sizeof(uint32_t), s, P, T, radii, g.means2D,
                     gx, gy, img.bin_hist);
}

// the tile totals and the unclamped tile starts live in the two extra rows of bin_hist (ImageState carves kBinGroups + 2)
void launch_bin_prefix(int P, int T, unsigned capacity, const ImageState& img, unsigned* user_counters, hipStream_t s) {
  uint32_t* totals = img.bin_hist + (size_t)(kBinGroups) * T;       // carved with two extra rows (ImageState)
  uint32_t* tile_start = img.bin_hist + (size_t)(kBinGroups + 1) * T;
  hipLaunchKernelGGL(bin_prefix_groups_kernel, dim3((T + 63) / 64), dim3(1024), 0, s, bin_groups(P), T, img.bin_hist, totals);
  hipLaunchKernelGGL(bin_prefix_tiles_kernel, dim3(1), dim3(1024), 0, s, T, capacity, totals, img.ranges, tile_start,
                     img.bin_counters, user_counters);
}

void launch_bin_scatter(int P, const int* radii, unsigned gx, unsigned gy, unsigned capacity, const GeomState& g,
                        const BinningState& b, const ImageState& img, hipStream_t s) {
  const int T = (int)(gx * gy);
  const uint32_t* tile_start = img.bin_hist + (size_t)(kBinGroups + 1) * T;
  hipLaunchKernelGGL(bin_scatter_kernel, dim3(bin_groups(P)), dim3(256), (size_t)T * sizeof(uint32_t), s, P, T, radii,
                     g.means2D, g.depths, gx, gy, capacity, bin_index_bits(P), img.bin_hist, tile_start, b.keys_unsorted);
}

// ---- split path launchers.  NB = T * B bins; the tables live in the binning chunk's sort_space (the split path
// does not use the global radix sort that space is sized for; sort_size_cached() covers both)
int split_buckets(size_t est_mean_list, size_t T) {
  // B such that a sub-list averages ~1000 keys: the 256-lane LDS sort's size; NB * 4 B of LDS per counting workgroup
  if (const char* e = getenv("GIGS_BIN_SPLIT")) {
    const int v = atoi(e);
    if (v == 1 || v == 2 || v == 4 || v == 8) return ((size_t)v * T <= (size_t)kSplitMaxBins) ? v : 1;
  }
  int B = 1;
  while (B < 8 && (size_t)B * 1024 < est_mean_list) B <<= 1;
  while (B > 1 && (size_t)B * T > (size_t)kSplitMaxBins) B >>= 1;
  return B;
}

static void split_lds_attr() {
  // more than 64 KB of dynamic LDS per workgroup has to be granted per kernel
  static const bool once = [] {
    const int bytes = kSplitMaxBins * (int)sizeof(uint32_t);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(split_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(split_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return true;
  }();
  (void)once;
}

void launch_split_count(int P, int T, int B, unsigned capacity, const int* radii, unsigned gx, unsigned gy, const GeomState& g,
                        const BinningState& b, const ImageState& img, unsigned* user_counters, hipStream_t s) {
  const int NB = T * B;
  SplitState st = SplitState::fromChunk(b.sort_space, (size_t)NB);
  split_lds_attr();
  hipLaunchKernelGGL(split_depth_hist_kernel, dim3(kDepthRows), dim3(1024), 0, s, P, radii, g.depths, g.tiles_touched,
                     st.depth_hist);
  hipLaunchKernelGGL(split_pick_kernel, dim3(1), dim3(1024), 0, s, (unsigned)B, st.depth_hist, st.splits);
  hipLaunchKernelGGL(split_count_kernel, dim3(bin_groups(P)), dim3(1024), (size_t)NB * sizeof(uint32_t), s, P, NB, (unsigned)B,
                     radii, g.means2D, g.depths, gx, gy, st.splits, st.hist);
  hipLaunchKernelGGL(bin_prefix_groups_kernel, dim3((NB + 63) / 64), dim3(1024), 0, s, bin_groups(P), NB, st.hist, st.totals);
  hipLaunchKernelGGL(split_prefix_bins_kernel, dim3(1), dim3(1024), 0, s, NB, (unsigned)B, T, capacity, st.totals, st.bin_start,
                     st.sub_ranges, img.ranges, st.cls_list, st.cls_count, img.bin_counters, user_counters);
}

void launch_split_scatter_sort(int P, int T, int B, unsigned capacity, const int* radii, unsigned gx, unsigned gy,
                               const GeomState& g, const BinningState& b, hipStream_t s) {
  const int NB = T * B;
  SplitState st = SplitState::fromChunk(b.sort_space, (size_t)NB);
  const unsigned ib = bin_index_bits(P);
  split_lds_attr();
  hipLaunchKernelGGL(split_scatter_kernel, dim3(bin_groups(P)), dim3(1024), (size_t)NB * sizeof(uint32_t), s, P, NB, (unsigned)B,
                     radii, g.means2D, g.depths, gx, gy, capacity, ib, st.splits, st.hist, st.bin_start, b.keys_unsorted);
  hipLaunchKernelGGL(split_sort_kernel<true>, dim3(256), dim3(1024), 0, s, NB, (unsigned)B, ib, st.cls_list, st.cls_count,
                     st.sub_ranges, b.keys_unsorted, b.keys, b.point_list);
  hipLaunchKernelGGL(split_sort_kernel<false>, dim3(512), dim3(1024), 0, s, NB, (unsigned)B, ib, st.cls_list, st.cls_count,
                     st.sub_ranges, b.keys_unsorted, b.keys, b.point_list);
  hipLaunchKernelGGL(split_sort_small_kernel, dim3(2048), dim3(256), 0, s, NB, (unsigned)B, ib, st.cls_list, st.cls_count,
                     st.sub_ranges, b.keys_unsorted, b.keys, b.point_list);
}

int launch_bin_sort(int T, int P, const BinningState& b, const ImageState& img, hipStream_t s) {
  // (Running size classes concurrently on forked streams was tried: with the light's side stream and two sort streams
  // the runtime ran out of hardware queues and folded the light filter onto the main queue -- 25 % slower.  Hence one
  // kernel that holds lists of every length up to 8192 keys, and a second, usually empty-handed, for the longer ones.)
  const unsigned ib = bin_index_bits(P);
  hipLaunchKernelGGL(bin_sort_kernel<true>, dim3(std::min(T, 256)), dim3(1024), 0, s, T, ib, img.tile_order, img.ranges,
                     b.keys_unsorted, b.keys, b.point_list);
  hipLaunchKernelGGL(bin_sort_kernel<false>, dim3(std::min(T, 512)), dim3(1024), 0, s, T, ib, img.tile_order, img.ranges,
                     b.keys_unsorted, b.keys, b.point_list);
  hipLaunchKernelGGL(bin_sort_small_kernel, dim3(std::min(T, 2048)), dim3(256), 0, s, T, ib, img.tile_order, img.ranges,
                     b.keys_unsorted, b.keys, b.point_list);
  return 0;
}

}  // namespace gigs
