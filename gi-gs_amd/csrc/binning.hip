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
#include <cstring>
#include <string.h>
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

}  // namespace gigs
