// binning.hip -- tile binning pass 2 (emit) and the per-tile depth sort.
//
// Replaces gsplat isect_tiles (second pass) + cub::DeviceRadixSort over n_isect 64-bit keys +
// isect_offset_encode (reference call site flow3d/scene_model.py:360-373).
//
// MI355X design: the global sort only has to group by tile, which the counting pass (atomic histogram in
// k_project_fwd + one scan) already did; what remains is a SHORT depth sort per tile, done by one workgroup
// per (sub-sample, tile) entirely in LDS (160 KB/CU) with a bitonic network whose comparators all point the
// same way (so lists need no +inf padding).  Key = float-bits(depth) << 32 | emission index; the emission
// index grows with the Gaussian id inside one sub-sample, which reproduces the stable-radix tie order of the
// reference path exactly.
#include "common.h"

namespace {

struct EmitArgs {
  D4gsDims d;
  const float *depths;
  const int32_t *tile_rects;
  const int32_t *tiles_touched;
  const int32_t *isect_offsets;
  int32_t *isect_offsets_out;  // fused scan (d4gs_fused_scan_chunks): this kernel computes and writes the offsets
  const int32_t *chunk_base;   // [S * nchunks] exclusive scan of k_count_tiles' chunk sums
  int nchunks;                 // 0: offsets come from k_scan_apply
  int32_t *tile_cursor;  // tile_counts: [0,T) splats per tile, [T,2T) slot cursors (zero on entry)
  const int32_t *tile_offsets;
  uint64_t *keys;
  int32_t *gid_of_emit;
  int tw, th;
  const int64_t *n_dev;  // device {total, longest list}: the launch is a no-op when they exceed what the caller sized
  int64_t cap, max_hint;
  int use_lds;
  LazyWs lazy;  // D4GS_LAZY_SORT instantiation: near / far partition of every list (common.h)
  const uint64_t *tile_masks;  // D4GS_EXACT_TILES: per instance, the tiles of its rectangle that are binned (0: all of them); or NULL
};

// D4gsIsect.n_isect is a CAPACITY: the host may size the lists from a guess and look at the real count afterwards.
__device__ __forceinline__ bool over_capacity(const int64_t *n_dev, int64_t cap, int64_t max_hint) {
  return n_dev && (n_dev[0] > cap || (max_hint > 0 && n_dev[1] > max_hint));
}

// Slot assignment without per-intersection global atomics.  Block b takes 4096 instances of sub-sample b % S
// (1024 lanes x 4, the chunks of k_count_tiles):
//   1. histogram of the chunk's tile touches in LDS;
//   2. ONE returning global atomic per non-empty tile reserves the chunk's run inside that tile's (unsorted) list;
//      the bin becomes the absolute first slot of the run;
//   3. every (instance, tile) takes the next slot of its bin with an LDS atomic and writes its key there.
// Which slot a key lands in is irrelevant (k_tile_sort orders the list by depth, then emission index), so nothing
// has to be carried from the counting pass, splats of any footprint take the same path, and the 8-byte key stores
// of a block fall into short contiguous runs.  Tile grids too big for the LDS histogram use global cursors.
constexpr int EMIT_THREADS = 1024;
// LZ (D4GS_LAZY_SORT): 1 - only the keys of the depth buckets up to each tile's pivot are emitted, to the front of the tile's list (the
// near part, lazy.near[t] keys); 2 - launched by d4gs_lazy_far_sort between the two composite passes: the other keys, behind the near
// part, but only of the tiles the first pass flagged (the far part of a list whose tile saturated inside its near part is never
// written, let alone sorted).  A pair's emission index does not depend on which launch writes it.
template <int EMIT_PER_THREAD, int LZ>  // EMIT_PER_THREAD = d4gs_chunk_per_thread(dims), like k_count_tiles
__global__ void __launch_bounds__(EMIT_THREADS) k_emit(const EmitArgs a) {
  extern __shared__ int bins[];  // [tiles] (+ LZ: [tiles] pivots)
  if (over_capacity(a.n_dev, a.cap, a.max_hint)) return;
  constexpr bool LAZY = LZ != 0;
  const int tiles = a.tw * a.th, tid = threadIdx.x;
  const int s = blockIdx.x % a.d.S, chunk = blockIdx.x / a.d.S;
  const int tbase = s * tiles, n_tiles_all = a.d.S * tiles;
  const bool lds = a.use_lds;
  __shared__ int fbox[4];  // LZ == 2: bounding box of the sub-sample's flagged tiles
  if (lds) {
    for (int z = tid; z < tiles; z += EMIT_THREADS) bins[z] = 0;
    if (LZ == 2 && tid == 0) fbox[0] = fbox[1] = 1 << 30, fbox[2] = fbox[3] = -1;
    __syncthreads();
  }
  const uint32_t zlo = LAZY ? ~a.lazy.zr[2 * s] : 0u, zhi = LAZY ? a.lazy.zr[2 * s + 1] : 0u;
  // LZ: the sub-sample's pivots, staged once per block (one read per key otherwise); 2: "never" for the tiles that were not flagged
  int *piv = bins + tiles;
  if (LAZY) {
    for (int z = tid; z < tiles; z += EMIT_THREADS) {
      int p = a.lazy.pivot[tbase + z];
      if (LZ == 2) {
        if (a.lazy.flag[tbase + z] == 1) {  // (2: an earlier d4gs_raster_fwd on these lists emitted + sorted that far part)
          const int ty = z / a.tw, tx = z - ty * a.tw;
          atomicMin(&fbox[0], tx), atomicMin(&fbox[1], ty), atomicMax(&fbox[2], tx), atomicMax(&fbox[3], ty);
        } else {
          p = 1 << 30;
        }
      }
      piv[z] = p;
    }
    __syncthreads();
    if (LZ == 2 && fbox[2] < 0) return;  // (block-uniform) no tile of this sub-sample needs its far part
  }
  auto taken = [&](int bk, int t) -> bool { return LZ == 1 ? bk <= piv[t] : bk > piv[t]; };
  int cnt[EMIT_PER_THREAD], rx[EMIT_PER_THREAD], ry[EMIT_PER_THREAD], bkq[EMIT_PER_THREAD];
  uint64_t msk[EMIT_PER_THREAD];
  float dep[EMIT_PER_THREAD];
  // All of the lane's loads first (count, rectangle, depth and mask of every instance, binned or not), waited for ONCE: written
  // instance by instance the kernel walked 2 x EMIT_PER_THREAD dependent round trips to memory (count -> branch ->
  // rectangle -> histogram, four times over) at 19 % VALU utilisation.  (The empty asm statements keep the compiler from sinking
  // the loads back behind the branches.)
#ifndef D4GS_EMIT_BATCHED_LOADS
#define D4GS_EMIT_BATCHED_LOADS 1
#endif
#pragma unroll
  for (int q = 0; q < EMIT_PER_THREAD; q++) {
    const int g = (chunk * EMIT_PER_THREAD + q) * EMIT_THREADS + tid;
    cnt[q] = 0, msk[q] = 0, rx[q] = 0, ry[q] = 0, dep[q] = 0.f;
    if (g < a.d.N) cnt[q] = a.tiles_touched[(int64_t)s * a.d.N + g];
  }
  if constexpr (D4GS_EMIT_BATCHED_LOADS) {
#pragma unroll
    for (int q = 0; q < EMIT_PER_THREAD; q++) {
      const int g = (chunk * EMIT_PER_THREAD + q) * EMIT_THREADS + tid;
      if (g < a.d.N) {
        const int64_t i = (int64_t)s * a.d.N + g;
        const int2 rc = *reinterpret_cast<const int2 *>(a.tile_rects + i * 2);
        rx[q] = rc.x, ry[q] = rc.y;
        dep[q] = a.depths[i];
        if (a.tile_masks) msk[q] = a.tile_masks[i];
      }
    }
#pragma unroll
    for (int q = 0; q < EMIT_PER_THREAD; q++) {
      uint32_t ml = (uint32_t)msk[q], mh = (uint32_t)(msk[q] >> 32);
      asm volatile("" : "+v"(cnt[q]), "+v"(rx[q]), "+v"(ry[q]), "+v"(dep[q]), "+v"(ml), "+v"(mh));
      msk[q] = ((uint64_t)mh << 32) | ml;
    }
  }
#pragma unroll
  for (int q = 0; q < EMIT_PER_THREAD; q++) {
    const int g = (chunk * EMIT_PER_THREAD + q) * EMIT_THREADS + tid;
    if (g < a.d.N) {
      const int64_t i = (int64_t)s * a.d.N + g;
      if (cnt[q] > 0) {
        if constexpr (!D4GS_EMIT_BATCHED_LOADS) {
          const int2 rc = *reinterpret_cast<const int2 *>(a.tile_rects + i * 2);
          rx[q] = rc.x, ry[q] = rc.y;
          dep[q] = a.depths[i];
        }
        const int x0 = rx[q] & 0xffff, x1 = rx[q] >> 16, y0 = ry[q] & 0xffff, y1 = ry[q] >> 16;
        if (LZ == 2 && (x0 > fbox[2] || x1 <= fbox[0] || y0 > fbox[3] || y1 <= fbox[1])) {
          cnt[q] = 0;  // touches no flagged tile
          continue;
        }
        bkq[q] = LAZY ? d4gs_depth_bucket(dep[q], zlo, zhi, a.lazy.nb) : 0;
        if constexpr (!D4GS_EMIT_BATCHED_LOADS) msk[q] = a.tile_masks ? a.tile_masks[i] : 0;
        if (cnt[q] > 0 && !a.tile_masks) msk[q] = 0;
        if (lds && msk[q]) {  // D4GS_EXACT_TILES: bit (ty - y0) * 8 + (tx - x0)
          for (uint64_t m = msk[q]; m; m &= m - 1) {
            const int b = __ffsll((long long)m) - 1, t = (y0 + (b >> 3)) * a.tw + x0 + (b & 7);
            if (!LAZY || taken(bkq[q], t)) atomicAdd(&bins[t], 1);
          }
        } else if (lds) {
          for (int ty = y0; ty < y1; ty++)
            for (int tx = x0; tx < x1; tx++) {
              const int t = ty * a.tw + tx;
              if (!LAZY || taken(bkq[q], t)) atomicAdd(&bins[t], 1);
            }
        }
      }
    }
  }
  if (lds) {
    __syncthreads();
    for (int z = tid; z < tiles; z += EMIT_THREADS) {
      const int c = bins[z];
      if (c == 0) continue;
      const int t = tbase + z;
      if (LZ == 1) bins[z] = a.tile_offsets[t] + atomicAdd(a.lazy.cur + 2 * t, c);
      else if (LZ == 2) bins[z] = a.tile_offsets[t] + a.lazy.near[t] + atomicAdd(a.lazy.cur + 2 * t + 1, c);
      else bins[z] = a.tile_offsets[t] + atomicAdd(a.tile_cursor + n_tiles_all + t, c);
    }
    __syncthreads();
  }
  // fused scan: emission index base of each of the block's 4096 instances = chunk base + exclusive scan of the counts in
  // instance order (q-major, then lane) - exactly what k_scan_apply would have written
  uint32_t ebase[EMIT_PER_THREAD];
  if (LZ != 2 && a.nchunks) {
    __shared__ int wsum[EMIT_THREADS / 64];
    int carry = a.chunk_base[s * a.nchunks + chunk];
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int q = 0; q < EMIT_PER_THREAD; q++) {
      int inc = cnt[q];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
      }
      if (lane == 63) wsum[wave] = inc;
      __syncthreads();
      int base = 0, tot = 0;
#pragma unroll
      for (int w = 0; w < EMIT_THREADS / 64; w++) {
        const int x = wsum[w];
        base += w < wave ? x : 0;
        tot += x;
      }
      ebase[q] = (uint32_t)(carry + base + inc - cnt[q]);
      carry += tot;
      __syncthreads();
      const int g = (chunk * EMIT_PER_THREAD + q) * EMIT_THREADS + tid;
      if (g < a.d.N) a.isect_offsets_out[(int64_t)s * a.d.N + g] = (int32_t)ebase[q];
    }
  }
#pragma unroll
  for (int q = 0; q < EMIT_PER_THREAD; q++) {
    if (cnt[q] == 0) continue;
    const int g = (chunk * EMIT_PER_THREAD + q) * EMIT_THREADS + tid;
    const int64_t i = (int64_t)s * a.d.N + g;
    const int x0 = rx[q] & 0xffff, x1 = rx[q] >> 16, y0 = ry[q] & 0xffff, y1 = ry[q] >> 16;
    const uint64_t hi = (uint64_t)__float_as_uint(dep[q]) << 32;
    uint32_t e = (LZ != 2 && a.nchunks) ? ebase[q] : (uint32_t)a.isect_offsets[i];
    auto put = [&](int t) {
      if (LAZY && !taken(bkq[q], t)) return;
      const int slot = (LAZY || lds) ? atomicAdd(&bins[t], 1)
                                     : a.tile_offsets[tbase + t] + atomicAdd(a.tile_cursor + n_tiles_all + tbase + t, 1);
      a.keys[slot] = hi | e;
      a.gid_of_emit[e] = g;
    };
    if (msk[q]) {  // the emission index counts the instance's BINNED tiles in ascending bit order (both lazy launches alike)
      for (uint64_t m = msk[q]; m; m &= m - 1, e++) {
        const int b = __ffsll((long long)m) - 1;
        put((y0 + (b >> 3)) * a.tw + x0 + (b & 7));
      }
      continue;
    }
    for (int ty = y0; ty < y1; ty++)
      for (int tx = x0; tx < x1; tx++, e++) put(ty * a.tw + tx);
  }
}

// all-ascending bitonic network on `n` keys (any n): step (k, j) compares i with its partner l > i.
// Generic form (used on global memory for lists beyond the LDS budget).
template <typename KeyPtr>
__device__ __forceinline__ void bitonic_sort(KeyPtr key, int n) {
  int P = 1;
  while (P < n) P <<= 1;
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
        const int blk = t / j, off = t - blk * j;
        const int i = blk * 2 * j + off;
        const int l = (j == (k >> 1)) ? (blk * 2 * j + (2 * j - 1 - off)) : (i + j);
        if (l < n) {
          uint64_t a = key[i], b = key[l];
          if (a > b) {
            key[i] = b;
            key[l] = a;
          }
        }
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
  const int lo = __shfl_xor((int)(uint32_t)v, m), hi = __shfl_xor((int)(v >> 32), m);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}

// Lane exchanges of the in-chunk bitonic steps.  `lane ^ M` for M = 1, 2, 3 (quad_perm), 7 (row_half_mirror), 15
// (row_mirror) is one DPP move on the VALU crossbar; 4 = 3 then 7 and 8 = 7 then 15 are two.  Only the exchanges that
// cross 16-lane rows (16, 31, 32, 63) go through ds_bpermute, i.e. the CU-wide LDS pipe that bounded this kernel.
template <int M>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
  if constexpr (M == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, false);        // [1,0,3,2]
  else if constexpr (M == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, false);   // [2,3,0,1]
  else if constexpr (M == 3) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x1B, 0xf, 0xf, false);   // [3,2,1,0]
  else if constexpr (M == 7) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, false);  // row_half_mirror
  else if constexpr (M == 15) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, false); // row_mirror
  else if constexpr (M == 4) return lane_xor<7>(lane_xor<3>(v));
  else if constexpr (M == 8) return lane_xor<15>(lane_xor<7>(v));
  else return (uint32_t)__shfl_xor((int)v, M);
}
// compare-exchange with lane ^ M; lanes whose bit BIT is clear keep the smaller key
template <int M, int BIT>
__device__ __forceinline__ void lane_cex(uint64_t &v, int lane) {
  const uint32_t lo = lane_xor<M>((uint32_t)v), hi = lane_xor<M>((uint32_t)(v >> 32));
  const uint64_t o = ((uint64_t)hi << 32) | lo;
  const bool lower = (lane & BIT) == 0;
  v = (lower == (v < o)) ? v : o;
}
template <int J, int NV>
__device__ __forceinline__ void chunk_plain(uint64_t (&v)[NV], int lane) {  // steps J, J/2, .., 1
  if constexpr (J >= 1) {
#pragma unroll
    for (int u = 0; u < NV; u++) lane_cex<J, J>(v[u], lane);
    chunk_plain<J / 2, NV>(v, lane);
  }
}
template <int K, int NV>
__device__ __forceinline__ void chunk_block(uint64_t (&v)[NV], int lane) {  // the whole k-block K <= 64 of the network
#pragma unroll
  for (int u = 0; u < NV; u++) lane_cex<K - 1, K / 2>(v[u], lane);  // flip step
  chunk_plain<K / 4, NV>(v, lane);
}

// ---- 512-key chunks in registers (round 3) ---------------------------------------------------------------------------
// A wave holds a chunk of 64 * NV keys as NV registers per lane (key index u * 64 + lane).  Inside the chunk every step
// of the network stays in registers: partner distances < 64 are the DPP / bpermute lane exchanges above, distances
// 64 .. 32 NV pair two REGISTERS of the same lane (a plain compare-exchange, no cross-lane traffic at all), and a
// k-block's flip step pairs register u of lane l with the mirrored register of lane 63 - l.  A list of up to 512 keys
// is therefore sorted by ONE wave without LDS or barriers (round 2: 6 LDS passes + barriers for 512 keys), and longer
// lists only go through LDS for distances >= 512 (4096 keys: 6 LDS passes instead of 21).
__device__ __forceinline__ uint64_t rev64(uint64_t v) { return shfl_xor_u64(v, 63); }
__device__ __forceinline__ void reg_cex(uint64_t &lo, uint64_t &hi) {  // lo = min, hi = max
  const uint64_t a = lo, b = hi;
  const bool sw = a > b;
  lo = sw ? b : a, hi = sw ? a : b;
}
template <int NV, int D>
__device__ __forceinline__ void reg_flip(uint64_t (&v)[NV]) {  // flip step of the k-block k = 64 D
#pragma unroll
  for (int base = 0; base < NV; base += D)
#pragma unroll
    for (int h = 0; h < D / 2; h++) {
      const int a = base + h, b = base + D - 1 - h;
      const uint64_t ra = rev64(v[a]), rb = rev64(v[b]);
      v[a] = v[a] < rb ? v[a] : rb;  // lower index keeps the smaller key
      v[b] = v[b] > ra ? v[b] : ra;
    }
}
template <int NV, int R>
__device__ __forceinline__ void reg_plain(uint64_t (&v)[NV]) {  // plain steps at register distances R, R/2, .., 1
  if constexpr (R >= 1) {
#pragma unroll
    for (int u = 0; u < NV; u++)
      if ((u / R) % 2 == 0) reg_cex(v[u], v[u + R]);
    reg_plain<NV, R / 2>(v);
  }
}
template <int NV>
__device__ __forceinline__ void chunk_sort(uint64_t (&v)[NV], int lane) {  // the whole network on 64 NV keys
  chunk_block<2, NV>(v, lane);
  chunk_block<4, NV>(v, lane);
  chunk_block<8, NV>(v, lane);
  chunk_block<16, NV>(v, lane);
  chunk_block<32, NV>(v, lane);
  chunk_block<64, NV>(v, lane);
  if constexpr (NV >= 2) {
    reg_flip<NV, 2>(v);
    chunk_plain<32, NV>(v, lane);
  }
  if constexpr (NV >= 4) {
    reg_flip<NV, 4>(v);
    reg_plain<NV, 1>(v);
    chunk_plain<32, NV>(v, lane);
  }
  if constexpr (NV >= 8) {
    reg_flip<NV, 8>(v);
    reg_plain<NV, 2>(v);
    chunk_plain<32, NV>(v, lane);
  }
}
constexpr int CNV = 8, CHUNK = 64 * CNV;  // keys per lane / per chunk

// LDS form.  `key[0..P)` holds the list padded with UINT64_MAX to a power of two P >= CHUNK.  Chunks of 512 keys are
// sorted in registers; only the steps with distance >= 512 go through LDS with a barrier each.
__device__ __forceinline__ void bitonic_sort_lds(uint64_t *key, int P, int n) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // keys [n, P) are UINT64_MAX padding: in this all-ascending network a pair whose upper partner is padding never
  // swaps, so chunks made of padding only and cross-chunk pairs reaching into it are skipped (work ~ n, not P)
  const int nchunks = (n + CHUNK - 1) / CHUNK;
  for (int c = wv; c < nchunks; c += nw) {
    uint64_t v[CNV];
#pragma unroll
    for (int u = 0; u < CNV; u++) v[u] = key[c * CHUNK + u * 64 + lane];
    chunk_sort<CNV>(v, lane);
#pragma unroll
    for (int u = 0; u < CNV; u++) key[c * CHUNK + u * 64 + lane] = v[u];
  }
  __syncthreads();
  for (int k = 2 * CHUNK; k <= P; k <<= 1) {
    for (int j = k >> 1; j >= CHUNK; j >>= 1) {  // cross-chunk steps
      for (int t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
        const int blk = t / j, off = t - blk * j;
        const int i = blk * 2 * j + off;
        const int l = (j == (k >> 1)) ? (blk * 2 * j + (2 * j - 1 - off)) : (i + j);
        if (l < n) {
          const uint64_t a = key[i], b = key[l];
          if (a > b) key[i] = b, key[l] = a;
        }
      }
      __syncthreads();
    }
    for (int c = wv; c < nchunks; c += nw) {  // distances 256 .. 1
      uint64_t v[CNV];
#pragma unroll
      for (int u = 0; u < CNV; u++) v[u] = key[c * CHUNK + u * 64 + lane];
      reg_plain<CNV, CNV / 2>(v);
      chunk_plain<32, CNV>(v, lane);
#pragma unroll
      for (int u = 0; u < CNV; u++) key[c * CHUNK + u * 64 + lane] = v[u];
    }
    __syncthreads();
  }
}

struct SortArgs {
  const int32_t *tile_offsets;
  uint64_t *keys;
  const int32_t *gid_of_emit;
  int32_t *sorted_gid;
  int32_t *sorted_emit;
  int lo, cap;  // this launch sorts the lists with lo < n <= cap in LDS; cap < 0: everything longer, in global memory
  const int64_t *n_dev;
  int64_t n_cap, max_hint;
  int n_lists;
  // D4GS_LAZY_SORT: pass 1 sorts the near part [base, base + near[t]) of every list, pass 2 the far part of the lists whose tile was
  // flagged by the forward composite (it did not saturate within the near part); pass 0: whole lists
  const int32_t *lazy_near, *lazy_flag;
  int pass;
};
// -> false: nothing to sort for this list in this pass
__device__ __forceinline__ bool lazy_range(const SortArgs &a, int t, int &base, int &n) {
  if (a.pass == 1) n = min(n, a.lazy_near[t]);
  else if (a.pass == 2) {
    const int nn = a.lazy_near[t];
    if (nn >= n || a.lazy_flag[t] != 1) return false;
    base += nn, n -= nn;
  }
  return true;
}

// lists of up to 512 keys: one WAVE per list, straight from global memory into registers and back - no LDS, no barrier
template <int NV>
__device__ __forceinline__ void wave_sort_list(const SortArgs &a, int base, int n, int lane) {
  uint64_t v[NV];
  const uint64_t *gk = a.keys + base;
#pragma unroll
  for (int u = 0; u < NV; u++) v[u] = (u * 64 + lane < n) ? gk[u * 64 + lane] : ~0ull;
  chunk_sort<NV>(v, lane);
#pragma unroll
  for (int u = 0; u < NV; u++) {
    const int p = u * 64 + lane;
    if (p < n) {
      const uint32_t e = (uint32_t)v[u];
      a.sorted_emit[base + p] = (int32_t)e;
      a.sorted_gid[base + p] = a.gid_of_emit[e];
    }
  }
}
// LONG (round 6): the workgroup also sorts those of its four lists that hold 513 ... 2 048 keys - all four waves through 16 KB of LDS, one
// list after the other, exactly the network of k_tile_sort's first class.  A scene whose lists are mostly short (the reference's training
// shape: p50 450 keys, a few hundred of 6 336 beyond 512; 720p) used to pay a whole launch of that class for the few: 19-21 / 11 us.
template <bool LONG>
__global__ void __launch_bounds__(256) k_tile_sort_w(const SortArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint64_t wkeys[];  // LONG: 2 048 keys
  if (over_capacity(a.n_dev, a.n_cap, a.max_hint)) return;  // (block-uniform)
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  int base = 0, n = 0;
  if (t < a.n_lists) {
    base = a.tile_offsets[t];
    n = a.tile_offsets[t + 1] - base;
    if (a.pass && !lazy_range(a, t, base, n)) n = 0;
  }
  if (n > 0 && n <= CHUNK) {
    if (n <= 64) wave_sort_list<1>(a, base, n, lane);
    else if (n <= 128) wave_sort_list<2>(a, base, n, lane);
    else if (n <= 256) wave_sort_list<4>(a, base, n, lane);
    else wave_sort_list<8>(a, base, n, lane);
  }
  if constexpr (LONG) {
    __shared__ int lb[4], ln[4];
    if (lane == 0) lb[threadIdx.x >> 6] = base, ln[threadIdx.x >> 6] = (n > CHUNK && n <= 4 * CHUNK) ? n : 0;
    __syncthreads();
    for (int q = 0; q < 4; q++) {
      const int nq = ln[q];  // (block-uniform)
      if (nq == 0) continue;
      const uint64_t *gk = a.keys + lb[q];
      int P = CHUNK;
      while (P < nq) P <<= 1;
      const int fill = ((nq + CHUNK - 1) / CHUNK) * CHUNK;
      for (int p = threadIdx.x; p < fill; p += blockDim.x) wkeys[p] = p < nq ? gk[p] : ~0ull;
      __syncthreads();
      bitonic_sort_lds(wkeys, P, nq);
      for (int p = threadIdx.x; p < nq; p += blockDim.x) {
        const uint32_t e = (uint32_t)wkeys[p];
        a.sorted_emit[lb[q] + p] = (int32_t)e;
        a.sorted_gid[lb[q] + p] = a.gid_of_emit[e];
      }
      __syncthreads();
    }
  }
}

__global__ void __launch_bounds__(1024) k_tile_sort(const SortArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint64_t skeys[];
  if (over_capacity(a.n_dev, a.n_cap, a.max_hint)) return;
  const int t = blockIdx.x;
  int base = a.tile_offsets[t];
  int n = a.tile_offsets[t + 1] - base;
  if (a.pass && !lazy_range(a, t, base, n)) return;
  if (n <= a.lo || (a.cap >= 0 && n > a.cap)) return;
  if (n <= CHUNK) {  // (only when lo == 0, the first class of a scene that also has longer lists: one launch for both kinds)
    if (threadIdx.x < 64) {  // one wave sorts the list in registers, exactly as k_tile_sort_w would
      const int lane = threadIdx.x;
      if (n <= 64) wave_sort_list<1>(a, base, n, lane);
      else if (n <= 128) wave_sort_list<2>(a, base, n, lane);
      else if (n <= 256) wave_sort_list<4>(a, base, n, lane);
      else wave_sort_list<8>(a, base, n, lane);
    }
    return;
  }
  uint64_t *gk = a.keys + base;
  if (a.cap >= 0) {
    int P = CHUNK;
    while (P < n) P <<= 1;
    for (int p = threadIdx.x; p < P; p += blockDim.x) skeys[p] = p < n ? gk[p] : ~0ull;
    __syncthreads();
    bitonic_sort_lds(skeys, P, n);
    for (int p = threadIdx.x; p < n; p += blockDim.x) {
      const uint32_t e = (uint32_t)skeys[p];
      a.sorted_emit[base + p] = (int32_t)e;
      a.sorted_gid[base + p] = a.gid_of_emit[e];
    }
  } else {  // list longer than the LDS budget: same network on global memory (rare; correctness path)
    bitonic_sort(gk, n);
    for (int p = threadIdx.x; p < n; p += blockDim.x) {
      const uint32_t e = (uint32_t)gk[p];
      a.sorted_emit[base + p] = (int32_t)e;
      a.sorted_gid[base + p] = a.gid_of_emit[e];
    }
  }
}


// Size classes, one launch each (a workgroup exits at once if its list is not in the class): lists of up to 512 keys take
// one wave each, in registers (k_tile_sort_w); beyond that the LDS a workgroup reserves and its width follow the list
// length - 16 KB / 256 lanes up to 2048 keys (8+ workgroups per CU) ... 128 KB / 1024 lanes up to 16384 keys, anything longer in
// global memory.  (Round 1 had only the 16 KB and 128 KB classes: a scene whose lists run to ~4 k keys sorted them one workgroup
// of 4 waves per CU - 1.54 ms on cfg2 with 4x larger splats.)  pass: 0 whole lists, 1 / 2 the near / flagged far parts
// (D4GS_LAZY_SORT).
int launch_sorts(const D4gsDims *dims, const D4gsProjOut *proj, const D4gsIsect *isect, int pass, hipStream_t stream) {
  const int tw = (dims->width + D4GS_TILE - 1) / D4GS_TILE, th = (dims->height + D4GS_TILE - 1) / D4GS_TILE;
  const int n_tiles = dims->S * tw * th;
  LazyWs lw{};
  if (pass) lw = d4gs_lazy_carve(proj->lazy_ws, dims->S, tw * th);
  // per call: the attribute belongs to the (function, device) pair and a process may drive several GPUs
  (void)hipFuncSetAttribute((const void *)k_tile_sort, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
  const int64_t longest = isect->max_tile_count > 0 ? isect->max_tile_count : isect->n_isect;
  // Lists of up to 512 keys: one wave each, in registers, four lists per workgroup (k_tile_sort_w) - unless the lists average
  // well above 512 keys (cfg2: 935): then that launch would find (almost) nothing to do and the few short lists ride in the first
  // LDS class's launch instead (wave 0 of their workgroup, same register sort): one launch less, 5 - 7 us per frame on cfg2.
  const bool merge_short = pass == 0 && longest > CHUNK && isect->n_isect >= (int64_t)768 * n_tiles;  // (capacity ~ 1.25 x the count)
  // ... and the other way round (round 6): where the lists are mostly SHORT, the few of 513 ... 2 048 keys are sorted by the workgroup of
  // k_tile_sort_w that owns them (all four waves, 16 KB of LDS) instead of a launch of the first LDS class for their sake.
  static const bool no_long = getenv("D4GS_SORT_NO_MERGE_LONG") != nullptr;  // A/B hook
  // Only whole-list passes whose lists average at most ~512 keys (capacity / tiles < 640): where many lists are long - the near / far
  // passes of a lazy sort, cfg5: 609 -> 739 us - four of them queue up inside one workgroup.  Measured (profiles/r06_ab_sort_merge_long.txt):
  // the reference's training shape 56.4 -> 46.8 us, cfg3 97.8 -> 87.7.
  const bool merge_long = !merge_short && !no_long && longest > CHUNK && pass == 0 && isect->n_isect < (int64_t)640 * n_tiles;
  if (!merge_short) {
    SortArgs s{proj->tile_offsets, isect->keys, isect->gid_of_emit, isect->sorted_gid, isect->sorted_emit,
               0, CHUNK, proj->n_isect, isect->n_isect, isect->max_tile_count, n_tiles, lw.near, lw.flag, pass};
    if (merge_long) D4GS_LAUNCH("k_tile_sort_w", k_tile_sort_w<true>, dim3((n_tiles + 3) / 4), dim3(256), 4 * CHUNK * 8, stream, s);
    else D4GS_LAUNCH("k_tile_sort_w", k_tile_sort_w<false>, dim3((n_tiles + 3) / 4), dim3(256), 0, stream, s);
  }
  const int classes[6][3] = {{merge_short ? 0 : CHUNK, 2048, 256}, {2048, 4096, 512}, {4096, 8192, 1024}, {8192, 16384, 1024}, {16384, -1, 1024}};
  for (int c = merge_long ? 1 : 0; c < 5; c++) {
    if (longest <= classes[c][0]) break;  // no list is that long
    SortArgs s{proj->tile_offsets, isect->keys, isect->gid_of_emit, isect->sorted_gid, isect->sorted_emit,
               classes[c][0], classes[c][1], proj->n_isect, isect->n_isect, isect->max_tile_count, n_tiles, lw.near, lw.flag, pass};
    const size_t lds = classes[c][1] > 0 ? (size_t)classes[c][1] * 8 : 0;
    D4GS_LAUNCH("k_tile_sort", k_tile_sort, dim3(n_tiles), dim3(classes[c][2]), lds, stream, s);
  }
  return d4gs_check_launch("k_tile_sort");
}

}  // namespace

namespace {

int fill_emit_args(EmitArgs &e, const D4gsDims *dims, const D4gsProjOut *proj, const D4gsIsect *isect) {
  e.d = *dims;
  e.depths = proj->depths;
  e.tile_rects = proj->tile_rects;
  e.tiles_touched = proj->tiles_touched;
  e.isect_offsets = proj->isect_offsets;
  e.isect_offsets_out = proj->isect_offsets, e.chunk_base = proj->scan_ws, e.nchunks = d4gs_fused_scan_chunks(dims);
  e.tile_cursor = proj->tile_counts;
  e.tile_masks = (dims->flags & D4GS_EXACT_TILES) ? proj->tile_masks : nullptr;
  e.tile_offsets = proj->tile_offsets;
  e.keys = isect->keys;
  e.gid_of_emit = isect->gid_of_emit;
  e.tw = (dims->width + D4GS_TILE - 1) / D4GS_TILE;
  e.th = (dims->height + D4GS_TILE - 1) / D4GS_TILE;
  e.n_dev = proj->n_isect, e.cap = isect->n_isect, e.max_hint = isect->max_tile_count;
  const int per_block = EMIT_THREADS * d4gs_chunk_per_thread(dims);
  if (e.nchunks && e.nchunks != (dims->N + per_block - 1) / per_block) {  // k_count_tiles' chunks must be this kernel's
    d4gs_set_error("internal: fused scan chunking mismatch (%d vs %d)", e.nchunks, (dims->N + per_block - 1) / per_block);
    return D4GS_EINVAL;
  }
  e.use_lds = sizeof(int) * (size_t)e.tw * e.th <= 64 * 1024;
  e.lazy = LazyWs{};
  return D4GS_OK;
}

// lz: 0 plain, 1 / 2 the near parts / the flagged tiles' far parts of a D4GS_LAZY_SORT frame (k_emit)
int launch_emit(const EmitArgs &e, const D4gsDims *dims, int lz, hipStream_t stream) {
  const int pt = d4gs_chunk_per_thread(dims), per_block = EMIT_THREADS * pt;
  const unsigned eblocks = (unsigned)(((dims->N + per_block - 1) / per_block) * dims->S);
  const size_t ebytes = e.use_lds ? sizeof(int) * (size_t)e.tw * e.th * (lz ? 2 : 1) : 0;
#define D4GS_EMIT(PT_, LZ_)                                                                                               \
  do {                                                                                                                    \
    if (ebytes > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_emit<PT_, LZ_>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024); \
    D4GS_LAUNCH("k_emit", (k_emit<PT_, LZ_>), dim3(eblocks), dim3(EMIT_THREADS), ebytes, stream, e);                     \
  } while (0)
  if (pt == 1) {
    if (lz == 0) D4GS_EMIT(1, 0);
    else if (lz == 1) D4GS_EMIT(1, 1);
    else D4GS_EMIT(1, 2);
  } else if (pt == 2) {
    if (lz == 0) D4GS_EMIT(2, 0);
    else if (lz == 1) D4GS_EMIT(2, 1);
    else D4GS_EMIT(2, 2);
  } else if (pt == 3) {
    if (lz == 0) D4GS_EMIT(3, 0);
    else if (lz == 1) D4GS_EMIT(3, 1);
    else D4GS_EMIT(3, 2);
  } else if (pt == 5) {
    if (lz == 0) D4GS_EMIT(5, 0);
    else if (lz == 1) D4GS_EMIT(5, 1);
    else D4GS_EMIT(5, 2);
  } else if (pt == 6) {
    if (lz == 0) D4GS_EMIT(6, 0);
    else if (lz == 1) D4GS_EMIT(6, 1);
    else D4GS_EMIT(6, 2);
  } else {
    if (lz == 0) D4GS_EMIT(4, 0);
    else if (lz == 1) D4GS_EMIT(4, 1);
    else D4GS_EMIT(4, 2);
  }
#undef D4GS_EMIT
  return d4gs_check_launch("k_emit");
}

}  // namespace

int d4gs_bin_sort_impl(const D4gsDims *dims, const D4gsProjOut *proj, const D4gsIsect *isect, hipStream_t stream) {
  if (isect->n_isect <= 0) return D4GS_OK;
  if (isect->n_isect >= (int64_t)0x7fffffff) {
    d4gs_set_error("n_isect=%lld exceeds int32 indexing", (long long)isect->n_isect);
    return D4GS_ECAPACITY;
  }
  EmitArgs e;
  int rc = fill_emit_args(e, dims, proj, isect);
  if (rc) return rc;
  const bool lazy = d4gs_lazy_on(dims, proj);
  if (lazy) {  // every tile's near / far pivot first (the caller's near_target is known here, not in d4gs_project_fwd)
    e.lazy = d4gs_lazy_carve(proj->lazy_ws, dims->S, e.tw * e.th);
    if ((rc = d4gs_lazy_pivot_launch(dims, proj, isect->near_target, stream))) return rc;
  }
  if ((rc = launch_emit(e, dims, lazy ? 1 : 0, stream))) return rc;
  return launch_sorts(dims, proj, isect, lazy ? 1 : 0, stream);
}

// D4GS_LAZY_SORT, second half (called by d4gs_raster_fwd between its two composite passes): the far parts of the lists whose tile
// the first pass flagged are emitted and sorted.  The far parts of the other lists - rows behind their tile's last contributor - are
// never written: nothing composites or replays them, and the backward of a lazy render always takes SPARSE gradient rows
// (raster_bwd.hip), which never touches a dead row (copying their emission indices for the dense zero-fill cost 176 us on cfg2 with
// 4x splats - more than the sort it saved).
int d4gs_lazy_far_sort(const D4gsDims *dims, const D4gsProjOut *proj, const D4gsIsect *isect, hipStream_t stream) {
  EmitArgs e;
  int rc = fill_emit_args(e, dims, proj, isect);
  if (rc) return rc;
  e.lazy = d4gs_lazy_carve(proj->lazy_ws, dims->S, e.tw * e.th);
  if ((rc = launch_emit(e, dims, 2, stream))) return rc;
  return launch_sorts(dims, proj, isect, 2, stream);
}
