// surfel_bin.hip -- tile binning and per-tile depth ordering of the 2D-surfel rasterizer, gfx950.
//
// Produces exactly what upstream's  InclusiveSum -> duplicateWithKeys -> DeviceRadixSort::SortPairs ->
// identifyTileRanges  chain produces (SURVEY.md A.1 "Binning"; call site /root/reference/nsr/gs_surfel.py:100-114):
//   ranges[tile] = [tile_start[tile], tile_start[tile+1])  and  point_list[] ordered by (tile, depth bits, index)
// but NOT by sorting D 64-bit keys device-wide.  MI355X-first formulation:
//   1. tile occupancy was counted by the preprocess kernel (LDS histograms flushed with L2 atomics);
//   2. surfel_tile_scan_kernel : one workgroup scans the V*tiles counters (a few thousand words) -> tile_start, D;
//   3. surfel_fill_kernel      : every (view, splat) claims slots in its tiles' segments (workgroup-aggregated
//                                returning atomics) and writes key = depth_bits<<32 | index (segment order arbitrary);
//   4. surfel_tile_sort_kernel : one workgroup per tile sorts its segment by that 64-bit key in LDS (bitonic network,
//                                up to 8192 entries = 64 KiB of the CU's 160 KiB) and writes the index list.  Keys are
//                                unique per tile, so the result equals the stable radix sort by depth with index as the
//                                tie-break -- the order upstream's stable sort yields.  Longer segments are sorted in
//                                8192-entry runs and merged by rank counting (correct for any length).
// HBM traffic is 8 B written + 8 B read + 4 B written per entry instead of ~144 B per entry for a 6-pass radix sort,
// and the launch count is 3 instead of ~18.
#include "surfel_common.h"

namespace ga {

// ---------------------------------------------------------------------------------------------------------------
// 2. exclusive scan of the per-(view, tile) counters; single workgroup of 1024 threads, wave-shuffle scans.
__global__ __launch_bounds__(1024) void surfel_tile_scan_kernel(const uint32_t *__restrict__ tile_count,
                                                                uint32_t *__restrict__ tile_start,
                                                                uint32_t *__restrict__ tile_cursor,
                                                                uint32_t *__restrict__ tile_order, int n,
                                                                int64_t capacity, int64_t *__restrict__ status)
{
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    __shared__ uint32_t maxc_s;
    __shared__ uint64_t wide_tot[16];
    __shared__ uint32_t bucket[33];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) { carry_s = 0; maxc_s = 0; }
    if (tid < 33) bucket[tid] = 0;
    __syncthreads();
    uint32_t local_max = 0;
    uint64_t local_sum = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const uint32_t c = i < n ? tile_count[i] : 0u;
        local_max = max(local_max, c);
        local_sum += c;
        if (i < n) atomicAdd(&bucket[c ? 32 - __builtin_clz(c) : 0], 1u);  // bucket b: 2^(b-1) <= c < 2^b
        uint32_t x = c;  // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wave_tot[wid] = x;
        __syncthreads();
        uint32_t wbase = 0;
        for (int w = 0; w < wid; ++w) wbase += wave_tot[w];
        const uint32_t carry = carry_s;
        const uint32_t excl = carry + wbase + x - c;
        if (i < n) { tile_start[i] = excl; tile_cursor[i] = excl; }
        __syncthreads();
        if (tid == 1023) carry_s = carry + wbase + x;
        __syncthreads();
    }
    // 64-bit total (the uint32 running offsets above wrap past 2^32; that case is reported as overflow).
    atomicMax(&maxc_s, local_max);
    uint64_t wsum = local_sum;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wsum += __shfl_down(wsum, o, 64);
    if (lane == 0) wide_tot[wid] = wsum;
    __syncthreads();
    if (tid == 0) {
        uint64_t total = 0;
        for (int w = 0; w < 16; ++w) total += wide_tot[w];
        tile_start[n] = (uint32_t)total;
        status[GA_STATUS_NUM_RENDERED] = (int64_t)total;
        status[GA_STATUS_OVERFLOW] = (total > (uint64_t)capacity || total > 0xFFFFFFFFull) ? 1 : 0;
        status[GA_STATUS_MAX_TILE] = (int64_t)maxc_s;
        // bucket start offsets, longest lists first
        uint32_t run = 0;
        for (int b = 32; b >= 0; --b) { const uint32_t c = bucket[b]; bucket[b] = run; run += c; }
    }
    __syncthreads();
    // Workgroup schedule for the per-tile kernels: tiles ordered by list length class, longest first, so the long
    // serial chains start at once and the short ones fill in behind them (order inside a class is irrelevant).
    for (int i = tid; i < n; i += 1024) {
        const uint32_t c = tile_count[i];
        const uint32_t pos = atomicAdd(&bucket[c ? 32 - __builtin_clz(c) : 0], 1u);
        tile_order[pos] = (uint32_t)i;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 3. fill.  One workgroup = 256 threads x kBinSplats consecutive Gaussians of one view (blockIdx.y), three LDS-aggregated
//    steps so that a tile's cursor sees one returning global atomic per workgroup instead of one per entry:
//      a) count the workgroup's entries per tile in an LDS histogram;
//      b) reserve [base, base+count) in each touched tile's segment: ONE returning atomic on the global cursor;
//      c) every entry takes its rank inside the workgroup from a returning LDS atomic and writes its key.
//    The order inside a segment is arbitrary (the per-tile sort fixes it).  Views with more than kLdsTiles tiles fall
//    back to one returning global atomic per entry.
template <bool kLds>
__global__ __launch_bounds__(256) void surfel_fill_kernel(const uint16_t *__restrict__ rect,
                                                          const float *__restrict__ depth, Dims dm,
                                                          uint32_t *__restrict__ tile_cursor,
                                                          uint64_t *__restrict__ keys,
                                                          const int64_t *__restrict__ status)
{
    extern __shared__ uint32_t lds[];  // [tiles] counts -> ranks, [tiles] segment bases
    if (status[GA_STATUS_OVERFLOW]) return;
    const int v = blockIdx.y;
    uint32_t *cur = tile_cursor + (size_t)v * dm.tiles;
    uint32_t *cnt = lds, *basep = lds + dm.tiles;
    ushort4 rcs[kBinSplats];
#pragma unroll
    for (int k = 0; k < kBinSplats; ++k) {
        const int i = (blockIdx.x * kBinSplats + k) * 256 + threadIdx.x;
        rcs[k] = i < dm.N ? *reinterpret_cast<const ushort4 *>(rect + 4 * ((size_t)v * dm.N + i)) : make_ushort4(0, 0, 0, 0);
    }
    if (kLds) {
        for (int t = threadIdx.x; t < dm.tiles; t += 256) cnt[t] = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kBinSplats; ++k)
            for (int ty = rcs[k].y; ty < rcs[k].w; ++ty)
                for (int tx = rcs[k].x; tx < rcs[k].z; ++tx) atomicAdd(cnt + ty * dm.gx + tx, 1u);
        __syncthreads();
        for (int t = threadIdx.x; t < dm.tiles; t += 256) {
            const uint32_t c = cnt[t];
            if (c) { basep[t] = atomicAdd(cur + t, c); cnt[t] = 0; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < kBinSplats; ++k) {
        const ushort4 rc = rcs[k];
        if (rc.z <= rc.x || rc.w <= rc.y) continue;
        const int i = (blockIdx.x * kBinSplats + k) * 256 + threadIdx.x;
        const uint64_t key = ((uint64_t)__float_as_uint(depth[(size_t)v * dm.N + i]) << 32) | (uint32_t)i;
        for (int ty = rc.y; ty < rc.w; ++ty)
            for (int tx = rc.x; tx < rc.z; ++tx) {
                const int t = ty * dm.gx + tx;
                const uint32_t pos = kLds ? basep[t] + atomicAdd(cnt + t, 1u) : atomicAdd(cur + t, 1u);
                keys[pos] = key;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 4. per-tile sort.  LDS bitonic network on u64 keys, padded with ~0 to the next power of two.
__device__ __forceinline__ void bitonic_sort_lds(uint64_t *s, int np, int tid, int nthreads)
{
    for (int k = 2; k <= np; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (np >> 1); t += nthreads) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int p = i | j;
                const uint64_t a = s[i], b = s[p];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { s[i] = b; s[p] = a; }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void surfel_tile_sort_kernel(const uint32_t *__restrict__ tile_start,
                                                               const uint32_t *__restrict__ tile_order, int ntiles,
                                                               uint64_t *__restrict__ keys,
                                                               uint32_t *__restrict__ point_list,
                                                               const int64_t *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t *s = reinterpret_cast<uint64_t *>(smem_raw);
    if (status[GA_STATUS_OVERFLOW]) return;
    (void)ntiles;
    const uint32_t tile = tile_order[blockIdx.x];
    const uint32_t beg = tile_start[tile], end = tile_start[tile + 1];
    const int n = (int)(end - beg);
    if (n <= 0) return;
    const int tid = threadIdx.x;
    if (n == 1) { if (tid == 0) point_list[beg] = (uint32_t)keys[beg]; return; }

    if (n <= kSortCap) {
        int np = 2; while (np < n) np <<= 1;
        for (int t = tid; t < np; t += 256) s[t] = t < n ? keys[beg + t] : ~0ull;
        __syncthreads();
        bitonic_sort_lds(s, np, tid, 256);
        for (int t = tid; t < n; t += 256) point_list[beg + t] = (uint32_t)s[t];
        return;
    }

    // Long segment: sort 8192-entry runs in LDS and write them back in place ...
    const int nruns = (n + kSortCap - 1) / kSortCap;
    for (int r = 0; r < nruns; ++r) {
        const int rb = r * kSortCap, rn = min(kSortCap, n - rb);
        int np = 2; while (np < rn) np <<= 1;
        for (int t = tid; t < np; t += 256) s[t] = t < rn ? keys[beg + rb + t] : ~0ull;
        __syncthreads();
        bitonic_sort_lds(s, np, tid, 256);
        for (int t = tid; t < rn; t += 256) keys[beg + rb + t] = s[t];
        __syncthreads();
    }
    __threadfence();
    __syncthreads();
    // ... then place every element at (its index in its own run) + sum over the other runs of (#keys smaller).
    // Keys are unique inside a tile, so ranks are a permutation.  Loads bypass the L1 (agent-scope atomics) because
    // the runs were rewritten by this very workgroup.
    for (int e = tid; e < n; e += 256) {
        const int er = e / kSortCap;
        const uint64_t key = __hip_atomic_load(keys + beg + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int rank = e - er * kSortCap;
        for (int r = 0; r < nruns; ++r) {
            if (r == er) continue;
            const int rb = r * kSortCap, rn = min(kSortCap, n - rb);
            int lo = 0, hi = rn;  // lower_bound
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const uint64_t kv = __hip_atomic_load(keys + beg + rb + mid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (kv < key) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        point_list[beg + rank] = (uint32_t)key;
    }
}

void launch_binning(const GaSurfelForwardArgs &a, const Dims &d, const Workspace &ws, hipStream_t s)
{
    const int nt = d.V * d.tiles;
    hipLaunchKernelGGL(surfel_tile_scan_kernel, dim3(1), dim3(1024), 0, s, ws.tile_count, ws.tile_start,
                       ws.tile_cursor, ws.tile_order, nt, a.capacity, ws.status);
    const dim3 grid((unsigned)((d.N + 256 * kBinSplats - 1) / (256 * kBinSplats)), (unsigned)d.V);
    if (d.tiles <= kLdsTiles)
        hipLaunchKernelGGL(surfel_fill_kernel<true>, grid, dim3(256), 2 * d.tiles * sizeof(uint32_t), s, ws.rect,
                           ws.depth, d, ws.tile_cursor, ws.keys, ws.status);
    else
        hipLaunchKernelGGL(surfel_fill_kernel<false>, grid, dim3(256), 0, s, ws.rect, ws.depth, d, ws.tile_cursor,
                           ws.keys, ws.status);
}

void launch_tile_sort(const GaSurfelForwardArgs &, const Dims &d, const Workspace &ws, hipStream_t s)
{
    const int nt = d.V * d.tiles;
    hipLaunchKernelGGL(surfel_tile_sort_kernel, dim3(nt), dim3(256), kSortCap * sizeof(uint64_t), s, ws.tile_start,
                       ws.tile_order, nt, ws.keys, ws.point_list, ws.status);
}

}  // namespace ga
