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
//   4. surfel_run_sort_kernel  : one workgroup per RUN (<= 2048 entries of one tile) sorts it by that 64-bit key in LDS
//                                (bitonic network, 16 KiB so 8+ workgroups share a CU).  Keys are unique per tile, so
//                                the result equals the stable radix sort by depth with index as the tie-break -- the
//                                order upstream's stable sort yields.  A tile that fits one run is finished here;
//   5. (last grid region of 4) : longer lists (several runs, sorted in parallel by different workgroups) are merged
//                                by rank counting: final position = index in own run + sum over the other runs of
//                                lower_bound(key) -- correct for any length, and the longest list of a real scene
//                                (~5 k entries) costs one 2048-sort plus 22 L2 probes per entry instead of a serial
//                                8192-wide network (measured 0.17 ms as the critical path, profiles/r1b_*).
// HBM traffic is 8 B written + 8 B read + 4 B written per entry instead of ~144 B per entry for a 6-pass radix sort,
// and the launch count is 4 instead of ~18.
#include "surfel_common.h"

#include <atomic>

namespace ga {

// ---------------------------------------------------------------------------------------------------------------
// 2. exclusive scan of the per-(view, tile) counters + the workgroup schedule of the per-tile kernels.
// A single workgroup of 1024 threads: everything here is latency, not throughput, so the kernel is organised to keep
// the dependent steps few -- counters are loaded 8 per thread at once into registers (chunks of 8192 = 8 views of
// 512 x 512), the eight wave scans of a chunk run back to back, the 128 wave totals are scanned by one wave, length
// classes are counted in wave-private LDS histograms (no contended atomics, no returned values) and ranks inside a
// class come from one returning LDS atomic per element on the wave's own histogram.
__device__ __forceinline__ int length_class(uint32_t c) { return c ? 32 - __builtin_clz(c) : 0; }  // 2^(b-1) <= c < 2^b

constexpr int kScanPer = 8;         // counters per thread and chunk
constexpr int kClasses = 33;
constexpr int kBigStash = 1024;     // (tile, count) of the lists longer than one sort run kept in LDS for the run table

struct ScanArgs {
    uint32_t *__restrict__ tile_count;
    uint32_t *__restrict__ seg_sync;
    uint32_t seg_sync_words;
    uint32_t *__restrict__ tile_start;
    uint32_t *__restrict__ tile_cursor;
    uint4 *__restrict__ tile_order;
    uint4 *__restrict__ run_table;
    int n;
    int64_t capacity, seg_capacity;
    uint32_t *__restrict__ seg_table;
    int64_t *__restrict__ status;
};

// kOwnLaunch (the only instantiation left): the scan is a launch of its own in front of the fill -- problems beyond the sizes
// surfel_fill_sched_kernel covers: it also sets the fill's cursors to the list begins and clears the tile counters it has consumed.
template <bool kOwnLaunch>
__device__ __forceinline__ void tile_scan_body(const ScanArgs &a)
{
    uint32_t *__restrict__ tile_count = a.tile_count;
    uint32_t *__restrict__ seg_sync = a.seg_sync;
    const uint32_t seg_sync_words = a.seg_sync_words;
    uint32_t *__restrict__ tile_start = a.tile_start;
    uint32_t *__restrict__ tile_cursor = a.tile_cursor;
    uint4 *__restrict__ tile_order = a.tile_order;
    uint4 *__restrict__ run_table = a.run_table;
    const int n = a.n;
    const int64_t capacity = a.capacity, seg_capacity = a.seg_capacity;
    uint32_t *__restrict__ seg_table = a.seg_table;
    int64_t *__restrict__ status = a.status;
    __shared__ uint32_t wt[kScanPer * 16], wt_ex[kScanPer * 16];
    __shared__ uint32_t hist[16][kClasses];       // wave-private class counts, later wave-private rank counters
    __shared__ uint32_t wave_off[16][kClasses];   // class start + population of the lower waves
    __shared__ uint32_t class_start[kClasses + 1];
    __shared__ uint32_t wave_tot[16];
    __shared__ uint64_t wide_tot[16];
    __shared__ uint32_t wave_max[16];
    __shared__ uint32_t carry_s, nbig_s;
    __shared__ uint32_t big_tile[kBigStash], big_cnt[kBigStash], big_beg[kBigStash];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 16 * kClasses; i += 1024) (&hist[0][0])[i] = 0;
    if (tid == 0) carry_s = 0;
    __syncthreads();

    const int nchunk = (n + kScanPer * 1024 - 1) / (kScanPer * 1024);
    uint32_t cnt[kScanPer], ex[kScanPer];  // the (last) chunk stays in registers for the second phase
    uint32_t local_max = 0;
    uint64_t local_sum = 0;
    // ---- phase A: tile_start / tile_cursor and the class histogram ------------------------------------------------
    for (int ch = 0; ch < nchunk; ++ch) {
        const int cbase = ch * kScanPer * 1024;
#pragma unroll
        for (int k = 0; k < kScanPer; ++k) {
            const int i = cbase + k * 1024 + tid;
            cnt[k] = i < n ? tile_count[i] : 0u;
        }
        uint32_t x[kScanPer];
#pragma unroll
        for (int k = 0; k < kScanPer; ++k) {
            local_max = max(local_max, cnt[k]);
            local_sum += cnt[k];
            if (cbase + k * 1024 + tid < n) atomicAdd(&hist[wid][length_class(cnt[k])], 1u);
            x[k] = cnt[k];  // inclusive scan inside the wave
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t y = __shfl_up(x[k], o, 64);
                if (lane >= o) x[k] += y;
            }
            if (lane == 63) wt[k * 16 + wid] = x[k];
        }
        __syncthreads();
        if (wid == 0) {  // exclusive scan of the 128 wave totals, two per lane, on top of the carry of earlier chunks
            const uint32_t v0 = wt[2 * lane], v1 = wt[2 * lane + 1];
            uint32_t sx = v0 + v1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t y = __shfl_up(sx, o, 64);
                if (lane >= o) sx += y;
            }
            const uint32_t carry = carry_s;
            wt_ex[2 * lane] = carry + sx - v0 - v1;
            wt_ex[2 * lane + 1] = carry + sx - v1;
            if (lane == 63) carry_s = carry + sx;  // read by everyone only after the next barrier
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kScanPer; ++k) {
            const int i = cbase + k * 1024 + tid;
            ex[k] = wt_ex[k * 16 + wid] + x[k] - cnt[k];
            if (i < n) { tile_start[i] = ex[k]; if (kOwnLaunch) tile_cursor[i] = ex[k]; }
        }
    }
    // ---- totals (64-bit: the uint32 running offsets above wrap past 2^32; that case is reported as overflow) -------
    {
        uint64_t wsum = local_sum;
        uint32_t wmax = local_max;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            wsum += __shfl_down(wsum, o, 64);
            wmax = max(wmax, (uint32_t)__shfl_down(wmax, o, 64));
        }
        if (lane == 0) { wide_tot[wid] = wsum; wave_max[wid] = wmax; }
    }
    __syncthreads();
    // class populations -> class starts, longest lists first (wave 0: lane b owns class 32 - b)
    if (wid == 0) {
        const int b = 32 - lane;
        uint32_t pop = 0;
        if (lane < kClasses)
            for (int w = 0; w < 16; ++w) pop += hist[w][b];
        uint32_t sx = pop;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(sx, o, 64);
            if (lane >= o) sx += y;
        }
        if (lane < kClasses) {
            class_start[b] = sx - pop;
            // every list longer than one sort run lives in a class >= kBigBucket: remember where those classes end
            constexpr int kBigBucket = 32 - __builtin_clz((unsigned)kSortCap + 1u);
            if (b == kBigBucket) nbig_s = sx;
        }
        uint32_t seg_wx = 0;
        {   // segmented blend: a tile of class b >= kSegClass becomes seg_count(b) work items; the classes are laid out
            // longest first, so lane order is work order.  seg_table[b] = (first tile_order slot, first work item)
            const uint32_t segs = (lane < kClasses && b >= kSegClass) ? pop * (uint32_t)seg_count(b) : 0u;
            uint32_t wx = segs;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t y = __shfl_up(wx, o, 64);
                if (lane >= o) wx += y;
            }
            if (lane < kClasses) {
                seg_table[2 * b] = sx - pop;
                seg_table[2 * b + 1] = wx - segs;
                if (b == kSegClass) {
                    status[GA_STATUS_LONG_TILES] = (int64_t)sx;
                    status[GA_STATUS_SEG_WORK] = (int64_t)wx;
                }
            }
            seg_wx = wx;
            if (lane == 0) {   // a new epoch for this launch's exchange words (never 0; any start value will do)
                uint32_t e = seg_table[kSegEpochWord] + 1u;
                if (e == 0u) e = 1u;
                seg_table[kSegEpochWord] = e;
            }
        }
        uint64_t total = lane < 16 ? wide_tot[lane] : 0ull;
        uint32_t mx = lane < 16 ? wave_max[lane] : 0u;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            total += __shfl_down(total, o, 64);
            mx = max(mx, (uint32_t)__shfl_down(mx, o, 64));
        }
        // segment work items of this launch (the inclusive scan value of lane 32 - kSegClass, class kSegClass being the last
        // segmented one): more than the exchange scratch holds is an overflow like D > capacity -- nothing is rendered, the
        // host reads GA_STATUS_SEG_WORK and grows seg_capacity
        const uint32_t seg_total = __shfl(seg_wx, 32 - kSegClass, 64);
        if (lane == 0) {
            tile_start[n] = (uint32_t)total;
            status[GA_STATUS_NUM_RENDERED] = (int64_t)total;
            status[GA_STATUS_OVERFLOW] = (total > (uint64_t)capacity || total > 0xFFFFFFFFull || (int64_t)seg_total > seg_capacity) ? 1 : 0;
            status[GA_STATUS_MAX_TILE] = (int64_t)mx;
        }
    }
    __syncthreads();
    for (int i = tid; i < 16 * kClasses; i += 1024) {  // (wave, class): start of the class + population of lower waves
        const int w = i / kClasses, b = i - w * kClasses;
        uint32_t off = class_start[b];
        for (int w2 = 0; w2 < w; ++w2) off += hist[w2][b];
        wave_off[w][b] = off;
    }
    __syncthreads();
    for (int i = tid; i < 16 * kClasses; i += 1024) (&hist[0][0])[i] = 0;  // now the per-wave rank counters
    __syncthreads();
    // ---- phase B: workgroup schedule of the per-tile kernels: tiles ordered by length class, longest first, so the
    // long serial chains start at once and the short ones fill in behind them (order inside a class is irrelevant) ---
    const uint32_t nbig = nbig_s;
    for (int ch = 0; ch < nchunk; ++ch) {
        const int cbase = ch * kScanPer * 1024;
        if (nchunk > 1) {
#pragma unroll
            for (int k = 0; k < kScanPer; ++k) {
                const int i = cbase + k * 1024 + tid;
                cnt[k] = i < n ? tile_count[i] : 0u;
                ex[k] = i < n ? tile_start[i] : 0u;  // written by this thread in phase A
            }
        }
#pragma unroll
        for (int k = 0; k < kScanPer; ++k) {
            const int i = cbase + k * 1024 + tid;
            if (i < n) {
                const int b = length_class(cnt[k]);
                const uint32_t pos = wave_off[wid][b] + atomicAdd(&hist[wid][b], 1u);
                tile_order[pos] = make_uint4((uint32_t)i, ex[k], cnt[k], 0u);
                if (pos < nbig && pos < (uint32_t)kBigStash) { big_tile[pos] = (uint32_t)i; big_cnt[pos] = cnt[k]; big_beg[pos] = ex[k]; }
            }
        }
    }
    // ---- run table of the per-tile sort: run 0 of every tile is implicit; list the runs 1.. of the lists longer than
    // one sort run (they sit at the front of tile_order) as (tile, run) pairs so that each gets its own workgroup -------
    if (nbig > (uint32_t)kBigStash) __threadfence();  // the overflow of the stash is re-read from tile_order
    __syncthreads();
    if (tid == 0) carry_s = 0;
    __syncthreads();
    const uint32_t table_cap = (uint32_t)(capacity / kSortCap + 1);
    for (uint32_t base = 0; base < nbig; base += 1024) {
        const uint32_t p = base + tid;
        uint32_t extra = 0, c = 0, tb = 0;
        if (p < nbig) {
            if (p < (uint32_t)kBigStash) { c = big_cnt[p]; tb = big_beg[p]; }
            else {
                const uint32_t *rec = reinterpret_cast<const uint32_t *>(tile_order + p);
                tb = __hip_atomic_load(rec + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                c = __hip_atomic_load(rec + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            extra = c > (uint32_t)kSortCap ? (c - 1) / kSortCap : 0;
        }
        uint32_t x = extra;
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
        uint32_t dst = carry + wbase + x - extra;
        for (uint32_t r = 1; r <= extra; ++r, ++dst)
            if (dst < table_cap) run_table[dst] = make_uint4(p, r, tb, c);   // (schedule slot of the list, run, list begin, list length)
        __syncthreads();
        if (tid == 1023) carry_s = carry + wbase + x;
        __syncthreads();
    }
    if (tid == 0) status[GA_STATUS_EXTRA_RUNS] = (int64_t)min(carry_s, table_cap);
    // Leave the accumulating words of the workspace head clean for the NEXT launch (the tile counters this kernel has consumed;
    // the arrival / saturation words, ticket and statistics words the blend of THIS launch starts from): a caller that passes
    // GA_SURFEL_FLAG_WORKSPACE_CLEAN then needs no clearing memset in front of the next forward (4.7 us + a launch boundary).
    if (kOwnLaunch)
        for (int i = tid; i < n; i += 1024) tile_count[i] = 0u;
    for (uint32_t i = tid; i < seg_sync_words; i += 1024) seg_sync[i] = 0u;
    if (tid >= 4 && tid < GA_STATUS_WORDS && tid != GA_STATUS_LONG_TILES && tid != GA_STATUS_SEG_WORK) status[tid] = 0;
}

__global__ __launch_bounds__(1024) void surfel_tile_scan_kernel(ScanArgs a) { tile_scan_body<true>(a); }

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
// 2 + 3 in ONE launch (the common sizes: views of at most kLdsTiles tiles, fewer than 65536 (view, tile) counters in all).
// Round 3 ran a single-workgroup scan (17 us of pure latency) in front of a fill that is ~8 us of latency itself; neither needs the
// other's RESULT if every workgroup derives what it needs on its own:
//   fill workgroups (grid rows 1 .. V, one view each, kFillSplats consecutive Gaussians per thread):
//       begin(v, t) = entries of the views before v  (view_total[], accumulated by the preprocess: 64 V words)
//                   + exclusive scan of view v's OWN tile counters up to t (<= kLdsTiles words from L2, scanned in LDS);
//       slots inside a list are claimed from a RELATIVE cursor (zero between launches);
//   schedule workgroups (grid row 0, the first nsched blocks; the others leave): tile_order (longest lists first, by length class) is
//       a counting sort over ALL V * tiles counters.  Each of them histograms all counters (lane-private LDS bins), which gives it the
//       class starts AND, for each class, the number of tiles in front of its own slice of ~1024 tiles; it then writes the schedule
//       entries, tile_start[] and the run-table entries of its slice (positions inside (class, slice) from LDS atomics: the order
//       inside a class is irrelevant), clears its share of the segment words, and block 0 writes the status words, the segment table and
//       the launch epoch.
// First cut of round 4: ONE schedule workgroup (the round-3 scan) beside the fill -- still the launch's critical path
// (tools/fill_stamps.py: 17.6 us, the fill workgroups were done after 8); the schedule work inside every fill workgroup (no row 0):
// 89 VGPRs, one 1024-thread workgroup per CU instead of two, 25 us.
#ifndef GA_FILL_THREADS
#define GA_FILL_THREADS 512
#endif
constexpr int kFT = GA_FILL_THREADS, kFW = kFT / 64;    // threads / waves of a workgroup of the fill launch
constexpr int kFillSplats = 2048 / kFT;                 // 2048 Gaussians per fill workgroup
constexpr int kSchedPre = 16;                       // (view, tile) counters per thread a schedule workgroup has in flight (8 views of 512 x 512 at 512 threads: all)

#ifdef GA_FILL_STAMPS   // measurement build: (start, end, row) per workgroup in the segment scratch (unused until the blend), 100 MHz clock
struct FillStamp {
    unsigned long long *p, t0;
    __device__ ~FillStamp() { if (threadIdx.x == 0) { p[0] = t0; p[1] = __builtin_amdgcn_s_memrealtime(); p[2] = 0x5A5A000000000000ull | blockIdx.y; } }
};
#endif

__device__ __forceinline__ void schedule_slice(const ScanArgs &sa, int nall, int nsched, uint32_t *__restrict__ big_scratch)
{
    __shared__ uint32_t histl[kClasses][64];     // one copy per lane: low half = tiles of class b, high half = those in front of my slice
    __shared__ uint32_t cls_all[kClasses], cls_before[kClasses], cls_start[kClasses], cls_work[kClasses], slice_cnt[kClasses];
    __shared__ uint32_t nbig_sh, long_tiles_sh, seg_total_sh, wave_tot[kFW], wave_maxc[kFW];
    __shared__ unsigned long long wave_front[kFW], wave_sum64[kFW];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int Q = (nall + nsched - 1) / nsched, S = (int)blockIdx.x * Q, S1 = min(nall, S + Q);   // my slice of the global tile index
    // the lists longer than one sort run (index, length), noted by this workgroup for its own use: at most capacity / kSortCap of them
    // when nothing overflows -- in my share of point_list, which nobody touches before the sort
    const uint32_t big_cap = (uint32_t)(sa.capacity / kSortCap + 1);
    uint32_t *big_i = big_scratch + (size_t)blockIdx.x * 2 * big_cap, *big_c = big_i + big_cap;
    for (int i = tid; i < kClasses * 64; i += kFT) (&histl[0][0])[i] = 0;
    if (tid < kClasses) slice_cnt[tid] = 0;
    if (tid == 0) nbig_sh = 0;
    __syncthreads();
    // class histogram of ALL lists (and of those in front of my slice), the entries in front of my slice, the lists longer than one
    // sort run, the longest list.  64 lane-private copies of the 33 bins: the atomics of one instruction never meet (a wave's 64
    // counters fall into a handful of classes, and LDS atomics of 64 lanes on ~5 addresses are served one lane at a time)
    uint32_t maxc = 0;
    unsigned long long front = 0, sum = 0;
    auto note = [&](int i, uint32_t c) {
        if (i < nall) {
            atomicAdd(&histl[length_class(c)][lane], 1u | (i < S ? 0x10000u : 0u));
            maxc = max(maxc, c);
            sum += c;
            if (i < S) front += c;
            if (c > (uint32_t)kSortCap) {
                const uint32_t q = atomicAdd(&nbig_sh, 1u);
                if (q < big_cap) { big_i[q] = (uint32_t)i; big_c[q] = c; }
            }
        }
    };
    for (int j0 = 0; j0 * kFT < nall; j0 += kSchedPre) {   // kSchedPre loads in flight
        uint32_t ca[kSchedPre];
#pragma unroll
        for (int j = 0; j < kSchedPre; ++j) {
            const int i = (j0 + j) * kFT + tid;
            ca[j] = i < nall ? sa.tile_count[i] : 0u;
        }
#pragma unroll
        for (int j = 0; j < kSchedPre; ++j) note((j0 + j) * kFT + tid, ca[j]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        maxc = max(maxc, (uint32_t)__shfl_down(maxc, o, 64));
        front += __shfl_down(front, o, 64);
        sum += __shfl_down(sum, o, 64);
    }
    if (lane == 0) { wave_maxc[wid] = maxc; wave_front[wid] = front; wave_sum64[wid] = sum; }
    __threadfence_block();    // (the noted lists are read by other threads of this workgroup)
    __syncthreads();
    for (int b = wid; b < kClasses; b += kFW) {   // a wave sums the 64 copies of a class (both halves at once)
        uint32_t h = histl[b][lane];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) h += __shfl_down(h, o, 64);
        if (lane == 0) { cls_all[b] = h & 0xFFFFu; cls_before[b] = h >> 16; }
    }
    __syncthreads();
    if (wid == 0) {   // class starts, longest lists first (lane b owns class 32 - b); segment work items of the classes in front
        const int b = 32 - lane;
        const uint32_t pop = lane < kClasses ? cls_all[b] : 0u;
        uint32_t sx = pop;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(sx, o, 64);
            if (lane >= o) sx += y;
        }
        const uint32_t segs = (lane < kClasses && b >= kSegClass) ? pop * (uint32_t)seg_count(b) : 0u;
        uint32_t wx = segs;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(wx, o, 64);
            if (lane >= o) wx += y;
        }
        if (lane < kClasses) { cls_start[b] = sx - pop; cls_work[b] = wx - segs; }
        if (b == kSegClass) { long_tiles_sh = sx; seg_total_sh = wx; }
    }
    unsigned long long fsum = 0, total = 0;
    uint32_t mx = 0;
#pragma unroll 2
    for (int w = 0; w < kFW; ++w) { fsum += wave_front[w]; total += wave_sum64[w]; mx = max(mx, wave_maxc[w]); }
    __syncthreads();
    const uint32_t long_tiles = long_tiles_sh, seg_total = seg_total_sh;
    const bool overflow = total > (unsigned long long)sa.capacity || total > 0xFFFFFFFFull || (int64_t)seg_total > sa.seg_capacity;
    const uint32_t nbig = min(nbig_sh, big_cap);
    const uint32_t table_cap = (uint32_t)(sa.capacity / kSortCap + 1);
    // the schedule entries, list begins and run-table entries of my slice, kFT tiles per round
    uint32_t carry = (uint32_t)fsum;
    for (int g0 = S; g0 < S1; g0 += kFT) {
        const int gi = g0 + tid;
        const uint32_t c = gi < S1 ? sa.tile_count[gi] : 0u;
        uint32_t x = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wave_tot[wid] = x;
        __syncthreads();
        uint32_t wbase = 0, round = 0;
#pragma unroll 4
        for (int w = 0; w < kFW; ++w) { wbase += w < wid ? wave_tot[w] : 0u; round += wave_tot[w]; }
        if (gi < S1) {
            const uint32_t beg = carry + wbase + x - c;
            const int b = length_class(c);
            const uint32_t pos = cls_start[b] + cls_before[b] + atomicAdd(&slice_cnt[b], 1u);
            sa.tile_order[pos] = make_uint4((uint32_t)gi, beg, c, 0u);
            sa.tile_start[gi] = beg;
            if (c > (uint32_t)kSortCap) {   // runs 1.. of a list longer than one sort run: table slots in the order of the tile index
                uint32_t dst = 0;
                for (uint32_t q = 0; q < nbig; ++q)
                    if (big_i[q] < (uint32_t)gi) dst += (big_c[q] - 1u) / kSortCap;
                for (uint32_t r = 1; r <= (c - 1u) / kSortCap; ++r, ++dst)
                    if (dst < table_cap) sa.run_table[dst] = make_uint4(pos, r, beg, c);
            }
        }
        carry += round;
        __syncthreads();
    }
    // the segment words, cleared in slices for the blend of THIS launch (GA_SURFEL_FLAG_WORKSPACE_CLEAN: see surfel_run_sort_kernel)
    {
        const uint32_t w0 = (uint32_t)((uint64_t)sa.seg_sync_words * blockIdx.x / nsched), w1 = (uint32_t)((uint64_t)sa.seg_sync_words * (blockIdx.x + 1) / nsched);
        for (uint32_t i = w0 + (uint32_t)tid; i < w1; i += kFT) sa.seg_sync[i] = 0u;
    }
    if (blockIdx.x == 0) {   // the launch's single words
        if (tid < kClasses) { sa.seg_table[2 * tid] = cls_start[tid]; sa.seg_table[2 * tid + 1] = cls_work[tid]; }
        if (tid == 0) {
            uint32_t extra = 0;
            for (uint32_t q = 0; q < nbig; ++q) extra += (big_c[q] - 1u) / kSortCap;
            sa.tile_start[nall] = (uint32_t)total;
            sa.status[GA_STATUS_NUM_RENDERED] = (int64_t)total;
            sa.status[GA_STATUS_OVERFLOW] = overflow ? 1 : 0;
            sa.status[GA_STATUS_MAX_TILE] = (int64_t)mx;
            sa.status[GA_STATUS_EXTRA_RUNS] = (int64_t)min(extra, table_cap);
            sa.status[GA_STATUS_LONG_TILES] = (int64_t)long_tiles;
            sa.status[GA_STATUS_SEG_WORK] = (int64_t)seg_total;
            uint32_t e = sa.seg_table[kSegEpochWord] + 1u;   // a new epoch for this launch's exchange words (never 0)
            if (e == 0u) e = 1u;
            sa.seg_table[kSegEpochWord] = e;
        }
        if (tid >= 4 && tid < GA_STATUS_WORDS && tid != GA_STATUS_LONG_TILES && tid != GA_STATUS_SEG_WORK) sa.status[tid] = 0;
    }
}

__global__ __launch_bounds__(kFT) void surfel_fill_sched_kernel(ScanArgs sa, const uint16_t *__restrict__ rect,
                                                                 const float *__restrict__ depth, Dims dm,
                                                                 const unsigned long long *__restrict__ view_total,
                                                                 uint64_t *__restrict__ keys, int nsched, uint32_t *__restrict__ big_scratch,
                                                                 unsigned long long *__restrict__ dbg)
{
    extern __shared__ uint32_t lds[];  // [tiles] counts -> ranks, [tiles] list begins -> segment bases of this workgroup
#ifdef GA_FILL_STAMPS
    FillStamp stamp{dbg + 3300000 + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4, __builtin_amdgcn_s_memrealtime()};   // (behind the words the blend's segments use at BASELINE configs[1])
#endif
    if (blockIdx.y == 0) {
        if ((int)blockIdx.x < nsched) schedule_slice(sa, dm.V * dm.tiles, nsched, big_scratch);
        return;
    }
    __shared__ uint32_t wave_sum[kFW];
    __shared__ unsigned long long wave_base[kFW], wave_all[kFW];
    const int v = (int)blockIdx.y - 1, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int T = dm.tiles;
    uint32_t *cnt = lds, *basep = lds + T;
    const uint32_t *tcv = sa.tile_count + (size_t)v * T;
    uint32_t *cur = sa.tile_cursor + (size_t)v * T;
    // everything this workgroup reads from memory, requested together
    ushort4 rcs[kFillSplats];
#pragma unroll
    for (int k = 0; k < kFillSplats; ++k) {
        const int i = (blockIdx.x * kFillSplats + k) * kFT + tid;
        rcs[k] = i < dm.N ? *reinterpret_cast<const ushort4 *>(rect + 4 * ((size_t)v * dm.N + i)) : make_ushort4(0, 0, 0, 0);
    }
    static_assert(kLdsTiles % kFT == 0, "a thread holds kLdsTiles / kFT consecutive tile counters");
    const int per = (T + kFT - 1) / kFT;       // consecutive tile counters per thread (<= kLdsTiles / kFT because T <= kLdsTiles)
    uint32_t tc[kLdsTiles / kFT];
#pragma unroll
    for (int j = 0; j < kLdsTiles / kFT; ++j) {
        const int t = tid * per + j;
        tc[j] = (j < per && t < T) ? tcv[t] : 0u;
    }
    unsigned long long before = 0, all = 0;     // entries of the views before mine / of all views
    for (int u = tid; u < dm.V * kViewSlots; u += kFT) {
        const unsigned long long c = view_total[u];
        all += c;
        if (u < v * kViewSlots) before += c;
    }
    for (int t = tid; t < T; t += kFT) cnt[t] = 0;
    // exclusive scan of my view's counters: thread-local, wave (shuffles), 16 wave totals
    uint32_t loc = 0;
#pragma unroll
    for (int j = 0; j < kLdsTiles / kFT; ++j) loc += tc[j];
    uint32_t x = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        before += __shfl_down(before, o, 64);
        all += __shfl_down(all, o, 64);
    }
    if (lane == 63) wave_sum[wid] = x;
    if (lane == 0) { wave_base[wid] = before; wave_all[wid] = all; }
    __syncthreads();
    unsigned long long vbase = 0, total = 0;
    uint32_t wbase = 0;
#pragma unroll
    for (int w = 0; w < kFW; ++w) {
        vbase += wave_base[w];
        total += wave_all[w];
        wbase += w < wid ? wave_sum[w] : 0u;
    }
    // the overflow the schedule workgroups report in the status words (D > capacity): nothing may be written
    if (total > (unsigned long long)sa.capacity || total > 0xFFFFFFFFull) return;
    {
        uint32_t run = (uint32_t)vbase + wbase + x - loc;
#pragma unroll
        for (int j = 0; j < kLdsTiles / kFT; ++j) {
            const int t = tid * per + j;
            if (j < per && t < T) basep[t] = run;
            run += tc[j];
        }
    }
    // a) the workgroup's entries per tile
#pragma unroll
    for (int k = 0; k < kFillSplats; ++k)
        for (int ty = rcs[k].y; ty < rcs[k].w; ++ty)
            for (int tx = rcs[k].x; tx < rcs[k].z; ++tx) atomicAdd(cnt + ty * dm.gx + tx, 1u);
    __syncthreads();
    // b) reserve [base, base + count) in each touched tile's list: ONE returning atomic on the tile's relative cursor
    for (int t = tid; t < T; t += kFT) {
        const uint32_t c = cnt[t];
        if (c) { basep[t] += atomicAdd(cur + t, c); cnt[t] = 0; }
    }
    __syncthreads();
    // c) ranks inside the workgroup from returning LDS atomics; keys
#pragma unroll
    for (int k = 0; k < kFillSplats; ++k) {
        const ushort4 rc = rcs[k];
        if (rc.z <= rc.x || rc.w <= rc.y) continue;
        const int i = (blockIdx.x * kFillSplats + k) * kFT + tid;
        const uint64_t key = ((uint64_t)__float_as_uint(depth[(size_t)v * dm.N + i]) << 32) | (uint32_t)i;
        for (int ty = rc.y; ty < rc.w; ++ty)
            for (int tx = rc.x; tx < rc.z; ++tx) {
                const int t = ty * dm.gx + tx;
                keys[basep[t] + atomicAdd(cnt + t, 1u)] = key;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 4. per-tile sort.  LDS bitonic network on u64 keys, padded with ~0 to the next power of two.
// Register-blocked: a thread holds E = 2^e keys whose indices differ in e consecutive bits [p, p+e) and performs up to
// e consecutive stages of the network on them in registers -- one LDS round trip per group of stages instead of one per
// stage (26 instead of 66 for a 2048-key run).
//
// The network is the direction-free form of the bitonic sorter: the block that merges sorted 2^(L-1)-subsequences into
// sorted 2^L-subsequences starts with a "flip" stage pairing i with i ^ (2^L - 1), followed by the ordinary stages
// i <-> i ^ 2^b (b = L-2 .. 0); every comparator is ascending.  A thread realises the flip by loading its upper-half
// keys from the mirrored positions i ^ (2^(L-1) - 1); for the rest of that group its upper-half register slots are in
// mirrored order, so their comparators are written the other way round -- all of it fixed at compile time.
//
// A compare-exchange is v_min_f64 + v_max_f64 on the raw key bits: keys are (depth bits << 32) | index with depth finite
// and > 0.2 (near cull in the preprocess), so they are the bit patterns of positive normal doubles, whose order is the
// unsigned order of the bits; padding is +infinity.  (A 64-bit integer compare and four selects cost about three times as
// many issue cycles; PMC showed the sort to be instruction-issue bound: 14 M VALU + 10 M SALU wave instructions.)
constexpr uint64_t kSortPad = 0x7FF0000000000000ull;

// LDS slot of key i: an XOR swizzle of the low five index bits (32 eight-byte keys span the 64 banks) that makes the
// gathers of the frequent groups (held bits at 0, 3, 6 for eight keys per thread) conflict-free -- unswizzled, a thread
// reading its eight consecutive keys is an 8-way bank conflict.
__device__ __forceinline__ int sort_slot(int i) { return i ^ ((i >> 5) & 7) ^ (((i >> 6) & 3) << 3); }

// v_min_f64 / v_max_f64 as such: through __builtin_fmin the compiler first canonicalises every loaded value (a
// v_max_f64 x, x per key and group) for the sake of signalling NaNs, which these keys never are.
__device__ __forceinline__ double key_min(double x, double y)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ double key_max(double x, double y)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

// Top group of block L: flip at slot bit FB (index bit L-1), then G-1 ordinary stages below it.
template <int E, int FB, int G>
__device__ __forceinline__ void sort_group_top(double *s, int base, int p, int L)
{
    // the swizzle is linear over XOR and base, m << p are bit-disjoint: slot(base | m << p) = slot(base) ^ slot(m << p),
    // where the second factor (and the mirror's) is wave-uniform -- one vector XOR per key
    // (byte offsets, so that the vector side is the XOR alone)
    const int sb = sort_slot(base) << 3, smirror = sort_slot((1 << (L - 1)) - 1) << 3;
    char *sc = reinterpret_cast<char *>(s);
    double v[E];
    int slot[E];
#pragma unroll
    for (int m = 0; m < E; ++m) {
        slot[m] = sb ^ ((sort_slot(m << p) << 3) ^ ((m & (1 << FB)) ? smirror : 0));
        v[m] = *reinterpret_cast<double *>(sc + slot[m]);
    }
#pragma unroll
    for (int q = FB; q > FB - G; --q) {
#pragma unroll
        for (int m = 0; m < E; ++m) {
            if (m & (1 << q)) continue;
            const double x = v[m], y = v[m | (1 << q)];
            const double mn = key_min(x, y), mx = key_max(x, y);
            const bool mirrored = q < FB && (m & (1 << FB));
            v[m] = mirrored ? mx : mn;
            v[m | (1 << q)] = mirrored ? mn : mx;
        }
    }
#pragma unroll
    for (int m = 0; m < E; ++m) *reinterpret_cast<double *>(sc + slot[m]) = v[m];
}

// Lower group: e ordinary stages on the held bits [p, p+e).
template <int E>
__device__ __forceinline__ void sort_group_low(double *s, int base, int p)
{
    constexpr int e = E == 8 ? 3 : (E == 4 ? 2 : 1);
    double v[E];
    int slot[E];
    const int sb = sort_slot(base) << 3;
    char *sc = reinterpret_cast<char *>(s);
#pragma unroll
    for (int m = 0; m < E; ++m) {
        slot[m] = sb ^ (sort_slot(m << p) << 3);
        v[m] = *reinterpret_cast<double *>(sc + slot[m]);
    }
#pragma unroll
    for (int q = e - 1; q >= 0; --q) {
#pragma unroll
        for (int m = 0; m < E; ++m) {
            if (m & (1 << q)) continue;
            const double x = v[m], y = v[m | (1 << q)];
            v[m] = key_min(x, y);
            v[m | (1 << q)] = key_max(x, y);
        }
    }
#pragma unroll
    for (int m = 0; m < E; ++m) *reinterpret_cast<double *>(sc + slot[m]) = v[m];
}

template <int E, bool kWave = false>
__device__ __forceinline__ void bitonic_sort_blocked(double *s, int np, int tid)
{
    constexpr int e = E == 8 ? 3 : (E == 4 ? 2 : 1);
    const int T = np / E;
    const bool act = tid < T;
    const int Ltot = 31 - __builtin_clz((unsigned)np);
    int prev_p = 0;  // the caller has a barrier after filling the array
    // With the held bits at or below bit 6 the 64 threads of a wave own the contiguous keys [64 E w, +64 E) in this group
    // and in the previous one (mirroring stays inside it): no workgroup barrier between two such groups, LDS operations
    // of one wave execute in order.
    auto sync = [&](int p) {
        if (!kWave && (p > 6 || prev_p > 6)) __syncthreads();
        else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        prev_p = p;
    };
    for (int L = 1; L <= Ltot; ++L) {
        const int g = (L % e) ? (L % e) : e;  // stages of the top group; the groups below it are full
        const int p = max(0, L - e);
        sync(p);
        if (act) {
            const int base = ((tid >> p) << (p + e)) | (tid & ((1 << p) - 1));
            if (L >= e) {
                if (e == 1 || g == e) sort_group_top<E, e - 1, e>(s, base, p, L);
                else if (g == 1) sort_group_top<E, e - 1, 1>(s, base, p, L);
                else sort_group_top<E, e - 1, (e > 2 ? 2 : 1)>(s, base, p, L);
            } else if (L == 1) sort_group_top<E, 0, 1>(s, base, p, L);
            else sort_group_top<E, (e > 1 ? 1 : 0), (e > 1 ? 2 : 1)>(s, base, p, L);
        }
        for (int hi = L - 1 - g; hi >= 0; hi -= e) {
            const int pl = hi - e + 1;
            sync(pl);
            if (act) sort_group_low<E>(s, ((tid >> pl) << (pl + e)) | (tid & ((1 << pl) - 1)), pl);
        }
    }
    if (kWave) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
}

__device__ __forceinline__ void bitonic_sort_lds(double *s, int np, int tid)
{
    // eight keys per thread as soon as that fills half a wave: fewer, fatter groups beat more active threads
    if (np >= 256) bitonic_sort_blocked<8>(s, np, tid);
    else if (np >= 64) bitonic_sort_blocked<4>(s, np, tid);
    else bitonic_sort_blocked<2>(s, np, tid);
}


constexpr int kWaveSort = 512;   // lists shorter than this are sorted by one wavefront with the keys in registers (64 lanes x <= 8 keys)

// ---- wave sort: E keys per lane IN REGISTERS (key index i = lane * E + m), no LDS image at all.
// The LDS network above costs ~12 VALU instructions per key and stage on short lists (slot arithmetic, 64-bit moves, a quarter of the
// lanes active for 256 keys): the launch was issue-bound at 14 M wave instructions.  Here the same direction-free network runs on
// registers: the stages whose partner differs in the low log2(E) index bits are v_min_f64 / v_max_f64 on two registers of the lane; a
// stage whose partner sits in another lane fetches the partner's key with two cross-lane moves (DPP inside a quad, ds_swizzle inside
// 32 lanes, ds_bpermute beyond -- no memory access), compares once (v_cmp_lt_u64) and keeps the smaller key in the lower lane.
template <int MASK>
__device__ __forceinline__ uint32_t xor_lane_u32(uint32_t x, int lane)
{
    if constexpr (MASK == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xf, 0xf, true);        // quad_perm [1,0,3,2]
    else if constexpr (MASK == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    else if constexpr (MASK == 3) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x1B, 0xf, 0xf, true);   // quad_perm [3,2,1,0]
    else if constexpr (MASK < 32) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x001F | (MASK << 10));   // bit mode: and 0x1f, xor MASK
    else return (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ MASK) << 2, (int)x);
}
template <int MASK>
__device__ __forceinline__ unsigned long long xor_lane_u64(unsigned long long x, int lane)
{
    return ((unsigned long long)xor_lane_u32<MASK>((uint32_t)(x >> 32), lane) << 32) | xor_lane_u32<MASK>((uint32_t)x, lane);
}

// one cross-lane stage: key[m] against the key[PM(m)] of lane ^ LMASK; the lane whose bit LBIT is clear keeps the smaller one
template <int E, int LMASK, int LBIT, bool MIRROR>
__device__ __forceinline__ void wave_stage_cross(unsigned long long (&key)[E], int lane)
{
    const bool lower = (lane & LBIT) == 0;
    constexpr int B = E < 8 ? E : 8;    // keys in flight (registers); a mirrored stage pairs block b with block E/B - 1 - b
    if constexpr (!MIRROR || E <= B) {
#pragma unroll
        for (int m0 = 0; m0 < E; m0 += B) {
            unsigned long long theirs[B];
#pragma unroll
            for (int m = 0; m < B; ++m) theirs[m] = xor_lane_u64<LMASK>(key[MIRROR ? E - 1 - (m0 + m) : m0 + m], lane);
#pragma unroll
            for (int m = 0; m < B; ++m) {
                const bool take = lower ? theirs[m] < key[m0 + m] : theirs[m] > key[m0 + m];
                key[m0 + m] = take ? theirs[m] : key[m0 + m];
            }
        }
    } else {
        // register m reads the partner's register E-1-m: the block that is read must not have been overwritten yet, so a pair of
        // mirrored blocks is fetched together
#pragma unroll
        for (int m0 = 0; m0 < E / 2; m0 += B) {
            unsigned long long ta[B], tb[B];
#pragma unroll
            for (int m = 0; m < B; ++m) {
                ta[m] = xor_lane_u64<LMASK>(key[E - 1 - (m0 + m)], lane);       // partners of registers m0 + m
                tb[m] = xor_lane_u64<LMASK>(key[m0 + m], lane);               // partners of registers E - 1 - (m0 + m)
            }
#pragma unroll
            for (int m = 0; m < B; ++m) {
                const int x = m0 + m, y = E - 1 - (m0 + m);
                const bool takex = lower ? ta[m] < key[x] : ta[m] > key[x];
                const bool takey = lower ? tb[m] < key[y] : tb[m] > key[y];
                key[x] = takex ? ta[m] : key[x];
                key[y] = takey ? tb[m] : key[y];
            }
        }
    }
}

// in-register compare-exchange: key[a] <= key[b] afterwards (keys are bit patterns of positive doubles, see key_min)
template <int E>
__device__ __forceinline__ void wave_ce(unsigned long long (&key)[E], int a, int b)
{
    const double x = __longlong_as_double((long long)key[a]), y = __longlong_as_double((long long)key[b]);
    key[a] = (unsigned long long)__double_as_longlong(key_min(x, y));
    key[b] = (unsigned long long)__double_as_longlong(key_max(x, y));
}

// ordinary stages j = J, J / 2, ..., 1 of a merge block (partner i ^ j, ascending), J a compile-time power of two
template <int E, int J>
__device__ __forceinline__ void wave_stages_down(unsigned long long (&key)[E], int lane)
{
    if constexpr (J >= 1) {
        if constexpr (J >= E) wave_stage_cross<E, J / E, J / E, false>(key, lane);
        else {
#pragma unroll
            for (int m = 0; m < E; ++m)
                if ((m & J) == 0) wave_ce<E>(key, m, m | J);
        }
        wave_stages_down<E, J / 2>(key, lane);
    }
}

// merge blocks K = 2, 4, ..., 64 E: the flip stage (partner i ^ (K - 1)), then the ordinary stages K / 4 .. 1
template <int E, int K>
__device__ __forceinline__ void wave_blocks(unsigned long long (&key)[E], int lane)
{
    if constexpr (K <= 64 * E) {
        if constexpr (K <= E) {
#pragma unroll
            for (int m = 0; m < E; ++m)
                if ((m & (K / 2)) == 0) wave_ce<E>(key, m, m ^ (K - 1));
        } else {
            wave_stage_cross<E, K / E - 1, K / (2 * E), true>(key, lane);    // lane ^ (K/E - 1), register E-1-m; lower: index bit K/2 clear
        }
        wave_stages_down<E, K / 4>(key, lane);
        wave_blocks<E, K * 2>(key, lane);
    }
}

// sort keys[beg, beg + n) (n <= 64 E) into point_list
template <int E>
__device__ __forceinline__ void wave_sort_list(const uint64_t *__restrict__ keys, uint32_t *__restrict__ point_list, uint32_t beg, int n, int lane)
{
    unsigned long long key[E];
#pragma unroll
    for (int m = 0; m < E; ++m) {
        const int t = lane * E + m;
        key[m] = t < n ? keys[beg + t] : kSortPad;
    }
    wave_blocks<E, 2>(key, lane);
#pragma unroll
    for (int m = 0; m < E; ++m) {
        const int t = lane * E + m;
        if (t < n) point_list[beg + t] = (uint32_t)key[m];
    }
}

// Workgroup -> (run, list begin, list length): one 16-byte load.  The first `max_extra` workgroups take the runs 1.. of
// the long lists from the run table (so the long lists start first), the rest take run 0 of schedule slot b - max_extra.
// The status words are read together with it (independent loads, one latency).
__device__ __forceinline__ bool sort_block_assignment(const uint4 *__restrict__ tile_order,
                                                      const uint4 *__restrict__ run_table,
                                                      const int64_t *__restrict__ status, uint32_t max_extra,
                                                      uint32_t b, uint32_t &run, uint32_t &beg, int &n)
{
    const bool extra = b < max_extra;
    const uint4 rec = extra ? run_table[b] : tile_order[b - max_extra];
    const int64_t overflow = status[GA_STATUS_OVERFLOW], extra_runs = status[GA_STATUS_EXTRA_RUNS];
    if (overflow || (extra && (int64_t)b >= extra_runs)) return false;
    run = extra ? rec.y : 0u;
    beg = extra ? rec.z : rec.y;
    n = (int)(extra ? rec.w : rec.z);
    return true;
}

// Rank merge of the sorted runs of a long list (keys unique inside a tile => ranks are a permutation), as the LAST GRID REGION of the
// sort launch (round 3: a launch of its own, 10 us behind the sort).  A run is split over kMergeParts workgroups (2 keys per thread);
// the other runs are copied into LDS one at a time, then every thread binary-searches its keys in them, branch-free.  The runs are
// sorted by earlier workgroups of the same launch: each of them, having written its run, fences and counts itself in word 5 of the
// list's seg_sync group (zero at launch: the scan clears it); a merge workgroup waits for the count to reach the number of runs --
// everything it waits for was dispatched before it (workgroups are dispatched in index order), so the wait cannot deadlock.
constexpr int kMergeParts = 4;

__device__ __forceinline__ void merge_runs(uint64_t *__restrict__ other, const uint64_t *__restrict__ keys, uint32_t *__restrict__ point_list,
                                           const uint32_t *__restrict__ run_count, uint32_t run, uint32_t beg, int n, int part)
{
    const int nruns = (n + kSortCap - 1) / kSortCap;
    if (threadIdx.x == 0)
        while (__hip_atomic_load(run_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)nruns) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    // (no fence: an agent-scope fence on gfx950 writes back / invalidates the XCD's whole L2 -- measured here: the launch went from 25 to
    // 78 us.  The runs are WRITTEN with agent-scope stores (through the L2) that have completed before their workgroup counts itself,
    // and READ with agent-scope loads, which do not take a line another workgroup of this XCD may have cached before the run was sorted.)
    const int rb = (int)run * kSortCap, rn = min(kSortCap, n - rb);
    const uint64_t *k = keys + beg;
    constexpr int kPer = kSortCap / kMergeParts / 256;  // keys of my part of the run per thread
    uint64_t mine[kPer];
    int rank[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        const int e = part * (kSortCap / kMergeParts) + (int)threadIdx.x + 256 * i;
        mine[i] = e < rn ? __hip_atomic_load(k + rb + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
        rank[i] = e;
    }
    for (int j = 0; j < nruns - 1; ++j) {  // the other runs, skipping my own
        const int r = j < (int)run ? j : j + 1;
        const int ob = r * kSortCap, on = min(kSortCap, n - ob);
        if (j) __syncthreads();
        uint64_t tmp[kSortCap / 256];
#pragma unroll
        for (int i = 0; i < kSortCap / 256; ++i) {
            const int t = (int)threadIdx.x + 256 * i;
            tmp[i] = t < on ? __hip_atomic_load(k + ob + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
        }
#pragma unroll
        for (int i = 0; i < kSortCap / 256; ++i) other[threadIdx.x + 256 * i] = tmp[i];
        __syncthreads();
        int lo[kPer], hi[kPer];
#pragma unroll
        for (int i = 0; i < kPer; ++i) { lo[i] = 0; hi[i] = on; }
        for (int step = 0; step < 12; ++step) {  // 2^11 = kSortCap: 12 halvings reach lo == hi
#pragma unroll
            for (int i = 0; i < kPer; ++i) {
                const bool open = lo[i] < hi[i];
                const int mid = (lo[i] + hi[i]) >> 1;       // < kSortCap whenever the interval is open
                const bool less = other[mid & (kSortCap - 1)] < mine[i];
                lo[i] = (open && less) ? mid + 1 : lo[i];
                hi[i] = (open && !less) ? mid : hi[i];
            }
        }
#pragma unroll
        for (int i = 0; i < kPer; ++i) rank[i] += lo[i];
    }
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        const int e = part * (kSortCap / kMergeParts) + (int)threadIdx.x + 256 * i;
        if (e < rn) point_list[beg + rank[i]] = (uint32_t)mine[i];
    }
}

// Grid: [ max_extra workgroups: runs 1.. of the lists longer than one run (run table) | nbig workgroups: run 0 of schedule slot j if its
// list has kWaveSort entries or more (the schedule is longest first; at most capacity / kWaveSort lists are that long, and a slot whose
// list is shorter leaves at once) | ceil(slots / 4) workgroups: slots 4 g .. 4 g + 3, ONE WAVEFRONT PER LIST with the keys in registers,
// lists shorter than kWaveSort ].  Round 3 gave every list a 256-thread workgroup and the LDS network: a typical short list (64 .. 256
// keys) ran it with 16 .. 32 active lanes.  Measured (tools/sort_stamps.py): the launch is bound by the lists of 512 .. 2048 entries in
// their workgroups (three quarters of all keys, 17 .. 24 us of life each, four of them per CU at a time); one wave per list of ANY length
// (32 keys per lane for a 2048-key run: 64-bit cross-lane stages cost ~30 cycles per key, a single wave's chain was 30 .. 40 us) was
// built and measured: 53 us instead of 36 for sort + merge -- not kept.
__global__ __launch_bounds__(256) void surfel_run_sort_kernel(const uint4 *__restrict__ tile_order,
                                                              const uint4 *__restrict__ run_table,
                                                              uint32_t max_extra, uint32_t nbig, uint32_t nslots,
                                                              uint64_t *__restrict__ keys,
                                                              uint32_t *__restrict__ point_list,
                                                              const int64_t *__restrict__ status,
                                                              uint32_t *__restrict__ tile_count, uint32_t *__restrict__ tile_cursor,
                                                              unsigned long long *__restrict__ view_total, int nviews,
                                                              uint32_t *__restrict__ seg_sync, unsigned long long *__restrict__ dbg,
                                                              Dims dm, const float *__restrict__ bg, float *__restrict__ out_color,
                                                              float *__restrict__ out_others)
{
    __shared__ __attribute__((aligned(16))) double s[kSortCap];  // raw key bits (see bitonic_sort_blocked)
    const int tid = threadIdx.x;
#ifdef GA_SORT_STAMPS   // measurement build: (start, end, list length) per wave in the (dead by now) depth array, 100 MHz clock
    struct StampExit {
        unsigned long long *p, t0; int n;
        __device__ ~StampExit() { p[0] = t0; p[1] = __builtin_amdgcn_s_memrealtime(); p[2] = 0x5A5A000000000000ull | (unsigned)n; }
    } stamp{dbg + ((size_t)blockIdx.x * 4 + (tid >> 6)) * 4, __builtin_amdgcn_s_memrealtime(), -1};
#define GA_STAMP_N(x) stamp.n = (x)
#else
#define GA_STAMP_N(x)
#endif
    const uint32_t nsort = max_extra + nbig + (nslots + 3u) / 4u;
    if (blockIdx.x >= nsort) {
        // ---- last region: merge of the runs of the lists longer than one run (see merge_runs)
        const uint32_t b = (blockIdx.x - nsort) / kMergeParts;
        uint32_t run, beg;
        int n;
        const uint32_t pos = b < max_extra ? run_table[b].x : b - max_extra;
        if (!sort_block_assignment(tile_order, run_table, status, max_extra, b, run, beg, n)) return;
        if (n <= kSortCap) return;
        merge_runs(reinterpret_cast<uint64_t *>(s), keys, point_list, seg_sync + 8 * (size_t)pos + 5, run, beg, n,
                   (int)((blockIdx.x - nsort) % kMergeParts));
        return;
    }
    if (blockIdx.x >= max_extra + nbig) {
        // ---- one wavefront per list
        const int lane = tid & 63, wave = tid >> 6;
        const uint32_t g = blockIdx.x - max_extra - nbig, j = 4u * g + (uint32_t)wave;
        const uint4 rec = tile_order[j < nslots ? j : 0u];
        const int64_t overflow = status[GA_STATUS_OVERFLOW];
        if (j >= nslots) return;
        // Leave the accumulating words of the binning clean for the NEXT launch (GA_SURFEL_FLAG_WORKSPACE_CLEAN): every schedule slot's
        // wave clears its tile's counter and fill cursor -- the fill launch that read them is complete -- and the first workgroup the
        // view totals.  (The segment / status words are cleared by the scan workgroup; the schedule is written whatever the overflow state.)
        if (lane == 0) { tile_count[rec.x] = 0u; tile_cursor[rec.x] = 0u; }
        if (g == 0)
            for (int u = tid; u < nviews * kViewSlots; u += 256) view_total[u] = 0ull;
        const uint32_t beg = rec.y;
        const int n = (int)rec.z;
        GA_STAMP_N(n < kWaveSort ? n : -2);
        if (!overflow && n == 0 && out_color != nullptr) {
            // Round 6: an EMPTY tile has no list to sort -- its wave writes the tile's background pixels (colour = bg, the seven allmap
            // channels 0: what the blend's fresh pixel gives, bit for bit) here, where the launch is latency-bound and the memory
            // system idle, instead of a blend workgroup doing it at the end of the blend.  Lane -> row lane >> 2, pixels 4 (lane & 3) ..
            const int v = (int)(rec.x / (uint32_t)dm.tiles), tile = (int)(rec.x - (uint32_t)v * dm.tiles);
            const int px = (tile % dm.gx) * kTile + 4 * (lane & 3), py = (tile / dm.gx) * kTile + (lane >> 2);
            if (py < dm.H && px < dm.W) {
                const size_t HW = (size_t)dm.H * dm.W, pid = (size_t)py * dm.W + px;
                float *oc = out_color + (size_t)v * 3 * HW + pid, *oo = out_others + (size_t)v * 7 * HW + pid;
                const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
                if (px + 3 < dm.W && (dm.W & 3) == 0 && (HW & 3) == 0) {   // 16-byte stores: whole 64-byte tile rows per four lanes
                    *reinterpret_cast<float4 *>(oc) = make_float4(b0, b0, b0, b0);
                    *reinterpret_cast<float4 *>(oc + HW) = make_float4(b1, b1, b1, b1);
                    *reinterpret_cast<float4 *>(oc + 2 * HW) = make_float4(b2, b2, b2, b2);
#pragma unroll
                    for (int c = 0; c < 7; ++c) *reinterpret_cast<float4 *>(oo + (size_t)c * HW) = make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    for (int e = 0; e < 4 && px + e < dm.W; ++e) {
                        oc[e] = b0; oc[HW + e] = b1; oc[2 * HW + e] = b2;
                        for (int c = 0; c < 7; ++c) oo[(size_t)c * HW + e] = 0.f;
                    }
                }
            }
            return;
        }
        if (overflow || n <= 0 || n >= kWaveSort) return;
        if (n == 1) { if (lane == 0) point_list[beg] = (uint32_t)keys[beg]; return; }
        if (n <= 64) wave_sort_list<1>(keys, point_list, beg, n, lane);
        else if (n <= 128) wave_sort_list<2>(keys, point_list, beg, n, lane);
        else if (n <= 256) wave_sort_list<4>(keys, point_list, beg, n, lane);
        else wave_sort_list<8>(keys, point_list, beg, n, lane);
        return;
    }
    uint32_t run, beg;
    int n;
    const uint32_t pos = blockIdx.x < max_extra ? run_table[blockIdx.x].x : blockIdx.x - max_extra;   // schedule slot of my list
    if (!sort_block_assignment(tile_order, run_table, status, max_extra, blockIdx.x, run, beg, n)) return;
    if (n < kWaveSort) return;     // (a wavefront of the next grid region sorts it)
    GA_STAMP_N(n);
    const int rb = (int)run * kSortCap, rn = min(kSortCap, n - rb);
    int np = 2; while (np < rn) np <<= 1;
    for (int t = tid; t < np; t += 256) s[sort_slot(t)] = __longlong_as_double((long long)(t < rn ? keys[beg + rb + t] : kSortPad));
    __syncthreads();
    bitonic_sort_lds(s, np, tid);
    if (n <= kSortCap) {
        for (int t = tid; t < rn; t += 256) point_list[beg + t] = (uint32_t)__double_as_longlong(s[sort_slot(t)]);   // single run: final order
    } else {
        // sorted run: merged by the last grid region (other workgroups, possibly on another XCD: see merge_runs)
        for (int t = tid; t < rn; t += 256)
            __hip_atomic_store(keys + beg + rb + t, (uint64_t)__double_as_longlong(s[sort_slot(t)]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my stores have completed
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(seg_sync + 8 * (size_t)pos + 5, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

void launch_binning(const GaSurfelForwardArgs &a, const Dims &d, const Workspace &ws, hipStream_t s)
{
    const int nt = d.V * d.tiles;
    const ScanArgs sa{ws.tile_count, ws.seg_sync, (uint32_t)(8 * ((size_t)a.capacity / 1024 + 1)), ws.tile_start, ws.tile_cursor,
                      ws.tile_order, ws.run_table, nt, a.capacity, seg_items(a.capacity, a.seg_capacity), ws.seg_table, ws.status};
    const unsigned nbx = (unsigned)std::max(1, (d.N + kFT * kFillSplats - 1) / (kFT * kFillSplats));
    const int nsched = (int)std::min<unsigned>(nbx, (unsigned)((nt + kFT - 1) / kFT));   // one schedule slot per thread of a row-0 workgroup
    // (16-bit halves in the schedule's class bins; the schedule workgroups' notes of the long lists fit point_list)
    // 2 * kLdsTiles words of dynamic LDS (64 KiB at 8192 tiles) beside the kernels' static LDS must be opted into: the attribute belongs
    // to the function ON ONE DEVICE, so it is set once per device (0 = not yet, 1 = granted, 2 = refused; racing threads both set it:
    // idempotent), and a refusal keeps the launches that would need it on the paths that fit the default 64 KiB
    static std::atomic<uint8_t> lds_optin[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int di = dev >= 0 && dev < 64 ? dev : 63;
    uint8_t st = dev == di ? lds_optin[di].load(std::memory_order_acquire) : (uint8_t)0;
    if (st == 0) {
        const int want = 2 * kLdsTiles * (int)sizeof(uint32_t);
        const bool ok = hipFuncSetAttribute((const void *)surfel_fill_sched_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess &&
                        hipFuncSetAttribute((const void *)surfel_fill_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        st = ok ? 1 : 2;
        lds_optin[di].store(st, std::memory_order_release);
    }
    // without the opt-in a launch may use 64 KiB of LDS in all; the fused kernel's static part is below 12 KiB (histl: 8.25), the plain fill's below 1 KiB
    const size_t dyn = 2 * (size_t)d.tiles * sizeof(uint32_t);
    const bool fused_fits = st == 1 || dyn + 12288 <= 65536, fill_fits = st == 1 || dyn + 1024 <= 65536;
    if (fused_fits && d.tiles <= kLdsTiles && nt <= 0xFFFF && (int64_t)nsched * 2 * (a.capacity / kSortCap + 1) <= a.capacity) {
        hipLaunchKernelGGL(surfel_fill_sched_kernel, dim3(nbx, (unsigned)d.V + 1u), dim3(kFT), dyn, s, sa,
                           ws.rect, ws.depth, d, ws.view_total, ws.keys, nsched, ws.point_list, ws.seg_scratch);
        return;
    }
    // larger problems: the single-workgroup scan in front of the fill
    hipLaunchKernelGGL(surfel_tile_scan_kernel, dim3(1), dim3(1024), 0, s, sa);
    const dim3 grid((unsigned)((d.N + 256 * kBinSplats - 1) / (256 * kBinSplats)), (unsigned)d.V);
    if (d.tiles <= kLdsTiles && fill_fits)
        hipLaunchKernelGGL(surfel_fill_kernel<true>, grid, dim3(256), dyn, s, ws.rect, ws.depth, d, ws.tile_cursor,
                           ws.keys, ws.status);
    else
        hipLaunchKernelGGL(surfel_fill_kernel<false>, grid, dim3(256), 0, s, ws.rect, ws.depth, d, ws.tile_cursor, ws.keys, ws.status);
}

void launch_tile_sort(const GaSurfelForwardArgs &a, const Dims &d, const Workspace &ws, hipStream_t s)
{
    const uint32_t nt = (uint32_t)(d.V * d.tiles);
    const uint32_t max_extra = (uint32_t)(a.capacity / kSortCap + 1);
    const uint32_t nbig = (uint32_t)std::min<int64_t>(nt, a.capacity / kWaveSort + 1);   // lists of kWaveSort entries or more
    // lists longer than one run sit at the front of tile_order and there are fewer than capacity / kSortCap of them
    const uint32_t max_big = (uint32_t)std::min<int64_t>(nt, a.capacity / kSortCap);
    hipLaunchKernelGGL(surfel_run_sort_kernel, dim3(max_extra + nbig + (nt + 3) / 4 + (max_big + max_extra) * kMergeParts), dim3(256), 0, s,
                       ws.tile_order, ws.run_table, max_extra, nbig, nt, ws.keys, ws.point_list, ws.status, ws.tile_count, ws.tile_cursor,
                       ws.view_total, d.V, ws.seg_sync, reinterpret_cast<unsigned long long *>(ws.depth), d, a.bg,
                       (a.flags & GA_SURFEL_FLAG_BG_IN_BLEND) ? nullptr : a.out_color, a.out_others);
}

}  // namespace ga
