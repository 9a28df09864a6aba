// surfel_blend.hip -- per-tile front-to-back alpha compositing of colour + depth + normal (+ median depth,
// distortion), gfx950.  Replaces upstream renderCUDA of diff_surfel_rasterization (call site
// /root/reference/nsr/gs_surfel.py:100-114; consumer of the 7 allmap channels :121-142); arithmetic per
// SURVEY.md Appendix A.1 "Blend".
//
// MI355X-first formulation, not the CUDA block-cooperative one.  The kernel is bound by VALU issue and LDS bandwidth
// (profiles/r1f_pmc.txt: 75 M wave instructions per launch, LDS 83 % busy, HBM far from its roof), so everything here is
// about instructions and LDS bytes per useful (pixel, splat) pair:
//   * WORK ITEM = one 16x16 tile, or one SEGMENT (256..512 list entries) of a tile whose list has >= 2048 entries; the
//     16x16 granularity is part of the semantics (the tile rect decides which pixels a splat may touch).
//   * WORKGROUP = four wavefronts, one per 8x8 pixel quadrant; all four share ONE LDS image of the item's list.
//   * STAGING (lanes = entries, 64 list entries = one chunk): gather the 96-byte records, test every entry's conservative
//     {alpha >= 1/255} pixel box against the 16 pixel columns and 16 pixel rows of the tile (32 ballots), rebase the plane
//     coefficients to the tile origin, leave records and masks in LDS.  Each record is fetched and prepared ONCE per tile
//     (round 1 did it once per quadrant: 2.3x the algorithmic HBM/L2 traffic and four times the staging instructions) and
//     once for BOTH passes of a segment.  The duty rotates: chunk k is staged by wave k % 4; the first eight chunks (two
//     per wave) before the walk starts -- for a segment and for most tiles that is everything -- and in a longer
//     unsegmented list (513..2047 entries) chunk i + 4 when its wave reaches chunk i, with the loads requested one duty
//     earlier.  (A dedicated fifth producer wave was built and measured first: 5-wave workgroups only fit twice per CU.)
//   * Consumers run with LANES = PIXELS and per-lane survivor lists: a lane ANDs the mask of its column with the mask of
//     its row and walks only those entries, kU per trip (the LDS gathers and alpha evaluations of a trip are
//     independent, the composites follow in order).  A two-chunk window lets a lane that has finished the older chunk
//     run ahead into the newer one.  The only synchronisation is one LDS word per chunk (DS operations of a wave execute
//     in order, so stamp-after-data is enough) plus, for wrapping lists, one progress word per wave: no workgroup barrier
//     in the list walk; a saturated quadrant leaves once its staging duties are done.
//   * SEGMENTS make the serial chain of a long list short (round 1: one workgroup, the hottest quadrant of the 5 127-entry
//     list, spanned the whole launch).  Pass 1: every segment multiplies (1 - alpha) over its pairs and publishes the
//     per-pixel product; pass 2: a segment enters the UNCHANGED sequential blend with the product of its predecessors, so
//     weights, the `T > 0.5` median test and the `T (1 - alpha) < 1e-4` stop rule see the global transmittance; the last
//     segment to finish a quadrant adds the partial sums in order, with the cross terms of the depth distortion
//     (M2_before * W_k - 2 M1_before * M1_k).  Segments of a tile are different workgroups (4, 8, 16 ... by list length);
//     they find each other through a scratch area and per-(segment, quadrant) flags in the workspace.  A segment only
//     ever waits for LOWER-numbered segments and segment work items are handed out by an atomic ticket, so whatever it
//     waits for has already started: no deadlock, whatever the dispatch order.  The only numerical difference to the
//     sequential order is the rounding of the prefix product.
//   * the ray/splat intersection uses the plane form p = C' + dx*A + dy*B (6 FMAs) instead of two 3-vector affine maps
//     and a cross product (18 ops); upstream's chain of `continue` filters is one branch-free predicate.
// Pixel results are compared with the oracle by MSE (<= 1e-5, tests/), so this TU may contract to FMA and uses
// v_rcp_f32 / v_exp_f32 instead of IEEE division and libm expf.
#include <hip/hip_fp16.h>

#include <atomic>

#include "surfel_common.h"

namespace ga {

typedef float f2 __attribute__((ext_vector_type(2)));  // one packed-fp32 operand (even-aligned register pair)

__device__ __forceinline__ f2 lo2(const float4 &q) { return f2{q.x, q.y}; }
__device__ __forceinline__ f2 hi2(const float4 &q) { return f2{q.z, q.w}; }

constexpr int kSlots = kItemChunks * 64;   // list entries of a work item that fit the LDS image
constexpr uint32_t kGone = 0xFFFFFFFFu;    // done[q]: this quadrant needs nothing any more

struct PixelAcc {  // pairs are updated by one v_pk_fma_f32
    float T, Dp, dist, median;
    f2 M;     // M1, M2
    f2 N01;   // normal x, y
    f2 N2C0;  // normal z, red
    f2 C12;   // green, blue
};

__device__ __forceinline__ PixelAcc fresh_pixel(float T)
{
    return PixelAcc{T, 0.0f, 0.0f, 0.0f, f2{0.0f, 0.0f}, f2{0.0f, 0.0f}, f2{0.0f, 0.0f}, f2{0.0f, 0.0f}};
}

struct Rec {  // one staged record (tile-relative, see stage_chunk) in registers; layout: surfel_common.h
    float4 q0, q1, q2, q3, q4;
    f2 q5;
};

// LDS image of a work item: six planes of kSlots entries (plane q holds quad q of every entry), so that the stager's writes
// are contiguous and a consumer's gather of 16 different entries spreads over all 64 banks; per chunk the 32 survivor masks
// (entries covering pixel column c / pixel row c - 16 of the tile) and a stamp.
struct __attribute__((aligned(16))) Ring {
    float4 planes[5][kSlots];
    f2 plane5[kSlots];
    unsigned long long masks[kItemChunks][32];
    uint32_t stamp[kItemChunks];   // k + 1 once chunk k of the item is in slot k % kItemChunks
    uint32_t done[4];              // unsegmented lists of more than kItemChunks chunks only: chunks finished by consumer q
    uint32_t work;                 // work item of this workgroup (ticket broadcast)
    uint32_t sat;                  // segmented tile: nsegs - k of the earliest segment k known to end every pixel's walk (0: none)
    uint32_t sat_waves[2];         // quadrants of this segment whose pixels are all finished after pass 1 / pass 2
};

__device__ __forceinline__ uint32_t lds_load(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store(uint32_t *p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// Cross-workgroup exchange of the segmented blend.  The L2 of an XCD is not coherent with the other seven, and an
// agent-scope release / acquire FENCE is a write-back / invalidate of that whole L2 (measured: 0.7 ms per launch for
// 1 340 segments), while plain agent-scope stores to different addresses are not ordered with respect to each other on
// their way to memory.  So every exchanged value is SELF-VALIDATING: one 64-bit agent-scope store (sc1: bypasses the L2)
// of (value, epoch of this launch), which the reader polls until the epoch matches -- no fence, no flag, no ordering
// assumption.  The epoch is a per-launch number the tile scan bumps in the workspace (seg_table[kSegEpochWord]: also under HIP-graph
// replay, where a kernel argument would be frozen), so words left by earlier launches never match.
__device__ __forceinline__ void xwg_store(unsigned long long *p, float v, uint32_t epoch)
{
    __hip_atomic_store(p, ((unsigned long long)epoch << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float xwg_load(const unsigned long long *p, uint32_t epoch)
{
    unsigned long long w;
    for (;;) {
        w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_ballot_w64((uint32_t)(w >> 32) != epoch) == 0) break;
        __builtin_amdgcn_s_sleep(4);
    }
    return __uint_as_float((uint32_t)w);
}

// compiler fence: LDS operations of one wave reach the LDS in program order, so ordering data and flag accesses in the
// instruction stream is all the protocol needs
#define GA_LDS_ORDER() asm volatile("" ::: "memory")

// One (pixel, splat) evaluation -- SURVEY.md A.1 "Blend" -- in two parts.  dxy: this lane's pixel relative to the
// tile origin.  eval_alpha is free of cross-entry dependences (several entries are evaluated back to back);
// composite is the short sequential part.  Upstream's chain of `continue` filters is evaluated branch-free into one
// predicate (the filters commute: each one only decides whether the pair is skipped).
struct Alpha {
    float alpha;
    f2 s;
    bool pass, use3d;
};

__device__ __forceinline__ Alpha eval_alpha(const Rec &r, f2 dxy)
{
    // p = C' + dx*A + dy*B
    const f2 pxy = dxy.y * hi2(r.q0) + (dxy.x * lo2(r.q0) + lo2(r.q1));
    const float p2 = fmaf(dxy.y, r.q1.w, fmaf(dxy.x, r.q1.z, r.q2.z));
    const float rz = __builtin_amdgcn_rcpf(p2);
    Alpha o;
    o.s = pxy * rz;
    const f2 ss = o.s * o.s;
    const float rho3d = ss.x + ss.y;
    const f2 e = lo2(r.q2) - dxy;  // centre - pixel
    const f2 ee = e * e;
    const float rho2d = kFilterInvSquare * (ee.x + ee.y);
    const float rho = fminf(rho3d, rho2d);
    o.use3d = rho3d <= rho2d;
    o.alpha = fminf(0.99f, r.q2.w * __builtin_amdgcn_exp2f(rho * -0.72134752044f));
    // upstream: p.z == 0 -> skip ; power = -0.5*rho > 0 -> skip (a NaN rho passes) ; alpha < 1/255 -> skip
    o.pass = p2 != 0.0f && !(rho < 0.0f) && !(o.alpha < 1.0f / 255.0f);
    return o;
}

__device__ __forceinline__ float pair_depth(const Rec &r, const Alpha &e)
{
    const f2 d = e.s * lo2(r.q3);
    return e.use3d ? (d.x + d.y) + r.q3.z : r.q3.z;
}

// `go`: this lane really has this pair (a live entry of its list that passed the filters, pixel not finished).  The
// update is branch-free: a pair that does not count enters with weight 0 and a harmless depth, so the wave executes one
// straight instruction stream instead of an EXEC-masked region per entry (of the 312 VALU + 105 SALU instructions of a
// four-entry trip, 41 v_mov and ~60 SALU were that masking and its register copies).  Measured: same time -- the trip
// loop is as much LDS-bound (88 bytes gathered per entry, 12 waves on one LDS pipe) as VALU-bound.
__device__ __forceinline__ void composite(const Rec &r, const Alpha &e, PixelAcc &a, bool &done, bool go)
{
    const float kM = kFar / (kFar - kNear);
    const float depth_raw = pair_depth(r, e);
    const bool near_ok = go && !(depth_raw < kNear);        // upstream: depth < near -> skip (before the alpha test)
    const float test_T = a.T * (1.0f - e.alpha);
    const bool stop = near_ok && test_T < 0.0001f;          // upstream: done = true
    const bool use = near_ok && !stop;
    done = done || stop;
    const float depth = use ? depth_raw : 1.0f;
    const float w = use ? e.alpha * a.T : 0.0f;
    const float A = 1.0f - a.T;
    const float m = kM * (1.0f - kNear * __builtin_amdgcn_rcpf(depth));
    const f2 mm = f2{m, m * m};
    a.dist += (mm.y * A + a.M.y - 2.0f * m * a.M.x) * w;
    a.Dp += depth * w;
    a.M += mm * w;
    a.median = (use && a.T > 0.5f) ? depth : a.median;
    a.N01 += lo2(r.q4) * w;
    a.N2C0 += hi2(r.q4) * w;
    a.C12 += r.q5 * w;
    a.T = use ? test_T : a.T;
}

struct Stats {
    unsigned iters, chunks, useful, lanemax;
};

// ---------------------------------------------------------------------------------------------------------------
// STAGING.  One chunk: the records of 64 consecutive list entries (lanes = entries) are already in registers.
struct ChunkRegs {
    float4 g0, g1, g2, g3, g4;
    f2 g5;
};

__device__ __forceinline__ ChunkRegs load_chunk(const float4 *__restrict__ rec4, uint32_t id)
{
    const float4 *r = rec4 + (size_t)id * (kRec / 4);
    return ChunkRegs{r[0], r[1], r[2], r[3], r[4], *reinterpret_cast<const f2 *>(r + 5)};
}

// masks, rebase, LDS image of chunk k (slot k % kItemChunks), stamp.  `valid`: my entry lies inside the list range.
__device__ __forceinline__ void stage_chunk(Ring &ring, int lane, uint32_t k, const ChunkRegs &g, bool valid, float tx0, float ty0)
{
    // centre relative to the tile origin and the cull half-extents (fp16 pair, +inf = unbounded, < 0 = never)
    const float ex0 = g.g2.x - tx0, ey0 = g.g2.y - ty0;
    const uint32_t cull = __float_as_uint(g.g3.w);
    const float rx = valid ? __half2float(__ushort_as_half((unsigned short)(cull & 0xffffu))) : -1.0f;
    const float ry = __half2float(__ushort_as_half((unsigned short)(cull >> 16)));
    // the 32 masks leave through lanes 0..31: lane c takes column mask c, lane 16 + c row mask c
    unsigned long long m = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const unsigned long long xm = __builtin_amdgcn_ballot_w64(fabsf((float)c - ex0) <= rx);
        const unsigned long long ym = __builtin_amdgcn_ballot_w64(fabsf((float)c - ey0) <= ry);
        m = lane == c ? xm : m;
        m = lane == 16 + c ? ym : m;
    }
    // rebase to the tile origin: C' = C + (t0.x - ox)*A + (t0.y - oy)*B with o = rint(centre); centre -= t0
    const float ux = tx0 - rintf(g.g2.x), uy = ty0 - rintf(g.g2.y);
    const int slot = (int)(k % kItemChunks), j = slot * 64 + lane;
    ring.planes[0][j] = g.g0;
    ring.planes[1][j] = make_float4(fmaf(uy, g.g0.z, fmaf(ux, g.g0.x, g.g1.x)), fmaf(uy, g.g0.w, fmaf(ux, g.g0.y, g.g1.y)), g.g1.z, g.g1.w);
    ring.planes[2][j] = make_float4(ex0, ey0, fmaf(uy, g.g1.w, fmaf(ux, g.g1.z, g.g2.z)), g.g2.w);
    ring.planes[3][j] = g.g3;
    ring.planes[4][j] = g.g4;
    ring.plane5[j] = g.g5;
    if (lane < 32) ring.masks[slot][lane] = m;
    GA_LDS_ORDER();
    if (lane == 0) lds_store(&ring.stamp[slot], k + 1);
    GA_LDS_ORDER();
}

// Staging duty: chunk k of an item is staged by wave k % 4.  The first kItemChunks chunks (two per wave) are staged before
// the walk starts -- for a segment and for most tiles that is everything.  In a longer unsegmented list wave w stages chunk
// i + 4 when it reaches its own chunk i (i % 4 == w), into the slot of chunk i - 4, once every wave has finished that one;
// its records were requested one duty earlier (ids two duties earlier).
struct Duty {
    uint32_t next, sbeg, send;   // my next chunk to stage; the list range
    uint32_t id1, id2;           // list entries of chunk `next` (arrived) and `next + 4` (in flight)
    ChunkRegs g;                 // records of chunk `next` (in flight)
    const uint32_t *__restrict__ point_list;
    const float4 *__restrict__ rec4;
    float tx0, ty0;
};

__device__ __forceinline__ uint32_t duty_entry(const Duty &d, uint32_t k, int lane)
{
    const uint32_t e = d.sbeg + k * 64 + lane;   // (clamped: out-of-range lanes re-read a valid entry)
    return e < d.send ? e : d.sbeg;
}

// stage chunks `quad` and `quad + 4`, then (longer lists) start the pipeline for `quad + 8`
__device__ __forceinline__ void duty_prologue(Ring &ring, Duty &d, int lane, int quad, uint32_t nch)
{
    const uint32_t k0 = (uint32_t)quad, k1 = (uint32_t)quad + 4;
    if (k0 >= nch) { d.next = nch; return; }
    const uint32_t ida = d.point_list[duty_entry(d, k0, lane)];
    const uint32_t idb = d.point_list[duty_entry(d, k1 < nch ? k1 : k0, lane)];
    const ChunkRegs ga = load_chunk(d.rec4, ida);
    ChunkRegs gb = ga;
    if (k1 < nch) gb = load_chunk(d.rec4, idb);
    // my second chunk is staged here only if its slot is not one of the first round's (a ring of fewer than eight chunks): otherwise it
    // becomes my first duty of the walk, its records already on their way
    const bool second = k1 < nch && k1 < (uint32_t)kItemChunks;
    d.next = second ? k1 + 4 : k1;
    if (d.next < nch) {   // wrapping list: ids of my next two duties
        d.id1 = second ? d.point_list[duty_entry(d, d.next, lane)] : idb;
        d.id2 = d.point_list[duty_entry(d, d.next + 4 < nch ? d.next + 4 : d.next, lane)];
    }
    stage_chunk(ring, lane, k0, ga, d.sbeg + k0 * 64 + lane < d.send, d.tx0, d.ty0);
    if (second) stage_chunk(ring, lane, k1, gb, d.sbeg + k1 * 64 + lane < d.send, d.tx0, d.ty0);
    if (d.next < nch) d.g = second ? load_chunk(d.rec4, d.id1) : gb;
}

// stage chunk d.next (its slot held chunk d.next - kItemChunks: every wave must have finished that one) and move the
// pipeline on.  Returns false when nobody is left to read it.
__device__ __forceinline__ bool duty_stage(Ring &ring, Duty &d, int lane, uint32_t nch)
{
    const uint32_t k = d.next;
    for (;;) {
        const uint32_t d0 = lds_load(&ring.done[0]), d1 = lds_load(&ring.done[1]);
        const uint32_t d2 = lds_load(&ring.done[2]), d3 = lds_load(&ring.done[3]);
        const uint32_t dmin = min(min(d0, d1), min(d2, d3));
        if (dmin == kGone) return false;
        if (dmin + kItemChunks > k) break;
        __builtin_amdgcn_s_sleep(2);
    }
    GA_LDS_ORDER();
    stage_chunk(ring, lane, k, d.g, d.sbeg + k * 64 + lane < d.send, d.tx0, d.ty0);
    d.next = k + 4;
    if (d.next < nch) {
        d.id1 = d.id2;
        d.g = load_chunk(d.rec4, d.id1);
        d.id2 = d.point_list[duty_entry(d, d.next + 4 < nch ? d.next + 4 : d.next, lane)];
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// CONSUMER: walk the `nch` chunks of the item for one 8x8 pixel quadrant.  FULL: the complete per-pixel blend;
// !FULL: only the transmittance product of the range (every contributing pair multiplies T by 1 - alpha; no stop rule
// except "the product alone is below the stop threshold").
struct Consumer {
    int lane, quad;
    f2 dxy;          // my pixel relative to the tile origin
    int col, row;    // my pixel column / row inside the tile
};

#ifndef GA_BLEND_KU
#define GA_BLEND_KU 4
#endif

// STORE (the forward of a differentiable call, GaSurfelForwardArgs.seg_T): at every second chunk boundary -- the 128-entry
// segments of ga_surfel_backward -- the pixel's transmittance (-1: its walk has ended) goes to tseg[(gc0 + i) / 2 * 256], gc0 = the
// item's first chunk counted from the list's begin.  Lanes run ahead into the new chunk as always; a lane notes its transmittance
// when it takes its first entry from there (three instructions per entry slot in those steps).
template <bool FULL, bool WRAP, bool STORE = false>
__device__ __forceinline__ void consume(Ring &ring, const Consumer &c, uint32_t nch, PixelAcc &a, bool &done, Stats &st, int flags,
                                        Duty &duty, float *__restrict__ tseg = nullptr, uint32_t gc0 = 0)
{
    // Per-lane survivor masks of a TWO-chunk window: `cur` = what is left of the previous chunk, `nxt` = the chunk fetched
    // in this step.  A lane that has finished `cur` runs ahead into `nxt` while slower lanes still work on `cur`; the step
    // ends when no lane has anything left in `cur`.
    unsigned long long cur = 0, nxt = 0;
    constexpr int kU = GA_BLEND_KU;
    float Tsnap = 0.0f;     // STORE: my transmittance on entering the chunk in `nxt` (-1: my walk ended before)
    bool snapped = false;
    auto trips = [&](int oldb, int newb, bool bstep = false) {
        // Lanes walk their own lists independently (compositing order only matters per pixel), kU entries per trip: the
        // kU gathers and alpha evaluations are mutually independent, then the contributing ones are composited in
        // order.  An exhausted lane re-reads its last slot with the live flag off.
        while (__builtin_amdgcn_ballot_w64(cur != 0) != 0) {
            int j[kU];
            bool live[kU], ahead[kU];
            int last = newb;   // (a staged slot: an exhausted lane reads finite values)
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const bool has = cur != 0;
                ahead[u] = !has;
                const unsigned long long sel = has ? cur : nxt;
                live[u] = sel != 0;
                j[u] = live[u] ? __builtin_ctzll(sel) + (has ? oldb : newb) : last;
                last = j[u];
                const unsigned long long rest = sel & (sel - 1);
                cur = has ? rest : 0ull;
                nxt = has ? nxt : rest;
            }
            st.iters += kU;
            Rec r[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                r[u].q0 = ring.planes[0][j[u]]; r[u].q1 = ring.planes[1][j[u]];
                r[u].q2 = ring.planes[2][j[u]]; r[u].q3 = ring.planes[3][j[u]];
                if (FULL) { r[u].q4 = ring.planes[4][j[u]]; r[u].q5 = ring.plane5[j[u]]; }
            }
            Alpha e[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) e[u] = eval_alpha(r[u], c.dxy);
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                if (STORE && bstep) {   // the first entry a lane takes from the new chunk: it crosses the segment boundary here
                    const bool cross = ahead[u] && live[u] && !snapped;
                    Tsnap = cross ? (done ? -1.0f : a.T) : Tsnap;
                    snapped = snapped || cross;
                }
                if (FULL) {
                    composite(r[u], e[u], a, done, live[u] && e[u].pass && !done);
                } else {
                    const float depth = pair_depth(r[u], e[u]);
                    if (live[u] && e[u].pass && !done && !(depth < kNear)) {
                        // T_global <= T_segment: once the segment product alone trips the stop rule the sequential
                        // loop has stopped at or before this pair, and every later segment is dead (P = 0)
                        const float t = a.T * (1.0f - e[u].alpha);
                        done = t < 0.0001f;
                        a.T = done ? 0.0f : t;
                    }
                }
            }
            if (done) { cur = 0; nxt = 0; }
        }
    };

    uint32_t i = 0;
    unsigned mine = 0;   // (statistics) entries of my own list
    for (; i < nch; ++i) {
        if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
        if (WRAP && duty.next == i + 4 && duty.next < nch) duty_stage(ring, duty, c.lane, nch);
        const int slot = (int)(i % kItemChunks);
        while (lds_load(&ring.stamp[slot]) != i + 1) __builtin_amdgcn_s_sleep(1);
        GA_LDS_ORDER();
        const unsigned long long mx = ring.masks[slot][c.col], my = ring.masks[slot][16 + c.row];
        const bool bstep = STORE && ((gc0 + i) & 1u) == 0u;   // chunk i begins a 128-entry segment of the backward
        snapped = false;
        nxt = done ? 0ull : (mx & my);
        if (flags & GA_SURFEL_FLAG_STATS) {
            unsigned pc = __builtin_popcountll(nxt);
            mine += pc;
            for (int o = 32; o > 0; o >>= 1) pc += __shfl_xor(pc, o, 64);
            st.useful += pc;
        }
        ++st.chunks;
        trips((int)((i + kItemChunks - 1) % kItemChunks) * 64, slot * 64, bstep);   // until every lane has finished the previous chunk
        if (bstep)   // (a lane that has not reached the new chunk yet enters it with what it has now)
            tseg[(size_t)((gc0 + i) >> 1) * 256] = snapped ? Tsnap : (done ? -1.0f : a.T);
        cur = nxt;           // what is left of this chunk becomes the "previous chunk" of the next step
        nxt = 0;
        if (WRAP) {          // chunks < i are finished: their slots may be reused
            GA_LDS_ORDER();
            if (c.lane == 0) lds_store(&ring.done[c.quad], i);
        }
    }
    trips((int)((i + kItemChunks - 1) % kItemChunks) * 64, 0);  // drain: `cur` is the last fetched chunk, `nxt` is empty
    if (STORE)   // (left the loop early: every pixel's walk has ended)
        for (uint32_t j = i; j < nch; ++j)
            if (((gc0 + j) & 1u) == 0u) tseg[(size_t)((gc0 + j) >> 1) * 256] = -1.0f;
    if (flags & GA_SURFEL_FLAG_STATS) {
        for (int o = 32; o > 0; o >>= 1) mine = max(mine, (unsigned)__shfl_xor(mine, o, 64));
        st.lanemax += mine;
    }
    if (WRAP) {
        GA_LDS_ORDER();
        if (c.lane == 0) lds_store(&ring.done[c.quad], kGone);
        GA_LDS_ORDER();
        while (duty.next < nch && duty_stage(ring, duty, c.lane, nch)) {}   // the others may still need my chunks
    }
}

__device__ __forceinline__ void write_pixel(const PixelAcc &a, const float *__restrict__ bg, const Dims &dm, int v, int pxi,
                                            int pyi, float *__restrict__ out_color, float *__restrict__ out_others)
{
    const size_t HW = (size_t)dm.H * dm.W, pid = (size_t)pyi * dm.W + pxi;
    float *oc = out_color + (size_t)v * 3 * HW + pid;
    float *oo = out_others + (size_t)v * 7 * HW + pid;
    oc[0] = a.N2C0.y + a.T * bg[0];
    oc[HW] = a.C12.x + a.T * bg[1];
    oc[2 * HW] = a.C12.y + a.T * bg[2];
    oo[0] = a.Dp;
    oo[HW] = 1.0f - a.T;
    oo[2 * HW] = a.N01.x;
    oo[3 * HW] = a.N01.y;
    oo[4 * HW] = a.N2C0.x;
    oo[5 * HW] = a.median;
    oo[6 * HW] = a.dist;
}

// ---------------------------------------------------------------------------------------------------------------
// One work item: tile_order slot `pos`, segment `seg` of `nsegs` (work item number `work` when segmented).
struct BlendArgs {
    const uint4 *__restrict__ tile_order;
    const uint32_t *__restrict__ point_list;
    const float *__restrict__ record;
    const float *__restrict__ bg;
    uint32_t *__restrict__ seg_sync;
    unsigned long long *__restrict__ seg_scratch;
    float *__restrict__ out_color;
    float *__restrict__ out_others;
    uint32_t epoch;
    int flags;
    float *__restrict__ seg_T;   // GaSurfelForwardArgs.seg_T (the STORE instantiation only)
    int64_t *__restrict__ prof;  // (GA_SPLIT_PROFILE builds) one row of section cycles per wave
};

// A segment's quadrant has published what it has to: count it; the last one of the tile's segments to arrive adds the partial
// sums in list order (with the cross terms of the depth distortion) and writes the pixels.  The loop ends at the first segment
// after which every pixel of the quadrant is finished: later segments may have left without partial sums (see `sat`).
__device__ __forceinline__ void finish_segment(const BlendArgs &k, const Dims &dm, int lane, uint32_t pos, uint32_t work0,
                                               uint32_t wstride, uint32_t nsegs, int px, int v, int pxi, int pyi, bool inside, uint32_t *arrive)
{
    uint32_t arrived = 0;
    if (lane == 0) arrived = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    if (arrived != nsegs - 1) return;
    PixelAcc r = fresh_pixel(1.0f);
    bool dead = !inside;
    for (uint32_t s2 = 0; s2 < nsegs; ++s2) {
        if (__builtin_amdgcn_ballot_w64(!dead) == 0) break;
        const unsigned long long *q = k.seg_scratch + (size_t)(work0 + s2 * wstride) * kSegFloats + 256 + px;
        float tt[14];
#pragma unroll
        for (int f = 0; f < 14; ++f) tt[f] = xwg_load(q + f * 256, k.epoch);
        if (!dead) {
            const float Wk = tt[13] - tt[11];
            r.N2C0.y += tt[0]; r.C12.x += tt[1]; r.C12.y += tt[2];
            r.N01.x += tt[3]; r.N01.y += tt[4]; r.N2C0.x += tt[5];
            r.Dp += tt[6];
            r.dist += tt[9] + r.M.y * Wk - 2.0f * r.M.x * tt[7];
            r.M.x += tt[7];
            r.M.y += tt[8];
            if (tt[10] >= 0.0f) r.median = tt[10];
            r.T = tt[11];
            dead = tt[12] != 0.0f;
        }
    }
    if (inside) write_pixel(r, k.bg, dm, v, pxi, pyi, k.out_color, k.out_others);
}

template <bool STORE>
__device__ __forceinline__ void blend_item(Ring &ring, const BlendArgs &k, const Dims &dm, int lane, int wave, uint32_t pos,
                                           const uint4 sched, uint32_t seg, uint32_t nsegs, uint32_t work, uint32_t wstride,
                                           Stats &st)
{
    const uint32_t *__restrict__ point_list = k.point_list;
    const float *__restrict__ record = k.record;
    const float *__restrict__ bg = k.bg;
    uint32_t *__restrict__ seg_sync = k.seg_sync;
    unsigned long long *__restrict__ seg_scratch = k.seg_scratch;
    float *__restrict__ out_color = k.out_color;
    float *__restrict__ out_others = k.out_others;
    const uint32_t epoch = k.epoch;
    const int flags = k.flags;
    const uint32_t vt = sched.x;          // schedule entry, longest lists first: (tile, list begin, list length)
    const int v = (int)(vt / (uint32_t)dm.tiles), tile = (int)(vt - (uint32_t)v * dm.tiles);
    const int tx = tile % dm.gx, ty = tile / dm.gx;
    const uint32_t beg = sched.y, n = sched.z;
    // my segment: chunks [seg * chunks / nsegs, (seg + 1) * chunks / nsegs) (at most kItemChunks of them when segmented)
    const uint32_t chunks = (n + 63) / 64;
    const uint32_t c0 = (uint32_t)((uint64_t)seg * chunks / nsegs), c1 = (uint32_t)((uint64_t)(seg + 1) * chunks / nsegs);
    const uint32_t sbeg = beg + c0 * 64, send = min(beg + n, beg + c1 * 64), nch = c1 - c0;
    const bool last_seg = seg + 1 == nsegs;
    const float4 *rec4 = reinterpret_cast<const float4 *>(record) + (size_t)v * dm.N * (kRec / 4);
    const float tx0 = (float)(tx * kTile), ty0 = (float)(ty * kTile);
    // my pixel's column of the backward's transmittance table (rows: the list's 128-entry segments, include/ga_surfel.h)
    float *tseg = STORE ? k.seg_T + ((size_t)(beg / 128u + vt) * 256 + (size_t)(wave * 64 + lane)) : nullptr;

    // SATURATED TILE.  Once every pixel of the tile has ended its walk inside segment k (the stop rule fired, or the
    // segment's own transmittance product is already below it), the segments after k contribute nothing: a segment that
    // finds this out before it starts (ring.sat, read once per workgroup) publishes a zero transmittance for the segments
    // that may still multiply it in, counts itself as arrived and leaves -- no staging, no walk.  On an opaque object most
    // of a long list lies behind the surface (1 M surfels x 8 views x 512^2, lists of up to 52 060 entries: blend 3.38 -> 2.38 ms).
    if (nsegs > 1 && ring.sat != 0 && seg > nsegs - ring.sat) {
        if (tx * kTile + (wave & 1) * 8 >= dm.W || ty * kTile + (wave >> 1) * 8 >= dm.H) return;   // quadrant outside the image
        const int col = (wave & 1) * 8 + (lane & 7), row = (wave >> 1) * 8 + (lane >> 3);
        const int pxi = tx * kTile + col, pyi = ty * kTile + row;
        const int px = wave * 64 + lane;
        if (STORE)
            for (uint32_t gc = c0; gc < c1; ++gc)
                if ((gc & 1u) == 0u) tseg[(size_t)(gc >> 1) * 256] = -1.0f;
        if (!last_seg) xwg_store(seg_scratch + (size_t)work * kSegFloats + px, 0.0f, epoch);
        finish_segment(k, dm, lane, pos, work - seg * wstride, wstride, nsegs, px, v, pxi, pyi, pxi < dm.W && pyi < dm.H,
                       seg_sync + 8 * (size_t)pos + wave);
        return;
    }

    // my share of the staging (the others wait for it, so even a quadrant outside the image does it)
    Duty duty;
    duty.sbeg = sbeg; duty.send = send; duty.point_list = point_list; duty.rec4 = rec4; duty.tx0 = tx0; duty.ty0 = ty0;
    duty_prologue(ring, duty, lane, wave, nch);
    const bool wraps = nch > (uint32_t)kItemChunks;   // only unsegmented lists

#ifdef GA_BLEND_STAMPS
    if (nch) {
        while (lds_load(&ring.stamp[0]) != 1) __builtin_amdgcn_s_sleep(1);
        if (lane == 0) k.seg_scratch[(size_t)GA_BLEND_STAMPS + ((size_t)blockIdx.x * 4 + wave) * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    Consumer c;
    c.lane = lane; c.quad = wave;
    c.col = (wave & 1) * 8 + (lane & 7); c.row = (wave >> 1) * 8 + (lane >> 3);
    c.dxy = f2{(float)c.col, (float)c.row};
    const int pxi = tx * kTile + c.col, pyi = ty * kTile + c.row;
    const bool inside = pxi < dm.W && pyi < dm.H;
    PixelAcc a = fresh_pixel(1.0f);
    bool done = !inside;
    if (tx * kTile + (wave & 1) * 8 >= dm.W || ty * kTile + (wave >> 1) * 8 >= dm.H) {  // quadrant outside the image
        if (lane == 0) lds_store(&ring.done[wave], kGone);
        GA_LDS_ORDER();
        while (wraps && duty.next < nch && duty_stage(ring, duty, lane, nch)) {}
        return;
    }

    if (nsegs == 1) {
        if (wraps) consume<true, true, STORE>(ring, c, nch, a, done, st, flags, duty, tseg, 0u);
        else consume<true, false, STORE>(ring, c, nch, a, done, st, flags, duty, tseg, 0u);
        if (inside) write_pixel(a, bg, dm, v, pxi, pyi, out_color, out_others);
    } else {
        const uint32_t work0 = work - seg * wstride;            // work item of segment 0 of this tile (segment k: + k * wstride)
        const int px = wave * 64 + lane;                        // pixel index inside the scratch records
        uint32_t *arrive = seg_sync + 8 * (size_t)pos + wave;
        uint32_t *satw = seg_sync + 8 * (size_t)pos + 4;     // nsegs - k of the earliest saturating segment k (atomic max)
        unsigned long long *mine = seg_scratch + (size_t)work * kSegFloats;
        if (!last_seg) {   // pass 1: transmittance of my segment (nobody needs that of the last one)
            bool d1 = !inside;
            consume<false, false>(ring, c, nch, a, d1, st, flags, duty);
            xwg_store(mine + px, a.T, epoch);
            // every pixel of my quadrant is finished by this segment alone; the fourth quadrant to say so marks the tile
            if (__builtin_amdgcn_ballot_w64(!d1) == 0 && lane == 0 && atomicAdd(&ring.sat_waves[0], 1u) == 3u)
                __hip_atomic_fetch_max(satw, nsegs - seg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // transmittance on entering my segment: product over the lower segments (they started before me: tickets)
        float P = 1.0f;
        for (uint32_t k = 0; k < seg; ++k) P *= xwg_load(seg_scratch + (size_t)(work0 + k * wstride) * kSegFloats + px, epoch);
        // pass 2: the sequential blend of my segment, entered with the global transmittance
        a = fresh_pixel(P);
        done = done || P < 0.0001f;  // T never falls below 1e-4 in the sequential loop: it stopped before this segment
        a.median = -1.0f;            // depths are >= near > 0: a negative median means "not set inside this segment"
        consume<true, false, STORE>(ring, c, nch, a, done, st, flags, duty, tseg, c0);
        unsigned long long *o = mine + 256 + px;
        const float part[14] = {a.N2C0.y, a.C12.x, a.C12.y, a.N01.x, a.N01.y, a.N2C0.x, a.Dp, a.M.x, a.M.y, a.dist,
                                a.median, a.T, done ? 1.0f : 0.0f, P};
#pragma unroll
        for (int f = 0; f < 14; ++f) xwg_store(o + f * 256, part[f], epoch);
        // ... or every pixel has ended its walk by the end of this segment (the exact, global statement)
        if (__builtin_amdgcn_ballot_w64(!done) == 0 && lane == 0 && atomicAdd(&ring.sat_waves[1], 1u) == 3u)
            __hip_atomic_fetch_max(satw, nsegs - seg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        finish_segment(k, dm, lane, pos, work0, wstride, nsegs, px, v, pxi, pyi, inside, arrive);
    }
}


// ---------------------------------------------------------------------------------------------------------------
// SPLIT WALK (round 4) of an unsegmented list: evaluate with lanes = PAIRS, composite with lanes = pixels.
//
// The fused walk above spends 78 VALU instructions per entry slot with 44 % of the lane slots doing work (round 3 counters): a
// lane = pixel waits for the longest survivor list of its wave, and evaluates (25 instructions, two transcendentals) inside that
// lock step.  Here a wavefront still owns one 8x8 quadrant -- so everything below is private to the wave, no barrier and no
// cross-wave protocol beyond the record ring -- but it walks the list in two interleaved phases:
//   A  lanes = (entry, pixel of the entry's cull box inside my quadrant).  Per 64-entry chunk the wave forms the boxes' areas
//      (lanes = entries), their prefix sums and a compact table of the entries that touch the quadrant, then runs over the pairs
//      64 at a time at ~90 % lane use: find my entry (run-start bits + v_mbcnt), my pixel (one multiply-shift division), fetch the
//      record (broadcast LDS reads: ~5 distinct entries per instruction), evaluate alpha and depth.  Pairs that pass are appended
//      to their PIXEL's list in LDS, in list order: pairs are enumerated entry-major, so inside one instruction the lower lane is the
//      earlier entry -- a lane's rank among the passing pairs of its pixel in this instruction is the number of lower lanes in a
//      64-bit mask the lanes OR their own bit into (DS operations of one wave execute in order: OR, read back, count with v_mbcnt),
//      its position is the pixel's tail + rank.  Lists are rank-major (row r = the r-th pending item of every pixel), kRows deep.
//   B  lanes = pixels: composite rows 0 .. (longest pending list) -- the only sequential part, 28 instructions per item and no
//      filter-failed slots -- whenever a pixel's list is full, before the ring of the records' colour / normal quads wraps, and at
//      the end.  tools/split_sim.py prices both phases on the bench scene's real lists.
// The record ring is split: q0..q3 (what A reads) in kSA chunk slots, q4 / q5 (what B gathers by slot) in kSB.
#ifndef GA_BLEND_SPLIT
#define GA_BLEND_SPLIT 1
#endif
#ifndef GA_SPLIT_A
#define GA_SPLIT_A 3
#endif
#ifndef GA_SPLIT_B
#define GA_SPLIT_B (GA_ITEM_CHUNKS >= 8 ? 7 : 6)
#endif
#ifndef GA_SPLIT_ROWS
#define GA_SPLIT_ROWS (GA_ITEM_CHUNKS >= 8 ? 8 : 4)
#endif
#ifndef GA_SPLIT_AU
#define GA_SPLIT_AU 1       // pair instructions per trip of phase A
#endif
#ifndef GA_SPLIT_BU
#define GA_SPLIT_BU 2       // rows per trip of phase B
#endif
#ifndef GA_SPLIT_RANK_ATOMIC
#define GA_SPLIT_RANK_ATOMIC 1   // 1: a pixel's list position from ONE returning LDS atomic (lane order, tools/lds_atomic_order.hip); 0: mask + v_mbcnt
#endif
#ifndef GA_SPLIT_THRESH
#define GA_SPLIT_THRESH 32       // composite opportunistically while at least this many pixels of the quadrant have a pending item
#endif
constexpr int kSA = GA_SPLIT_A, kSB = GA_SPLIT_B, kRows = GA_SPLIT_ROWS, kAU = GA_SPLIT_AU, kBU = GA_SPLIT_BU;

struct WaveLists {                     // private to one wavefront (its quadrant)
    float alpha[kRows][64];            // pending items: item number t of pixel p sits in row t % kRows ([row][pixel])
    float depth[kRows][64];
#if !GA_SPLIT_RANK_ATOMIC
    unsigned long long mask[64];       // per pixel: lanes of the current instruction holding a passing pair of it (zero between)
#endif
    unsigned long long centry[64];     // compact table of the chunk's entries that touch the quadrant
    uint32_t tail[64];                 // per pixel: items appended so far
    uint32_t head[64];                 // per pixel: items composited so far (the pixel's lane keeps the live value in a register)
    uint32_t bits[128];                // bit p: pair slot p of the chunk is the first pair of its entry
    unsigned short slot[kRows][64];    // composite-ring slot of the item's entry
};
static_assert((kRows & (kRows - 1)) == 0, "rows are addressed modulo kRows");

struct __attribute__((aligned(16))) Ring2 {
    float4 pa[4][kSA * 64];            // q0 .. q3 (tile-relative, see stage_chunk2)
    float4 pb4[kSB * 64];              // q4: normal, red
    f2 pb5[kSB * 64];                  // q5: green, blue
    WaveLists wl[4];
    uint32_t stamp[kSA];               // k + 1 once chunk k is staged
    uint32_t doneA[4];                 // chunks wave q has finished evaluating (kGone: it needs nothing any more)
    uint32_t doneB[4];                 // oldest chunk wave q still holds uncomposited items of
};
#ifndef GA_SPLIT_WGS_PER_CU
#define GA_SPLIT_WGS_PER_CU 3
#endif
// (measured: three workgroups of 53 808 bytes do NOT fit a CU -- the launch then runs two per CU -- three of 52 272 do)
static_assert(GA_SPLIT_WGS_PER_CU != 3 || sizeof(Ring2) <= 52272, "three workgroups per CU");
static_assert(GA_ITEM_CHUNKS >= 8 || sizeof(Ring2) <= 39 * 1024, "four workgroups per CU");
static_assert(GA_SPLIT_WGS_PER_CU * (sizeof(Ring2) + 64) <= 160 * 1024, "workgroups per CU");

template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t x)   // (lanes without a source read 0)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROWS, 0xf, true);
}
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x)
{
    x += dpp_u32<0x111>(x); x += dpp_u32<0x112>(x); x += dpp_u32<0x114>(x); x += dpp_u32<0x118>(x);   // row_shr 1, 2, 4, 8
    x += dpp_u32<0x142, 0xa>(x);   // row_bcast:15 into rows 1, 3
    x += dpp_u32<0x143, 0xc>(x);   // row_bcast:31 into rows 2, 3
    return x;
}
__device__ __forceinline__ uint32_t lanes_below(unsigned long long m)   // set bits of m in lanes below mine
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// records of chunk k into their ring slots (rebased to the tile origin as in stage_chunk; no masks: the boxes are formed by the
// consumers).  An entry beyond the list's end gets the "never" box.
__device__ __forceinline__ void stage_chunk2(Ring2 &ring, int lane, uint32_t k, const ChunkRegs &g, bool valid, float tx0, float ty0)
{
    const float ex0 = g.g2.x - tx0, ey0 = g.g2.y - ty0;
    const float ux = tx0 - rintf(g.g2.x), uy = ty0 - rintf(g.g2.y);
    const int ja = (int)(k % kSA) * 64 + lane, jb = (int)(k % kSB) * 64 + lane;
    ring.pa[0][ja] = g.g0;
    ring.pa[1][ja] = make_float4(fmaf(uy, g.g0.z, fmaf(ux, g.g0.x, g.g1.x)), fmaf(uy, g.g0.w, fmaf(ux, g.g0.y, g.g1.y)), g.g1.z, g.g1.w);
    ring.pa[2][ja] = make_float4(ex0, ey0, fmaf(uy, g.g1.w, fmaf(ux, g.g1.z, g.g2.z)), g.g2.w);
    ring.pa[3][ja] = make_float4(g.g3.x, g.g3.y, g.g3.z, valid ? g.g3.w : __uint_as_float(0xBC00BC00u));   // (fp16 -1, -1)
    ring.pb4[jb] = g.g4;
    ring.pb5[jb] = g.g5;
    GA_LDS_ORDER();
    if (lane == 0) lds_store(&ring.stamp[k % kSA], k + 1);
    GA_LDS_ORDER();
}

// Staging duties as above (chunk k by wave k % 4), against the two rings: chunk k takes the evaluation slot of chunk k - kSA and the
// composite slot of chunk k - kSB, so every wave must have evaluated the former and composited every item of the latter.
__device__ __forceinline__ void duty_prologue2(Ring2 &ring, Duty &d, int lane, int quad, uint32_t nch)
{
    const uint32_t k0 = (uint32_t)quad;
    d.next = k0;
    if (k0 >= nch) { d.next = nch; return; }
    if (k0 < (uint32_t)kSA) {
        const uint32_t ida = d.point_list[duty_entry(d, k0, lane)];
        d.next = k0 + 4;
        if (d.next < nch) d.id1 = d.point_list[duty_entry(d, d.next, lane)];
        const ChunkRegs ga = load_chunk(d.rec4, ida);
        if (d.next < nch) d.id2 = d.point_list[duty_entry(d, d.next + 4 < nch ? d.next + 4 : d.next, lane)];
        stage_chunk2(ring, lane, k0, ga, d.sbeg + k0 * 64 + lane < d.send, d.tx0, d.ty0);
    } else {
        d.id1 = d.point_list[duty_entry(d, d.next, lane)];
        d.id2 = d.point_list[duty_entry(d, d.next + 4 < nch ? d.next + 4 : d.next, lane)];
    }
    if (d.next < nch) d.g = load_chunk(d.rec4, d.id1);
}

__device__ __forceinline__ bool duty_stage2(Ring2 &ring, Duty &d, int lane, uint32_t nch)
{
    const uint32_t k = d.next;
    for (;;) {
        const uint32_t a0 = lds_load(&ring.doneA[0]), a1 = lds_load(&ring.doneA[1]), a2 = lds_load(&ring.doneA[2]), a3 = lds_load(&ring.doneA[3]);
        const uint32_t b0 = lds_load(&ring.doneB[0]), b1 = lds_load(&ring.doneB[1]), b2 = lds_load(&ring.doneB[2]), b3 = lds_load(&ring.doneB[3]);
        const uint32_t amin = min(min(a0, a1), min(a2, a3)), bmin = min(min(b0, b1), min(b2, b3));
        if (amin == kGone) return false;
        if (amin + kSA > k && bmin + kSB > k) break;
        __builtin_amdgcn_s_sleep(2);
    }
    GA_LDS_ORDER();
    stage_chunk2(ring, lane, k, d.g, d.sbeg + k * 64 + lane < d.send, d.tx0, d.ty0);
    d.next = k + 4;
    if (d.next < nch) {
        d.id1 = d.id2;
        d.g = load_chunk(d.rec4, d.id1);
        d.id2 = d.point_list[duty_entry(d, d.next + 4 < nch ? d.next + 4 : d.next, lane)];
    }
    return true;
}

// one pending item of my pixel (B).  `go`: the item exists and my pixel's walk has not ended.
__device__ __forceinline__ void composite_item(float alpha, float depth_raw, const float4 &q4, const f2 &q5, PixelAcc &a, bool &done, bool go)
{
    const float kM = kFar / (kFar - kNear);
    const float test_T = a.T * (1.0f - alpha);
    const bool stop = go && test_T < 0.0001f;               // upstream: done = true
    const bool use = go && !stop;
    done = done || stop;
    const float depth = use ? depth_raw : 1.0f;
    const float w = use ? alpha * a.T : 0.0f;
    const float A = 1.0f - a.T;
    const float m = kM * (1.0f - kNear * __builtin_amdgcn_rcpf(depth));
    const f2 mm = f2{m, m * m};
    a.dist += (mm.y * A + a.M.y - 2.0f * m * a.M.x) * w;
    a.Dp += depth * w;
    a.M += mm * w;
    a.median = (use && a.T > 0.5f) ? depth : a.median;
    a.N01 += lo2(q4) * w;
    a.N2C0 += hi2(q4) * w;
    a.C12 += q5 * w;
    a.T = use ? test_T : a.T;
}

struct SplitStats {
    unsigned a_iters, a_pairs, b_rows, b_items;
};
// GA_SPLIT_PROFILE (measurement builds): cycles a wave spends per section, one row per wave in the binning's depth array (tools/split_profile.py)
#ifdef GA_SPLIT_PROFILE
#define GA_PROF_DECL unsigned long long prof_t = __builtin_readcyclecounter(), prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define GA_PROF(sec) do { const unsigned long long prof_n = __builtin_readcyclecounter(); prof_acc[sec] += prof_n - prof_t; prof_t = prof_n; } while (0)
#else
#define GA_PROF_DECL
#define GA_PROF(sec)
#endif

__device__ __forceinline__ void blend_tile_split(Ring2 &ring, const BlendArgs &k, const Dims &dm, int lane, int wave, const uint4 sched,
                                                 SplitStats &st)
{
    const uint32_t vt = sched.x;
    const int v = (int)(vt / (uint32_t)dm.tiles), tile = (int)(vt - (uint32_t)v * dm.tiles);
    const int tx = tile % dm.gx, ty = tile / dm.gx;
    const uint32_t beg = sched.y, n = sched.z, nch = (n + 63) / 64;
    const float4 *rec4 = reinterpret_cast<const float4 *>(k.record) + (size_t)v * dm.N * (kRec / 4);
    WaveLists &wl = ring.wl[wave];
    GA_PROF_DECL;
#if !GA_SPLIT_RANK_ATOMIC
    wl.mask[lane] = 0ull;
#endif
    wl.tail[lane] = 0u; wl.head[lane] = 0u; wl.bits[lane] = 0u; wl.bits[64 + lane] = 0u;

    Duty duty;
    duty.sbeg = beg; duty.send = beg + n; duty.point_list = k.point_list; duty.rec4 = rec4;
    duty.tx0 = (float)(tx * kTile); duty.ty0 = (float)(ty * kTile);
    duty_prologue2(ring, duty, lane, wave, nch);
    GA_PROF(0);

    const int qx = (wave & 1) * 8, qy = (wave >> 1) * 8;          // my quadrant inside the tile
    const int pxi = tx * kTile + qx + (lane & 7), pyi = ty * kTile + qy + (lane >> 3);
    const bool inside = pxi < dm.W && pyi < dm.H;
    PixelAcc a = fresh_pixel(1.0f);
    bool done = !inside;
    unsigned long long done_mask = __builtin_amdgcn_ballot_w64(done);
#if !GA_SPLIT_RANK_ATOMIC
    const unsigned long long lanebit = 1ull << lane;
#endif
    const float qxf = (float)qx, qyf = (float)qy;
    bool pending = false;        // (wave-uniform) items appended since the last composite pass
    uint32_t oldest = 0;         // ... the chunk of the first of them

    // B: composite pending items, one per pixel and step, kBU steps per trip.  all: until nothing is pending; otherwise at least one
    // trip and on while at least kThresh pixels have something (a step costs the same however many lanes have work).  Returns whether
    // every list is empty afterwards.
    uint32_t hd = 0;             // items of my pixel composited so far
    auto drain = [&](bool all) -> bool {
#ifdef GA_SPLIT_PROFILE
        const unsigned long long fl_t0 = __builtin_readcyclecounter();
#endif
        // (a tail may run ahead of what is stored: the pairs of the current instruction that found their list full are written
        // after this pass -- only the kRows items from my head on are there)
        uint32_t cnt = min(wl.tail[lane] - hd, (uint32_t)kRows);
        bool first = true;
        unsigned long long act;
        for (;;) {
            act = __builtin_amdgcn_ballot_w64(cnt != 0u);
            if (act == 0ull) break;
            if (!all && !first && __builtin_popcountll(act) < GA_SPLIT_THRESH) break;
            first = false;
            float al[kBU], dp[kBU];
            uint32_t sl[kBU];
            bool g[kBU];
#pragma unroll
            for (int u = 0; u < kBU; ++u) {
                const uint32_t row = (hd + (uint32_t)u) & (uint32_t)(kRows - 1);
                al[u] = wl.alpha[row][lane]; dp[u] = wl.depth[row][lane];
                g[u] = (uint32_t)u < cnt;
                sl[u] = g[u] ? (uint32_t)wl.slot[row][lane] : 0u;   // (rows beyond my list hold stale slots: stay inside the ring)
            }
            float4 q4[kBU];
            f2 q5[kBU];
#pragma unroll
            for (int u = 0; u < kBU; ++u) { q4[u] = ring.pb4[sl[u]]; q5[u] = ring.pb5[sl[u]]; }
#pragma unroll
            for (int u = 0; u < kBU; ++u) composite_item(al[u], dp[u], q4[u], q5[u], a, done, g[u] && !done);
            const uint32_t adv = min(cnt, (uint32_t)kBU);
            hd += adv; cnt -= adv;
            if (k.flags & GA_SURFEL_FLAG_STATS) {
                st.b_rows += kBU;
                uint32_t c = adv;
                for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
                st.b_items += c;
            }
        }
        wl.head[lane] = hd;
        GA_LDS_ORDER();
        done_mask = __builtin_amdgcn_ballot_w64(done);
#ifdef GA_SPLIT_PROFILE
        { const unsigned long long fl_n = __builtin_readcyclecounter(); prof_acc[5] += fl_n - fl_t0; prof_t += fl_n - fl_t0; }
#endif
        return act == 0ull;
    };

    uint32_t i = 0;
    for (; i < nch; ++i) {
        if (done_mask == ~0ull) break;
        // the composite ring is about to wrap onto entries my pending items point at
        if (pending && oldest + (uint32_t)(kSB - kSA) <= i) { drain(true); pending = false; }
        GA_LDS_ORDER();
        if (lane == 0) { lds_store(&ring.doneA[wave], i); lds_store(&ring.doneB[wave], pending ? oldest : i); }
        GA_LDS_ORDER();
        GA_PROF(4);
        while (duty.next < nch && duty.next <= i + (uint32_t)kSA - 1u && duty_stage2(ring, duty, lane, nch)) {}
        GA_PROF(1);
        const int sa = (int)(i % kSA) * 64;
        while (lds_load(&ring.stamp[i % kSA]) != i + 1) __builtin_amdgcn_s_sleep(1);
        GA_LDS_ORDER();
        GA_PROF(2);
        // ---- lanes = entries: cull box inside my quadrant, pair prefix, compact table
        uint32_t total;
        {
            const float4 c2 = ring.pa[2][sa + lane];
            const uint32_t cull = __float_as_uint(ring.pa[3][sa + lane].w);
            const float rx = __half2float(__ushort_as_half((unsigned short)(cull & 0xffffu)));
            const float ry = __half2float(__ushort_as_half((unsigned short)(cull >> 16)));
            // pixel columns c with |c - ex0| <= rx are ceil(ex0 - rx) .. floor(ex0 + rx)  (+inf: all; negative: none)
            const float xlo = fmaxf(ceilf(c2.x - rx), qxf), xhi = fminf(floorf(c2.x + rx), qxf + 7.0f);
            const float ylo = fmaxf(ceilf(c2.y - ry), qyf), yhi = fminf(floorf(c2.y + ry), qyf + 7.0f);
            const int w = (int)(xhi - xlo) + 1, h = (int)(yhi - ylo) + 1;
            const bool nz = xhi >= xlo && yhi >= ylo;
            const uint32_t area = nz ? (uint32_t)(w * h) : 0u;
            const uint32_t incl = wave_inclusive_scan(area);
            total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (total != 0) {
                const unsigned long long nzm = __builtin_amdgcn_ballot_w64(nz);
                if (nz) {
                    const uint32_t base = incl - area;
                    const uint32_t magic = (uint32_t)ceilf(65536.0f * __builtin_amdgcn_rcpf((float)w));   // (k * magic) >> 16 == k / w for k < 64
                    const uint32_t lo = (uint32_t)lane | ((uint32_t)((int)xlo - qx) << 6) | ((uint32_t)((int)ylo - qy) << 9) | ((uint32_t)w << 12) | (base << 16);
                    wl.centry[lanes_below(nzm)] = ((unsigned long long)magic << 32) | lo;
                    __hip_atomic_fetch_or(&wl.bits[base >> 5], 1u << (base & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        GA_LDS_ORDER();
        GA_PROF(3);
        // ---- lanes = pairs, kAU instructions' worth per trip: their table look-ups, record fetches and evaluations are independent of
        // each other (the LDS round trips overlap), the appends follow in order
        uint32_t cbase = 0;
        const uint32_t bslot0 = (i % (uint32_t)kSB) * 64u;
        for (uint32_t p0 = 0; p0 < total; p0 += 64u * kAU) {
            bool valid[kAU];
            uint32_t lo[kAU], pq[kAU];
            Rec r[kAU];
            f2 dxy[kAU];
#pragma unroll
            for (int u = 0; u < kAU; ++u) {
                const uint32_t p = p0 + 64u * u + (uint32_t)lane;
                valid[u] = p < total;
                const uint32_t word = wl.bits[(p >> 5) & 127u];
                const bool start = valid[u] && ((word >> (p & 31u)) & 1u) != 0u;
                const unsigned long long S = __builtin_amdgcn_ballot_w64(start);
                const uint32_t c = valid[u] ? cbase + lanes_below(S) + (start ? 1u : 0u) - 1u : 0u;
                cbase += (uint32_t)__builtin_popcountll(S);
                const unsigned long long ce = wl.centry[c];
                lo[u] = (uint32_t)ce;
                const uint32_t magic = (uint32_t)(ce >> 32);
                const uint32_t kk = valid[u] ? p - (lo[u] >> 16) : 0u, w = (lo[u] >> 12) & 15u;
                const uint32_t dy = __umul24(kk, magic) >> 16, dx = kk - __umul24(dy, w);
                const uint32_t px = ((lo[u] >> 6) & 7u) + dx, py = ((lo[u] >> 9) & 7u) + dy;
                pq[u] = (py * 8u + px) & 63u;
                dxy[u] = f2{(float)(qx + (int)px), (float)(qy + (int)py)};
                const int j = sa + (int)(lo[u] & 63u);
                r[u].q0 = ring.pa[0][j]; r[u].q1 = ring.pa[1][j]; r[u].q2 = ring.pa[2][j]; r[u].q3 = ring.pa[3][j];
            }
            Alpha e[kAU];
            float depth[kAU];
            bool pass[kAU];
#pragma unroll
            for (int u = 0; u < kAU; ++u) {
                e[u] = eval_alpha(r[u], dxy[u]);
                depth[u] = pair_depth(r[u], e[u]);
                pass[u] = valid[u] && e[u].pass && !(depth[u] < kNear);
                if (done_mask != 0ull) pass[u] = pass[u] && ((done_mask >> pq[u]) & 1ull) == 0ull;   // (pixels whose walk has ended take nothing)
                if (k.flags & GA_SURFEL_FLAG_STATS) {
                    const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid[u]);
                    st.a_iters += vm != 0ull; st.a_pairs += (unsigned)__builtin_popcountll(vm);
                }
            }
#pragma unroll
            for (int u = 0; u < kAU; ++u) {
                if (__builtin_amdgcn_ballot_w64(pass[u]) == 0ull) continue;
                if (!pending) { pending = true; oldest = i; }
                uint32_t hq = wl.head[pq[u]];
#if GA_SPLIT_RANK_ATOMIC
                // my place in my pixel's list from one returning atomic: the LDS serves the lanes of an instruction that hit the same
                // address in lane order (tools/lds_atomic_order.hip: 5.1 M conflicting lanes, none out of order; tests/test_surfel_gpu.py
                // runs it), and pairs are enumerated entry-major -- the lower lane is the earlier entry
                uint32_t pos = 0;
                if (pass[u]) pos = __hip_atomic_fetch_add(&wl.tail[pq[u]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
                // my place in my pixel's list: its tail + the passing pairs of the same pixel in lower lanes (= earlier entries)
                if (pass[u]) __hip_atomic_fetch_or(&wl.mask[pq[u]], lanebit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                GA_LDS_ORDER();
                const unsigned long long m = wl.mask[pq[u]];
                const uint32_t t = wl.tail[pq[u]];
                GA_LDS_ORDER();
                const uint32_t rank = lanes_below(m);
                if (pass[u] && rank == 0u) { wl.tail[pq[u]] = t + (uint32_t)__builtin_popcountll(m); wl.mask[pq[u]] = 0ull; }
                GA_LDS_ORDER();
                uint32_t pos = t + rank;
#endif
                bool pend = pass[u];
                for (;;) {
                    const bool fit = pend && pos - hq < (uint32_t)kRows;
                    if (fit) {
                        const uint32_t row = pos & (uint32_t)(kRows - 1);
                        wl.alpha[row][pq[u]] = e[u].alpha; wl.depth[row][pq[u]] = depth[u];
                        wl.slot[row][pq[u]] = (unsigned short)(bslot0 + (lo[u] & 63u));
                    }
                    pend = pend && !fit;
                    GA_LDS_ORDER();
                    if (__builtin_amdgcn_ballot_w64(pend) == 0ull) break;
                    drain(false);          // a list is full: composite (at least one trip), the rest of this instruction's pairs follow
                    hq = wl.head[pq[u]];
                }
            }
        }
        // composite now if most of the quadrant has something pending (full lanes for the sequential part)
        if (pending) {
            const unsigned long long has = __builtin_amdgcn_ballot_w64(wl.tail[lane] != hd);
            if (__builtin_popcountll(has) >= GA_SPLIT_THRESH && drain(false)) pending = false;
        }
        // the run-start bits of this chunk
        for (uint32_t wd = (uint32_t)lane; wd * 32u < total; wd += 64u) wl.bits[wd] = 0u;
        GA_LDS_ORDER();
    }
    GA_PROF(4);
    if (pending) drain(true);
    GA_LDS_ORDER();
    if (lane == 0) { lds_store(&ring.doneA[wave], kGone); lds_store(&ring.doneB[wave], kGone); }
    GA_LDS_ORDER();
    if (inside) write_pixel(a, k.bg, dm, v, pxi, pyi, k.out_color, k.out_others);
    GA_PROF(4);
    while (duty.next < nch && duty_stage2(ring, duty, lane, nch)) {}   // the others may still need my chunks
    GA_PROF(6);
#ifdef GA_SPLIT_PROFILE
    // sections: 0 prologue, 1 staging duty (with its waits), 2 stamp wait, 3 chunk table, 4 pair instructions (+ bookkeeping),
    // 5 composite passes, 6 trailing duties; word 11: waves
    if (lane == 0) {   // one row of 8 words per wave in the (dead by now) depth array of the binning: plain stores, summed by tools/split_profile.py
        int64_t *row = k.prof + ((size_t)blockIdx.x * 4 + wave) * 8;
        unsigned long long tot = 0;
        for (int q = 0; q < 7; ++q) { row[q] = (int64_t)prof_acc[q]; tot += prof_acc[q]; }
        row[7] = (int64_t)(tot | (0x5A5Aull << 48));   // (tag: the array also holds what the binning left there)
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Grid: [ segment region | one workgroup per (view, tile) ].  The kSegWGs workgroups of the segment region take tickets
// (status[GA_STATUS_SEG_TICKET]) until the status[GA_STATUS_SEG_WORK] segment work items are handed out (the host does not
// know their number; a region sized for the worst case, capacity / 256 workgroups that mostly find nothing to do, cost
// 20 % of the launch); the segmented tiles are the first status[GA_STATUS_LONG_TILES] slots of tile_order, which the
// tile region skips.
#ifndef GA_BLEND_WAVES
#define GA_BLEND_WAVES 3
#endif
#ifndef GA_BLEND_XCD_RUNS
#define GA_BLEND_XCD_RUNS 3   // log2 of the run length (0: off)
#endif
template <bool STORE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GA_BLEND_WAVES, GA_BLEND_WAVES))) void surfel_blend_kernel(BlendArgs k, Dims dm, int ntiles,
                                                           uint32_t seg_region,
                                                           const uint32_t *__restrict__ seg_table,
                                                           int64_t *__restrict__ status)
{
    // one LDS allocation, two images: the fused walk's ring (segments, the differentiable forward) / the split walk's
    __shared__ __attribute__((aligned(16))) unsigned char lds_image[sizeof(Ring) > sizeof(Ring2) ? sizeof(Ring) : sizeof(Ring2)];
    Ring &ring = *reinterpret_cast<Ring *>(lds_image);
    // (requested before the status words are looked at: on overflow the entry is stale but the slot exists)
    // schedule slot of a tile workgroup: runs of 2^GA_BLEND_XCD_RUNS consecutive slots -- tiles of one view and one pair of tile rows that
    // fall into the same length class (the order the tile scan leaves inside a class) -- go to workgroups eight apart, i.e.
    // to ONE XCD (workgroup ids are dealt round-robin to the eight XCDs), whose L2 then serves the records neighbouring tiles
    // share; a transposition inside blocks of 8 runs, so the longest-first order and the balance between XCDs stay
    uint32_t tslot = blockIdx.x >= seg_region ? blockIdx.x - seg_region : 0u;
#if GA_BLEND_XCD_RUNS
    {
        constexpr uint32_t kRun = 1u << GA_BLEND_XCD_RUNS, kBlock = 8u * kRun;   // run length, slots per transposed block
        if ((tslot | (kBlock - 1u)) < (uint32_t)ntiles) tslot = (tslot & ~(kBlock - 1u)) | ((tslot & 7u) * kRun) | ((tslot / 8u) & (kRun - 1u));
    }
#endif
    const uint4 my_sched = k.tile_order[tslot];
    const int64_t overflow = status[GA_STATUS_OVERFLOW], nlong64 = status[GA_STATUS_LONG_TILES];
    const int64_t segwork64 = status[GA_STATUS_SEG_WORK];
    if (overflow) return;
    const uint32_t nlong = (uint32_t)nlong64, segwork = (uint32_t)segwork64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Stats st{};
#ifdef GA_BLEND_STAMPS
    // measurement build: (entry, exit) wall-clock stamps and hardware id per wave, in the upper half of the segment scratch
    unsigned long long *stamps = k.seg_scratch + (size_t)GA_BLEND_STAMPS + ((size_t)blockIdx.x * 4 + wave) * 4;
    const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
    struct StampExit {
        unsigned long long *p, t0; int lane;
        __device__ ~StampExit() {
            if (lane == 0) { p[0] = t0; p[1] = __builtin_amdgcn_s_memrealtime(); p[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); }
        }
    } stamp_exit{stamps, t_entry, lane};
#endif
    if (blockIdx.x < seg_region) {
#ifdef GA_BLEND_PROBE
        if (GA_BLEND_PROBE == 2) return;
#endif
        if (blockIdx.x >= segwork) return;   // more workgroups than work items: whole workgroups leave
        BlendArgs ks = k;
        ks.epoch = seg_table[kSegEpochWord];   // this launch's epoch (written by the tile scan, a kernel boundary ago)
        for (;;) {
            if (threadIdx.x == 0)
                ring.work = (uint32_t)atomicAdd(reinterpret_cast<unsigned long long *>(status + GA_STATUS_SEG_TICKET), 1ull);
            if (threadIdx.x < kItemChunks) ring.stamp[threadIdx.x] = 0;
            if (threadIdx.x < 4) ring.done[threadIdx.x] = 0;
            __syncthreads();
            const uint32_t work = ring.work;
            if (work >= segwork) break;
            // length class of the work item: classes are laid out longest first, seg_table[b] = (first slot, first work
            // item) (non-increasing in b; class b owns [first(b), first(b - 1)); classes above the longest populated one
            // start at 0)
            int b = kSegClass;
            while (b < 32 && seg_table[2 * b + 1] > work) ++b;
            // inside a class the work items run segment-major (segment 0 of all its tiles, then segment 1, ...): a tile's later
            // segments start late enough to find the tile saturated, and still only ever wait for lower tickets
            const uint32_t nsegs = seg_count(b), rel = work - seg_table[2 * b + 1];
            const uint32_t ntile = seg_table[2 * (b - 1)] - seg_table[2 * b];
            const uint32_t pos = seg_table[2 * b] + rel % ntile;
            if (threadIdx.x == 0) {   // one reading of the tile's saturation word for the whole workgroup
                ring.sat = __hip_atomic_load(k.seg_sync + 8 * (size_t)pos + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ring.sat_waves[0] = 0; ring.sat_waves[1] = 0;
            }
            __syncthreads();
            blend_item<STORE>(ring, ks, dm, lane, wave, pos, k.tile_order[pos], rel / ntile, nsegs, work, ntile, st);
            __syncthreads();   // everybody has left the item: the LDS image and the ticket word may be overwritten
        }
    } else {
        // schedule slot = block index: no dependence on the status words (one load latency less before the first record
        // arrives); the slots of the segmented tiles lead the schedule and belong to the segment region
        const uint32_t pos = tslot;
        if (pos < nlong) return;
#ifdef GA_BLEND_PROBE   // timing-only builds (wrong images): 1 = leave the tiles with lists of fewer than 512 entries out, 2 = only those
        if (GA_BLEND_PROBE == 1 && my_sched.z > 0 && my_sched.z < 512) return;
        if (GA_BLEND_PROBE == 2 && my_sched.z >= 512) return;
#endif
        if (my_sched.z == 0) {   // empty list (more than half of the tiles at BASELINE configs[1]): background pixels, nothing else
            if (!(k.flags & GA_SURFEL_FLAG_BG_IN_BLEND)) return;   // (round 6: written by the sort launch's waves, surfel_bin.hip)
            const int v = (int)(my_sched.x / (uint32_t)dm.tiles), tile = (int)(my_sched.x - (uint32_t)v * dm.tiles);
            const int pxi = (tile % dm.gx) * kTile + (wave & 1) * 8 + (lane & 7), pyi = (tile / dm.gx) * kTile + (wave >> 1) * 8 + (lane >> 3);
            if (pxi < dm.W && pyi < dm.H) write_pixel(fresh_pixel(1.0f), k.bg, dm, v, pxi, pyi, k.out_color, k.out_others);
            return;
        }
#if GA_BLEND_SPLIT
        if (!STORE && (k.flags & GA_SURFEL_FLAG_SPLIT_WALK)) {
            Ring2 &ring2 = *reinterpret_cast<Ring2 *>(lds_image);
            if (threadIdx.x < kSA) ring2.stamp[threadIdx.x] = 0;
            if (threadIdx.x < 4) { ring2.doneA[threadIdx.x] = 0; ring2.doneB[threadIdx.x] = 0; }
            __syncthreads();
            SplitStats ss{};
            blend_tile_split(ring2, k, dm, lane, wave, my_sched, ss);
            if (k.flags & GA_SURFEL_FLAG_STATS) {
                // the counters of the fused walk, in the split walk's terms: wave-level slots = A instructions + B rows; lanes with
                // work = pairs evaluated + items composited
                st.iters = ss.a_iters + ss.b_rows; st.useful = ss.a_pairs + ss.b_items; st.chunks = 0; st.lanemax = ss.b_rows;
                if (lane == 0) {
                    atomicAdd(reinterpret_cast<unsigned long long *>(status + GA_STATUS_SPLIT_A_ITERS), (unsigned long long)ss.a_iters);
                    atomicAdd(reinterpret_cast<unsigned long long *>(status + GA_STATUS_SPLIT_A_PAIRS), (unsigned long long)ss.a_pairs);
                    atomicAdd(reinterpret_cast<unsigned long long *>(status + GA_STATUS_SPLIT_B_ROWS), (unsigned long long)ss.b_rows);
                    atomicAdd(reinterpret_cast<unsigned long long *>(status + GA_STATUS_SPLIT_B_ITEMS), (unsigned long long)ss.b_items);
                }
            }
        } else
#endif
        {
            if (threadIdx.x < kItemChunks) ring.stamp[threadIdx.x] = 0;
            if (threadIdx.x < 4) ring.done[threadIdx.x] = 0;
            __syncthreads();
            blend_item<STORE>(ring, k, dm, lane, wave, pos, my_sched, 0u, 1u, 0u, 0u, st);
        }
    }
    if ((k.flags & GA_SURFEL_FLAG_STATS) && lane == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(status + GA_STATUS_BLEND_ITERS), (unsigned long long)st.iters);
        atomicMax(reinterpret_cast<unsigned long long *>(status + GA_STATUS_BLEND_MAX_ITERS), (unsigned long long)st.iters);
        atomicAdd(reinterpret_cast<unsigned long long *>(status + GA_STATUS_BLEND_LANE_SLOTS), (unsigned long long)st.useful);
        atomicAdd(reinterpret_cast<unsigned long long *>(status + GA_STATUS_BLEND_CHUNKS), (unsigned long long)st.chunks);
        atomicAdd(reinterpret_cast<unsigned long long *>(status + GA_STATUS_BLEND_LANE_MAX), (unsigned long long)st.lanemax);
    }
}

void launch_blend(const GaSurfelForwardArgs &a, const Dims &d, const Workspace &ws, hipStream_t s)
{
    const int nt = d.V * d.tiles;
    // a segmented tile of class b holds >= 2^(b-1) entries and takes seg_count(b) = 2^(b-9) = 2^(b-1) / 256 work items
    constexpr uint32_t kSegWGs = 512;   // two of the three workgroup slots of a CU; beyond that they loop
    const uint32_t seg_region = (uint32_t)std::min<int64_t>(a.capacity / 256, kSegWGs);
    const BlendArgs k{ws.tile_order, ws.point_list, ws.record, a.bg, ws.seg_sync, ws.seg_scratch, a.out_color, a.out_others,
                      0u /* the segment workgroups read the launch epoch from the workspace */, a.flags, a.seg_T, reinterpret_cast<int64_t *>(ws.depth)};
    if (a.seg_T)
        hipLaunchKernelGGL(surfel_blend_kernel<true>, dim3(seg_region + (unsigned)nt), dim3(256), 0, s, k, d, nt, seg_region,
                           ws.seg_table, ws.status);
    else
        hipLaunchKernelGGL(surfel_blend_kernel<false>, dim3(seg_region + (unsigned)nt), dim3(256), 0, s, k, d, nt, seg_region,
                           ws.seg_table, ws.status);
}

}  // namespace ga
