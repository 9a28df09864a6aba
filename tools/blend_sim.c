/* tools/blend_sim.c -- trip-count model of lanes=pixels blend schedules on real tile lists (design aid, not product).
 * Input per view: depth-ordered tile lists, per-splat centre + cull half-extents, per-pixel number of list entries visited
 * (oracle n_walked).  For every tile the survivor count of every pixel in every 64-entry chunk is formed; schedules are
 * then simulated on those counts.  Build: gcc -O2 -shared -fPIC -o tools/_build/libblend_sim.so tools/blend_sim.c */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double pairs;          /* useful (pixel, entry) evaluations */
    double slots;          /* lane slots spent in trips (trips * kU * 64) */
    double trips;          /* wave-level trips */
    double chunks;         /* wave-level chunk stagings */
    double waves;          /* waves that did anything */
    double max_wave_cost;  /* instruction-model cost of the most expensive wave */
    double total_cost;
    double coupled_cost;   /* sum over work items of 4 x the most expensive quadrant (quadrants in lock step) */
} SimOut;

#define MAXCH 512
static int g_chunk = 64;   /* entries per window unit (64: the kernel's chunks; 32: half-chunk windows, round 5 study) */
void blend_sim_set_chunk(int c) { g_chunk = c; }

/* windowed schedule on one wave: cnt[lane][chunk]; window of W chunks; kU entries per trip; a step ends when the tail
 * chunk is exhausted in every lane (W = 2 is the current kernel).  Returns trips; *chunks = chunks staged. */
static long sim_window(const uint16_t (*cnt)[MAXCH], const int *lanes_last_chunk, int nch, int W, int kU, long *chunks)
{
    static uint16_t rem[64][MAXCH];
    int last = -1;
    for (int l = 0; l < 64; ++l) if (lanes_last_chunk[l] > last) last = lanes_last_chunk[l];
    *chunks = last + 1;
    if (last < 0) return 0;
    for (int l = 0; l < 64; ++l) memcpy(rem[l], cnt[l], sizeof(uint16_t) * (last + 1));
    long trips = 0;
    /* staged chunks: tail .. head-1; kernel: step k stages chunk k then runs trips until chunk k-1 is exhausted */
    for (int k = 0; k <= last + 1; ++k) {
        /* chunk k staged (if k <= last); tail = k - (W - 1): run until chunk `tail` exhausted in all lanes */
        int tail = k - (W - 1);
        if (k == last + 1) tail = last;              /* drain everything */
        if (tail < 0) continue;
        for (;;) {
            int any = 0;
            for (int l = 0; l < 64 && !any; ++l)
                for (int c = (tail - (W - 1) > 0 ? tail - (W - 1) : 0); c <= tail; ++c) if (rem[l][c]) { any = 1; break; }
            if (!any) break;
            ++trips;
            const int hi = k <= last ? k : last;
            for (int l = 0; l < 64; ++l) {
                int take = kU;
                for (int c = 0; c <= hi && take > 0; ++c) {   /* oldest first */
                    if (!rem[l][c]) continue;
                    const int t = rem[l][c] < take ? rem[l][c] : take;
                    rem[l][c] -= (uint16_t)t; take -= t;
                }
            }
        }
    }
    return trips;
}

/* per-lane chunk pointers: a lane takes up to kU entries of ITS current chunk per trip and moves on to the next chunk
 * (mask fetch) when that one is exhausted; it may be at most W - 1 chunks ahead of the slowest lane of the wave.
 * skip = 1: empty chunks cost the lane nothing (it fetches masks until it finds a non-empty one within the window). */
static long sim_pointer(const uint16_t (*cnt)[MAXCH], const int *lanes_last_chunk, int nch, int W, int kU, int skip)
{
    static uint16_t rem[64][MAXCH];
    int pc[64], last = -1;
    for (int l = 0; l < 64; ++l) if (lanes_last_chunk[l] > last) last = lanes_last_chunk[l];
    if (last < 0) return 0;
    for (int l = 0; l < 64; ++l) { memcpy(rem[l], cnt[l], sizeof(uint16_t) * (last + 1)); pc[l] = 0; }
    long trips = 0;
    for (;;) {
        int tail = last + 1;
        for (int l = 0; l < 64; ++l) {
            while (pc[l] <= last && rem[l][pc[l]] == 0 && (skip || 1)) { if (!skip) break; ++pc[l]; }
            if (pc[l] < tail) tail = pc[l];
        }
        if (!skip) { /* without skipping: a lane advances one chunk per trip when its chunk is empty */ }
        if (tail > last) break;
        ++trips;
        for (int l = 0; l < 64; ++l) {
            if (pc[l] > last) continue;
            if (pc[l] >= tail + W) continue;               /* outside the window: idle */
            if (rem[l][pc[l]] == 0) { ++pc[l]; continue; }  /* (only reached when skip = 0) */
            const int t = rem[l][pc[l]] < kU ? rem[l][pc[l]] : kU;
            rem[l][pc[l]] -= (uint16_t)t;
        }
    }
    return trips;
}

/* ideal: every lane walks its own list back to back: trips = ceil(max lane total / kU) */
static long sim_ideal(const uint16_t (*cnt)[MAXCH], int nch, int kU)
{
    long mx = 0;
    for (int l = 0; l < 64; ++l) { long s = 0; for (int c = 0; c < nch; ++c) s += cnt[l][c]; if (s > mx) mx = s; }
    return (mx + kU - 1) / kU;
}

/* map: 0 = 8x8 quadrants (wave w = quadrant), 1 = 2x2-interleaved lattice (wave w takes pixels with (x&1, y&1) = w) */
void blend_sim(int H, int W_img, int ntiles_x, int ntiles_y, const uint32_t *ranges, const uint32_t *point_list,
               const float *cx, const float *cy, const float *rx, const float *ry, const uint32_t *n_walked,
               int window, int kU, int map, int seg_min, int nseg, double c_stage, double c_trip, double c_entry,
               SimOut *win, SimOut *ideal)
{
    static uint16_t cnt[4][64][MAXCH];
    memset(win, 0, sizeof(*win)); memset(ideal, 0, sizeof(*ideal));
    for (int ty = 0; ty < ntiles_y; ++ty)
        for (int tx = 0; tx < ntiles_x; ++tx) {
            const int tile = ty * ntiles_x + tx;
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            const int n = (int)(r1 - r0);
            if (n <= 0) continue;
            int segs = (n >= seg_min) ? nseg : 1;
            const int chunks_total = (n + 63) / 64;
            int cps = (chunks_total + segs - 1) / segs;
            for (int sg = 0; sg < segs; ++sg) {
                const int sb = sg * cps * 64, se = (sb + cps * 64 < n) ? sb + cps * 64 : n;
                if (sb >= n) break;
                const int nch = (se - sb + g_chunk - 1) / g_chunk;
                if (nch > MAXCH) continue;
                memset(cnt, 0, sizeof(cnt));
                int lastc[4][64];
                static long proxy[4][64];   /* survivors of the first 8 chunks whether or not the pixel is still alive */
                memset(proxy, 0, sizeof(proxy));
                for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) lastc[w][l] = -1;
                double pairs = 0;
                for (int j = sb; j < se; ++j) {
                    const uint32_t id = point_list[r0 + j];
                    const int ch = (j - sb) / g_chunk;
                    uint32_t colm = 0, rowm = 0;
                    for (int k = 0; k < 16; ++k) {
                        const float px = (float)(tx * 16 + k), py = (float)(ty * 16 + k);
                        float dx = px - cx[id]; if (dx < 0) dx = -dx;
                        float dy = py - cy[id]; if (dy < 0) dy = -dy;
                        if (dx <= rx[id]) colm |= 1u << k;
                        if (dy <= ry[id]) rowm |= 1u << k;
                    }
                    for (int y = 0; y < 16; ++y) {
                        const int pyi = ty * 16 + y;
                        if (pyi >= H) break;
                        for (int x = 0; x < 16; ++x) {
                            const int pxi = tx * 16 + x;
                            if (pxi >= W_img) break;
                            const int walked = (int)n_walked[(size_t)pyi * W_img + pxi];
                            int w, l;
                            if (map == 0) { w = (y >> 3) * 2 + (x >> 3); l = (y & 7) * 8 + (x & 7); }
                            else { w = (y & 1) * 2 + (x & 1); l = (y >> 1) * 8 + (x >> 1); }
                            if (ch < 8 * 64 / g_chunk && ((colm >> x) & 1) && ((rowm >> y) & 1)) proxy[w][l]++;
                            if (j < walked) {
                                /* the lane is alive at this entry: its wave has to stage this chunk */
                                if (ch > lastc[w][l]) lastc[w][l] = ch;
                                if (((colm >> x) & 1) && ((rowm >> y) & 1)) { cnt[w][l][ch]++; pairs += 1; }
                            }
                        }
                    }
                }
                if (map >= 2) {
                    /* regroup the tile's 256 pixels into 4 waves by total list length (longest 64 together, ...) */
                    static uint16_t flat[256][MAXCH]; static int fl[256]; long tot[256]; int ord[256];
                    for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) {
                        memcpy(flat[w * 64 + l], cnt[w][l], sizeof(uint16_t) * nch); fl[w * 64 + l] = lastc[w][l];
                        long s_ = 0; for (int c = 0; c < nch; ++c) s_ += cnt[w][l][c]; tot[w * 64 + l] = map >= 4 ? proxy[w][l] : s_; ord[w * 64 + l] = w * 64 + l;
                    }
                    for (int a = 1; a < 256; ++a) { int v = ord[a], b = a - 1; while (b >= 0 && tot[ord[b]] < tot[v]) { ord[b + 1] = ord[b]; --b; } ord[b + 1] = v; }
                    for (int r = 0; r < 256; ++r) {
                        int w, l;
                        if (map == 2 || map == 4) { w = r / 64; l = r % 64; }           /* sorted groups (4: by the proxy) */
                        else { w = r % 4; l = r / 4; }                      /* map 3: round-robin (every wave sees the same mix) */
                        memcpy(cnt[w][l], flat[ord[r]], sizeof(uint16_t) * nch); lastc[w][l] = fl[ord[r]];
                    }
                }
                const int passes = segs > 1 ? 2 : 1;   /* segmented lists are walked twice (transmittance pre-pass) */
                double tile_max_w = 0, tile_max_i = 0;
                for (int w = 0; w < 4; ++w) {
                    long chunks = 0;
                    long tr = sim_window((const uint16_t (*)[MAXCH])cnt[w], lastc[w], nch, window > 0 ? window : 2, kU, &chunks);
                    if (window < 0) tr = sim_pointer((const uint16_t (*)[MAXCH])cnt[w], lastc[w], nch, -window, kU, 1);
                    const long ti = sim_ideal((const uint16_t (*)[MAXCH])cnt[w], nch, kU);
                    if (chunks == 0) continue;
                    const double cw = passes * (chunks * c_stage + tr * (c_trip + kU * c_entry));
                    const double ci = passes * (chunks * c_stage + ti * (c_trip + kU * c_entry));
                    win->trips += passes * tr; win->slots += passes * tr * kU * 64.0; win->chunks += passes * chunks; win->waves += 1;
                    win->total_cost += cw; if (cw > win->max_wave_cost) win->max_wave_cost = cw;
                    if (cw > tile_max_w) tile_max_w = cw;
                    if (ci > tile_max_i) tile_max_i = ci;
                    ideal->trips += passes * ti; ideal->slots += passes * ti * kU * 64.0; ideal->chunks += passes * chunks; ideal->waves += 1;
                    ideal->total_cost += ci; if (ci > ideal->max_wave_cost) ideal->max_wave_cost = ci;
                }
                win->coupled_cost += 4 * tile_max_w; ideal->coupled_cost += 4 * tile_max_i;
                win->pairs += passes * pairs; ideal->pairs += passes * pairs;
            }
        }
}
