/* tools/split_sim.c -- lane-slot model of the SPLIT blend (design aid, not product): phase A evaluates the (entry, pixel of its
 * cull box) pairs of a group of G list entries with lanes = pairs; phase B composites with lanes = pixels over the compact
 * per-pixel lists of the pairs that passed.  Counts, on the real tile lists of a view:
 *   cand_all / pass_all      every pair of every cull box inside its tile / those that pass the alpha + near tests
 *   cand_walk / pass_walk    ... restricted to entries a pixel really visits (oracle n_walked)
 *   candA                    pairs phase A evaluates: all pairs of the groups that start before the tile's last pixel stops
 *   passA                    pairs it leaves in the per-pixel lists
 *   slotsB_quad / _sorted    lane slots of phase B (64 x longest list of the wave, per group) with waves = 8x8 quadrants / the
 *                            256 pixels regrouped by list length inside each group
 *   slotsB_sorted_tile       regrouped once per tile by the total list length
 * Build: gcc -O2 -shared -fPIC -o tools/_build/libsplit_sim.so tools/split_sim.c -lm */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NPOL 12
static const int polK[NPOL] = {8, 8, 10, 10, 12, 12, 16, 16, 24, 32, 10, 12};
static const int polF[NPOL] = {4, 8, 4, 8, 4, 8, 4, 8, 8, 8, 1000, 1000};
typedef struct {
    double polslots[NPOL], polflush[NPOL], quad_iters, quad_chunks;
    double cand_all, pass_all, cand_walk, pass_walk, candA, passA, slotsB_quad, slotsB_sorted, slotsB_sorted_tile, groups, itersA,
        max_group_pairs, entriesA;
} SplitOut;

static int cmp_desc(const void *a, const void *b) { return *(const int *)b - *(const int *)a; }

void split_sim(int H, int W_img, int ntx, int nty, const uint32_t *ranges, const uint32_t *point_list, const float *cx,
               const float *cy, const float *rx, const float *ry, const float *trans, const float *opa,
               const uint32_t *n_walked, int G, int kU, SplitOut *out)
{
    memset(out, 0, sizeof(*out));
    for (int ty = 0; ty < nty; ++ty)
        for (int tx = 0; tx < ntx; ++tx) {
            const int tile = ty * ntx + tx;
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            const int n = (int)(r1 - r0);
            if (n <= 0) continue;
            int walked[256], maxwalk = 0;
            for (int y = 0; y < 16; ++y)
                for (int x = 0; x < 16; ++x) {
                    const int pxi = tx * 16 + x, pyi = ty * 16 + y;
                    walked[y * 16 + x] = (pxi < W_img && pyi < H) ? (int)n_walked[(size_t)pyi * W_img + pxi] : 0;
                    if (walked[y * 16 + x] > maxwalk) maxwalk = walked[y * 16 + x];
                }
            int tot[256];
            memset(tot, 0, sizeof(tot));
            int pc[NPOL][256], pmax[NPOL][4], plast[NPOL][4];   /* policy state: per-pixel counts, per-quadrant max, chunk of the last flush */
            memset(pc, 0, sizeof(pc)); memset(pmax, 0, sizeof(pmax)); memset(plast, 0, sizeof(plast));
            int qpairs[4] = {0, 0, 0, 0}, qchunk = -1;
            static int gcnt[4096][256];   /* per group per pixel passing-and-walked count */
            const int ngroups = (n + G - 1) / G;
            if (ngroups > 4096) continue;
            for (int g = 0; g < ngroups; ++g) memset(gcnt[g], 0, sizeof(gcnt[g]));
            for (int g = 0; g < ngroups; ++g) {
                const int gb = g * G, ge = gb + G < n ? gb + G : n;
                const int evaluated = gb < maxwalk;
                double gp = 0;
                for (int j = gb; j < ge; ++j) {
                    const uint32_t id = point_list[r0 + j];
                    if (G == 64 && kU == 1) {   /* (policies are evaluated once: in the G = 64, kU = 1 run) */
                        const int ch = j / 64;
                        if (ch != qchunk) {
                            for (int q = 0; q < 4; ++q) { if (qchunk >= 0 && j <= maxwalk + 63) { out->quad_iters += (qpairs[q] + 63) / 64; out->quad_chunks += 1; } qpairs[q] = 0; }
                            qchunk = ch;
                            for (int p = 0; p < NPOL; ++p)
                                for (int q = 0; q < 4; ++q)
                                    if (pmax[p][q] > 0 && ch - plast[p][q] >= polF[p]) {   /* forced flush: the record ring wraps */
                                        out->polslots[p] += 64.0 * pmax[p][q]; out->polflush[p] += 1; pmax[p][q] = 0; plast[p][q] = ch;
                                        for (int l = 0; l < 64; ++l) pc[p][((q >> 1) * 8 + (l >> 3)) * 16 + (q & 1) * 8 + (l & 7)] = 0;
                                    }
                        }
                    }
                    const float *Tu = trans + 9 * id, *Tv = Tu + 3, *Tw = Tu + 6;
                    for (int y = 0; y < 16; ++y) {
                        const float pyf = (float)(ty * 16 + y);
                        if (!(fabsf(pyf - cy[id]) <= ry[id])) continue;
                        for (int x = 0; x < 16; ++x) {
                            const float pxf = (float)(tx * 16 + x);
                            if (!(fabsf(pxf - cx[id]) <= rx[id])) continue;
                            if (tx * 16 + x >= W_img || ty * 16 + y >= H) continue;
                            const float kx = pxf * Tw[0] - Tu[0], ky = pxf * Tw[1] - Tu[1], kz = pxf * Tw[2] - Tu[2];
                            const float lx_ = pyf * Tw[0] - Tv[0], ly_ = pyf * Tw[1] - Tv[1], lz_ = pyf * Tw[2] - Tv[2];
                            const float p0 = ky * lz_ - kz * ly_, p1 = kz * lx_ - kx * lz_, p2 = kx * ly_ - ky * lx_;
                            int pass = 0;
                            if (p2 != 0.0f) {
                                const float sx = p0 / p2, sy = p1 / p2, rho3d = sx * sx + sy * sy;
                                const float dx = cx[id] - pxf, dy = cy[id] - pyf, rho2d = 2.0f * (dx * dx + dy * dy);
                                const float rho = fminf(rho3d, rho2d);
                                const float depth = (rho3d <= rho2d) ? (sx * Tw[0] + sy * Tw[1]) + Tw[2] : Tw[2];
                                const float alpha = fminf(0.99f, opa[id] * expf(-0.5f * rho));
                                pass = depth >= 0.2f && !(-0.5f * rho > 0.0f) && alpha >= 1.0f / 255.0f;
                            }
                            const int w = j < walked[y * 16 + x];
                            out->cand_all += 1; out->pass_all += pass;
                            out->cand_walk += w; out->pass_walk += pass && w;
                            if (evaluated) { out->candA += 1; out->passA += pass; gp += 1; }
                            if (pass && w) { gcnt[g][y * 16 + x]++; tot[y * 16 + x]++; }
                            if (G == 64 && kU == 1) {
                                const int q = (y >> 3) * 2 + (x >> 3);
                                if (evaluated) qpairs[q]++;
                                if (pass && w)
                                    for (int p = 0; p < NPOL; ++p) {
                                        if (pc[p][y * 16 + x] == polK[p]) {   /* full: flush the quadrant now, then append */
                                            out->polslots[p] += 64.0 * pmax[p][q]; out->polflush[p] += 1; pmax[p][q] = 0; plast[p][q] = j / 64;
                                            for (int l = 0; l < 64; ++l) pc[p][((q >> 1) * 8 + (l >> 3)) * 16 + (q & 1) * 8 + (l & 7)] = 0;
                                        }
                                        if (++pc[p][y * 16 + x] > pmax[p][q]) pmax[p][q] = pc[p][y * 16 + x];
                                    }
                            }
                        }
                    }
                }
                if (evaluated) {
                    out->groups += 1; out->itersA += ceil(gp / 64.0); out->entriesA += ge - gb;
                    if (gp > out->max_group_pairs) out->max_group_pairs = gp;
                }
            }
            if (G == 64 && kU == 1) {
                for (int q = 0; q < 4; ++q) { if (qchunk >= 0) { out->quad_iters += (qpairs[q] + 63) / 64; out->quad_chunks += 1; } }
                for (int p = 0; p < NPOL; ++p)
                    for (int q = 0; q < 4; ++q)
                        if (pmax[p][q] > 0) { out->polslots[p] += 64.0 * pmax[p][q]; out->polflush[p] += 1; }
            }
            /* phase B lane slots */
            int ord[256];
            for (int g = 0; g < ngroups; ++g) {
                if (!(g * G < maxwalk)) break;
                for (int q = 0; q < 4; ++q) {
                    int mx = 0;
                    for (int l = 0; l < 64; ++l) {
                        const int x = (q & 1) * 8 + (l & 7), y = (q >> 1) * 8 + (l >> 3);
                        if (gcnt[g][y * 16 + x] > mx) mx = gcnt[g][y * 16 + x];
                    }
                    out->slotsB_quad += 64.0 * ((mx + kU - 1) / kU * kU);
                }
                memcpy(ord, gcnt[g], sizeof(ord));
                qsort(ord, 256, sizeof(int), cmp_desc);
                for (int q = 0; q < 4; ++q) out->slotsB_sorted += 64.0 * ((ord[q * 64] + kU - 1) / kU * kU);
                /* regrouped once per tile by total length: wave q = ranks 64q.. of tot; per group its max */
            }
            {
                int idx[256];
                for (int i = 0; i < 256; ++i) idx[i] = i;
                for (int a = 1; a < 256; ++a) { int v = idx[a], b = a - 1; while (b >= 0 && tot[idx[b]] < tot[v]) { idx[b + 1] = idx[b]; --b; } idx[b + 1] = v; }
                for (int g = 0; g < ngroups; ++g) {
                    if (!(g * G < maxwalk)) break;
                    for (int q = 0; q < 4; ++q) {
                        int mx = 0;
                        for (int l = 0; l < 64; ++l) if (gcnt[g][idx[q * 64 + l]] > mx) mx = gcnt[g][idx[q * 64 + l]];
                        out->slotsB_sorted_tile += 64.0 * ((mx + kU - 1) / kU * kU);
                    }
                }
            }
        }
}
